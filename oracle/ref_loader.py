"""TEST INFRASTRUCTURE ONLY -- loads the reference's UNMODIFIED hot-path files from /root/reference.

`import GPy` fails here because the third-party `paramz` is absent (reference `setup.py:146`), so
this loader registers empty stub *packages* whose ``__path__`` points into ``/root/reference/GPy``
(skipping the package ``__init__``s that import everything) and puts the test-only paramz stand-in
(``oracle/paramz_stub``) on ``sys.path``.  The modules then imported are the reference's own files:

* ``GPy/util/linalg.py``, ``GPy/util/diag.py``                       (jitchol/pdinv/dpotrs/tdot ...)
* ``GPy/kern/src/stationary.py``, ``rbf.py``, ``kern.py``, ``kernel_slice_operations.py``
* ``GPy/inference/latent_function_inference/exact_gaussian_inference.py``, ``posterior.py``
* ``GPy/likelihoods/gaussian.py``

Nothing is copied; nothing under /root/reference is written (``sys.dont_write_bytecode``).
Only ``tests/`` (when /root/reference exists) and ``oracle/make_golden.py`` may use this module.
It is NOT available on the GPU box (no /root/reference there); the committed ``tests/golden/*.npz``
produced by ``oracle/make_golden.py`` carry the reference's outputs instead.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("GPY_REFERENCE_ROOT", "/root/reference")
_REF = os.path.join(REF_ROOT, "GPy")
_HERE = os.path.dirname(os.path.abspath(__file__))
_loaded = None


def available():
    return os.path.isdir(_REF)


def _stub(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def load(cython_build_dir=None):
    """Returns a namespace with the reference's classes (RBF, Matern52, ..., ExactGaussianInference,
    Gaussian, linalg).  ``cython_build_dir``: optional directory holding out-of-tree builds of the
    reference's ``stationary_cython`` / ``linalg_cython`` (see ``oracle/build_ref_cython.py``)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % _REF)
    sys.dont_write_bytecode = True
    stub_dir = os.path.join(_HERE, "paramz_stub")
    if stub_dir not in sys.path:
        sys.path.insert(0, stub_dir)

    G = _stub("GPy", _REF)
    U = _stub("GPy.util", _REF + "/util")
    C = _stub("GPy.core", _REF + "/core")
    _stub("GPy.kern", _REF + "/kern")
    KS = _stub("GPy.kern.src", _REF + "/kern/src")
    _stub("GPy.inference", _REF + "/inference")
    LFI = _stub("GPy.inference.latent_function_inference", _REF + "/inference/latent_function_inference")
    _stub("GPy.likelihoods", _REF + "/likelihoods")

    class LatentFunctionInference(object):
        def on_optimization_start(self):
            pass

        def on_optimization_end(self):
            pass

        def _save_to_input_dict(self):
            return {}

    LFI.LatentFunctionInference = LatentFunctionInference

    if cython_build_dir is None:
        cand = os.path.join(_HERE, "_ref", "cython")
        if os.path.isdir(cand):
            cython_build_dir = cand
    if cython_build_dir:
        U.__path__.append(os.path.join(cython_build_dir, "GPy", "util"))
        KS.__path__.append(os.path.join(cython_build_dir, "GPy", "kern", "src"))

    for m in ("config", "linalg", "diag"):
        setattr(U, m, importlib.import_module("GPy.util." + m))
    G.util = U
    P = importlib.import_module("GPy.core.parameterization")
    C.parameterization = P
    C.Param = P.Param
    C.Parameterized = P.Parameterized
    G.core = C

    st = importlib.import_module("GPy.kern.src.stationary")
    rbf = importlib.import_module("GPy.kern.src.rbf")
    egi = importlib.import_module("GPy.inference.latent_function_inference.exact_gaussian_inference")
    post = importlib.import_module("GPy.inference.latent_function_inference.posterior")
    gauss = importlib.import_module("GPy.likelihoods.gaussian")

    ns = types.SimpleNamespace(
        RBF=rbf.RBF, Matern52=st.Matern52, Matern32=st.Matern32, Exponential=st.Exponential,
        ExpQuad=st.ExpQuad, Stationary=st.Stationary,
        ExactGaussianInference=egi.ExactGaussianInference, PosteriorExact=post.PosteriorExact,
        Gaussian=gauss.Gaussian, linalg=U.linalg, diag=U.diag,
        use_stationary_cython=st.use_stationary_cython,
        use_linalg_cython=U.linalg.use_linalg_cython,
        stationary=st,
    )
    _loaded = ns
    return ns


def load_sum_kernels(ns):
    """Adds the reference's `Add`, `Prod`, `White`, `Bias` (GPy/kern/src/add.py, prod.py, static.py) to the namespace.  `Add.__init__` does
    `from .. import RBF, Linear, Bias, White` (add.py:34), so those names are put on the stub `GPy.kern` package."""
    if hasattr(ns, "Add"):
        return ns
    st = importlib.import_module("GPy.kern.src.static")
    K = sys.modules["GPy.kern"]
    K.RBF, K.White, K.Bias = ns.RBF, st.White, st.Bias
    try:
        K.Linear = importlib.import_module("GPy.kern.src.linear").Linear
    except Exception:
        K.Linear = type("Linear", (), {})                 # only used in isinstance checks of the psi-statistics path
    add = importlib.import_module("GPy.kern.src.add")
    ns.Add, ns.White, ns.Bias = add.Add, st.White, st.Bias
    ns.Prod = importlib.import_module("GPy.kern.src.prod").Prod
    return ns


KERNELS = {"rbf": "RBF", "matern52": "Matern52", "matern32": "Matern32", "exponential": "Exponential"}


def make_kernel(ns, kind, input_dim, variance, lengthscale, ARD):
    cls = getattr(ns, KERNELS[kind])
    return cls(input_dim, variance=variance, lengthscale=lengthscale, ARD=ARD)


def run_iteration(ns, kind, X, Y, variance, lengthscale, ARD, noise):
    """One `GP.parameters_changed` (reference core/gp.py:278-280) through the reference's own code."""
    import numpy as np
    k = make_kernel(ns, kind, X.shape[1], variance, lengthscale, ARD)
    lik = ns.Gaussian(variance=noise)
    post, lml, gd = ns.ExactGaussianInference().inference(k, X, lik, Y)
    lik.update_gradients(gd["dL_dthetaL"])
    k.update_gradients_full(gd["dL_dK"], X)
    return dict(
        lml=float(lml), alpha=np.asarray(post.woodbury_vector), L=np.asarray(post.woodbury_chol),
        dL_dK=np.asarray(gd["dL_dK"]), K=np.asarray(post._K),
        dvar=np.asarray(k.variance.gradient, dtype=float).copy(),
        dlen=np.asarray(k.lengthscale.gradient, dtype=float).copy(),
        dnoise=np.asarray(lik.variance.gradient, dtype=float).copy(),
    )
