"""ctypes binding of libmi355gp.so (C-ABI declared in include/mi355gp.h).

No PyTorch, no CPU fallback: if the library is missing it is built with hipcc (gfx950) on first use, and
every compute entry point raises when no MI355X is visible.  The CPU oracle under `oracle/` is never
imported from here.
"""
import ctypes
import os
import subprocess

import numpy as np
from numpy.ctypeslib import ndpointer

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355GP_LIB") or os.path.join(_HERE, "libmi355gp.so")
CSRC = os.path.join(_HERE, "csrc")

KIND_IDS = {"rbf": 0, "matern52": 1, "matern32": 2, "exponential": 3, "white": 4, "bias": 5}


class Part(ctypes.Structure):
    """`mi355gp_part` of include/mi355gp.h: one part of a sum-of-products kernel expression."""
    _fields_ = [("kind", ctypes.c_int), ("ard", ctypes.c_int), ("n_active", ctypes.c_int),
                ("active_dims", ctypes.POINTER(ctypes.c_int)), ("theta", ctypes.POINTER(ctypes.c_double)),
                ("term", ctypes.c_int)]


def make_parts(specs):
    """specs: [(kind_name, ARD, theta array, active_dims array or None[, term])] -> (ctypes array, keep-alive list,
    n_theta).  term (optional, default 0): parts with the same non-zero term id are multiplied (product kernels)."""
    arr = (Part * len(specs))()
    keep, ntheta = [], 0
    for i, spec in enumerate(specs):
        kind, ARD, theta, dims = spec[:4]
        arr[i].term = int(spec[4]) if len(spec) > 4 else 0
        th = np.ascontiguousarray(theta, dtype=np.float64)
        keep.append(th)
        arr[i].kind = KIND_IDS[kind]
        arr[i].ard = int(bool(ARD))
        arr[i].theta = th.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        if dims is None:
            arr[i].n_active = 0
            arr[i].active_dims = None
        else:
            d = np.ascontiguousarray(dims, dtype=np.int32)
            keep.append(d)
            arr[i].n_active = d.size
            arr[i].active_dims = d.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        ntheta += th.size
    return arr, keep, ntheta
FETCH_L, FETCH_KINV, FETCH_DLDK, FETCH_K = 0, 1, 2, 3
OUT_LML, OUT_LOGDET, OUT_DATAFIT, OUT_DNOISE, OUT_TRKINV, NUM_OUT = 0, 1, 2, 3, 4, 8
STAGE_NAMES = ("kbuild", "potrf", "trtri", "lauum", "solve", "grad", "total")
NUM_T = 8
PROFILE_FAMILIES = ("update_nt", "trtri", "lauum", "diag128", "trsm128", "update_nt64", "potrf_persist", "trtri_early")

_lib = None


class MI355GPError(RuntimeError):
    pass


DIAG_LIB_PATH = os.path.join(_HERE, "libmi355gp_diag.so")


def build(force=False):
    """Compile every HIP translation unit for gfx950 and link, in-tree (make -C gpy_amd/csrc): libmi355gp.so, the product, and
    libmi355gp_diag.so, the same sources with -DMI355GP_DIAG (schedule overrides, fault injectors and bounding experiments that
    the product build compiles out; loaded only by the tests / tools that ask for it through MI355GP_LIB)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(_HERE, "..", "include", "mi355gp.h"))
    srcs.append(os.path.join(_HERE, "..", "include", "mi355gp_debug.h"))
    product = os.path.join(_HERE, "libmi355gp.so")
    if not force and os.path.exists(product) and os.path.exists(DIAG_LIB_PATH):
        t = min(os.path.getmtime(product), os.path.getmtime(DIAG_LIB_PATH))
        if all(os.path.getmtime(s) <= t for s in srcs if os.path.exists(s)):
            return product
    r = subprocess.run(["make", "-C", CSRC, "-j8", "all", "diag"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(product) or not os.path.exists(DIAG_LIB_PATH):
        raise MI355GPError("building libmi355gp.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return product


_dp = ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_c_dp = ctypes.POINTER(ctypes.c_double)


def _opt(a):
    """optional output array -> pointer or NULL"""
    return None if a is None else a.ctypes.data_as(_c_dp)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = ctypes.CDLL(LIB_PATH)
    i64, ci, cd, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p
    L.mi355gp_last_error.restype = ctypes.c_char_p
    L.mi355gp_version.restype = ctypes.c_char_p
    L.mi355gp_device_count.argtypes = [ctypes.POINTER(ci)]
    L.mi355gp_device_synchronize.argtypes = [ci]
    L.mi355gp_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.mi355gp_destroy.argtypes = [vp]
    L.mi355gp_set_data.argtypes = [vp, _dp, i64, ci, _dp, ci]
    L.mi355gp_set_targets.argtypes = [vp, _dp, ci]
    L.mi355gp_kern_K.argtypes = [ci, ci, ci, _dp, _dp, i64, _c_dp, i64, ci, _dp]
    L.mi355gp_kern_Kdiag.argtypes = [ci, _dp, i64, _dp]
    L.mi355gp_update_gradients_full.argtypes = [ci, ci, ci, _dp, _dp, _dp, i64, _c_dp, i64, ci, _dp]
    L.mi355gp_gradients_X.argtypes = [ci, ci, ci, _dp, _dp, _dp, i64, _c_dp, i64, ci, _dp]
    L.mi355gp_exact_inference.argtypes = [vp, ci, ci, _dp, _dp, i64, cd, cd, _dp, _c_dp, _c_dp, _c_dp, _c_dp]
    L.mi355gp_exact_inference_sum.argtypes = [vp, ci, ctypes.POINTER(Part), _dp, i64, cd, cd, _dp, _c_dp, _c_dp, _c_dp,
                                              _c_dp]
    L.mi355gp_exact_studentt_sum.argtypes = [vp, ci, ctypes.POINTER(Part), cd, cd, cd, _dp, _c_dp, _c_dp, _c_dp]
    L.mi355gp_covariance_between_points.argtypes = [vp, ci, ctypes.POINTER(Part), _dp, i64, _dp, i64, _dp]
    L.mi355gp_predictive_gradients_sum.argtypes = [vp, ci, ctypes.POINTER(Part), _dp, i64, _c_dp, _c_dp]
    L.mi355gp_predict_sum.argtypes = [vp, ci, ctypes.POINTER(Part), _dp, i64, _c_dp, _c_dp, ci]
    L.mi355gp_inference_given_K.argtypes = [vp, _dp, _dp, i64, cd, cd, _dp, _c_dp, _c_dp, _c_dp]
    L.mi355gp_fetch.argtypes = [vp, ci, _dp, ci]
    L.mi355gp_predict.argtypes = [vp, ci, ci, _dp, _dp, i64, _c_dp, _c_dp, ci]
    L.mi355gp_potrf.argtypes = [ci, _dp, i64, _c_dp]
    L.mi355gp_pdinv.argtypes = [ci, _dp, i64, _c_dp, _c_dp, _c_dp, _c_dp]
    L.mi355gp_pdinv_full.argtypes = [ci, _dp, i64, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]
    L.mi355gp_set_option.argtypes = [vp, ci, ci]
    L.mi355gp_get_option.argtypes = [vp, ci, ctypes.POINTER(ci)]
    L.mi355gp_sparse_get_profile.argtypes = [vp, _dp]
    L.mi355gp_dbg_comm_selftest.argtypes = [ci, ctypes.c_char_p, ci, ci, ci, ci, i64, ci, _dp]
    L.mi355gp_dbg_comm_selftest.restype = ci
    L.mi355gp_bench_factor.argtypes = [ci, i64, ci, _c_dp, _c_dp, _c_dp]
    L.mi355gp_bench_factor.restype = ci
    L.mi355gp_get_profile.argtypes = [vp, _dp, _dp, ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
    L.mi355gp_grid_unique_id.argtypes = [ctypes.c_char_p]
    L.mi355gp_grid_create.argtypes = [ci, ci, ci, ci, ci, ci, ctypes.c_char_p, ctypes.POINTER(vp)]
    L.mi355gp_grid_destroy.argtypes = [vp]
    L.mi355gp_grid_set_data.argtypes = [vp, _dp, i64, ci, _dp, ci]
    L.mi355gp_grid_exact_inference.argtypes = [vp, ci, ci, _dp, _dp, i64, cd, cd, _dp, _c_dp, _c_dp, _c_dp, _c_dp]
    L.mi355gp_grid_fetch.argtypes = [vp, ci, _dp]
    L.mi355gp_grid_set_option.argtypes = [vp, ci, ci]
    L.mi355gp_grid_coll_log.argtypes = [vp, ci, _dp]
    L.mi355gp_grid_coll_log.restype = ci
    L.mi355gp_dbg_grid_multi.argtypes = [ci, ci, ci, ci, _dp]
    L.mi355gp_dbg_update_nt.argtypes = [ci, ci, ctypes.POINTER(ci), ci, ci, _dp]
    L.mi355gp_dbg_update_rect.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ci, ci, _dp]
    L.mi355gp_grid_get_option.argtypes = [vp, ci, ctypes.POINTER(ci)]
    L.mi355gp_sparse_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.mi355gp_sparse_destroy.argtypes = [vp]
    L.mi355gp_sparse_set_data.argtypes = [vp, _dp, i64, ci, _dp, ci]
    L.mi355gp_vardtc_inference.argtypes = [vp, ci, ci, _dp, _dp, i64, cd, cd, _dp, _c_dp, _c_dp, _c_dp, _c_dp]
    L.mi355gp_sparse_fetch.argtypes = [vp, ci, _dp]
    L.mi355gp_vardtc_inference_sum.argtypes = [vp, ci, ctypes.POINTER(Part), _dp, i64, _dp, i64, cd, _dp, _c_dp, _c_dp, _c_dp,
                                               _c_dp, _c_dp, _c_dp]
    L.mi355gp_sparse_predict.argtypes = [vp, ci, ctypes.POINTER(Part), _dp, i64, _c_dp, _c_dp, ci]
    L.mi355gp_sparse_fetch_dLdKnm.argtypes = [vp, i64, i64, _dp]
    L.mi355gp_sparse_attach_loopback.argtypes = [vp, ci, ci, ci]
    L.mi355gp_sparse_attach_comm.argtypes = [vp, ci, ci, ctypes.c_char_p]
    L.mi355gp_dbg_mfma.argtypes = [ci, _dp, _dp, _dp]
    L.mi355gp_dbg_gemm.argtypes = [ci, ci, ci, i64, i64, i64, _dp, _dp, _dp, cd, cd, ci, _c_dp]
    L.mi355gp_dbg_peaks.argtypes = [ci, _dp]
    L.mi355gp_dbg_pipe_share.argtypes = [ci, _dp]
    L.mi355gp_dbg_graph_factor.argtypes = [ci, i64, ci, _dp]
    L.mi355gp_dbg_gemm_clock.argtypes = [_c_dp, _c_dp]
    L.mi355gp_dbg_mask_probe.argtypes = [ci, ci, ci, _dp]
    L.mi355gp_dbg_lauum_plan.argtypes = [ci, ctypes.POINTER(ci), ci, ctypes.POINTER(ci)]
    L.mi355gp_dbg_lauum_plan.restype = ci
    L.mi355gp_dbg_persist_owners.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.mi355gp_dbg_persist_owners.restype = ci
    L.mi355gp_dbg_persist.argtypes = [ci, i64, ci, ci, _dp]
    L.mi355gp_dbg_ipc_selftest.argtypes = [ctypes.c_char_p, ci, ci, ci, ci, i64, _dp]
    for name in ("device_count", "create", "destroy", "set_data", "set_targets", "kern_K", "kern_Kdiag",
                 "update_gradients_full", "exact_inference", "inference_given_K", "fetch", "predict", "potrf",
                 "pdinv", "dbg_mfma", "dbg_gemm", "dbg_peaks", "set_option", "get_profile", "grid_unique_id",
                 "grid_create", "grid_destroy", "grid_set_data", "grid_exact_inference", "grid_fetch", "grid_set_option",
                 "grid_get_option", "sparse_create",
                 "sparse_destroy", "sparse_set_data", "vardtc_inference", "sparse_fetch", "gradients_X", "sparse_attach_comm", "exact_inference_sum",
                 "predict_sum", "dbg_gemm_clock", "covariance_between_points", "exact_studentt_sum", "dbg_mask_probe",
                 "vardtc_inference_sum", "sparse_predict", "sparse_fetch_dLdKnm", "sparse_attach_loopback",
                 "predictive_gradients_sum", "dbg_pipe_share", "pdinv_full", "dbg_graph_factor", "get_option",
                 "sparse_get_profile", "dbg_persist", "dbg_grid_multi", "dbg_update_nt", "dbg_update_rect",
                 "dbg_ipc_selftest"):
        getattr(L, "mi355gp_" + name).restype = ci
    _lib = L
    return L


EXPORTED = ("mi355gp_last_error", "mi355gp_version", "mi355gp_device_count", "mi355gp_device_synchronize", "mi355gp_create", "mi355gp_destroy",
            "mi355gp_set_data", "mi355gp_set_targets", "mi355gp_kern_K", "mi355gp_kern_Kdiag",
            "mi355gp_update_gradients_full", "mi355gp_exact_inference", "mi355gp_inference_given_K",
            "mi355gp_fetch", "mi355gp_predict", "mi355gp_potrf", "mi355gp_pdinv", "mi355gp_bench_factor",
            "mi355gp_set_option", "mi355gp_get_profile", "mi355gp_grid_unique_id", "mi355gp_grid_create",
            "mi355gp_grid_destroy", "mi355gp_grid_set_data", "mi355gp_grid_exact_inference", "mi355gp_grid_fetch",
            "mi355gp_grid_set_option", "mi355gp_grid_get_option",
            "mi355gp_sparse_create", "mi355gp_sparse_destroy", "mi355gp_sparse_set_data", "mi355gp_vardtc_inference",
            "mi355gp_sparse_fetch", "mi355gp_gradients_X", "mi355gp_sparse_attach_comm",
            "mi355gp_exact_inference_sum", "mi355gp_predict_sum", "mi355gp_dbg_gemm_clock",
            "mi355gp_covariance_between_points", "mi355gp_exact_studentt_sum", "mi355gp_vardtc_inference_sum",
            "mi355gp_sparse_predict", "mi355gp_sparse_fetch_dLdKnm", "mi355gp_sparse_attach_loopback",
            "mi355gp_predictive_gradients_sum", "mi355gp_dbg_pipe_share", "mi355gp_pdinv_full", "mi355gp_dbg_graph_factor",
            "mi355gp_dbg_mfma", "mi355gp_dbg_gemm", "mi355gp_dbg_peaks", "mi355gp_dbg_mask_probe", "mi355gp_get_option",
            "mi355gp_sparse_get_profile", "mi355gp_dbg_persist", "mi355gp_dbg_grid_multi", "mi355gp_dbg_update_nt", "mi355gp_dbg_update_rect",
            "mi355gp_dbg_ipc_selftest", "mi355gp_grid_coll_log", "mi355gp_dbg_lauum_plan", "mi355gp_dbg_persist_owners")


# mi355gp_set_option / mi355gp_get_option ids (include/mi355gp.h, MI355GP_OPT_*)
OPTIONS = {"profile": 0, "lookahead": 1, "tri_overlap": 2, "tri_min_nt": 3, "tri_h": 4, "tri_wgs": 5, "tri_half": 6,
           "part1_on_panel": 7, "nbo": 8, "solve_overlap": 9, "diag_excl_first": 10, "graph": 11, "persist": 12, "agg2": 13,
           "persist_test": 14, "persist_aborts": 15, "persist_skip": 16, "persist_sched": 17}


def last_error():
    return lib().mi355gp_last_error().decode()


def check(rc, what):
    """negative rc -> exception; positive rc (LAPACK info) is returned to the caller"""
    if rc < 0:
        raise MI355GPError("%s failed (rc=%d): %s" % (what, rc, last_error()))
    return rc


def device_count():
    n = ctypes.c_int(0)
    lib().mi355gp_device_count(ctypes.byref(n))
    return n.value


def device_synchronize(device=0):
    """hipDeviceSynchronize through the library (no second HIP runtime user in the process)."""
    require_device(device)
    check(lib().mi355gp_device_synchronize(int(device)), "mi355gp_device_synchronize")


def require_device(device=0):
    n = device_count()
    if device >= n:
        raise MI355GPError("no MI355X visible to HIP (device %d requested, %d present): the gpy_amd backend has "
                           "no CPU fallback" % (device, n))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def theta_vec(variance, lengthscale, ARD, D):
    ls = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64)).ravel()
    if ARD:
        if ls.size == 1:
            ls = np.full(D, ls[0])
        assert ls.size == D, "ARD kernel needs one lengthscale per input dimension"
    else:
        assert ls.size == 1, "isotropic kernel has a single lengthscale"
    return np.concatenate([[float(np.asarray(variance).ravel()[0])], ls])


class Context(object):
    """One device context = one uploaded data set (X, R = Y - mean) with its N x N buffers in HBM."""

    def __init__(self, device=0):
        require_device(device)
        self._h = ctypes.c_void_p()
        check(lib().mi355gp_create(device, ctypes.byref(self._h)), "mi355gp_create")
        self.device = device
        self.N = self.D = self.Dy = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mi355gp_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, X, R):
        X, R = f64(X), f64(R)
        assert X.ndim == 2 and R.ndim == 2 and X.shape[0] == R.shape[0]
        self.N, self.D = X.shape
        self.Dy = R.shape[1]
        check(lib().mi355gp_set_data(self._h, X, self.N, self.D, R, self.Dy), "mi355gp_set_data")

    def set_targets(self, R):
        R = f64(R)
        check(lib().mi355gp_set_targets(self._h, R, R.shape[1]), "mi355gp_set_targets")

    def exact_inference(self, kind, ARD, theta, noise, jitter=1e-8, extra_jitter=0.0, want_alpha=True,
                        want_diag=False, want_stage_ms=False):
        """Returns (info, dict).  info > 0: not positive definite (the caller runs GPy's jitter ladder)."""
        theta = f64(theta)
        noise = f64(np.atleast_1d(noise))
        out = np.zeros(NUM_OUT)
        alpha = np.empty((self.N, self.Dy)) if want_alpha else None
        dtheta = np.zeros(theta.size)
        diag = np.empty(self.N) if want_diag else None
        ms = np.zeros(NUM_T) if want_stage_ms else None
        rc = check(lib().mi355gp_exact_inference(self._h, KIND_IDS[kind], int(bool(ARD)), theta, noise, noise.size,
                                                 jitter, extra_jitter, out, _opt(alpha), _opt(dtheta), _opt(diag),
                                                 _opt(ms)), "mi355gp_exact_inference")
        res = dict(lml=out[OUT_LML], logdet=out[OUT_LOGDET], datafit=out[OUT_DATAFIT], dnoise=out[OUT_DNOISE],
                   trKinv=out[OUT_TRKINV], alpha=alpha, dtheta=dtheta, diag_dL_dK=diag)
        if ms is not None:
            res["stage_ms"] = dict(zip(STAGE_NAMES, ms[:len(STAGE_NAMES)]))
        return rc, res

    def exact_inference_sum(self, specs, noise, jitter=1e-8, extra_jitter=0.0, want_alpha=True, want_diag=False,
                            want_stage_ms=False):
        """Sum kernel: specs = [(kind, ARD, theta, active_dims or None)]; dtheta is the concatenation over the parts."""
        arr, keep, ntheta = make_parts(specs)
        noise = f64(np.atleast_1d(noise))
        out = np.zeros(NUM_OUT)
        alpha = np.empty((self.N, self.Dy)) if want_alpha else None
        dtheta = np.zeros(ntheta)
        diag = np.empty(self.N) if want_diag else None
        ms = np.zeros(NUM_T) if want_stage_ms else None
        rc = check(lib().mi355gp_exact_inference_sum(self._h, len(specs), arr, noise, noise.size, jitter, extra_jitter,
                                                     out, _opt(alpha), _opt(dtheta), _opt(diag), _opt(ms)),
                   "mi355gp_exact_inference_sum")
        res = dict(lml=out[OUT_LML], logdet=out[OUT_LOGDET], datafit=out[OUT_DATAFIT], dnoise=out[OUT_DNOISE],
                   trKinv=out[OUT_TRKINV], alpha=alpha, dtheta=dtheta, diag_dL_dK=diag)
        if ms is not None:
            res["stage_ms"] = dict(zip(STAGE_NAMES, ms[:len(STAGE_NAMES)]))
        return rc, res

    def predict_sum(self, specs, Xnew, full_cov=False, want_var=True):
        arr, keep, _ = make_parts(specs)
        Xnew = f64(Xnew)
        M = Xnew.shape[0]
        assert Xnew.shape[1] == self.D
        mu = np.empty((M, self.Dy))
        var = (np.empty((M, M)) if full_cov else np.empty(M)) if want_var else None
        check(lib().mi355gp_predict_sum(self._h, len(specs), arr, Xnew, M, mu.ctypes.data_as(_c_dp), _opt(var),
                                        int(bool(full_cov))), "mi355gp_predict_sum")
        if var is not None and not full_cov:
            var = var[:, None]
        return mu, var

    def exact_studentt_sum(self, specs, nu, jitter=1e-8, extra_jitter=0.0):
        """Student-t process: (info, dict(lml, logdet, beta, scale, alpha, dtheta))"""
        arr, keep, ntheta = make_parts(specs)
        out = np.zeros(NUM_OUT)
        alpha = np.empty((self.N, self.Dy))
        dtheta = np.zeros(ntheta)
        rc = check(lib().mi355gp_exact_studentt_sum(self._h, len(specs), arr, float(nu), jitter, extra_jitter, out,
                                                    _opt(alpha), _opt(dtheta), None), "mi355gp_exact_studentt_sum")
        return rc, dict(lml=out[OUT_LML], logdet=out[OUT_LOGDET], beta=out[OUT_DATAFIT], scale=out[5], alpha=alpha,
                        dtheta=dtheta)

    def predictive_gradients(self, specs, Xnew, want_var=True):
        """(dmu (M x D x Dy), dvar (M x D)) of GP.predictive_gradients (core/gp.py:418-474), reduced on the device."""
        arr, keep, _ = make_parts(specs)
        Xnew = f64(Xnew)
        M = Xnew.shape[0]
        assert Xnew.shape[1] == self.D
        dmu = np.empty((M, self.D, self.Dy))
        dvar = np.empty((M, self.D)) if want_var else None
        check(lib().mi355gp_predictive_gradients_sum(self._h, len(specs), arr, Xnew, M, dmu.ctypes.data_as(_c_dp), _opt(dvar)),
              "mi355gp_predictive_gradients_sum")
        return dmu, dvar

    def covariance_between_points(self, specs, X1, X2):
        arr, keep, _ = make_parts(specs)
        X1, X2 = f64(X1), f64(X2)
        out = np.empty((X1.shape[0], X2.shape[0]))
        check(lib().mi355gp_covariance_between_points(self._h, len(specs), arr, X1, X1.shape[0], X2, X2.shape[0], out),
              "mi355gp_covariance_between_points")
        return out

    def inference_given_K(self, K, noise, jitter=1e-8, extra_jitter=0.0, want_diag=False, want_stage_ms=False):
        K = f64(K)
        assert K.shape == (self.N, self.N)
        noise = f64(np.atleast_1d(noise))
        out = np.zeros(NUM_OUT)
        alpha = np.empty((self.N, self.Dy))
        diag = np.empty(self.N) if want_diag else None
        ms = np.zeros(NUM_T) if want_stage_ms else None
        rc = check(lib().mi355gp_inference_given_K(self._h, K, noise, noise.size, jitter, extra_jitter, out,
                                                   _opt(alpha), _opt(diag), _opt(ms)), "mi355gp_inference_given_K")
        res = dict(lml=out[OUT_LML], logdet=out[OUT_LOGDET], datafit=out[OUT_DATAFIT], dnoise=out[OUT_DNOISE],
                   trKinv=out[OUT_TRKINV], alpha=alpha, diag_dL_dK=diag)
        if ms is not None:
            res["stage_ms"] = dict(zip(STAGE_NAMES, ms[:len(STAGE_NAMES)]))
        return rc, res

    def set_option(self, name, value):
        """Options of THIS context through the C-ABI (`mi355gp_set_option`; the MI355GP_* environment variables only give the
        process-wide defaults).  'profile': 0 = off, 1 = every kernel family, or a tuple of family names; 'lookahead': 0/1;
        the schedule switches of OPTIONS take an int, -1 = back to the process default."""
        if name == "profile" and not isinstance(value, (int, bool)):
            value = sum(1 << PROFILE_FAMILIES.index(f) for f in value) << 1
        check(lib().mi355gp_set_option(self._h, OPTIONS[name], int(value)), "mi355gp_set_option")

    def get_option(self, name):
        v = ctypes.c_int(0)
        check(lib().mi355gp_get_option(self._h, OPTIONS[name], ctypes.byref(v)), "mi355gp_get_option")
        return int(v.value)

    def get_profile(self):
        """{family: (ms, algorithmic flops, launches)} of the last inference call made with option 'profile' on."""
        ms, fl, n = np.zeros(8), np.zeros(8), np.zeros(8, dtype=np.int32)
        check(lib().mi355gp_get_profile(self._h, ms, fl, n), "mi355gp_get_profile")
        return {k: (ms[i], fl[i], int(n[i])) for i, k in enumerate(PROFILE_FAMILIES)}

    def predict(self, kind, ARD, theta, Xnew, full_cov=False, want_var=True):
        """(mu (M x Dy), var (M x 1) or cov (M x M)) of the latent function at Xnew, computed on the device."""
        Xnew = f64(Xnew)
        M = Xnew.shape[0]
        assert Xnew.shape[1] == self.D
        mu = np.empty((M, self.Dy))
        var = (np.empty((M, M)) if full_cov else np.empty(M)) if want_var else None
        check(lib().mi355gp_predict(self._h, KIND_IDS[kind], int(bool(ARD)), f64(theta), Xnew, M,
                                    mu.ctypes.data_as(_c_dp), _opt(var), int(bool(full_cov))), "mi355gp_predict")
        if var is not None and not full_cov:
            var = var[:, None]
        return mu, var

    def fetch(self, which, fortran_order=False):
        out = np.empty((self.N, self.N))
        check(lib().mi355gp_fetch(self._h, which, out, int(fortran_order)), "mi355gp_fetch")
        if fortran_order:
            return out.T      # same memory viewed as an F-contiguous array
        return out


class SparseContext(object):
    """One device context of the sparse (VarDTC) path = one uploaded (X, Y); Z and theta change per call."""
    FETCH_DLDKMM, FETCH_WOODBURY_INV, FETCH_LM, FETCH_KMM, FETCH_PSI2 = 0, 1, 2, 3, 4

    def __init__(self, device=0):
        require_device(device)
        self._h = ctypes.c_void_p()
        check(lib().mi355gp_sparse_create(device, ctypes.byref(self._h)), "mi355gp_sparse_create")
        self.N = self.D = self.Dy = self.M = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mi355gp_sparse_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def attach_comm(self, rank, world, id_bytes):
        """Row-sharded multi-GPU mode: this rank will upload only its slice of (X, Y) (see gpy_amd.grid.shard_rows)."""
        check(lib().mi355gp_sparse_attach_comm(self._h, rank, world, id_bytes), "mi355gp_sparse_attach_comm")

    def set_data(self, X, Y):
        X, Y = f64(X), f64(Y)
        self.N, self.D = X.shape
        self.Dy = Y.shape[1]
        check(lib().mi355gp_sparse_set_data(self._h, X, self.N, self.D, Y, self.Dy), "mi355gp_sparse_set_data")

    def get_profile(self):
        """Launch timing of the two MFMA kernels of the LAST vardtc call (hipEvent pairs on the launching stream):
        {"gemm_T": (ms, algorithmic flops, launches), "gram_psi2": (...)} (`mi355gp_sparse_get_profile`)."""
        o = np.zeros(6)
        check(lib().mi355gp_sparse_get_profile(self._h, o), "mi355gp_sparse_get_profile")
        return {"gemm_T": (o[0], o[1], int(o[2])), "gram_psi2": (o[3], o[4], int(o[5]))}

    def vardtc(self, kind, ARD, theta, Z, noise_var, extra_jitter=0.0, want_stage_ms=False):
        """(info, dict(lml, dnoise, dtheta, dZ, woodbury_vector[, stage_ms]))"""
        theta, Z = f64(theta), f64(Z)
        self.M = Z.shape[0]
        assert Z.shape[1] == self.D
        out = np.zeros(NUM_OUT)
        dtheta = np.zeros(theta.size)
        dZ = np.zeros((self.M, self.D))
        wv = np.zeros((self.M, self.Dy))
        ms = np.zeros(4) if want_stage_ms else None
        rc = check(lib().mi355gp_vardtc_inference(self._h, KIND_IDS[kind], int(bool(ARD)), theta, Z, self.M,
                                                  float(noise_var), float(extra_jitter), out, _opt(dtheta), _opt(dZ),
                                                  _opt(wv), _opt(ms)), "mi355gp_vardtc_inference")
        res = dict(lml=out[0], dnoise=out[1], trA=out[2], data_fit=out[3], dtheta=dtheta, dZ=dZ, woodbury_vector=wv)
        if ms is not None:
            res["stage_ms"] = dict(pass1=ms[0], mxm=ms[1], pass2=ms[2], total=ms[3])
        return rc, res

    def fetch(self, which):
        out = np.empty((self.M, self.M))
        check(lib().mi355gp_sparse_fetch(self._h, which, out), "mi355gp_sparse_fetch")
        return out

    def attach_loopback(self, rank, world, group_key):
        """Row-sharded mode over the in-process loopback transport (one host thread per logical rank)."""
        check(lib().mi355gp_sparse_attach_loopback(self._h, rank, world, int(group_key)), "mi355gp_sparse_attach_loopback")

    def vardtc_sum(self, specs, Z, noise, extra_jitter=0.0, want_dL_dm=False, want_stage_ms=False):
        """Sum-of-parts kernel (specs as for `Context.exact_inference_sum`), scalar or per-point noise variances.
        (info, dict(lml, dnoise (scalar; per-point noise: the N-vector dL_dR, N x Dy for several output columns), dtheta
        (concatenated), dZ, woodbury_vector[, dL_dm, stage_ms]))"""
        arr, keep, ntheta = make_parts(specs)
        Z = f64(Z)
        self.M = Z.shape[0]
        assert Z.shape[1] == self.D
        noise = f64(np.atleast_1d(noise)).ravel()
        het = noise.size > 1
        out = np.zeros(NUM_OUT)
        dtheta = np.zeros(ntheta)
        dZ = np.zeros((self.M, self.D))
        wv = np.zeros((self.M, self.Dy))
        rows = np.zeros((self.N, self.Dy)) if het else None       # dL_dR per point and output column (var_dtc.py:240-256)
        dm = np.zeros((self.N, self.Dy)) if want_dL_dm else None
        ms = np.zeros(4) if want_stage_ms else None
        rc = check(lib().mi355gp_vardtc_inference_sum(self._h, len(specs), arr, Z, self.M, noise, noise.size,
                                                      float(extra_jitter), out, _opt(dtheta), _opt(dZ), _opt(wv), _opt(rows),
                                                      _opt(dm), _opt(ms)), "mi355gp_vardtc_inference_sum")
        res = dict(lml=out[0], dnoise=(rows[:, 0] if self.Dy == 1 else rows) if het else out[1], trA=out[2],
                   data_fit=out[3], dtheta=dtheta, dZ=dZ, woodbury_vector=wv, dL_dm=dm)
        if ms is not None:
            res["stage_ms"] = dict(pass1=ms[0], mxm=ms[1], pass2=ms[2], total=ms[3])
        return rc, res

    def predict(self, specs, Xnew, full_cov=False, want_var=True):
        """(mu (M* x Dy), var (M* x 1) or cov) of the sparse posterior of the last call, on the device."""
        arr, keep, _ = make_parts(specs)
        Xnew = f64(Xnew)
        Mn = Xnew.shape[0]
        assert Xnew.shape[1] == self.D
        mu = np.empty((Mn, self.Dy))
        var = (np.empty((Mn, Mn)) if full_cov else np.empty(Mn)) if want_var else None
        check(lib().mi355gp_sparse_predict(self._h, len(specs), arr, Xnew, Mn, mu.ctypes.data_as(_c_dp), _opt(var),
                                           int(bool(full_cov))), "mi355gp_sparse_predict")
        if var is not None and not full_cov:
            var = var[:, None]
        return mu, var

    def fetch_dL_dKnm(self, row0, nrows):
        out = np.empty((nrows, self.M))
        check(lib().mi355gp_sparse_fetch_dLdKnm(self._h, int(row0), int(nrows), out), "mi355gp_sparse_fetch_dLdKnm")
        return out


def kern_K(kind, ARD, theta, X, X2=None, device=0):
    require_device(device)
    X = f64(X)
    N, D = X.shape
    if X2 is None:
        M, p2 = N, None
    else:
        X2 = f64(X2)
        M, p2 = X2.shape[0], X2.ctypes.data_as(_c_dp)
        assert X2.shape[1] == D
    out = np.empty((N, M))
    check(lib().mi355gp_kern_K(device, KIND_IDS[kind], int(bool(ARD)), f64(theta), X, N, p2, M, D, out),
          "mi355gp_kern_K")
    return out


def kern_Kdiag(kind, theta, N):
    out = np.empty(N)
    check(lib().mi355gp_kern_Kdiag(KIND_IDS[kind], f64(theta), N, out), "mi355gp_kern_Kdiag")
    return out


def update_gradients_full(kind, ARD, theta, dL_dK, X, X2=None, device=0):
    require_device(device)
    X = f64(X)
    N, D = X.shape
    if X2 is None:
        M, p2 = N, None
    else:
        X2 = f64(X2)
        M, p2 = X2.shape[0], X2.ctypes.data_as(_c_dp)
    G = f64(dL_dK)
    assert G.shape == (N, M), "dL_dK must be N x M"
    theta = f64(theta)
    out = np.zeros(theta.size)
    check(lib().mi355gp_update_gradients_full(device, KIND_IDS[kind], int(bool(ARD)), theta, G, X, N, p2, M, D, out),
          "mi355gp_update_gradients_full")
    return out


def gradients_X(kind, ARD, theta, dL_dK, X, X2=None, device=0):
    """dL/dX (N x D) from dL_dK (N x M) (reference `Stationary.gradients_X`, stationary.py:245-252)."""
    require_device(device)
    X = f64(X)
    N, D = X.shape
    if X2 is None:
        M, p2 = N, None
    else:
        X2 = f64(X2)
        M, p2 = X2.shape[0], X2.ctypes.data_as(_c_dp)
    G = f64(dL_dK)
    assert G.shape == (N, M), "dL_dK must be N x M"
    out = np.zeros((N, D))
    check(lib().mi355gp_gradients_X(device, KIND_IDS[kind], int(bool(ARD)), f64(theta), G, X, N, p2, M, D, out),
          "mi355gp_gradients_X")
    return out


def potrf(A, device=0):
    """Lower Cholesky factor of A (dpotrf equivalent) -> (L, info, ms)."""
    require_device(device)
    L = f64(A).copy()
    ms = ctypes.c_double(0.0)
    info = check(lib().mi355gp_potrf(device, L, L.shape[0], ctypes.byref(ms)), "mi355gp_potrf")
    return L, info, ms.value


def pdinv(A, device=0):
    """(Ainv, L, logdet, info, ms): pdinv equivalent (GPy/util/linalg.py:193-214)."""
    require_device(device)
    A = f64(A)
    n = A.shape[0]
    Ai, L = np.empty((n, n)), np.empty((n, n))
    ld, ms = ctypes.c_double(0.0), ctypes.c_double(0.0)
    info = check(lib().mi355gp_pdinv(device, A, n, Ai.ctypes.data_as(_c_dp), L.ctypes.data_as(_c_dp),
                                     ctypes.byref(ld), ctypes.byref(ms)), "mi355gp_pdinv")
    return Ai, L, ld.value, info, ms.value


def pdinv_full(A, device=0):
    """(Ainv, L, Li, logdet, info): all four members of GPy's pdinv tuple (GPy/util/linalg.py:193-214)."""
    require_device(device)
    A = f64(A)
    n = A.shape[0]
    Ai, L, Li = np.empty((n, n)), np.empty((n, n)), np.empty((n, n))
    ld, ms = ctypes.c_double(0.0), ctypes.c_double(0.0)
    info = check(lib().mi355gp_pdinv_full(device, A, n, Ai.ctypes.data_as(_c_dp), L.ctypes.data_as(_c_dp),
                                          Li.ctypes.data_as(_c_dp), ctypes.byref(ld), ctypes.byref(ms)), "mi355gp_pdinv_full")
    return Ai, L, Li, ld.value, info


def bench_factor(N, reps=3, device=0):
    """Device-only timing of potrf / trtri / lauum on a synthetic SPD matrix: dict of ms and TFLOP/s (N^3/3 each)."""
    require_device(device)
    p, t, l = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    check(lib().mi355gp_bench_factor(device, N, reps, ctypes.byref(p), ctypes.byref(t), ctypes.byref(l)),
          "mi355gp_bench_factor")
    fl = float(N) ** 3 / 3.0
    return {"potrf_ms": p.value, "trtri_ms": t.value, "lauum_ms": l.value,
            "potrf_tflops": fl / p.value / 1e9, "trtri_tflops": fl / t.value / 1e9, "lauum_tflops": fl / l.value / 1e9}


def dbg_persist(N, reps=3, kcap=0, device=0):
    """The persistent dataflow Cholesky (persist.hip) next to the launch-per-step schedule on the same resident SPD matrix:
    dict(ms_steps, ms_persist, mismatches (doubles of the lower triangle of L that differ BITWISE), info, steps = per chain
    step the six wall-clock stamps in microseconds relative to the first)."""
    require_device(device)
    nt = (int(N) + 127) // 128
    out = np.zeros(8 + 32 * nt)
    check(lib().mi355gp_dbg_persist(device, int(N), int(reps), int(kcap), out), "mi355gp_dbg_persist")
    st = out[8:8 + 8 * nt].reshape(nt, 8)
    near = out[8 + 8 * nt:8 + 20 * nt].reshape(nt, 3, 4)   # [..., 3] of entry (j+1, 0): the chain published row j+1
    far = out[8 + 20 * nt:].reshape(nt, 12)                # far tile (i, i-3): last pass picked / done, solve picked / staged / solved / published
    return dict(ms_steps=out[0], ms_persist=out[1], mismatches=int(out[2]), info=int(out[3]), abort=int(out[4]), nt=nt,
                steps=(st - st[0, 0]) / 100.0, near=(near - st[0, 0]) / 100.0, far=(far - st[0, 0]) / 100.0)


def dbg_mask_probe(pct=75, order=0, device=0):
    """ms of a 4096^3 GEMM on (plain, CU-masked, CU-masked again) streams: is the mask in force?"""
    require_device(device)
    out = np.zeros(3)
    check(lib().mi355gp_dbg_mask_probe(device, pct, order, out), "mi355gp_dbg_mask_probe")
    return out


def dbg_mfma(a, b, device=0):
    require_device(device)
    d = np.zeros(256)
    check(lib().mi355gp_dbg_mfma(device, f64(a), f64(b), d), "dbg_mfma")
    return d.reshape(64, 4)


def dbg_gemm(A, B, C, a_mcontig, b_ncontig, alpha=1.0, beta=0.0, reps=0, device=0):
    """C = alpha*op(A) op(B) + beta*C.  A: (M,K) [k-contig] or (K,M) [m-contig]; B: (N,K) or (K,N)."""
    require_device(device)
    A, B, C = f64(A), f64(B), f64(C).copy()
    M, K = (A.shape[1], A.shape[0]) if a_mcontig else A.shape
    N = B.shape[1] if b_ncontig else B.shape[0]
    ms = ctypes.c_double(0.0)
    check(lib().mi355gp_dbg_gemm(device, int(a_mcontig), int(b_ncontig), M, N, K, A, B, C, alpha, beta, reps,
                                 ctypes.byref(ms)), "dbg_gemm")
    return C, ms.value


def dbg_gemm_clock():
    mhz, cyc = ctypes.c_double(0.0), ctypes.c_double(0.0)
    check(lib().mi355gp_dbg_gemm_clock(ctypes.byref(mhz), ctypes.byref(cyc)), "dbg_gemm_clock")
    return mhz.value, cyc.value


def dbg_peaks(device=0):
    require_device(device)
    out = np.zeros(8)
    check(lib().mi355gp_dbg_peaks(device, out), "dbg_peaks")
    return dict(mfma_f64_tflops=out[0], valu_f64_tflops=out[1], hbm_copy_gbs=out[2], hbm_fill_gbs=out[3],
                mfma_cycles_per_inst_1wave=out[4], shader_mhz_under_load=out[5], mfma_1wave_tflops=out[6],
                mfma_cycles_per_inst_loaded=out[7])


def dbg_grid_multi(T, nb=512, reps=3, device=0):
    """ms of the deep-K X^T X pass: [k_lauum row-major, grid kernel on two panel stores, on one panel store, grid kernel
    on the row-major matrix] (mi355gp_dbg_grid_multi)."""
    require_device(device)
    out = np.zeros(5)
    check(lib().mi355gp_dbg_grid_multi(device, int(T), int(nb), int(reps), out), "mi355gp_dbg_grid_multi")
    return out


def dbg_update_nt(nt, ks, reps=5, device=0):
    """ms per launch of the trailing-update kernel alone on the lower triangle of nt x nt tiles, one entry per panel depth."""
    require_device(device)
    ka = (ctypes.c_int * len(ks))(*[int(k) for k in ks])
    out = np.zeros(len(ks))
    check(lib().mi355gp_dbg_update_nt(device, int(nt), ka, len(ks), int(reps), out), "mi355gp_dbg_update_nt")
    return out


def dbg_update_rect(ntr, ntc, ks, reps=5, device=0):
    """the same for rows [ntc, ntr) x columns [0, ntc) of tiles (part 1 of a step: the next panel's columns)"""
    require_device(device)
    ka = (ctypes.c_int * len(ks))(*[int(k) for k in ks])
    out = np.zeros(len(ks))
    check(lib().mi355gp_dbg_update_rect(device, int(ntr), int(ntc), ka, len(ks), int(reps), out), "mi355gp_dbg_update_rect")
    return out


def persist_owners(nt, nw, tune=0):
    """Host only: the ownership map of a persistent launch (persist.hip): (owner matrix nt x nt as described in
    include/mi355gp_debug.h, near owners, half owners, far workers, most tiles one worker holds, tiles claimed twice)."""
    owner = (ctypes.c_int * (nt * nt))()
    out4 = (ctypes.c_int * 4)()
    dup = lib().mi355gp_dbg_persist_owners(nt, nw, tune, owner, out4)
    return np.frombuffer(owner, dtype=np.int32, count=nt * nt).reshape(nt, nt).copy(), out4[0], out4[1], out4[2], out4[3], dup


def lauum_plan(nt):
    """Host only: the X^T X work list for nt x nt tiles (gemm.hip): (items as an int array of rows (ti, tj, q, k0, klen, part),
    number of partial tiles, edge of an item's output tile, longest chunk in rows)."""
    out4 = (ctypes.c_int * 4)()
    lib().mi355gp_dbg_lauum_plan(nt, None, 0, out4)
    n = out4[0]
    buf = (ctypes.c_int * (6 * max(n, 1)))()
    check(lib().mi355gp_dbg_lauum_plan(nt, buf, n, out4), "mi355gp_dbg_lauum_plan")
    return np.frombuffer(buf, dtype=np.int32, count=6 * n).reshape(n, 6).copy(), out4[1], out4[2], out4[3]
