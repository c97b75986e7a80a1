import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
DIAG_LIB = os.path.join(ROOT, "gpy_amd", "libmi355gp_diag.so")


def diag_lib(fn):
    """Run this test in a CHILD pytest process that loads the DIAGNOSTICS build of the library (libmi355gp_diag.so,
    -DMI355GP_DIAG): the fault injectors and schedule overrides it drives are compiled out of the product library, which is
    what every other test -- and this process -- loads.  The child runs exactly this test node with MI355GP_LIB pointing at
    the diagnostics build; its failure output becomes this test's."""
    import functools
    import subprocess

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if os.environ.get("MI355GP_TEST_DIAG_CHILD") == "1":
            return fn(*args, **kwargs)
        assert os.path.exists(DIAG_LIB), "diagnostics build missing: make -C gpy_amd/csrc diag (or __graft_entry__.build())"
        node = os.environ["PYTEST_CURRENT_TEST"].rsplit(" ", 1)[0]
        r = subprocess.run([sys.executable, "-m", "pytest", node, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT,
                           env=dict(os.environ, MI355GP_LIB=DIAG_LIB, MI355GP_TEST_DIAG_CHILD="1"), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, "child pytest (diagnostics library) failed:\n" + r.stdout[-6000:] + r.stderr[-3000:]
    return wrapper


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    # sparse_* fixtures: tests/test_oracle_sparse.py; sum_* / prod_*: tests/test_oracle_sum.py
    # baseline_* (full BASELINE.json sizes, inputs regenerated from the seed): tests/test_gpu_baseline.py
    return [n for n in names if not n.startswith(("sparse_", "sparse2_", "sparse3_", "sum_", "studentt_", "prod_", "baseline_", "predict_", "predsparse_"))]


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    d["kind"] = str(d["kind"])
    d["ARD"] = bool(d["ARD"])
    d["variance"] = float(d["variance"])
    # regenerate the (X2, A) pair of the generic update_gradients_full case exactly as oracle/make_golden.py did
    N, D = d["X"].shape
    M = int(d["M"])
    rngA = np.random.default_rng(int(d["seedA"]))
    d["X2"] = rngA.standard_normal((M, D))
    d["A"] = rngA.standard_normal((N, M))
    return d


@pytest.fixture(scope="session")
def oracle_native_built():
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return True


def baseline_golden(name):
    """tests/golden/baseline_*.npz (oracle/make_golden_baseline.py): outputs of the REFERENCE at the full BASELINE.json
    sizes; the inputs are regenerated from the stored seed."""
    p = os.path.join(GOLDEN_DIR, name + ".npz")
    if not os.path.exists(p):
        pytest.skip("golden fixture %s not generated" % name)
    d = dict(np.load(p, allow_pickle=False))
    for k in ("kind", "source"):
        d[k] = str(d[k])
    d["ARD"] = bool(d["ARD"])
    for k in ("N", "D", "seed", "M"):
        if k in d:
            d[k] = int(d[k])
    for k in ("variance", "noise", "lml"):
        d[k] = float(d[k])
    return d
