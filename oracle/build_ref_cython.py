"""TEST INFRASTRUCTURE ONLY -- builds the REFERENCE'S OWN native extensions of the hot path out of tree.

A pip-installed GPy runs `update_gradients_full` / `gradients_X` / `symmetrify` through three Cython / C files
(reference `setup.py:100-118`; `GPy/kern/src/stationary_cython.pyx:53-62`, `GPy/kern/src/stationary_utils.c:1-14`,
`GPy/util/linalg_cython.pyx:9-18`; enabled by `GPy/defaults.cfg:26-27` `working = True`).  This recipe compiles
those sources WHERE THEY LIE under /root/reference with the reference's own flags (`-fopenmp -O3`, `-lgomp`) and
writes every product (generated .c, .so) under `oracle/_ref/cython/GPy/...` only -- nothing is copied into the
repository (oracle/_ref/ is git-ignored) and nothing under /root/reference is written.

`oracle/ref_loader.load()` appends `oracle/_ref/cython/GPy/{util,kern/src}` to the stub packages' `__path__`, so the
reference's `from . import stationary_cython` / `from . import linalg_cython` pick the builds up
(`use_stationary_cython == use_linalg_cython == True`): the CPU baseline then times the reference exactly as an
installed GPy would run it.

    python oracle/build_ref_cython.py        (no-op when /root/reference is absent, e.g. on the GPU box)
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("GPY_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "_ref", "cython")

# (module, package dir relative to the reference root, [extra C sources])
EXTENSIONS = [
    ("stationary_cython", "GPy/kern/src", ["GPy/kern/src/stationary_utils.c"]),
    ("linalg_cython", "GPy/util", []),
]


def built_path(mod, pkg):
    return os.path.join(OUT, pkg, mod + sysconfig.get_config_var("EXT_SUFFIX"))


def build(verbose=False):
    """Returns the list of built extension files (empty when the reference tree is not present)."""
    if not os.path.isdir(os.path.join(REF_ROOT, "GPy")):
        return []
    import numpy as np
    inc_py = sysconfig.get_paths()["include"]
    inc_np = np.get_include()
    outs = []
    for mod, pkg, extra in EXTENSIONS:
        pyx = os.path.join(REF_ROOT, pkg, mod + ".pyx")
        odir = os.path.join(OUT, pkg)
        os.makedirs(odir, exist_ok=True)
        so = built_path(mod, pkg)
        srcs = [pyx] + [os.path.join(REF_ROOT, e) for e in extra]
        if os.path.exists(so) and all(os.path.getmtime(s) <= os.path.getmtime(so) for s in srcs):
            outs.append(so)
            continue
        c_file = os.path.join(odir, mod + ".c")
        # cython: .pyx -> .c (output redirected out of the reference tree)
        cmd = [sys.executable, "-m", "cython", "-3", "-o", c_file, pyx]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OUT)
        if r.returncode != 0:
            raise RuntimeError("cython failed for %s:\n%s%s" % (pyx, r.stdout, r.stderr))
        # gcc with the reference's flags (setup.py:93-94): -fopenmp -O3, link -lgomp
        cmd = ["gcc", "-shared", "-fPIC", "-fopenmp", "-O3", "-w", "-I", inc_py, "-I", inc_np,
               "-I", os.path.join(REF_ROOT, pkg), "-I", REF_ROOT, c_file] + srcs[1:] + ["-lgomp", "-o", so]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed for %s:\n%s%s" % (mod, r.stdout, r.stderr))
        if verbose:
            print("built", so)
        outs.append(so)
    return outs


if __name__ == "__main__":
    built = build(verbose=True)
    print("%d reference extension(s) under %s" % (len(built), OUT) if built else "reference tree absent: nothing built")
