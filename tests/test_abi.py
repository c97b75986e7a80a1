"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mi355gp.h (the drop-in boundary) and
include/mi355gp_debug.h (diagnostics for tests / tools) declare.
No compute calls here (no GPU in this container); device entry points must fail loudly instead of falling back."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from gpy_amd import _lib


def _declared_symbols(header="mi355gp.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355gp_[a-z_0-9A-Z]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = _lib.build()
    assert os.path.exists(path)
    lib = _lib.lib()
    product, debug = _declared_symbols(), _declared_symbols("mi355gp_debug.h")
    assert len(product) >= 20
    assert not [s for s in product if s.startswith("mi355gp_dbg_")], "diagnostics belong in mi355gp_debug.h"
    declared = product + debug
    for sym in declared:
        assert hasattr(lib, sym), "libmi355gp.so does not export %s" % sym
    assert set(_lib.EXPORTED) <= set(declared)
    assert b"gfx950" in lib.mi355gp_version()


PRODUCT_ENV = {"MI355GP_TRANSPORT", "MI355GP_IPC_HOST", "MI355GP_IPC_TIMEOUT_S", "MI355GP_GRID_CHECK_SEQ", "MI355GP_GRAPH",
               "MI355GP_PERSIST", "MI355GP_PERSIST_AUTO", "MI355GP_TRI_OVERLAP", "MI355GP_GRID_LOOKAHEAD"}


def _env_names(path):
    data = open(path, "rb").read()
    return set(m.decode() for m in re.findall(rb"MI355GP_[A-Z0-9_]+", data)) - {"MI355GP_OPT_PERSIST_TEST"}


def test_the_product_library_reads_only_the_documented_environment_variables():
    """the shipped library used to read 37 MI355GP_* variables, one of which (a bounding experiment)
    produced wrong results.  Now the schedule overrides of the A/B tools, the fault injectors and the experiments are compiled
    out of libmi355gp.so (common.h DIAG_ENV) and live in libmi355gp_diag.so only; what is left is the transport of the
    multi-process mode, one self-check and five schedule choices that give the same bits either way."""
    _lib.build()
    names = _env_names(os.path.join(ROOT, "gpy_amd", "libmi355gp.so"))
    assert names == PRODUCT_ENV, sorted(names ^ PRODUCT_ENV)
    diag = _env_names(_lib.DIAG_LIB_PATH)
    assert PRODUCT_ENV < diag and {"MI355GP_DBG_UPD_QUEUE", "MI355GP_DENSE_PERSIST_TEST", "MI355GP_AGG2"} <= diag
    # the experiment that computes wrong numbers is not even linked into the product
    sym = lambda p: open(p, "rb").read().count(b"k_update_nt_queue")
    assert sym(os.path.join(ROOT, "gpy_amd", "libmi355gp.so")) == 0 and sym(_lib.DIAG_LIB_PATH) > 0
    readme = open(os.path.join(ROOT, "README.md")).read()
    for n in PRODUCT_ENV:
        assert n in readme, "README.md does not document %s" % n


def test_code_object_is_gfx950_only():
    import subprocess
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", _lib.LIB_PATH],
                         capture_output=True, text=True).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"} or not archs, archs


@pytest.mark.skipif(_lib.device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback_without_device():
    assert _lib.device_count() == 0
    with pytest.raises(_lib.MI355GPError):
        _lib.Context(0)
    X = np.zeros((4, 2))
    with pytest.raises(_lib.MI355GPError):
        _lib.kern_K("rbf", False, np.array([1.0, 1.0]), X)
    with pytest.raises(_lib.MI355GPError):
        _lib.potrf(np.eye(4))
    import gpy_amd
    with pytest.raises(_lib.MI355GPError):
        gpy_amd.GPRegression(X, np.zeros((4, 1)))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "gpy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    # bench.py: the oracle is the CHECKER (parity gate before the timed region) and the CPU-baseline leg, nothing else
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    allowed = {"cpu_baseline", "_oracle_seconds", "parity_gate", "sparse_cpu_baseline"}

    def oracle_imports(node):
        return [n for n in ast.walk(node) if (isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle")) or
                (isinstance(n, ast.Import) and any(a.name.startswith("oracle") for a in n.names))]
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in allowed:
            continue
        assert not oracle_imports(node), getattr(node, "name", node)


def test_kdiag_is_host_side_and_exact():
    out = _lib.kern_Kdiag("matern52", np.array([1.7, 0.3]), 5)
    assert np.array_equal(out, np.full(5, 1.7))


def test_option_ids_of_the_python_binding_match_the_header():
    """`Context.set_option` configures ONE context through the C-ABI: the name -> id table of the binding
    is the header's MI355GP_OPT_* enum."""
    text = open(os.path.join(ROOT, "include", "mi355gp.h")).read()
    m = re.search(r"enum \{ (MI355GP_OPT_PROFILE.*?) \};", text, flags=re.S)
    ids = {k.strip()[len("MI355GP_OPT_"):].lower(): int(v) for k, v in
           (item.split("=") for item in m.group(1).replace("\n", " ").split(","))}
    num = ids.pop("num")
    assert ids == _lib.OPTIONS and num == len(ids)
