"""CPU: host-side logic that mirrors the reference's interface (no device work)."""
import pickle

import numpy as np
import pytest

import gpy_amd
from gpy_amd import _lib
from gpy_amd.inference import ExactGaussianInference
from gpy_amd.lazy import DeviceResult, kernel_signature
from gpy_amd.param import Param


def test_param_gradient_broadcast_and_pickle():
    p = Param("lengthscale", [1.0, 2.0, 3.0])
    p.gradient = 0.
    assert p.gradient.shape == (3,) and np.all(p.gradient == 0)
    p.gradient = np.array([1.0, 2.0, 3.0])
    q = pickle.loads(pickle.dumps(p))
    assert q.name == "lengthscale" and np.array_equal(q.gradient, [1, 2, 3]) and np.array_equal(q.values, [1, 2, 3])


def test_kernel_constructor_semantics_follow_the_reference():
    # reference GPy/kern/src/stationary.py:60-81: non-ARD takes one lengthscale, ARD expands a scalar
    k = gpy_amd.RBF(3)
    assert k.lengthscale.size == 1 and not k.ARD
    k = gpy_amd.Matern52(3, lengthscale=2.0, ARD=True)
    assert np.array_equal(k.lengthscale.values, [2.0, 2.0, 2.0])
    with pytest.raises(AssertionError):
        gpy_amd.RBF(3, lengthscale=[1.0, 2.0])
    with pytest.raises(AssertionError):
        gpy_amd.RBF(3, lengthscale=[1.0, 2.0], ARD=True)
    # link order: variance, then lengthscale (stationary.py:78-81)
    assert [p.name for p in k.parameters] == ["variance", "lengthscale"]
    assert np.array_equal(k._theta(), [1.0, 2.0, 2.0, 2.0])


def test_rbf_inverse_lengthscale_parameterisation():
    # reference rbf.py:29-33,328-330,373-375
    k = gpy_amd.RBF(2, lengthscale=0.5, inv_l=True)
    assert [p.name for p in k.parameters] == ["variance", "inv_lengthscale"]
    assert np.allclose(k.inv_l.values, 4.0)
    k.inv_l[:] = 16.0
    k.parameters_changed()
    assert np.allclose(k.lengthscale.values, 0.25)
    k._install_gradients(np.array([0.3, 2.0]))
    assert np.allclose(k.inv_l.gradient, 2.0 * (0.25 ** 3 / -2.0))


def test_active_dims_slicing():
    k = gpy_amd.Matern32(2, active_dims=[0, 3])
    X = np.arange(20.0).reshape(4, 5)
    assert np.array_equal(k._slice_X(X), X[:, [0, 3]])
    with pytest.raises(AssertionError):
        k._slice_X(X[:, :3])


def test_to_dict_round_trip_uses_gpy_class_strings():
    k = gpy_amd.Matern52(3, variance=1.5, lengthscale=[1, 2, 3], ARD=True, name="m52")
    d = k.to_dict()
    assert d["class"] == "GPy.kern.Matern52"
    k2 = gpy_amd.Matern52.from_dict(d)
    assert np.array_equal(k2._theta(), k._theta()) and k2.name == "m52"
    assert gpy_amd.RBF(1).to_dict()["class"] == "GPy.kern.RBF"
    inf = ExactGaussianInference()
    assert inf.to_dict()["class"].endswith("exact_gaussian_inference.ExactGaussianInference")
    assert pickle.loads(pickle.dumps(inf))._state is None


def test_model_flat_parameter_order_and_names():
    from gpy_amd.param import Parameterized
    k = gpy_amd.RBF(2, variance=1.3, lengthscale=[0.5, 0.7], ARD=True)
    lik = gpy_amd.Gaussian(0.1)
    m = Parameterized("gp")
    m.link_parameter(k)
    m.link_parameter(lik)
    assert np.allclose(m.param_array, [1.3, 0.5, 0.7, 0.1])          # [variance, lengthscale..., noise]
    assert m.parameter_names() == ["rbf.variance", "rbf.lengthscale[0]", "rbf.lengthscale[1]", "Gaussian_noise.variance"]
    m.param_array = [2.0, 1.0, 3.0, 0.5]
    assert float(k.variance[0]) == 2.0 and np.array_equal(k.lengthscale.values, [1.0, 3.0]) and float(lik.variance[0]) == 0.5


class _FakeAttempts(object):
    def __init__(self, fail_first):
        self.fail_first, self.calls = fail_first, []

    def __call__(self, extra):
        self.calls.append(extra)
        return (7, None) if len(self.calls) <= self.fail_first else (0, "ok")


def test_jitter_ladder_matches_jitchol():
    """reference GPy/util/linalg.py:56-75 and GPy/testing/test_linalg.py:20-37."""
    inf = ExactGaussianInference()
    diag = np.full(10, 2.0)
    a = _FakeAttempts(0)
    assert inf._run_with_ladder(a, diag) == "ok" and a.calls == [0.0]
    a = _FakeAttempts(3)
    assert inf._run_with_ladder(a, diag) == "ok"
    assert np.allclose(a.calls, [0.0, 2e-6, 2e-5, 2e-4])
    a = _FakeAttempts(6)                         # plain try + 5 ladder tries all fail
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite, even with jitter."):
        inf._run_with_ladder(a, diag)
    assert len(a.calls) == 6 and np.isclose(a.calls[-1], 2e-2)
    with pytest.raises(np.linalg.LinAlgError, match="not pd: non-positive diagonal elements"):
        inf._run_with_ladder(_FakeAttempts(1), np.array([1.0, -1.0]))


def test_device_result_proxy_contract():
    class Ctx(object):
        call_token = 3

        def fetch(self, which, fortran_order=False):
            return np.arange(9.0).reshape(3, 3)
    k = gpy_amd.RBF(2)
    r = DeviceResult(Ctx(), _lib.FETCH_DLDK, 3, token=3, kernel_sig=kernel_signature(k), fused_dtheta=np.array([1., 2.]))
    assert r.shape == (3, 3) and r.matches_kernel(k)
    k.variance[:] = 2.0
    assert not r.matches_kernel(k)              # parameters changed -> fused gradients are stale
    assert np.array_equal(np.asarray(r), np.arange(9.0).reshape(3, 3))
    assert np.array_equal(np.diag(r), [0, 4, 8]) and np.array_equal((r * 2)[0], [0, 2, 4])
    stale = DeviceResult(Ctx(), _lib.FETCH_L, 3, token=2)
    with pytest.raises(RuntimeError):
        np.asarray(stale)


def test_theta_vec_and_dataset_helpers():
    assert np.array_equal(_lib.theta_vec(1.5, 2.0, True, 3), [1.5, 2, 2, 2])
    assert np.array_equal(_lib.theta_vec(np.array([1.5]), np.array([2.0]), False, 3), [1.5, 2])
    from gpy_amd.datasets import synthetic, default_theta
    from oracle import gp_oracle as O
    for a, b in zip(synthetic(40, 3, seed=1, Dy=2), O.synthetic(40, 3, seed=1, Dy=2)):
        assert np.array_equal(a, b)
    assert np.array_equal(default_theta(5, True)[1], O.default_theta(5, True)[1])


def test_standardize_normalizer_matches_reference_formulas():
    """gpy_amd.models.Standardize == GPy/util/normalizer.py:85-112 (host-only bookkeeping)."""
    from gpy_amd.models import Standardize
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((50, 3)) * [1.0, 5.0, 0.1] + [3.0, -2.0, 0.5]
    s = Standardize()
    assert not s.scaled()
    s.scale_by(Y)
    Z = s.normalize(Y)
    assert np.allclose(Z.mean(0), 0) and np.allclose(Z.std(0), 1)
    assert np.allclose(s.inverse_mean(Z), Y)
    v = rng.random((50, 1))
    assert np.allclose(s.inverse_variance(v), v * Y.std(0) ** 2)
    assert s.inverse_covariance(np.eye(4)).shape == (4, 4, 3)
    const = Standardize()
    const.scale_by(np.ones((5, 1)))
    assert const.std[0] == 1.0                      # zero standard deviation resets to 1 (normalizer.py:94-97)


def test_combination_kernels_flatten_and_describe_themselves_to_the_c_abi():
    """`Add` / `Prod` follow the reference's flattening (add.py:24-33, prod.py:33-41); `part_specs()` is the
    `mi355gp_part` list: factors of a product share a non-zero term id, plain summands carry term 0; parameter /
    gradient order is the leaves' link order."""
    import ctypes
    from gpy_amd import _lib as L
    r1 = gpy_amd.RBF(2, variance=1.5, lengthscale=0.7, active_dims=[0, 1])
    m32 = gpy_amd.Matern32(1, variance=0.4, lengthscale=2.0, active_dims=[2])
    b = gpy_amd.Bias(3, 0.3)
    w = gpy_amd.White(3, 0.05)
    r2 = gpy_amd.RBF(3, variance=0.9, lengthscale=[1.0, 2.0, 3.0], ARD=True)
    k = (r1 * m32) * b + w + (r2 + gpy_amd.Bias(3, 0.1))
    assert isinstance(k, gpy_amd.Add) and [type(p).__name__ for p in k.parts] == ["Prod", "White", "RBF", "Bias"]
    assert [type(p).__name__ for p in k.parts[0].parts] == ["RBF", "Matern32", "Bias"]          # nested Prod flattened
    specs = k.part_specs()
    assert [s[0] for s in specs] == ["rbf", "matern32", "bias", "white", "rbf", "bias"]
    assert [s[4] for s in specs] == [1, 1, 1, 0, 0, 0]
    assert [p.name for p in k.leaves()] == ["rbf", "Mat32", "bias", "white", "rbf", "bias"]
    # Kdiag of the expression: sum over summands of the product of variances
    assert abs(k.diag_variance() - (1.5 * 0.4 * 0.3 + 0.05 + 0.9 + 0.1)) < 1e-15
    arr, keep, ntheta = L.make_parts(specs)
    assert ntheta == 2 + 2 + 1 + 1 + 4 + 1 and [arr[i].term for i in range(6)] == [1, 1, 1, 0, 0, 0]
    assert arr[1].n_active == 1 and arr[1].active_dims[0] == 2 and arr[4].ard == 1
    assert ctypes.sizeof(L.Part) == 40                            # int, int, int, (pad), ptr, ptr, int, (pad): include/mi355gp.h
    d = k.to_dict()
    assert d["class"] == "GPy.kern.Add" and d["parts"][0]["class"] == "GPy.kern.Prod"
    # two products in one sum get distinct term ids
    k2 = r1 * m32 + r2 * b
    assert [s[4] for s in k2.part_specs()] == [1, 1, 2, 2]
    assert gpy_amd.lazy.kernel_signature(k2) != gpy_amd.lazy.kernel_signature(r1 * m32 + r2 + b)


def test_gpy_style_import_paths():
    """`import gpy_amd as GPy` reads like the reference on this path (no device needed for the lookups)."""
    import gpy_amd as GPy
    assert GPy.kern.RBF is GPy.RBF and GPy.kern.Matern52 is GPy.Matern52
    assert GPy.models.GPRegression is GPy.GPRegression and GPy.models.SparseGPRegression is GPy.SparseGPRegression
    assert GPy.models.GPHeteroscedasticRegression.__mro__[1] is GPy.core.GP
    assert GPy.likelihoods.Gaussian is GPy.Gaussian
    lfi = GPy.inference.latent_function_inference
    assert lfi.ExactGaussianInference is GPy.ExactGaussianInference and lfi.VarDTC is GPy.VarDTC
    assert lfi.exact_gaussian_inference.ExactGaussianInference is GPy.ExactGaussianInference
    for name in ("predict", "predict_noiseless", "predict_quantiles", "predictive_gradients", "log_predictive_density",
                 "posterior_samples_f", "posterior_samples", "posterior_covariance_between_points"):
        assert callable(getattr(GPy.core.GP, name)) and callable(getattr(GPy.core.SparseGP, name))


def test_small_stationary_methods_follow_the_reference():
    import gpy_amd
    k = gpy_amd.OU(3, variance=2.0, lengthscale=[0.5, 1.0, 2.0], ARD=True)
    assert k.kind == "exponential" and k.to_dict()["class"] == "GPy.kern.OU"
    assert np.allclose(k.input_sensitivity(), 2.0 / np.array([0.25, 1.0, 4.0]))       # stationary.py:363-364
    k.update_gradients_direct(1.5, np.array([0.1, 0.2, 0.3]))                           # stationary.py:215-223
    assert np.allclose(k.variance.gradient, 1.5) and np.allclose(k.lengthscale.gradient, [0.1, 0.2, 0.3])
    k.update_gradients_diag(np.ones(7), np.zeros((7, 3)))                               # stationary.py:182-191
    assert np.allclose(k.variance.gradient, 7.0) and np.allclose(k.lengthscale.gradient, 0.0)


def test_array_identity_never_misses_an_in_place_edit():
    """ADVICE r2 (high / medium): a host array is recognised either by identity -- only when it is frozen through its whole
    base chain -- or by a full comparison; never by a sampled fingerprint or by the address of a writable buffer."""
    from gpy_amd.lazy import ArrayIdentity, freeze, frozen
    rng = np.random.default_rng(0)
    a = rng.standard_normal((1000, 7))
    ida = ArrayIdentity(a)
    assert ida.matches(a) and ida.matches(a.copy())
    a[501, 3] += 1e-9                            # one element that 64 strided samples of 7000 would not see
    assert not ida.matches(a)
    f = freeze(a)
    assert frozen(f) and not frozen(a) and not f.flags.writeable and f is not a
    with pytest.raises(ValueError):
        f[0, 0] = 1.0
    idf = ArrayIdentity(f)
    assert idf.obj is f and idf.copy is None and idf.matches(f)
    assert idf.matches(f.copy()) and not idf.matches(f + 1.0)        # other objects: compared, not assumed different
    v = a[:500]                                  # a read-only VIEW of a writable buffer is not frozen: the base can be edited
    v.setflags(write=False)
    assert not frozen(v)
    idv = ArrayIdentity(v)
    a[10, 0] = 42.0
    assert not idv.matches(v)
    assert ArrayIdentity(None).matches(None) and not ArrayIdentity(None).matches(a) and not ida.matches(None)


def test_kernel_K_cache_confirms_hits_with_the_data():
    """`Stationary.K`'s Cache_this(limit=3) stand-in: same-shaped arrays at a reused address, or an edited buffer, must
    not hit (the loop `Xp = base.copy(); Xp[i, j] += eps; k.K(Xp)` false-hit 19 times out of 19 in round 2)."""
    from gpy_amd.kern import _KCache
    c = _KCache(limit=3)
    calls = []

    def compute_for(X):
        def compute():
            calls.append(1)
            return np.array([[float((X * np.arange(X.size).reshape(X.shape)).sum())]])
        return compute
    base = np.random.default_rng(1).standard_normal((300, 4))
    theta = np.array([1.0, 2.0])
    seen = set()
    for i in range(19):
        Xp = base.copy()
        Xp[(7 * i + 3) % 300, i % 4] += 1e-3
        seen.add(float(c.get(Xp, None, theta, compute_for(Xp))[0, 0]))
        del Xp
    assert len(calls) == 19 and len(seen) == 19
    Xb = base.copy()
    k1 = c.get(Xb, None, theta, compute_for(Xb))
    n = len(calls)
    assert c.get(Xb, None, theta, compute_for(Xb)) is k1 and len(calls) == n        # a true hit, confirmed by comparison
    assert not k1.flags.writeable
    Xb[123, 2] -= 0.5                                                                # in-place edit of the same object
    assert c.get(Xb, None, theta, compute_for(Xb))[0, 0] != k1[0, 0] and len(calls) == n + 1
    assert c.get(Xb, None, theta * 2, compute_for(Xb)) is not None and len(calls) == n + 2   # parameters are part of the key


def test_device_state_uploads_again_after_an_unsampled_in_place_edit():
    """`_DeviceState.ensure_data` with a recording stand-in for the device context (no GPU work)."""
    from gpy_amd.inference import _DeviceState
    from gpy_amd.lazy import freeze

    class Rec(object):
        def __init__(self):
            self.log = []

        def set_data(self, X, R):
            self.log.append(("data", float(X.sum()), float(R.sum())))

        def set_targets(self, R):
            self.log.append(("targets", float(R.sum())))
    st = _DeviceState.__new__(_DeviceState)
    st.ctx, st._idX, st._idR, st.call_token, st.kern = Rec(), None, None, 0, None
    rng = np.random.default_rng(2)
    X, Y = rng.standard_normal((4000, 3)), rng.standard_normal((4000, 1))
    st.ensure_data(X, Y)
    st.ensure_data(X, Y)
    assert [e[0] for e in st.ctx.log] == ["data"]
    X[1777, 1] += 1e-3                            # the update pattern `X_buf[i] = x_new; m.set_XY(X_buf, Y_buf)`
    st.ensure_data(X, Y)
    assert [e[0] for e in st.ctx.log] == ["data", "data"]
    Y[3001, 0] -= 1.0
    st.ensure_data(X, Y)
    assert [e[0] for e in st.ctx.log] == ["data", "data", "targets"]
    Xf, Yf = freeze(X), freeze(Y)                 # what the model drivers hold: O(1) identity hits from now on
    st.ensure_data(Xf, Yf)
    n = len(st.ctx.log)                           # equal contents: recognised by comparison, nothing uploaded
    st.ensure_data(Xf, Yf)
    st.ensure_data(Xf, Yf)
    assert len(st.ctx.log) == n
    st.invalidate()
    st.ensure_data(Xf, Yf)
    assert st.ctx.log[-1][0] == "data"


def test_generated_potf2_header_is_what_the_generator_emits():
    import os
    """gpy_amd/csrc/potf2_asm.h is GENERATED (tools/gen_potf2.py: the hand-scheduled 16 x 16 Cholesky + inverse of the chain
    kernels, with the DPP / transcendental wait states inserted by the generator's hazard tracker): the committed header must be
    the generator's current output, and no DPP read may follow the VALU write of its source by fewer than two wait states."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_potf2.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(root, "gpy_amd", "csrc", "potf2_asm.h")).read()
    lines = [ln.strip().strip('"').replace("\\n\\t", "") for ln in out.splitlines() if ln.strip().startswith('"')]
    last_write = {}
    slot = 0
    for ln in lines:
        m = re.match(r"(\S+)\s+(.*)", ln)
        op, args = m.group(1), m.group(2)
        if op == "s_nop":
            slot += int(args) + 1
            continue
        regs = re.findall(r"%\d+|v\[\d+:\d+\]", args)
        if "row_newbcast" in ln:
            src = regs[1]                                      # operand read through the DPP
            assert slot - last_write.get(src, -10) >= 3, ln
        if op.startswith("v_") and regs:
            last_write[regs[0]] = slot
        slot += 1


def test_bench_line_stays_under_the_drivers_tail():
    import os
    """bench.compact_line: the one JSON line keeps every leg and stays under 6 KB (the driver keeps an 8 KB tail of stdout)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    full = json.load(open(os.path.join(root, "profiles", "r4b_bench.json")))
    line = bench.compact_line(full)
    rec = json.loads(line)
    assert len(line) < 6144
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "c2", "grid", "c5", "c4_single", "legs_ms"):
        assert k in rec, k
    assert list(rec)[-1] == "legs_ms" and set(rec["legs_ms"]) == {"c2", "grid", "c5", "c4_single"}
    assert rec["cpu_baseline"]["value"] > 0 and rec["c2"]["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("nt", [9, 16, 23, 24, 25, 32, 36, 44])
def test_lauum_work_list_covers_every_k_range_exactly_once(nt):
    """gemm.hip `lauum_split_plan` (host code, no GPU): W(ti, tj) = sum over the tile rows k >= ti, cut into chunks.  Every output
    tile (quadrant below nt = 24) must be covered by chunks that tile [128 ti, 128 nt) without gap or overlap, a single chunk
    stores W itself (part = -1), several chunks store consecutive partials, items come longest first, and from nt = 24 the
    chunk length is the packing model's pick: a multiple of 128 rows."""
    from gpy_amd import _lib as L
    items, nparts, edge, longest = L.lauum_plan(nt)
    assert edge == (128 if nt >= 24 else 64) and longest % 128 == 0 and 0 < longest <= 128 * nt
    assert list(items[:, 4]) == sorted(items[:, 4], reverse=True)
    groups = {}
    for ti, tj, q, k0, klen, part in items:
        groups.setdefault((ti, tj, q if edge == 64 else 0), []).append((k0, klen, part))
    nq = 4 if edge == 64 else 1
    assert len(groups) == nq * nt * (nt + 1) // 2
    parts_seen = []
    for (ti, tj, q), chunks in groups.items():
        assert 0 <= tj <= ti < nt
        chunks.sort()
        pos = 128 * ti
        for k0, klen, part in chunks:
            assert k0 == pos and klen > 0 and klen % 128 == 0 and klen <= longest
            pos += klen
        assert pos == 128 * nt
        if len(chunks) == 1:
            assert chunks[0][2] == -1
        else:
            ps = [c[2] for c in chunks]
            assert ps == list(range(ps[0], ps[0] + len(ps))) and ps[0] >= 0      # summed in chunk order
            parts_seen += ps
    assert sorted(parts_seen) == list(range(nparts))


@pytest.mark.parametrize("tune", [0, 1 << 22, 1 << 20, 1 << 21, (5 << 8) | (24 << 23)])
def test_every_tile_of_a_persistent_launch_has_exactly_one_owner(tune):
    """persist.hip `Ownership` (host mirror, no GPU): for every size the launch takes (nt = 2 .. 64) and several worker counts, every
    tile of the lower triangle but block (0, 0) has exactly one owner -- the tiles of the second sub-diagonal one owner per 64-row half
    when they are held in halves (default up to nt = 40; tune bit 22 switches them off) -- near owners, half owners and far workers
    are disjoint ranges, nobody holds more tiles than the workers' per-tile state has room for (48), and far tiles dealt out
    column-major (nt >= 33, or forced) cover the same set as row-major."""
    from gpy_amd import _lib as L
    for nt in list(range(2, 41)) + [44, 48, 56, 64]:
        for nw in (15, 31, 136, 254):
            nfar = (nt - 3) * (nt - 2) // 2 if nt > 3 else 0
            owner, H, Hh, W, most, dup = L.persist_owners(nt, nw, tune)
            if nfar > 0 and W < 1:
                continue                                      # the host refuses such a launch (persist_tiles_fit)
            assert dup == 0, (nt, nw)
            assert H >= 1 and Hh >= 0 and H + Hh + W == nw
            halves = Hh > 0
            for i in range(nt):
                for k in range(i + 1):
                    o = owner[i, k]
                    if (i, k) == (0, 0):
                        assert o == -1
                        continue
                    assert o >= 0, (nt, nw, i, k)
                    if halves and i - k == 2:
                        lo = owner[k, i]
                        assert H <= o < H + Hh and H <= lo < H + Hh and lo == o + 1 and (o - H) % 2 == 0, (nt, nw, i, k, o, lo)
                    elif i - k <= 2:
                        assert 0 <= o < H, (nt, nw, i, k, o)
                    else:
                        assert H + Hh <= o < nw, (nt, nw, i, k, o)
            upper = [owner[k, i] for i in range(nt) for k in range(i) if not (halves and i - k == 2)]
            assert all(v == -2 for v in upper)
            if nw >= 136:
                assert most <= 48, (nt, nw, most)
