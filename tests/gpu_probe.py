"""Developer probe (run on the GPU box): every kernel family against NumPy / the oracle, plus timings.

    python tests/gpu_probe.py [quick|full]

Prints one line per check and never stops at the first failure: one gpurun call gives the whole picture.
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpy_amd import _lib as L  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "quick"


def section(name):
    def deco(f):
        def run():
            t0 = time.time()
            try:
                f()
            except Exception:
                print("[%s] EXCEPTION\n%s" % (name, traceback.format_exc()))
            print("[%s] done in %.2fs" % (name, time.time() - t0), flush=True)
        return run
    return deco


@section("mfma-layout")
def probe_mfma():
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(64), rng.standard_normal(64)
    d = L.dbg_mfma(a, b)
    lane = np.arange(64)
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[lane & 15, lane >> 4] = a
    B[lane >> 4, lane & 15] = b
    ref = A @ B
    got_f64 = np.zeros((16, 16)); got_f32 = np.zeros((16, 16))
    for r in range(4):
        got_f64[(lane >> 4) + 4 * r, lane & 15] = d[:, r]
        got_f32[4 * (lane >> 4) + r, lane & 15] = d[:, r]
    print("  layout row=(l>>4)+4r : max err %.3e" % np.abs(got_f64 - ref).max())
    print("  layout row=4(l>>4)+r : max err %.3e" % np.abs(got_f32 - ref).max())


@section("gemm")
def probe_gemm():
    rng = np.random.default_rng(1)
    M, N, K = 256, 384, 144 if False else 160
    for am in (0, 1):
        for bn in (0, 1):
            A = rng.standard_normal((K, M) if am else (M, K))
            B = rng.standard_normal((K, N) if bn else (N, K))
            C0 = rng.standard_normal((M, N))
            opA = A.T if am else A
            opB = B if bn else B.T
            ref = 0.7 * opA @ opB - 1.3 * C0
            C, _ = L.dbg_gemm(A, B, C0, am, bn, alpha=0.7, beta=-1.3)
            print("  gemm a_mcontig=%d b_ncontig=%d : max err %.3e (|ref| %.2e)" % (am, bn, np.abs(C - ref).max(), np.abs(ref).max()))
    for n in ((2048, 512), (4096, 512), (8192, 512), (4096, 4096)) if MODE == "full" else ((2048, 512),):
        M = N = n[0]; K = n[1]
        A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K)); C0 = np.zeros((M, N))
        _, ms = L.dbg_gemm(A, B, C0, 0, 0, reps=5)
        print("  gemm NT %dx%dx%d: %.3f ms  %.1f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))


def spd(n, seed=0, D=4):
    X, _ = O.synthetic(n, D, seed=seed)
    return O.kern_K("rbf", X, None, 1.3, 1.4, False) + 0.1 * np.eye(n)


@section("potrf")
def probe_potrf():
    import scipy.linalg as sla
    for n in (16, 100, 128, 129, 256, 640, 1000, 1536) + ((4096,) if MODE == "full" else ()):
        A = spd(n, seed=n)
        Lr = sla.cholesky(A, lower=True)
        Lg, info, ms = L.potrf(A)
        print("  potrf n=%5d info=%d  max|L-Lref| %.3e  resid %.3e  (%.2f ms)" % (
            n, info, np.abs(Lg - Lr).max(), np.abs(Lg @ Lg.T - A).max(), ms))
    A = spd(300, seed=3); A[200, 200] = -1.0
    _, info, _ = L.potrf(A)
    print("  potrf non-PD (pivot 201 negative): info=%d" % info)


@section("pdinv")
def probe_pdinv():
    for n in (64, 128, 300, 640, 1100) + ((3000,) if MODE == "full" else ()):
        A = spd(n, seed=n + 1)
        Ai, Lg, ld, info, ms = L.pdinv(A)
        ref = np.linalg.inv(A)
        sign, ldr = np.linalg.slogdet(A)
        print("  pdinv n=%5d info=%d  max|Ai-ref| %.3e (|ref| %.2e)  logdet err %.3e  asym %.1e (%.2f ms)" % (
            n, info, np.abs(Ai - ref).max(), np.abs(ref).max(), abs(ld - ldr), np.abs(Ai - Ai.T).max(), ms))


@section("kern")
def probe_kern():
    rng = np.random.default_rng(5)
    for kind in O.KINDS:
        for ARD in (False, True):
            N, M, D = 203, 77, 5
            X = rng.standard_normal((N, D)); X2 = rng.standard_normal((M, D))
            var, ls, _ = O.default_theta(D, ARD)
            th = L.theta_vec(var, ls, ARD, D)
            K = L.kern_K(kind, ARD, th, X)
            Kx = L.kern_K(kind, ARD, th, X, X2)
            e1 = np.abs(K - O.kern_K(kind, X, None, var, ls, ARD)).max()
            e2 = np.abs(Kx - O.kern_K(kind, X, X2, var, ls, ARD)).max()
            G = rng.standard_normal((N, M)); Gs = rng.standard_normal((N, N))
            g = L.update_gradients_full(kind, ARD, th, G, X, X2)
            dv, dl = O.update_gradients_full(kind, G, X, X2, var, ls, ARD)
            gs = L.update_gradients_full(kind, ARD, th, Gs, X)
            dvs, dls = O.update_gradients_full(kind, Gs, X, None, var, ls, ARD)
            ref = np.concatenate([[dv], dl]); refs = np.concatenate([[dvs], dls])
            print("  %-11s ARD=%d  K %.2e  Kx %.2e  grad(X,X2) rel %.2e  grad(X) rel %.2e" % (
                kind, ARD, e1, e2, np.abs(g - ref).max() / np.abs(ref).max(), np.abs(gs - refs).max() / np.abs(refs).max()))


@section("inference")
def probe_inference():
    ctx = L.Context(0)
    cases = [("rbf", False, 512, 2, 1), ("matern52", True, 512, 2, 1), ("matern32", False, 300, 3, 3),
             ("exponential", True, 130, 4, 1), ("rbf", True, 1000, 8, 1)]
    if MODE == "full":
        cases.append(("matern52", True, 2048, 8, 1))
    for kind, ARD, N, D, Dy in cases:
        X, Y = O.synthetic(N, D, seed=N, Dy=Dy)
        var, ls, noise = O.default_theta(D, ARD)
        ref = O.parameters_changed(kind, X, Y, var, ls, ARD, noise)
        ctx.set_data(X, Y)
        th = L.theta_vec(var, ls, ARD, D)
        info, r = ctx.exact_inference(kind, ARD, th, noise, want_diag=True)
        gref = np.concatenate([[ref["dvar"]], ref["dlen"]])
        print("  %-11s ARD=%d N=%4d D=%d Dy=%d info=%d  lml rel %.2e  alpha rel %.2e  dtheta rel %.2e  dnoise rel %.2e  diag %.2e" % (
            kind, ARD, N, D, Dy, info, abs(r["lml"] - ref["lml"]) / abs(ref["lml"]),
            np.linalg.norm(r["alpha"] - ref["alpha"]) / np.linalg.norm(ref["alpha"]),
            np.abs(r["dtheta"] - gref).max() / np.abs(gref).max(),
            abs(r["dnoise"] - ref["dL_dnoise"]) / abs(ref["dL_dnoise"]),
            np.abs(r["diag_dL_dK"] - ref["diag_dL_dK"]).max() / np.abs(ref["diag_dL_dK"]).max()))
        Lg = ctx.fetch(L.FETCH_L)
        G = ctx.fetch(L.FETCH_DLDK)
        Kg = ctx.fetch(L.FETCH_K)
        Wi = ctx.fetch(L.FETCH_KINV)
        print("      fetch: L %.2e  dL_dK %.2e (|ref| %.1e)  K %.2e  Kinv %.2e" % (
            np.abs(Lg - ref["L"]).max(), np.abs(G - ref["dL_dK"]).max(), np.abs(ref["dL_dK"]).max(),
            np.abs(Kg - ref["K"]).max(), np.abs(Wi - ref["Wi"]).max()))
        i2, r2 = ctx.inference_given_K(ref["K"], noise)
        print("      given K: lml rel %.2e alpha rel %.2e" % (abs(r2["lml"] - ref["lml"]) / abs(ref["lml"]),
              np.linalg.norm(r2["alpha"] - ref["alpha"]) / np.linalg.norm(ref["alpha"])))
    ctx.close()


@section("timing")
def probe_timing():
    print("  peaks:", {k: round(v, 1) for k, v in L.dbg_peaks().items()})
    ctx = L.Context(0)
    sizes = [("rbf", False, 4096, 8)]
    if MODE == "full":
        sizes += [("rbf", False, 8192, 8), ("matern52", True, 16384, 32)]
    for kind, ARD, N, D in sizes:
        X, Y = O.synthetic(N, D, seed=0)
        var, ls, noise = O.default_theta(D, ARD)
        ctx.set_data(X, Y)
        th = L.theta_vec(var, ls, ARD, D)
        for it in range(3):
            t0 = time.time()
            info, r = ctx.exact_inference(kind, ARD, th, noise, want_alpha=False, want_stage_ms=True)
            wall = (time.time() - t0) * 1e3
        ms = r["stage_ms"]
        print("  N=%5d D=%2d %s: wall %.2f ms  stages %s" % (N, D, kind, wall, {k: round(v, 3) for k, v in ms.items()}))
        print("      potrf %.1f TF/s  trtri %.1f  lauum %.1f  | lml %.6f info %d" % (
            N ** 3 / 3 / ms["potrf"] / 1e9, N ** 3 / 3 / ms["trtri"] / 1e9, N ** 3 / 3 / ms["lauum"] / 1e9, r["lml"], info))
    ctx.close()


@section("host-classes")
def probe_host():
    import gpy_amd
    X, Y = O.synthetic(900, 4, seed=21, Dy=2)
    var, ls, noise = O.default_theta(4, True)
    ref = O.parameters_changed("matern32", X, Y, var, ls, True, noise)
    m = gpy_amd.GPRegression(X, Y, gpy_amd.Matern32(4, variance=var, lengthscale=ls, ARD=True), noise_var=noise)
    gref = np.concatenate([[ref["dvar"]], ref["dlen"], [ref["dL_dnoise"]]])
    print("  GPRegression lml rel %.2e grad rel %.2e" % (abs(m.log_likelihood() - ref["lml"]) / abs(ref["lml"]),
          np.abs(m.gradient - gref).max() / np.abs(gref).max()))
    Xs = np.random.default_rng(3).standard_normal((333, 4))
    mu, v = m.predict_noiseless(Xs)
    mur, vr = O.predict("matern32", X, Xs, ref["L"], ref["alpha"], var, ls, True)
    mu2, C = m.predict_noiseless(Xs, full_cov=True)
    _, Cr = O.predict("matern32", X, Xs, ref["L"], ref["alpha"], var, ls, True, full_cov=True)
    print("  predict: mu %.2e var %.2e cov %.2e" % (np.abs(mu - mur).max(), np.abs(v - vr).max(), np.abs(C - Cr).max()))
    # jitter ladder: duplicated rows, zero noise -> singular K; must succeed like jitchol does
    Xd = np.vstack([X[:200], X[:200]]); Yd = np.vstack([Y[:200], Y[:200]])
    try:
        md = gpy_amd.GPRegression(Xd, Yd, gpy_amd.RBF(4, variance=1.0, lengthscale=1.5), noise_var=1e-300)
        K = O.kern_K("rbf", Xd, None, 1.0, 1.5, False)
        try:
            r2 = O.exact_inference(K, Yd, 1e-300)
            print("  ladder: device lml %.6f oracle lml %.6f" % (md.log_likelihood(), r2["lml"]))
        except Exception as e:
            print("  ladder: device lml %.6f, oracle raised %r" % (md.log_likelihood(), e))
    except Exception as e:
        print("  ladder: device raised %r" % (e,))
    res = m.optimize(max_iters=15)
    print("  optimize: %d its, objective %.4f -> params %s" % (res.nit, res.fun, np.round(m.param_array, 4)))


@section("lookahead")
def probe_lookahead():
    ctx = L.Context(0)
    for kind, ARD, N, D in [("rbf", False, 4096, 8)] + ([("matern52", True, 16384, 32)] if MODE == "full" else []):
        X, Y = O.synthetic(N, D, seed=0)
        var, ls, noise = O.default_theta(D, ARD)
        ctx.set_data(X, Y)
        th = L.theta_vec(var, ls, ARD, D)
        out = {}
        for la in (0, 1):
            ctx.set_option("lookahead", la)
            for prof in (0, 1):
                ctx.set_option("profile", prof)
                for it in range(3):
                    info, r = ctx.exact_inference(kind, ARD, th, noise, want_alpha=False, want_stage_ms=True)
                print("  N=%5d lookahead=%d profile=%d: total %.3f ms potrf %.3f trtri %.3f lauum %.3f lml %.9f" % (
                    N, la, prof, r["stage_ms"]["total"], r["stage_ms"]["potrf"], r["stage_ms"]["trtri"],
                    r["stage_ms"]["lauum"], r["lml"]))
                if prof:
                    pf = ctx.get_profile()
                    print("      profile:", {k: "%.3f ms %.1f TF/s x%d" % (v[0], v[1] / max(v[0], 1e-9) / 1e9, v[2]) for k, v in pf.items()})
            out[la] = r["lml"]
        print("  lml identical with/without look-ahead:", out[0] == out[1])
    ctx.close()


if __name__ == "__main__":
    print("lib:", L.lib().mi355gp_version().decode(), "devices:", L.device_count(), "mode:", MODE, flush=True)
    allf = dict(mfma=probe_mfma, gemm=probe_gemm, potrf=probe_potrf, pdinv=probe_pdinv, kern=probe_kern,
                inference=probe_inference, timing=probe_timing, host=probe_host,
                lookahead=probe_lookahead)
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(allf)
    for nm in names:
        allf[nm]()
