#!/usr/bin/env python3
"""bench.py -- exact-GP log-likelihood + gradient throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full `GP.parameters_changed` equivalent on one synthetic data set resident in HBM:
K build -> Cholesky -> alpha -> log marginal likelihood -> Ky^-1 -> all gradients (one C-ABI call).
Workload: BASELINE.json configs[2] (the N=16k configuration the metric is quoted on):
Matern-5/2 ARD, N=16384, D=32, Dy=1, fp64.

Multi-GPU: the exact path does not shard without a panel exchange every 128 columns, so
--gpus N runs N independent replicas (one process per GPU, no data-path collective, weak scaling); ranks only
meet at the barriers that bracket the timed region and for the MAX-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6       # MI355X fp64 matrix (= vector) peak, AMD spec
WORKLOAD = dict(kind="matern52", ARD=True, N=16384, D=32, Dy=1)


class Comm(object):
    """Rank plumbing for the bracketing barriers and the MAX-over-ranks (torch.distributed; RCCL when CUDA
    tensors are available, gloo otherwise).  Not on the data path."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.dist = None
        self.device_tensor = False
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device_tensor = True
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            self.torch = torch

    def _sync_device(self):
        if self.dist is not None and self.device_tensor:
            self.torch.cuda.synchronize()

    def barrier(self):
        self._sync_device()
        if self.dist is not None:
            self.dist.barrier()
        self._sync_device()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64,
                              device="cuda" if self.device_tensor else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed_region(comm, step_fn, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync; returns MAX-over-ranks seconds."""
    for _ in range(warmup):
        step_fn()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()              # each step ends with a stream synchronize inside the C-ABI call
    comm.barrier()
    dt = time.perf_counter() - t0
    return comm.max_over_ranks(dt)


def cpu_baseline(kind, ARD, D, n_sample, n_full):
    """The oracle (NumPy/SciPy restatement of GPy's CPU path, oracle/gp_oracle.py) on the host cores, on a bounded
    sample: one full iteration at N = n_sample, scaled to N = n_full by (n_full/n_sample)^3."""
    from oracle import gp_oracle as O
    try:
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    except Exception:
        pass
    X, Y = O.synthetic(n_sample, D, seed=0)
    var, ls, noise = O.default_theta(D, ARD)
    O.parameters_changed(kind, X[:512], Y[:512], var, ls, ARD, noise)        # warm BLAS threads
    t0 = time.perf_counter()
    O.parameters_changed(kind, X, Y, var, ls, ARD, noise, cached=True)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    scale = (float(n_full) / n_sample) ** 3
    return {"value": 1.0 / (dt * scale), "unit": "iters/s", "cores": int(threads), "kind": "port",
            "sample": "one full iteration of the NumPy/SciPy oracle (GPy's CPU algorithm, paramz-style K/r caching) at "
                      "N=%d D=%d measured %.2f s on %d BLAS threads (host has %d cores; the ARD gradient loop is "
                      "single-threaded as in GPy), scaled by (%d/%d)^3 to N=%d" % (
                          n_sample, D, dt, threads, os.cpu_count() or 1, n_full, n_sample, n_full),
            "measured_seconds": dt, "sample_N": n_sample}


def profiled_traffic(kernel_prefix):
    """HBM-side bytes per launch of the roofline kernel from the committed PMC summary (profiles/*_traffic.json, written
    by tools/summarize_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if no summary is committed."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")) if "sparse" not in f)
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        for name, v in ks.items():
            if name.startswith(kernel_prefix):
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=WORKLOAD["N"])
    ap.add_argument("--d", type=int, default=WORKLOAD["D"])
    ap.add_argument("--kind", default=WORKLOAD["kind"])
    ap.add_argument("--iso", action="store_true", help="single lengthscale instead of ARD")
    ap.add_argument("--cpu-sample-n", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grid", default="", help="PrxPc: ONE problem on a 2D block-cyclic process grid (RCCL panel "
                    "broadcasts, strong scaling) instead of the default independent replicas; with one process the "
                    "logical ranks share the GPU (loopback transport)")
    ap.add_argument("--nb", type=int, default=512, help="tile edge of the block-cyclic layout")
    ap.add_argument("--sparse", action="store_true", help="BASELINE configs[4]: SparseGPRegression (VarDTC) N=200000 "
                    "rows per GPU, M=2048, D=16; rows sharded across GPUs with one RCCL all-reduce per pass")
    ap.add_argument("--m", type=int, default=2048, help="inducing points (--sparse)")
    args = ap.parse_args()
    if args.grid:
        return main_grid(args)
    if args.sparse:
        return main_sparse(args)

    comm = Comm()
    assert comm.world == max(1, args.gpus) or comm.world == 1, "launch with torch.distributed.run for --gpus > 1"
    from gpy_amd import _lib as L
    from gpy_amd.datasets import default_theta, synthetic

    ARD = not args.iso
    N, D = args.n, args.d
    X, Y = synthetic(N, D, seed=comm.rank)                    # every replica gets its own data set
    var, ls, noise = default_theta(D, ARD)
    theta = L.theta_vec(var, ls, ARD, D)
    ctx = L.Context(comm.local_rank)
    ctx.set_data(X, Y)
    ctx.set_option("profile", ("update_nt", "lauum"))         # hipEvent pairs around every k_update_nt / k_lauum launch
    last = {}

    def step():
        info, r = ctx.exact_inference(args.kind, ARD, theta, noise, want_alpha=False, want_stage_ms=True)
        assert info == 0
        last["r"] = r

    dt = timed_region(comm, step, args.steps, args.warmup)
    n_gpus = comm.world
    its = n_gpus * args.steps / dt
    if comm.rank == 0:
        r = last["r"]
        st = r["stage_ms"]
        pf = ctx.get_profile()                                # last timed step
        upd_ms, upd_flops, upd_n = pf["update_nt"]
        achieved = upd_flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        out = {
            "metric": "exact-GP log_lik+grad iters/sec", "value": its, "unit": "iters/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %s exact GP, one parameters_changed (K build + Cholesky + alpha + LML + Ky^-1 + "
                                   "all gradients), N=%d D=%d Dy=1 per GPU" % (args.kind, "ARD" if ARD else "iso", N, D),
                       "N": N, "D": D, "kernel": args.kind, "ARD": ARD, "parallelism": "replicas x%d" % n_gpus},
            "cholesky_gflops": (N ** 3 / 3.0) / (st["potrf"] * 1e-3) / 1e9,
            "cholesky_frac_of_fp64_peak": (N ** 3 / 3.0) / (st["potrf"] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
            "iteration_tflops": float(N) ** 3 / (st["total"] * 1e-3) / 1e12,
            "iteration_frac_of_fp64_peak": float(N) ** 3 / (st["total"] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
            "stage_ms": {k: round(float(v), 4) for k, v in st.items()},
            "roofline": {"bound": "mfma", "kernel": "k_update_nt (fp64 MFMA trailing update of the blocked Cholesky)",
                         "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_TFLOPS,
                         "traffic": profiled_traffic("k_update_nt") if (N, D, args.kind) == (16384, 32, "matern52")
                         else None,
                         "launches_per_step": upd_n, "avg_launch_ms": upd_ms / max(upd_n, 1),
                         "algorithmic_flops_per_step": upd_flops,
                         "note": ("launch durations include the time k_update_nt shares the GPU with the overlapped "
                                  "inverse of the leading block" +
                                  (" (MI355GP_TRI_OVERLAP=0: 0.57)" if (N, D, args.kind) == (16384, 32, "matern52") else ""))
                         if N >= 6144 else None},
            # the same tile-GEMM device routine in its single uncontended launch (W = X^T X, N^3/3 flops)
            "roofline_k_lauum": {"achieved": pf["lauum"][1] / (pf["lauum"][0] * 1e-3) / 1e12 if pf["lauum"][0] > 0 else 0.0,
                                 "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                                 "frac": (pf["lauum"][1] / (pf["lauum"][0] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS)
                                 if pf["lauum"][0] > 0 else 0.0},
            "lml": r["lml"],
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kind, ARD, D, min(args.cpu_sample_n, N), N)
        print(json.dumps(out), flush=True)
    ctx.close()
    comm.close()


def main_sparse(args):
    """BASELINE configs[4]: one SparseGP.parameters_changed (VarDTC + all gradients) per step; rows sharded over ranks
    (weak scaling: N rows PER GPU), Z / theta replicated, psi2 + gradient sums all-reduced over RCCL."""
    comm = Comm()
    from gpy_amd import _lib as L
    from gpy_amd import grid as G
    from gpy_amd.datasets import default_theta, synthetic
    n_per = 200000 if args.n == WORKLOAD["N"] else args.n
    D = 16 if args.d == WORKLOAD["D"] else args.d
    M, world = args.m, comm.world
    X, Y = synthetic(n_per * world, D, seed=0)                 # every rank generates the same global set ...
    Z = X[np.random.default_rng(1).permutation(X.shape[0])[:M]].copy()
    lo, hi = G.shard_rows(X.shape[0], comm.rank, world)        # ... and uploads only its shard
    var, ls, noise = default_theta(D, False)
    theta = L.theta_vec(var, ls, False, D)
    c = L.SparseContext(comm.local_rank)
    if world > 1:
        idb = G.unique_id() if comm.rank == 0 else b"\0" * G.ID_BYTES
        c.attach_comm(comm.rank, world, G.exchange_id_torch(idb, comm.rank))
    c.set_data(X[lo:hi], Y[lo:hi])
    last = {}

    def step():
        info, r = c.vardtc("rbf", False, theta, Z, noise, want_stage_ms=True)
        assert info == 0
        last["r"] = r

    dt = timed_region(comm, step, args.steps, args.warmup)
    if comm.rank == 0:
        r = last["r"]
        N = X.shape[0]
        flops = 3.0 * N * M * M                               # psi2 (lower half) N M^2 + Kfu dL_dpsi2 2 N M^2
        out = {"metric": "sparse-GP (VarDTC) log_lik+grad iters/sec", "value": args.steps / dt, "unit": "iters/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "RBF SparseGPRegression (VarDTC), one parameters_changed incl. kernel and "
                                      "inducing-input gradients, N=%d (%d per GPU) M=%d D=%d" % (N, n_per, M, D),
                          "N": N, "M": M, "D": D, "parallelism": "rows sharded x%d" % world},
               "rows_per_s": N * args.steps / dt, "gemm_tflops": flops / (dt / args.steps) / 1e12,
               "gemm_frac_of_fp64_peak": flops / (dt / args.steps) / 1e12 / (PEAK_FP64_TFLOPS * world),
               "stage_ms": {k: round(float(v), 3) for k, v in r["stage_ms"].items()}, "lml": r["lml"]}
        print(json.dumps(out), flush=True)
    c.close()
    comm.close()


def main_grid(args):
    """Optional mode (north_star config 4): all GPUs factor ONE N x N problem, 2D block-cyclic over Pr x Pc."""
    comm = Comm()
    from gpy_amd import _lib as L
    from gpy_amd import grid as G
    from gpy_amd.datasets import default_theta, synthetic
    Pr, Pc = (int(v) for v in args.grid.lower().split("x"))
    ARD = not args.iso
    N, D = args.n, args.d
    X, Y = synthetic(N, D, seed=0)                            # the same problem on every rank
    var, ls, noise = default_theta(D, ARD)
    theta = L.theta_vec(var, ls, ARD, D)
    if comm.world > 1:
        assert comm.world == Pr * Pc, "--grid PrxPc must match the number of processes"
        g = G.GridContext.from_env(Pr, Pc, args.nb)
    else:
        g = G.GridContext.loopback(Pr, Pc, args.nb, device=comm.local_rank)
    g.set_data(X, Y)
    last = {}

    def step():
        info, r = g.exact_inference(args.kind, ARD, theta, noise, want_stage_ms=True)
        assert info == 0
        last["r"] = r

    dt = timed_region(comm, step, args.steps, args.warmup)
    if comm.rank == 0:
        r = last["r"]
        its = args.steps / dt                                 # one problem, all GPUs
        flops = float(N) ** 3
        traffic = G.step_traffic_bytes(N, args.nb, Pr, Pc)
        out = {
            "metric": "exact-GP log_lik+grad iters/sec", "value": its, "unit": "iters/s", "n_gpus": comm.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %s exact GP, one parameters_changed on a %dx%d block-cyclic grid (nb=%d, %s "
                                   "transport), N=%d D=%d Dy=1" % (args.kind, "ARD" if ARD else "iso", Pr, Pc, args.nb,
                                                                  "loopback" if g.is_loopback else "RCCL", N, D),
                       "N": N, "D": D, "kernel": args.kind, "ARD": ARD, "parallelism": "grid %dx%d" % (Pr, Pc)},
            "iteration_tflops": flops / (dt / args.steps) / 1e12,
            "iteration_frac_of_fp64_peak": flops / (dt / args.steps) / 1e12 / (PEAK_FP64_TFLOPS * comm.world),
            "stage_ms": {k: round(float(v), 4) for k, v in r["stage_ms"].items()},
            "comm_bytes_per_step": traffic, "lml": r["lml"],
        }
        print(json.dumps(out), flush=True)
    g.close()
    comm.close()


if __name__ == "__main__":
    main()
