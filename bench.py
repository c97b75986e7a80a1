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

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).  Besides the headline (configs[2]) the
line carries one sub-record per other GPU configuration of BASELINE.json, each measured by a CHILD process with a timeout
(a fault there cannot take the headline down) and each with its own parity block against the committed golden fixture:
    "c2"             configs[1]  RBF N=4096 D=8 (one GPU; roofline + cpu_baseline)
    "grid"           configs[3]  RBF N=32768 D=8 on the 2D block-cyclic grid: 2x4 LOGICAL ranks over the loopback transport
                                 when the launch has one GPU (the real block-cyclic code, all ranks time-sharing the device),
                                 RCCL over all ranks of the launch otherwise
    "c5"             configs[4]  VarDTC N=200000 (per GPU) M=2048 D=16 (rows sharded over the ranks of the launch; roofline of
                                 its dominant MFMA GEMM + cpu_baseline from the sparse oracle)
    "c4_single"      configs[3]'s problem (RBF N=32768 D=8) on the dedicated SINGLE-GPU path (rank 0's GPU): the N = 1 anchor
                                 of the strong-scaling series, parity against the N=32768 golden
Every `cpu_baseline` of the line is measured on rank 0 AFTER the GPU legs have ended, one after the other (nothing else of the
run on the host): the headline configuration DIRECTLY at N=16384, configs[1] directly at N=4096, the sparse oracle at two sizes.
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
    """Rank plumbing for the bracketing barriers and the MAX-over-ranks.  Not on the data path.

    Default for world > 1: torch.distributed over GLOO with CPU tensors, and NO torch.cuda call anywhere in the process -- the
    library binds RCCL itself (dlopen, csrc/grid.hip) for the grid / sparse legs, and a process that also held PyTorch's own RCCL
    communicator and torch.cuda contexts would be a combination nothing has ever executed.  The device fence
    of the bracket is hipDeviceSynchronize through the library (mi355gp_device_synchronize).  MI355GP_BENCH_BACKEND=nccl opts
    into the rank plumbing over torch's NCCL (= RCCL) backend with CUDA tensors instead."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.dist = None
        self.device_tensor = False
        self.backend = None
        self._sync = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend is None:
                backend = os.environ.get("MI355GP_BENCH_BACKEND") or "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device_tensor = True
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            self.torch = torch
            self.backend = backend
            if os.environ.get("MI355GP_TRANSPORT") == "ipc":
                # dry run on a box with fewer GPUs than ranks (tools/multiproc_dryrun.sh): ranks share devices
                try:
                    from gpy_amd import _lib as _L
                    self.local_rank %= max(1, _L.device_count())
                except Exception:
                    pass

    def _sync_device(self):
        if self.device_tensor:
            self.torch.cuda.synchronize()
            return
        if self._sync is None:
            try:
                from gpy_amd import _lib as _L
                self._sync = _L.device_synchronize if _L.device_count() > self.local_rank else False
            except Exception:
                self._sync = False
        if self._sync:
            self._sync(self.local_rank)

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

    def gather(self, values):
        """[values of rank 0, values of rank 1, ...] (lists of floats of equal length) on every rank."""
        vals = [float(v) for v in values]
        if self.dist is None:
            return [vals]
        t = self.torch.tensor(vals, dtype=self.torch.float64, device="cuda" if self.device_tensor else "cpu")
        outs = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [[float(x) for x in o.cpu().tolist()] for o in outs]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed_region(comm, step_fn, steps, warmup, per_step=None):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync; returns MAX-over-ranks seconds.
    per_step (a list): this rank's host time of every timed step in ms (each step ends with a stream synchronize inside the
    C-ABI call) -- for runs of a handful of steps, where a slow first step is visible in the mean."""
    for _ in range(warmup):
        step_fn()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        step_fn()              # each step ends with a stream synchronize inside the C-ABI call
        if per_step is not None:
            per_step.append(round(1e3 * (time.perf_counter() - t1), 3))
    comm.barrier()
    dt = time.perf_counter() - t0
    return comm.max_over_ranks(dt)


def _cpu_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        return os.cpu_count() or 1


def _oracle_seconds(kind, ARD, D, n):
    from oracle import gp_oracle as O
    X, Y = O.synthetic(n, D, seed=0)
    var, ls, noise = O.default_theta(D, ARD)
    t0 = time.perf_counter()
    O.parameters_changed(kind, X, Y, var, ls, ARD, noise, cached=True)
    return time.perf_counter() - t0


def committed_cpu_record(kind, ARD, D, n_full):
    """profiles/*cpu_baseline*.json: full-size measurements committed from earlier runs (direct N = n_full timing of the
    port on a GPU box host; the reference's own code vs the port on the build container).  None if none matches."""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*cpu_baseline*.json"))):
        try:
            for rec in json.load(open(f)).get("records", []):
                if (rec.get("kind"), bool(rec.get("ARD")), rec.get("D"), rec.get("N")) == (kind, bool(ARD), D, n_full):
                    out.append(dict(rec, file=os.path.basename(f)))
        except Exception:
            pass
    return out or None


def cpu_baseline(kind, ARD, D, n_small, n_full, full=False, fit=True):
    """The oracle (NumPy/SciPy restatement of GPy's CPU path, oracle/gp_oracle.py; kind "port") on the host cores.
    full=True (the default run): ONE full iteration timed DIRECTLY at n_full -- `value` is that measurement; the two-point
    model t(N) = a N^2 + b N^3 through (n_small, 2 n_small) is kept as `fit_cross_check` only (its cubic coefficient wanders from
    run to run: the path is ~N^2 at the sample sizes -- a dozen single-threaded N x N NumPy passes and the serial ARD gradient
    loop; only dpotrf / dtrtri / dpotri are N^3 and multi-threaded).  full=False: the fit alone (quick runs)."""
    from oracle import gp_oracle as O
    try:
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    except Exception:
        pass
    _oracle_seconds(kind, ARD, D, 512)                                         # warm BLAS threads
    threads = _cpu_threads()
    rec = {"unit": "iters/s", "cores": int(threads), "kind": "port", "host_cores": os.cpu_count() or 1}
    direct = full or 2 * n_small >= n_full
    if direct:
        dt = _oracle_seconds(kind, ARD, D, n_full)
        rec.update(value=1.0 / dt, measured_seconds=dt, sample_N=n_full,
                   sample="1 full iteration of the oracle (port of GPy's CPU path) timed directly at N=%d D=%d: %.1f s, "
                          "%d BLAS threads of %d host cores" % (n_full, D, dt, threads, os.cpu_count() or 1))
    if (fit or not direct) and 2 * n_small < n_full:
        n1, n2 = n_small, 2 * n_small
        t1, t2 = _oracle_seconds(kind, ARD, D, n1), _oracle_seconds(kind, ARD, D, n2)
        # t = a N^2 + b N^3 through (n1, t1), (n2, t2)
        b = (t2 / n2 ** 2 - t1 / n1 ** 2) / (n2 - n1)
        a = t1 / n1 ** 2 - b * n1
        if b < 0.0:                      # noise at tiny sizes: fall back to the pure N^2 law through the larger sample
            a, b = t2 / n2 ** 2, 0.0
        if a < 0.0:
            a, b = 0.0, t2 / n2 ** 3
        est = a * n_full ** 2 + b * n_full ** 3
        fitrec = {"estimated_seconds": est, "a_N2": a, "b_N3": b, "n": [n1, n2], "seconds": [t1, t2]}
        if direct:
            rec["fit_cross_check"] = fitrec
        else:
            rec.update(value=1.0 / est, estimated_seconds=est, fit=fitrec,
                       sample="oracle (port) at N=%d (%.2f s) and N=%d (%.2f s), D=%d, %d BLAS threads of %d host cores; "
                              "t = a N^2 + b N^3 through both, evaluated at N=%d (%.0f s)" % (
                                  n1, t1, n2, t2, D, threads, os.cpu_count() or 1, n_full, est))
    committed = committed_cpu_record(kind, ARD, D, n_full)
    if committed:
        rec["committed_full_size"] = committed
    return rec


REFERENCE_RATIO_FILE = "profiles/r3_cpu_baseline_reference_cached.json"


def reference_ratio(kind, ARD, D, N):
    """port_cached / reference_cached seconds of the committed same-host comparison (tools/cpu_reference_vs_port.py on the build
    container, where /root/reference exists; the GPU box has no reference tree, so `cpu_baseline.kind` is "port"): how much the
    port UNDERSTATES the reference's own time for this configuration.  None when the configuration was not compared."""
    try:
        for rec in json.load(open(os.path.join(ROOT, REFERENCE_RATIO_FILE))).get("records", []):
            if (rec.get("kind"), bool(rec.get("ARD")), rec.get("D"), rec.get("N")) == (kind, bool(ARD), D, N):
                sec = rec["seconds"]
                return round(sec["port_cached"] / sec["reference_cached"], 3)
    except Exception:
        pass
    return None


def annotate_baseline(rec, gpu_value, kind, ARD, D, N):
    """What the CPU baseline is and is not: `speedup` = GPU value / the PORT's value on `cores` BLAS threads
    (reported, not credit); `reference_ratio` = the port's time / the reference's own time on one host (committed file)."""
    if isinstance(rec, dict) and rec.get("value"):
        rec["speedup"] = gpu_value / rec["value"]
        r = reference_ratio(kind, ARD, D, N)
        if r is not None:
            rec["reference_ratio"] = r
            rec["reference_ratio_source"] = REFERENCE_RATIO_FILE
    return rec


def timed_baseline(fn):
    """One CPU baseline, measured with NOTHING else of this run on the host (after the GPU legs have ended: a baseline timed
    underneath the legs' host loops competes with them for cores and perturbs the latency-bound small legs)."""
    t0 = time.perf_counter()
    try:
        rec = fn()
        rec["baseline_wall_s"] = round(time.perf_counter() - t0, 1)
        return rec
    except Exception as e:                                # noqa: BLE001 -- a failed baseline must not take the line down
        return {"error": repr(e)[-300:]}


def profiled_traffic(kernel_prefix, with_source=False):
    """HBM-side bytes per launch of the roofline kernel from the committed PMC summary (profiles/*_traffic.json, written
    by tools/summarize_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if no summary is committed."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))
                   if "sparse" not in f and "_c2_" not in f)
    hit = _latest_traffic(files, kernel_prefix)
    if with_source:
        return hit if hit else (None, None)
    return hit[0] if hit else None


def _latest_traffic(files, kernel_prefix):
    """(bytes per launch, file name) from the newest summary that HAS the kernel (a summary whose counter pass failed holds an
    empty table and must not shadow an older good one), or None."""
    for f in reversed(files):
        try:
            for name, v in json.load(open(f)).get("kernels", {}).items():
                if name.startswith(kernel_prefix) and v.get("hbm_bytes_per_launch"):
                    return v["hbm_bytes_per_launch"], os.path.basename(f)
        except Exception:
            pass
    return None


def profiled_c2_traffic(kernel_prefix):
    """HBM-side bytes per launch of a kernel of configs[1] (N=4096) from the committed PMC summary (profiles/*_c2_traffic.json);
    (bytes, file) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c2_traffic.json")))
    return _latest_traffic(files, kernel_prefix) or (None, None)


def profiled_sparse_traffic(kernel_prefix):
    """HBM-side bytes per launch of a sparse-path kernel from the committed PMC summary (profiles/*sparse_traffic.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*sparse_traffic.json")))
    hit = _latest_traffic(files, kernel_prefix)
    return hit[0] if hit else None


def parity_gate(kind, ARD, D, device):
    """no timing counts before parity.  A small same-kernel evaluation through the drop-in classes against
    the CPU oracle (N = 2048, the bench's own D / kernel / ARD).  Returns the observed errors; raises on a miss."""
    import gpy_amd
    from oracle import gp_oracle as O
    n = 2048
    X, Y = O.synthetic(n, D, seed=3)
    var, ls, noise = O.default_theta(D, ARD)
    ref = O.parameters_changed(kind, X, Y, var, ls, ARD, noise)
    cls = gpy_amd.kern.KERNEL_CLASSES[kind]
    m = gpy_amd.GPRegression(X, Y, cls(D, variance=var, lengthscale=ls, ARD=ARD, device=device), noise_var=noise,
                             device=device)
    gref = np.concatenate([[ref["dvar"]], ref["dlen"], [ref["dL_dnoise"]]])
    err = {"n": n, "lml_rel": abs(m.log_likelihood() - ref["lml"]) / abs(ref["lml"]),
           "alpha_rel": float(np.linalg.norm(m.posterior.woodbury_vector - ref["alpha"]) / np.linalg.norm(ref["alpha"])),
           "grad_rel": float(np.abs(m.gradient - gref).max() / np.abs(gref).max()), "against": "oracle (port)"}
    m.inference_method._state.ctx.close()
    if not (err["lml_rel"] <= 1e-10 and err["alpha_rel"] <= 1e-9 and err["grad_rel"] <= 1e-8):
        raise SystemExit("bench.py: parity gate failed before timing: %r" % (err,))
    return err


def golden_check(kind, ARD, N, D, seed, lml, alpha, grad):
    """If tests/golden holds the REFERENCE's outputs for exactly this workload (oracle/make_golden_baseline.py), compare
    the timed configuration's own results with them.  None when there is no such fixture."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "baseline_c*.npz")):
        g = np.load(f, allow_pickle=False)
        if "M" in g.files or (str(g["kind"]), bool(g["ARD"]), int(g["N"]), int(g["D"]), int(g["seed"])) != (
                kind, bool(ARD), N, D, seed):
            continue
        gref = np.concatenate([g["dvar"], g["dlen"], g["dnoise"]])
        err = {"fixture": os.path.basename(f), "against": str(g["source"]),
               "lml_rel": abs(lml - float(g["lml"])) / abs(float(g["lml"])),
               "alpha_rel": float(np.linalg.norm(alpha - g["alpha"]) / np.linalg.norm(g["alpha"])),
               "grad_rel": float(np.abs(grad - gref).max() / np.abs(gref).max())}
        if not (err["lml_rel"] <= 1e-10 and err["alpha_rel"] <= 1e-9 and err["grad_rel"] <= 1e-8):
            raise SystemExit("bench.py: the timed configuration disagrees with the reference golden: %r" % (err,))
        return err
    return None


def child_env(port_offset, **extra):
    """Environment of a child leg that keeps this rank's RANK / WORLD_SIZE / LOCAL_RANK (the children of all ranks form their own
    process group): MASTER_PORT moved by port_offset and EVERY TORCHELASTIC_* variable removed.  torch.distributed.run exports
    TORCHELASTIC_USE_AGENT_STORE=True to its workers, which makes env:// rendezvous expect the launcher's agent to host the store
    at MASTER_PORT: a child that inherits it only ever CONNECTS to the moved port, where nobody listens, and the leg hangs to its
    time-out (found by the 8-process dry run of round 4; tests/test_bench_dist.py covers it over gloo)."""
    env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset))
    nonce = "%s+%d" % (os.environ.get("TORCHELASTIC_RUN_ID", os.environ.get("MASTER_PORT", "job")), port_offset)
    for k in list(env):
        if k.startswith("TORCHELASTIC_"):
            env.pop(k)
    env["MI355GP_JOB_NONCE"] = nonce                        # what gpy_amd.grid's file exchange keys on when torch is not used
    env.update(extra)
    return env


def run_child(argv, timeout, env=None, single=True):
    """`python bench.py <argv>` in a child process; returns the parsed JSON line (or {"error": ...}).  single: the child is a
    one-process run of its own (rank variables of the parent launch are removed)."""
    import subprocess
    env = dict(os.environ if env is None else env)
    if single:
        for k in list(env):
            if k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK") or k.startswith("TORCHELASTIC_"):
                env.pop(k, None)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, capture_output=True, text=True,
                           timeout=timeout)
    except subprocess.TimeoutExpired as e:
        tail = e.stderr if isinstance(e.stderr, str) else (e.stderr or b"").decode("utf-8", "replace")
        return {"error": "timed out after %.0f s" % timeout, "stderr_tail": tail[-600:]}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": ("rc %d: " % r.returncode) + (r.stderr or r.stdout)[-400:]}
    try:
        rec = json.loads(lines[-1])
    except ValueError as e:
        return {"error": "unparsable child output: %s" % e}
    rec["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    return rec


C2_KEEP = ("ms_per_step", "step_ms", "value", "unit", "steps", "warmup", "config", "stage_ms", "iteration_tflops", "stage_sum_frac_of_fp64_peak",
           "iteration_frac_of_fp64_peak", "cholesky_gflops", "cholesky_frac_of_fp64_peak", "roofline", "roofline_k_lauum",
           "families", "parity_checked", "parity", "cpu_baseline", "host_path", "lml", "leg_wall_s", "error", "small_n_schedule", "persist_aborts")


def c2_leg(comm, args):
    """BASELINE configs[1]: RBF iso, N=4096, D=8 on ONE GPU (rank 0's), through the drop-in classes."""
    steps = ["--steps", "20", "--warmup", "5"] if args.dry_run_sizes else ["--steps", "300", "--warmup", "20"]
    rec = run_child(["--n", "4096", "--d", "8", "--kind", "rbf", "--iso"] + steps + ["--no-legs",
                     "--device", str(comm.local_rank), "--no-cpu-baseline"], timeout=240.0)
    return {k: rec[k] for k in C2_KEEP if k in rec}


def c4_single_leg(comm, args):
    """BASELINE configs[3]'s problem (RBF iso, N=32768, D=8) on the DEDICATED single-GPU path: the N = 1 anchor of the
    strong-scaling series north_star asks for (N in {4k, 16k, 32k} at 1/2/4/8 GPUs), next to `grid` (the same problem on the
    block-cyclic code).  26 GB resident; parity against the N=32768 golden."""
    rec = run_child(["--n", str(args.grid_n), "--d", "8", "--kind", "rbf", "--iso", "--steps", "4", "--warmup", "2", "--no-legs",
                     "--device", str(comm.local_rank), "--no-cpu-baseline"], timeout=420.0)
    return {k: rec[k] for k in C2_KEEP if k in rec}


def sparse_leg(comm, args, timeout=300.0):
    """BASELINE configs[4] over ALL ranks of this launch (rows sharded, one all-reduce per pass) -- every rank spawns its own
    child with its rank variables, like the grid leg."""
    env = child_env(29)
    argv = ["--sparse", "--steps", "12", "--warmup", "3", "--no-cpu-baseline"]
    if args.dry_run_sizes:
        argv = ["--sparse", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--n", "20000", "--m", "512"]
    if comm.world == 1:
        argv += ["--device", str(comm.local_rank)]
    rec = run_child(argv, timeout, env=env, single=comm.world == 1)
    if comm.rank != 0:
        return None
    rec.pop("metric", None)
    return rec


GRID_LEGS = {
    # name: (N, D, kind, iso, steps, warmup, MASTER_PORT offset of the children's own process group)
    "grid":    (None, 8, "rbf", True, 3, 1, 17),               # BASELINE configs[3]: N = --grid-n (32768)
    "grid16k": (16384, 32, "matern52", False, 4, 1, 19),       # the headline problem (configs[2]) strong-scaled over the launch
    "grid4k":  (4096, 8, "rbf", True, 10, 2, 23),              # configs[1] strong-scaled over the launch
}


def grid_leg(comm, args, name="grid", timeout=180.0):
    """One strong-scaling leg: ONE problem on the 2D block-cyclic grid over ALL ranks of this launch (grid_shape(world);
    loopback transport when there is one process), run in CHILD processes with a timeout so that a fault of the
    never-before-timed multi-GPU path cannot take the headline line down.  north_star asks for N in {4k, 16k, 32k} at 1/2/4/8
    GPUs: "grid" is BASELINE configs[3] (RBF, N=32768, D=8), "grid16k" the headline problem (configs[2]: Matern-5/2 ARD, D=32),
    "grid4k" configs[1]; each checks its result against the golden of that configuration.  Returns the sub-record (rank 0)."""
    import subprocess
    from gpy_amd import grid as G
    N, D, kind, iso, steps, warmup, port = GRID_LEGS[name]
    N = args.grid_n if N is None else (N if not args.dry_run_sizes else min(N, args.grid_n))
    if args.dry_run_sizes:
        steps, warmup = 2, 1
    # one GPU: the 2 x 4 grid of configs[3] as LOGICAL ranks over the loopback transport -- the block-cyclic code itself
    # (tiles, panel broadcasts as device copies, look-ahead streams), not the degenerate 1 x 1 grid
    Pr, Pc = G.grid_shape(comm.world) if comm.world > 1 else (2, 4)
    out_path = os.path.join(ROOT, "gpurun_out", "%s_leg_%s_%d.json" % (name, os.environ.get("MASTER_PORT", "0"), os.getpid()))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    env = child_env(port, MI355GP_GRID_LEG_OUT=out_path if comm.rank == 0 else "")
    cmd = [sys.executable, os.path.abspath(__file__), "--grid", "%dx%d" % (Pr, Pc), "--n", str(N), "--d", str(D),
           "--kind", kind, "--steps", str(steps), "--warmup", str(warmup), "--nb", str(args.nb), "--grid-child"]
    if iso:
        cmd.append("--iso")
    rec = {"workload": "%s %s exact GP N=%d D=%d, one parameters_changed on a %dx%d block-cyclic grid (%s)" % (
        kind, "iso" if iso else "ARD", N, D, Pr, Pc, "RCCL, one rank per GPU" if comm.world > 1 else
        "8 logical ranks time-sharing ONE GPU over the loopback transport: the per-rank code path of the multi-GPU mode, no xGMI")}
    if comm.world == 1:
        cmd += ["--device", str(comm.local_rank)]
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        ok = r.returncode == 0
        err = (r.stderr or r.stdout)[-400:]
    except subprocess.TimeoutExpired as e:
        tail = e.stderr if isinstance(e.stderr, str) else (e.stderr or b"").decode("utf-8", "replace")
        ok, err = False, "timed out after %.0f s; stderr: %s" % (timeout, tail[-500:])
    if comm.rank != 0:
        return None
    if ok and os.path.exists(out_path):
        rec.update(json.load(open(out_path)))
        os.unlink(out_path)
    else:
        rec["error"] = err
    return rec


# ---- the printed line -------------------------------------------------------------------------------------------------------
# The driver keeps an 8 KB tail of stdout: the line is kept under 6 KB so that every leg is in it.  Prose (what a sample was,
# what a kernel does, committed full-size records) lives in DESIGN.md section 5 and profiles/; the untrimmed record of a run goes
# to gpurun_out/bench_full.json.
LINE_DROP = ("note", "what", "committed_full_size", "fit", "fit_cross_check", "cholesky_standalone", "host_cores", "baseline_wall_s",
             "abi_steps", "drop_in_ms_per_step", "algorithmic_flops_per_step", "k_update_nt64", "against", "fixture",
             "cholesky_frac_of_fp64_peak", "iteration_tflops", "rows_per_s", "gemm_tflops", "higher_is_better", "vs_baseline")
LEG_DROP = ("config", "families", "host_path", "roofline_k_lauum", "unit", "steps", "warmup", "value", "dtype", "data", "scaling",
            "n_gpus", "parity_checked", "workload", "sample", "sample_N", "comm_bytes_per_step", "leg_wall_s")


def _slim(o, drop, sig=5):
    if isinstance(o, dict):
        return {k: _slim(v, drop, sig) for k, v in o.items() if k not in drop}
    if isinstance(o, (list, tuple)):
        return [_slim(v, drop, sig) for v in o]
    if isinstance(o, float):
        return float("%.*g" % (sig, o))
    return o


def compact_line(out):
    """The one JSON line of the run: the headline record trimmed of prose, the legs trimmed to their numbers, and -- last, so that
    it survives any tail cut -- `legs_ms`, every leg's ms per step."""
    full_path = os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(full_path), exist_ok=True)
        with open(full_path, "w") as f:
            json.dump(out, f)
    except OSError:
        pass
    line = {}
    legs = ("c2", "grid", "grid16k", "grid4k", "c5", "c4_single")
    for k, v in out.items():
        if k in legs:
            continue
        line[k] = _slim(v, LINE_DROP) if k not in ("higher_is_better", "vs_baseline") else v
    line["higher_is_better"], line["vs_baseline"] = out.get("higher_is_better", True), out.get("vs_baseline")
    if isinstance(line.get("families"), dict):               # name -> [summed launch ms, launches, TFLOP/s]
        line["families"] = {k: [round(v["ms"], 3), v["launches"], round(v["tflops"], 2)] for k, v in out["families"].items()}
    for leg in legs:
        if isinstance(out.get(leg), dict):                    # LEG_DROP at the top level of a leg only (its cpu_baseline keeps `value`)
            line[leg] = _slim({k: v for k, v in out[leg].items() if k not in LEG_DROP}, LINE_DROP + ("sample", "sample_N", "unit", "reference_ratio_source"))
    line["legs_ms"] = {leg: (round(out[leg]["ms_per_step"], 4) if isinstance(out.get(leg), dict) and "ms_per_step" in out[leg]
                             else (out.get(leg) or {}).get("error", None) and "error") for leg in legs if leg in out}
    return json.dumps(line, separators=(",", ":"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=70, help="timed steps (default: >= 5 s of timed work at the default workload)")
    ap.add_argument("--warmup", type=int, default=3)
    # (--size / --dims: aliases that are no prefix of a torch.distributed.run option -- its argparse rejects "--n" as ambiguous)
    ap.add_argument("--n", "--size", dest="n", type=int, default=WORKLOAD["N"])
    ap.add_argument("--d", "--dims", dest="d", type=int, default=WORKLOAD["D"])
    ap.add_argument("--kind", default=WORKLOAD["kind"])
    ap.add_argument("--iso", action="store_true", help="single lengthscale instead of ARD")
    ap.add_argument("--cpu-sample-n", type=int, default=3072, help="the CPU baseline is timed at this N and at twice it")
    ap.add_argument("--cpu-full", action="store_true", help="(default since round 4) time the CPU baseline directly at the full N")
    ap.add_argument("--cpu-fit-only", action="store_true", help="quick runs: CPU baseline from the two-point fit only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true")
    ap.add_argument("--no-grid-leg", action="store_true", help="skip the block-cyclic sub-record (configs[3])")
    ap.add_argument("--no-legs", action="store_true", help="headline only: no c2 / grid / c5 sub-records")
    ap.add_argument("--device", type=int, default=-1, help="HIP device of a one-process run (default: LOCAL_RANK)")
    ap.add_argument("--grid-n", type=int, default=32768)
    ap.add_argument("--grid-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run-sizes", action="store_true", help="small child legs (c2 20 steps, c5 20000 rows x 512 inducing "
                    "points, grid / c4_single at --grid-n): the multi-rank dry run of tools/multiproc_dryrun.sh on one GPU")
    ap.add_argument("--abi-only", action="store_true", help="time the bare C-ABI call instead of the drop-in classes")
    ap.add_argument("--grid", default="", help="PrxPc: ONE problem on a 2D block-cyclic process grid (RCCL panel "
                    "broadcasts, strong scaling) instead of the default independent replicas; with one process the "
                    "logical ranks share the GPU (loopback transport)")
    ap.add_argument("--nb", type=int, default=512, help="tile edge of the block-cyclic layout")
    ap.add_argument("--sparse", action="store_true", help="BASELINE configs[4]: SparseGPRegression (VarDTC) N=200000 "
                    "rows per GPU, M=2048, D=16; rows sharded across GPUs with one RCCL all-reduce per pass")
    ap.add_argument("--m", "--inducing", dest="m", type=int, default=2048, help="inducing points (--sparse)")
    args = ap.parse_args()
    if args.grid:
        return main_grid(args)
    if args.sparse:
        return main_sparse(args)

    comm = Comm()
    assert comm.world == max(1, args.gpus) or comm.world == 1, "launch with torch.distributed.run for --gpus > 1"
    if args.device >= 0 and comm.world == 1:
        comm.local_rank = args.device
    if args.no_legs:
        args.no_grid_leg = True
    import gpy_amd
    from gpy_amd import _lib as L
    from gpy_amd.datasets import default_theta, synthetic

    ARD = not args.iso
    N, D = args.n, args.d
    parity = None
    if not args.no_parity_gate:
        parity = parity_gate(args.kind, ARD, D, comm.local_rank)       # every rank gates its own device
    X, Y = synthetic(N, D, seed=comm.rank)                    # every replica gets its own data set
    var, ls, noise = default_theta(D, ARD)
    theta = L.theta_vec(var, ls, ARD, D)
    # ---- the drop-in path: GPRegression.parameters_changed through the reference's three call signatures ------------
    kern = gpy_amd.kern.KERNEL_CLASSES[args.kind](D, variance=var, lengthscale=ls, ARD=ARD, device=comm.local_rank)
    m = gpy_amd.GPRegression(X, Y, kern, noise_var=noise, device=comm.local_rank)
    ctx = m.inference_method._state.ctx
    x0 = m.param_array.copy()
    last = {}

    def step_model():
        # what an optimiser does: write the parameter vector, which runs parameters_changed (log_likelihood + all gradients)
        m.param_array = x0
        last["lml"], last["grad"] = m.log_likelihood(), m.gradient

    def step_abi(want_stage_ms=False):
        info, r = ctx.exact_inference(args.kind, ARD, theta, noise, want_alpha=False, want_stage_ms=want_stage_ms)
        assert info == 0
        last["r"] = r

    per_step = [] if args.steps <= 8 else None
    dt = timed_region(comm, step_abi if args.abi_only else step_model, args.steps, args.warmup, per_step)
    n_gpus = comm.world
    its = n_gpus * args.steps / dt
    out = None
    if comm.rank == 0:
        lml = last["r"]["lml"] if args.abi_only else last["lml"]
        # stage breakdown: one more evaluation with the stage events recorded (the timed steps run without them; below the
        # overlapped-inverse threshold they replay the factorisation from a hipGraph, which carries no timing events)
        step_abi(want_stage_ms=True)
        st = last["r"]["stage_ms"]
        aborts_timed = ctx.get_option("persist_aborts")       # persistent launches called off / aborted so far (each redone on launches)
        # roofline leg: the same step once more with hipEvent pairs around every k_update_nt / k_lauum launch (the timed
        # region above runs without them: ~2 us per bracketed launch)
        for attempt in range(2):
            before = ctx.get_option("persist_aborts")
            ctx.set_option("profile", 1)
            step_abi()
            step_abi()
            pf = ctx.get_profile()
            ctx.set_option("profile", 0)
            if ctx.get_option("persist_aborts") == before:
                break
            # a persistent launch was called off at its co-residency gate inside the bracketed steps (redone on launches inside the
            # same call): the context stays on launches for PS_SKIP_AFTER_CLEAN evaluations -- sit them out, bracket again
            for _ in range(ctx.get_option("persist_skip") if ctx.get_option("persist_skip") < 64 else 0):
                step_abi()
        # every bracketed kernel family of ONE evaluation: summed launch ms, launches, algorithmic flops (the chain kernels
        # k_diag128 / k_trsm128 run on the panel stream underneath the updates: their sum is not wall time)
        families = {k: {"ms": round(v[0], 4), "launches": v[2], "flops": v[1],
                        "tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else 0.0} for k, v in pf.items() if v[2] > 0}
        if "trtri" in families:
            # the triangular inverse runs partly UNDERNEATH potrf (leading block + its share of T21 on the CU-masked side
            # stream: "trtri_early", elapsed on that stream) and partly after it ("exposed"): N^3/3 flops over BOTH, so that no
            # family is billed more flops than the time it was given (round 3 divided all of them by the exposed part only)
            early = families.pop("trtri_early", None)
            t = families["trtri"]
            ov_ms = early["ms"] if early else 0.0
            t.update(exposed_ms=t["ms"], overlapped_ms=round(ov_ms, 4), ms=round(t["ms"] + ov_ms, 4),
                     tflops=t["flops"] / ((t["ms"] + ov_ms) * 1e-3) / 1e12 if t["ms"] + ov_ms > 0 else 0.0,
                     note="ms = exposed_ms (after potrf, main stream) + overlapped_ms (side stream underneath potrf)")
        upd_ms, upd_flops, upd_n = pf["update_nt"]
        roof_kernel = "k_update_nt<4, true>"       # fp64 MFMA trailing update of the blocked Cholesky, 128 x 128 tiles, in situ
        if upd_n == 0 and pf.get("potrf_persist", (0, 0, 0))[2] > 0:
            # small N: the whole factorisation is ONE persistent dataflow launch (persist.hip) -- it IS the dominant kernel;
            # algorithmic flops N^3/3, bound by the fp64 MFMA pipe in principle, by its one-CU chain in practice
            upd_ms, upd_flops, upd_n = pf["potrf_persist"]
            roof_kernel = "k_potrf_persist"
        achieved = upd_flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        traffic = traffic_source = None
        if (N, D, args.kind) == (16384, 32, "matern52"):
            traffic, traffic_source = profiled_traffic("k_update_nt<", True)
        elif (N, D, args.kind) == (4096, 8, "rbf"):
            traffic, traffic_source = profiled_c2_traffic("k_potrf_persist" if pf.get("update_nt", (0, 0, 0))[2] == 0 else "k_update_nt<")
        traffic_source = ("profiles/%s" % traffic_source) if traffic_source else None
        out = {
            "metric": "exact-GP log_lik+grad iters/sec", "value": its, "unit": "iters/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %s %s exact GP, one GPRegression.parameters_changed, N=%d D=%d Dy=1 per GPU" % (
                           args.kind, "ARD" if ARD else "iso", N, D),
                       "N": N, "D": D, "kernel": args.kind, "ARD": ARD, "parallelism": "replicas x%d" % n_gpus,
                       "path": "C-ABI" if args.abi_only else "drop-in classes"},
            # "cholesky_gflops" (the metric's Cholesky GF/s, filled in below) is the factorisation timed ALONE on a resident
            # matrix; the pipeline's potrf stage also hosts the overlapped inverse kernels and is only reported as stage_ms
            # N^3 flops over the WALL CLOCK of a timed step; the device-side stage sum of one evaluation under its own name
            "iteration_tflops": float(N) ** 3 / (dt / args.steps) / 1e12,
            "iteration_frac_of_fp64_peak": float(N) ** 3 / (dt / args.steps) / 1e12 / PEAK_FP64_TFLOPS,
            "stage_sum_frac_of_fp64_peak": float(N) ** 3 / (st["total"] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
            "stage_ms": {k: round(float(v), 4) for k, v in st.items()},
            "roofline": {"bound": "mfma", "kernel": roof_kernel,
                         "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_TFLOPS,
                         # HBM-side bytes per launch from the PMC passes of THIS command (rocprofv3 cannot run inside the
                         # timed process): the committed summary of the shipped schedule, named in traffic_source
                         "traffic": traffic, "traffic_source": traffic_source,
                         "launches_per_step": upd_n, "avg_launch_ms": upd_ms / max(upd_n, 1),
                         "algorithmic_flops_per_step": upd_flops,
                         "k_update_nt64": {"launches_per_step": pf["update_nt64"][2],
                                           "avg_launch_ms": pf["update_nt64"][0] / max(pf["update_nt64"][2], 1),
                                           "algorithmic_flops_per_step": pf["update_nt64"][1]}},
            # the same tile-GEMM device routine in its single uncontended launch (W = X^T X, N^3/3 flops)
            "roofline_k_lauum": {"achieved": pf["lauum"][1] / (pf["lauum"][0] * 1e-3) / 1e12 if pf["lauum"][0] > 0 else 0.0,
                                 "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                                 "frac": (pf["lauum"][1] / (pf["lauum"][0] * 1e-3) / 1e12 / PEAK_FP64_TFLOPS)
                                 if pf["lauum"][0] > 0 else 0.0},
            "families": families,
            "lml": lml,
        }
        if per_step:
            out["step_ms"] = per_step
        out["persist_aborts"] = {"timed_steps": aborts_timed, "profile_steps": ctx.get_option("persist_aborts") - aborts_timed}
        if N <= 128 * 36:
            # which schedule of a small factorisation the context's own timing settled on (boxes differ by 2x on the persistent
            # launch): the timed steps above ran on it
            out["small_n_schedule"] = ("undecided", "persistent launch", "launches")[ctx.get_option("persist_sched")]
        if not args.abi_only:
            # host overhead of the drop-in classes: the same evaluation through the bare C-ABI, same context, same data
            k2 = max(3, min(args.steps, 10))
            for _ in range(3):                   # plain, capture, first replay (the profile leg above dropped the graph)
                step_abi()
            t0 = time.perf_counter()
            for _ in range(k2):
                step_abi()
            abi_ms = 1e3 * (time.perf_counter() - t0) / k2
            out["host_path"] = {"drop_in_ms_per_step": out["ms_per_step"], "abi_ms_per_step": abi_ms,
                                "overhead_frac": out["ms_per_step"] / abi_ms - 1.0, "abi_steps": k2}
            step_model()
        # the Cholesky ALONE (what a jitchol / lapack.dpotrf call costs, util/linalg.py:56-75): device-resident synthetic SPD
        # matrix of the same N, factorisation only, without the leading inverse that the pipeline overlaps with it
        old = os.environ.get("MI355GP_TRI_OVERLAP")
        os.environ["MI355GP_TRI_OVERLAP"] = "0"
        try:
            bf = L.bench_factor(N, reps=3, device=comm.local_rank)
        finally:
            if old is None:
                os.environ.pop("MI355GP_TRI_OVERLAP", None)
            else:
                os.environ["MI355GP_TRI_OVERLAP"] = old
        out["cholesky_gflops"] = 1e3 * bf["potrf_tflops"]
        out["cholesky_frac_of_fp64_peak"] = bf["potrf_tflops"] / PEAK_FP64_TFLOPS
        out["cholesky_standalone"] = {"ms": bf["potrf_ms"], "gflops": 1e3 * bf["potrf_tflops"],
                                      "frac_of_fp64_peak": bf["potrf_tflops"] / PEAK_FP64_TFLOPS,
                                      "note": "the factorisation timed alone on a resident SPD matrix (mi355gp_bench_factor), "
                                              "no overlapped inverse"}
        out["parity_checked"] = parity is not None
        if parity is not None:
            out["parity"] = {"gate": parity}
            if not args.no_legs:
                # the N=32768 legs (grid, c4_single) compare with a golden the reference itself could not produce here (its pdinv
                # needs ~8 live N^2 temporaries; OpenBLAS dpotrf segfaults at this size in the image): a two-hop pin
                out["parity"]["c4_golden_source"] = "lean oracle, pinned to gp_oracle (tests/test_oracle_baseline.py), which is pinned to the reference"
            if not args.abi_only:
                g = golden_check(args.kind, ARD, N, D, 0, last["lml"], m.posterior.woodbury_vector, last["grad"])
                if g is not None:
                    out["parity"]["timed_config_vs_reference"] = g
    ctx.close()
    del m
    if not args.no_legs:
        comm.barrier()
        if comm.rank == 0:
            out["c2"] = c2_leg(comm, args)                    # configs[1], one GPU
        comm.barrier()
    if not args.no_grid_leg:
        # strong scaling of ONE problem over all ranks of the launch, N in {32768, 16384, 4096} (north_star)
        for name, tmo in (("grid", 240.0), ("grid16k", 180.0), ("grid4k", 120.0)):
            comm.barrier()
            rec = grid_leg(comm, args, name, timeout=tmo)
            if out is not None:
                out[name] = rec
    if not args.no_legs:
        comm.barrier()
        rec = sparse_leg(comm, args)                          # configs[4], rows sharded over all ranks of the launch
        if out is not None:
            out["c5"] = rec
    if not args.no_legs:
        comm.barrier()
        if comm.rank == 0:
            out["c4_single"] = c4_single_leg(comm, args)      # configs[3]'s problem on the dedicated single-GPU path (rank 0's GPU)
        comm.barrier()
    if comm.rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            # the CPU baselines, one after the other, after every GPU leg has ended (nothing else of this run uses the host)
            out["cpu_baseline"] = annotate_baseline(timed_baseline(lambda: cpu_baseline(
                args.kind, ARD, D, min(args.cpu_sample_n, N // 2), N, full=not args.cpu_fit_only, fit=False)), out["value"], args.kind, ARD, D, N)
            # vs_baseline stays null: BASELINE.md holds no published number for this metric.  The ratio to the CPU baseline is
            # cpu_baseline.speedup -- against the PORT on cpu_baseline.cores BLAS threads; the reference's own code takes
            # 1 / cpu_baseline.reference_ratio times as long on one host.
            out["vs_baseline_why"] = "no published number; see cpu_baseline.speedup (vs port) and .reference_ratio"
            if not args.no_legs:
                if isinstance(out.get("c2"), dict) and "error" not in out["c2"]:
                    out["c2"]["cpu_baseline"] = annotate_baseline(timed_baseline(lambda: cpu_baseline(
                        "rbf", False, 8, 2048, 4096, full=True, fit=False)), out["c2"].get("value", 0.0), "rbf", False, 8, 4096)
                if isinstance(out.get("c5"), dict) and "error" not in out["c5"]:
                    out["c5"]["cpu_baseline"] = timed_baseline(lambda: sparse_cpu_baseline(16, args.m, 200000))
        if args.dry_run_sizes:
            out["dry_run_sizes"] = True
        print(compact_line(out), flush=True)
    comm.close()


def sparse_cpu_baseline(D, M, n_full, n_small=4000):
    """The sparse oracle (NumPy/SciPy restatement of GPy's VarDTC + SparseGP._update_gradients, oracle/sparse_oracle.py;
    kind "port") on the host cores at TWO bounded sizes; the path is a + b N in the number of rows at fixed M (three N-column
    dtrtrs, one syrk and one GEMM of M^2 N each, N M elementwise passes; the M^3 algebra is the constant), fitted through
    both and evaluated at n_full."""
    from oracle import gp_oracle as O
    from oracle import sparse_oracle as S
    var, ls, noise = O.default_theta(D, False)

    def seconds(n):
        X, Y = O.synthetic(n, D, seed=0)
        Z = S.synthetic_Z(X, M, 0)
        t0 = time.perf_counter()
        S.vardtc("rbf", X, Z, Y, var, ls, False, noise)
        return time.perf_counter() - t0
    seconds(600)                                               # warm BLAS threads
    threads = _cpu_threads()
    n1, n2 = n_small, 2 * n_small
    t1, t2 = seconds(n1), seconds(n2)
    b = max((t2 - t1) / (n2 - n1), 0.0)
    a = max(t1 - b * n1, 0.0)
    if b == 0.0:
        a, b = 0.0, t2 / n2
    est = a + b * n_full
    rec = {"unit": "iters/s", "cores": int(threads), "kind": "port", "host_cores": os.cpu_count() or 1, "value": 1.0 / est,
           "estimated_seconds": est, "fit": {"a": a, "b_per_row": b, "n": [n1, n2], "seconds": [t1, t2]},
           "sample": "sparse oracle (port of GPy's VarDTC) at N=%d (%.2f s) and N=%d (%.2f s), M=%d D=%d, %d BLAS threads of %d "
                     "host cores; t = a + b N through both, evaluated at N=%d (%.0f s)" % (
                         n1, t1, n2, t2, M, D, threads, os.cpu_count() or 1, n_full, est)}
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "baseline_c5_sparse_rbf_n200000_m2048_d16.npz"), allow_pickle=False)
        if (int(g["N"]), int(g["M"]), int(g["D"])) == (n_full, M, D):
            rec["committed_full_size"] = {"kind": "reference", "seconds": float(g["seconds"]), "host_cores": 8,
                                          "what": "the REFERENCE'S OWN VarDTC + SparseGP._update_gradients at the full size on "
                                                  "the 8-core build container (oracle/make_golden_baseline.py, stored in the "
                                                  "golden fixture)"}
    except Exception:
        pass
    return rec


def sparse_golden_check(N, M, D, r):
    """tests/golden/baseline_c5_*.npz holds the REFERENCE's outputs for exactly configs[4] (seed 0)."""
    f = os.path.join(ROOT, "tests", "golden", "baseline_c5_sparse_rbf_n200000_m2048_d16.npz")
    if not os.path.exists(f):
        return None
    g = np.load(f, allow_pickle=False)
    if (int(g["N"]), int(g["M"]), int(g["D"])) != (N, M, D):
        return None
    err = {"fixture": os.path.basename(f), "against": str(g["source"]),
           "lml_rel": abs(r["lml"] - float(g["lml"])) / abs(float(g["lml"])),
           "dtheta_rel": float(np.abs(r["dtheta"] - g["dtheta"]).max() / np.abs(g["dtheta"]).max()),
           "dnoise_rel": abs(r["dnoise"] - float(g["dnoise"])) / abs(float(g["dnoise"])),
           "dZ_rel": float(np.abs(r["dZ"] - g["dZ"]).max() / np.abs(g["dZ"]).max()),
           "woodbury_vector_rel": float(np.linalg.norm(r["woodbury_vector"] - g["woodbury_vector"]) /
                                        np.linalg.norm(g["woodbury_vector"]))}
    if not (err["lml_rel"] <= 1e-9 and err["dtheta_rel"] <= 1e-6 and err["dnoise_rel"] <= 1e-6 and err["dZ_rel"] <= 1e-6):
        raise SystemExit("bench.py --sparse: the timed configuration disagrees with the reference golden: %r" % (err,))
    return err


def main_sparse(args):
    """BASELINE configs[4]: one SparseGP.parameters_changed (VarDTC + all gradients) per step; rows sharded over ranks
    (weak scaling: N rows PER GPU), Z / theta replicated, psi2 + gradient sums all-reduced over RCCL."""
    comm = Comm()
    if args.device >= 0 and comm.world == 1:
        comm.local_rank = args.device
    from gpy_amd import _lib as L
    from gpy_amd import grid as G
    from gpy_amd.datasets import default_theta, synthetic, synthetic_Z
    n_per = 200000 if args.n == WORKLOAD["N"] else args.n
    D = 16 if args.d == WORKLOAD["D"] else args.d
    M, world = args.m, comm.world
    X, Y = synthetic(n_per * world, D, seed=0)                 # every rank generates the same global set ...
    Z = synthetic_Z(X, M, 0)                                   # (the draw of the golden fixture)
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
        pf = c.get_profile()                                  # launch timing of the two MFMA kernels of the last timed step
        gm_ms, gm_fl, gm_n = pf["gemm_T"]
        gr_ms, gr_fl, gr_n = pf["gram_psi2"]
        out = {"metric": "sparse-GP (VarDTC) log_lik+grad iters/sec", "value": args.steps / dt, "unit": "iters/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "RBF SparseGPRegression (VarDTC), one parameters_changed incl. kernel and "
                                      "inducing-input gradients, N=%d (%d per GPU) M=%d D=%d" % (N, n_per, M, D),
                          "N": N, "M": M, "D": D, "parallelism": "rows sharded x%d" % world},
               "rows_per_s": N * args.steps / dt, "gemm_tflops": flops / (dt / args.steps) / 1e12,
               "gemm_frac_of_fp64_peak": flops / (dt / args.steps) / 1e12 / (PEAK_FP64_TFLOPS * world),
               "stage_ms": {k: round(float(v), 3) for k, v in r["stage_ms"].items()}, "lml": r["lml"],
               # dominant kernel of the path: T = Kfu dL_dpsi2 (var_dtc.py:231), 2 N M^2 flops on this rank's rows
               "roofline": {"bound": "mfma", "kernel": "k_gemm_full<true, false, 4>",     # T = Kfu dL_dpsi2 of pass 2, K = M
                            "achieved": gm_fl / (gm_ms * 1e-3) / 1e12 if gm_ms > 0 else 0.0, "peak": PEAK_FP64_TFLOPS,
                            "unit": "TFLOP/s", "frac": (gm_fl / (gm_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS) if gm_ms > 0 else 0.0,
                            "traffic": profiled_sparse_traffic("k_gemm_full<true, false"),
                            "launches_per_step": gm_n, "avg_launch_ms": gm_ms / max(gm_n, 1),
                            "algorithmic_flops_per_step": gm_fl,
                            "k_gram_splitk": {"launches_per_step": gr_n, "avg_launch_ms": gr_ms / max(gr_n, 1),
                                              "algorithmic_flops_per_step": gr_fl,
                                              "frac": (gr_fl / (gr_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS) if gr_ms > 0 else 0.0}}}
        if world == 1:
            gc = sparse_golden_check(N, M, D, r)
            out["parity_checked"] = gc is not None
            if gc is not None:
                out["parity"] = {"timed_config_vs_reference": gc}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = sparse_cpu_baseline(D, M, N)
        print(json.dumps(out), flush=True)
    c.close()
    comm.close()


def main_grid(args):
    """Optional mode (north_star config 4): all GPUs factor ONE N x N problem, 2D block-cyclic over Pr x Pc."""
    comm = Comm()
    if args.device >= 0 and comm.world == 1:
        comm.local_rank = args.device
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
        g = G.GridContext.from_env(Pr, Pc, args.nb, device=comm.local_rank)
    else:
        g = G.GridContext.loopback(Pr, Pc, args.nb, device=comm.local_rank)
    g.set_data(X, Y)
    last = {}

    def step():
        info, r = g.exact_inference(args.kind, ARD, theta, noise, want_stage_ms=True)
        assert info == 0
        last["r"] = r

    dt = timed_region(comm, step, args.steps, args.warmup)
    stage_keys = ("kbuild", "factor", "solve", "grad", "total")
    per_rank = comm.gather([last["r"]["stage_ms"][k] for k in stage_keys])     # where a real node's ranks differ
    transport = "loopback" if g.is_loopback else ("hipIpc (ranks = processes sharing a GPU)" if os.environ.get("MI355GP_TRANSPORT") == "ipc"
                                                  else "RCCL")
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
                                                                  transport, N, D),
                       "N": N, "D": D, "kernel": args.kind, "ARD": ARD, "parallelism": "grid %dx%d" % (Pr, Pc)},
            "iteration_tflops": flops / (dt / args.steps) / 1e12,
            "iteration_frac_of_fp64_peak": flops / (dt / args.steps) / 1e12 / (PEAK_FP64_TFLOPS * comm.world),
            "stage_ms": {k: round(float(v), 4) for k, v in r["stage_ms"].items()},
            "comm_bytes_per_step": traffic, "lml": r["lml"],
        }
        if len(per_rank) > 1:                                 # [min, max] over the ranks of the launch, per stage
            out["stage_ms_rank_min_max"] = {k: [round(min(p[i] for p in per_rank), 3), round(max(p[i] for p in per_rank), 3)]
                                            for i, k in enumerate(stage_keys)}
        gc = golden_check(args.kind, ARD, N, D, 0, r["lml"], r["alpha"],
                          np.concatenate([r["dtheta"], [r["dnoise"]]]))
        if gc is not None:
            out["parity_vs_golden"] = gc
        leg = os.environ.get("MI355GP_GRID_LEG_OUT", "")
        if leg:                                               # child of the default bench: hand the sub-record back
            keep = ("ms_per_step", "value", "n_gpus", "scaling", "iteration_tflops", "iteration_frac_of_fp64_peak",
                    "stage_ms", "stage_ms_rank_min_max", "comm_bytes_per_step", "lml", "parity_vs_golden", "steps", "warmup")
            sub = {k: out[k] for k in keep if k in out}
            sub.update(grid="%dx%d" % (Pr, Pc), nb=args.nb, transport=transport, N=N)
            with open(leg, "w") as f:
                json.dump(sub, f)
        print(json.dumps(out), flush=True)
    g.close()
    comm.close()


if __name__ == "__main__":
    main()
