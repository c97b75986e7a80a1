"""CPU, world_size 2 over gloo: the rank plumbing of bench.py (barriers around the timed region, MAX over ranks,
whole-job aggregation for independent replicas).  The data path itself has no collective (replicas only)."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import bench
    comm = bench.Comm()                             # the DEFAULT rank plumbing of a multi-rank launch
    calls = []
    def step():
        calls.append(1)
        time.sleep(0.02 * (1 + comm.rank))        # rank 1 is the slow replica
    dt = bench.timed_region(comm, step, steps=5, warmup=2)
    with open(os.path.join(%r, "rank%%d.json" %% comm.rank), "w") as f:
        import torch
        json.dump({"rank": comm.rank, "world": comm.world, "dt": dt, "calls": len(calls), "backend": comm.backend,
                   "device_tensor": comm.device_tensor, "cuda_initialized": bool(torch.cuda.is_initialized()),
                   "gathered": comm.gather([comm.rank, 10.0 + comm.rank])}, f)
    comm.close()
""")


def test_two_rank_timed_region_takes_max_over_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    env.pop("MI355GP_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)]
    assert sorted(x["rank"] for x in recs) == [0, 1]
    assert all(x["world"] == 2 and x["calls"] == 7 for x in recs)          # 2 warm-up + exactly 5 timed steps
    # gloo with CPU tensors unless MI355GP_BENCH_BACKEND=nccl asks otherwise, and torch.cuda never touched: the library's own
    # dlopen'ed RCCL communicator (grid / sparse legs) must not share a process with PyTorch's
    assert all(x["backend"] == "gloo" and not x["device_tensor"] and not x["cuda_initialized"] for x in recs)
    assert all(x["gathered"] == [[0.0, 10.0], [1.0, 11.0]] for x in recs)   # per-rank stage times reach every rank in rank order
    assert abs(recs[0]["dt"] - recs[1]["dt"]) < 1e-9                        # both ranks hold the MAX
    assert recs[0]["dt"] >= 5 * 0.04 * 0.95                                 # the slow rank's time
    # whole-job throughput of independent replicas = n_gpus * steps / max time
    assert 2 * 5 / recs[0]["dt"] < 2 * 5 / (5 * 0.02)


def test_single_process_comm_is_a_no_op():
    sys.path.insert(0, ROOT)
    import bench
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        comm = bench.Comm()
        assert comm.world == 1 and comm.rank == 0
        n = []
        dt = bench.timed_region(comm, lambda: n.append(1), steps=3, warmup=1)
        assert len(n) == 4 and dt >= 0
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v


LEG_PARENT = textwrap.dedent("""
    import json, os, subprocess, sys
    sys.path.insert(0, %r)
    import bench
    comm = bench.Comm(backend="gloo")
    # what grid_leg / sparse_leg do: every rank spawns ITS child with its own rank variables; the children form their own group
    child = "import sys; sys.path.insert(0, %r); import bench; c = bench.Comm(backend='gloo'); c.barrier(); " \\
            "print('CHILD', c.rank, c.world, c.max_over_ranks(c.rank)); c.close()"
    try:
        r = subprocess.run([sys.executable, "-c", child], env=bench.child_env(17), capture_output=True, text=True, timeout=60)
        out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "rc %%d %%s" %% (r.returncode, r.stderr[-200:])
    except subprocess.TimeoutExpired:
        out = "TIMEOUT"
    with open(os.path.join(%r, "leg%%d.txt" %% comm.rank), "w") as f:
        f.write(out)
    comm.barrier()
    comm.close()
""")


def test_child_legs_of_a_torchrun_launch_can_rendezvous(tmp_path):
    """The `grid` and `c5` records of the default line come from CHILD processes that every rank spawns with its own rank variables
    and a moved MASTER_PORT.  Under `torch.distributed.run` the workers' environment carries TORCHELASTIC_USE_AGENT_STORE=True: a
    child inheriting it never starts the store of its own group and hangs to the leg's time-out -- `bench.child_env` strips it.
    (Found by round 4's 8-process dry run; on an 8-GPU node both legs would have timed out.)"""
    script = tmp_path / "parent.py"
    script.write_text(LEG_PARENT % (ROOT, ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29741", str(script)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    outs = [open(tmp_path / ("leg%d.txt" % i)).read() for i in range(2)]
    assert outs[0].startswith("CHILD 0 2 1.0") and outs[1].startswith("CHILD 1 2 1.0"), outs
