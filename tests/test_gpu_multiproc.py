"""GPU (-m gpu), ONE GPU: the multi-PROCESS flow of the block-cyclic mode (north_star config 4) and of the row-sharded
sparse path (`var_dtc_parallel.py:121-130`) with REAL ranks -- one process per rank, rank variables from the environment, the id
shipped through the host channel, communicators split into process rows / columns, every process running the per-rank code of
csrc/grid.hip that an 8-GPU node will run -- over the hipIpc transport (`MI355GP_TRANSPORT=ipc`, csrc/ipc_comm.hip) that is bound
in RCCL's place because RCCL refuses two ranks on one device.  The bar: the bits of the single-process loopback run, and, per
rank and communicator, exactly the collective sequence (operation, root, size, order) the loopback run logged for that rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gpy_amd import _lib as L
from gpy_amd import grid as G
from oracle import gp_oracle as O
from oracle import sparse_oracle as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from gpy_amd import _lib as L, grid as G
from gpy_amd.datasets import default_theta, synthetic, synthetic_Z
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode, N, D, Pr, Pc, nb, M = %(mode)r, %(N)d, %(D)d, %(Pr)d, %(Pc)d, %(nb)d, %(M)d
X, Y = synthetic(N, D, seed=3)
var, ls, noise = default_theta(D, True)
th = L.theta_vec(var, ls, True, D)
if mode == "grid":
    g = G.GridContext.from_env(Pr, Pc, nb)            # RANK / WORLD_SIZE / LOCAL_RANK, id through the file channel
    g.set_option("check_seq", 1)
    g.set_data(X, Y)
    outs = []
    for _ in range(2):
        info, r = g.exact_inference("matern52", True, th, noise)
        assert info == 0
        outs.append((float(r["lml"]).hex(), r["dtheta"].tobytes().hex(), r["alpha"].tobytes().hex()))
    assert outs[0] == outs[1]
    log = g.coll_log()
    res = dict(lml=outs[0][0], dtheta=outs[0][1], alpha=outs[0][2][:512], log={k: [v[0], str(v[1])] for k, v in log.items()})
    g.close()
else:
    Z = synthetic_Z(X, M, 0)
    idb = G.unique_id() if rank == 0 else b"\0" * G.ID_BYTES
    idb = G.exchange_id_file(idb, rank, os.path.join(os.environ["MI355GP_ID_DIR"], "sparse_id"), world=world)
    lo, hi = G.shard_rows(N, rank, world)
    c = L.SparseContext(0)
    c.attach_comm(rank, world, idb)
    c.set_data(X[lo:hi], Y[lo:hi])
    info, r = c.vardtc("rbf", True, th, Z, noise)
    assert info == 0
    res = dict(lml=float(r["lml"]).hex(), dtheta=r["dtheta"].tobytes().hex(), dZ=r["dZ"].tobytes().hex()[:512])
    c.close()
print("RESULT " + json.dumps(res))
"""


def _spawn(world, tmp_path, **kw):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT, **kw))
    env = dict(os.environ, WORLD_SIZE=str(world), LOCAL_RANK="0", MI355GP_TRANSPORT="ipc", MI355GP_ID_DIR=str(tmp_path),
               MI355GP_JOB_NONCE="t%d" % os.getpid(), MI355GP_IPC_TIMEOUT_S="240", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MASTER_PORT=str(29000 + os.getpid() % 2000))
    env.pop("TORCHELASTIC_RUN_ID", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    res = []
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, se[-3000:])
        res.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    return res


@pytest.mark.parametrize("Pr,Pc,nb,N", [(1, 2, 256, 1300), (2, 2, 256, 1536), (2, 4, 512, 4096), (3, 2, 128, 1100)])
def test_one_process_per_rank_matches_the_single_process_loopback_bit_for_bit(Pr, Pc, nb, N, tmp_path):
    from gpy_amd.datasets import default_theta, synthetic
    D, world = 5, Pr * Pc
    X, Y = synthetic(N, D, seed=3)
    var, ls, noise = default_theta(D, True)
    th = L.theta_vec(var, ls, True, D)
    g = G.GridContext.loopback(Pr, Pc, nb)
    try:
        g.set_option("check_seq", 1)
        g.set_data(X, Y)
        info, ref = g.exact_inference("matern52", True, th, noise)
        assert info == 0
        logs = [g.coll_log(r) for r in range(world)]
    finally:
        g.close()
    # the loopback run against the oracle and against the model of what one evaluation enqueues
    o = O.parameters_changed("matern52", X, Y, var, ls, True, noise)
    assert abs(ref["lml"] - o["lml"]) <= 1e-10 * abs(o["lml"])
    for r in range(world):
        want = G.expected_collectives(N, nb, Pr, Pc, r)
        assert {k: v[0] for k, v in logs[r].items()} == want, (r, logs[r], want)
    res = _spawn(world, tmp_path, mode="grid", N=N, D=D, Pr=Pr, Pc=Pc, nb=nb, M=0)
    for r, rec in enumerate(res):
        assert rec["lml"] == float(ref["lml"]).hex(), (r, rec["lml"], float(ref["lml"]).hex())
        assert rec["dtheta"] == ref["dtheta"].tobytes().hex()
        assert rec["alpha"] == ref["alpha"].tobytes().hex()[:512]
        assert {k: (v[0], int(v[1])) for k, v in rec["log"].items()} == logs[r], (r, rec["log"], logs[r])


_SELFTEST = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
from gpy_amd import grid as G
bad, checksum, rrow, rcol = G.comm_selftest(%(Pr)d, %(Pc)d, count=%(count)d, rounds=%(rounds)d)
rank = int(os.environ["RANK"])
assert (rrow, rcol) == (rank %% %(Pc)d, rank // %(Pc)d), (rank, rrow, rcol)
print("RESULT " + json.dumps(dict(bad=bad, checksum=checksum)))
"""


@pytest.mark.parametrize("Pr,Pc", [(1, 2), (2, 2), (2, 4)])
def test_transport_selftest_grouped_multi_root_broadcasts(Pr, Pc, tmp_path):
    """`mi355gp_dbg_comm_selftest` drives the BOUND transport through the function table the grid mode uses: communicator splits,
    several broadcasts with different roots inside one group, all-reduces, payloads checked.  Here over the hipIpc stand-in, one
    process per rank on one GPU; on a node with more GPUs `tools/rccl_first_light.sh` runs the same call over RCCL."""
    world = Pr * Pc
    script = tmp_path / "selftest.py"
    script.write_text(_SELFTEST % dict(root=ROOT, Pr=Pr, Pc=Pc, count=70000, rounds=7))
    env = dict(os.environ, WORLD_SIZE=str(world), LOCAL_RANK="0", MI355GP_TRANSPORT="ipc", MI355GP_ID_DIR=str(tmp_path),
               MI355GP_JOB_NONCE="s%d" % os.getpid(), MI355GP_IPC_TIMEOUT_S="240", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TORCHELASTIC_RUN_ID", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, se[-3000:])
        rec = json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:])
        assert rec["bad"] == 0, (r, rec)


def test_row_sharded_sparse_path_with_one_process_per_rank(tmp_path):
    """Three processes, rows sharded 3 ways, the two all-reduces of the reference's MPI design over the hipIpc transport: every
    rank returns the unsharded result (oracle) and all ranks the same bits."""
    from gpy_amd.datasets import default_theta, synthetic, synthetic_Z
    N, M, D, world = 30000, 256, 4, 3
    res = _spawn(world, tmp_path, mode="sparse", N=N, D=D, Pr=1, Pc=1, nb=0, M=M)
    assert len({(r["lml"], r["dtheta"], r["dZ"]) for r in res}) == 1
    X, Y = synthetic(N, D, seed=3)
    Z = synthetic_Z(X, M, 0)
    var, ls, noise = default_theta(D, True)
    ref = S.vardtc("rbf", X, Z, Y, var, ls, True, noise)
    lml = float.fromhex(res[0]["lml"])
    dth = np.frombuffer(bytes.fromhex(res[0]["dtheta"]))
    assert abs(lml - ref["lml"]) <= 1e-9 * abs(ref["lml"])
    assert np.abs(dth - ref["dtheta"]).max() <= 1e-6 * np.abs(ref["dtheta"]).max()
