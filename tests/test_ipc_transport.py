"""CPU (no GPU): the multi-PROCESS transport of csrc/ipc_comm.hip -- the provider of the RCCL entry points that
`MI355GP_TRANSPORT=ipc` binds for ranks that are processes sharing one GPU (the dry run of the 8-rank flow on a 1-GPU box,
SURVEY 8e / north_star config 4) -- in HOST mode (`MI355GP_IPC_HOST=1`: buffers are host memory, no HIP call): barriers,
communicator split with the rank numbering of `ncclCommSplit(colour, key)` exactly as `mi355gp_grid_create` uses it (row
communicator: colour pr, key pc; column communicator: colour pc, key pr), root-as-coordinate broadcasts by members only,
chunking over more than one staging buffer, sums in communicator-rank order."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import ctypes, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from gpy_amd import _lib as L
rank, world, Pr, Pc, count, idfile = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
                                      sys.argv[6])
lib = L.lib()
if rank == 0:
    buf = ctypes.create_string_buffer(128)
    assert lib.mi355gp_grid_unique_id(buf) == 0, L.last_error()
    with open(idfile + ".tmp", "wb") as f:
        f.write(buf.raw)
    os.replace(idfile + ".tmp", idfile)
    idb = buf.raw
else:
    import time
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    idb = open(idfile, "rb").read()
out = np.zeros(4)
rc = lib.mi355gp_dbg_ipc_selftest(idb, rank, world, Pr, Pc, count, out)
print(json.dumps({"rc": rc, "bad": out[0], "sum": out[1], "row_rank": out[2], "col_rank": out[3]}))
"""


def _run(Pr, Pc, count, tmp_path):
    world = Pr * Pc
    env = dict(os.environ, MI355GP_TRANSPORT="ipc", MI355GP_IPC_HOST="1", MI355GP_IPC_TIMEOUT_S="60")
    idfile = str(tmp_path / ("id_%dx%d" % (Pr, Pc)))
    code = _CHILD % {"root": ROOT}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world), str(Pr), str(Pc), str(count), idfile], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    recs = []
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, "rank %d: %s" % (r, se[-1500:])
        recs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    return recs


@pytest.mark.parametrize("Pr,Pc,count", [(1, 2, 5000), (2, 2, 70000), (2, 4, 2200000), (3, 1, 1000)])
def test_ipc_transport_protocol_in_host_mode(Pr, Pc, count, tmp_path):
    """(2, 4, 2.2e6 doubles): the 2 x 4 grid of BASELINE configs[3] with messages larger than the 16 MiB staging buffer.
    rc == 0 also covers the selftest's refusals (negative codes -20..-23): a split colour outside 0..15, and a split whose
    slot number lands on a communicator that is still alive -- refused by every member, after which the live one still works."""
    recs = _run(Pr, Pc, count, tmp_path)
    for r, rec in enumerate(recs):
        assert rec["rc"] == 0 and rec["bad"] == 0.0, (r, rec)
        # ncclCommSplit numbering: inside a process row the rank is the grid column, inside a process column the grid row
        assert (rec["row_rank"], rec["col_rank"]) == (r % Pc, r // Pc)
    # ranks of one process row received the same row broadcasts and sums: checksums differ only through the column traffic
    assert len({rec["sum"] for rec in recs}) >= 1
