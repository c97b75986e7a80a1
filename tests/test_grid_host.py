"""CPU: host logic of the 2D block-cyclic multi-GPU mode (gpy_amd/grid.py) -- ownership / local-index algebra
checked against a brute-force restatement, the communication budget DESIGN.md quotes, and the N>1 plumbing
(world_size 2 over gloo: RCCL id exchange by torch.distributed and by file; every rank derives the same maps)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from gpy_amd import grid as G


@pytest.mark.parametrize("world,expect", [(1, (1, 1)), (2, (1, 2)), (4, (2, 2)), (8, (2, 4)), (6, (2, 3)), (7, (1, 7))])
def test_grid_shape(world, expect):
    assert G.grid_shape(world) == expect


@pytest.mark.parametrize("T,Pr,Pc", [(1, 1, 1), (5, 2, 2), (7, 2, 4), (9, 3, 2), (16, 4, 2), (6, 1, 3)])
def test_every_tile_has_exactly_one_owner_and_local_order_is_increasing(T, Pr, Pc):
    owned = {}
    for r in range(Pr * Pc):
        pr, pc = G.rank_coords(r, Pc)
        rows, cols = G.local_tiles(T, pr, Pr), G.local_tiles(T, pc, Pc)
        assert rows == sorted(rows) and cols == sorted(cols)
        assert len(rows) == G.count_le(T - 1, pr, Pr) and len(cols) == G.count_le(T - 1, pc, Pc)
        for I in rows:
            for J in cols:
                assert (I, J) not in owned
                owned[(I, J)] = r
                assert G.tile_owner(I, J, Pr, Pc) == r
    assert len(owned) == T * T


def test_count_le_and_global_index():
    for P in (1, 2, 3, 4):
        for p in range(P):
            for k in range(0, 12):
                assert G.count_le(k, p, P) == sum(1 for t in range(k + 1) if t % P == p)
    nb = 4
    for P in (1, 2, 3):
        seen = set()
        for p in range(P):
            for l in range(5 * nb):
                gidx = G.global_index(l, p, P, nb)
                assert (gidx // nb) % P == p and gidx % nb == l % nb
                seen.add(gidx)
        assert len(seen) == P * 5 * nb


def test_communication_budget_matches_survey_estimate():
    # SURVEY.md 7 (hard part 8): N=32k on 2x4: every L tile goes to Pc-1 = 3 GPUs of its row and Pr-1 = 1 of its column
    t = G.step_traffic_bytes(32768, 512, 2, 4)
    half = 8 * 32768 ** 2 / 2
    assert abs(t["row_panel"] - 3 * half) / (3 * half) < 0.05
    assert abs(t["col_panel"] - 1 * half) / half < 0.05
    # the inverse and Ky^-1 ride on a second pair of panel broadcasts of the same size class
    assert abs(t["x_row"] - 1 * half) / half < 0.05 and abs(t["x_row_t"] - 3 * half) / (3 * half) < 0.05
    assert t["total"] == sum(v for k, v in t.items() if k != "total")
    one = G.step_traffic_bytes(4096, 512, 1, 1)
    assert one["total"] == 0


WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from gpy_amd import grid as G
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = bytes(range(128)) if rank == 0 else bytes(128)          # stands in for mi355gp_grid_unique_id (needs a GPU)
    got_t = G.exchange_id_torch(fake, rank)
    got_f = G.exchange_id_file(fake, rank, os.path.join(%r, "id.bin"))
    Pr, Pc = G.grid_shape(world)
    pr, pc = G.rank_coords(rank, Pc)
    T = 7
    rec = {"rank": rank, "id_t": list(got_t), "id_f": list(got_f), "grid": [Pr, Pc], "coords": [pr, pc],
           "rows": G.local_tiles(T, pr, Pr), "cols": G.local_tiles(T, pc, Pc)}
    json.dump(rec, open(os.path.join(%r, "rank%%d.json" %% rank), "w"))
    dist.destroy_process_group()
""")


def test_two_rank_id_exchange_and_partition(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path), str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29741", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = sorted((json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)), key=lambda x: x["rank"])
    for x in recs:
        assert x["id_t"] == list(range(128)) and x["id_f"] == list(range(128))
        assert x["grid"] == [1, 2]
    assert recs[0]["coords"] == [0, 0] and recs[1]["coords"] == [0, 1]
    assert recs[0]["rows"] == recs[1]["rows"] == list(range(7))
    assert sorted(recs[0]["cols"] + recs[1]["cols"]) == list(range(7))
    assert set(recs[0]["cols"]).isdisjoint(recs[1]["cols"])


@pytest.mark.parametrize("N,world", [(10, 1), (10, 3), (200000, 8), (7, 7), (5, 8)])
def test_shard_rows_partitions_like_divide_data(N, world):
    """gpy_amd.grid.shard_rows == the reference's divide_data (GPy/util/parallel.py:14-30)."""
    covered = []
    for r in range(world):
        lo, hi = G.shard_rows(N, r, world)
        residue = N % world
        size = N // world + (1 if r < residue else 0)
        offset = size * r if r < residue else size * r + residue
        assert (lo, hi) == (offset, offset + size)
        covered.extend(range(lo, hi))
    assert covered == list(range(N))


def test_id_file_is_validated_by_job_nonce_not_by_age(tmp_path, monkeypatch):
    """a reader accepts an id file only with its own launch nonce -- however old the file is -- and never the
    left-over of another launch."""
    import threading
    import time
    from gpy_amd import grid as G
    path = str(tmp_path / "id.bin")
    ida, idb = bytes(range(128)), bytes(reversed(range(128)))
    G.exchange_id_file(ida, 0, path, nonce="job-A#0")              # rank 0 of a launch that then crashed
    old = time.time() - 3600.0
    os.utime(path, (old, old))                                     # ... an hour ago
    assert G.exchange_id_file(b"", 1, path, timeout=1.0, nonce="job-A#0") == ida       # same launch: age does not matter
    with pytest.raises(Exception):
        G.exchange_id_file(b"", 1, path, timeout=0.3, nonce="job-A#1")                 # the relaunch must not read it
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("id", G.exchange_id_file(b"", 1, path, timeout=10.0, nonce="job-A#1")))
    t.start()
    time.sleep(0.2)
    G.exchange_id_file(idb, 0, path, nonce="job-A#1")              # rank 0 of the relaunch replaces the stale file
    t.join()
    assert got["id"] == idb
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "abc")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "2")
    monkeypatch.delenv("MI355GP_JOB_NONCE", raising=False)
    assert G.job_nonce() == "abc#2"


def test_expected_collectives_equals_a_brute_force_enumeration_of_one_evaluation():
    """`grid.expected_collectives` (closed form per rank, asserted against the real collective logs on the GPU) against an
    enumeration of every broadcast `crit(k)` of csrc/grid.hip enqueues -- (communicator kind, communicator index, doubles) -- with
    the membership rule of `grid_bcast` (a rank takes part iff it belongs to that process row / column; empty broadcasts are
    skipped by everybody)."""
    from gpy_amd import grid as G
    for N, nb, Pr, Pc in ((4096, 512, 2, 4), (1536, 256, 2, 2), (1300, 256, 1, 2), (3000, 128, 3, 2), (700, 128, 1, 1), (5000, 128, 4, 6)):
        T = -(-N // nb)
        tile = nb * nb
        calls = []                                   # (kind, index, doubles)
        for k in range(T):
            opr, opc = k % Pr, k % Pc
            calls.append(("col", opc, tile))                                     # D down its process column
            calls.append(("row", opr, tile))                                     # D along its process row
            for pr in range(Pr):                                                 # row panel along every process row
                below = sum(1 for i in range(k + 1, T) if i % Pr == pr)
                calls.append(("row", pr, below * tile))
            for pc in range(Pc):                                                 # column-panel tiles: one run per root
                for root in range(Pr):
                    run = [j for j in range(k + 1, T) if j % Pc == pc and j % Pr == root]
                    calls.append(("col", pc, len(run) * tile))
            for pc in range(Pc):                                                 # X row panel down every process column
                upto = sum(1 for j in range(k + 1) if j % Pc == pc)
                calls.append(("col", pc, upto * tile))
            for pr in range(Pr):                                                 # transposed X tiles: one run per root
                for root in range(Pc):
                    run = [i for i in range(k + 1) if i % Pr == pr and i % Pc == root]
                    calls.append(("row", pr, len(run) * tile))
        for rank in range(Pr * Pc):
            pr, pc = rank // Pc, rank % Pc
            row = sum(1 for kind, idx, cnt in calls if kind == "row" and idx == pr and cnt > 0)
            col = sum(1 for kind, idx, cnt in calls if kind == "col" and idx == pc and cnt > 0)
            assert G.expected_collectives(N, nb, Pr, Pc, rank) == {"world": 5, "row": row, "col": col}, (N, nb, Pr, Pc, rank)


def test_collectives_per_step_and_rank_stay_within_the_budget_of_the_merged_panel_broadcasts():
    """at most 8 collectives per step and rank on the 2 x 4 grid of BASELINE configs[3] (the per-tile column
    broadcasts of rounds 2-5 were up to T - 1 = 63 per step); in general at most 4 + Pr + Pc."""
    from gpy_amd import grid as G
    for N, nb, Pr, Pc in ((32768, 512, 2, 4), (16384, 512, 2, 4), (4096, 512, 2, 4), (8192, 512, 2, 2), (3000, 128, 3, 2)):
        T = -(-N // nb)
        for rank in range(Pr * Pc):
            e = G.expected_collectives(N, nb, Pr, Pc, rank)
            per_step = (e["row"] + e["col"]) / T
            assert per_step <= 4 + Pr + Pc, (N, Pr, Pc, rank, per_step)
            if (Pr, Pc) == (2, 4):
                assert per_step <= 8, (N, rank, per_step)
