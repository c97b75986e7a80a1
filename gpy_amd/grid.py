"""Optional multi-GPU mode: the exact-GP evaluation on a Pr x Pc process grid, N x N matrices 2D block-cyclic
(csrc/grid.hip, C-ABI `mi355gp_grid_*`; north_star config 4).

    # one process per GPU (torch.distributed.run / mpirun / anything that sets RANK, WORLD_SIZE, LOCAL_RANK)
    g = GridContext.from_env(Pr=2, Pc=4, nb=512)          # RCCL transport, id exchanged over `exchange`
    g.set_data(X, R)                                      # replicated inputs
    info, res = g.exact_inference("rbf", False, theta, noise)      # collective; results replicated

    g = GridContext.loopback(Pr=2, Pc=2, nb=256)          # all logical ranks on one device (tests on a 1-GPU box)

The host-side index algebra (who owns which tile, where it sits locally) lives here in pure Python as well so it
can be unit-tested without a GPU; csrc/grid.hip implements the same maps.
"""
import ctypes
import os
import sys

import numpy as np

from . import _lib
from ._lib import KIND_IDS, NUM_OUT, NUM_T, OUT_DATAFIT, OUT_DNOISE, OUT_LML, OUT_LOGDET, OUT_TRKINV, _opt, check, f64

FETCH_L, FETCH_LINV = 0, 100
ID_BYTES = 128


def _process_start():
    """Wall-clock start of this process (an id file older than this belongs to an earlier job)."""
    import time
    try:
        import psutil
        return psutil.Process().create_time()
    except Exception:
        return time.time()


_PROCESS_START = _process_start()


# ---- 2D block-cyclic index algebra (mirrors csrc/grid.hip) ----------------------------------------------------
def grid_shape(world):
    """Most square Pr x Pc with Pr <= Pc and Pr*Pc == world (8 -> 2 x 4)."""
    pr = int(np.floor(np.sqrt(world)))
    while world % pr:
        pr -= 1
    return pr, world // pr


def rank_coords(rank, Pc):
    return rank // Pc, rank % Pc


def tile_owner(I, J, Pr, Pc):
    """Rank owning global tile (I, J)."""
    return (I % Pr) * Pc + (J % Pc)


def local_tiles(T, p, P):
    """Global tile indices t < T with t % P == p, in local order."""
    return list(range(p, T, P))


def count_le(k, p, P):
    """Number of tiles t <= k with t % P == p."""
    return (k - p) // P + 1 if k >= p else 0


def global_index(local, p, P, nb):
    """Global row/col index of local row/col `local` on grid coordinate p."""
    return ((local // nb) * P + p) * nb + local % nb


def step_traffic_bytes(N, nb, Pr, Pc):
    """Bytes each collective family moves in total over the one-pass factorisation (per destination GPU summed):
    used by DESIGN.md's communication budget."""
    T = -(-N // nb)
    tile = nb * nb * 8
    row_panel = col_panel = xrow = xrowT = diag = 0
    for k in range(T):
        below = T - 1 - k
        row_panel += below * tile * (Pc - 1)          # L_ik to the other Pc-1 GPUs of its process row
        col_panel += below * tile * (Pr - 1)          # L_jk to the other Pr-1 GPUs of its process column
        xrow += (k + 1) * tile * (Pr - 1)             # X_kj down the process columns
        xrowT += (k + 1) * tile * (Pc - 1)            # X_ki along the process rows
        diag += tile * (Pr - 1 + Pc - 1)
    return dict(row_panel=row_panel, col_panel=col_panel, x_row=xrow, x_row_t=xrowT, diag=diag,
                total=row_panel + col_panel + xrow + xrowT + diag)


def _has_tile(lo, hi, a, Pa, b, Pb):
    """Is there a tile t in [lo, hi) with t % Pa == a and t % Pb == b?  (Chinese remainder: the residues must agree modulo
    gcd(Pa, Pb); the solutions then repeat with period lcm(Pa, Pb).)"""
    import math
    g = math.gcd(Pa, Pb)
    if (a - b) % g:
        return False
    L = Pa // g * Pb
    t0 = next(t for t in range(L) if t % Pa == a and t % Pb == b)
    first = t0 if t0 >= lo else t0 + -(-(lo - t0) // L) * L
    return first < hi


def expected_collectives(N, nb, Pr, Pc, rank, Dy=1, D=1):
    """Number of collectives ONE evaluation enqueues on the world / process-row / process-column communicator of grid rank
    `rank` (csrc/grid.hip, crit(k)): the diagonal inverse down its column and along its row, the row panel along every process
    row, the column-panel tiles as ONE broadcast per (process column, root process row) that has tiles, the X row panel down
    every process column, the transposed X tiles as one broadcast per (process row, root process column); then five
    all-reduces.  At most 2 + 1 + Pr + 1 + Pc per step and rank (2 x 4: 7 or fewer).  Independent of the look-ahead / grouping
    options: only the ORDER of compute changes with them, never the collectives."""
    T = -(-N // nb)
    pr, pc = rank // Pc, rank % Pc
    row = col = 0
    for k in range(T):
        opr, opc = k % Pr, k % Pc
        col += 1 if pc == opc else 0                                   # (b) D down process column opc
        row += 1 if pr == opr else 0                                   # (b) D along process row opr
        row += 1 if count_le(T - 1, pr, Pr) - count_le(k, pr, Pr) > 0 else 0     # (d) row panel (skipped when empty)
        col += sum(1 for root in range(Pr) if _has_tile(k + 1, T, pc, Pc, root, Pr))   # (e) column-panel runs, one per root
        col += 1 if count_le(k, pc, Pc) > 0 else 0                     # (h) X row panel down process column pc
        row += sum(1 for root in range(Pc) if _has_tile(0, k + 1, pr, Pr, root, Pc))    # (i) transposed X runs, one per root
    return {"world": 5, "row": row, "col": col}


def shard_rows(N, rank, world):
    """Contiguous row range [lo, hi) of rank `rank` (the reference's divide_data, util/parallel.py:14-30: the first
    N % world ranks get one extra row)."""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---- id exchange ---------------------------------------------------------------------------------------------
def exchange_id_torch(id_bytes, rank):
    """Broadcast rank 0's RCCL id over an initialised torch.distributed process group (gloo or nccl)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.zeros(ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        t = torch.tensor(list(id_bytes), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist())


def job_nonce():
    """Identity of THIS launch of the job, common to all its ranks: MI355GP_JOB_NONCE if set, else what torchrun exports
    (TORCHELASTIC_RUN_ID + restart count), else MASTER_ADDR:MASTER_PORT; '' when the launcher gives nothing."""
    n = os.environ.get("MI355GP_JOB_NONCE")
    if n:
        return n
    rid = os.environ.get("TORCHELASTIC_RUN_ID")
    if rid:
        return "%s#%s" % (rid, os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    if os.environ.get("MASTER_PORT"):
        return "%s:%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ["MASTER_PORT"])
    return ""


def exchange_id_file(id_bytes, rank, path, timeout=120.0, world=None, nonce=None):
    """Rank 0 writes the id to `path` (atomically, mode 0600); the others poll for it.  The file carries the launch's
    `job_nonce()` in front of the id and a reader accepts only a file with ITS OWN nonce: a rank that starts minutes after
    rank 0 still accepts it, a left-over of a crashed earlier launch (other restart count / run id) never is.
    Without any nonce from the launcher the modification time decides (not older than this process's start minus 30 s).
    With `world` given, every reader drops an acknowledgement next to it and rank 0 removes the files once all have read."""
    import time
    t_start = _PROCESS_START
    nonce = (job_nonce() if nonce is None else nonce).encode()
    header = b"MI355GPID" + bytes([len(nonce) >> 8, len(nonce) & 255]) + nonce
    if rank == 0:
        for old in (path,):
            try:
                os.unlink(old)
            except OSError:
                pass
        tmp = path + ".tmp%d" % os.getpid()
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(header + id_bytes)
        os.replace(tmp, path)
        if world:
            t0 = time.time()
            acks = [path + ".ack%d" % r for r in range(1, world)]
            while time.time() - t0 < timeout and not all(os.path.exists(a) for a in acks):
                time.sleep(0.02)
            for a in acks + [path]:
                try:
                    os.unlink(a)
                except OSError:
                    pass
        return id_bytes
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            st = os.stat(path)
            if st.st_size == len(header) + ID_BYTES:
                with open(path, "rb") as f:
                    data = f.read()
                fresh = st.st_mtime >= t_start - 30.0 if not nonce else True
                if data[:len(header)] == header and len(data) == len(header) + ID_BYTES and fresh:
                    if world:
                        open(path + ".ack%d" % rank, "wb").close()
                    return data[len(header):]
        except OSError:
            pass
        time.sleep(0.05)
    raise _lib.MI355GPError("timed out waiting for the RCCL id at %s" % path)


def comm_selftest(Pr, Pc, count=1 << 16, rounds=6, device=None, id_dir=None):
    """This rank's part of `mi355gp_dbg_comm_selftest` (the transport's CommInitRank / CommSplit / grouped multi-root
    broadcasts / all-reduces with checked payloads; RANK / WORLD_SIZE / LOCAL_RANK from the launcher, the id through the file
    channel).  Returns (mismatches, checksum, rank inside the row communicator, rank inside the column communicator)."""
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == Pr * Pc
    local = int(os.environ.get("LOCAL_RANK", str(rank))) if device is None else int(device)
    if device is None and os.environ.get("MI355GP_TRANSPORT") == "ipc":
        local %= max(1, _lib.device_count())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    idb = unique_id() if rank == 0 else b"\0" * ID_BYTES
    if world > 1:
        d = id_dir or os.environ.get("MI355GP_ID_DIR") or "/tmp"
        idb = exchange_id_file(idb, rank, os.path.join(d, "mi355gp_selftest_id_%dx%d" % (Pr, Pc)), world=world)
    out = np.zeros(4)
    check(_lib.lib().mi355gp_dbg_comm_selftest(local, idb, rank, world, Pr, Pc, int(count), int(rounds), out),
          "mi355gp_dbg_comm_selftest")
    return int(out[0]), float(out[1]), int(out[2]), int(out[3])


def unique_id():
    buf = ctypes.create_string_buffer(ID_BYTES)
    check(_lib.lib().mi355gp_grid_unique_id(buf), "mi355gp_grid_unique_id")
    return buf.raw


class GridContext(object):
    def __init__(self, device, rank, world, Pr, Pc, nb, id_bytes):
        _lib.require_device(device)
        assert Pr * Pc == world
        self._h = ctypes.c_void_p()
        self.rank, self.world, self.Pr, self.Pc, self.nb = rank, world, Pr, Pc, nb
        self.is_loopback = id_bytes is None
        check(_lib.lib().mi355gp_grid_create(device, rank, world, Pr, Pc, nb, id_bytes, ctypes.byref(self._h)),
              "mi355gp_grid_create")
        self.N = self.D = self.Dy = 0

    @classmethod
    def loopback(cls, Pr, Pc, nb=256, device=0):
        return cls(device, 0, Pr * Pc, Pr, Pc, nb, None)

    @classmethod
    def from_env(cls, Pr=None, Pc=None, nb=512, exchange=None, device=None):
        """One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK from the launcher; `exchange(id_bytes, rank)` ships
        rank 0's id (default: torch.distributed if initialised, else a file under $MI355GP_ID_DIR or /tmp).  device: HIP device
        of this rank (default LOCAL_RANK; under MI355GP_TRANSPORT=ipc, where ranks SHARE devices, LOCAL_RANK modulo the number
        of visible devices)."""
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", str(rank))) if device is None else int(device)
        if device is None and os.environ.get("MI355GP_TRANSPORT") == "ipc":
            local %= max(1, _lib.device_count())
        if Pr is None or Pc is None:
            Pr, Pc = grid_shape(world)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        idb = unique_id() if rank == 0 else b"\0" * ID_BYTES
        if world > 1:
            if exchange is None:
                # torch is never imported from here: the id travels over torch.distributed only if the CALLER already set a
                # process group up (bench.py under torch.distributed.run does); the file exchange is the default channel
                dist = sys.modules.get("torch.distributed")
                try:
                    use_torch = dist is not None and dist.is_available() and dist.is_initialized()
                except Exception:
                    use_torch = False
                if use_torch:
                    idb = exchange_id_torch(idb, rank)
                else:
                    # per-user private directory; the file name carries the launcher's job identity (torchrun exports
                    # TORCHELASTIC_RUN_ID; MASTER_PORT otherwise) so concurrent jobs never read each other's id
                    d = os.environ.get("MI355GP_ID_DIR") or os.path.join(
                        os.environ.get("XDG_RUNTIME_DIR") or os.path.expanduser("~"), ".mi355gp")
                    os.makedirs(d, mode=0o700, exist_ok=True)
                    tag = "%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "job"), os.environ.get("MASTER_PORT", "0"))
                    idb = exchange_id_file(idb, rank, os.path.join(d, "mi355gp_id_%s" % tag), world=world)
            else:
                idb = exchange(idb, rank)
        return cls(local, rank, world, Pr, Pc, nb, idb)

    OPTIONS = {"lookahead": 0, "G": 1, "GW": 2, "check_seq": 3}

    def set_option(self, name, value):
        """Schedule option of this grid (include/mi355gp.h MI355GP_GRID_OPT_*; -1 = process default); every rank must set
        the same value."""
        check(_lib.lib().mi355gp_grid_set_option(self._h, self.OPTIONS[name], int(value)), "mi355gp_grid_set_option")

    def get_option(self, name):
        v = ctypes.c_int(0)
        check(_lib.lib().mi355gp_grid_get_option(self._h, self.OPTIONS[name], ctypes.byref(v)), "mi355gp_grid_get_option")
        return v.value

    def coll_log(self, rank=0):
        """Collectives of the LAST evaluation as logical rank `rank` logged them (loopback: any rank; one rank per process:
        this process's own): {"world" | "row" | "col": (count, 64-bit hash over (operation, root, doubles) in order)}."""
        out = np.zeros(9)
        check(_lib.lib().mi355gp_grid_coll_log(self._h, int(rank), out), "mi355gp_grid_coll_log")
        return {name: (int(out[c]), (int(out[3 + 2 * c]) << 32) | int(out[4 + 2 * c]))
                for c, name in enumerate(("world", "row", "col"))}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().mi355gp_grid_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, X, R):
        X, R = f64(X), f64(R)
        self.N, self.D = X.shape
        self.Dy = R.shape[1]
        check(_lib.lib().mi355gp_grid_set_data(self._h, X, self.N, self.D, R, self.Dy), "mi355gp_grid_set_data")

    def exact_inference(self, kind, ARD, theta, noise, jitter=1e-8, extra_jitter=0.0, want_diag=False,
                        want_stage_ms=False):
        theta = f64(theta)
        noise = f64(np.atleast_1d(noise))
        out = np.zeros(NUM_OUT)
        alpha = np.empty((self.N, self.Dy))
        dtheta = np.zeros(theta.size)
        diag = np.empty(self.N) if want_diag else None
        ms = np.zeros(NUM_T) if want_stage_ms else None
        rc = check(_lib.lib().mi355gp_grid_exact_inference(self._h, KIND_IDS[kind], int(bool(ARD)), theta, noise,
                                                           noise.size, jitter, extra_jitter, out, _opt(alpha),
                                                           _opt(dtheta), _opt(diag), _opt(ms)),
                   "mi355gp_grid_exact_inference")
        res = dict(lml=out[OUT_LML], logdet=out[OUT_LOGDET], datafit=out[OUT_DATAFIT], dnoise=out[OUT_DNOISE],
                   trKinv=out[OUT_TRKINV], alpha=alpha, dtheta=dtheta, diag_dL_dK=diag)
        if ms is not None:
            res["stage_ms"] = dict(kbuild=ms[0], factor=ms[1], solve=ms[4], grad=ms[5], total=ms[6])
        return rc, res

    def fetch(self, which):
        """Lower triangle of L (FETCH_L) or L^-1 (FETCH_LINV): the tiles this process hosts, zeros elsewhere."""
        out = np.zeros((self.N, self.N))
        check(_lib.lib().mi355gp_grid_fetch(self._h, which, out), "mi355gp_grid_fetch")
        return out
