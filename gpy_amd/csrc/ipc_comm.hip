// ipc_comm.hip -- a second provider of the handful of RCCL entry points that grid.hip / sparse.hip use (ncclGetUniqueId,
// ncclCommInitRank, ncclCommSplit, ncclBroadcast, ncclAllReduce, ncclGroupStart/End, ncclCommDestroy), with the SAME
// signatures and the same rank-numbering semantics, for ranks that are separate PROCESSES sharing ONE GPU:
//   MI355GP_TRANSPORT=ipc python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...
// RCCL refuses two ranks on one device, so on a 1-GPU box the multi-process flow of the block-cyclic mode (SURVEY 8e,
// north_star config 4) and of the row-sharded sparse path (var_dtc_parallel.py:121-130) -- rank variables from the launcher, id
// exchange, communicator split into process rows / columns, one process per rank calling the per-rank code of grid.hip with
// g->ranks.size() == 1 -- could only be exercised at world size 1.  With this provider bound into the same function table,
// every line of that code except the RCCL library itself runs with 8 real processes.
//
// How: a POSIX shared-memory control block named by the 128-byte id (barriers with a generation counter per communicator,
// the colours / keys of a split, every rank's hipIpcMemHandle), one 16 MiB staging buffer per rank in device memory, exported
// with hipIpcGetMemHandle and mapped by every peer.  A collective is BLOCKING on the host: root copies send -> its staging
// buffer, stream sync, barrier, peers copy staging -> recv, stream sync, barrier (chunked by the staging size).  All ranks issue
// the collectives of a communicator in one global order (grid.hip enqueues the same sequence everywhere), so blocking
// semantics cannot deadlock and ncclGroupStart / End are no-ops.  Sums are taken in communicator-rank order (the order of the
// single-process loopback transport: same bits).  Slow by construction; it is test infrastructure for the rank plumbing, not
// a data path anyone should time.
//
// MI355GP_IPC_HOST=1: buffers are HOST memory (staging in POSIX shared memory, copies are memcpy, no HIP call at all): the
// protocol itself -- barriers, split numbering, chunking, root semantics -- is unit-tested on a machine without a GPU.
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

#define IPC_MAXR 64                        // ranks
#define IPC_MAXCOMM 256                    // communicator slots: 0 = world, 1 + split * 16 + colour
#define IPC_STAGE_BYTES (16u << 20)

namespace {

struct CtlRank {
    hipIpcMemHandle_t handle;
    std::atomic<int> ready;
};
struct CtlComm {
    std::atomic<unsigned> count;
    std::atomic<unsigned> gen;
    std::atomic<int> users;                // live members bound to this slot: more than the communicator's size = two communicators share it
    int color[IPC_MAXR], key[IPC_MAXR];
};
struct Ctl {
    std::atomic<int> world;
    std::atomic<int> failed;               // a rank gave up (timeout): everyone else stops waiting too
    CtlRank rank[IPC_MAXR];
    CtlComm comm[IPC_MAXCOMM];
};

struct World {
    Ctl* ctl = nullptr;
    char name[96] = {0};
    int world = 0, rank = 0;
    bool host = false;
    double* stage[IPC_MAXR] = {nullptr};   // every rank's staging buffer as mapped HERE
    double timeout_s = 120.0;
    int refs = 0;
};

struct Comm {
    World* w = nullptr;
    int slot = 0, n = 0, me = 0, nsplit = 0;
    int global[IPC_MAXR] = {0};            // communicator rank -> world rank
};

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// spin until pred() or timeout / a peer's failure flag; polite to the CPU (the peers of a 1-GPU dry run share the cores)
template <class Pred>
bool wait_for(World* w, Pred pred) {
    const double t0 = now_s();
    for (unsigned it = 0;; ++it) {
        if (pred()) return true;
        if (w->ctl->failed.load(std::memory_order_relaxed)) return false;
        if ((it & 255) == 255) {
            if (now_s() - t0 > w->timeout_s) {
                w->ctl->failed.store(1);
                return false;
            }
            usleep(20);
        } else {
            sched_yield();
        }
    }
}

bool barrier(Comm* c) {
    if (c->n == 1) return true;
    CtlComm& b = c->w->ctl->comm[c->slot];
    const unsigned g = b.gen.load(std::memory_order_acquire);
    if (b.count.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)c->n) {
        b.count.store(0, std::memory_order_relaxed);
        b.gen.fetch_add(1, std::memory_order_release);
        return true;
    }
    return wait_for(c->w, [&]() { return b.gen.load(std::memory_order_acquire) != g; });
}

__global__ void k_ipc_sum(double* __restrict__ out, const double* const* __restrict__ src, int n, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = src[0][i];
    for (int r = 1; r < n; ++r) s += src[r][i];             // communicator-rank order: the loopback transport's order
    out[i] = s;
}

bool copy(World* w, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (w->host) {
        memcpy(dst, src, bytes);
        return true;
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
}
bool sync(World* w, hipStream_t st) { return w->host || hipStreamSynchronize(st) == hipSuccess; }

void* map_shm(const char* name, size_t bytes, bool create) {
    const int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (create && ftruncate(fd, (off_t)bytes) != 0) {
        close(fd);
        shm_unlink(name);
        return nullptr;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}

}  // namespace

// ---- the entry points (signatures of <rccl/rccl.h>) -----------------------------------------------------------------------
ncclResult_t ipcGetUniqueId(ncclUniqueId* id) {
    static std::atomic<int> counter{0};
    memset(id, 0, sizeof(*id));
    char* s = id->internal;
    snprintf(s, sizeof(id->internal), "/mi355gp_ipc_%d_%d_%lx", (int)getpid(), counter.fetch_add(1), (unsigned long)(now_s() * 1e6));
    void* p = map_shm(s, sizeof(Ctl), true);                // ftruncate zero-fills: every atomic starts at 0
    if (!p) return ncclSystemError;
    munmap(p, sizeof(Ctl));
    return ncclSuccess;
}

// everything a World holds: mapped / opened staging buffers, the control block, the object itself
static void world_release(World* w) {
    for (int r = 0; r < w->world; ++r) {
        if (!w->stage[r]) continue;
        if (w->host) munmap(w->stage[r], IPC_STAGE_BYTES);
        else if (r == w->rank) (void)hipFree(w->stage[r]);
        else (void)hipIpcCloseMemHandle(w->stage[r]);
    }
    if (w->ctl) munmap(w->ctl, sizeof(Ctl));
    delete w;
}

ncclResult_t ipcCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > IPC_MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    World* w = new World();
    auto fail = [&](ncclResult_t rc) {                     // a failed initialisation leaves nothing mapped, opened or allocated
        if (w->host && w->stage[rank]) {
            char nm[128];
            snprintf(nm, sizeof(nm), "%s_r%d", w->name, rank);
            shm_unlink(nm);
        }
        world_release(w);
        return rc;
    };
    memcpy(w->name, id.internal, sizeof(w->name) - 1);
    w->world = nranks;
    w->rank = rank;
    {
        const char* e = PRODUCT_ENV("IPC_HOST");
        w->host = e && atoi(e) != 0;
        const char* t = PRODUCT_ENV("IPC_TIMEOUT_S");
        if (t && atof(t) > 0.0) w->timeout_s = atof(t);
    }
    w->ctl = (Ctl*)map_shm(w->name, sizeof(Ctl), false);
    if (!w->ctl) return fail(ncclSystemError);
    w->ctl->world.store(nranks);
    // my staging buffer, published through the control block
    char sname[128];
    if (w->host) {
        snprintf(sname, sizeof(sname), "%s_r%d", w->name, rank);
        w->stage[rank] = (double*)map_shm(sname, IPC_STAGE_BYTES, true);
        if (!w->stage[rank]) return fail(ncclSystemError);
    } else {
        if (hipMalloc(&w->stage[rank], IPC_STAGE_BYTES) != hipSuccess) return fail(ncclUnhandledCudaError);
        if (hipIpcGetMemHandle(&w->ctl->rank[rank].handle, w->stage[rank]) != hipSuccess) return fail(ncclUnhandledCudaError);
    }
    w->ctl->rank[rank].ready.store(1, std::memory_order_release);
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) continue;
        if (!wait_for(w, [&]() { return w->ctl->rank[r].ready.load(std::memory_order_acquire) != 0; })) return fail(ncclSystemError);
        if (w->host) {
            snprintf(sname, sizeof(sname), "%s_r%d", w->name, r);
            w->stage[r] = (double*)map_shm(sname, IPC_STAGE_BYTES, false);
            if (!w->stage[r]) return fail(ncclSystemError);
        } else {
            void* p = nullptr;
            if (hipIpcOpenMemHandle(&p, w->ctl->rank[r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                return fail(ncclUnhandledCudaError);
            w->stage[r] = (double*)p;
        }
    }
    Comm* c = new Comm();
    c->w = w;
    c->slot = 0;
    c->n = nranks;
    c->me = rank;
    for (int r = 0; r < nranks; ++r) c->global[r] = r;
    w->refs = 1;
    if (!barrier(c)) {                                     // everyone has mapped everything: the names can go
        delete c;
        return fail(ncclSystemError);
    }
    if (rank == 0) shm_unlink(w->name);
    if (w->host) {
        snprintf(sname, sizeof(sname), "%s_r%d", w->name, rank);
        shm_unlink(sname);
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

// ncclCommSplit: ranks with the same colour form a communicator, ordered by key (ties: parent rank)
ncclResult_t ipcCommSplit(ncclComm_t parent, int color, int key, ncclComm_t* out, ncclConfig_t*) {
    Comm* p = (Comm*)parent;
    if (color < 0 || color >= 16) {
        mi355gp_set_error("ipc transport: split colour %d outside 0..15 (a process grid of at most 16 rows / columns)", color);
        return ncclInvalidArgument;
    }
    CtlComm& b = p->w->ctl->comm[p->slot];
    b.color[p->me] = color;
    b.key[p->me] = key;
    if (!barrier(p)) return ncclSystemError;
    std::vector<std::pair<std::pair<int, int>, int>> mem;   // ((key, parent rank), world rank)
    for (int r = 0; r < p->n; ++r)
        if (b.color[r] == color) mem.push_back({{b.key[r], r}, p->global[r]});
    std::sort(mem.begin(), mem.end());
    Comm* c = new Comm();
    c->w = p->w;
    c->slot = 1 + (p->slot * 7 + p->nsplit) % 15 * 16 + color;   // same on every member: same parent, same split number
    p->nsplit += 1;
    c->n = (int)mem.size();
    for (int i = 0; i < c->n; ++i) {
        c->global[i] = mem[(size_t)i].second;
        if (mem[(size_t)i].second == p->w->rank) c->me = i;
    }
    CtlComm& mine = p->w->ctl->comm[c->slot];
    mine.users.fetch_add(1, std::memory_order_acq_rel);
    if (!barrier(p)) {                                     // colours / keys may be overwritten by the next split
        mine.users.fetch_sub(1, std::memory_order_acq_rel);
        delete c;
        return ncclSystemError;
    }
    const int bound = mine.users.load(std::memory_order_acquire);
    if (!barrier(p)) {                                     // everyone has read the count before anyone takes its own back
        mine.users.fetch_sub(1, std::memory_order_acq_rel);
        delete c;
        return ncclSystemError;
    }
    if (bound != c->n) {
        // the slot number is a hash of (parent slot, split number, colour): a second live communicator landed on it and the two
        // would share one barrier counter.  Every member of both sees the same count, so all of them refuse together.
        mi355gp_set_error("ipc transport: communicator slot %d is shared by two live communicators (%d members bound, %d expected); "
                          "destroy earlier splits first", c->slot, bound, c->n);
        mine.users.fetch_sub(1, std::memory_order_acq_rel);
        delete c;
        return ncclInvalidUsage;
    }
    p->w->refs += 1;
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ipcCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    World* w = c->w;
    if (c->slot != 0) w->ctl->comm[c->slot].users.fetch_sub(1, std::memory_order_acq_rel);
    delete c;
    if (--w->refs == 0) world_release(w);
    return ncclSuccess;
}

ncclResult_t ipcBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm,
                          hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (type != ncclFloat64 || root < 0 || root >= c->n) return ncclInvalidArgument;
    World* w = c->w;
    const size_t cap = IPC_STAGE_BYTES / sizeof(double);
    const bool is_root = (c->me == root);
    double* stage = w->stage[c->global[root]];
    for (size_t off = 0; off < count; off += cap) {
        const size_t nchunk = (count - off < cap) ? count - off : cap;
        if (is_root) {
            if (!copy(w, stage, (const double*)send + off, nchunk * sizeof(double), st)) return ncclUnhandledCudaError;
            if (recv != send && !copy(w, (double*)recv + off, (const double*)send + off, nchunk * sizeof(double), st))
                return ncclUnhandledCudaError;
            if (!sync(w, st)) return ncclUnhandledCudaError;
        }
        if (!barrier(c)) return ncclSystemError;           // the chunk is in the root's staging buffer
        if (!is_root) {
            if (!copy(w, (double*)recv + off, stage, nchunk * sizeof(double), st)) return ncclUnhandledCudaError;
            if (!sync(w, st)) return ncclUnhandledCudaError;
        }
        if (!barrier(c)) return ncclSystemError;           // everybody has read it: the buffer may be overwritten
    }
    return ncclSuccess;
}

ncclResult_t ipcAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                          hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (type != ncclFloat64 || op != ncclSum) return ncclInvalidArgument;
    World* w = c->w;
    if (c->n == 1) {
        if (recv != send && !copy(w, recv, send, count * sizeof(double), st)) return ncclUnhandledCudaError;
        return ncclSuccess;
    }
    const size_t cap = IPC_STAGE_BYTES / sizeof(double);
    const double** dsrc = nullptr;                          // device copy of the pointer table (device mode)
    std::vector<const double*> src((size_t)c->n);
    for (int r = 0; r < c->n; ++r) src[(size_t)r] = w->stage[c->global[r]];
    if (!w->host) {
        if (hipMalloc((void**)&dsrc, sizeof(double*) * c->n) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpy((void*)dsrc, src.data(), sizeof(double*) * c->n, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree((void*)dsrc);
            return ncclUnhandledCudaError;
        }
    }
    ncclResult_t rc = ncclSuccess;
    for (size_t off = 0; off < count && rc == ncclSuccess; off += cap) {
        const size_t nchunk = (count - off < cap) ? count - off : cap;
        if (!copy(w, w->stage[w->rank], (const double*)send + off, nchunk * sizeof(double), st) || !sync(w, st)) {
            rc = ncclUnhandledCudaError;
            break;
        }
        if (!barrier(c)) { rc = ncclSystemError; break; }  // every contribution is staged
        double* out = (double*)recv + off;
        if (w->host) {
            for (size_t i = 0; i < nchunk; ++i) {
                double s = src[0][i];
                for (int r = 1; r < c->n; ++r) s += src[(size_t)r][i];
                out[i] = s;
            }
        } else {
            hipLaunchKernelGGL(k_ipc_sum, dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, st, out, dsrc, c->n, nchunk);
            if (!sync(w, st)) { rc = ncclUnhandledCudaError; break; }
        }
        if (!barrier(c)) rc = ncclSystemError;             // everybody has summed: the buffers may be overwritten
    }
    if (dsrc) (void)hipFree((void*)dsrc);
    return rc;
}

ncclResult_t ipcGroupStart() { return ncclSuccess; }
ncclResult_t ipcGroupEnd() { return ncclSuccess; }
const char* ipcGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "success";
        case ncclInvalidArgument: return "ipc transport: invalid argument";
        case ncclSystemError: return "ipc transport: shared-memory set-up failed or a peer did not arrive within MI355GP_IPC_TIMEOUT_S";
        case ncclUnhandledCudaError: return "ipc transport: a HIP call failed (hipIpcOpenMemHandle needs HSA_ENABLE_IPC_MODE_LEGACY=0 here)";
        default: return "ipc transport: error";
    }
}

// ---- protocol self-test (tests/test_ipc_transport.py; runs in HOST mode on a machine without a GPU) --------------------------
// The communicator set-up of mi355gp_grid_create (world, rows by colour pr / key pc, columns by colour pc / key pr) followed by a
// scripted sequence in the shape of grid.hip's traffic: a world broadcast, one broadcast inside every process row and every
// process column (members only, root given as a grid COORDINATE as in grid_bcast), a world all-reduce and a row all-reduce,
// each `count` doubles (more than one staging buffer: chunked).  out[0] = mismatching doubles on this rank, out[1] = checksum
// of everything received, out[2] / out[3] = this rank's number inside its row / column communicator.
extern "C" int mi355gp_dbg_ipc_selftest(const void* id128, int rank, int world, int Pr, int Pc, int64_t count, double* out) {
    if (!id128 || !out || world != Pr * Pc || count <= 0) return -1;
    const char* e = PRODUCT_ENV("IPC_HOST");
    const bool host = e && atoi(e) != 0;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t cw = nullptr, crow = nullptr, ccol = nullptr;
    if (ipcCommInitRank(&cw, world, id, rank) != ncclSuccess) return -2;
    const int pr = rank / Pc, pc = rank % Pc;
    if (ipcCommSplit(cw, pr, pc, &crow, nullptr) != ncclSuccess) return -3;
    if (ipcCommSplit(cw, pc, pr, &ccol, nullptr) != ncclSuccess) return -3;
    std::vector<double> hsend((size_t)count), hrecv((size_t)count);
    double *dsend = hsend.data(), *drecv = hrecv.data();
    if (!host) {
        if (hipMalloc(&dsend, sizeof(double) * count) != hipSuccess || hipMalloc(&drecv, sizeof(double) * count) != hipSuccess) return -4;
    }
    auto put = [&](const std::vector<double>& v) {
        if (!host) (void)hipMemcpy(dsend, v.data(), sizeof(double) * count, hipMemcpyHostToDevice);
    };
    auto get = [&]() {
        if (!host) (void)hipMemcpy(hrecv.data(), drecv, sizeof(double) * count, hipMemcpyDeviceToHost);
    };
    double bad = 0.0, sum = 0.0;
    auto fill = [&](double tag) {
        for (int64_t i = 0; i < count; ++i) hsend[(size_t)i] = tag + 1e-3 * (double)(i % 1000);
        put(hsend);
    };
    auto expect = [&](double tag) {
        get();
        for (int64_t i = 0; i < count; ++i) {
            const double want = tag + 1e-3 * (double)(i % 1000);
            if (hrecv[(size_t)i] != want) bad += 1.0;
            sum += hrecv[(size_t)i];
        }
    };
    int rc = 0;
    // (1) world broadcast from the last rank, out of place everywhere
    fill(100.0 + rank);
    if (ipcBroadcast(dsend, drecv, (size_t)count, ncclFloat64, world - 1, cw, 0) != ncclSuccess) rc = -5;
    expect(100.0 + (world - 1));
    // (2) inside every process row: root = grid column (1 % Pc), i.e. communicator rank pc of the root
    for (int r = 0; r < Pr && rc == 0; ++r) {
        if (pr != r) continue;                              // grid_bcast: non-members skip the call
        const int rootc = 1 % Pc;
        fill(1000.0 * (r + 1) + pc);
        if (ipcBroadcast(dsend, pc == rootc ? dsend : drecv, (size_t)count, ncclFloat64, rootc, crow, 0) != ncclSuccess) rc = -6;
        if (pc != rootc) expect(1000.0 * (r + 1) + rootc);
    }
    // (3) inside every process column: root = grid row (Pr - 1)
    for (int cidx = 0; cidx < Pc && rc == 0; ++cidx) {
        if (pc != cidx) continue;
        const int rootr = Pr - 1;
        fill(5000.0 * (cidx + 1) + pr);
        if (ipcBroadcast(dsend, pr == rootr ? dsend : drecv, (size_t)count, ncclFloat64, rootr, ccol, 0) != ncclSuccess) rc = -7;
        if (pr != rootr) expect(5000.0 * (cidx + 1) + rootr);
    }
    // (4) world all-reduce in place, (5) row all-reduce out of place: sums in communicator-rank order
    if (rc == 0) {
        fill((double)rank);
        if (ipcAllReduce(dsend, dsend, (size_t)count, ncclFloat64, ncclSum, cw, 0) != ncclSuccess) rc = -8;
        if (!host) (void)hipMemcpy(hrecv.data(), dsend, sizeof(double) * count, hipMemcpyDeviceToHost);
        else hrecv = hsend;
        for (int64_t i = 0; i < count; ++i) {
            double want = 0.0 + 1e-3 * (double)(i % 1000);
            for (int r = 1; r < world; ++r) want += (double)r + 1e-3 * (double)(i % 1000);
            if (hrecv[(size_t)i] != want) bad += 1.0;
            sum += hrecv[(size_t)i];
        }
    }
    if (rc == 0) {
        fill(10.0 * rank);
        if (ipcAllReduce(dsend, drecv, (size_t)count, ncclFloat64, ncclSum, crow, 0) != ncclSuccess) rc = -9;
        get();
        for (int64_t i = 0; i < count; ++i) {
            double want = 10.0 * (pr * Pc) + 1e-3 * (double)(i % 1000);
            for (int c2 = 1; c2 < Pc; ++c2) want += 10.0 * (pr * Pc + c2) + 1e-3 * (double)(i % 1000);
            if (hrecv[(size_t)i] != want) bad += 1.0;
            sum += hrecv[(size_t)i];
        }
    }
    // (6) refusals: a colour outside 0..15, and a split whose slot number hashes onto the still-live process-row communicator
    //     (split number 15 of the world lands where split number 0 did) -- refused by every member, nothing left bound
    if (rc == 0) {
        ncclComm_t t = nullptr;
        if (ipcCommSplit(cw, 16, 0, &t, nullptr) != ncclInvalidArgument) rc = -20;
        for (int s = 2; s < 15 && rc == 0; ++s) {
            if (ipcCommSplit(cw, pr, pc, &t, nullptr) != ncclSuccess) rc = -21;
            else ipcCommDestroy(t);
        }
        if (rc == 0 && Pc > 1 && ipcCommSplit(cw, pr, pc, &t, nullptr) != ncclInvalidUsage) rc = -22;
        if (rc == 0 && Pc > 1 && ipcAllReduce(dsend, drecv, (size_t)count, ncclFloat64, ncclSum, crow, 0) != ncclSuccess) rc = -23;
    }
    out[0] = bad;
    out[1] = sum;
    out[2] = (double)((Comm*)crow)->me;
    out[3] = (double)((Comm*)ccol)->me;
    if (!host) {
        (void)hipFree(dsend);
        (void)hipFree(drecv);
    }
    ipcCommDestroy(crow);
    ipcCommDestroy(ccol);
    ipcCommDestroy(cw);
    return rc;
}
