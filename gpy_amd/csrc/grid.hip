// grid.hip -- optional multi-GPU mode: the exact-GP evaluation on a Pr x Pc process grid with the N x N matrices
// 2D block-cyclic distributed (tile edge nb), panels exchanged with RCCL broadcasts over xGMI and the
// O(N) results combined with all-reduces (north_star config 4).  GPy has no counterpart: its only
// parallel code is the mpi4py data-parallel sparse GP (GPy/inference/latent_function_inference/var_dtc_parallel.py).
//
// One pass, right-looking in all three factors, so every exchange is a panel broadcast and every update a rank-nb
// outer product on the local tiles (the same MFMA tile GEMM as the single-GPU path):
//   step k:  owner(k,k): L_kk = chol(A_kk), D = L_kk^-1          -> bcast D down process column k%Pc and along row k%Pr
//            column k%Pc: L_ik = A_ik D^T (i > k)                 -> bcast along process rows       (row panel RP)
//            RP tiles with i%Pc == my column                      -> bcast down process columns     (col panel CP)
//            A_ij -= L_ik L_jk^T                (i >= j > k)        [potrf trailing update]
//            row k%Pr: X_kj = D B_kj (j < k), X_kk = D            -> bcast down process columns     (XR)
//            XR tiles with j%Pr == my row                         -> bcast along process rows       (XRr)
//            B_ij -= L_ik X_kj                  (i > k >= j)        [right-looking triangular inverse, B -> X = L^-1]
//            W_ij += X_ki^T X_kj                (k >= i >= j)       [Ky^-1 = X^T X accumulated as X rows complete]
// Afterwards alpha = X^T (X R), diag W and the gradient reduction run on the local tiles and are summed across ranks.
//
// Two transports behind one driver: RCCL (one rank per process, one GPU each; librccl.so is dlopen'ed so the
// single-GPU library has no RCCL dependency) and LOOPBACK (all Pr*Pc logical ranks inside one process on one
// device, broadcasts are device copies) used to test the algorithm on a 1-GPU box.
#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/mi355gp.h"
#include "../../include/mi355gp_debug.h"
#include "gemm_tile.h"
#include "internal.h"

#define GP_STRIDE 34
#define LOG_2_PI 1.8378770664093454836
#define ARGCHK(cond, msg)                 \
    do {                                  \
        if (!(cond)) {                    \
            mi355gp_set_error("%s", msg); \
            return -1;                    \
        }                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// RCCL: types, enums and prototypes come from the installed <rccl/rccl.h>; the library itself is dlopen'ed (so the
// single-GPU library has no link-time RCCL dependency) and every entry point is bound through decltype(&ncclXxx): a
// signature change in RCCL is a compile error here, not silent ABI drift.
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == 128, "mi355gp_grid_unique_id hands out 128-byte ids");
// the second provider of the same entry points (ipc_comm.hip): ranks = processes sharing ONE GPU, MI355GP_TRANSPORT=ipc
ncclResult_t ipcGetUniqueId(ncclUniqueId* id);
ncclResult_t ipcCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank);
ncclResult_t ipcCommSplit(ncclComm_t parent, int color, int key, ncclComm_t* out, ncclConfig_t* cfg);
ncclResult_t ipcCommDestroy(ncclComm_t comm);
ncclResult_t ipcBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t st);
ncclResult_t ipcAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t st);
ncclResult_t ipcGroupStart();
ncclResult_t ipcGroupEnd();
const char* ipcGetErrorString(ncclResult_t r);

struct Rccl {
    void* h = nullptr;
    bool ipc = false;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommSplit) CommSplit = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load() {
        if (h) return true;
        {
            const char* t = PRODUCT_ENV("TRANSPORT");
            if (t && strcmp(t, "ipc") == 0) {                // multi-process transport over hipIpc + shared memory (ipc_comm.hip)
                GetUniqueId = ipcGetUniqueId;
                CommInitRank = ipcCommInitRank;
                CommSplit = ipcCommSplit;
                CommDestroy = ipcCommDestroy;
                Broadcast = ipcBroadcast;
                AllReduce = ipcAllReduce;
                GroupStart = ipcGroupStart;
                GroupEnd = ipcGroupEnd;
                GetErrorString = ipcGetErrorString;
                ipc = true;
                h = (void*)this;
                return true;
            }
        }
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            mi355gp_set_error("cannot dlopen librccl.so: %s", dlerror());
            return false;
        }
#define SYM(field, name)                                                   \
    *(void**)(&field) = dlsym(h, name);                                    \
    if (!field) {                                                          \
        mi355gp_set_error("librccl.so lacks %s", name);                    \
        return false;                                                      \
    }
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommSplit, "ncclCommSplit");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(Broadcast, "ncclBroadcast");
        SYM(AllReduce, "ncclAllReduce");
        SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        return true;
    }
};
static Rccl g_rccl;
#define NCCL_CHECK(expr)                                                                                  \
    do {                                                                                                  \
        ncclResult_t _r = (expr);                                                                         \
        if (_r != ncclSuccess) {                                                                          \
            mi355gp_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return -(2000 + (int)_r);                                                                       \
        }                                                                                                 \
    } while (0)

// plain world communicator for the row-sharded sparse path (sparse.hip): one all-reduce per streaming pass
int rccl_comm_create(int rank, int world, const void* id128, void** comm) {
    if (!g_rccl.load()) return -20;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    NCCL_CHECK(g_rccl.CommInitRank(&c, world, id, rank));
    *comm = c;
    return 0;
}
int rccl_allreduce_sum(void* comm, double* buf, size_t count, hipStream_t st) {
    NCCL_CHECK(g_rccl.AllReduce(buf, buf, count, ncclFloat64, ncclSum, (ncclComm_t)comm, st));
    return 0;
}
void rccl_comm_destroy(void* comm) {
    if (comm && g_rccl.h) g_rccl.CommDestroy((ncclComm_t)comm);
}

// ---------------------------------------------------------------------------------------------------
// device kernels specific to the distributed layout

// C (op)= A * B on the 128x128 tiles of a local region; operands may be "tile-major" panels ([tile][nb][nb], ld = nb).
//   mode 0: C = acc, 1: C += acc, 2: C -= acc.   pred: keep only tiles whose global nb-tile indices satisfy I >= J.
struct GridPred {
    int on, Pr, pr, Pc, pc, roff, coff;   // I = (roff + ti/q)*Pr + pr ;  J = (coff + tj/q)*Pc + pc
};
template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void k_grid_gemm(double* __restrict__ C, long ldc, int c_tm,
                                                      const double* __restrict__ A, long lda, int a_tm,
                                                      const double* __restrict__ B, long ldb, int b_tm, int K,
                                                      int ntc, int q, long nb, int mode, GridPred pd) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ti = blockIdx.x / ntc, tj = blockIdx.x % ntc;
    if (pd.on) {
        const int I = (pd.roff + ti / q) * pd.Pr + pd.pr, J = (pd.coff + tj / q) * pd.Pc + pd.pc;
        if (J > I) return;
    }
    const double* Ap;
    const double* Bp;
    if (AK) Ap = A + (long)ti * NB * lda;
    else Ap = a_tm ? A + (long)(ti / q) * nb * nb + (long)(ti % q) * NB : A + (long)ti * NB;
    if (BK) Bp = B + (long)tj * NB * ldb;
    else Bp = b_tm ? B + (long)(tj / q) * nb * nb + (long)(tj % q) * NB : B + (long)tj * NB;
    double* Cp = c_tm ? C + (long)(tj / q) * nb * nb + (long)ti * NB * nb + (long)(tj % q) * NB
                      : C + (long)ti * NB * ldc + (long)tj * NB;
    d4 acc[4][4];
    gt_zero<4>(acc);
    gemm_tile_128<AK, BK, 4>(Ap, lda, Bp, ldb, K, acc, smem);
    if (mode == 0) gt_store<0, 4>(Cp, ldc, acc);
    else if (mode == 1) gt_store<3, 4>(Cp, ldc, acc, 1.0, 1.0);
    else gt_store<2, 4>(Cp, ldc, acc);
}

struct GOp {
    const double* p;
    long ld;
    int tm;
};
template <bool AK, bool BK>
static void grid_gemm_t(hipStream_t st, double* C, long ldc, int c_tm, GOp a, GOp b, long M, long N, long K, long nb,
                        int mode, GridPred pd) {
    if (M <= 0 || N <= 0) return;
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_gemm<AK, BK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES);
        opted = true;
    }
    const int ntr = (int)(M / NB), ntc = (int)(N / NB);
    hipLaunchKernelGGL((k_grid_gemm<AK, BK>), dim3((unsigned)(ntr * ntc)), dim3(256), GT_LDS_BYTES, st, C, ldc, c_tm,
                       a.p, a.ld, a.tm, b.p, b.ld, b.tm, (int)K, ntc, (int)(nb / NB), nb, mode, pd);
}

// The aggregated updates: C (op)= sum over the panels k in [ks, k1) of A_k * B_k on the 128x128 tiles of a local region,
// ONE pass over C for the whole panel list (the tile pipeline rolls across panels: gemm_tile_128_v3_multi).
//   MODE 0: C = acc,  1: C += acc,  2: C -= acc   (1 / 2 read C before the k-loop: its latency hides behind the prologue)
//   Atab[k] / Btab[k]: base of panel k such that LOCAL nb-tile l of the operand starts at base + l * nb * nb
//       AK / BK = true : the panel is a tall [rows][nb] matrix (k contiguous): L_ik row panels / L_jk column panels
//       AK / BK = false: the panel is a sequence of [nb(k)][nb] tiles (m/n contiguous): X_ki / X_kj row panels
//   ti0 / tj0: first local 128-tile row / column of the region.   pd: keep tiles with I >= J (global nb-tile indices).
//   kpred: the first panel that touches tile (I, J):  0: k0;  1: max(k0, J) (B -= L X: X_kj = 0 for j > k);
//          2: max(k0, I, J) (W += X^T X).  A tile with no panel left keeps its C (MODE 0: stores zeros).
// Live-tile enumeration of a region under the lower predicate: pre[i] = number of live nb-tile pairs in the region's
// rows before row i (host-computed per launch, passed BY VALUE in the kernel arguments: <= 4 KB).  Consecutive
// workgroups then are consecutive LIVE tiles: a plain rows x columns grid whose dead half exits at once runs the same
// tiles 30 % slower (N=16384 X^T X: 48 vs 69 TF/s, mi355gp_dbg_grid_multi) -- workgroups go to the 8 XCDs round-robin
// by index, dead ones included, and the XCDs drift out of balance.
#define GRID_ROWTAB_MAX 960
struct GridRowTab {
    int n;                                    // rows in the table; 0 = plain rows x columns enumeration
    int pre[GRID_ROWTAB_MAX + 1];
};
template <bool AK, bool BK, int MODE>
__global__ __launch_bounds__(256, 2) void k_grid_gemm_multi(double* __restrict__ C, long ldc,
                                                            const double* const* __restrict__ Atab,
                                                            const double* const* __restrict__ Btab, int k0, int k1,
                                                            int ti0, int tj0, int ntc, int q, long nb, int kpred,
                                                            GridPred pd, long ldo, int tm, GridRowTab rt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int ti, tj;
    if (tm == 2) {                                           // experiment: triangular enumeration of a square region at (0, 0)
        const int bid = blockIdx.x;
        ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((long)ti * (ti + 1) / 2 > bid) --ti;
        while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
        tj = bid - (int)((long)ti * (ti + 1) / 2);
    } else if (rt.n > 0) {                                   // live nb-tile pairs in row order, q x q tiles per pair
        const int qq = q * q, p = (int)(blockIdx.x / qq), sub = (int)(blockIdx.x % qq);
        int lo = 0, hi = rt.n - 1;                           // last row with pre[row] <= p
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rt.pre[mid] <= p) lo = mid;
            else hi = mid - 1;
        }
        ti = ti0 + lo * q + sub / q;
        tj = tj0 + (p - rt.pre[lo]) * q + sub % q;
    } else {
        ti = ti0 + blockIdx.x / ntc;
        tj = tj0 + blockIdx.x % ntc;
    }
    const int I = (ti / q) * pd.Pr + pd.pr, J = (tj / q) * pd.Pc + pd.pc;
    if (pd.on && J > I) return;
    int ks = k0;
    if (kpred >= 1 && J > ks) ks = J;
    if (kpred >= 2 && I > ks) ks = I;
    double* Ct = C + (long)ti * NB * ldc + (long)tj * NB;
    d4 acc[4][4];
    if (ks >= k1) {
        if (MODE == 0) {
            gt_zero<4>(acc);
            gt_store<0, 4>(Ct, ldc, acc);
        }
        return;
    }
    if (MODE == 0) gt_zero<4>(acc);
    else gt_load_buf<4>(Ct, ldc, acc);
    // ldo / tm: operand row stride and tile-major flag (the panel stores: ldo = nb, tm = 1; the microbenchmark also reads a
    // plain row-major matrix: ldo = its leading dimension, tm = 0)
    const long aoff = AK ? (long)ti * NB * ldo : (tm == 1 ? (long)(ti / q) * nb * nb + (long)(ti % q) * NB : (long)ti * NB);
    const long boff = BK ? (long)tj * NB * ldo : (tm == 1 ? (long)(tj / q) * nb * nb + (long)(tj % q) * NB : (long)tj * NB);
    gemm_tile_128_v3_multi<AK, BK, MODE == 2>(Atab, aoff, ldo, Btab, boff, ldo, ks, k1, (int)(nb / 16), acc, smem);
    gt_store<0, 4>(Ct, ldc, acc);
}

// region [ti0, ti1) x [tj0, tj1) in local nb-tiles
template <bool AK, bool BK, int MODE>
static void grid_gemm_multi(hipStream_t st, double* C, long ldc, const double* const* Atab, const double* const* Btab,
                            int k0, int k1, int ti0, int ti1, int tj0, int tj1, long nb, int kpred, GridPred pd,
                            long ldo = 0, int tm = 1) {
    if (ti1 <= ti0 || tj1 <= tj0 || k1 <= k0) return;
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_gemm_multi<AK, BK, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES);
        opted = true;
    }
    const int q = (int)(nb / NB), ntr = (ti1 - ti0) * q, ntc = (tj1 - tj0) * q;
    long nblk = (tm == 2) ? (long)ntr * (ntr + 1) / 2 : (long)ntr * ntc;
    GridRowTab rt;
    rt.n = 0;
    if (pd.on && tm != 2 && ti1 - ti0 <= GRID_ROWTAB_MAX) {
        // live columns of local row li: lj with lj * Pc + pc <= li * Pr + pr, clipped to [tj0, tj1) (a prefix of the range)
        long tot = 0;
        int first = -1, last = -1;
        for (int li = ti0; li < ti1; ++li) {
            const long I = (long)li * pd.Pr + pd.pr;
            long m = (I >= pd.pc) ? (I - pd.pc) / pd.Pc + 1 : 0;             // local columns with J <= I
            long c = m - tj0;
            if (c < 0) c = 0;
            if (c > tj1 - tj0) c = tj1 - tj0;
            if (c > 0) {
                if (first < 0) first = li;
                last = li;
            }
            if (first >= 0) rt.pre[li - first] = (int)tot;
            tot += c;
        }
        if (tot == 0) return;
        rt.n = last - first + 1;
        rt.pre[rt.n] = (int)tot;
        ti0 = first;
        nblk = tot * q * q;
    }
    hipLaunchKernelGGL((k_grid_gemm_multi<AK, BK, MODE>), dim3((unsigned)nblk), dim3(256), GT_LDS_BYTES, st, C, ldc,
                       Atab, Btab, k0, k1, ti0 * q, tj0 * q, ntc, q, nb, kpred, pd, ldo ? ldo : nb, tm, rt);
}

// local diagonal fix-up after the cross-covariance build: entries with equal global index get noise + jitter
// (real points) or 1 (padding); one thread per element of every diagonal nb-tile this rank owns.
__global__ void k_grid_fix_diag(double* __restrict__ A, long ld, long nb, long n, int ndiag,
                                const int* __restrict__ dl_r, const int* __restrict__ dl_c,
                                const int* __restrict__ dg, const double* __restrict__ noise, long noise_len,
                                double jit) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)ndiag * nb) return;
    const int d = (int)(idx / nb);
    const long e = idx % nb, g = (long)dg[d] * nb + e;
    double* p = A + ((long)dl_r[d] * nb + e) * ld + (long)dl_c[d] * nb + e;
    if (g < n) *p += noise[noise_len > 1 ? g : 0] + jit;
    else *p = 1.0;
}

// G = w * 0.5 * (alpha_i . alpha_j - Dy * W_ij), w = 2 (global i > j), 1 (i == j), 0 (i < j): the lower-triangle
// weighting of the symmetric dL_dK (exact_gaussian_inference.py:70) on a block-cyclic local tile set.
__global__ void k_grid_dldk(const double* __restrict__ W, double* __restrict__ G, long ld, long rows, long cols,
                            const long* __restrict__ gr, const long* __restrict__ gc,
                            const double* __restrict__ alpha, int Dy, long n) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= cols || i >= rows) return;
    const long gi = gr[i], gj = gc[j];
    double g = 0.0;
    if (gi < n && gj < n && gj <= gi) {
        double aa = 0.0;
        for (int d = 0; d < Dy; ++d) aa = fma(alpha[gi * Dy + d], alpha[gj * Dy + d], aa);
        g = 0.5 * (aa - (double)Dy * W[i * ld + j]);
        if (gj < gi) g *= 2.0;
    }
    G[i * ld + j] = g;
}

// y[i][d] = sum_j M[i][j] * v[g(j)][d]   (one wave per local row; v indexed by global column index, 0 beyond n)
__global__ __launch_bounds__(256) void k_grid_row_reduce(const double* __restrict__ M, long ld, long rows, long cols,
                                                         const long* __restrict__ gc, const double* __restrict__ v,
                                                         int Dy, int d, long n, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    double s = 0.0;
    for (long j = lane; j < cols; j += 64) {
        const long g = gc[j];
        if (g < n) s = fma(M[i * ld + j], v[g * Dy + d], s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) y[i] = s;
}

// out[j] = sum_i M[i][j] * (v ? v[g(i)][d] : M[i][j]) in two fixed-order stages: partials per chunk of CR_ROWS rows (64
// columns x CR_ROWS rows per block), then the sum over chunks.  M = X = L^-1 is lower block triangular in GLOBAL nb-tiles:
// a chunk whose tile row lies above the column's tile contributes exact zeros and is not read.
#define CR_ROWS 128
__global__ __launch_bounds__(256) void k_grid_col_reduce_part(const double* __restrict__ M, long ld, long rows, long cols,
                                                              const long* __restrict__ gr, const double* __restrict__ v,
                                                              int Dy, int d, long n, double* __restrict__ part, long nb,
                                                              int Pr, int pr, int Pc, int pc) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + tx, i0 = (long)blockIdx.y * CR_ROWS;
    const long I = (i0 / nb) * Pr + pr, J = (((long)blockIdx.x * 64) / nb) * Pc + pc;
    double s = 0.0;
    if (j < cols && I >= J) {
        const long i1 = (i0 + CR_ROWS < rows) ? i0 + CR_ROWS : rows;
        for (long i = i0 + g; i < i1; i += 4) {
            const double x = M[i * ld + j];
            if (v) {
                const long gi = gr[i];
                if (gi < n) s = fma(x, v[gi * Dy + d], s);
            } else {
                s = fma(x, x, s);
            }
        }
    }
    red[g][tx] = s;
    __syncthreads();
    if (g == 0 && j < cols) part[(long)blockIdx.y * cols + j] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
__global__ void k_grid_col_combine(const double* __restrict__ part, long nchunks, long cols, double* __restrict__ out) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    double s = 0.0;
    for (long c = 0; c < nchunks; ++c) s += part[c * cols + j];
    out[j] = s;
}

// gvec[g(l)*stride + d] = loc[l]  for g(l) < n
__global__ void k_grid_scatter(const double* __restrict__ loc, long cnt, const long* __restrict__ gidx, long n,
                               int stride, int d, double* __restrict__ gvec) {
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= cnt) return;
    const long g = gidx[l];
    if (g < n) gvec[g * stride + d] = loc[l];
}

__global__ void k_grid_axpy(double* __restrict__ dst, const double* __restrict__ src, long cnt) {
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l < cnt) dst[l] += src[l];
}

// after a tile factorisation: scal[0] += sum(logsum[0..q)) ; info_g = first failure in global numbering
__global__ void k_grid_tile_stats(const double* __restrict__ logsum, int q, const int* __restrict__ info_tile,
                                  long col0, double* __restrict__ scal, int* __restrict__ info_g) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < q; ++i) s += logsum[i];
    scal[0] += s;
    if (info_tile[0] >= PS_ABORT_INFO) {
        scal[2] += 1.0;                                  // a persistent tile factorisation did not run: every rank redoes the evaluation
    } else if (info_tile[0] != 0) {
        scal[1] += 1.0;                                  // summed over ranks: "some tile failed" is known everywhere
        if (info_g[0] == 0) info_g[0] = (int)(col0 + info_tile[0]);
    }
}

// ---------------------------------------------------------------------------------------------------
struct GridRank {
    int rank = 0, pr = 0, pc = 0;
    int TLr = 0, TLc = 0;            // local tile rows / cols actually owned
    long LR = 0, LC = 0;             // allocated local rows / cols (uniform over ranks)
    long nvr = 0, nvc = 0;           // local rows / cols whose global index is < n (a prefix of the local order)
    double *A = nullptr, *X = nullptr, *W = nullptr;
    // Panel stores (alloc_panel_stores):
    //   RPs: L_ik for the local row tiles with I > k          CPs: L_jk for the local column tiles with J > k
    //   XRs: X_kj for the local column tiles with J <= k      XRrs: X_ki for the local row tiles with I <= k
    // The L panels of a step are read only by the updates of its own group (near, part 1 on sc; bulk on st), and the
    // critical path of group g+2 starts after sc has waited for bulk(g): they live in a RING of 2 G full-height slots (panel
    // k in slot k % 2G), 2 G (N/Pr + N/Pc) nb doubles instead of ~N^2/2 (1/Pr + 1/Pc).  The X panels are also read by the
    // deferred W = X^T X on the low-priority stream, which may trail by any number of steps (all of them with GW = 0): every
    // step keeps its own, packed back to back.
    // hRP[k] .. are the panels' VIRTUAL bases (local tile l of panel k at base + l * nb * nb; only the tiles the panel holds
    // are ever addressed), dRP .. the same tables in device memory for k_grid_gemm_multi.
    double *RPs = nullptr, *CPs = nullptr, *XRs = nullptr, *XRrs = nullptr;
    long ring_slots = 0;             // L-panel ring slots the stores were sized for (2 G at allocation time)
    std::vector<double*> hRP, hCP, hXR, hXRr;
    const double **dRP = nullptr, **dCP = nullptr, **dXR = nullptr, **dXRr = nullptr;
    double* cpart = nullptr;         // column-reduction partials [row chunk][LC]
    double *Dt = nullptr, *Dv = nullptr, *Ds = nullptr;
    double* bstage = nullptr;        // one-rank-per-process transports: packed tiles of a strided panel broadcast (grid_bcast_tile_runs)
    double *XtR = nullptr, *XtC = nullptr, *XsR = nullptr, *XsC = nullptr;   // scaled dimension-major / raw row-major
    long *gR = nullptr, *gC = nullptr;                                       // global index of every local row / col
    int *dl_r = nullptr, *dl_c = nullptr, *dg = nullptr;
    int ndiag = 0;
    double *vloc = nullptr, *gvec = nullptr, *gvec2 = nullptr, *alpha = nullptr, *ybuf = nullptr, *Rg = nullptr;
    double *scal = nullptr, *gradPart = nullptr, *gradOut = nullptr, *invls = nullptr, *noise = nullptr;
    int* info_g = nullptr;
    // Log of the collectives this rank took part in during the current evaluation, per communicator (0 world, 1 process row,
    // 2 process column): count and an FNV-1a hash over (operation, root, doubles).  RCCL matches collectives by ORDER: two members
    // of a communicator that enqueue different sequences dead-lock or, worse, exchange the wrong panels.  check_seq compares
    // the logs of all members after every evaluation (mi355gp_grid_coll_log reads them).
    uint64_t coll_hash[3] = {0, 0, 0};
    long coll_count[3] = {0, 0, 0};
    double* seqbuf = nullptr;        // 3 * max(world, Pr, Pc) doubles: the exchange buffer of the sequence check
    int seqbuf_members = 0;          // members it was sized for
    FactorWs ws;
    hipStream_t st = nullptr;        // bulk updates and everything outside the factorisation loop
    hipStream_t sc = nullptr;        // the critical path of a step: diagonal tile, panel solves, all panel broadcasts
    hipStream_t sw = nullptr;        // W = X^T X updates
};

// loopback transport: destinations of one broadcast / the broadcasts of one group (kernel arguments of k_bcast_copy / k_bcast_batch)
#define BCAST_MAX_DST 8
#define BCAST_BATCH 24
struct BcastDst { double* p[BCAST_MAX_DST]; };
struct BcastItem { const double* src; double* dst[BCAST_MAX_DST]; int nd; };
struct BcastBatch { BcastItem it[BCAST_BATCH]; };

struct mi355gp_grid {
    int device = 0, world = 1, Pr = 1, Pc = 1, my_rank = 0;
    long nb = 512;
    bool loopback = true;
    std::vector<GridRank> ranks;     // logical ranks hosted by this process (loopback: all, RCCL: one)
    ncclComm_t comm_world = nullptr, comm_row = nullptr, comm_col = nullptr;
    hipStream_t st = nullptr;        // loopback: shared by all logical ranks
    hipStream_t sc = nullptr;        // second stream (high priority): panel factorisation + broadcasts, one step ahead
    hipStream_t sw = nullptr;        // third stream (low priority): the W = X^T X updates, which nothing waits for until the end
    hipEvent_t ev_w = nullptr;
    std::vector<hipEvent_t> ev_cr, ev_p1;   // [k]: panels of step k are in place / the part-1 updates of step k are done
    int lookahead = 1;               // MI355GP_GRID_LOOKAHEAD=0: everything in order on one stream
    int check_seq = 0;               // MI355GP_GRID_CHECK_SEQ=1: compare the collective logs of all ranks after every evaluation
    int G = 1;                       // MI355GP_GRID_G: steps per group of the two-level blocked factorisation (K = G * nb updates)
    int GW = 4;                      // MI355GP_GRID_GW: steps per W = X^T X update; 0 = one deep-K pass after the last step
    // diagnostics build, MI355GP_GRID_DBG_CRIT (WRONG RESULTS; tools/grid_crit_probe.py): 1 = every update (near / part1 / bulk /
    // W) skipped: the critical path crit(k) of every step alone; 2 = ... and its broadcasts skipped; 3 = ... and only phase (a),
    // the factorisation + inverse of the diagonal tile, left
    int dbg_crit = 0;
    long n = 0, npad = 0, T = 0;
    int D = 0, Dy = 0;
    hipEvent_t ev[6] = {};
    bool have_result = false;
    // Pr = Pc = 1 over the loopback transport degenerates to the dedicated single-GPU pipeline ("with Pr = Pc = 1
    // degenerate to the single-GPU path"): deep-K trtri / lauum tiles instead of three K = nb read-modify-write updates per
    // step (N=32768: 527 vs 623 ms).  MI355GP_GRID_FORCE_GENERIC=1 keeps the generic one-pass code (tests).
    mi355gp_ctx* single = nullptr;
    // loopback transport: the broadcasts of an open group, launched together at its end
    bool batching = false;
    size_t batch_count = 0;
    hipStream_t batch_stream = nullptr;
    std::vector<BcastItem> batch;
};

// process defaults of the schedule options: environment, else built-in
static void grid_default_option(mi355gp_grid* g, int o) {
    if (o == MI355GP_GRID_OPT_LOOKAHEAD) {
        const char* e = PRODUCT_ENV("GRID_LOOKAHEAD");
        g->lookahead = (e && *e) ? (atoi(e) ? 1 : 0) : 1;
    } else if (o == MI355GP_GRID_OPT_G) {
        const char* e = DIAG_ENV("GRID_G");
        g->G = (e && *e && atoi(e) >= 1) ? atoi(e) : 1;
    } else if (o == MI355GP_GRID_OPT_GW) {
        const char* e = DIAG_ENV("GRID_GW");
        g->GW = (e && *e && atoi(e) >= 0) ? atoi(e) : 4;
    } else if (o == MI355GP_GRID_OPT_CHECK_SEQ) {
        const char* e = PRODUCT_ENV("GRID_CHECK_SEQ");
        g->check_seq = (e && *e && atoi(e)) ? 1 : 0;
    }
}

static void coll_log(GridRank& r, int comm, int op, int root, size_t count) {
    uint64_t h = r.coll_hash[comm];
    const uint64_t words[3] = {(uint64_t)op, (uint64_t)(unsigned)root, (uint64_t)count};
    for (uint64_t w : words)
        for (int b = 0; b < 8; ++b) {
            h ^= (w >> (8 * b)) & 0xffu;
            h *= 1099511628211ull;
        }
    r.coll_hash[comm] = h;
    r.coll_count[comm] += 1;
}

static int cnt_le(long k, int p, int P) { return (k >= p) ? (int)((k - p) / P + 1) : 0; }   // tiles t <= k with t % P == p
static int cnt_lt(long k, int p, int P) { return (k > 0) ? cnt_le(k - 1, p, P) : 0; }

static void free_rank(GridRank& r) {
    void* ptrs[] = {r.A, r.X, r.W, r.RPs, r.CPs, r.XRs, r.XRrs, (void*)r.dRP, (void*)r.dCP, (void*)r.dXR, (void*)r.dXRr, r.cpart, r.Dt, r.Dv, r.Ds, r.bstage, r.XtR, r.XtC, r.XsR, r.XsC, r.gR, r.gC,
                    r.dl_r, r.dl_c, r.dg, r.vloc, r.gvec, r.gvec2, r.alpha, r.ybuf, r.Rg, r.scal, r.gradPart,
                    r.gradOut, r.invls, r.noise, r.info_g, r.seqbuf};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    factor_ws_free(&r.ws);
    r = GridRank();
}

// ---- transports ------------------------------------------------------------------------------------------
// Broadcast inside one process row (group = GROUP_ROW, index = pr) or column (GROUP_COL, index = pc).
// `buf(rank, is_root)` returns the send pointer for the root and the receive pointer for everyone (the root's
// receive pointer may differ from its send pointer: out-of-place on the root, like ncclBroadcast).
enum { GROUP_ROW = 0, GROUP_COL = 1 };
// loopback transport: a broadcast is a device copy from the root's buffer into every other member's
__global__ __launch_bounds__(256) void k_bcast_copy(const double* __restrict__ src, BcastDst dst, int nd, long count) {
    const long n2 = count >> 1;                               // tiles and panels are 16-byte aligned and even-sized
    const d2* s2 = reinterpret_cast<const d2*>(src);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        const d2 v = s2[i];
        for (int q = 0; q < nd; ++q) reinterpret_cast<d2*>(dst.p[q])[i] = v;
    }
    if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        for (int q = 0; q < nd; ++q) dst.p[q][count - 1] = src[count - 1];
}
// a GROUP of loopback broadcasts (grid_group_start .. grid_group_end: the per-tile broadcasts of crit(k), up to k + 1 of them
// with different roots) as ONE launch per 24 of them instead of one per broadcast: blockIdx.y picks the broadcast
__global__ __launch_bounds__(256) void k_bcast_batch(BcastBatch b, long count) {
    const BcastItem& it = b.it[blockIdx.y];
    const long n2 = count >> 1;
    const d2* s2 = reinterpret_cast<const d2*>(it.src);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        const d2 v = s2[i];
        for (int q = 0; q < it.nd; ++q) reinterpret_cast<d2*>(it.dst[q])[i] = v;
    }
    if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        for (int q = 0; q < it.nd; ++q) it.dst[q][count - 1] = it.src[count - 1];
}
// nt tiles of nb x nb, contiguous at src, into the row block dst (row stride ldd): tile t at columns t * nb ..  (one launch
// instead of one 2-D copy per tile: 7,000 of them per evaluation at N = 32768 on a 2 x 4 grid)
__global__ __launch_bounds__(256) void k_tiles_to_rowblock(double* __restrict__ dst, long ldd, const double* __restrict__ src, int nb) {
    const double* s = src + (long)blockIdx.y * nb * nb;
    double* d = dst + (long)blockIdx.y * nb;
    const int half = nb >> 1;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)nb * half; e += (long)gridDim.x * blockDim.x) {
        const long i = e / half, j = 2 * (e - i * half);
        *reinterpret_cast<d2*>(d + i * ldd + j) = *reinterpret_cast<const d2*>(s + i * nb + j);
    }
}
static void launch_bcast_copy(hipStream_t st, const double* src, const BcastDst& dst, int nd, size_t count) {
    long blocks = (long)((count / 2 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_bcast_copy, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, nd, (long)count);
}
typedef std::function<double*(GridRank&, bool)> BufFn;

// loopback transport: launch the broadcasts collected since grid_group_start
static int grid_flush_bcasts(mi355gp_grid* g) {
    const size_t n = g->batch.size();
    for (size_t i0 = 0; i0 < n; i0 += BCAST_BATCH) {
        BcastBatch b;
        const int m = (int)((n - i0 < BCAST_BATCH) ? n - i0 : BCAST_BATCH);
        for (int q = 0; q < m; ++q) b.it[q] = g->batch[i0 + (size_t)q];
        long blocks = (long)((g->batch_count / 2 + 255) / 256);
        if (blocks > 256) blocks = 256;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_bcast_batch, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, g->batch_stream, b, (long)g->batch_count);
    }
    g->batch.clear();
    g->batch_count = 0;
    HIP_CHECK(hipGetLastError());
    return 0;
}
static int grid_bcast(mi355gp_grid* g, int group, int index, int root_coord, size_t count, const BufFn& buf) {
    if (count == 0) return 0;
    hipStream_t lst = g->lookahead ? g->sc : g->st;            // every panel broadcast travels on the communication stream
    if (g->loopback) {
        GridRank* root = nullptr;
        for (GridRank& r : g->ranks) {
            const bool in = (group == GROUP_ROW) ? (r.pr == index) : (r.pc == index);
            const int coord = (group == GROUP_ROW) ? r.pc : r.pr;
            if (in && coord == root_coord) root = &r;
        }
        const double* src = buf(*root, true);
        if (g->batching && (g->batch_count != count || g->batch_stream != lst)) {
            if (int rc = grid_flush_bcasts(g)) return rc;      // (a group of equal-sized tiles in practice: one flush at its end)
        }
        BcastItem item;                                       // every member's copy in ONE launch (one blit per member cost the
        item.src = src;                                       // loopback run ~13,000 copy launches per evaluation at N = 32768)
        item.nd = 0;
        auto emit = [&]() {
            if (item.nd == 0) return;
            if (g->batching) {
                g->batch_count = count;
                g->batch_stream = lst;
                g->batch.push_back(item);
            } else {
                BcastDst dsts;
                for (int q = 0; q < item.nd; ++q) dsts.p[q] = item.dst[q];
                launch_bcast_copy(lst, src, dsts, item.nd, count);
            }
            item.nd = 0;
        };
        for (GridRank& r : g->ranks) {
            const bool in = (group == GROUP_ROW) ? (r.pr == index) : (r.pc == index);
            if (!in) continue;
            coll_log(r, group == GROUP_ROW ? 1 : 2, 1, root_coord, count);
            double* dst = buf(r, false);
            if (dst == src) continue;
            if (item.nd == BCAST_MAX_DST) emit();
            item.dst[item.nd++] = dst;
        }
        emit();
        return 0;
    }
    GridRank& r = g->ranks[0];
    const bool in = (group == GROUP_ROW) ? (r.pr == index) : (r.pc == index);
    if (!in) return 0;
    const int coord = (group == GROUP_ROW) ? r.pc : r.pr;
    const bool is_root = coord == root_coord;
    const double* send = is_root ? buf(r, true) : buf(r, false);
    coll_log(r, group == GROUP_ROW ? 1 : 2, 1, root_coord, count);
    NCCL_CHECK(g_rccl.Broadcast(send, buf(r, false), count, ncclFloat64, root_coord,
                                group == GROUP_ROW ? g->comm_row : g->comm_col, lst));
    return 0;
}
static int grid_group_start(mi355gp_grid* g) {
    if (!g->loopback) NCCL_CHECK(g_rccl.GroupStart());
    else g->batching = true;
    return 0;
}
static int grid_group_end(mi355gp_grid* g) {
    if (!g->loopback) {
        NCCL_CHECK(g_rccl.GroupEnd());
        return 0;
    }
    g->batching = false;
    return grid_flush_bcasts(g);
}
// ---- panel tiles that change hands between process rows / columns: ONE broadcast per (communicator, root) ----------------------
// Phases (e) and (i) of crit(k) move single tiles whose source and destination are both determined by the tile's GLOBAL index:
// tile j of a column panel lives on process row j % Pr (local tile j / Pr of the row panel there) and is wanted by process
// column j % Pc (local tile j / Pc of its column panel).  For one communicator and one root the tiles form an arithmetic
// progression in j with step lcm(Pr, Pc): a run of `n` tiles at local index s0 + m * ss on the root and d0 + m * ds on every
// member.  Rounds 2-5 sent every tile as its own broadcast inside a group (up to T - k - 1 of them per step: ~4,000 RCCL calls
// per evaluation at N = 32768); now a run travels as ONE broadcast of n tiles: packed into the staging buffer by one small kernel
// where the root's tiles are not contiguous (ss != 1), received in place where the members' are (ds == 1) and unpacked by one
// kernel where they are not.  On a Pr x Pc grid with Pr | Pc (2 x 4) a column panel needs no unpack and an X panel no pack.
// The loopback transport moves the same tiles with its batched copy kernel and LOGS the merged broadcast, so that the collective
// sequences of the two transports stay comparable (tests/test_gpu_multiproc.py).
struct TileRun { int group, index, root, n; long s0, ss, d0, ds; };
__global__ __launch_bounds__(256) void k_tiles_restride(double* __restrict__ dst, long dstride, const double* __restrict__ src,
                                                        long sstride, long tile) {
    const d2* s2 = reinterpret_cast<const d2*>(src + (long)blockIdx.y * sstride);
    d2* q2 = reinterpret_cast<d2*>(dst + (long)blockIdx.y * dstride);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (tile >> 1); i += (long)gridDim.x * blockDim.x) q2[i] = s2[i];
}
typedef std::function<double*(GridRank&)> BaseFn;
static int grid_bcast_tile_runs(mi355gp_grid* g, const std::vector<TileRun>& runs, size_t tile, const BaseFn& srcbase,
                                const BaseFn& dstbase) {
    hipStream_t lst = g->lookahead ? g->sc : g->st;
    if (g->loopback) {
        if (int rc = grid_group_start(g)) return rc;
        for (const TileRun& tr : runs) {
            if (tr.n <= 0) continue;
            for (GridRank& r : g->ranks) {
                const bool in = (tr.group == GROUP_ROW) ? (r.pr == tr.index) : (r.pc == tr.index);
                if (in) coll_log(r, tr.group == GROUP_ROW ? 1 : 2, 1, tr.root, (size_t)tr.n * tile);
            }
            for (int m = 0; m < tr.n; ++m) {
                if (g->batch_count != tile || g->batch_stream != lst)
                    if (int rc = grid_flush_bcasts(g)) return rc;
                BcastItem item;
                item.nd = 0;
                item.src = nullptr;
                for (GridRank& r : g->ranks) {
                    const bool in = (tr.group == GROUP_ROW) ? (r.pr == tr.index) : (r.pc == tr.index);
                    const int coord = (tr.group == GROUP_ROW) ? r.pc : r.pr;
                    if (in && coord == tr.root) item.src = srcbase(r) + (tr.s0 + m * tr.ss) * (long)tile;
                }
                auto emit = [&]() {
                    if (item.nd == 0) return;
                    g->batch_count = tile;
                    g->batch_stream = lst;
                    g->batch.push_back(item);
                    item.nd = 0;
                };
                for (GridRank& r : g->ranks) {
                    const bool in = (tr.group == GROUP_ROW) ? (r.pr == tr.index) : (r.pc == tr.index);
                    if (!in) continue;
                    if (item.nd == BCAST_MAX_DST) emit();
                    item.dst[item.nd++] = dstbase(r) + (tr.d0 + m * tr.ds) * (long)tile;
                }
                emit();
            }
        }
        return grid_group_end(g);
    }
    GridRank& r = g->ranks[0];
    struct Mine { const TileRun* tr; bool is_root; const double* send; double* recv; double* stage; };
    std::vector<Mine> mine;
    long used = 0;
    for (const TileRun& tr : runs) {
        if (tr.n <= 0) continue;
        const bool in = (tr.group == GROUP_ROW) ? (r.pr == tr.index) : (r.pc == tr.index);
        if (!in) continue;
        Mine m;
        m.tr = &tr;
        m.is_root = ((tr.group == GROUP_ROW) ? r.pc : r.pr) == tr.root;
        const bool need_stage = (m.is_root && tr.ss != 1 && tr.n > 1) || (tr.ds != 1 && tr.n > 1);
        m.stage = need_stage ? r.bstage + used * (long)tile : nullptr;
        if (need_stage) used += tr.n;
        double* dst0 = dstbase(r) + tr.d0 * (long)tile;
        m.recv = (tr.ds != 1 && tr.n > 1) ? m.stage : dst0;
        m.send = m.recv;
        if (m.is_root) {
            const double* src0 = srcbase(r) + tr.s0 * (long)tile;
            if (tr.ss != 1 && tr.n > 1) {                    // pack the root's tiles
                hipLaunchKernelGGL(k_tiles_restride, dim3(64, (unsigned)tr.n), dim3(256), 0, lst, m.stage, (long)tile, src0,
                                   tr.ss * (long)tile, (long)tile);
                m.send = m.stage;
            } else {
                m.send = src0;
            }
        }
        mine.push_back(m);
    }
    if (mine.empty()) return 0;
    HIP_CHECK(hipGetLastError());
    if (mine.size() > 1) NCCL_CHECK(g_rccl.GroupStart());
    for (const Mine& m : mine) {
        coll_log(r, m.tr->group == GROUP_ROW ? 1 : 2, 1, m.tr->root, (size_t)m.tr->n * tile);
        NCCL_CHECK(g_rccl.Broadcast(m.send, m.recv, (size_t)m.tr->n * tile, ncclFloat64, m.tr->root,
                                    m.tr->group == GROUP_ROW ? g->comm_row : g->comm_col, lst));
    }
    if (mine.size() > 1) NCCL_CHECK(g_rccl.GroupEnd());
    for (const Mine& m : mine)
        if (m.tr->ds != 1 && m.tr->n > 1)                    // unpack into the members' panel
            hipLaunchKernelGGL(k_tiles_restride, dim3(64, (unsigned)m.tr->n), dim3(256), 0, lst,
                               dstbase(r) + m.tr->d0 * (long)tile, m.tr->ds * (long)tile, (const double*)m.stage, (long)tile, (long)tile);
    HIP_CHECK(hipGetLastError());
    return 0;
}
// the tiles t in (lo, hi] ... [lo, hi) with t % Pa == a and t % Pb == b, as first / step / count (step = lcm(Pa, Pb))
static void tile_progression(long lo, long hi, int a, int Pa, int b, int Pb, long* first, long* step, int* count) {
    long L = Pa;
    while (L % Pb != 0) L += Pa;
    *step = L;
    *first = -1;
    *count = 0;
    for (long t = lo; t < hi && t < lo + L; ++t)
        if (t % Pa == a && t % Pb == b) { *first = t; break; }
    if (*first >= 0) *count = (int)((hi - 1 - *first) / L + 1);
}
// ---- self-test of the bound transport (RCCL, or the hipIpc stand-in under MI355GP_TRANSPORT=ipc) --------------------------------
// Exactly the communicator set-up and the call patterns of the per-rank grid code, without the numerics: world communicator,
// row / column communicators by CommSplit (colour = own grid row / column, key = the other coordinate), then
//   (1) a world broadcast, (2) inside ONE group: `rounds` tile broadcasts on the row communicator with rotating roots (the
//   pattern of crit(k) step (i)), (3) the same on the column communicator (steps (d) / (f)), (4) a world all-reduce in place,
//   (5) a row-communicator all-reduce -- all on one non-default stream, every payload checked element by element.
// out4 = [mismatching doubles, checksum, rank inside the row communicator, rank inside the column communicator].
// This is the first thing to run on a node with more than one GPU (tools/rccl_first_light.sh): it separates "RCCL does not do
// what grid.hip assumes" (split numbering, several roots in one group, out-of-place broadcast on the root) from everything else.
extern "C" int mi355gp_dbg_comm_selftest(int device, const void* id128, int rank, int world, int Pr, int Pc, int64_t count,
                                         int rounds, double* out4) {
    if (!id128 || !out4 || world != Pr * Pc || count <= 0 || rounds < 1 || rank < 0 || rank >= world) {
        mi355gp_set_error("mi355gp_dbg_comm_selftest: invalid argument");
        return -1;
    }
    if (!g_rccl.load()) return -2;
    HIP_CHECK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t cw = nullptr, crow = nullptr, ccol = nullptr;
    NCCL_CHECK(g_rccl.CommInitRank(&cw, world, id, rank));
    const int pr = rank / Pc, pc = rank % Pc;
    NCCL_CHECK(g_rccl.CommSplit(cw, pr, pc, &crow, nullptr));
    NCCL_CHECK(g_rccl.CommSplit(cw, pc, pr, &ccol, nullptr));
    hipStream_t st;
    HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t n = (size_t)count, total = n * (size_t)(rounds + 1);
    double *dsend = nullptr, *drecv = nullptr;
    HIP_CHECK(hipMalloc(&dsend, sizeof(double) * total));
    HIP_CHECK(hipMalloc(&drecv, sizeof(double) * total));
    std::vector<double> hs(total), hr(total);
    double bad = 0.0, sum = 0.0;
    auto val = [](double tag, size_t i) { return tag + 1e-3 * (double)(i % 1000); };
    auto fill = [&](int slot, double tag) {
        for (size_t i = 0; i < n; ++i) hs[(size_t)slot * n + i] = val(tag, i);
    };
    auto upload = [&]() -> int {
        HIP_CHECK(hipMemcpyAsync(dsend, hs.data(), sizeof(double) * total, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemsetAsync(drecv, 0, sizeof(double) * total, st));
        return 0;
    };
    auto download = [&]() -> int {
        HIP_CHECK(hipMemcpyAsync(hr.data(), drecv, sizeof(double) * total, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    };
    auto expect = [&](int slot, double tag) {
        for (size_t i = 0; i < n; ++i) {
            const double got = hr[(size_t)slot * n + i];
            if (got != val(tag, i)) bad += 1.0;
            sum += got;
        }
    };
    // (1) world broadcast from the last rank, out of place everywhere (the root included, as grid_bcast does it)
    fill(0, 100.0 + rank);
    if (int rc = upload()) return rc;
    NCCL_CHECK(g_rccl.Broadcast(dsend, drecv, n, ncclFloat64, world - 1, cw, st));
    if (int rc = download()) return rc;
    expect(0, 100.0 + (world - 1));
    // (2) / (3): one GROUP of `rounds` broadcasts with rotating roots on the row, then on the column communicator
    for (int which = 0; which < 2; ++which) {
        const int size = which == 0 ? Pc : Pr, me = which == 0 ? pc : pr, line = which == 0 ? pr : pc;
        const ncclComm_t comm = which == 0 ? crow : ccol;
        for (int q = 0; q < rounds; ++q) fill(q, 1000.0 * (which + 1) + 10.0 * line + me + 0.25 * q);
        if (int rc = upload()) return rc;
        NCCL_CHECK(g_rccl.GroupStart());
        for (int q = 0; q < rounds; ++q)
            NCCL_CHECK(g_rccl.Broadcast(dsend + (size_t)q * n, drecv + (size_t)q * n, n, ncclFloat64, q % size, comm, st));
        NCCL_CHECK(g_rccl.GroupEnd());
        if (int rc = download()) return rc;
        for (int q = 0; q < rounds; ++q) expect(q, 1000.0 * (which + 1) + 10.0 * line + (q % size) + 0.25 * q);
    }
    // (4) world all-reduce in place: sum over ranks of (rank + pattern)
    fill(0, (double)rank);
    if (int rc = upload()) return rc;
    NCCL_CHECK(g_rccl.AllReduce(dsend, dsend, n, ncclFloat64, ncclSum, cw, st));
    HIP_CHECK(hipMemcpyAsync(hr.data(), dsend, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (size_t i = 0; i < n; ++i) {
        double want = 0.0;
        for (int r = 0; r < world; ++r) want += val((double)r, i);          // rank order = the order the loopback transport sums in
        if (fabs(hr[i] - want) > 1e-9 * fabs(want) + 1e-12) bad += 1.0;
        sum += hr[i];
    }
    // (5) row all-reduce out of place
    fill(0, 10.0 * rank);
    if (int rc = upload()) return rc;
    NCCL_CHECK(g_rccl.AllReduce(dsend, drecv, n, ncclFloat64, ncclSum, crow, st));
    if (int rc = download()) return rc;
    for (size_t i = 0; i < n; ++i) {
        double want = 0.0;
        for (int c = 0; c < Pc; ++c) want += val(10.0 * (pr * Pc + c), i);
        if (fabs(hr[i] - want) > 1e-9 * fabs(want) + 1e-12) bad += 1.0;
        sum += hr[i];
    }
    out4[0] = bad;
    out4[1] = sum;
    out4[2] = (double)pc;                                          // key of the row split = grid column: must be the rank inside crow
    out4[3] = (double)pr;
    (void)hipFree(dsend);
    (void)hipFree(drecv);
    (void)hipStreamDestroy(st);
    NCCL_CHECK(g_rccl.CommDestroy(crow));
    NCCL_CHECK(g_rccl.CommDestroy(ccol));
    NCCL_CHECK(g_rccl.CommDestroy(cw));
    return 0;
}

// sum `count` doubles at `pick(rank)` over all ranks, result everywhere
static int grid_allreduce(mi355gp_grid* g, size_t count, const std::function<double*(GridRank&)>& pick) {
    for (GridRank& r : g->ranks) coll_log(r, 0, 2, 0, count);
    if (g->loopback) {
        double* acc = pick(g->ranks[0]);
        const unsigned nblk = (unsigned)((count + 255) / 256);
        for (size_t i = 1; i < g->ranks.size(); ++i)
            hipLaunchKernelGGL(k_grid_axpy, dim3(nblk), dim3(256), 0, g->st, acc, pick(g->ranks[i]), (long)count);
        for (size_t i = 1; i < g->ranks.size(); ++i)
            HIP_CHECK(hipMemcpyAsync(pick(g->ranks[i]), acc, count * sizeof(double), hipMemcpyDeviceToDevice, g->st));
        return 0;
    }
    GridRank& r = g->ranks[0];
    NCCL_CHECK(g_rccl.AllReduce(pick(r), pick(r), count, ncclFloat64, ncclSum, g->comm_world, r.st));
    return 0;
}

// (Re)allocates the four panel stores of a rank and their pointer tables for the current group size G (see GridRank).
static int alloc_panel_stores(mi355gp_grid* g, GridRank& r) {
    const long nb = g->nb, T = g->T;
    const size_t tile = (size_t)nb * nb;
    void* old[] = {r.RPs, r.CPs, r.XRs, r.XRrs, (void*)r.dRP, (void*)r.dCP, (void*)r.dXR, (void*)r.dXRr};
    for (void* p : old)
        if (p) (void)hipFree(p);
    r.RPs = r.CPs = r.XRs = r.XRrs = nullptr;
    r.dRP = r.dCP = r.dXR = r.dXRr = nullptr;
    const long G = g->G < 1 ? 1 : g->G;
    long slots = 2 * G;
    if (slots > T) slots = T;
    r.ring_slots = slots;
    std::vector<long> oXR((size_t)T + 1, 0), oXRr((size_t)T + 1, 0);
    for (long k = 0; k < T; ++k) {
        oXR[k + 1] = oXR[k] + cnt_le(k, r.pc, g->Pc);
        oXRr[k + 1] = oXRr[k] + cnt_le(k, r.pr, g->Pr);
    }
    const size_t bytes[4] = {sizeof(double) * tile * (size_t)(slots * r.TLr + 1), sizeof(double) * tile * (size_t)(slots * r.TLc + 1),
                             sizeof(double) * tile * (size_t)(oXR[T] + 1), sizeof(double) * tile * (size_t)(oXRr[T] + 1)};
    double** dst[4] = {&r.RPs, &r.CPs, &r.XRs, &r.XRrs};
    for (int i = 0; i < 4; ++i)
        if (hipMalloc(dst[i], bytes[i]) != hipSuccess) {
            (void)hipGetLastError();
            mi355gp_set_error("grid mode: out of device memory for the panel stores of rank %d (N=%ld on %dx%d, nb=%ld: L-panel ring "
                              "%.2f + %.2f GB for G=%ld, X panels %.2f + %.2f GB, next to 3 x %.2f GB of local matrices); use more "
                              "GPUs or a smaller G",
                              r.rank, g->n, g->Pr, g->Pc, nb, bytes[0] / 1e9, bytes[1] / 1e9, G, bytes[2] / 1e9, bytes[3] / 1e9,
                              sizeof(double) * (double)r.LR * (double)r.LC / 1e9);
            return -4;
        }
    r.hRP.resize((size_t)T); r.hCP.resize((size_t)T); r.hXR.resize((size_t)T); r.hXRr.resize((size_t)T);
    for (long k = 0; k < T; ++k) {
        r.hRP[k] = r.RPs + (k % slots) * (long)r.TLr * (long)tile;       // full-height slot: local tile l at base + l * tile
        r.hCP[k] = r.CPs + (k % slots) * (long)r.TLc * (long)tile;
        r.hXR[k] = r.XRs + oXR[k] * (long)tile;
        r.hXRr[k] = r.XRrs + oXRr[k] * (long)tile;
    }
    const size_t tb = sizeof(double*) * (size_t)T;
    HIP_CHECK(hipMalloc((void**)&r.dRP, tb));
    HIP_CHECK(hipMalloc((void**)&r.dCP, tb));
    HIP_CHECK(hipMalloc((void**)&r.dXR, tb));
    HIP_CHECK(hipMalloc((void**)&r.dXRr, tb));
    HIP_CHECK(hipMemcpy((void*)r.dRP, r.hRP.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((void*)r.dCP, r.hCP.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((void*)r.dXR, r.hXR.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((void*)r.dXRr, r.hXRr.data(), tb, hipMemcpyHostToDevice));
    return 0;
}

// Debug mode (MI355GP_GRID_CHECK_SEQ=1 / MI355GP_GRID_OPT_CHECK_SEQ): after an evaluation, every member of a communicator must
// have logged the SAME sequence of collectives on it.  Loopback: the logical ranks are compared on the host.  One rank per
// process: every member contributes (hash high, hash low, count) at its own position of a vector that is sum-all-reduced inside
// the communicator (the three numbers are integers below 2^32: exact in fp64), then compares all positions with its own.
static int grid_check_sequences(mi355gp_grid* g) {
    if (g->loopback) {
        for (const GridRank& a : g->ranks)
            for (const GridRank& b : g->ranks) {
                const bool same[3] = {true, a.pr == b.pr, a.pc == b.pc};
                for (int c = 0; c < 3; ++c)
                    if (same[c] && (a.coll_hash[c] != b.coll_hash[c] || a.coll_count[c] != b.coll_count[c])) {
                        mi355gp_set_error("grid mode: ranks %d and %d logged different collective sequences on communicator %d "
                                          "(%ld vs %ld collectives)", a.rank, b.rank, c, a.coll_count[c], b.coll_count[c]);
                        return -7;
                    }
            }
        return 0;
    }
    GridRank& r = g->ranks[0];
    const ncclComm_t comms[3] = {g->comm_world, g->comm_row, g->comm_col};
    const int sizes[3] = {g->world, g->Pc, g->Pr}, me[3] = {r.rank, r.pc, r.pr};
    if (r.seqbuf_members < g->world) {                      // world >= Pr, Pc: one buffer serves the three communicators
        if (r.seqbuf) (void)hipFree(r.seqbuf);
        r.seqbuf = nullptr;
        HIP_CHECK(hipMalloc(&r.seqbuf, sizeof(double) * 3 * (size_t)g->world));
        r.seqbuf_members = g->world;
    }
    const uint64_t hash[3] = {r.coll_hash[0], r.coll_hash[1], r.coll_hash[2]};       // before the check's own collectives
    const long cnt[3] = {r.coll_count[0], r.coll_count[1], r.coll_count[2]};
    for (int c = 0; c < 3; ++c) {
        std::vector<double> v((size_t)3 * sizes[c], 0.0);
        v[(size_t)3 * me[c]] = (double)(hash[c] >> 32);
        v[(size_t)3 * me[c] + 1] = (double)(hash[c] & 0xffffffffull);
        v[(size_t)3 * me[c] + 2] = (double)cnt[c];
        HIP_CHECK(hipMemcpyAsync(r.seqbuf, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice, r.st));
        NCCL_CHECK(g_rccl.AllReduce(r.seqbuf, r.seqbuf, v.size(), ncclFloat64, ncclSum, comms[c], r.st));
        HIP_CHECK(hipMemcpyAsync(v.data(), r.seqbuf, sizeof(double) * v.size(), hipMemcpyDeviceToHost, r.st));
        HIP_CHECK(hipStreamSynchronize(r.st));
        for (int m = 0; m < sizes[c]; ++m)
            for (int q = 0; q < 3; ++q)
                if (v[(size_t)3 * m + q] != v[(size_t)3 * me[c] + q]) {
                    mi355gp_set_error("grid mode: rank %d and member %d of communicator %d logged different collective sequences "
                                      "(%.0f vs %ld collectives)", r.rank, m, c, v[(size_t)3 * m + 2], cnt[c]);
                    return -7;
                }
    }
    return 0;
}

extern "C" {

int mi355gp_grid_unique_id(void* id128) {
    ARGCHK(id128 != nullptr, "mi355gp_grid_unique_id: NULL");
    if (!g_rccl.load()) return -20;
    ncclUniqueId id;
    NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int mi355gp_grid_create(int device, int rank, int world, int Pr, int Pc, int nb, const void* id128,
                        mi355gp_grid** out) {
    ARGCHK(out && Pr >= 1 && Pc >= 1 && nb >= NB && nb % NB == 0, "mi355gp_grid_create: Pr, Pc >= 1, nb % 128 == 0");
    ARGCHK(world == Pr * Pc, "mi355gp_grid_create: world must equal Pr*Pc");
    ARGCHK(rank >= 0 && rank < world, "mi355gp_grid_create: bad rank");
    int ndev = 0;
    mi355gp_device_count(&ndev);
    if (device < 0 || device >= ndev) {
        mi355gp_set_error("mi355gp_grid_create: device %d not available (%d HIP devices visible)", device, ndev);
        return -2;
    }
    HIP_CHECK(hipSetDevice(device));
    mi355gp_grid* g = new mi355gp_grid();
    g->device = device;
    g->world = world;
    g->Pr = Pr;
    g->Pc = Pc;
    g->nb = nb;
    g->my_rank = rank;
    g->loopback = (id128 == nullptr);
    {
        const char* envf = DIAG_ENV("GRID_FORCE_GENERIC");
        if (g->loopback && world == 1 && !(envf && atoi(envf))) {
            if (int rc = mi355gp_create(device, &g->single)) return rc;
            *out = g;
            return 0;
        }
    }
    HIP_CHECK(hipStreamCreateWithFlags(&g->st, hipStreamNonBlocking));
    {
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&g->sc, hipStreamNonBlocking, greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&g->sw, hipStreamNonBlocking, least));
        HIP_CHECK(hipEventCreateWithFlags(&g->ev_w, hipEventDisableTiming));
        for (int o = 0; o < MI355GP_GRID_OPT_NUM; ++o) grid_default_option(g, o);
        g->dbg_crit = diag_env_int(DIAG_ENV("GRID_DBG_CRIT"), 0);
    }
    for (auto& e : g->ev) HIP_CHECK(hipEventCreate(&e));
    if (g->loopback) {
        g->ranks.resize((size_t)world);
        for (int r = 0; r < world; ++r) {
            g->ranks[r].rank = r;
            g->ranks[r].pr = r / Pc;
            g->ranks[r].pc = r % Pc;
            g->ranks[r].st = g->st;
            g->ranks[r].sc = g->sc;
            g->ranks[r].sw = g->sw;
        }
    } else {
        if (!g_rccl.load()) return -20;
        g->ranks.resize(1);
        GridRank& r = g->ranks[0];
        r.rank = rank;
        r.pr = rank / Pc;
        r.pc = rank % Pc;
        r.st = g->st;
        r.sc = g->sc;
        r.sw = g->sw;
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        NCCL_CHECK(g_rccl.CommInitRank(&g->comm_world, world, id, rank));
        NCCL_CHECK(g_rccl.CommSplit(g->comm_world, r.pr, r.pc, &g->comm_row, nullptr));   // rank inside = pc
        NCCL_CHECK(g_rccl.CommSplit(g->comm_world, r.pc, r.pr, &g->comm_col, nullptr));   // rank inside = pr
    }
    *out = g;
    return 0;
}

int mi355gp_grid_destroy(mi355gp_grid* g) {
    if (!g) return 0;
    if (g->single) {
        mi355gp_destroy(g->single);
        delete g;
        return 0;
    }
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->st);
    for (GridRank& r : g->ranks) free_rank(r);
    if (g->comm_row) g_rccl.CommDestroy(g->comm_row);
    if (g->comm_col) g_rccl.CommDestroy(g->comm_col);
    if (g->comm_world) g_rccl.CommDestroy(g->comm_world);
    for (auto& e : g->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : g->ev_cr) (void)hipEventDestroy(e);
    for (auto& e : g->ev_p1) (void)hipEventDestroy(e);
    if (g->sc) {
        (void)hipStreamSynchronize(g->sc);
        (void)hipStreamDestroy(g->sc);
    }
    if (g->sw) {
        (void)hipStreamSynchronize(g->sw);
        (void)hipStreamDestroy(g->sw);
    }
    if (g->ev_w) (void)hipEventDestroy(g->ev_w);
    if (g->st) (void)hipStreamDestroy(g->st);
    delete g;
    return 0;
}

// Every rank passes the full (replicated) X and R: N*D*8 bytes is small next to the N^2/P matrix share.
int mi355gp_grid_set_data(mi355gp_grid* g, const double* X, int64_t N, int D, const double* R, int Dy) {
    ARGCHK(g && X && R && N > 0 && D > 0 && Dy > 0, "mi355gp_grid_set_data: bad arguments");
    if (g->single) {
        g->n = N;
        return mi355gp_set_data(g->single, X, N, D, R, Dy);
    }
    HIP_CHECK(hipSetDevice(g->device));
    HIP_CHECK(hipStreamSynchronize(g->st));
    HIP_CHECK(hipStreamSynchronize(g->sc));
    HIP_CHECK(hipStreamSynchronize(g->sw));
    const long nb = g->nb;
    g->n = N;
    g->T = (N + nb - 1) / nb;
    while ((long)g->ev_cr.size() < g->T + 1) {
        hipEvent_t e1, e2;
        HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        g->ev_cr.push_back(e1);
        g->ev_p1.push_back(e2);
    }
    g->npad = g->T * nb;
    g->D = D;
    g->Dy = Dy;
    g->have_result = false;
    const long T = g->T, TLrM = (T + g->Pr - 1) / g->Pr, TLcM = (T + g->Pc - 1) / g->Pc;
    const int groups = (D + 31) / 32;
    for (GridRank& r : g->ranks) {
        const int rank = r.rank, pr = r.pr, pc = r.pc;
        hipStream_t st = r.st, sc = r.sc, sw = r.sw;
        free_rank(r);
        r.rank = rank; r.pr = pr; r.pc = pc; r.st = st; r.sc = sc; r.sw = sw;
        r.TLr = cnt_le(T - 1, pr, g->Pr);
        r.TLc = cnt_le(T - 1, pc, g->Pc);
        r.LR = TLrM * nb;
        r.LC = TLcM * nb;
        const size_t mat = sizeof(double) * r.LR * r.LC;
        HIP_CHECK(hipMalloc(&r.A, mat));
        HIP_CHECK(hipMalloc(&r.X, mat));
        HIP_CHECK(hipMalloc(&r.W, mat));
        if (int rc = alloc_panel_stores(g, r)) return rc;
        HIP_CHECK(hipMalloc(&r.cpart, sizeof(double) * (size_t)(r.LR / CR_ROWS + 1) * r.LC));
        HIP_CHECK(hipMalloc(&r.Dt, sizeof(double) * nb * nb));
        HIP_CHECK(hipMalloc(&r.Dv, sizeof(double) * nb * nb));
        HIP_CHECK(hipMalloc(&r.Ds, sizeof(double) * nb * nb));
        if (!g->loopback) HIP_CHECK(hipMalloc(&r.bstage, sizeof(double) * nb * nb * (size_t)(TLrM > TLcM ? TLrM : TLcM)));
        // local point sets (host-side gather, uploaded once)
        std::vector<long> gR((size_t)r.LR), gC((size_t)r.LC);
        std::vector<double> XsR((size_t)r.LR * D, 0.0), XsC((size_t)r.LC * D, 0.0);
        r.nvr = r.nvc = 0;
        for (long l = 0; l < r.LR; ++l) {
            const long lt = l / nb;
            const long gi = (lt < r.TLr) ? (lt * g->Pr + pr) * nb + l % nb : g->npad + l;   // unused rows: beyond n
            gR[l] = gi;
            if (gi < N) { memcpy(&XsR[(size_t)l * D], X + gi * D, sizeof(double) * D); r.nvr = l + 1; }
        }
        for (long l = 0; l < r.LC; ++l) {
            const long lt = l / nb;
            const long gj = (lt < r.TLc) ? (lt * g->Pc + pc) * nb + l % nb : g->npad + l;
            gC[l] = gj;
            if (gj < N) { memcpy(&XsC[(size_t)l * D], X + gj * D, sizeof(double) * D); r.nvc = l + 1; }
        }
        HIP_CHECK(hipMalloc(&r.gR, sizeof(long) * r.LR));
        HIP_CHECK(hipMalloc(&r.gC, sizeof(long) * r.LC));
        HIP_CHECK(hipMalloc(&r.XsR, sizeof(double) * r.LR * D));
        HIP_CHECK(hipMalloc(&r.XsC, sizeof(double) * r.LC * D));
        HIP_CHECK(hipMalloc(&r.XtR, sizeof(double) * r.LR * D));
        HIP_CHECK(hipMalloc(&r.XtC, sizeof(double) * r.LC * D));
        HIP_CHECK(hipMemcpy(r.gR, gR.data(), sizeof(long) * r.LR, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(r.gC, gC.data(), sizeof(long) * r.LC, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(r.XsR, XsR.data(), sizeof(double) * r.LR * D, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(r.XsC, XsC.data(), sizeof(double) * r.LC * D, hipMemcpyHostToDevice));
        // diagonal tiles owned by this rank
        std::vector<int> dlr, dlc, dg;
        for (long t = 0; t < T; ++t)
            if (t % g->Pr == pr && t % g->Pc == pc) {
                dlr.push_back((int)(t / g->Pr));
                dlc.push_back((int)(t / g->Pc));
                dg.push_back((int)t);
            }
        r.ndiag = (int)dg.size();
        HIP_CHECK(hipMalloc(&r.dl_r, sizeof(int) * (r.ndiag + 1)));
        HIP_CHECK(hipMalloc(&r.dl_c, sizeof(int) * (r.ndiag + 1)));
        HIP_CHECK(hipMalloc(&r.dg, sizeof(int) * (r.ndiag + 1)));
        if (r.ndiag) {
            HIP_CHECK(hipMemcpy(r.dl_r, dlr.data(), sizeof(int) * r.ndiag, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(r.dl_c, dlc.data(), sizeof(int) * r.ndiag, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(r.dg, dg.data(), sizeof(int) * r.ndiag, hipMemcpyHostToDevice));
        }
        const long vmax = (r.LR > r.LC ? r.LR : r.LC);
        HIP_CHECK(hipMalloc(&r.vloc, sizeof(double) * vmax));
        HIP_CHECK(hipMalloc(&r.gvec, sizeof(double) * N * Dy));
        HIP_CHECK(hipMalloc(&r.gvec2, sizeof(double) * N));
        HIP_CHECK(hipMalloc(&r.alpha, sizeof(double) * N * Dy));
        HIP_CHECK(hipMalloc(&r.ybuf, sizeof(double) * N * Dy));
        HIP_CHECK(hipMalloc(&r.Rg, sizeof(double) * N * Dy));
        HIP_CHECK(hipMemcpy(r.Rg, R, sizeof(double) * N * Dy, hipMemcpyHostToDevice));
        HIP_CHECK(hipMalloc(&r.scal, sizeof(double) * 8));
        HIP_CHECK(hipMalloc(&r.gradPart, sizeof(double) * groups * 2048 * GP_STRIDE));
        HIP_CHECK(hipMalloc(&r.gradOut, sizeof(double) * groups * GP_STRIDE));
        HIP_CHECK(hipMalloc(&r.invls, sizeof(double) * D));
        HIP_CHECK(hipMalloc(&r.noise, sizeof(double) * N));
        HIP_CHECK(hipMalloc(&r.info_g, sizeof(int) * 4));
        if (factor_ws_alloc(&r.ws, nb) != 0) return -3;
        r.ws.lookahead = 0;      // the nb x nb diagonal tile is factored in order on the rank's stream
    }
    return 0;
}

}  // extern "C"

// one evaluation on all local ranks
static int grid_run(mi355gp_grid* g, KernParams kp, const double* theta, const std::vector<double>& inv_ls,
                    const double* noise, int64_t noise_len, double jit, double* out_scalars, double* alpha_out,
                    double* dtheta_out, double* diag_out, double* stage_ms, int attempt = 0) {
    const long nb = g->nb, T = g->T, n = g->n;
    const int Pr = g->Pr, Pc = g->Pc, Dy = g->Dy, D = g->D, q = (int)(nb / NB);
    const size_t tile = (size_t)nb * nb;
    const GridPred nopred{0, 1, 0, 1, 0, 0, 0};
    {   // the L-panel ring is sized for the group size in effect when the data was set: follow a later change of G
        long want = 2 * (g->G < 1 ? 1 : g->G);
        if (want > T) want = T;
        for (GridRank& r : g->ranks)
            if (r.ring_slots != want) {
                HIP_CHECK(hipDeviceSynchronize());
                if (int rc = alloc_panel_stores(g, r)) return rc;
            }
    }
    for (GridRank& r : g->ranks)
        for (int c = 0; c < 3; ++c) {
            r.coll_hash[c] = 14695981039346656037ull;
            r.coll_count[c] = 0;
        }
    HIP_CHECK(hipEventRecord(g->ev[0], g->st));
    // ---- covariance tiles: K(X_rows, X_cols) + diagonal fix-up -------------------------------------------
    for (GridRank& r : g->ranks) {
        hipStream_t st = r.st;
        HIP_CHECK(hipMemcpyAsync(r.invls, inv_ls.data(), sizeof(double) * D, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(r.noise, noise, sizeof(double) * noise_len, hipMemcpyHostToDevice, st));
        launch_scale_inputs(st, r.XsR, r.LR, D, r.invls, kp.ard, r.XtR, r.LR);
        launch_scale_inputs(st, r.XsC, r.LC, D, r.invls, kp.ard, r.XtC, r.LC);
        HIP_CHECK(hipMemsetAsync(r.A, 0, sizeof(double) * r.LR * r.LC, st));
        HIP_CHECK(hipMemsetAsync(r.X, 0, sizeof(double) * r.LR * r.LC, st));
        HIP_CHECK(hipMemsetAsync(r.W, 0, sizeof(double) * r.LR * r.LC, st));
        HIP_CHECK(hipMemsetAsync(r.scal, 0, sizeof(double) * 8, st));
        HIP_CHECK(hipMemsetAsync(r.info_g, 0, sizeof(int) * 4, st));
        if (r.nvr > 0 && r.nvc > 0) launch_kbuild_cross(st, kp, r.XtR, r.LR, r.nvr, r.XtC, r.LC, r.nvc, r.A, r.LC);
        if (r.ndiag > 0) {
            const long cnt = (long)r.ndiag * nb;
            hipLaunchKernelGGL(k_grid_fix_diag, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, r.A, r.LC, nb, n,
                               r.ndiag, r.dl_r, r.dl_c, r.dg, r.noise, (long)noise_len, jit);
        }
    }
    HIP_CHECK(hipEventRecord(g->ev[1], g->st));
    // ---- the one-pass factorisation / inversion, two-level blocked -----------------------------------------------
    // Steps are taken in GROUPS of G (MI355GP_GRID_G).  Every step k runs its critical path crit(k) -- diagonal tile k,
    // D = L_kk^-1, the panel solves, row k of X and ALL panel broadcasts -- and then brings only the REST OF ITS GROUP up to
    // date with K = nb updates (near(k): tile columns k+1 .. ge-1 of A, tile rows k+1 .. ge-1 of B; ge = first step of the
    // next group).  Everything beyond the group receives the group's G panels in ONE pass over C with K = G * nb
    // (k_grid_gemm_multi): first the next group's columns / rows (part1(g), on the critical path), then the bulk.  W = X^T X is
    // not read by anything before the end: its updates are aggregated over GW steps (MI355GP_GRID_GW; 0 = one deep-K pass
    // after the last step, the distributed lauum).  The three read-modify-write passes per step of the one-level form
    // become one pass per G (A, B) and per GW (W) steps: the C traffic of the updates drops accordingly and the tile
    // pipeline runs K = G * nb deep (k_grid_gemm at K = 512: 0.61 of the fp64 peak; lauum-deep: 0.87).
    // Two streams per rank, one group of look-ahead:
    //   sc (high priority): crit(k), near(k) for the steps of group g, then part1(g) [after bulk(g-1): both write the
    //                       columns / rows of group g+1], then group g+1
    //   st                : bulk(g) [after crit of the group's last step]
    //   sw (low priority) : the W updates [after crit of the last step they read]; st joins it after the loop
    // bulk(g) touches columns / rows >= the start of group g+2 only, the critical path of group g+1 stays inside group
    // g+1's columns / rows: they run concurrently.  The X panels keep one buffer per step for the whole evaluation (the
    // deferred W = X^T X reads them late); the L panels live in a RING of 2 G full-height slots (alloc_panel_stores): slot
    // reuse is safe because sc waits for ev_p1[gi-1] -- i.e. for bulk(g-1) on st -- before part1(g), so every reader of group
    // g's panels has been enqueued behind the event that group g+2's crit (the next writer of those slots) waits for.
    // Every rank enqueues the same sequence of collectives on sc, so their order is consistent across the grid.
    const bool la = g->lookahead != 0;
    hipStream_t scs = la ? g->sc : g->st;
    const long G = g->G < 1 ? 1 : g->G, GW = g->GW;
    if (la) {
        HIP_CHECK(hipEventRecord(g->ev[5], g->st));          // the covariance tiles precede everything on sc / sw
        HIP_CHECK(hipStreamWaitEvent(g->sc, g->ev[5], 0));
        HIP_CHECK(hipStreamWaitEvent(g->sw, g->ev[5], 0));
    }
    auto crit = [&](long k) -> int {
        const int opr = (int)(k % Pr), opc = (int)(k % Pc);
        const long lkr = k / Pr, lkc = k / Pc;
        // (a) diagonal tile: L_kk and D = L_kk^-1 on its owner
        for (GridRank& r : g->ranks) {
            if (r.pr != opr || r.pc != opc) continue;
            hipStream_t s = la ? r.sc : r.st;
            double* At = r.A + lkr * nb * r.LC + lkc * nb;
            HIP_CHECK(hipMemcpy2DAsync(r.Dt, sizeof(double) * nb, At, sizeof(double) * r.LC, sizeof(double) * nb, nb,
                                       hipMemcpyDeviceToDevice, s));
            potrf_device(s, r.Dt, nb, &r.ws);
            hipLaunchKernelGGL(k_grid_tile_stats, dim3(1), dim3(64), 0, s, r.ws.logsum, q, r.ws.info, k * nb, r.scal,
                               r.info_g);
            HIP_CHECK(hipMemsetAsync(r.Dv, 0, sizeof(double) * tile, s));
            trtri_device(s, r.Dt, r.Dv, r.Ds, nb, &r.ws);
            HIP_CHECK(hipMemcpy2DAsync(At, sizeof(double) * r.LC, r.Dt, sizeof(double) * nb, sizeof(double) * nb, nb,
                                       hipMemcpyDeviceToDevice, s));
        }
        if (g->dbg_crit >= 3) return 0;
        const bool nobc = g->dbg_crit >= 2;
        // (b) D to the panel owners (process column opc) and to the owners of row k of X (process row opr)
        if (!nobc) {
            if (int rc = grid_bcast(g, GROUP_COL, opc, opr, tile, [](GridRank& r, bool) { return r.Dv; })) return rc;
            if (int rc = grid_bcast(g, GROUP_ROW, opr, opc, tile, [](GridRank& r, bool) { return r.Dv; })) return rc;
        }
        // (c) panel solve L_ik = A_ik D^T on process column opc
        for (GridRank& r : g->ranks) {
            if (r.pc != opc) continue;
            hipStream_t s = la ? r.sc : r.st;
            const int lr0 = cnt_le(k, r.pr, Pr);
            const long rows = (long)(r.TLr - lr0) * nb;
            if (rows <= 0) continue;
            double* Acol = r.A + (long)lr0 * nb * r.LC + lkc * nb;
            grid_gemm_t<true, true>(s, r.hRP[k] + (long)lr0 * tile, nb, 0, GOp{Acol, r.LC, 0}, GOp{r.Dv, nb, 0}, rows, nb,
                                    nb, nb, 0, nopred);
            HIP_CHECK(hipMemcpy2DAsync(Acol, sizeof(double) * r.LC, r.hRP[k] + (long)lr0 * tile, sizeof(double) * nb,
                                       sizeof(double) * nb, rows, hipMemcpyDeviceToDevice, s));
        }
        // (d) row panel along every process row
        for (int pr = 0; pr < Pr && !nobc; ++pr) {
            const int lr0 = cnt_le(k, pr, Pr), TLr = cnt_le(T - 1, pr, Pr);
            const size_t cnt = (size_t)(TLr - lr0) * tile;
            if (int rc = grid_bcast(g, GROUP_ROW, pr, opc, cnt,
                                    [&](GridRank& r, bool) { return r.hRP[k] + (long)lr0 * tile; }))
                return rc;
        }
        // (e) column panel: L_jk for the local columns j > k comes from process row j % Pr -- one broadcast per (process column,
        //     root) pair: at most Pr per column (grid_bcast_tile_runs)
        if (!nobc) {
            std::vector<TileRun> runs;
            for (int pc = 0; pc < Pc; ++pc)
                for (int root = 0; root < Pr; ++root) {
                    long first, step;
                    int cnt;
                    tile_progression(k + 1, T, pc, Pc, root, Pr, &first, &step, &cnt);
                    if (cnt > 0) runs.push_back(TileRun{GROUP_COL, pc, root, cnt, first / Pr, step / Pr, first / Pc, step / Pc});
                }
            if (int rc = grid_bcast_tile_runs(g, runs, tile, [&](GridRank& r) { return r.hRP[k]; }, [&](GridRank& r) { return r.hCP[k]; }))
                return rc;
        }
        // (g) row k of X on process row opr: X_kj = D * B_kj (j < k), X_kk = D
        for (GridRank& r : g->ranks) {
            if (r.pr != opr) continue;
            hipStream_t s = la ? r.sc : r.st;
            const int lcB = cnt_lt(k, r.pc, Pc);                      // local columns with J < k
            const double* Brow = r.X + lkr * nb * r.LC;
            grid_gemm_t<true, false>(s, r.hXR[k], nb, 1, GOp{r.Dv, nb, 0}, GOp{Brow, r.LC, 0}, nb, (long)lcB * nb, nb, nb,
                                     0, nopred);
            if (r.pc == opc) HIP_CHECK(hipMemcpyAsync(r.hXR[k] + (long)lcB * tile, r.Dv, sizeof(double) * tile,
                                                      hipMemcpyDeviceToDevice, s));
            const int lc0 = cnt_le(k, r.pc, Pc);
            if (lc0 > 0)                                                // final X row block back into the local matrix
                hipLaunchKernelGGL(k_tiles_to_rowblock, dim3(64, (unsigned)lc0), dim3(256), 0, s, r.X + lkr * nb * r.LC, (long)r.LC,
                                   (const double*)r.hXR[k], (int)nb);
        }
        // (h) X row panel down every process column
        for (int pc = 0; pc < Pc && !nobc; ++pc) {
            const size_t cnt = (size_t)cnt_le(k, pc, Pc) * tile;
            if (int rc = grid_bcast(g, GROUP_COL, pc, opr, cnt, [&](GridRank& r, bool) { return r.hXR[k]; })) return rc;
        }
        // (i) X_ki for the local rows i <= k comes from process column i % Pc -- at most Pc broadcasts per process row
        if (!nobc) {
            std::vector<TileRun> runs;
            for (int pr = 0; pr < Pr; ++pr)
                for (int root = 0; root < Pc; ++root) {
                    long first, step;
                    int cnt;
                    tile_progression(0, k + 1, pr, Pr, root, Pc, &first, &step, &cnt);
                    if (cnt > 0) runs.push_back(TileRun{GROUP_ROW, pr, root, cnt, first / Pc, step / Pc, first / Pr, step / Pr});
                }
            if (int rc = grid_bcast_tile_runs(g, runs, tile, [&](GridRank& r) { return r.hXR[k]; }, [&](GridRank& r) { return r.hXRr[k]; }))
                return rc;
        }
        return 0;
    };
    // updates of A and B by the panels [k0, k1) on local tile columns J in [ca, cb) (A, rows I >= ca) and tile rows I in
    // [ca, cb) (B, columns J < k1; panel k reaches the columns J <= k)
    auto update_AB = [&](hipStream_t (*pick)(GridRank&, bool), long k0, long k1, long ca, long cb) {
        if (g->dbg_crit) return;
        for (GridRank& r : g->ranks) {
            hipStream_t s = pick(r, la);
            const GridPred lower{1, Pr, r.pr, Pc, r.pc, 0, 0}, all{0, Pr, r.pr, Pc, r.pc, 0, 0};
            grid_gemm_multi<true, true, 2>(s, r.A, r.LC, r.dRP, r.dCP, (int)k0, (int)k1, cnt_lt(ca, r.pr, Pr), r.TLr,
                                           cnt_lt(ca, r.pc, Pc), cnt_lt(cb, r.pc, Pc), nb, 0, lower);
            grid_gemm_multi<true, false, 2>(s, r.X, r.LC, r.dRP, r.dXR, (int)k0, (int)k1, cnt_lt(ca, r.pr, Pr),
                                            cnt_lt(cb, r.pr, Pr), 0, cnt_lt(k1, r.pc, Pc), nb, 1, all);
        }
    };
    auto on_sc = [](GridRank& r, bool la_) -> hipStream_t { return la_ ? r.sc : r.st; };
    auto on_st = [](GridRank& r, bool) -> hipStream_t { return r.st; };
    long kw0 = 0;                                            // first step whose X panels W has not received yet
    const long ngroups = (T + G - 1) / G;
    for (long gi = 0; gi < ngroups; ++gi) {
        const long kb = gi * G, ge = (kb + G < T) ? kb + G : T, ge2 = (ge + G < T) ? ge + G : T;
        for (long k = kb; k < ge; ++k) {
            if (int rc = crit(k)) return rc;
            if (k + 1 < ge) update_AB(on_sc, k, k + 1, k + 1, ge);                 // near(k)
        }
        if (la) HIP_CHECK(hipEventRecord(g->ev_cr[gi], scs));
        if (ge < T) {                                                              // part1(g): the next group's columns / rows
            if (la && gi > 0) HIP_CHECK(hipStreamWaitEvent(g->sc, g->ev_p1[gi - 1], 0));
            update_AB(on_sc, kb, ge, ge, ge2);
        }
        if (la) HIP_CHECK(hipStreamWaitEvent(g->st, g->ev_cr[gi], 0));
        if (ge2 < T) update_AB(on_st, kb, ge, ge2, T);                             // bulk(g)
        if (la) HIP_CHECK(hipEventRecord(g->ev_p1[gi], g->st));
        // W_ij += sum_k X_ki^T X_kj over the finished steps, k >= i >= j
        const bool flush = !g->dbg_crit && ((ge == T) || (GW > 0 && ge - kw0 >= GW));
        if (flush) {                                          // on the low-priority stream: fills whatever the other two leave idle
            if (la) HIP_CHECK(hipStreamWaitEvent(g->sw, g->ev_cr[gi], 0));
            for (GridRank& r : g->ranks) {
                const GridPred lower{1, Pr, r.pr, Pc, r.pc, 0, 0};
                grid_gemm_multi<false, false, 1>(la ? r.sw : r.st, r.W, r.LC, r.dXRr, r.dXR, (int)kw0, (int)ge, 0,
                                                 cnt_lt(ge, r.pr, Pr), 0, cnt_lt(ge, r.pc, Pc), nb, 2, lower);
            }
            kw0 = ge;
        }
    }
    if (la) {
        HIP_CHECK(hipEventRecord(g->ev_w, g->sw));
        HIP_CHECK(hipStreamWaitEvent(g->st, g->ev_w, 0));
    }
    HIP_CHECK(hipEventRecord(g->ev[2], g->st));
    // ---- alpha = X^T (X R), diag W, logdet ------------------------------------------------------------------
    const unsigned nblkN = (unsigned)((n * Dy + 255) / 256);
    for (GridRank& r : g->ranks) {                       // y = X R (partial over the local columns)
        HIP_CHECK(hipMemsetAsync(r.ybuf, 0, sizeof(double) * n * Dy, r.st));
        for (int d = 0; d < Dy; ++d) {
            hipLaunchKernelGGL(k_grid_row_reduce, dim3((unsigned)((r.LR + 3) / 4)), dim3(256), 0, r.st, r.X, r.LC, r.LR,
                               r.LC, r.gC, r.Rg, Dy, d, n, r.vloc);
            hipLaunchKernelGGL(k_grid_scatter, dim3((unsigned)((r.LR + 255) / 256)), dim3(256), 0, r.st, r.vloc, r.LR,
                               r.gR, n, Dy, d, r.ybuf);
        }
    }
    // NB: a rank's partial y covers only its local columns, and ranks of one process row write the same global
    // rows: the scatter above must not overwrite -- every rank owns a private ybuf, and the all-reduce sums them.
    if (int rc = grid_allreduce(g, (size_t)n * Dy, [](GridRank& r) { return r.ybuf; })) return rc;
    auto col_reduce = [&](GridRank& r, const double* v, int d) {    // vloc[j] = sum_i X_ij * (v ? v[g(i)][d] : X_ij)
        const long nch = (r.LR + CR_ROWS - 1) / CR_ROWS;
        hipLaunchKernelGGL(k_grid_col_reduce_part, dim3((unsigned)((r.LC + 63) / 64), (unsigned)nch), dim3(256), 0, r.st, r.X,
                           r.LC, r.LR, r.LC, r.gR, v, v ? Dy : 1, d, n, r.cpart, nb, Pr, r.pr, Pc, r.pc);
        hipLaunchKernelGGL(k_grid_col_combine, dim3((unsigned)((r.LC + 255) / 256)), dim3(256), 0, r.st, r.cpart, nch, r.LC,
                           r.vloc);
    };
    for (GridRank& r : g->ranks) {                       // alpha = X^T y (partial over the local rows), diag W
        HIP_CHECK(hipMemsetAsync(r.alpha, 0, sizeof(double) * n * Dy, r.st));
        HIP_CHECK(hipMemsetAsync(r.gvec2, 0, sizeof(double) * n, r.st));
        for (int d = 0; d < Dy; ++d) {
            col_reduce(r, r.ybuf, d);
            hipLaunchKernelGGL(k_grid_scatter, dim3((unsigned)((r.LC + 255) / 256)), dim3(256), 0, r.st, r.vloc, r.LC,
                               r.gC, n, Dy, d, r.alpha);
        }
        col_reduce(r, nullptr, 0);
        hipLaunchKernelGGL(k_grid_scatter, dim3((unsigned)((r.LC + 255) / 256)), dim3(256), 0, r.st, r.vloc, r.LC, r.gC,
                           n, 1, 0, r.gvec2);
    }
    if (int rc = grid_allreduce(g, (size_t)n * Dy, [](GridRank& r) { return r.alpha; })) return rc;
    if (int rc = grid_allreduce(g, (size_t)n, [](GridRank& r) { return r.gvec2; })) return rc;
    if (int rc = grid_allreduce(g, 8, [](GridRank& r) { return r.scal; })) return rc;
    (void)nblkN;
    HIP_CHECK(hipEventRecord(g->ev[3], g->st));
    // ---- gradient reduction on the local tiles ----------------------------------------------------------------
    const int groups = (D + 31) / 32;
    std::vector<int> nblocks(g->ranks.size(), 0);
    for (size_t ri = 0; ri < g->ranks.size(); ++ri) {
        GridRank& r = g->ranks[ri];
        HIP_CHECK(hipMemsetAsync(r.gradOut, 0, sizeof(double) * groups * GP_STRIDE, r.st));
        if (r.nvr <= 0 || r.nvc <= 0) continue;
        // local dL_dK tiles into A (the factor is no longer needed there?  no: keep L for fetch) -> use CP/RP? too small:
        // W is consumed in place: G overwrites W after diag W has been taken (fetch of Kinv re-derives from X if needed).
        hipLaunchKernelGGL(k_grid_dldk, dim3((unsigned)((r.nvc + 255) / 256), (unsigned)r.nvr), dim3(256), 0, r.st, r.W,
                           r.W, r.LC, r.nvr, r.nvc, r.gR, r.gC, r.alpha, Dy, n);
        const int nbk = grad_generic_num_blocks(r.nvr, r.nvc);
        nblocks[ri] = nbk;
        launch_grad_generic(r.st, kp, r.XtR, r.LR, r.nvr, r.XtC, r.LC, r.nvc, 0, r.W, r.LC, r.gradPart, GP_STRIDE);
        for (int gi = 0; gi < (kp.ard ? groups : 1); ++gi)
            launch_reduce_partials(r.st, r.gradPart + (long)gi * nbk * GP_STRIDE, nbk, GP_STRIDE,
                                   r.gradOut + (long)gi * GP_STRIDE);
    }
    if (int rc = grid_allreduce(g, (size_t)groups * GP_STRIDE, [](GridRank& r) { return r.gradOut; })) return rc;
    HIP_CHECK(hipEventRecord(g->ev[4], g->st));
    // ---- results (replicated on every rank) -----------------------------------------------------------------------
    GridRank& r0 = g->ranks[0];
    std::vector<double> alpha((size_t)n * Dy), dW((size_t)n), R((size_t)n * Dy), sums((size_t)groups * GP_STRIDE);
    double scal[8];
    int info = 0;
    HIP_CHECK(hipMemcpyAsync(alpha.data(), r0.alpha, sizeof(double) * n * Dy, hipMemcpyDeviceToHost, g->st));
    HIP_CHECK(hipMemcpyAsync(dW.data(), r0.gvec2, sizeof(double) * n, hipMemcpyDeviceToHost, g->st));
    HIP_CHECK(hipMemcpyAsync(R.data(), r0.Rg, sizeof(double) * n * Dy, hipMemcpyDeviceToHost, g->st));
    HIP_CHECK(hipMemcpyAsync(sums.data(), r0.gradOut, sizeof(double) * groups * GP_STRIDE, hipMemcpyDeviceToHost, g->st));
    HIP_CHECK(hipMemcpyAsync(scal, r0.scal, sizeof(double) * 8, hipMemcpyDeviceToHost, g->st));
    HIP_CHECK(hipStreamSynchronize(g->st));
    // Every rank must take the same branch of the caller's jitter ladder: "some tile failed" travels in the all-reduced
    // scal[1]; the failing column itself is only known to the ranks hosting that tile (others report N).
    for (GridRank& r : g->ranks) {
        int ir = 0;
        HIP_CHECK(hipMemcpy(&ir, r.info_g, sizeof(int), hipMemcpyDeviceToHost));
        if (ir > 0 && (info == 0 || ir < info)) info = ir;
    }
    if (scal[1] > 0.0 && info == 0) info = (int)n;
    HIP_CHECK(hipGetLastError());
    if (g->check_seq)
        if (int rc = grid_check_sequences(g)) return rc;
    if (scal[2] > 0.0) {
        // A persistent factorisation of a diagonal tile was called off (co-residency gate) or aborted on some rank: the count
        // travels in the all-reduced scal[2], so EVERY rank takes this branch and the collectives of the redone evaluation
        // line up.  The tile factorisations stay on the launch-per-step schedule from here on.
        for (GridRank& r : g->ranks) {
            r.ws.persist_aborts += 1;
            r.ws.persist = 0;
        }
        if (attempt == 0)
            return grid_run(g, kp, theta, inv_ls, noise, noise_len, jit, out_scalars, alpha_out, dtheta_out, diag_out, stage_ms, 1);
        mi355gp_set_error("mi355gp_grid_exact_inference: a tile factorisation aborted twice");
        return -6;
    }
    if (stage_ms) {
        for (int i = 0; i < MI355GP_NUM_T; ++i) stage_ms[i] = 0.0;
        float ms;
        const int map[4] = {MI355GP_T_KBUILD, MI355GP_T_POTRF, MI355GP_T_SOLVE, MI355GP_T_GRAD};
        for (int i = 0; i < 4; ++i) {
            HIP_CHECK(hipEventElapsedTime(&ms, g->ev[i], g->ev[i + 1]));
            stage_ms[map[i]] = ms;
        }
        HIP_CHECK(hipEventElapsedTime(&ms, g->ev[0], g->ev[4]));
        stage_ms[MI355GP_T_TOTAL] = ms;
    }
    // a failed factorisation poisons logdet/alpha with NaN on every rank: detect it everywhere, not only on the owner
    double datafit = 0.0, alpha2 = 0.0, trw = 0.0;
    for (long i = 0; i < n * Dy; ++i) {
        datafit += alpha[i] * R[i];
        alpha2 += alpha[i] * alpha[i];
    }
    for (long i = 0; i < n; ++i) trw += dW[i];
    const double logdet = 2.0 * scal[0];
    if (info == 0 && !(std::isfinite(logdet) && std::isfinite(datafit))) info = (int)n;
    if (info > 0) {
        g->have_result = false;
        return info > n ? (int)n : info;
    }
    g->have_result = true;
    for (int i = 0; i < MI355GP_NUM_OUT; ++i) out_scalars[i] = 0.0;
    out_scalars[MI355GP_OUT_LML] = 0.5 * (-(double)n * Dy * LOG_2_PI - Dy * logdet - datafit);
    out_scalars[MI355GP_OUT_LOGDET] = logdet;
    out_scalars[MI355GP_OUT_DATAFIT] = datafit;
    out_scalars[MI355GP_OUT_DNOISE] = 0.5 * (alpha2 - Dy * trw);
    out_scalars[MI355GP_OUT_TRKINV] = trw;
    if (alpha_out) memcpy(alpha_out, alpha.data(), sizeof(double) * n * Dy);
    if (diag_out)
        for (long i = 0; i < n; ++i) {
            double a2 = 0.0;
            for (int d = 0; d < Dy; ++d) a2 += alpha[i * Dy + d] * alpha[i * Dy + d];
            diag_out[i] = 0.5 * (a2 - Dy * dW[i]);
        }
    if (dtheta_out) {
        dtheta_out[0] = sums[0] / kp.variance;
        if (!kp.ard) dtheta_out[1] = -sums[1] / theta[1];
        else
            for (int qd = 0; qd < D; ++qd) dtheta_out[1 + qd] = -sums[(qd / 32) * GP_STRIDE + 2 + (qd % 32)] / theta[1 + qd];
    }
    return 0;
}

extern "C" {

int mi355gp_grid_exact_inference(mi355gp_grid* g, int kind, int ard, const double* theta, const double* noise,
                                 int64_t noise_len, double jitter, double extra_jitter, double* out_scalars,
                                 double* alpha_out, double* dtheta_out, double* diag_dLdK_out, double* stage_ms) {
    ARGCHK(g && g->n > 0, "mi355gp_grid_exact_inference: set_data first");
    ARGCHK(out_scalars && theta && noise, "mi355gp_grid_exact_inference: NULL argument");
    if (g->single) {
        double ms[MI355GP_NUM_T];
        const int rc = mi355gp_exact_inference(g->single, kind, ard, theta, noise, noise_len, jitter, extra_jitter, out_scalars,
                                               alpha_out, dtheta_out, diag_dLdK_out, stage_ms ? ms : nullptr);
        if (stage_ms) {       // the grid's stage layout: the whole factorisation + inversion under POTRF
            for (int i = 0; i < MI355GP_NUM_T; ++i) stage_ms[i] = 0.0;
            stage_ms[MI355GP_T_KBUILD] = ms[MI355GP_T_KBUILD];
            stage_ms[MI355GP_T_POTRF] = ms[MI355GP_T_POTRF] + ms[MI355GP_T_TRTRI] + ms[MI355GP_T_LAUUM];
            stage_ms[MI355GP_T_SOLVE] = ms[MI355GP_T_SOLVE];
            stage_ms[MI355GP_T_GRAD] = ms[MI355GP_T_GRAD];
            stage_ms[MI355GP_T_TOTAL] = ms[MI355GP_T_TOTAL];
        }
        g->have_result = (rc == 0);
        return rc;
    }
    ARGCHK(kind >= 0 && kind <= 3, "unknown covariance kind");
    ARGCHK(noise_len == 1 || noise_len == g->n, "noise must have 1 or N entries");
    ARGCHK(theta[0] > 0.0, "variance must be positive");
    HIP_CHECK(hipSetDevice(g->device));
    std::vector<double> inv_ls((size_t)g->D, 0.0);
    const int nl = ard ? g->D : 1;
    for (int qd = 0; qd < nl; ++qd) {
        ARGCHK(theta[1 + qd] > 0.0, "lengthscales must be positive");
        inv_ls[qd] = 1.0 / theta[1 + qd];
    }
    KernParams kp{kind, ard ? 1 : 0, g->D, theta[0]};
    return grid_run(g, kp, theta, inv_ls, noise, noise_len, jitter + extra_jitter, out_scalars, alpha_out, dtheta_out,
                    diag_dLdK_out, stage_ms);
}

int mi355gp_grid_set_option(mi355gp_grid* g, int option, int value) {
    ARGCHK(g && option >= 0 && option < MI355GP_GRID_OPT_NUM, "mi355gp_grid_set_option: unknown option");
    ARGCHK(value >= -1, "mi355gp_grid_set_option: value must be >= 0 (or -1 for the default)");
    if (g->single) return 0;                              // the degenerate 1 x 1 grid runs the single-GPU pipeline
    if (value < 0) {
        grid_default_option(g, option);
        return 0;
    }
    if (option == MI355GP_GRID_OPT_LOOKAHEAD) g->lookahead = value ? 1 : 0;
    else if (option == MI355GP_GRID_OPT_G) {
        ARGCHK(value >= 1, "mi355gp_grid_set_option: G >= 1");
        g->G = value;
    } else if (option == MI355GP_GRID_OPT_CHECK_SEQ) g->check_seq = value ? 1 : 0;
    else g->GW = value;
    return 0;
}

int mi355gp_grid_get_option(mi355gp_grid* g, int option, int* value) {
    ARGCHK(g && value && option >= 0 && option < MI355GP_GRID_OPT_NUM, "mi355gp_grid_get_option: unknown option");
    *value = option == MI355GP_GRID_OPT_LOOKAHEAD ? g->lookahead : option == MI355GP_GRID_OPT_G ? g->G
             : option == MI355GP_GRID_OPT_CHECK_SEQ ? g->check_seq : g->GW;
    return 0;
}

// Collective log of logical rank `rank` of this process (loopback: any rank of the grid; one rank per process: its own, rank is
// ignored) for the LAST evaluation: out9 = [count world, row, column, then per communicator the hash as (high 32 bits, low 32 bits)].
int mi355gp_grid_coll_log(mi355gp_grid* g, int rank, double* out9) {
    ARGCHK(g && out9 && !g->single, "mi355gp_grid_coll_log: not available on the degenerate 1 x 1 grid");
    const GridRank* r = &g->ranks[0];
    if (g->loopback) {
        ARGCHK(rank >= 0 && rank < (int)g->ranks.size(), "mi355gp_grid_coll_log: bad rank");
        r = &g->ranks[(size_t)rank];
    }
    for (int c = 0; c < 3; ++c) {
        out9[c] = (double)r->coll_count[c];
        out9[3 + 2 * c] = (double)(r->coll_hash[c] >> 32);
        out9[4 + 2 * c] = (double)(r->coll_hash[c] & 0xffffffffull);
    }
    return 0;
}

// Host copy of the tiles owned by this process's ranks, placed at their global position in an N x N row-major array
// (entries owned by other processes are left untouched: callers zero `out` first and sum over ranks).
// which: MI355GP_FETCH_L (lower tiles of L), MI355GP_FETCH_KINV is not available after the gradient pass consumed W;
// 100 = X = L^-1 (lower).
int mi355gp_grid_fetch(mi355gp_grid* g, int which, double* out) {
    ARGCHK(g && out && g->n > 0 && g->have_result, "mi355gp_grid_fetch: run an inference call first");
    ARGCHK(which == MI355GP_FETCH_L || which == 100, "mi355gp_grid_fetch: L (0) or L^-1 (100)");
    if (g->single) return mi355gp_fetch(g->single, which, out, 0);
    HIP_CHECK(hipSetDevice(g->device));
    const long nb = g->nb, n = g->n;
    std::vector<double> tilebuf((size_t)nb * nb);
    for (GridRank& r : g->ranks) {
        const double* M = (which == MI355GP_FETCH_L) ? r.A : r.X;
        for (int li = 0; li < r.TLr; ++li)
            for (int lj = 0; lj < r.TLc; ++lj) {
                const long I = (long)li * g->Pr + r.pr, J = (long)lj * g->Pc + r.pc;
                if (J > I) continue;
                HIP_CHECK(hipMemcpy2D(tilebuf.data(), sizeof(double) * nb, M + (long)li * nb * r.LC + (long)lj * nb,
                                      sizeof(double) * r.LC, sizeof(double) * nb, nb, hipMemcpyDeviceToHost));
                for (long a = 0; a < nb; ++a) {
                    const long gi = I * nb + a;
                    if (gi >= n) break;
                    for (long b = 0; b < nb; ++b) {
                        const long gj = J * nb + b;
                        if (gj >= n || gj > gi) break;
                        out[gi * n + gj] = tilebuf[(size_t)a * nb + b];
                    }
                }
            }
    }
    return 0;
}

// Microbenchmark of the deep-K W = X^T X pass on ONE device (1 x 1 layout, T tiles of nb): out_ms[v] for
//   v = 0: k_lauum on the row-major matrix (the single-GPU path's kernel)
//   v = 1: k_grid_gemm_multi on the tile-major panel stores (two copies: XRr and XR), as the grid mode runs it
//   v = 2: the same with ONE panel store for both operands
//   v = 3: k_grid_gemm_multi reading the row-major matrix (panel k = rows k*nb ..)
int mi355gp_dbg_grid_multi(int device, int T, int nb, int reps, double* out_ms) {
    ARGCHK(T >= 1 && nb >= NB && nb % NB == 0 && reps >= 1 && out_ms, "mi355gp_dbg_grid_multi: bad arguments");
    HIP_CHECK(hipSetDevice(device));
    const long N = (long)T * nb;
    const size_t tile = (size_t)nb * nb;
    double *X = nullptr, *W = nullptr, *P1 = nullptr, *P2 = nullptr;
    const double **t1 = nullptr, **t2 = nullptr, **t3 = nullptr;
    HIP_CHECK(hipMalloc(&X, sizeof(double) * N * N));
    HIP_CHECK(hipMalloc(&W, sizeof(double) * N * N));
    const long ntl = (long)T * (T + 1) / 2;
    HIP_CHECK(hipMalloc(&P1, sizeof(double) * tile * ntl));
    HIP_CHECK(hipMalloc(&P2, sizeof(double) * tile * ntl));
    {
        std::vector<double> h((size_t)N * 64);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1024) - 0.5;
        for (long r = 0; r < N; r += 64) HIP_CHECK(hipMemcpy(X + r * N, h.data(), sizeof(double) * N * 64, hipMemcpyHostToDevice));
    }
    std::vector<const double*> h1((size_t)T), h2((size_t)T), h3((size_t)T);
    long off = 0;
    for (long k = 0; k < T; ++k) {
        h1[k] = P1 + off * tile;
        h2[k] = P2 + off * tile;
        h3[k] = X + k * nb * N;
        for (long j = 0; j <= k; ++j) {
            HIP_CHECK(hipMemcpy2D(P1 + (off + j) * tile, sizeof(double) * nb, X + k * nb * N + j * nb, sizeof(double) * N,
                                  sizeof(double) * nb, nb, hipMemcpyDeviceToDevice));
        }
        off += k + 1;
    }
    HIP_CHECK(hipMemcpy(P2, P1, sizeof(double) * tile * ntl, hipMemcpyDeviceToDevice));
    const size_t tb = sizeof(double*) * (size_t)T;
    HIP_CHECK(hipMalloc((void**)&t1, tb));
    HIP_CHECK(hipMalloc((void**)&t2, tb));
    HIP_CHECK(hipMalloc((void**)&t3, tb));
    HIP_CHECK(hipMemcpy((void*)t1, h1.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((void*)t2, h2.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((void*)t3, h3.data(), tb, hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    const GridPred lower{1, 1, 0, 1, 0, 0, 0};
    for (int v = 0; v < 5; ++v) {
        for (int rep = -1; rep < reps; ++rep) {
            if (rep == 0) HIP_CHECK(hipEventRecord(e0, st));
            if (v == 0) launch_lauum(st, X, W, N, (int)(N / NB));
            else if (v == 1) grid_gemm_multi<false, false, 0>(st, W, N, t1, t2, 0, T, 0, T, 0, T, nb, 2, lower);
            else if (v == 2) grid_gemm_multi<false, false, 0>(st, W, N, t1, t1, 0, T, 0, T, 0, T, nb, 2, lower);
            else if (v == 3) grid_gemm_multi<false, false, 0>(st, W, N, t3, t3, 0, T, 0, T, 0, T, nb, 2, lower, N, 0);
            else grid_gemm_multi<false, false, 0>(st, W, N, t3, t3, 0, T, 0, T, 0, T, nb, 2, lower, N, 2);
        }
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        out_ms[v] = ms / reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    for (void* p : {(void*)X, (void*)W, (void*)P1, (void*)P2, (void*)t1, (void*)t2, (void*)t3}) (void)hipFree(p);
    return 0;
}

}  // extern "C"
