// gemm.hip -- fp64 MFMA tile-GEMM kernels behind the blocked Cholesky (potrf), the batched triangular
// inverse (trtri) and X^T X (lauum).  One 128x128 output tile per workgroup; see gemm_tile.h.
#include "gemm_tile.h"
#include "internal.h"

// kernels here use 72 KiB of dynamic LDS: opt in once per kernel
#define LDS_OPT_IN(kernel)                                                                              \
    do {                                                                                                \
        static bool done = false;                                                                       \
        if (!done) {                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES);        \
            done = true;                                                                                \
        }                                                                                               \
    } while (0)

#include <algorithm>
#include <cstdlib>
#include <vector>
// tuning overrides of the A/B tools: diagnostics build only (common.h "environment switches")
#define env_int(name, dflt) diag_env_int(DIAG_ENV(name), dflt)
#define LB(NW) __launch_bounds__((NW) * 64, (NW) / 2)

// ------------------------------------------------------------------------------------------------
// Trailing update of the right-looking Cholesky (the dsyrk/dgemm inside LAPACK dpotrf, which GPy reaches
// through GPy/util/linalg.py:58):  C[ti,tj] -= A[ti,:] * B[tj,:]^T.
// `tri`: region is square on the diagonal -> enumerate the lower triangle only.
template <int NW, bool PRE>
__global__ LB(NW) void k_update_nt(double* __restrict__ C, long ldc,
                                                      const double* __restrict__ A, long lda,
                                                      const double* __restrict__ B, long ldb, int K, int ntc,
                                                      int row0t, int col0t, int tri, long ntiles) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    for (long bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
        int ti, tj;
        if (tri) {
            ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
            while ((long)ti * (ti + 1) / 2 > bid) --ti;
            while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
            tj = (int)(bid - (long)ti * (ti + 1) / 2);
        } else {
            ti = (int)(bid / ntc);
            tj = (int)(bid - (long)ti * ntc);
            if (col0t + tj > row0t + ti) continue;
        }
        d4 acc[4][GTCfg<NW>::NI];
        double* Ct = C + (long)ti * NB * ldc + (long)tj * NB;
        if (PRE) {
            gt_load_buf<NW>(Ct, ldc, acc);
            gemm_tile_128<true, true, NW, true>(A + (long)ti * NB * lda, lda, B + (long)tj * NB * ldb, ldb, K, acc, smem);
            gt_store<0, NW>(Ct, ldc, acc);
        } else {
            gt_zero<NW>(acc);
            gemm_tile_128<true, true, NW>(A + (long)ti * NB * lda, lda, B + (long)tj * NB * ldb, ldb, K, acc, smem);
            gt_store<2, NW>(Ct, ldc, acc);
        }
    }
}

// The same update cut into 64 x 64 quadrants (gemm_tile.h, "64 x 64 output tiles"): for launches with fewer 128-tiles than
// CUs -- the K = nbo "part 1" and the K = 128 in-panel updates on the panel chain.  Workgroup b: 128-tile b >> 2, quadrant
// b & 3.  Reads C first, accumulates -A*B on it, stores: the arithmetic of k_update_nt<4, true>, bit for bit.
__global__ __launch_bounds__(256) void k_update_nt64(double* __restrict__ C, long ldc, const double* __restrict__ A, long lda,
                                                     const double* __restrict__ B, long ldb, int K, int ntc, int row0t,
                                                     int col0t, int tri) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long bid = blockIdx.x >> 2;
    const int qi = (blockIdx.x >> 1) & 1, qj = blockIdx.x & 1;
    int ti, tj;
    if (tri) {
        ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((long)ti * (ti + 1) / 2 > bid) --ti;
        while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
        tj = (int)(bid - (long)ti * (ti + 1) / 2);
    } else {
        ti = (int)(bid / ntc);
        tj = (int)(bid - (long)ti * ntc);
        if (col0t + tj > row0t + ti) return;
    }
    d4 acc[2][2];
    double* Ct = C + ((long)ti * NB + qi * 64) * ldc + (long)tj * NB + qj * 64;
    gt64_load(Ct, ldc, acc);
    gemm_tile_64_v3<true, true, true>(A + ((long)ti * NB + qi * 64) * lda, lda, B + ((long)tj * NB + qj * 64) * ldb, ldb, K,
                                      acc, smem);
    gt64_store<0>(Ct, ldc, acc);
}

// does a launch of this shape take the 64 x 64 tile kernel?  (profile family of the caller)
bool update_nt_uses_64(int ntr, int ntc, int row0t, int col0t) {
    static const int upd64_max = env_int("UPD64_MAX", GEMM_DEFAULT_UPD64_MAX);
    const int tri = (row0t == col0t && ntr == ntc) ? 1 : 0;
    const long nblocks = tri ? (long)ntr * (ntr + 1) / 2 : (long)ntr * ntc;
    return nblocks <= upd64_max;
}

void launch_update_nt(hipStream_t st, double* C, long ldc, const double* A, long lda, const double* B, long ldb,
                      int K, int ntr, int ntc, int row0t, int col0t) {
    if (ntr <= 0 || ntc <= 0) return;
    const int tri = (row0t == col0t && ntr == ntc) ? 1 : 0;
    const long nblocks = tri ? (long)ntr * (ntr + 1) / 2 : (long)ntr * ntc;
    // Few tiles: the launch is the latency of ONE tile on ONE CU -> four times as many 64 x 64 tiles (bit-identical result)
    if (update_nt_uses_64(ntr, ntc, row0t, col0t)) {
        hipLaunchKernelGGL(k_update_nt64, dim3((unsigned)(4 * nblocks)), dim3(256), GT64_LDS_BYTES, st, C, ldc, A, lda, B, ldb,
                           K, ntc, row0t, col0t, tri);
        return;
    }
    // C is read through buffer loads BEFORE the k-loop, which then runs with negated A fragments, and stored plainly
    // (potrf 33.0 -> 32.3 ms against read-modify-write after the loop)
    LDS_OPT_IN((k_update_nt<4, true>));
    hipLaunchKernelGGL((k_update_nt<4, true>), dim3((unsigned)nblocks), dim3(256), GT_LDS_BYTES, st, C, ldc, A, lda, B, ldb, K,
                       ntc, row0t, col0t, tri, nblocks);
}

#ifdef MI355GP_DIAG   // diagnostics build only: WRONG RESULTS by construction, not in the product library
// ---- bounding experiment for a persistent trailing updater (tools/upd_queue_probe.py) ------------------------------
// ONE launch of 2 x CUs resident workgroups that drains the tile lists of ALL "part 2" updates of a factorisation from a single
// atomic queue, as if every dependence were already satisfied: the same tile code as k_update_nt<4, true> on the same operands, so
// the instruction stream and the memory traffic are those of the real updates, but no launch boundary, no ramp-up / drain per
// launch and no wait for the panel chain.  The RESULT IS WRONG BY CONSTRUCTION (tiles are updated before their panels are final):
// it measures what the launch boundaries cost in situ, i.e. an upper bound on what a dataflow updater with ready flags could gain.
__global__ LB(4) void k_update_nt_queue(double* __restrict__ A, long ld, const UpdTask* __restrict__ tasks, int ntasks,
                                        int* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ long s_tile;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_tile = (long)atomicAdd(counter, 1);
        __syncthreads();
        long bid = s_tile;
        int q = 0;
        while (q < ntasks && bid >= tasks[q].ntiles) bid -= tasks[q++].ntiles;
        if (q >= ntasks) return;
        const UpdTask tk = tasks[q];
        int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((long)ti * (ti + 1) / 2 > bid) --ti;
        while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
        const int tj = (int)(bid - (long)ti * (ti + 1) / 2);
        d4 acc[4][GTCfg<4>::NI];
        double* Ct = A + tk.c_off + (long)ti * NB * ld + (long)tj * NB;
        const double* P = A + tk.p_off;
        gt_load_buf<4>(Ct, ld, acc);
        gemm_tile_128<true, true, 4, true>(P + (long)ti * NB * ld, ld, P + (long)tj * NB * ld, ld, tk.K, acc, smem);
        gt_store<0, 4>(Ct, ld, acc);
    }
}

void launch_update_nt_queue(hipStream_t st, double* A, long ld, const UpdTask* tasks_dev, int ntasks, int* counter, int wgs) {
    LDS_OPT_IN(k_update_nt_queue);
    hipLaunchKernelGGL(k_update_nt_queue, dim3((unsigned)wgs), dim3(256), GT_LDS_BYTES, st, A, ld, tasks_dev, ntasks, counter);
}
#endif   // MI355GP_DIAG

// ------------------------------------------------------------------------------------------------
// Batched bottom-up triangular inverse (the dtrtri half of LAPACK dpotri, GPy/util/linalg.py:127-145).
// Level s merges diagonal blocks of nbt = 2^s tiles pairwise:
//   [X11 0; X21 X22] with X21 = -X22 * L21 * X11.
// stage 1: T21 = L21 * X11   (X11 lower triangular -> k from tj to the end of the left block)
// stage 2: X21 = -X22 * T21  (X22 lower triangular -> k from the start of the right block to ti)
// Tiles of one level: pair p joins the diagonal blocks [2 p nbt, (2p+1) nbt) and [(2p+1) nbt, (2p+2) nbt); its X21 has nbt x nbt
// tiles -- except the LAST pair of a matrix whose tile count is not a multiple of 2 nbt, whose right block has only
// rows_last = nt - right0 rows (possibly none).  Only LIVE tiles are enumerated: a grid padded to nbt x nbt per pair hands most
// of its workgroups an early exit (N = 4608, top level: 7 of 8), and since workgroups go to the XCDs round-robin by index the
// live ones then land unevenly (the dead-workgroup effect of section 6 of DESIGN.md: trtri 1.82 -> 0.9 ms at N = 4608).
__host__ __device__ __forceinline__ int trtri_rows_last(int nt, int nbt) {
    const int pairs = (nt + 2 * nbt - 1) / (2 * nbt), right0 = (2 * (pairs - 1) + 1) * nbt;
    const int rows = nt - right0;
    return rows < 0 ? 0 : (rows > nbt ? nbt : rows);
}
__host__ __device__ __forceinline__ long trtri_level_tiles(int nt, int nbt) {
    const int pairs = (nt + 2 * nbt - 1) / (2 * nbt);
    return (long)(pairs - 1) * nbt * nbt + (long)nbt * trtri_rows_last(nt, nbt);
}
template <int STAGE>
__device__ __forceinline__ void trtri_map(int bid, int nt, int nbt, int& p, int& ri, int& cj) {
    const int per = nbt * nbt, pairs = (nt + 2 * nbt - 1) / (2 * nbt), full = (pairs - 1) * per;
    if (bid < full) {
        p = bid / per;
        const int rem = bid - p * per;
        if (STAGE == 1) {          // heavy tiles (small cj) first
            cj = rem / nbt;
            ri = rem - cj * nbt;
        } else {                   // heavy tiles (large ri) first
            ri = nbt - 1 - rem / nbt;
            cj = rem % nbt;
        }
    } else {
        p = pairs - 1;
        const int rem = bid - full, rows = trtri_rows_last(nt, nbt);
        if (STAGE == 1) {
            cj = rem / rows;
            ri = rem - cj * rows;
        } else {
            ri = rows - 1 - rem / nbt;
            cj = rem % nbt;
        }
    }
}

template <int STAGE, int NW>
__device__ __forceinline__ void trtri_stage_tile(int bid, const double* __restrict__ L, double* __restrict__ X,
                                                 double* __restrict__ T, long ld, int nt, int nbt, double* smem) {
    int p, ri, cj;
    trtri_map<STAGE>(bid, nt, nbt, p, ri, cj);
    const int left0 = 2 * p * nbt, right0 = left0 + nbt;
    const int ti = right0 + ri, tj = left0 + cj;
    if (ti >= nt) return;
    d4 acc[4][GTCfg<NW>::NI];
    gt_zero<NW>(acc);
    if (STAGE == 1) {
        const int K = (right0 - tj) * NB;
        gemm_tile_128<true, false, NW>(L + (long)ti * NB * ld + (long)tj * NB, ld,
                                   X + (long)tj * NB * ld + (long)tj * NB, ld, K, acc, smem);
        gt_store<0, NW>(T + (long)ti * NB * ld + (long)tj * NB, ld, acc);
    } else {
        const int K = (ti - right0 + 1) * NB;
        gemm_tile_128<true, false, NW>(X + (long)ti * NB * ld + (long)right0 * NB, ld,
                                   T + (long)right0 * NB * ld + (long)tj * NB, ld, K, acc, smem);
        gt_store<1, NW>(X + (long)ti * NB * ld + (long)tj * NB, ld, acc);
    }
}

// 64 x 64 quadrants of the same tiles (launches with few tiles: the low levels of every inverse, every level of a small one)
template <int STAGE>
__global__ __launch_bounds__(256) void k_trtri_stage64(const double* __restrict__ L, double* __restrict__ X,
                                                       double* __restrict__ T, long ld, int nt, int nbt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bid = blockIdx.x >> 2, qi = (blockIdx.x >> 1) & 1, qj = blockIdx.x & 1;
    int p, ri, cj;
    trtri_map<STAGE>(bid, nt, nbt, p, ri, cj);
    const int left0 = 2 * p * nbt, right0 = left0 + nbt;
    const int ti = right0 + ri, tj = left0 + cj;
    if (ti >= nt) return;
    d4 acc[2][2];
    gt64_zero(acc);
    const long r0 = (long)ti * NB + qi * 64, c0 = (long)tj * NB + qj * 64;
    if (STAGE == 1) {
        const int K = (right0 - tj) * NB;
        gemm_tile_64_v3<true, false>(L + r0 * ld + (long)tj * NB, ld, X + (long)tj * NB * ld + c0, ld, K, acc, smem);
        gt64_store<0>(T + r0 * ld + c0, ld, acc);
    } else {
        const int K = (ti - right0 + 1) * NB;
        gemm_tile_64_v3<true, false>(X + r0 * ld + (long)right0 * NB, ld, T + (long)right0 * NB * ld + c0, ld, K, acc, smem);
        gt64_store<1>(X + r0 * ld + c0, ld, acc);
    }
}

template <int STAGE, int NW>
__global__ LB(NW) void k_trtri_stage(const double* __restrict__ L, double* __restrict__ X,
                                                        double* __restrict__ T, long ld, int nt, int nbt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    trtri_stage_tile<STAGE, NW>((int)blockIdx.x, L, X, T, ld, nt, nbt, smem);
}

// Stage 1 with a shared tile counter: several launches of this kernel (on different streams, with different grids) drain
// ONE tile list.  The look-ahead factorisation puts an instance with a bounded grid on a CU-masked stream while potrf
// still runs and a second, machine-wide instance on the main stream afterwards: whatever the first one did not get to is
// picked up at full speed, nothing is left behind on the masked stream.
__global__ LB(4) void k_trtri_stage1_steal(const double* __restrict__ L, double* __restrict__ X, double* __restrict__ T,
                                           long ld, int nt, int nbt, int* __restrict__ counter, int nblocks) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_bid;
    for (;;) {
        if (threadIdx.x == 0) s_bid = atomicAdd(counter, 1);
        __syncthreads();
        const int bid = s_bid;
        __syncthreads();
        if (bid >= nblocks) return;
        trtri_stage_tile<1, 4>(bid, L, X, T, ld, nt, nbt, smem);
    }
}

void launch_trtri_stage1_steal(hipStream_t st, const double* L, double* X, double* T, long ld, int nt, int level,
                               int* counter, int grid) {
    const int nbt = 1 << level;
    if (nbt >= nt) return;
    const int nblocks = (int)trtri_level_tiles(nt, nbt);
    if (nblocks <= 0) return;
    LDS_OPT_IN(k_trtri_stage1_steal);
    if (grid > nblocks) grid = nblocks;
    hipLaunchKernelGGL(k_trtri_stage1_steal, dim3((unsigned)grid), dim3(256), GT_LDS_BYTES, st, L, X, T, ld, nt, nbt,
                       counter, nblocks);
}

template <int NW>
static void launch_trtri_level_t(hipStream_t st, long nblocks, const double* L, double* X, double* T, long ld, int nt,
                                 int nbt, int stages) {
    LDS_OPT_IN((k_trtri_stage<1, NW>));
    LDS_OPT_IN((k_trtri_stage<2, NW>));
    if (stages & 1)
        hipLaunchKernelGGL((k_trtri_stage<1, NW>), dim3((unsigned)nblocks), dim3(NW * 64), GT_LDS_BYTES, st, L, X, T, ld, nt, nbt);
    if (stages & 2)
        hipLaunchKernelGGL((k_trtri_stage<2, NW>), dim3((unsigned)nblocks), dim3(NW * 64), GT_LDS_BYTES, st, L, X, T, ld, nt, nbt);
}

// stages: bit 0 = T21 = L21 X11, bit 1 = X21 = -X22 T21 (3 = the whole level)
void launch_trtri_level(hipStream_t st, const double* L, double* X, double* T, long ld, int nt, int level, int stages) {
    const int nbt = 1 << level;
    if (nbt >= nt) return;
    const long nblocks = trtri_level_tiles(nt, nbt);
    if (nblocks <= 0) return;
    static const int tri64_max = env_int("TRI64_MAX", GEMM_DEFAULT_TRI64_MAX);
    if (nblocks <= tri64_max) {
        if (stages & 1)
            hipLaunchKernelGGL((k_trtri_stage64<1>), dim3((unsigned)(4 * nblocks)), dim3(256), GT64_LDS_BYTES, st, L, X, T, ld, nt, nbt);
        if (stages & 2)
            hipLaunchKernelGGL((k_trtri_stage64<2>), dim3((unsigned)(4 * nblocks)), dim3(256), GT64_LDS_BYTES, st, L, X, T, ld, nt, nbt);
        return;
    }
    launch_trtri_level_t<4>(st, nblocks, L, X, T, ld, nt, nbt, stages);
}

// ------------------------------------------------------------------------------------------------
// W = X^T X for lower-triangular X (the dlauum half of LAPACK dpotri): W[ti,tj] = sum_{tk>=ti} X[tk,ti]^T X[tk,tj].
template <int NW>
__global__ LB(NW) void k_lauum(const double* __restrict__ X, double* __restrict__ W, long ld, int nt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bid = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
    while ((long)ti * (ti + 1) / 2 > bid) --ti;
    while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    const int tj = bid - (int)((long)ti * (ti + 1) / 2);
    d4 acc[4][GTCfg<NW>::NI];
    gt_zero<NW>(acc);
    const int K = (nt - ti) * NB;
    gemm_tile_128<false, false, NW>(X + (long)ti * NB * ld + (long)ti * NB, ld, X + (long)ti * NB * ld + (long)tj * NB, ld,
                                K, acc, smem);
    gt_store<0, NW>(W + (long)ti * NB * ld + (long)tj * NB, ld, acc);
}

__global__ __launch_bounds__(256) void k_lauum64(const double* __restrict__ X, double* __restrict__ W, long ld, int nt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bid = blockIdx.x >> 2, qi = (blockIdx.x >> 1) & 1, qj = blockIdx.x & 1;
    int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
    while ((long)ti * (ti + 1) / 2 > bid) --ti;
    while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    const int tj = bid - (int)((long)ti * (ti + 1) / 2);
    d4 acc[2][2];
    gt64_zero(acc);
    const int K = (nt - ti) * NB;
    const long r0 = (long)ti * NB + qi * 64, c0 = (long)tj * NB + qj * 64;
    gemm_tile_64_v3<false, false>(X + (long)ti * NB * ld + r0, ld, X + (long)ti * NB * ld + c0, ld, K, acc, smem);
    gt64_store<0>(W + r0 * ld + c0, ld, acc);
}

// The same on a work list with the long k ranges CUT (small matrices: nt <= LAUUM_SPLIT_MAX_NT).  W(k,l) sums over the tile rows i >= k, so the
// tiles of the first block columns carry K ~ N while most carry a fraction of it: one quadrant of tile (0,0) at N = 4096 is 3.4e7
// flops, 0.44 ms at the quarter of a CU it gets while four workgroups share the CU -- alone more than the whole product needs at
// the machine's rate (0.29 ms).  Every quadrant's k range is cut into chunks of at most LAUUM_KC rows; a quadrant with ONE chunk
// stores W directly, the others store partial products that k_lauum64_reduce adds up in chunk order (fixed order: run-to-run
// bit-reproducible; no atomics).  Items are issued longest first.
#define LAUUM_KC 1024
#define LAUUM_T128_MIN_NT 24
// 128 x 128 items: the chunk length decides how the list packs onto the machine (two workgroups per CU, 512 slots): at nt = 32 chunks
// of 1024 rows are 1000 items = two rounds of 8 k-steps where the work is 11.7 per slot.  The length is therefore chosen PER nt by
// running the list (longest first, next item to the first free slot -- what the dispatcher does) through a cost model calibrated on
// the two measured points (nt = 24: 0.25 ms, nt = 32: 0.47 ms at 1024): one k-step of 128 rows 27.4 us with two workgroups on a
// CU, 8 us per item, plus the reduction's traffic (every partial is 128 KB written and read).  MI355GP_LAUUM_KC (rows) overrides.
// Measured against fixed lengths (tools/lauum_kc_probe.sh): the model's pick is the fastest or within 1 % of it at every size tried
// (N=3072: 768 rows, 0.253 -> 0.236 ms; N=4096: 1536, 0.470 -> 0.455; N=4608: 2176, 0.655 at 1024 -> 0.607; N=5120: 0.848 -> 0.790).
static int lauum_kc_rows_128(int nt) {
    static const int forced = env_int("LAUUM_KC", 0);
    if (forced >= NB) return forced / NB * NB;
    int best = LAUUM_KC;
    double best_us = 1e30;
    for (int kc = 2; kc <= 32 && kc <= nt; ++kc) {             // chunk length in k-steps of 128 rows
        std::vector<int> len;
        long parts = 0;
        for (int ti = 0; ti < nt; ++ti) {
            const int K = nt - ti, nch = (K + kc - 1) / kc;
            for (int tj = 0; tj <= ti; ++tj)
                for (int c = 0; c < nch; ++c) len.push_back(c + 1 < nch ? kc : K - c * kc);
            if (nch > 1) parts += (long)nch * (ti + 1);
        }
        std::sort(len.begin(), len.end(), [](int a, int b) { return a > b; });
        std::vector<double> slot(512, 0.0);                    // a binary heap would do; 512 x a few thousand items is nothing
        std::make_heap(slot.begin(), slot.end(), [](double a, double b) { return a > b; });
        double end = 0.0;
        for (int l : len) {
            std::pop_heap(slot.begin(), slot.end(), [](double a, double b) { return a > b; });
            const double t = slot.back() + 27.4 * l + 8.0;
            slot.back() = t;
            if (t > end) end = t;
            std::push_heap(slot.begin(), slot.end(), [](double a, double b) { return a > b; });
        }
        const double us = end + (double)parts * (NB * NB * 8.0) / 3.0e6;   // the reduction reads the partials at ~3 TB/s
        if (us < best_us - 0.5) {
            best_us = us;
            best = kc * NB;
        }
    }
    return best;
}
__global__ __launch_bounds__(256) void k_lauum64_items(const double* __restrict__ X, double* __restrict__ W, long ld,
                                                       const LauumItem* __restrict__ items, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LauumItem it = items[blockIdx.x];
    d4 acc[2][2];
    gt64_zero(acc);
    const long r0 = (long)it.ti * NB + (it.q >> 1) * 64, c0 = (long)it.tj * NB + (it.q & 1) * 64;
    gemm_tile_64_v3<false, false>(X + (long)it.k0 * ld + r0, ld, X + (long)it.k0 * ld + c0, ld, it.klen, acc, smem);
    if (it.part < 0) gt64_store<0>(W + r0 * ld + c0, ld, acc);
    else gt64_store<0>(part + (long)it.part * 4096, 64, acc);
}

// From N ~ 3000 on the chunks are long enough for the 128 x 128 tile (half the operand traffic per flop of the 64 x 64 one): the
// same list with whole tiles as items (q = -1), partials of 128 x 128.
__global__ LB(4) void k_lauum128_items(const double* __restrict__ X, double* __restrict__ W, long ld,
                                       const LauumItem* __restrict__ items, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LauumItem it = items[blockIdx.x];
    d4 acc[4][GTCfg<4>::NI];
    gt_zero<4>(acc);
    const long r0 = (long)it.ti * NB, c0 = (long)it.tj * NB;
    gemm_tile_128<false, false, 4>(X + (long)it.k0 * ld + r0, ld, X + (long)it.k0 * ld + c0, ld, it.klen, acc, smem);
    if (it.part < 0) gt_store<0, 4>(W + r0 * ld + c0, ld, acc);
    else gt_store<0, 4>(part + (long)it.part * (NB * NB), NB, acc);
}

__global__ __launch_bounds__(256) void k_lauum128_reduce(double* __restrict__ W, long ld, const LauumSum* __restrict__ sums,
                                                         const double* __restrict__ part) {
    const LauumSum sm = sums[blockIdx.x];
    const long r0 = (long)sm.ti * NB, c0 = (long)sm.tj * NB;
    for (int e = threadIdx.x; e < NB * NB / 2; e += 256) {
        const int i = e >> 6, j = 2 * (e & 63);
        d2 s = *reinterpret_cast<const d2*>(part + (long)sm.first * (NB * NB) + i * NB + j);
        for (int c = 1; c < sm.n; ++c) s += *reinterpret_cast<const d2*>(part + (long)(sm.first + c) * (NB * NB) + i * NB + j);
        *reinterpret_cast<d2*>(W + (r0 + i) * ld + c0 + j) = s;
    }
}

__global__ __launch_bounds__(256) void k_lauum64_reduce(double* __restrict__ W, long ld, const LauumSum* __restrict__ sums,
                                                        const double* __restrict__ part) {
    const LauumSum sm = sums[blockIdx.x];
    const long r0 = (long)sm.ti * NB + (sm.q >> 1) * 64, c0 = (long)sm.tj * NB + (sm.q & 1) * 64;
    for (int e = threadIdx.x; e < 2048; e += 256) {           // two doubles per element index
        const int i = e >> 5, j = 2 * (e & 31);
        d2 s = *reinterpret_cast<const d2*>(part + (long)sm.first * 4096 + i * 64 + j);
        for (int c = 1; c < sm.n; ++c) s += *reinterpret_cast<const d2*>(part + (long)(sm.first + c) * 4096 + i * 64 + j);
        *reinterpret_cast<d2*>(W + (r0 + i) * ld + c0 + j) = s;
    }
}

// work list of the split lauum for nt x nt tiles (host side, cached by the caller): items longest first
int lauum_split_tile(int nt) { return nt >= LAUUM_T128_MIN_NT ? 128 : 64; }

void lauum_split_plan(int nt, std::vector<LauumItem>& items, std::vector<LauumSum>& sums, int* nparts) {
    items.clear();
    sums.clear();
    int np = 0;
    const int nq = lauum_split_tile(nt) == 128 ? 1 : 4;       // whole tiles (q = -1 in spirit: q is ignored) or quadrants
    // quadrant items: 512 rows up to nt = 16, 1024 above (tools/lauum_kc64_probe.sh: N=1536 0.064 -> 0.051 ms, N=2048 0.091 -> 0.085;
    // N=2560 / 2944 are fastest at 1024); MI355GP_LAUUM_KC64 (rows) overrides
    static const int kc64 = env_int("LAUUM_KC64", 0) / NB * NB;
    const int kc = (nq == 1) ? lauum_kc_rows_128(nt) : (kc64 >= NB ? kc64 : (nt <= 16 ? 512 : LAUUM_KC));
    for (int ti = 0; ti < nt; ++ti)
        for (int tj = 0; tj <= ti; ++tj)
            for (int q = 0; q < nq; ++q) {
                const int k0 = ti * NB, K = (nt - ti) * NB, nch = (K + kc - 1) / kc;
                if (nch > 1) sums.push_back(LauumSum{ti, tj, q, np, nch});
                for (int c = 0; c < nch; ++c) {
                    const int len = (c + 1 < nch) ? kc : K - c * kc;
                    items.push_back(LauumItem{ti, tj, q, k0 + c * kc, len, nch > 1 ? np++ : -1});
                }
            }
    std::stable_sort(items.begin(), items.end(), [](const LauumItem& a, const LauumItem& b) { return a.klen > b.klen; });
    *nparts = np;
}

void launch_lauum_split(hipStream_t st, const double* X, double* W, long ld, int nt, const LauumItem* items_dev, int nitems,
                        const LauumSum* sums_dev, int nsums, double* part) {
    if (lauum_split_tile(nt) == 128) {
        LDS_OPT_IN(k_lauum128_items);
        hipLaunchKernelGGL(k_lauum128_items, dim3((unsigned)nitems), dim3(256), GT_LDS_BYTES, st, X, W, ld, items_dev, part);
        if (nsums > 0) hipLaunchKernelGGL(k_lauum128_reduce, dim3((unsigned)nsums), dim3(256), 0, st, W, ld, sums_dev, part);
        return;
    }
    hipLaunchKernelGGL(k_lauum64_items, dim3((unsigned)nitems), dim3(256), GT64_LDS_BYTES, st, X, W, ld, items_dev, part);
    if (nsums > 0) hipLaunchKernelGGL(k_lauum64_reduce, dim3((unsigned)nsums), dim3(256), 0, st, W, ld, sums_dev, part);
}

template <int NW>
static void launch_lauum_t(hipStream_t st, long nblocks, const double* X, double* W, long ld, int nt) {
    LDS_OPT_IN((k_lauum<NW>));
    hipLaunchKernelGGL((k_lauum<NW>), dim3((unsigned)nblocks), dim3(NW * 64), GT_LDS_BYTES, st, X, W, ld, nt);
}

bool lauum_uses_64(int nt) {
    static const int lauum64_max = env_int("LAUUM64_MAX", GEMM_DEFAULT_LAUUM64_MAX);
    return (long)nt * (nt + 1) / 2 <= lauum64_max;
}

void launch_lauum(hipStream_t st, const double* X, double* W, long ld, int nt) {
    const long nblocks = (long)nt * (nt + 1) / 2;
    static const int lauum64_max = env_int("LAUUM64_MAX", GEMM_DEFAULT_LAUUM64_MAX);
    if (nblocks <= lauum64_max) {
        hipLaunchKernelGGL(k_lauum64, dim3((unsigned)(4 * nblocks)), dim3(256), GT64_LDS_BYTES, st, X, W, ld, nt);
        return;
    }
    launch_lauum_t<4>(st, nblocks, X, W, ld, nt);
}

// ------------------------------------------------------------------------------------------------
// Out[ti,tj] = sum_{tk<=ti} X[ti,tk] * B[tk,tj] with X lower triangular (npad x npad), B (npad x mpad):
// L^-1 K(X, X*) of the predictive variance (dtrtrs in GPy/inference/latent_function_inference/posterior.py:286,296).
template <int NW>
__global__ LB(NW) void k_trmm_lower(const double* __restrict__ X, long ldx,
                                                       const double* __restrict__ B, long ldb,
                                                       double* __restrict__ Out, long ldo, int ntc) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ti = (int)(gridDim.x / ntc) - 1 - (int)(blockIdx.x / ntc), tj = blockIdx.x % ntc;     // longest k range first
    d4 acc[4][GTCfg<NW>::NI];
    gt_zero<NW>(acc);
    gemm_tile_128<true, false, NW>(X + (long)ti * NB * ldx, ldx, B + (long)tj * NB, ldb, (ti + 1) * NB, acc, smem);
    gt_store<0, NW>(Out + (long)ti * NB * ldo + (long)tj * NB, ldo, acc);
}

// Out[ti,tj] = sum_{tk>=ti} X[tk,ti]^T * B[tk,tj]: the transposed product X^T B (with k_trmm_lower: Ky^-1 B = X^T (X B) for
// the variance part of GP.predictive_gradients, core/gp.py:463-465)
template <int NW>
__global__ LB(NW) void k_trmm_lower_T(const double* __restrict__ X, long ldx, const double* __restrict__ B, long ldb,
                                      double* __restrict__ Out, long ldo, int ntc, int nt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ti = blockIdx.x / ntc, tj = blockIdx.x % ntc;
    d4 acc[4][GTCfg<NW>::NI];
    gt_zero<NW>(acc);
    gemm_tile_128<false, false, NW>(X + (long)ti * NB * ldx + (long)ti * NB, ldx, B + (long)ti * NB * ldb + (long)tj * NB, ldb,
                                    (nt - ti) * NB, acc, smem);
    gt_store<0, NW>(Out + (long)ti * NB * ldo + (long)tj * NB, ldo, acc);
}

void launch_trmm_lower_T(hipStream_t st, const double* X, long ldx, const double* B, long ldb, double* Out, long ldo,
                         int ntr, int ntc) {
    LDS_OPT_IN((k_trmm_lower_T<4>));
    hipLaunchKernelGGL((k_trmm_lower_T<4>), dim3((unsigned)(ntr * ntc)), dim3(256), GT_LDS_BYTES, st, X, ldx, B, ldb, Out,
                       ldo, ntc, ntr);
}

template <int NW>
static void launch_trmm_lower_t(hipStream_t st, const double* X, long ldx, const double* B, long ldb, double* Out,
                                long ldo, int ntr, int ntc) {
    LDS_OPT_IN((k_trmm_lower<NW>));
    hipLaunchKernelGGL((k_trmm_lower<NW>), dim3((unsigned)(ntr * ntc)), dim3(NW * 64), GT_LDS_BYTES, st, X, ldx, B, ldb,
                       Out, ldo, ntc);
}

void launch_trmm_lower(hipStream_t st, const double* X, long ldx, const double* B, long ldb, double* Out, long ldo,
                       int ntr, int ntc) {
    launch_trmm_lower_t<4>(st, X, ldx, B, ldb, Out, ldo, ntr, ntc);
}

// The four triangular products of the M x M algebra on 64 x 64 quadrants: a 2048^2 output has only 256 128-tiles -- one per CU, so
// the launch lasts as long as its LONGEST k range and skipping the zero half of X buys nothing; 1024 quadrants with the long
// ranges first balance (the full product is 0.27 ms at M = 2048, the triangular one half the flops).
//   MODE 0: Out = X B      (k <  (ti+1) 128)     MODE 1: Out = X^T B   (k >= ti 128)
//   MODE 2: Out = B X^T    (k <  (tj+1) 128)     MODE 3: Out = B X     (k >= tj 128)          X lower triangular, nt x nt tiles
template <int MODE>
__global__ __launch_bounds__(256) void k_trmm64(const double* __restrict__ X, long ldx, const double* __restrict__ B, long ldb,
                                                double* __restrict__ Out, long ldo, int nt, int ntother, double alpha) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x & 3, qi = q >> 1, qj = q & 1, bid = blockIdx.x >> 2;
    int ti, tj;                                                // heavy tiles first
    if (MODE == 0) { ti = nt - 1 - bid / ntother; tj = bid % ntother; }
    else if (MODE == 1) { ti = bid / ntother; tj = bid % ntother; }
    else if (MODE == 2) { tj = nt - 1 - bid / ntother; ti = bid % ntother; }
    else { tj = bid / ntother; ti = bid % ntother; }
    const long r0 = (long)ti * NB + qi * 64, c0 = (long)tj * NB + qj * 64;
    d4 acc[2][2];
    gt64_zero(acc);
    if (MODE == 0) gemm_tile_64_v3<true, false>(X + r0 * ldx, ldx, B + c0, ldb, (ti + 1) * NB, acc, smem);
    else if (MODE == 1) gemm_tile_64_v3<false, false>(X + (long)ti * NB * ldx + r0, ldx, B + (long)ti * NB * ldb + c0, ldb, (nt - ti) * NB, acc, smem);
    else if (MODE == 2) gemm_tile_64_v3<true, true>(B + r0 * ldb, ldb, X + c0 * ldx, ldx, (tj + 1) * NB, acc, smem);
    else gemm_tile_64_v3<true, false>(B + r0 * ldb + (long)tj * NB, ldb, X + (long)tj * NB * ldx + c0, ldx, (nt - tj) * NB, acc, smem);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] *= alpha;
    gt64_store<0>(Out + r0 * ldo + c0, ldo, acc);
}

// mode: 0 X B, 1 X^T B, 2 B X^T, 3 B X; X: nt x nt tiles (lower triangular), the other dimension of B / Out: ntother tiles
void launch_trmm64(hipStream_t st, int mode, const double* X, long ldx, const double* B, long ldb, double* Out, long ldo, int nt,
                   int ntother, double alpha) {
    const unsigned grid = (unsigned)(4 * nt * ntother);
    switch (mode) {
        case 0: hipLaunchKernelGGL((k_trmm64<0>), dim3(grid), dim3(256), GT64_LDS_BYTES, st, X, ldx, B, ldb, Out, ldo, nt, ntother, alpha); break;
        case 1: hipLaunchKernelGGL((k_trmm64<1>), dim3(grid), dim3(256), GT64_LDS_BYTES, st, X, ldx, B, ldb, Out, ldo, nt, ntother, alpha); break;
        case 2: hipLaunchKernelGGL((k_trmm64<2>), dim3(grid), dim3(256), GT64_LDS_BYTES, st, X, ldx, B, ldb, Out, ldo, nt, ntother, alpha); break;
        default: hipLaunchKernelGGL((k_trmm64<3>), dim3(grid), dim3(256), GT64_LDS_BYTES, st, X, ldx, B, ldb, Out, ldo, nt, ntother, alpha); break;
    }
}

// C (mpad x mpad, ldc) = alpha * A^T A + beta * C with A (K x mpad): the K** - tmp^T tmp of full_cov prediction
void launch_gemm_tn_sq(hipStream_t st, const double* A, long lda, long K, double* C, long ldc, int nt, double alpha,
                       double beta);

// ------------------------------------------------------------------------------------------------
// General C = alpha*op(A)*op(B) + beta*C on full tiles (diagnostics / prediction).
// shader / wall clock ticks of workgroup 0 of the last k_gemm_full launch (effective shader MHz under the GEMM's own load)
__device__ long long g_gemm_clk[2];

template <bool AK, bool BK, int NW>
__global__ LB(NW) void k_gemm_full(const double* __restrict__ A, long lda,
                                                      const double* __restrict__ B, long ldb, double* __restrict__ C,
                                                      long ldc, int K, int ntc, double alpha, double beta, int swz) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int bid = blockIdx.x;
    if (swz & 1) {   // XCD-aware order: workgroup b runs on XCD b % 8; give each XCD 8x8 super-tiles (needs ntr, ntc % 8 == 0)
        const int xcd = bid & 7, loc = bid >> 3;
        const int stiles = ntc >> 3;                      // super-tiles per row
        const int sidx = (loc >> 6) * 8 + xcd;            // super-tile number handled by this XCD in this round
        const int in = loc & 63;
        bid = ((sidx / stiles) * 8 + (in >> 3)) * ntc + (sidx % stiles) * 8 + (in & 7);
    }
    const int ti = bid / ntc, tj = bid % ntc;
    d4 acc[4][GTCfg<NW>::NI];
    gt_zero<NW>(acc);
    const double* Ap = AK ? A + (long)ti * NB * lda : A + (long)ti * NB;
    const double* Bp = BK ? B + (long)tj * NB * ldb : B + (long)tj * NB;
    const long long c0 = clock64(), w0 = wall_clock64();
    gemm_tile_128<AK, BK, NW>(Ap, lda, Bp, ldb, K, acc, smem);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_gemm_clk[0] = clock64() - c0;
        g_gemm_clk[1] = wall_clock64() - w0;
    }
    double* Ct = C + (long)ti * NB * ldc + (long)tj * NB;
    if (beta == 0.0) {   // never read C: it may be uninitialised memory (0 * NaN would poison the result)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < GTCfg<NW>::NI; ++ni) acc[mi][ni] *= alpha;
        gt_store<0, NW>(Ct, ldc, acc);
    } else {
        gt_store<3, NW>(Ct, ldc, acc, alpha, beta);
    }
}

template <bool AK, bool BK>
static void launch_gemm_full(hipStream_t st, unsigned nblocks, const double* A, long lda, const double* B, long ldb,
                             double* C, long ldc, int K, int ntc, double alpha, double beta) {
    static const int dbg_ld0 = env_int("DBG_LD0", 0), dbg_swz = env_int("DBG_SWZ", 0);
    if (dbg_ld0) lda = ldb = 0;                           // every operand row aliases row 0: cache-resident loads
    static const int dbg_1wg = env_int("DBG_1WG", 0);
    const int swz = (dbg_swz && (ntc % 8 == 0) && ((nblocks / ntc) % 8 == 0)) ? 1 : 0;
    const size_t lds = dbg_1wg ? 100 * 1024 : GT_LDS_BYTES;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_full<AK, BK, 4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_gemm_full<AK, BK, 4>), dim3(nblocks), dim3(256), lds, st, A, lda, B, ldb, C, ldc, K,
                       ntc, alpha, beta, swz);
}

void launch_dbg_gemm(hipStream_t st, int a_mcontig, int b_ncontig, long M, long N, long K, const double* A,
                     const double* B, double* C, double alpha, double beta) {
    const int ntr = (int)(M / NB), ntc = (int)(N / NB);
    const unsigned grid = (unsigned)(ntr * ntc);
    const long lda = a_mcontig ? M : K, ldb = b_ncontig ? N : K;
    if (!a_mcontig && !b_ncontig) launch_gemm_full<true, true>(st, grid, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
    else if (!a_mcontig && b_ncontig) launch_gemm_full<true, false>(st, grid, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
    else if (a_mcontig && b_ncontig) launch_gemm_full<false, false>(st, grid, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
    else launch_gemm_full<false, true>(st, grid, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
}

void launch_gemm_tn_sq(hipStream_t st, const double* A, long lda, long K, double* C, long ldc, int nt, double alpha,
                       double beta) {
    launch_gemm_full<false, false>(st, (unsigned)(nt * nt), A, lda, A, lda, C, ldc, (int)K, nt, alpha, beta);
}

// effective shader clock (MHz) and shader cycles seen by workgroup 0 of the last k_gemm_full launch
int gemm_last_clock(double* mhz, double* cycles) {
    long long h[2] = {0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_clk), sizeof(h)) != hipSuccess) return -1;
    *mhz = h[1] > 0 ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    *cycles = (double)h[0];
    return 0;
}

void launch_gemm(hipStream_t st, int a_mcontig, int b_ncontig, long M, long N, long K, const double* A, long lda,
                 const double* B, long ldb, double* C, long ldc, double alpha, double beta) {
    const int ntr = (int)(M / NB), ntc = (int)(N / NB);
    const unsigned grid = (unsigned)(ntr * ntc);
    if (!a_mcontig && !b_ncontig) launch_gemm_full<true, true>(st, grid, A, lda, B, ldb, C, ldc, (int)K, ntc, alpha, beta);
    else if (!a_mcontig && b_ncontig) launch_gemm_full<true, false>(st, grid, A, lda, B, ldb, C, ldc, (int)K, ntc, alpha, beta);
    else if (a_mcontig && b_ncontig) launch_gemm_full<false, false>(st, grid, A, lda, B, ldb, C, ldc, (int)K, ntc, alpha, beta);
    else launch_gemm_full<false, true>(st, grid, A, lda, B, ldb, C, ldc, (int)K, ntc, alpha, beta);
}

// ------------------------------------------------------------------------------------------------
// psi2 = Kuf Kfu of the sparse path (the tdot of var_dtc.py:134 / the psi2 accumulation of
// var_dtc_parallel.py:91-116): Gram matrix of a tall (rows x mp) panel.  The output has only (mp/128)(mp/128+1)/2
// lower tiles (136 at M = 2048), far fewer than the 512 workgroup slots, so K is split S ways into separate partial
// matrices (summed in a fixed order afterwards: no atomics, bit-reproducible).
__global__ __launch_bounds__(256, 2) void k_gram_splitk(const double* __restrict__ P, long ldp, long rows_per_split,
                                                        long mp, int ntl, int accumulate, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int s = blockIdx.x / ntl, bid = blockIdx.x % ntl;
    int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
    while ((long)ti * (ti + 1) / 2 > bid) --ti;
    while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    const int tj = bid - (int)((long)ti * (ti + 1) / 2);
    const double* Ps = P + (long)s * rows_per_split * ldp;
    d4 acc[4][4];
    gt_zero<4>(acc);
    gemm_tile_128<false, false, 4>(Ps + (long)ti * NB, ldp, Ps + (long)tj * NB, ldp, (int)rows_per_split, acc, smem);
    double* C = part + (long)s * mp * mp + (long)ti * NB * mp + (long)tj * NB;
    if (accumulate) gt_store<3, 4>(C, mp, acc, 1.0, 1.0);
    else gt_store<0, 4>(C, mp, acc);
}

void launch_gram_splitk(hipStream_t st, const double* P, long ldp, long rows, long mp, int S, int accumulate,
                        double* part) {
    LDS_OPT_IN(k_gram_splitk);
    const int nt = (int)(mp / NB), ntl = nt * (nt + 1) / 2;
    hipLaunchKernelGGL(k_gram_splitk, dim3((unsigned)(ntl * S)), dim3(256), GT_LDS_BYTES, st, P, ldp, rows / S, mp, ntl,
                       accumulate, part);
}

