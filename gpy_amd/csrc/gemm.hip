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

// ------------------------------------------------------------------------------------------------
// Trailing update of the right-looking Cholesky (the dsyrk/dgemm inside LAPACK dpotrf, which GPy reaches
// through GPy/util/linalg.py:58):  C[ti,tj] -= A[ti,:] * B[tj,:]^T.
// `tri`: region is square on the diagonal -> enumerate the lower triangle only.
__global__ __launch_bounds__(256, 2) void k_update_nt(double* __restrict__ C, long ldc,
                                                      const double* __restrict__ A, long lda,
                                                      const double* __restrict__ B, long ldb, int K, int ntc,
                                                      int row0t, int col0t, int tri) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int ti, tj;
    const int bid = blockIdx.x;
    if (tri) {
        ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
        while ((long)ti * (ti + 1) / 2 > bid) --ti;
        while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
        tj = bid - (int)((long)ti * (ti + 1) / 2);
    } else {
        ti = bid / ntc;
        tj = bid - ti * ntc;
        if (col0t + tj > row0t + ti) return;
    }
    d4 acc[4][4];
    gt_zero(acc);
    gemm_tile_128<true, true>(A + (long)ti * NB * lda, lda, B + (long)tj * NB * ldb, ldb, K, acc, smem);
    gt_store<2>(C + (long)ti * NB * ldc + (long)tj * NB, ldc, acc);
}

void launch_update_nt(hipStream_t st, double* C, long ldc, const double* A, long lda, const double* B, long ldb,
                      int K, int ntr, int ntc, int row0t, int col0t) {
    if (ntr <= 0 || ntc <= 0) return;
    const int tri = (row0t == col0t && ntr == ntc) ? 1 : 0;
    const long nblocks = tri ? (long)ntr * (ntr + 1) / 2 : (long)ntr * ntc;
    LDS_OPT_IN(k_update_nt);
    hipLaunchKernelGGL(k_update_nt, dim3((unsigned)nblocks), dim3(256), GT_LDS_BYTES, st, C, ldc, A, lda, B, ldb,
                       K, ntc, row0t, col0t, tri);
}

// ------------------------------------------------------------------------------------------------
// Batched bottom-up triangular inverse (the dtrtri half of LAPACK dpotri, GPy/util/linalg.py:127-145).
// Level s merges diagonal blocks of nbt = 2^s tiles pairwise:
//   [X11 0; X21 X22] with X21 = -X22 * L21 * X11.
// stage 1: T21 = L21 * X11   (X11 lower triangular -> k from tj to the end of the left block)
// stage 2: X21 = -X22 * T21  (X22 lower triangular -> k from the start of the right block to ti)
template <int STAGE>
__global__ __launch_bounds__(256, 2) void k_trtri_stage(const double* __restrict__ L, double* __restrict__ X,
                                                        double* __restrict__ T, long ld, int nt, int nbt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int per = nbt * nbt;
    const int p = blockIdx.x / per;
    const int rem = blockIdx.x - p * per;
    int ri, cj;
    if (STAGE == 1) {          // heavy tiles (small cj) first
        cj = rem / nbt;
        ri = rem - cj * nbt;
    } else {                   // heavy tiles (large ri) first
        ri = nbt - 1 - rem / nbt;
        cj = rem % nbt;
    }
    const int left0 = 2 * p * nbt, right0 = left0 + nbt;
    const int ti = right0 + ri, tj = left0 + cj;
    if (ti >= nt) return;
    d4 acc[4][4];
    gt_zero(acc);
    if (STAGE == 1) {
        const int K = (right0 - tj) * NB;
        gemm_tile_128<true, false>(L + (long)ti * NB * ld + (long)tj * NB, ld,
                                   X + (long)tj * NB * ld + (long)tj * NB, ld, K, acc, smem);
        gt_store<0>(T + (long)ti * NB * ld + (long)tj * NB, ld, acc);
    } else {
        const int K = (ti - right0 + 1) * NB;
        gemm_tile_128<true, false>(X + (long)ti * NB * ld + (long)right0 * NB, ld,
                                   T + (long)right0 * NB * ld + (long)tj * NB, ld, K, acc, smem);
        gt_store<1>(X + (long)ti * NB * ld + (long)tj * NB, ld, acc);
    }
}

void launch_trtri_level(hipStream_t st, const double* L, double* X, double* T, long ld, int nt, int level) {
    const int nbt = 1 << level;
    if (nbt >= nt) return;
    const int pairs = (nt + 2 * nbt - 1) / (2 * nbt);
    const long nblocks = (long)pairs * nbt * nbt;
    LDS_OPT_IN(k_trtri_stage<1>);
    LDS_OPT_IN(k_trtri_stage<2>);
    hipLaunchKernelGGL(k_trtri_stage<1>, dim3((unsigned)nblocks), dim3(256), GT_LDS_BYTES, st, L, X, T, ld, nt, nbt);
    hipLaunchKernelGGL(k_trtri_stage<2>, dim3((unsigned)nblocks), dim3(256), GT_LDS_BYTES, st, L, X, T, ld, nt, nbt);
}

// ------------------------------------------------------------------------------------------------
// W = X^T X for lower-triangular X (the dlauum half of LAPACK dpotri): W[ti,tj] = sum_{tk>=ti} X[tk,ti]^T X[tk,tj].
__global__ __launch_bounds__(256, 2) void k_lauum(const double* __restrict__ X, double* __restrict__ W, long ld,
                                                  int nt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bid = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
    while ((long)ti * (ti + 1) / 2 > bid) --ti;
    while ((long)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    const int tj = bid - (int)((long)ti * (ti + 1) / 2);
    d4 acc[4][4];
    gt_zero(acc);
    const int K = (nt - ti) * NB;
    gemm_tile_128<false, false>(X + (long)ti * NB * ld + (long)ti * NB, ld, X + (long)ti * NB * ld + (long)tj * NB, ld,
                                K, acc, smem);
    gt_store<0>(W + (long)ti * NB * ld + (long)tj * NB, ld, acc);
}

void launch_lauum(hipStream_t st, const double* X, double* W, long ld, int nt) {
    const long nblocks = (long)nt * (nt + 1) / 2;
    LDS_OPT_IN(k_lauum);
    hipLaunchKernelGGL(k_lauum, dim3((unsigned)nblocks), dim3(256), GT_LDS_BYTES, st, X, W, ld, nt);
}

// ------------------------------------------------------------------------------------------------
// Out[ti,tj] = sum_{tk<=ti} X[ti,tk] * B[tk,tj] with X lower triangular (npad x npad), B (npad x mpad):
// L^-1 K(X, X*) of the predictive variance (dtrtrs in GPy/inference/latent_function_inference/posterior.py:286,296).
__global__ __launch_bounds__(256, 2) void k_trmm_lower(const double* __restrict__ X, long ldx,
                                                       const double* __restrict__ B, long ldb,
                                                       double* __restrict__ Out, long ldo, int ntc) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ti = blockIdx.x / ntc, tj = blockIdx.x % ntc;
    d4 acc[4][4];
    gt_zero(acc);
    gemm_tile_128<true, false>(X + (long)ti * NB * ldx, ldx, B + (long)tj * NB, ldb, (ti + 1) * NB, acc, smem);
    gt_store<0>(Out + (long)ti * NB * ldo + (long)tj * NB, ldo, acc);
}

void launch_trmm_lower(hipStream_t st, const double* X, long ldx, const double* B, long ldb, double* Out, long ldo,
                       int ntr, int ntc) {
    LDS_OPT_IN(k_trmm_lower);
    hipLaunchKernelGGL(k_trmm_lower, dim3((unsigned)(ntr * ntc)), dim3(256), GT_LDS_BYTES, st, X, ldx, B, ldb, Out, ldo,
                       ntc);
}

// C (mpad x mpad, ldc) = alpha * A^T A + beta * C with A (K x mpad): the K** - tmp^T tmp of full_cov prediction
void launch_gemm_tn_sq(hipStream_t st, const double* A, long lda, long K, double* C, long ldc, int nt, double alpha,
                       double beta);

// ------------------------------------------------------------------------------------------------
// General C = alpha*op(A)*op(B) + beta*C on full tiles (diagnostics / prediction).
template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void k_gemm_full(const double* __restrict__ A, long lda,
                                                      const double* __restrict__ B, long ldb, double* __restrict__ C,
                                                      long ldc, int K, int ntc, double alpha, double beta) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ti = blockIdx.x / ntc, tj = blockIdx.x % ntc;
    d4 acc[4][4];
    gt_zero(acc);
    const double* Ap = AK ? A + (long)ti * NB * lda : A + (long)ti * NB;
    const double* Bp = BK ? B + (long)tj * NB * ldb : B + (long)tj * NB;
    gemm_tile_128<AK, BK>(Ap, lda, Bp, ldb, K, acc, smem);
    gt_store<3>(C + (long)ti * NB * ldc + (long)tj * NB, ldc, acc, alpha, beta);
}

void launch_dbg_gemm(hipStream_t st, int a_mcontig, int b_ncontig, long M, long N, long K, const double* A,
                     const double* B, double* C, double alpha, double beta) {
    const int ntr = (int)(M / NB), ntc = (int)(N / NB);
    const dim3 grid((unsigned)(ntr * ntc)), block(256);
    const long lda = a_mcontig ? M : K, ldb = b_ncontig ? N : K;
    LDS_OPT_IN((k_gemm_full<true, true>));
    LDS_OPT_IN((k_gemm_full<true, false>));
    LDS_OPT_IN((k_gemm_full<false, false>));
    LDS_OPT_IN((k_gemm_full<false, true>));
    if (!a_mcontig && !b_ncontig)
        hipLaunchKernelGGL((k_gemm_full<true, true>), grid, block, GT_LDS_BYTES, st, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
    else if (!a_mcontig && b_ncontig)
        hipLaunchKernelGGL((k_gemm_full<true, false>), grid, block, GT_LDS_BYTES, st, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
    else if (a_mcontig && b_ncontig)
        hipLaunchKernelGGL((k_gemm_full<false, false>), grid, block, GT_LDS_BYTES, st, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
    else
        hipLaunchKernelGGL((k_gemm_full<false, true>), grid, block, GT_LDS_BYTES, st, A, lda, B, ldb, C, N, (int)K, ntc, alpha, beta);
}

void launch_gemm_tn_sq(hipStream_t st, const double* A, long lda, long K, double* C, long ldc, int nt, double alpha,
                       double beta) {
    LDS_OPT_IN((k_gemm_full<false, false>));
    hipLaunchKernelGGL((k_gemm_full<false, false>), dim3((unsigned)(nt * nt)), dim3(256), GT_LDS_BYTES, st, A, lda, A,
                       lda, C, ldc, (int)K, nt, alpha, beta);
}
