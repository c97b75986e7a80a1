// factor.hip -- blocked drivers: the dpotrf / dtrtri / dlauum equivalents GPy reaches through
// GPy/util/linalg.py:56-75 (jitchol -> lapack.dpotrf) and :127-145,193-227 (pdinv: dtrtri + dpotri).
//
// Layout: one padded (npad x npad, npad % 128 == 0) row-major fp64 buffer per matrix, lower triangle
// significant; padding rows/cols carry the identity, which factorises and inverts to itself.
#include "internal.h"

int factor_ws_alloc(FactorWs* ws, long npad) {
    ws->nblk = npad / NB;
    HIP_CHECK(hipMalloc(&ws->dinv, sizeof(double) * ws->nblk * 8 * 256));
    HIP_CHECK(hipMalloc(&ws->logsum, sizeof(double) * ws->nblk));
    HIP_CHECK(hipMalloc(&ws->info, sizeof(int) * 4));
    return 0;
}

void factor_ws_free(FactorWs* ws) {
    if (ws->dinv) (void)hipFree(ws->dinv);
    if (ws->logsum) (void)hipFree(ws->logsum);
    if (ws->info) (void)hipFree(ws->info);
    ws->dinv = ws->logsum = nullptr;
    ws->info = nullptr;
}

// Two-level right-looking Cholesky.  Outer panels of NBO = 512 columns keep the big trailing update at
// K = 512 (64 flop per byte of C traffic); inside a panel, 128-column steps:
//   diag128 (one CU) -> trsm128 on the rows below -> rank-128 update of the rest of the outer panel.
void potrf_device(hipStream_t st, double* A, long npad, FactorWs* ws) {
    const long ld = npad;
    (void)hipMemsetAsync(ws->info, 0, sizeof(int) * 4, st);
    for (long K0 = 0; K0 < npad; K0 += NBO) {
        const long W = (npad - K0 < NBO) ? (npad - K0) : NBO;
        for (long j = 0; j < W; j += NB) {
            const long c = K0 + j;
            const long blk = c / NB;
            launch_diag128(st, A, ld, c, ws->dinv + blk * 8 * 256, ws->logsum + blk, ws->info);
            const long below = npad - (c + NB);
            if (below <= 0) continue;
            launch_trsm128(st, A, ld, c, c + NB, below, ws->dinv + blk * 8 * 256);
            // inner update: rows [c+NB, npad) x cols [c+NB, K0+W)
            const long ncols = K0 + W - (c + NB);
            if (ncols > 0) {
                double* C = A + (c + NB) * ld + (c + NB);
                const double* P = A + (c + NB) * ld + c;
                launch_update_nt(st, C, ld, P, ld, P, ld, NB, (int)(below / NB), (int)(ncols / NB),
                                 (int)((c + NB) / NB), (int)((c + NB) / NB));
            }
        }
        const long rest = npad - (K0 + W);
        if (rest > 0) {
            double* C = A + (K0 + W) * ld + (K0 + W);
            const double* P = A + (K0 + W) * ld + K0;
            launch_update_nt(st, C, ld, P, ld, P, ld, (int)W, (int)(rest / NB), (int)(rest / NB),
                             (int)((K0 + W) / NB), (int)((K0 + W) / NB));
        }
    }
}

// X = L^-1: diagonal 128-blocks on single CUs (all blocks concurrently), then log2(nt) batched levels.
void trtri_device(hipStream_t st, const double* L, double* X, double* T, long npad, FactorWs* ws) {
    const int nt = (int)(npad / NB);
    launch_inv128(st, L, X, npad, nt, ws->dinv);
    for (int level = 0; (1 << level) < nt; ++level) launch_trtri_level(st, L, X, T, npad, nt, level);
}

void lauum_device(hipStream_t st, const double* X, double* W, long npad) {
    launch_lauum(st, X, W, npad, (int)(npad / NB));
}
