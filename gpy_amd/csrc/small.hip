// small.hip -- single-CU fp64 MFMA kernels on 128x128 diagonal blocks: the sequential part of the
// blocked Cholesky (dpotf2 + the triangular solves that LAPACK dpotrf/dtrtri do on diagonal blocks;
// GPy reaches them through GPy/util/linalg.py:58 and :217-227).
//
// All three kernels use "register chaining" of v_mfma_f64_16x16x4_f64: the accumulator register r of a
// 16x16 product (row (l>>4)+4r, col l&15) is exactly the B operand of k-slice r of the next product
// when the A operand is fetched with k = (l>>4)+4s, so chains like  Dinv * (P - L*Y)  never leave VGPRs.
#include "common.h"
#include "internal.h"

#define DS 130   // LDS row stride (doubles) of the 128x128 block image
#define IS 18    // LDS row stride of a 16x16 inverse tile

// ------------------------------------------------------------------------------------------------
// 16x16 Cholesky + inverse of the factor, entirely in the registers of ONE wave.
// Lane l holds row i = l&15, columns 4g..4g+3 (g = l>>4) of the tile in v[] and of the running
// right-hand side (identity -> L^-1) in x[].  Right-looking: step j finishes column j of L and row j of L^-1.
__device__ __forceinline__ int potf2_inv_16(double (&v)[4], double (&x)[4], int lane) {
    const int i = lane & 15, g = lane >> 4;
    int fail = 0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) x[cc] = (4 * g + cc == i) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int gj = j >> 2, rj = j & 3;
        const double colv = v[rj];
        double piv = __shfl(colv, j + 16 * gj);
        if (!(piv > 0.0)) {           // wave-uniform: not positive definite (or NaN)
            if (fail == 0) fail = j + 1;
            piv = 1.0;
        }
        const double d = sqrt(piv);
        const double rd = 1.0 / d;
        const double lij = __shfl(colv, i + 16 * gj) * rd;     // L[i][j] (meaningful for i > j)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = 4 * g + cc;
            const double lcj = __shfl(colv, c + 16 * gj) * rd;  // L[c][j]
            if (c > j && i >= c) v[cc] -= lij * lcj;
            const double xjc = __shfl(x[cc], j + 16 * g) * rd;  // (L^-1)[j][c]
            if (i > j) x[cc] -= lij * xjc;
            else if (i == j) x[cc] = xjc;
        }
        if (g == gj) v[rj] = (i > j) ? lij : ((i == j) ? d : v[rj]);
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
        if (4 * g + cc > i) v[cc] = 0.0;
    return fail;
}

// ------------------------------------------------------------------------------------------------
// In-place Cholesky of one 128x128 block held in LDS, 16 waves.  Per 16-column step:
//   wave 0: potf2 + inverse of the diagonal tile  |  all: P = A21 * Dinv^T (MFMA)  |  all: A22 -= P P^T (MFMA)
__global__ __launch_bounds__(1024) void k_diag128(double* __restrict__ A, long ld, long c0,
                                                  double* __restrict__ dinv, double* __restrict__ logsum,
                                                  int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* M = sm;                    // [128][DS]
    double* Dv = sm + 128 * DS;        // [8][16][IS]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double* Ab = A + c0 * ld + c0;
    for (int idx = t; idx < 128 * 128; idx += 1024) {
        const int r = idx >> 7, c = idx & 127;
        M[r * DS + c] = Ab[(long)r * ld + c];
    }
    __syncthreads();
    const int fi = lane & 15, fk = lane >> 4;      // MFMA operand coordinates of this lane
    for (int jb = 0; jb < 8; ++jb) {
        const int o = jb * 16;
        if (w == 0) {
            double v[4], x[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) v[cc] = M[(o + fi) * DS + o + 4 * fk + cc];
            const int fail = potf2_inv_16(v, x, lane);
            if (fail != 0 && lane == 0) atomicCAS(info, 0, (int)(c0 + o + fail));
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                M[(o + fi) * DS + o + 4 * fk + cc] = v[cc];
                Dv[jb * 16 * IS + fi * IS + 4 * fk + cc] = x[cc];
            }
        }
        __syncthreads();
        // panel: tile ib (rows below) <- tile * Dinv^T ; one tile per wave
        const int nbelow = 7 - jb;
        if (w < nbelow) {
            const int ro = (jb + 1 + w) * 16;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = M[(ro + fi) * DS + o + fk + 4 * s];
                const double b = Dv[jb * 16 * IS + fi * IS + fk + 4 * s];   // B[k][col] = Dinv[col][k]
                acc = mfma_f64(a, b, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(ro + fk + 4 * r) * DS + o + fi] = acc[r];
        }
        __syncthreads();
        // trailing update inside the block: C[ib,kb] -= P[ib] P[kb]^T for jb < kb <= ib
        const int ntile = nbelow * (nbelow + 1) / 2;
        for (int q = w; q < ntile; q += 16) {
            int a_ = 0, rem = q;
            while (rem > a_) { rem -= a_ + 1; ++a_; }   // q -> (a_, rem) with rem <= a_
            const int ro = (jb + 1 + a_) * 16, co = (jb + 1 + rem) * 16;
            d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = M[(ro + fk + 4 * r) * DS + co + fi];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -M[(ro + fi) * DS + o + fk + 4 * s];
                const double b = M[(co + fi) * DS + o + fk + 4 * s];
                acc = mfma_f64(a, b, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(ro + fk + 4 * r) * DS + co + fi] = acc[r];
        }
        __syncthreads();
    }
    // write back L (whole block; the strict upper part is never consumed on device), the tile inverses
    // and sum(log diag)
    for (int idx = t; idx < 128 * 128; idx += 1024) {
        const int r = idx >> 7, c = idx & 127;
        Ab[(long)r * ld + c] = M[r * DS + c];
    }
    for (int idx = t; idx < 8 * 256; idx += 1024) {
        const int jb = idx >> 8, r = (idx >> 4) & 15, c = idx & 15;
        dinv[idx] = Dv[jb * 16 * IS + r * IS + c];
    }
    if (w == 0) {
        double s = log(M[lane * DS + lane]) + log(M[(lane + 64) * DS + lane + 64]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) logsum[0] = s;
    }
}

void launch_diag128(hipStream_t st, double* A, long ld, long c0, double* dinv, double* logsum, int* info) {
    const size_t lds = (size_t)(128 * DS + 8 * 16 * IS) * sizeof(double);
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_diag128), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        opted = true;
    }
    hipLaunchKernelGGL(k_diag128, dim3(1), dim3(1024), lds, st, A, ld, c0, dinv, logsum, info);
}

// ------------------------------------------------------------------------------------------------
// Panel solve P <- P * L_cc^{-T} for 16 rows of P per wave (transposed: L_cc Y = P^T, Y chained in VGPRs).
__global__ __launch_bounds__(256) void k_trsm128(double* __restrict__ A, long ld, long c0, long r0, long mrows,
                                                 const double* __restrict__ dinv) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long prow0 = r0 + ((long)blockIdx.x * 4 + w) * 16;
    if (prow0 >= r0 + mrows) return;
    const int fi = lane & 15, fk = lane >> 4;
    double* P = A + (prow0 + fi) * ld + c0;                 // this lane's row of the panel
    const double* Lb = A + (c0 + fi) * ld + c0;             // L_cc, row fi of tile-row 0
    d4 Y[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = P[jb * 16 + fk + 4 * r];
#pragma unroll
        for (int k = 0; k < jb; ++k)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -Lb[(long)(jb * 16) * ld + k * 16 + fk + 4 * s];
                acc = mfma_f64(a, Y[k][s], acc);
            }
        d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double a = dinv[jb * 256 + fi * 16 + fk + 4 * s];
            y = mfma_f64(a, acc[s], y);
        }
        Y[jb] = y;
#pragma unroll
        for (int r = 0; r < 4; ++r) P[jb * 16 + fk + 4 * r] = y[r];
    }
}

void launch_trsm128(hipStream_t st, double* A, long ld, long c0, long r0, long mrows, const double* dinv) {
    if (mrows <= 0) return;
    const long nwaves = mrows / 16;
    hipLaunchKernelGGL(k_trsm128, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, A, ld, c0, r0, mrows, dinv);
}

// ------------------------------------------------------------------------------------------------
// X_cc = L_cc^{-1} for every 128x128 diagonal block; wave JB owns tile-column JB of the block.
template <int JB>
__device__ __forceinline__ void inv128_col(const double* __restrict__ Lb, double* __restrict__ Xb, long ld,
                                           const double* __restrict__ dv, int lane) {
    const int fi = lane & 15, fk = lane >> 4;
    d4 Xt[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) Xt[JB][r] = dv[JB * 256 + (fk + 4 * r) * 16 + fi];
#pragma unroll
    for (int r = 0; r < 4; ++r) Xb[(long)(JB * 16 + fk + 4 * r) * ld + JB * 16 + fi] = Xt[JB][r];
#pragma unroll
    for (int ib = 0; ib < JB; ++ib)      // tiles above the diagonal of the block: exact zeros
#pragma unroll
        for (int r = 0; r < 4; ++r) Xb[(long)(ib * 16 + fk + 4 * r) * ld + JB * 16 + fi] = 0.0;
#pragma unroll
    for (int ib = JB + 1; ib < 8; ++ib) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = JB; k < ib; ++k)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = Lb[(long)(ib * 16 + fi) * ld + k * 16 + fk + 4 * s];
                acc = mfma_f64(a, Xt[k][s], acc);
            }
        d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double a = -dv[ib * 256 + fi * 16 + fk + 4 * s];
            y = mfma_f64(a, acc[s], y);
        }
        Xt[ib] = y;
#pragma unroll
        for (int r = 0; r < 4; ++r) Xb[(long)(ib * 16 + fk + 4 * r) * ld + JB * 16 + fi] = y[r];
    }
}

__global__ __launch_bounds__(512) void k_inv128(const double* __restrict__ L, double* __restrict__ X, long ld,
                                                const double* __restrict__ dinv_all) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long c0 = (long)blockIdx.x * 128;
    const double* Lb = L + c0 * ld + c0;
    double* Xb = X + c0 * ld + c0;
    const double* dv = dinv_all + (long)blockIdx.x * 8 * 256;
    switch (w) {
        case 0: inv128_col<0>(Lb, Xb, ld, dv, lane); break;
        case 1: inv128_col<1>(Lb, Xb, ld, dv, lane); break;
        case 2: inv128_col<2>(Lb, Xb, ld, dv, lane); break;
        case 3: inv128_col<3>(Lb, Xb, ld, dv, lane); break;
        case 4: inv128_col<4>(Lb, Xb, ld, dv, lane); break;
        case 5: inv128_col<5>(Lb, Xb, ld, dv, lane); break;
        case 6: inv128_col<6>(Lb, Xb, ld, dv, lane); break;
        default: inv128_col<7>(Lb, Xb, ld, dv, lane); break;
    }
}

void launch_inv128(hipStream_t st, const double* L, double* X, long ld, int nblk, const double* dinv_all) {
    hipLaunchKernelGGL(k_inv128, dim3((unsigned)nblk), dim3(512), 0, st, L, X, ld, dinv_all);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_dbg_mfma(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = mfma_f64(a[l], b[l], acc);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

void launch_dbg_mfma(hipStream_t st, const double* a, const double* b, double* d) {
    hipLaunchKernelGGL(k_dbg_mfma, dim3(1), dim3(64), 0, st, a, b, d);
}
