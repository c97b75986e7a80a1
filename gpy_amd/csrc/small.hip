// small.hip -- single-CU fp64 MFMA kernels on 128x128 diagonal blocks: the sequential part of the
// blocked Cholesky (dpotf2 + the triangular solves that LAPACK dpotrf/dtrtri do on diagonal blocks;
// GPy reaches them through GPy/util/linalg.py:58 and :217-227).
//
// All three kernels use "register chaining" of v_mfma_f64_16x16x4_f64: the accumulator register r of a
// 16x16 product (row (l>>4)+4r, col l&15) is exactly the B operand of k-slice r of the next product
// when the A operand is fetched with k = (l>>4)+4s, so chains like  Dinv * (P - L*Y)  never leave VGPRs.
//
// LDS image of a 128x128 lower-triangular block: its 36 lower 16x16 tiles, each [16][18] doubles
// (row stride 18 = 2*odd keeps the MFMA fragment read row*18+k bank-conflict free): 82,944 B, so the
// kernel fits on a CU next to one 72 KB GEMM workgroup (look-ahead overlap with the trailing update).
#include "common.h"
#include "internal.h"
#include "gemm_tile.h"

#include "chain_dev.h"

__global__ __launch_bounds__(256) void k_diag128(double* __restrict__ A, long ld, long c0,
                                                  double* __restrict__ dinv, double* __restrict__ logsum,
                                                  int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (PRIO_CHAIN) __builtin_amdgcn_s_setprio(3);   // win instruction arbitration against co-resident update waves
    diag128_body<false>(A, ld, c0, dinv, logsum, info, sm);
}



#define DIAG_LDS_BYTES ((NTILE * TSZ + TSZ) * 8)

// exclusive != 0: request DIAG_EXCL_LDS_BYTES of LDS, more than a CU has left next to ONE 69.6 KB tile-GEMM workgroup, so
// the block only starts on a CU that runs no trailing-update workgroup (the CUs the update stream's CU mask leaves out):
// its dependent fp64 VALU chain then never queues behind a neighbour's fp64 MFMAs.
#define DIAG_EXCL_LDS_BYTES (96 * 1024)
void launch_diag128(hipStream_t st, double* A, long ld, long c0, double* dinv, double* logsum, int* info, int exclusive) {
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_diag128), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  DIAG_EXCL_LDS_BYTES);
        opted = true;
    }
    hipLaunchKernelGGL(k_diag128, dim3(1), dim3(256), exclusive ? DIAG_EXCL_LDS_BYTES : DIAG_LDS_BYTES, st, A, ld, c0, dinv,
                       logsum, info);
}

__global__ __launch_bounds__(256) void k_trsm128(double* __restrict__ A, long ld, long c0, long r0, long mrows,
                                                 const double* __restrict__ dinv) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (PRIO_CHAIN) __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    trsm_stage_L(A, ld, c0, dinv, sm);
    __syncthreads();
    const long prow0 = r0 + ((long)blockIdx.x * 4 + w) * 16;
    if (prow0 >= r0 + mrows) return;
    trsm_strip<false>(A, ld, c0, prow0, sm, lane);
}

void launch_trsm128(hipStream_t st, double* A, long ld, long c0, long r0, long mrows, const double* dinv) {
    if (mrows <= 0) return;
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm128), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  TRSM_LDS_BYTES);
        opted = true;
    }
    const long nwaves = mrows / 16;
    hipLaunchKernelGGL(k_trsm128, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), TRSM_LDS_BYTES, st, A, ld, c0, r0, mrows,
                       dinv);
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
