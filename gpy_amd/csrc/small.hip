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

#define PRIO_CHAIN 0        // s_setprio 3 in the chain kernels: no measurable effect (fp64 VALU shares the DP pipe with MFMA)
#define TS 18                 // row stride (doubles) inside a 16x16 tile
#define TSZ (16 * TS)         // doubles per tile image
#define NTILE 36              // lower tiles of a 128x128 block

__device__ __forceinline__ int tix(int I, int J) { return I * (I + 1) / 2 + J; }
// inverse of tix for compile-time tile numbers (loops over u are fully unrolled)
__device__ __forceinline__ constexpr int tile_I(int u) {
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= u) ++I;
    return I;
}
__device__ __forceinline__ constexpr int tile_J(int u) { return u - tile_I(u) * (tile_I(u) + 1) / 2; }

__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// Agent-scope coherent accesses (global_load/store ... sc1): data handed from one workgroup to another INSIDE a kernel
// (k_panel_fused) must not be served from / parked in the per-XCD L2 of the producer or the consumer.
template <bool COH> __device__ __forceinline__ double ldg(const double* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ void stg(double* p, double v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// value of lane (row*16 + j) for every lane of each 16-lane row: one v_mov_b64_dpp row_newbcast (no SGPR round trip,
// no readlane->VALU hazard nops).  j is a compile-time constant after unrolling; the switch folds.
#define BC16_CASE(J) case J: return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true);
__device__ __forceinline__ double bcast16(double v, int j) {
    switch (j) {
        BC16_CASE(0) BC16_CASE(1) BC16_CASE(2) BC16_CASE(3) BC16_CASE(4) BC16_CASE(5) BC16_CASE(6) BC16_CASE(7)
        BC16_CASE(8) BC16_CASE(9) BC16_CASE(10) BC16_CASE(11) BC16_CASE(12) BC16_CASE(13) BC16_CASE(14)
        default: return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + 15, 0xf, 0xf, true);
    }
}

// ------------------------------------------------------------------------------------------------
// 16x16 Cholesky + inverse of the factor in the registers of one wave.  Lane l works on row i = l & 15.
// a[c] = A[i][c] is mirrored in the four 16-lane rows of the wave (the factorisation itself is replicated);
// the running right-hand side I -> L^-1 is SPLIT over them: lane group g = l >> 4 keeps columns 4k+g in xs[k],
// so the inverse costs a quarter of the instructions.  Right-looking; every cross-row value is a DPP row
// broadcast.  Returns 0 or the 1-based index of the first non-positive pivot.
__device__ __forceinline__ int potf2_inv_16(double (&a)[16], double (&xs)[4], int lane) {
    const int i = lane & 15, g = lane >> 4;
    int fail = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[k] = (4 * k + g == i) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double piv = readlane_d(a[j], j);
        if (!(piv > 0.0)) {                      // wave-uniform: not positive definite (or NaN)
            if (fail == 0) fail = j + 1;
            piv = 1.0;
        }
        const double rd = rsqrt(piv);            // one dependent chain (v_rsq_f64 + refinement) instead of sqrt + div
        a[j] *= rd;                              // row j: piv*rd = L[j][j]; rows below: L[i][j]; rows above: don't care
        const double lm = (i > j) ? a[j] : 0.0;  // L[i][j] on the rows that still change, 0 elsewhere
        const double sc = (i == j) ? rd : 1.0;   // row j of the right-hand side becomes row j of L^-1
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = fma(-lm, bcast16(a[j], c), a[c]);
        // columns 4k+g <= j change; a column 4k+g > j in the last slot has a zero in row j, so it passes through unchanged
#pragma unroll
        for (int k = 0; k <= (j >> 2); ++k) {
            xs[k] *= sc;
            xs[k] = fma(-lm, bcast16(xs[k], j), xs[k]);
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c > i) a[c] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (4 * k + g > i) xs[k] = 0.0;
    }
    return fail;
}

// C tile (D layout) -= P[ib] * P[kb]^T with both panels in column jp of the packed block
__device__ __forceinline__ void diag_update_tile(double* Tt, int ib, int kb, int jp, int fi, int fk) {
    double* C = Tt + tix(ib, kb) * TSZ;
    const double* Pa = Tt + tix(ib, jp) * TSZ;
    const double* Pb = Tt + tix(kb, jp) * TSZ;
    d4 acc;
    double af[4], bf[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = C[(fk + 4 * r) * TS + fi];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        af[s] = -Pa[fi * TS + fk + 4 * s];
        bf[s] = Pb[fi * TS + fk + 4 * s];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma_f64(af[s], bf[s], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(fk + 4 * r) * TS + fi] = acc[r];
}

// ------------------------------------------------------------------------------------------------
// In-place Cholesky of one 128x128 block, 4 waves (one per SIMD, <= 280 VGPRs: co-resident with a GEMM workgroup),
// two barriers per 16-column step:
//   wave 0: finish tile (jb,jb), factor + invert it (potf2_inv_16)   ||   waves 1..3: rest of step jb-1's update
//   barrier; all waves: P[ib] = A[ib,jb] * Dinv^T for the tiles below (MFMA); barrier
template <bool COH, bool COHLD = false>
__device__ __forceinline__ void diag128_body(double* __restrict__ A, long ld, long c0, double* __restrict__ dinv,
                                             double* __restrict__ logsum, int* __restrict__ info, double* sm) {
    double* Tt = sm;                       // [36][16][18]
    double* Dv = sm + NTILE * TSZ;         // [16][18] inverse of the current diagonal tile
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double* Ab = A + c0 * ld + c0;
    const int er = t >> 4, ec = t & 15;     // element (er, ec) of tile u for u = 0..35 (256 threads = one tile per trip)
    {   // issue all 36 loads before the first LDS write
        double v[NTILE];
#pragma unroll
        for (int u = 0; u < NTILE; ++u) v[u] = ldg<COHLD>(Ab + (long)(tile_I(u) * 16 + er) * ld + tile_J(u) * 16 + ec);
#pragma unroll
        for (int u = 0; u < NTILE; ++u) Tt[u * TSZ + er * TS + ec] = v[u];
    }
    __syncthreads();
    const int fi = lane & 15, fk = lane >> 4;
    for (int jb = 0; jb < 8; ++jb) {
        if (w == 0) {
            if (jb > 0) {
                diag_update_tile(Tt, jb, jb, jb - 1, fi, fk);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            double* Td = Tt + tix(jb, jb) * TSZ;
            double a[16], xs[4];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = Td[fi * TS + c];
            const int fail = potf2_inv_16(a, xs, lane);
            if (fail != 0 && lane == 0) atomicCAS(info, 0, (int)(c0 + jb * 16 + fail));
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) Td[fi * TS + c] = a[c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {              // lane group fk holds columns 4k+fk of the inverse
                Dv[fi * TS + 4 * k + fk] = xs[k];
                stg<COH>(dinv + jb * 256 + fi * 16 + 4 * k + fk, xs[k]);
            }
        } else if (jb > 0) {
            // remaining tiles of step jb-1's trailing update: (ib,kb), jb <= kb <= ib <= 7, except (jb,jb)
            const int m = 8 - jb, ntile = m * (m + 1) / 2;
            for (int q = w; q < ntile; q += 3) {        // q = 0 is (jb,jb): skipped (wave 0 did it)
                int a_ = 0, rem = q;
                while (rem > a_) { rem -= a_ + 1; ++a_; }
                diag_update_tile(Tt, jb + a_, jb + rem, jb - 1, fi, fk);
            }
        }
        __syncthreads();
        const int nbelow = 7 - jb;
        for (int pt = w; pt < nbelow; pt += 4) {   // panel tile ib = jb+1+pt:  P = A[ib,jb] * Dinv^T
            double* T = Tt + tix(jb + 1 + pt, jb) * TSZ;
            double af[4], bf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                af[s] = T[fi * TS + fk + 4 * s];
                bf[s] = Dv[fi * TS + fk + 4 * s];          // B[k][col] = Dinv[col][k]
            }
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_f64(af[s], bf[s], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(fk + 4 * r) * TS + fi] = acc[r];
        }
        __syncthreads();
    }
    // write back the lower tiles of L and sum(log diag)
#pragma unroll
    for (int u = 0; u < NTILE; ++u)
        stg<COH>(Ab + (long)(tile_I(u) * 16 + er) * ld + tile_J(u) * 16 + ec, Tt[u * TSZ + er * TS + ec]);
    if (w == 0) {
        const int i0 = lane, i1 = lane + 64;
        double s = log(Tt[tix(i0 >> 4, i0 >> 4) * TSZ + (i0 & 15) * TS + (i0 & 15)]) +
                   log(Tt[tix(i1 >> 4, i1 >> 4) * TSZ + (i1 & 15) * TS + (i1 & 15)]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) logsum[0] = s;
    }
}

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

// ------------------------------------------------------------------------------------------------
// Panel solve P <- P * L_cc^{-T}, 16 panel rows per wave (transposed: L_cc Y = P^T with the 16-column strips of Y
// chained in VGPRs).  No LDS and < 128 VGPRs, so these waves slot in next to the trailing-update workgroups of the
// look-ahead schedule.  All panel loads are issued up front and all stores at the end: the L_cc / Dinv operand loads
// (L2-resident, shared by every wave) then carry no dependence on earlier steps and the compiler hoists them.
// LDS image for the panel solves: the 28 strictly-lower 16x16 tiles of L_cc and the 8 inverted diagonal tiles, each
// [16][18] (same conflict-free fragment layout as k_diag128): 82,944 B.  Every MFMA operand of the 16-row strip chains
// then comes from LDS instead of a dependent L2 round trip (a strip went from ~25 us to a few us).
#define TRSM_LDS_BYTES (NTILE * TSZ * 8)
__device__ __forceinline__ int tix_sl(int I, int J) { return I * (I - 1) / 2 + J; }     // I > J
__device__ __forceinline__ constexpr int sl_I(int u) {
    int I = 1;
    while ((I + 1) * I / 2 <= u) ++I;
    return I;
}
__device__ __forceinline__ constexpr int sl_J(int u) { return u - sl_I(u) * (sl_I(u) - 1) / 2; }

// all 256 threads of the workgroup; the caller synchronises
__device__ __forceinline__ void trsm_stage_L(const double* __restrict__ A, long ld, long c0,
                                             const double* __restrict__ dinv, double* sm) {
    const int t = threadIdx.x, er = t >> 4, ec = t & 15;
    const double* Ab = A + c0 * ld + c0;
    double v[NTILE];
#pragma unroll
    for (int u = 0; u < 28; ++u) v[u] = Ab[(long)(sl_I(u) * 16 + er) * ld + sl_J(u) * 16 + ec];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[28 + u] = dinv[u * 256 + t];
#pragma unroll
    for (int u = 0; u < NTILE; ++u) sm[u * TSZ + er * TS + ec] = v[u];
}

template <bool COH>
__device__ __forceinline__ void trsm_strip(double* __restrict__ A, long ld, long c0, long prow0, const double* sm,
                                           int lane) {
    const int fi = lane & 15, fk = lane >> 4;
    double* P = A + (prow0 + fi) * ld + c0;                 // this lane's row of the panel
    const double* Ls = sm + fi * TS + fk;                   // L_cc tiles: row fi, this lane's k offset
    const double* Ds = sm + 28 * TSZ + fi * TS + fk;        // inverted diagonal tiles
    d4 Pin[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pin[jb][r] = P[jb * 16 + fk + 4 * r];
    d4 Y[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d4 acc0 = Pin[jb], acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < jb; ++k) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -Ls[tix_sl(jb, k) * TSZ + 4 * s];
                if (k & 1) acc1 = mfma_f64(a, Y[k][s], acc1);
                else acc0 = mfma_f64(a, Y[k][s], acc0);
            }
        }
        const d4 acc = acc0 + acc1;
        d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) y = mfma_f64(Ds[jb * TSZ + 4 * s], acc[s], y);
        Y[jb] = y;
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg<COH>(P + jb * 16 + fk + 4 * r, Y[jb][r]);
}

// the same strip with every operand loaded straight from global memory / L2 (no LDS: these waves fit next to any
// resident workgroup)
__device__ __forceinline__ void trsm_strip_g(double* __restrict__ A, long ld, long c0, long prow0,
                                             const double* __restrict__ dinv, int lane) {
    const int fi = lane & 15, fk = lane >> 4;
    double* P = A + (prow0 + fi) * ld + c0;
    const double* Lb = A + (c0 + fi) * ld + c0 + fk;
    const double* Db = dinv + fi * 16 + fk;
    d4 Pin[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pin[jb][r] = P[jb * 16 + fk + 4 * r];
    d4 Y[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d4 acc0 = Pin[jb], acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < jb; ++k) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -Lb[(long)(jb * 16) * ld + k * 16 + 4 * s];
                if (k & 1) acc1 = mfma_f64(a, Y[k][s], acc1);
                else acc0 = mfma_f64(a, Y[k][s], acc0);
            }
        }
        const d4 acc = acc0 + acc1;
        d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) y = mfma_f64(Db[jb * 256 + 4 * s], acc[s], y);
        Y[jb] = y;
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[jb * 16 + fk + 4 * r] = Y[jb][r];
}

__global__ __launch_bounds__(256) void k_trsm128_g(double* __restrict__ A, long ld, long c0, long r0, long mrows,
                                                   const double* __restrict__ dinv) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long prow0 = r0 + ((long)blockIdx.x * 4 + w) * 16;
    if (prow0 >= r0 + mrows) return;
    trsm_strip_g(A, ld, c0, prow0, dinv, lane);
}

// Two strips per wave, interleaved: the two dependent MFMA chains fill each other's latency and share every L / Dinv
// fragment read, so a workgroup covers 128 panel rows in about the time it needs for 64 -- half as many workgroups hold
// a CU slot (and keep a 74 KB trailing-update workgroup out of it) while the chain runs under a busy GPU.
__device__ __forceinline__ void trsm_strip2(double* __restrict__ A, long ld, long c0, long prow0, long prow1, const double* sm,
                                            int lane) {
    const int fi = lane & 15, fk = lane >> 4;
    double* P0 = A + (prow0 + fi) * ld + c0;
    double* P1 = A + (prow1 + fi) * ld + c0;
    const double* Ls = sm + fi * TS + fk;
    const double* Ds = sm + 28 * TSZ + fi * TS + fk;
    d4 Y0[8], Y1[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d4 a0, a1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a0[r] = P0[jb * 16 + fk + 4 * r];
            a1[r] = P1[jb * 16 + fk + 4 * r];
        }
#pragma unroll
        for (int k = 0; k < jb; ++k) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -Ls[tix_sl(jb, k) * TSZ + 4 * s];
                a0 = mfma_f64(a, Y0[k][s], a0);
                a1 = mfma_f64(a, Y1[k][s], a1);
            }
        }
        d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double d = Ds[jb * TSZ + 4 * s];
            y0 = mfma_f64(d, a0[s], y0);
            y1 = mfma_f64(d, a1[s], y1);
        }
        Y0[jb] = y0;
        Y1[jb] = y1;
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            P0[jb * 16 + fk + 4 * r] = Y0[jb][r];
            P1[jb * 16 + fk + 4 * r] = Y1[jb][r];
        }
}

__global__ __launch_bounds__(256) void k_trsm128x2(double* __restrict__ A, long ld, long c0, long r0, long mrows,
                                                   const double* __restrict__ dinv) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    trsm_stage_L(A, ld, c0, dinv, sm);
    __syncthreads();
    const long base = r0 + (long)blockIdx.x * 128 + w * 16;          // strips w and w + 4 of this workgroup's 128 rows
    const long end = r0 + mrows;
    if (base >= end) return;
    if (base + 64 < end) trsm_strip2(A, ld, c0, base, base + 64, sm, lane);
    else trsm_strip<false>(A, ld, c0, base, sm, lane);
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

void launch_trsm128(hipStream_t st, double* A, long ld, long c0, long r0, long mrows, const double* dinv, int lds) {
    if (mrows <= 0) return;
    if (!lds) {
        hipLaunchKernelGGL(k_trsm128_g, dim3((unsigned)((mrows / 16 + 3) / 4)), dim3(256), 0, st, A, ld, c0, r0, mrows, dinv);
        return;
    }
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm128), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  TRSM_LDS_BYTES);
        opted = true;
    }
    const long nwaves = mrows / 16;
    if (lds == 2) {
        static bool opted2 = false;
        if (!opted2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm128x2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      TRSM_LDS_BYTES);
            opted2 = true;
        }
        hipLaunchKernelGGL(k_trsm128x2, dim3((unsigned)((mrows + 127) / 128)), dim3(256), TRSM_LDS_BYTES, st, A, ld, c0, r0,
                           mrows, dinv);
        return;
    }
    hipLaunchKernelGGL(k_trsm128, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), TRSM_LDS_BYTES, st, A, ld, c0, r0, mrows,
                       dinv);
}

// ------------------------------------------------------------------------------------------------
// One launch per outer panel (columns [c0, c0 + 128 ns), rows [c0, c0 + 128 nrb)): the whole
//   ns x ( diag128 -> trsm of the rows below -> rank-128 update of the panel's remaining columns )
// sequence with workgroup-to-workgroup hand-offs through flags in global memory instead of kernel boundaries.
// Workgroup g owns the 128-row blocks g, g + G, ... (G = gridDim.x >= ns: block j < ns, a diagonal block, is the
// first block of workgroup j).  Step k:
//   workgroup k        factors tile (k,k) (diag128_body), publishes L_kk and its 16x16 inverses, raises F[k];
//   every workgroup    waits for F[k], solves its blocks b > k against L_kk (trsm_strip), raises T[k][b] for b < ns;
//   every workgroup    for j = k+1 .. ns-1 waits for T[k][j] and updates its tiles (b, j), b >= j (one 128^3 tile GEMM).
// Workgroup k+1 only has tile (k+1,k+1) to update, so it reaches the next factorisation while the others still work:
// the dependency chain of a panel is ns x (diag + one strip solve + one tile GEMM) instead of ns x three machine-wide
// launches, and the workgroups keep their CU slots for the whole panel while a trailing update fills the GPU.
// Hand-off data (L_kk, dinv, solved blocks) is written with sc1 (write-through) stores and the flags are sc1 atomics: no
// L2-wide write-back / invalidate is needed.  The consumers read with plain loads: no cache of a consumer can hold
// a stale copy of those lines, because nothing but their producer touches them between the start of the kernel (caches
// invalidated by the dispatch) and the flag -- sc1 loads in the dependent MFMA chains cost a memory round trip each.  Flags carry the launch generation (never reset).  Every wait is bounded: on a
// timeout info[1] is raised, every workgroup stops waiting and the host reports the failure.
#define PF_SPIN_LIMIT (1 << 21)
__device__ __forceinline__ void wg_sync_mem() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void pf_raise(int* flag, int gen) {
    wg_sync_mem();                                      // every thread's stores have been acknowledged
    if (threadIdx.x == 0) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pf_wait(int* flag, int gen, int* info) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 255) == 0) {
                if (spins > PF_SPIN_LIMIT) __hip_atomic_store(info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            }
        }
    }
    wg_sync_mem();
}

__global__ __launch_bounds__(256, 2) void k_panel_fused(double* __restrict__ A, long ld, long c0, int ns, int nrb,
                                                        double* __restrict__ dinv, double* __restrict__ logsum,
                                                        int* __restrict__ info, int* __restrict__ flags, int gen,
                                                        long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int g = blockIdx.x, G = gridDim.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // diagnostics (MI355GP_PANEL_DBG=1): 100 MHz timestamps [workgroup][step][0 start, 1 F ready, 2 solved, 3 updated]
#define PF_STAMP(k, i) if (dbg && threadIdx.x == 0) dbg[((long)g * 4 + (k)) * 4 + (i)] = (long long)wall_clock64()
    int* F = flags;                 // F[k]
    int* T = flags + 4;             // T[k * 4 + j]
    for (int k = 0; k < ns; ++k) {
        const long ck = c0 + (long)k * NB;
        double* dv = dinv + (long)k * 8 * 256;
        PF_STAMP(k, 0);
        if (g == k) {
            diag128_body<true>(A, ld, ck, dv, logsum + k, info, sm);
            pf_raise(F + k, gen);
        } else {
            pf_wait(F + k, gen, info);
        }
        PF_STAMP(k, 1);
        bool staged = false;
        for (int b = g; b < nrb; b += G) {
            if (b <= k) continue;
            if (!staged) {              // L_kk and its inverted diagonal tiles -> LDS, once per step
                trsm_stage_L(A, ld, ck, dv, sm);
                __syncthreads();
                staged = true;
            }
            const long r0 = c0 + (long)b * NB;
            trsm_strip<true>(A, ld, ck, r0 + 16 * w, sm, lane);
            trsm_strip<true>(A, ld, ck, r0 + 16 * (w + 4), sm, lane);
            if (b < ns) pf_raise(T + k * 4 + b, gen);
        }
        wg_sync_mem();              // the solved blocks of this workgroup are operands of its own tile GEMMs
        PF_STAMP(k, 2);
        for (int j = k + 1; j < ns; ++j) {
            bool any = false;
            for (int b = g; b < nrb; b += G) any = any || (b >= j);
            if (!any) continue;
            if (g != j) pf_wait(T + k * 4 + j, gen, info);
            const double* Pj = A + (c0 + (long)j * NB) * ld + ck;
            for (int b = g; b < nrb; b += G) {
                if (b < j) continue;
                double* Ct = A + (c0 + (long)b * NB) * ld + c0 + (long)j * NB;
                d4 acc[4][GTCfg<4>::NI];
                gt_load_buf<4>(Ct, ld, acc);
                gemm_tile_128<true, true, 4, true>(A + (c0 + (long)b * NB) * ld + ck, ld, Pj, ld, NB, acc, sm);
                gt_store<0, 4>(Ct, ld, acc);
                __syncthreads();    // the LDS stages are reused by the next tile / the next factorisation
            }
        }
        wg_sync_mem();              // tile (k+1,k+1) and the strips of step k+1 were written by other waves of this workgroup
        PF_STAMP(k, 3);
    }
    // a timed-out hand-off turns into a negative info[0]: every host entry point reports it as an error
    if (threadIdx.x == 0 && __hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        __hip_atomic_store(info, -7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// Diagonal-block server: ONE workgroup that stays resident for a whole factorisation on a CU of its own (its LDS request
// leaves no room for a tile-GEMM workgroup next to it) and factors the 128x128 diagonal blocks one after the other as the
// panel chain asks for them.  k_diag128 launched into a GPU that a trailing update has filled shares its CU -- and the
// fp64 pipe of its SIMD -- with MFMA-saturating waves and takes 100-250 us instead of 30; the server always runs alone.
//   chain stream:  ... -> k_diag_call(blk): raise ready[blk], spin until done[blk]  -> trsm -> update -> ...
//   server      :  for blk: wait ready[blk]; factor (tile read with sc1 loads, results written with sc1 stores); raise done[blk]
// The producers of the tile and the consumers of L_kk / dinv are ordinary kernels before / after k_diag_call on the
// chain stream: kernel boundaries make their side coherent.  All waits are bounded (info[1] = abort).
#define DIAG_SERVER_LDS_BYTES (92 * 1024)
__global__ __launch_bounds__(256) void k_diag_server(double* __restrict__ A, long ld, int nblk, double* __restrict__ dinv,
                                                     double* __restrict__ logsum, int* __restrict__ info,
                                                     int* __restrict__ ready, int* __restrict__ done, int gen) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    for (int blk = 0; blk < nblk; ++blk) {
        pf_wait(ready + blk, gen, info);
        if (__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // uniform: same word
        diag128_body<true, true>(A, ld, (long)blk * NB, dinv + (long)blk * 8 * 256, logsum + blk, info, sm);
        pf_raise(done + blk, gen);
    }
    if (threadIdx.x == 0 && __hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        __hip_atomic_store(info, -7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void k_diag_call(int* __restrict__ ready, int* __restrict__ done, int blk, int gen,
                                                  int* __restrict__ info) {
    if (threadIdx.x != 0) return;
    __hip_atomic_store(ready + blk, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(done + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 255) == 0) {
            if (spins > PF_SPIN_LIMIT) __hip_atomic_store(info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        }
    }
}

void launch_diag_server(hipStream_t st, double* A, long ld, int nblk, double* dinv, double* logsum, int* info, int* ready,
                        int* done, int gen) {
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_diag_server), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  DIAG_SERVER_LDS_BYTES);
        opted = true;
    }
    hipLaunchKernelGGL(k_diag_server, dim3(1), dim3(256), DIAG_SERVER_LDS_BYTES, st, A, ld, nblk, dinv, logsum, info, ready,
                       done, gen);
}
void launch_diag_call(hipStream_t st, int* ready, int* done, int blk, int gen, int* info) {
    hipLaunchKernelGGL(k_diag_call, dim3(1), dim3(64), 0, st, ready, done, blk, gen, info);
}

int launch_panel_fused(hipStream_t st, double* A, long ld, long c0, int ns, int nrb, double* dinv, double* logsum,
                       int* info, int* flags, int gen, int max_wgs, long long* dbg) {
    static bool opted = false;
    const int lds = DIAG_LDS_BYTES > GT_LDS_BYTES ? DIAG_LDS_BYTES : GT_LDS_BYTES;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_panel_fused), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        opted = true;
    }
    const int G = nrb < max_wgs ? nrb : max_wgs;
    if (G < ns) return -1;
    hipLaunchKernelGGL(k_panel_fused, dim3((unsigned)G), dim3(256), lds, st, A, ld, c0, ns, nrb, dinv, logsum, info, flags, gen, dbg);
    return 0;
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
