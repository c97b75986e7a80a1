// chain_dev.h -- device code of the sequential part of the blocked Cholesky, shared by the single-CU kernels of small.hip
// (k_diag128 / k_trsm128: one launch per 128-column step) and by the single-launch dataflow factorisation of persist.hip
// (the same arithmetic, instruction for instruction, so both give the same bits).
//
// "Register chaining" of v_mfma_f64_16x16x4_f64: the accumulator register r of a 16x16 product (row (l>>4)+4r, col l&15)
// is exactly the B operand of k-slice r of the next product when the A operand is fetched with k = (l>>4)+4s, so chains
// like  Dinv * (P - L*Y)  never leave VGPRs.
//
// LDS image of a 128x128 lower-triangular block: its 36 lower 16x16 tiles, each [16][18] doubles (row stride 18 = 2*odd
// keeps the MFMA fragment read row*18+k bank-conflict free): 82,944 B.
#pragma once
#include "common.h"

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

// COH = true: agent-scope relaxed atomics = global_load / global_store ... sc1 (write-through, L1-bypassing): what a value
// that ANOTHER workgroup of the same launch reads or wrote must travel through (MI355X_MICROARCH.md, inter-workgroup
// visibility); COH = false: plain accesses (a kernel boundary orders them).
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
// The three phases of the in-place Cholesky of one 128x128 block (4 waves = 256 threads):
//   diag128_load   : the 36 lower tiles global -> LDS image Tt
//   diag128_factor : the factorisation proper, LDS -> LDS; two barriers per 16-column step:
//        wave 0: finish tile (jb,jb), factor + invert it (potf2_inv_16)   ||   waves 1..3: rest of step jb-1's update
//        barrier; all waves: P[ib] = A[ib,jb] * Dinv^T for the tiles below (MFMA); barrier
//     the inverse of diagonal tile jb goes to global dinv[jb*256 ..] and to the LDS tile Dv + jb*DVS (DVS = 0: one scratch
//     tile, DVS = TSZ: all eight kept, the layout the panel solve reads)
//   diag128_store  : L (lower tiles) LDS -> global, sum(log diag) -> logsum[0]
template <bool COHLD>
__device__ __forceinline__ void diag128_load(const double* __restrict__ Ab, long ld, double* Tt) {
    const int t = threadIdx.x, er = t >> 4, ec = t & 15;     // element (er, ec) of tile u for u = 0..35 (256 threads = one tile per trip)
    double v[NTILE];                                          // issue all 36 loads before the first LDS write
#pragma unroll
    for (int u = 0; u < NTILE; ++u) v[u] = ldg<COHLD>(Ab + (long)(tile_I(u) * 16 + er) * ld + tile_J(u) * 16 + ec);
#pragma unroll
    for (int u = 0; u < NTILE; ++u) Tt[u * TSZ + er * TS + ec] = v[u];
}

template <bool COH, int DVS>
__device__ __forceinline__ void diag128_factor(double* Tt, double* Dv0, long c0, double* __restrict__ dinv,
                                               int* __restrict__ info) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int fi = lane & 15, fk = lane >> 4;
    for (int jb = 0; jb < 8; ++jb) {
        double* Dv = Dv0 + jb * DVS;
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
}

template <bool COH>
__device__ __forceinline__ void diag128_store(double* __restrict__ Ab, long ld, const double* Tt, double* __restrict__ logsum) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, er = t >> 4, ec = t & 15;
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

// In-place Cholesky of one 128x128 block, 4 waves (one per SIMD, <= 280 VGPRs: co-resident with a GEMM workgroup)
template <bool COH, bool COHLD = false>
__device__ __forceinline__ void diag128_body(double* __restrict__ A, long ld, long c0, double* __restrict__ dinv,
                                             double* __restrict__ logsum, int* __restrict__ info, double* sm) {
    double* Tt = sm;                       // [36][16][18]
    double* Dv = sm + NTILE * TSZ;         // [16][18] inverse of the current diagonal tile
    double* Ab = A + c0 * ld + c0;
    diag128_load<COHLD>(Ab, ld, Tt);
    __syncthreads();
    diag128_factor<COH, 0>(Tt, Dv, c0, dinv, info);
    diag128_store<COH>(Ab, ld, Tt, logsum);
}

// ------------------------------------------------------------------------------------------------
// Panel solve P <- P * L_cc^{-T}, 16 panel rows per wave (transposed: L_cc Y = P^T with the 16-column strips of Y
// chained in VGPRs).  No LDS and < 128 VGPRs, so these waves slot in next to the trailing-update workgroups of the
// look-ahead schedule.  All panel loads are issued up front and all stores at the end: the L_cc / Dinv operand loads
// (L2-resident, shared by every wave) then carry no dependence on earlier steps and the compiler hoists them.
// (A variant with every operand read straight from L2 and one with two interleaved strips per wave were measured slower:
//  DESIGN.md 6e.)
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

// Y = P * L_cc^{-T} for one 16-row strip held in registers (Pin: tile jb, register r = P[row fk+4r... in the chained
// layout]).  Lt(jb, k): LDS tile image of L_cc[jb][k] (k < jb); Dt(jb): LDS tile image of the inverted diagonal tile jb.
template <class LT, class DT>
__device__ __forceinline__ void trsm_strip_core(const d4 (&Pin)[8], d4 (&Y)[8], LT Lt, DT Dt, int lane) {
    const int fi = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d4 acc0 = Pin[jb], acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < jb; ++k) {
            const double* Ls = Lt(jb, k) + fi * TS + fk;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -Ls[4 * s];
                if (k & 1) acc1 = mfma_f64(a, Y[k][s], acc1);
                else acc0 = mfma_f64(a, Y[k][s], acc0);
            }
        }
        const d4 acc = acc0 + acc1;
        d4 y = {0.0, 0.0, 0.0, 0.0};
        const double* Ds = Dt(jb) + fi * TS + fk;
#pragma unroll
        for (int s = 0; s < 4; ++s) y = mfma_f64(Ds[4 * s], acc[s], y);
        Y[jb] = y;
    }
}

// Two strips at once: the same operations per strip, in the same order (same bits), with the two dependence chains
// interleaved -- the L / Dinv fragments are read once for both and each MFMA has an independent neighbour to hide behind.
template <class LT, class DT>
__device__ __forceinline__ void trsm_strip_core2(const d4 (&P0)[8], const d4 (&P1)[8], d4 (&Y0)[8], d4 (&Y1)[8], LT Lt, DT Dt,
                                                 int lane) {
    const int fi = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d4 a0 = P0[jb], a1 = {0.0, 0.0, 0.0, 0.0}, b0 = P1[jb], b1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < jb; ++k) {
            const double* Ls = Lt(jb, k) + fi * TS + fk;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -Ls[4 * s];
                if (k & 1) {
                    a1 = mfma_f64(a, Y0[k][s], a1);
                    b1 = mfma_f64(a, Y1[k][s], b1);
                } else {
                    a0 = mfma_f64(a, Y0[k][s], a0);
                    b0 = mfma_f64(a, Y1[k][s], b0);
                }
            }
        }
        const d4 acca = a0 + a1, accb = b0 + b1;
        d4 ya = {0.0, 0.0, 0.0, 0.0}, yb = {0.0, 0.0, 0.0, 0.0};
        const double* Ds = Dt(jb) + fi * TS + fk;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double dd = Ds[4 * s];
            ya = mfma_f64(dd, acca[s], ya);
            yb = mfma_f64(dd, accb[s], yb);
        }
        Y0[jb] = ya;
        Y1[jb] = yb;
    }
}

// strips at rows prow0 and prow0 + 64 of the panel at column c0 together (trsm_stage_L's packing of L_cc)
template <bool COH>
__device__ __forceinline__ void trsm_strip2(double* __restrict__ A, long ld, long c0, long prow0, const double* sm, int lane) {
    const int fi = lane & 15, fk = lane >> 4;
    double* Pa = A + (prow0 + fi) * ld + c0;
    double* Pb = A + (prow0 + 64 + fi) * ld + c0;
    d4 P0[8], P1[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            P0[jb][r] = Pa[jb * 16 + fk + 4 * r];
            P1[jb][r] = Pb[jb * 16 + fk + 4 * r];
        }
    d4 Y0[8], Y1[8];
    trsm_strip_core2(P0, P1, Y0, Y1, [sm](int jb, int k) { return sm + tix_sl(jb, k) * TSZ; },
                     [sm](int jb) { return sm + (28 + jb) * TSZ; }, lane);
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            stg<COH>(Pa + jb * 16 + fk + 4 * r, Y0[jb][r]);
            stg<COH>(Pb + jb * 16 + fk + 4 * r, Y1[jb][r]);
        }
}

// the same, one strip after the other (half the live registers), results left in registers (the caller stores them)
__device__ __forceinline__ void trsm_strip2_regs(const double* __restrict__ A, long ld, long c0, long prow0, const double* sm,
                                                 int lane, d4 (&Y0)[8], d4 (&Y1)[8]) {
    const int fi = lane & 15, fk = lane >> 4;
    const double* Pa = A + (prow0 + fi) * ld + c0;
    const double* Pb = A + (prow0 + 64 + fi) * ld + c0;
    d4 P0[8], P1[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            P0[jb][r] = Pa[jb * 16 + fk + 4 * r];
            P1[jb][r] = Pb[jb * 16 + fk + 4 * r];
        }
    trsm_strip_core(P0, Y0, [sm](int jb, int k) { return sm + tix_sl(jb, k) * TSZ; }, [sm](int jb) { return sm + (28 + jb) * TSZ; },
                    lane);
    trsm_strip_core(P1, Y1, [sm](int jb, int k) { return sm + tix_sl(jb, k) * TSZ; }, [sm](int jb) { return sm + (28 + jb) * TSZ; },
                    lane);
}

// strip rows [prow0, prow0+16) of the panel at column c0, L_cc image = trsm_stage_L's packing
template <bool COH>
__device__ __forceinline__ void trsm_strip(double* __restrict__ A, long ld, long c0, long prow0, const double* sm,
                                           int lane) {
    const int fi = lane & 15, fk = lane >> 4;
    double* P = A + (prow0 + fi) * ld + c0;                 // this lane's row of the panel
    d4 Pin[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pin[jb][r] = P[jb * 16 + fk + 4 * r];
    d4 Y[8];
    trsm_strip_core(Pin, Y, [sm](int jb, int k) { return sm + tix_sl(jb, k) * TSZ; },
                    [sm](int jb) { return sm + (28 + jb) * TSZ; }, lane);
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg<COH>(P + jb * 16 + fk + 4 * r, Y[jb][r]);
}

