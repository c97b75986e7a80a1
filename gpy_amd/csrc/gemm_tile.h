// gemm_tile.h -- the fp64 MFMA workhorse: one 128x128 output tile per 256-thread workgroup.
//
// Geometry (MI355X / gfx950): 4 waves in a 2x2 arrangement, each wave owns a 64x64 sub-tile =
// 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 accumulator VGPRs).  K is consumed in slabs of 16
// (four MFMA k-slices); slabs are double-buffered in LDS (2 x (A 18 KB + B 18 KB) = 72 KB, two
// workgroups per CU) and the next slab's global loads are issued before the current slab's MFMAs.
//
// Operand storage (row-major buffers with leading dimension ld):
//   k-contiguous  : element (i, k) at P[i*ld + k]   -> LDS image [128][18]  (stride 18 = 2*odd: the
//                   fragment read row*18+k is bank-conflict free for ds_read_b64)
//   m/n-contiguous: element (k, i) at P[k*ld + i]   -> LDS image [16][144] (stride 144 = 16 mod 32)
// NT = (A k-contig, B k-contig), NN = (A k-contig, B n-contig), TN = (A m-contig, B n-contig).
#pragma once
#include "common.h"

#define GT_BK 16
#define GT_SKC 18
#define GT_SMN 144
#define GT_TILE 2304                      // doubles per staged operand slab (128*18 == 16*144)
#define GT_LDS_BYTES (4 * GT_TILE * 8)    // 73,728 B

template <bool KC>
__device__ __forceinline__ void gt_g2r(const double* __restrict__ P, long ld, int k0, d2 (&r)[4], int t) {
    if (KC) {
        const int row = t >> 3, kp = (t & 7) * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            r[it] = *reinterpret_cast<const d2*>(P + (long)(row + 32 * it) * ld + k0 + kp);
    } else {
        const int k = t >> 6, cp = (t & 63) * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            r[it] = *reinterpret_cast<const d2*>(P + (long)(k0 + k + 4 * it) * ld + cp);
    }
}

template <bool KC>
__device__ __forceinline__ void gt_r2s(double* s, const d2 (&r)[4], int t) {
    if (KC) {
        const int row = t >> 3, kp = (t & 7) * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<d2*>(s + (row + 32 * it) * GT_SKC + kp) = r[it];
    } else {
        const int k = t >> 6, cp = (t & 63) * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<d2*>(s + (k + 4 * it) * GT_SMN + cp) = r[it];
    }
}

template <bool KC>
__device__ __forceinline__ double gt_frag(const double* s, int idx, int kk) {
    return KC ? s[idx * GT_SKC + kk] : s[kk * GT_SMN + idx];
}

// acc[mi][ni] += sum_{k<K} opA(i,k) * opB(k,j) for this wave's 64x64 part of the 128x128 tile.
// A points at the tile's first row (k-contig) / first column (m-contig) at k = 0; same for B.  K % 16 == 0.
template <bool AK, bool BK>
__device__ __forceinline__ void gemm_tile_128(const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb, int K,
                                              d4 (&acc)[4][4], double* smem) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    d2 ra[4], rb[4];
    const int nk = K / GT_BK;
    gt_g2r<AK>(A, lda, 0, ra, t);
    gt_g2r<BK>(B, ldb, 0, rb, t);
    gt_r2s<AK>(smem, ra, t);
    gt_r2s<BK>(smem + GT_TILE, rb, t);
    __syncthreads();
    const int arow = wr * 64 + (lane & 15), bcol = wc * 64 + (lane & 15), kq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            gt_g2r<AK>(A, lda, (kt + 1) * GT_BK, ra, t);
            gt_g2r<BK>(B, ldb, (kt + 1) * GT_BK, rb, t);
        }
        const double* a_s = smem + cur * 2 * GT_TILE;
        const double* b_s = a_s + GT_TILE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = 4 * s + kq;
            double af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = gt_frag<AK>(a_s, arow + mi * 16, kk);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = gt_frag<BK>(b_s, bcol + ni * 16, kk);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_f64(af[mi], bf[ni], acc[mi][ni]);
        }
        if (kt + 1 < nk) {
            double* nxt = smem + (cur ^ 1) * 2 * GT_TILE;
            gt_r2s<AK>(nxt, ra, t);
            gt_r2s<BK>(nxt + GT_TILE, rb, t);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void gt_zero(d4 (&acc)[4][4]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};
}

// Epilogue: C (pointer to the tile's (0,0) element) = alpha*acc + beta*C.
// MODE 0: C = acc;  1: C = -acc;  2: C -= acc;  3: C = alpha*acc + beta*C.
// Read-modify-write modes first gather 16 C values (one 16-row band) into registers, then store:
// a load->store->load chain through one pointer would serialise on HBM latency.
template <int MODE>
__device__ __forceinline__ void gt_store(double* __restrict__ C, long ldc, const d4 (&acc)[4][4],
                                         double alpha = 1.0, double beta = 0.0) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    double* base = C + (long)(wr * 64 + (lane >> 4)) * ldc + wc * 64 + (lane & 15);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        double old[4][4];
        if (MODE >= 2) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[ni][r] = base[(long)(mi * 16 + 4 * r) * ldc + ni * 16];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* p = base + (long)(mi * 16 + 4 * r) * ldc + ni * 16;
                const double v = acc[mi][ni][r];
                if (MODE == 0) *p = v;
                else if (MODE == 1) *p = -v;
                else if (MODE == 2) *p = old[ni][r] - v;
                else *p = alpha * v + beta * old[ni][r];
            }
    }
}
