// gemm_tile.h -- the fp64 MFMA workhorse: one 128x128 output tile per workgroup.
//
// Geometry (MI355X / gfx950), two wave arrangements selected by the template parameter NW:
//   NW = 4 (256 threads): waves 2x2, each owns a 64x64 sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators
//                         (128 accumulator VGPRs; 2 workgroups per CU = 2 waves per SIMD)
//   NW = 8 (512 threads): waves 2x4, each owns a 64x32 sub-tile = 4x2 accumulators (64 VGPRs;
//                         2 workgroups per CU = 4 waves per SIMD, so the matrix pipe always finds a ready wave)
// K is consumed in slabs of 16 (four MFMA k-slices); slabs are double-buffered in LDS
// (2 x (A 18 KB + B 18 KB) = 72 KB, two workgroups per CU) and the next slab's global loads are issued
// before the current slab's MFMAs.
//
// Operand storage (row-major buffers with leading dimension ld):
//   k-contiguous  : element (i, k) at P[i*ld + k]   -> LDS image [128][18]  (stride 18 = 2*odd: the
//                   fragment read row*18+k is bank-conflict free for ds_read_b64)
//   m/n-contiguous: element (k, i) at P[k*ld + i]   -> LDS image [16][144] (stride 144 = 16 mod 32)
// NT = (A k-contig, B k-contig), NN = (A k-contig, B n-contig), TN = (A m-contig, B n-contig).
#pragma once
#include "common.h"

#define GT_BK 16
#ifndef GT_USE_V3
#define GT_USE_V3 1                       // 4-wave tile kernels take the LDS-DMA pipeline (v3); 0 = register-staged v1
#endif
#define GT_SKC 18
#define GT_SMN 144
#define GT_TILE 2304                      // doubles per staged operand slab (128*18 == 16*144)
#define GT_LDS_BYTES (4 * GT_TILE * 8)    // 73,728 B

template <int NW>
struct GTCfg {
    static constexpr int NTHR = NW * 64;
    static constexpr int NLD = 1024 / NTHR;          // 16-byte loads per operand slab per thread
    static constexpr int NI = (NW == 8) ? 2 : 4;     // 16-column accumulator blocks per wave
    static constexpr int WCOLS = NI * 16;            // columns of the wave's sub-tile
    static constexpr int WPR = 128 / WCOLS;          // waves per tile row
};

// k-contiguous staging: which (row, 16-byte chunk c of the 16-wide slab row) thread t moves in trip `it`.
// ds_write_b128 is serviced in 16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31} (+32); with row stride 18
// doubles a group is conflict-free iff its four lane-quads hit rows = 0,4,8,12 (mod 16) with the same chunk
// half, hence the row/chunk permutation below (global reads stay 128 B contiguous per 8 lanes).
// Trip `it` of the staging loops moves row0 + it*NW*8.
template <int NW>
__device__ __forceinline__ void gt_kc_map(int t, int& row0, int& c) {
    const int l = t & 63, w = t >> 6, o = l >> 3, m = o & 3;
    const int e = w * 2 + (o >> 2);                      // trip `it` adds NW*2 to e, i.e. NW*8 rows
    row0 = 16 * (e >> 2) + 4 * m + (e & 3);
    c = (l & 7) ^ ((m == 1 || m == 2) ? 4 : 0);
}
#define GT_KC_RSTEP(NW) ((NW) * 8)

template <bool KC, int NW>
__device__ __forceinline__ void gt_g2r(const double* __restrict__ P, long ld, int k0, d2 (&r)[GTCfg<NW>::NLD], int t) {
    if (KC) {
        int row0, c;
        gt_kc_map<NW>(t, row0, c);
        const double* p = P + (long)row0 * ld + k0 + 2 * c;
#pragma unroll
        for (int it = 0; it < GTCfg<NW>::NLD; ++it) r[it] = *reinterpret_cast<const d2*>(p + (long)(it * GT_KC_RSTEP(NW)) * ld);
    } else {
        const int k = t >> 6, cp = (t & 63) * 2;
#pragma unroll
        for (int it = 0; it < GTCfg<NW>::NLD; ++it)
            r[it] = *reinterpret_cast<const d2*>(P + (long)(k0 + k + NW * it) * ld + cp);
    }
}

template <bool KC, int NW>
__device__ __forceinline__ void gt_r2s(double* s, const d2 (&r)[GTCfg<NW>::NLD], int t) {
    if (KC) {
        int row0, c;
        gt_kc_map<NW>(t, row0, c);
#pragma unroll
        for (int it = 0; it < GTCfg<NW>::NLD; ++it)
            *reinterpret_cast<d2*>(s + (row0 + it * GT_KC_RSTEP(NW)) * GT_SKC + 2 * c) = r[it];
    } else {
        const int k = t >> 6, cp = (t & 63) * 2;
#pragma unroll
        for (int it = 0; it < GTCfg<NW>::NLD; ++it)
            *reinterpret_cast<d2*>(s + (k + NW * it) * GT_SMN + cp) = r[it];
    }
}

template <bool KC>
__device__ __forceinline__ double gt_frag(const double* s, int idx, int kk) {
    return KC ? s[idx * GT_SKC + kk] : s[kk * GT_SMN + idx];
}

template <bool AK, bool BK, bool NEGA = false>
__device__ __forceinline__ void gemm_tile_128_v3(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                 long ldb, int K, d4 (&acc)[4][4], double* smem);

// acc[mi][ni] += sum_{k<K} opA(i,k) * opB(k,j) for this wave's part of the 128x128 tile.
// A points at the tile's first row (k-contig) / first column (m-contig) at k = 0; same for B.  K % 16 == 0.
template <bool AK, bool BK, int NW, bool NEGA = false>
__device__ __forceinline__ void gemm_tile_128(const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb, int K,
                                              d4 (&acc)[4][GTCfg<NW>::NI], double* smem, int dbg_nosync = 0,
                                              int reverse_k = 0) {
    constexpr int NI = GTCfg<NW>::NI;
    if constexpr (GT_USE_V3 && NW == 4) {                    // LDS-DMA pipeline (see the v3 block at the end of this file)
        if (!dbg_nosync && !reverse_k) {
            gemm_tile_128_v3<AK, BK, NEGA>(A, lda, B, ldb, K, acc, smem);
            return;
        }
    }
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / GTCfg<NW>::WPR, wc = w % GTCfg<NW>::WPR;
    d2 ra[GTCfg<NW>::NLD], rb[GTCfg<NW>::NLD];
    const int nk = K / GT_BK;
    // reverse_k: walk the slabs from k = K-16 down to 0.  Tiles of one launch whose k-ranges END at a common point
    // (lauum, trtri stage 1) then sweep the same operand slabs at the same time, which is what keeps them in L2.
    const int kbeg = reverse_k ? K - GT_BK : 0, kinc = reverse_k ? -GT_BK : GT_BK;
    gt_g2r<AK, NW>(A, lda, kbeg, ra, t);
    gt_g2r<BK, NW>(B, ldb, kbeg, rb, t);
    gt_r2s<AK, NW>(smem, ra, t);
    gt_r2s<BK, NW>(smem + GT_TILE, rb, t);
    __syncthreads();
    const int arow = wr * 64 + (lane & 15), bcol = wc * GTCfg<NW>::WCOLS + (lane & 15), kq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            gt_g2r<AK, NW>(A, lda, kbeg + (kt + 1) * kinc, ra, t);
            gt_g2r<BK, NW>(B, ldb, kbeg + (kt + 1) * kinc, rb, t);
        }
        const double* a_s = smem + cur * 2 * GT_TILE;
        const double* b_s = a_s + GT_TILE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = 4 * s + kq;
            double af[4], bf[NI];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = NEGA ? -gt_frag<AK>(a_s, arow + mi * 16, kk) : gt_frag<AK>(a_s, arow + mi * 16, kk);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[ni] = gt_frag<BK>(b_s, bcol + ni * 16, kk);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma_f64(af[mi], bf[ni], acc[mi][ni]);
        }
        if (dbg_nosync & 1) continue;   // diagnostics only: MFMA + LDS-read steady state without staging / barriers
        if (kt + 1 < nk) {
            double* nxt = smem + (cur ^ 1) * 2 * GT_TILE;
            gt_r2s<AK, NW>(nxt, ra, t);
            gt_r2s<BK, NW>(nxt + GT_TILE, rb, t);
        }
        if (dbg_nosync & 2) continue;   // diagnostics only (racy, wrong results): staging but no barrier
        __syncthreads();
    }
}

template <int NW>
__device__ __forceinline__ void gt_zero(d4 (&acc)[4][GTCfg<NW>::NI]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < GTCfg<NW>::NI; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};
}

template <int NW>
__device__ __forceinline__ double* gt_cbase(double* C, long ldc) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / GTCfg<NW>::WPR, wc = w % GTCfg<NW>::WPR;
    return C + (long)(wr * 64 + (lane >> 4)) * ldc + wc * GTCfg<NW>::WCOLS + (lane & 15);
}

// acc = C through buffer loads: one 32-bit lane offset + scalar row offsets + immediate column offsets, so the 64
// loads of a wave cost no address VGPRs (64-bit flat addresses for them spill the 128-register accumulator budget).
// Ct must be workgroup-uniform.  Issued before the k-loop, the read half of "C -= A*B" hides behind the operand
// prologue; the k-loop then runs with negated A fragments and the epilogue is a plain store.
template <int NW>
__device__ __forceinline__ void gt_load_buf(const double* Ct, long ldc, d4 (&acc)[4][GTCfg<NW>::NI]) {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w / GTCfg<NW>::WPR, wc = w % GTCfg<NW>::WPR;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Ct), 0, 0x7fffffff, 0x00020000);
    const int ldb = (int)ldc * 8;
    const int voff = (wr * 64 + (lane >> 4)) * ldb + (wc * GTCfg<NW>::WCOLS + (lane & 15)) * 8;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int soff = (mi * 16 + 4 * r) * ldb;
#pragma unroll
            for (int ni = 0; ni < GTCfg<NW>::NI; ++ni) {
                const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + ni * 128, soff, 0);
                acc[mi][ni][r] = __builtin_bit_cast(double, v);
            }
        }
}

// acc = -C: the read half of "C -= A*B" issued before the k-loop, so its HBM latency hides behind the
// operand prologue instead of sitting between the last MFMA and the store.
template <int NW>
__device__ __forceinline__ void gt_load_neg(const double* __restrict__ C, long ldc, d4 (&acc)[4][GTCfg<NW>::NI]) {
    const double* base = gt_cbase<NW>(const_cast<double*>(C), ldc);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < GTCfg<NW>::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = -base[(long)(mi * 16 + 4 * r) * ldc + ni * 16];
}

// Epilogue: C (pointer to the tile's (0,0) element) = alpha*acc + beta*C.
// MODE 0: C = acc;  1: C = -acc;  2: C -= acc;  3: C = alpha*acc + beta*C.
// Read-modify-write modes first gather 16 C values (one 16-row band) into registers, then store:
// a load->store->load chain through one pointer would serialise on HBM latency.
template <int MODE, int NW>
__device__ __forceinline__ void gt_store(double* __restrict__ C, long ldc, const d4 (&acc)[4][GTCfg<NW>::NI],
                                         double alpha = 1.0, double beta = 0.0) {
    constexpr int NI = GTCfg<NW>::NI;
    double* base = gt_cbase<NW>(C, ldc);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        double old[NI][4];
        if (MODE >= 2) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[ni][r] = base[(long)(mi * 16 + 4 * r) * ldc + ni * 16];
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
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

// =====================================================================================================================
// v3 pipeline (the shipping one for 4-wave workgroups): operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging
// VGPRs, no ds_write pass.  The DMA destination is wave-uniform base + lane*16 B (linear), so the bank-conflict-free
// layout is obtained by permuting the per-lane SOURCE address:
//   k-contiguous operand : [128][16] unpadded; one DMA moves 8 rows x 128 B; lane l = (row l>>3, slot l&7) fetches the
//                          16-byte granule (slot ^ h(row)) of its row, h(r) = ((r>>1)&7) ^ (2 if 4 <= r&15 <= 11);
//                          the fragment of lane (row, kq) is two ds_read_b128 at slots (2kq+e) ^ h(row), e = 0,1
//                          -> physical k = 4kq + s for slice s (all four 16-lane b128 groups hit 16 distinct bank quads)
//   m/n-contiguous       : [16][132]; one DMA moves one k-row of 128 doubles; fragments are ds_read_b64 at row 4kq + s
//                          (stride 132 = 4 mod 8 puts the kq = 0 / 1 halves of a 32-lane group 32 banks apart)
// Two LDS stages; the DMAs of slab kt+1 are issued at the top of slab kt and retired (vmcnt(0)) before its barrier.
#define GT3_SMN 132
#define GT3_OP 2176                         // doubles reserved per operand per stage (max(128*16, 16*132) rounded up)
#define GT3_LDS_BYTES (2 * 2 * GT3_OP * 8)  // 69,632 B

__device__ __forceinline__ int gt3_h(int r) { return ((r >> 1) & 7) ^ ((((r & 15) >= 4) && ((r & 15) <= 11)) ? 2 : 0); }

// per-lane byte offsets (relative to the operand tile's base pointer at k = 0) of this wave's four DMAs per slab
template <bool KC>
__device__ __forceinline__ void gt3_src_offsets(long ld, int lane, int w, int (&voff)[4]) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * w + ii;
        if (KC) {
            const int row = 8 * i + (lane >> 3), g = (lane & 7) ^ gt3_h(row);
            voff[ii] = (int)((row * ld + 2 * g) * 8);
        } else {
            voff[ii] = (int)((i * ld + 2 * lane) * 8);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void gt3_issue(__amdgpu_buffer_rsrc_t rs, const int (&voff)[4], int soff, double* sdst, int w) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * w + ii;
        double* d = KC ? sdst + i * 128 : sdst + i * GT3_SMN;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d, 16, voff[ii], soff, 0, 0);
    }
}

template <bool AK, bool BK, bool NEGA>
__device__ __forceinline__ void gemm_tile_128_v3(const double* __restrict__ A, long lda,
                                                 const double* __restrict__ B, long ldb, int K, d4 (&acc)[4][4],
                                                 double* smem) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int nk = K / 16;
    const int arow = wr * 64 + (lane & 15), bcol = wc * 64 + (lane & 15), kq = lane >> 4;
    auto rsrc = [](const double* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, 0x7fffffff, 0x00020000);
    };
    int va[4], vb[4];
    gt3_src_offsets<AK>(lda, lane, w, va);
    gt3_src_offsets<BK>(ldb, lane, w, vb);
    // the descriptor base advances with k (a 32-bit offset K*ld*8 overflows from N = 16384 on): 16 columns or 16 rows per slab
    const long sa = AK ? 16 : 16 * lda, sb = BK ? 16 : 16 * ldb;
    // fragment read offsets (doubles) inside an operand image
    int fa0[4], fb0[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int r = arow + mi * 16, c = bcol + mi * 16;
        fa0[mi] = AK ? r * 16 : r;
        fb0[mi] = BK ? c * 16 : c;
    }
    const int ha = gt3_h(arow), hb = gt3_h(bcol);            // h depends on row & 15 only: the same for all mi / ni
    gt3_issue<AK>(rsrc(A), va, 0, smem, w);
    gt3_issue<BK>(rsrc(B), vb, 0, smem + GT3_OP, w);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the DMAs have landed
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            double* nxt = smem + (cur ^ 1) * 2 * GT3_OP;
            gt3_issue<AK>(rsrc(A + (kt + 1) * sa), va, 0, nxt, w);
            gt3_issue<BK>(rsrc(B + (kt + 1) * sb), vb, 0, nxt + GT3_OP, w);
        }
        const double* a_s = smem + cur * 2 * GT3_OP;
        const double* b_s = a_s + GT3_OP;
#pragma unroll
        for (int e = 0; e < 2; ++e) {                        // slices 2e, 2e+1
            d2 af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                if (AK) af[mi] = *reinterpret_cast<const d2*>(a_s + fa0[mi] + 2 * ((2 * kq + e) ^ ha));
                else af[mi] = (d2){a_s[(4 * kq + 2 * e) * GT3_SMN + fa0[mi]], a_s[(4 * kq + 2 * e + 1) * GT3_SMN + fa0[mi]]};
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if (BK) bf[ni] = *reinterpret_cast<const d2*>(b_s + fb0[ni] + 2 * ((2 * kq + e) ^ hb));
                else bf[ni] = (d2){b_s[(4 * kq + 2 * e) * GT3_SMN + fb0[ni]], b_s[(4 * kq + 2 * e + 1) * GT3_SMN + fb0[ni]]};
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = mfma_f64(NEGA ? -af[mi][s] : af[mi][s], bf[ni][s], acc[mi][ni]);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) before the barrier: slab kt+1 is in LDS
        __syncthreads();
    }
}

// The same pipeline with the K dimension running over a LIST of operand panels (grid.hip: the aggregated updates of the
// block-cyclic mode apply panels k0 .. k1-1 of `slabs` 16-wide slabs each in ONE pass over C).  Atab[k] / Btab[k] are the
// panels' base pointers (workgroup-uniform: scalar loads), aoff / boff this tile's offset inside every panel.  The two
// LDS stages roll across panel boundaries: no pipeline drain between panels.
template <bool AK, bool BK, bool NEGA>
__device__ __forceinline__ void gemm_tile_128_v3_multi(const double* const* __restrict__ Atab, long aoff, long lda,
                                                       const double* const* __restrict__ Btab, long boff, long ldb,
                                                       int k0, int k1, int slabs, d4 (&acc)[4][4], double* smem) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int nk = (k1 - k0) * slabs;
    const int arow = wr * 64 + (lane & 15), bcol = wc * 64 + (lane & 15), kq = lane >> 4;
    auto rsrc = [](const double* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, 0x7fffffff, 0x00020000);
    };
    int va[4], vb[4];
    gt3_src_offsets<AK>(lda, lane, w, va);
    gt3_src_offsets<BK>(ldb, lane, w, vb);
    const long sa = AK ? 16 : 16 * lda, sb = BK ? 16 : 16 * ldb;
    int fa0[4], fb0[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int r = arow + mi * 16, c = bcol + mi * 16;
        fa0[mi] = AK ? r * 16 : r;
        fb0[mi] = BK ? c * 16 : c;
    }
    const int ha = gt3_h(arow), hb = gt3_h(bcol);
    int pk = k0, ps = 0;                                     // panel / slab-in-panel of the NEXT slab to fetch
    const double* Ab = Atab[pk] + aoff;
    const double* Bb = Btab[pk] + boff;
    gt3_issue<AK>(rsrc(Ab), va, 0, smem, w);
    gt3_issue<BK>(rsrc(Bb), vb, 0, smem + GT3_OP, w);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            if (++ps == slabs) {
                ps = 0;
                ++pk;
                Ab = Atab[pk] + aoff;
                Bb = Btab[pk] + boff;
            }
            double* nxt = smem + (cur ^ 1) * 2 * GT3_OP;
            gt3_issue<AK>(rsrc(Ab + ps * sa), va, 0, nxt, w);
            gt3_issue<BK>(rsrc(Bb + ps * sb), vb, 0, nxt + GT3_OP, w);
        }
        const double* a_s = smem + cur * 2 * GT3_OP;
        const double* b_s = a_s + GT3_OP;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            d2 af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                if (AK) af[mi] = *reinterpret_cast<const d2*>(a_s + fa0[mi] + 2 * ((2 * kq + e) ^ ha));
                else af[mi] = (d2){a_s[(4 * kq + 2 * e) * GT3_SMN + fa0[mi]], a_s[(4 * kq + 2 * e + 1) * GT3_SMN + fa0[mi]]};
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if (BK) bf[ni] = *reinterpret_cast<const d2*>(b_s + fb0[ni] + 2 * ((2 * kq + e) ^ hb));
                else bf[ni] = (d2){b_s[(4 * kq + 2 * e) * GT3_SMN + fb0[ni]], b_s[(4 * kq + 2 * e + 1) * GT3_SMN + fb0[ni]]};
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = mfma_f64(NEGA ? -af[mi][s] : af[mi][s], bf[ni][s], acc[mi][ni]);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
}

// =====================================================================================================================
// 64 x 64 output tiles on the v3 (LDS-DMA) pipeline, for launches with FEWER 128-tiles than the chip has CUs.
// fp64 MFMA peak is per CU (4 SIMDs x one 16x16x4 MFMA per 64 cycles = 0.31 TFLOP/s): a 128x128 tile confined to one
// workgroup = one CU needs >= 55 us for K = 512 and >= 14 us for K = 128 however empty the rest of the GPU is.  The
// latency-bound kernels of the panel chain (the K = 512 "part 1" update of the next panel's columns, the K = 128 in-panel
// updates) have 10 .. 120 such tiles; cut into 64 x 64 quadrants they occupy four times as many CUs and finish in about a
// third of the time.  Same slab size (16), same k -> MFMA-slice assignment (physical k = 4 kq + s) and the same
// accumulation order as gemm_tile_128_v3: results are bit-identical, whichever of the two a launch takes.
//   waves 2 x 2, each a 32 x 32 sub-tile = 2 x 2 accumulators (32 VGPRs)
//   k-contiguous operand : [64][16], 8 DMAs of 8 rows per slab (two per wave), source-side granule swizzle gt3_h
//   m/n-contiguous       : 16 k-rows of 64 doubles, one DMA moves the PAIR (2i, 2i+1) (lanes 0-31 / 32-63); pairs are 136
//                          doubles apart (136 = 8 mod 16 puts the kq = 0 / 1 halves of a 32-lane ds_read_b64 group 32
//                          banks apart)
// LDS: 2 stages x 2 operands x 1088 doubles = 34,816 B.
#define GT64_OP 1088
#define GT64_PAIR 136
#define GT64_LDS_BYTES (2 * 2 * GT64_OP * 8)

template <bool KC>
__device__ __forceinline__ void gt64_src_offsets(long ld, int lane, int w, int (&voff)[2]) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * w + ii;
        if (KC) {
            const int row = 8 * i + (lane >> 3), g = (lane & 7) ^ gt3_h(row);
            voff[ii] = (int)((row * ld + 2 * g) * 8);
        } else {
            voff[ii] = (int)(((2 * i + (lane >> 5)) * ld + 2 * (lane & 31)) * 8);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void gt64_issue(__amdgpu_buffer_rsrc_t rs, const int (&voff)[2], double* sdst, int w) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * w + ii;
        double* d = KC ? sdst + i * 128 : sdst + i * GT64_PAIR;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d, 16, voff[ii], 0, 0, 0);
    }
}

// acc[mi][ni] += sum_{k<K} opA(i,k) * opB(k,j) for this wave's 32 x 32 part of a 64 x 64 tile (A, B as in gemm_tile_128)
template <bool AK, bool BK, bool NEGA = false>
__device__ __forceinline__ void gemm_tile_64_v3(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                long ldb, int K, d4 (&acc)[2][2], double* smem) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int nk = K / 16;
    const int arow = wr * 32 + (lane & 15), bcol = wc * 32 + (lane & 15), kq = lane >> 4;
    auto rsrc = [](const double* p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, 0x7fffffff, 0x00020000);
    };
    int va[2], vb[2];
    gt64_src_offsets<AK>(lda, lane, w, va);
    gt64_src_offsets<BK>(ldb, lane, w, vb);
    const long sa = AK ? 16 : 16 * lda, sb = BK ? 16 : 16 * ldb;
    int fa0[2], fb0[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = arow + mi * 16, c = bcol + mi * 16;
        fa0[mi] = AK ? r * 16 : r;
        fb0[mi] = BK ? c * 16 : c;
    }
    const int ha = gt3_h(arow), hb = gt3_h(bcol);
    gt64_issue<AK>(rsrc(A), va, smem, w);
    gt64_issue<BK>(rsrc(B), vb, smem + GT64_OP, w);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            double* nxt = smem + (cur ^ 1) * 2 * GT64_OP;
            gt64_issue<AK>(rsrc(A + (kt + 1) * sa), va, nxt, w);
            gt64_issue<BK>(rsrc(B + (kt + 1) * sb), vb, nxt + GT64_OP, w);
        }
        const double* a_s = smem + cur * 2 * GT64_OP;
        const double* b_s = a_s + GT64_OP;
#pragma unroll
        for (int e = 0; e < 2; ++e) {                        // slices 2e, 2e+1
            d2 af[2], bf[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (AK) af[mi] = *reinterpret_cast<const d2*>(a_s + fa0[mi] + 2 * ((2 * kq + e) ^ ha));
                else af[mi] = (d2){a_s[(2 * kq + e) * GT64_PAIR + fa0[mi]], a_s[(2 * kq + e) * GT64_PAIR + 64 + fa0[mi]]};
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (BK) bf[ni] = *reinterpret_cast<const d2*>(b_s + fb0[ni] + 2 * ((2 * kq + e) ^ hb));
                else bf[ni] = (d2){b_s[(2 * kq + e) * GT64_PAIR + fb0[ni]], b_s[(2 * kq + e) * GT64_PAIR + 64 + fb0[ni]]};
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = mfma_f64(NEGA ? -af[mi][s] : af[mi][s], bf[ni][s], acc[mi][ni]);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
}

__device__ __forceinline__ void gt64_zero(d4 (&acc)[2][2]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};
}

// element (mi, ni, r) of this lane lives at C[(wr*32 + mi*16 + 4r + lane>>4) * ldc + wc*32 + ni*16 + (lane & 15)]
__device__ __forceinline__ void gt64_load(const double* __restrict__ C, long ldc, d4 (&acc)[2][2]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const double* base = C + (long)(wr * 32 + (lane >> 4)) * ldc + wc * 32 + (lane & 15);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = base[(long)(mi * 16 + 4 * r) * ldc + ni * 16];
}

// MODE 0: C = acc;  1: C = -acc
template <int MODE>
__device__ __forceinline__ void gt64_store(double* __restrict__ C, long ldc, const d4 (&acc)[2][2]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    double* base = C + (long)(wr * 32 + (lane >> 4)) * ldc + wc * 32 + (lane & 15);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) base[(long)(mi * 16 + 4 * r) * ldc + ni * 16] = (MODE == 1) ? -acc[mi][ni][r] : acc[mi][ni][r];
}
