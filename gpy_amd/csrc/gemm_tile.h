// gemm_tile.h -- the fp64 MFMA workhorse: one 128x128 output tile per workgroup.
//
// Geometry (MI355X / gfx950): 256 threads = 4 waves arranged 2x2, each owns a 64x64 sub-tile = 4x4 v_mfma_f64_16x16x4_f64
// accumulators (128 accumulator VGPRs; 2 workgroups per CU = 2 waves per SIMD).  K is consumed in slabs of 16 (four MFMA
// k-slices); slabs are double-buffered in LDS and move global -> LDS by LDS-DMA (the pipeline is described at
// gemm_tile_128_v3 below; the register-staged and 8-wave pipelines of rounds 1-2 measured within +-4 % of it and are gone,
// DESIGN.md 6e).
//
// Operand storage (row-major buffers with leading dimension ld):
//   k-contiguous  : element (i, k) at P[i*ld + k]
//   m/n-contiguous: element (k, i) at P[k*ld + i]
// NT = (A k-contig, B k-contig), NN = (A k-contig, B n-contig), TN = (A m-contig, B n-contig).
#pragma once
#include "common.h"

// dynamic LDS asked for by the 128-tile kernels: the pipeline uses GT3_LDS_BYTES (69,632 B); the request stays at the 72 KiB
// every measured schedule ran with (two workgroups per CU either way)
#define GT_LDS_BYTES 73728

template <int NW>
struct GTCfg {
    static_assert(NW == 4, "the tile kernels run 4-wave workgroups");
    static constexpr int NTHR = NW * 64;
    static constexpr int NI = 4;                     // 16-column accumulator blocks per wave
    static constexpr int WCOLS = NI * 16;            // columns of the wave's sub-tile
    static constexpr int WPR = 128 / WCOLS;          // waves per tile row
};

template <bool AK, bool BK, bool NEGA = false>
__device__ __forceinline__ void gemm_tile_128_v3(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                 long ldb, int K, d4 (&acc)[4][4], double* smem);

// acc[mi][ni] += sum_{k<K} opA(i,k) * opB(k,j) for this wave's part of the 128x128 tile.
// A points at the tile's first row (k-contig) / first column (m-contig) at k = 0; same for B.  K % 16 == 0.
template <bool AK, bool BK, int NW, bool NEGA = false>
__device__ __forceinline__ void gemm_tile_128(const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb, int K,
                                              d4 (&acc)[4][GTCfg<NW>::NI], double* smem) {
    gemm_tile_128_v3<AK, BK, NEGA>(A, lda, B, ldb, K, acc, smem);
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
// The tile pipeline: operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging
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
