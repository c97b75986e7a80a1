/* mi355gp_debug.h -- diagnostic and micro-benchmark entry points of libmi355gp.so, used by tests/ and tools/ only.
 * NOT part of the drop-in boundary: the product contract is include/mi355gp.h (the entry points GPy's hot path binds,
 * SURVEY.md 8(b)).  Nothing here has a counterpart in the reference. */
#ifndef MI355GP_DEBUG_H
#define MI355GP_DEBUG_H
#include "mi355gp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Device-only benchmark of the factorisation on a synthetic SPD matrix already resident in HBM:
 * returns average milliseconds of potrf / trtri / lauum over `reps` runs. */
int mi355gp_bench_factor(int device, int64_t N, int reps, double* ms_potrf, double* ms_trtri, double* ms_lauum);

/* Self-test of the multi-PROCESS transport (csrc/ipc_comm.hip; MI355GP_TRANSPORT=ipc binds it in place of RCCL for ranks that
 * are processes sharing one GPU): communicator set-up as in mi355gp_grid_create, then broadcasts / all-reduces of `count` doubles
 * on the world, row and column communicators.  With MI355GP_IPC_HOST=1 every buffer is host memory (no HIP call): the protocol
 * is testable without a GPU.  out4 = [mismatching doubles, checksum, rank inside the row communicator, inside the column one].
 * id128: from mi355gp_grid_unique_id under MI355GP_TRANSPORT=ipc. */
int mi355gp_dbg_ipc_selftest(const void* id128, int rank, int world, int Pr, int Pc, int64_t count, double* out4);
/* Self-test of the BOUND transport -- RCCL, or the hipIpc stand-in under MI355GP_TRANSPORT=ipc -- through the function table the
 * grid mode calls: CommInitRank, two CommSplits (row / column), one world broadcast, `rounds` broadcasts with rotating roots
 * inside ONE GroupStart / GroupEnd on the row and on the column communicator (the pattern of grid.hip's crit(k)), a world and a
 * row all-reduce, on a non-default stream, every payload checked.  out4 = [mismatching doubles, checksum, grid column, grid row].
 * The first step of tools/rccl_first_light.sh on a node with more than one GPU. */
int mi355gp_dbg_comm_selftest(int device, const void* id128, int rank, int world, int Pr, int Pc, int64_t count, int rounds,
                              double* out4);
/* diagnostics: the deep-K X^T X pass of the grid mode against the single-GPU lauum kernel */
int mi355gp_dbg_grid_multi(int device, int T, int nb, int reps, double* out_ms4);
/* diagnostics: the trailing-update kernel alone, lower triangle of nt x nt tiles, panel depths ks[0..nk) */
int mi355gp_dbg_update_nt(int device, int nt, const int* ks, int nk, int reps, double* out_ms);
int mi355gp_dbg_update_rect(int device, int ntr, int ntc, const int* ks, int nk, int reps, double* out_ms);

/* ---- diagnostics (used by tests/ and tools/) --------------------------------------------------------- */
/* raw lane dump of one v_mfma_f64_16x16x4_f64: a[64], b[64] -> d[64*4] */
int mi355gp_dbg_mfma(int device, const double* a, const double* b, double* d);
/* C (M x N) = alpha*op(A) op(B) + beta*C with the tiled MFMA kernel; M,N,K multiples of 128.
 * transa/transb: 0 = operand stored k-contiguous (A: M x K row-major, B: N x K row-major), 1 = stored K x M / K x N. */
int mi355gp_dbg_gemm(int device, int a_mcontig, int b_ncontig, int64_t M, int64_t N, int64_t K,
                     const double* A, const double* B, double* C, double alpha, double beta, int reps, double* ms);
/* microbenchmarks, out8: [0] fp64 MFMA TFLOP/s (8 workgroups/CU), [1] fp64 VALU FMA TFLOP/s, [2] HBM copy GB/s,
 * [3] HBM fill GB/s, [4] shader cycles per v_mfma_f64_16x16x4 at one wave/SIMD, [5] effective shader MHz under the
 * full MFMA load, [6] MFMA TFLOP/s at one wave/SIMD, [7] shader cycles per MFMA per SIMD under the full load */
int mi355gp_dbg_peaks(int device, double* out8);
/* effective shader clock (MHz) and shader cycles of workgroup 0 of the last mi355gp_dbg_gemm launch */
int mi355gp_dbg_gemm_clock(double* mhz, double* cycles);
/* diagnostics: where do workgroups land?  out[2b] = HW_REG_HW_ID, out[2b+1] = HW_REG_XCC_ID of workgroup b of a launch of nwg
 * spinning workgroups, machine-wide (mask_bit < 0) or on a stream whose CU mask has the single bit mask_bit
 * (tools/cu_map.py: logical CU b is CU (b/8)/4 of shader engine (b/8)%4 of XCD b%8; an XCD WITHOUT a mask bit is unrestricted) */
int mi355gp_dbg_cu_map(int device, int nwg, int mask_bit, unsigned* out);
/* Diagnostic: do fp64 MFMA and fp64 VALU FMA share one issue pipe?  out4 = ms of the same launch shape with all workgroups
 * on the MFMA stream / all on the FMA stream / alternating, and the MFMA TF/s of the first. */
int mi355gp_dbg_pipe_share(int device, double* out4);
/* Diagnostic: build + potrf + trtri + lauum of a synthetic N x N problem launched kernel by kernel vs replayed from a hipGraph
 * captured from the same streams.  out3 = ms launched, ms replayed, graph nodes. */
int mi355gp_dbg_graph_factor(int device, int64_t N, int reps, double* out3);
/* Diagnostic: the persistent dataflow Cholesky (persist.hip) against the launch-per-step schedule on the same resident SPD
 * matrix.  out[0] ms per factorisation launch-per-step, [1] persistent, [2] doubles of the lower triangle of L that differ
 * bitwise between the two, [3] info, [4] abort word; out[8 + 8 j + q]: wall-clock stamps (100 MHz ticks) of chain step j
 * (q = 0 factor start, 1 factor end, 2 sub-diagonal tile seen, 3 solve end, 4 diagonal tile seen, 5 update end).
 * then for tile row i and d = i - k in 0..2 (the near tiles): out[8 + 8 nt + 4 (3 i + d) + q], q = 0 last task picked, 1 computed,
 * 2 published.  kcap: columns a worker applies per pass (0 = default).  out: 8 + 20 ceil(N / 128) doubles. */
int mi355gp_dbg_persist(int device, int64_t N, int reps, int kcap, double* out);
/* Host only (no GPU call): the work list of X^T X for an nt x nt tile matrix as gemm.hip plans it.
 * items: up to max_items rows of 6 ints (ti, tj, q, k0, klen, part); out4 = number of items, number of partial tiles, edge of an
 * item's output tile (64 or 128), longest chunk in rows.  Returns 0, or -1 if max_items is too small (out4[0] still set). */
int mi355gp_dbg_lauum_plan(int nt, int* items, int max_items, int* out4);
/* Host only (no GPU call): who owns which 128 x 128 tile in a persistent launch (persist.hip) of nw worker workgroups for an nt x nt
 * tile matrix under the diagnostics tune word (0 = what ships).  owner: nt * nt ints; owner[i * nt + k] for k <= i = worker of tile
 * (i, k), -1 for block (0, 0) (the chain's); a tile (i, i - 2) held in 64-row halves reports its top half's owner there and its bottom
 * half's owner in the mirrored entry owner[k * nt + i].  out4 = near owners, half owners, far workers, most tiles / rows one worker
 * holds.  Returns the number of tiles claimed twice or out of range (0), -1 on bad arguments. */
int mi355gp_dbg_persist_owners(int nt, int nw, int tune, int* owner, int* out4);
/* diagnostics: is a CU-masked stream really confined?  out3 = ms of a 4096^3 GEMM on [plain, masked, masked] streams */
int mi355gp_dbg_mask_probe(int device, int pct, int order, double* out3);

#ifdef __cplusplus
}
#endif
#endif
