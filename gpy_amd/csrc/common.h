// common.h -- shared definitions for the gfx950 kernels of libmi355gp.so (CDNA4 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define NB 128            // base block: diagonal blocks, GEMM tiles and padding granule
#define NBO 512           // outer panel width of the two-level right-looking Cholesky
#define FACTOR_NBO_SMALL_N 0      // npad up to which the outer panels are FACTOR_NBO_SMALL wide (0: never)
#define FACTOR_NBO_SMALL 256
#define FACTOR_DEFAULT_TRI_OVERLAP 1    // 1: inverse of the leading block overlapped with the second half of potrf
#define FACTOR_DEFAULT_PERSIST 1        // 1: small factorisations run as ONE persistent dataflow launch (persist.hip)
#define FACTOR_PERSIST_MAX_NT 64        // ... up to this many 128-tiles per dimension (N <= 8192 = the sync block's PS_MAXNT).  Round 5
                                        // stopped at 36 (N=5120: +4 %, the far-tile owners saturated and starved the near tiles);
                                        // with round 6's reserve policy and column-major far ownership (persist.hip) the launch wins
                                        // up to the layout's limit: whole evaluations, one box, N=5120 4.86 -> 3.83 ms, N=6144 6.92 ->
                                        // 5.61, N=7168 9.13 -> 8.08, N=8192 12.16 -> 11.17 (profiles/r6_persist_range.txt);
                                        // MI355GP_PERSIST_MAX_NT (diagnostics build) overrides
// info[0] codes of a persistent launch that did not complete (far above any column index):
//   PS_ABORT_INFO : a wait inside the dataflow timed out -- the matrix is partly overwritten, the caller rebuilds it
//   PS_ABORT_CLEAN: the launch was called off at its co-residency gate before anything was written -- the matrix is intact
// Either way the factorisation is redone with the launch-per-step schedule inside the same C-ABI call (never mapped to a
// LAPACK info, never surfaced to the jitter ladder).
#define PS_ABORT_INFO (1 << 30)
#define PS_ABORT_CLEAN ((1 << 30) + 1)
#define PS_SKIP_AFTER_CLEAN 16          // evaluations that stay on the launch-per-step schedule after a called-off launch
#define FACTOR_DEFAULT_DIAG_EXCL_FIRST 1   // small factorisations only (N < 6144): from N = 8192 on it measured slower
#define GEMM_DEFAULT_TRI64_MAX 512    // levels of the triangular inverse with at most this many 128-tiles per stage run as 64 x 64 quadrants
#define GEMM_DEFAULT_LAUUM64_MAX 528  // X^T X of at most this many lower 128-tiles (nt <= 32) runs as 64 x 64 quadrants
#define LAUUM_SPLIT_MAX_NT 44          // X^T X runs from a work list with the long k ranges cut up to this many 128-tiles per dimension
                                       // (measured against the single launch of whole-K tiles: N=4608 0.80 -> 0.61 ms, N=5120 0.94 -> 0.79,
                                       //  N=6144 1.305 -> 1.295, N=7168 1.96 -> 2.00; MI355GP_LAUUM_SPLIT=n > 1 overrides the bound)
#define GEMM_DEFAULT_UPD64_MAX 128  // trailing-update launches of at most this many 128-tiles run as 64 x 64 quadrants (0 = never)

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) * B(4x16) + C.  Per lane (l = 0..63):
//   A operand: A[row = l & 15][k = l >> 4]        (one double)
//   B operand: B[k = l >> 4][col = l & 15]        (one double)
//   C/D      : D[row = (l >> 4) + 4 * r][col = l & 15], r = 0..3   (four doubles)
// A K=16 product is four MFMAs; slice s uses k = (l >> 4) + 4 s so that accumulator register r of a
// previous product can be fed back directly as the B operand of slice r ("register chaining").
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// launcher-side error plumbing
void mi355gp_set_error(const char* fmt, ...);
#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            mi355gp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                              __LINE__);                                                         \
            return -(1000 + (int)_e);                                                            \
        }                                                                                        \
    } while (0)

static inline int64_t round_up(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

// ---- environment switches ----------------------------------------------------------------------------------------------
// The PRODUCT library (libmi355gp.so) reads nine variables, all through PRODUCT_ENV: the transport of the multi-process
// mode (TRANSPORT, IPC_HOST, IPC_TIMEOUT_S), the collective-sequence self-check of the grid mode (GRID_CHECK_SEQ) and five
// schedule choices that give the SAME BITS either way (GRAPH, PERSIST, PERSIST_AUTO, TRI_OVERLAP, GRID_LOOKAHEAD).
// Everything else -- the schedule overrides of the A/B tools under tools/, the fault injectors of
// tests/test_gpu_persist_safety.py, the bounding experiments (one of which computes WRONG numbers by construction) -- goes
// through DIAG_ENV and exists only in the diagnostics build (make -C gpy_amd/csrc diag -> libmi355gp_diag.so, -DMI355GP_DIAG);
// in the product build the macro is a null pointer, the names are not in the binary and the code behind them is dead.
// tests/test_abi.py counts the names in the shipped library.
#include <cstdlib>
#define PRODUCT_ENV(name) getenv("MI355GP_" name)
#ifdef MI355GP_DIAG
#define DIAG_ENV(name) getenv("MI355GP_" name)
#else
#define DIAG_ENV(name) ((const char*)nullptr)
#endif
static inline int diag_env_int(const char* v, int dflt) { return (v && *v) ? atoi(v) : dflt; }
