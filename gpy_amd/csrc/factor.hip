// factor.hip -- blocked drivers: the dpotrf / dtrtri / dlauum equivalents GPy reaches through
// GPy/util/linalg.py:56-75 (jitchol -> lapack.dpotrf) and :127-145,193-227 (pdinv: dtrtri + dpotri).
//
// Layout: one padded (npad x npad, npad % 128 == 0) row-major fp64 buffer per matrix, lower triangle
// significant; padding rows/cols carry the identity, which factorises and inverts to itself.
#include "internal.h"

// ---- profiling ------------------------------------------------------------------------------------
void KernelProf::begin(hipStream_t st, int fam, double flops) {
    open = on && ((mask >> fam) & 1u);
    if (!open) return;
    const size_t e0 = 2 * recs.size();
    while (pool.size() < e0 + 2) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        pool.push_back(e);
    }
    recs.push_back(Rec{fam, flops, e0});
    (void)hipEventRecord(pool[e0], st);
}

void KernelProf::end(hipStream_t st) {
    if (!open) return;
    open = false;
    (void)hipEventRecord(pool[recs.back().e0 + 1], st);
}

int KernelProf::collect(double* ms, double* flops, int* launches) {
    for (int f = 0; f < PF_NUM; ++f) { ms[f] = 0.0; flops[f] = 0.0; launches[f] = 0; }
    for (const Rec& r : recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, pool[r.e0], pool[r.e0 + 1]) != hipSuccess) return -1;
        ms[r.fam] += t;
        flops[r.fam] += r.flops;
        launches[r.fam] += 1;
    }
    return 0;
}

void KernelProf::destroy() {
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    pool.clear();
    recs.clear();
}

// ---- workspace --------------------------------------------------------------------------------------
int factor_ws_alloc(FactorWs* ws, long npad) {
    ws->nblk = npad / NB;
    HIP_CHECK(hipMalloc(&ws->dinv, sizeof(double) * ws->nblk * 8 * 256));
    HIP_CHECK(hipMalloc(&ws->logsum, sizeof(double) * ws->nblk));
    HIP_CHECK(hipMalloc(&ws->info, sizeof(int) * 4));
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_CHECK(hipStreamCreateWithPriority(&ws->st_panel, hipStreamNonBlocking, greatest));
    const size_t nouter = (size_t)(npad + NBO - 1) / NBO + 2;
    ws->ev_panel.resize(nouter);
    ws->ev_upd.resize(nouter);
    for (size_t i = 0; i < nouter; ++i) {
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_panel[i], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_upd[i], hipEventDisableTiming));
    }
    return 0;
}

void factor_ws_free(FactorWs* ws) {
    if (ws->dinv) (void)hipFree(ws->dinv);
    if (ws->logsum) (void)hipFree(ws->logsum);
    if (ws->info) (void)hipFree(ws->info);
    ws->dinv = ws->logsum = nullptr;
    ws->info = nullptr;
    for (hipEvent_t e : ws->ev_panel) (void)hipEventDestroy(e);
    for (hipEvent_t e : ws->ev_upd) (void)hipEventDestroy(e);
    ws->ev_panel.clear();
    ws->ev_upd.clear();
    if (ws->st_panel) (void)hipStreamDestroy(ws->st_panel);
    ws->st_panel = nullptr;
    ws->prof.destroy();
}

// algorithmic flops of a rank-K update of the lower triangle (incl. diagonal) of an (n x n) block / of an (m x n) block
static double syrk_flops(double n, double K) { return K * n * (n + 1.0); }
static double gemm_flops(double m, double n, double K) { return 2.0 * m * n * K; }

// One outer panel: columns [K0, K0+W), rows [K0, npad).  128-column steps:
//   diag128 (one CU) -> trsm128 on the rows below -> rank-128 update of the remaining columns of the panel.
static void factor_panel(hipStream_t s, double* A, long npad, long K0, long W, FactorWs* ws) {
    const long ld = npad;
    for (long j = 0; j < W; j += NB) {
        const long c = K0 + j, blk = c / NB;
        double* dv = ws->dinv + blk * 8 * 256;
        ws->prof.begin(s, PF_DIAG, (double)NB * NB * NB / 3.0);
        launch_diag128(s, A, ld, c, dv, ws->logsum + blk, ws->info);
        ws->prof.end(s);
        const long below = npad - (c + NB);
        if (below <= 0) continue;
        ws->prof.begin(s, PF_TRSM, (double)below * NB * NB);
        launch_trsm128(s, A, ld, c, c + NB, below, dv);
        ws->prof.end(s);
        const long ncols = K0 + W - (c + NB);
        if (ncols > 0) {
            double* C = A + (c + NB) * ld + (c + NB);
            const double* P = A + (c + NB) * ld + c;
            ws->prof.begin(s, PF_UPDATE, syrk_flops((double)ncols, NB) + gemm_flops((double)(below - ncols), (double)ncols, NB));
            launch_update_nt(s, C, ld, P, ld, P, ld, NB, (int)(below / NB), (int)(ncols / NB), (int)((c + NB) / NB),
                             (int)((c + NB) / NB));
            ws->prof.end(s);
        }
    }
}

// Two-level right-looking Cholesky with one panel of look-ahead.  Outer panels of NBO = 512 columns keep the big
// trailing update at K = 512 (64 flop per byte of C traffic).  The update of outer step k is split into the next
// panel's 512 columns (part 1) and the rest (part 2); panel k+1 is factored on a second, high-priority stream while
// part 2 of step k still runs, so the latency-bound diag/trsm chain hides behind MFMA-bound work.
void potrf_device(hipStream_t st, double* A, long npad, FactorWs* ws) {
    const long ld = npad;
    (void)hipMemsetAsync(ws->info, 0, sizeof(int) * 4, st);
    hipStream_t sp = ws->lookahead ? ws->st_panel : st;
    if (ws->lookahead) {
        (void)hipEventRecord(ws->ev_upd[0], st);                 // panel 0 follows everything queued on st so far
        (void)hipStreamWaitEvent(sp, ws->ev_upd[0], 0);
    }
    const long W0 = (npad < NBO) ? npad : NBO;
    factor_panel(sp, A, npad, 0, W0, ws);
    size_t k = 0;
    for (long K0 = 0; K0 < npad; K0 += NBO, ++k) {
        const long W = (npad - K0 < NBO) ? (npad - K0) : NBO;
        const long R0 = K0 + W;                                  // first row/col of the trailing matrix
        const long rest = npad - R0;
        if (rest <= 0) break;
        const long W1 = (rest < NBO) ? rest : NBO;               // width of the next panel
        const double* P = A + R0 * ld + K0;
        if (ws->lookahead) {
            (void)hipEventRecord(ws->ev_panel[k], sp);
            (void)hipStreamWaitEvent(st, ws->ev_panel[k], 0);
        }
        // part 1: the next panel's columns
        ws->prof.begin(st, PF_UPDATE, syrk_flops((double)W1, (double)W) + gemm_flops((double)(rest - W1), (double)W1, (double)W));
        launch_update_nt(st, A + R0 * ld + R0, ld, P, ld, P, ld, (int)W, (int)(rest / NB), (int)(W1 / NB), (int)(R0 / NB),
                         (int)(R0 / NB));
        ws->prof.end(st);
        if (ws->lookahead) {
            (void)hipEventRecord(ws->ev_upd[k + 1], st);
            (void)hipStreamWaitEvent(sp, ws->ev_upd[k + 1], 0);
        }
        factor_panel(sp, A, npad, R0, W1, ws);
        // part 2: everything to the right of the next panel
        const long rest2 = rest - W1;
        if (rest2 > 0) {
            const long R1 = R0 + W1;
            const double* P2 = A + R1 * ld + K0;
            ws->prof.begin(st, PF_UPDATE, syrk_flops((double)rest2, (double)W));
            launch_update_nt(st, A + R1 * ld + R1, ld, P2, ld, P2, ld, (int)W, (int)(rest2 / NB), (int)(rest2 / NB),
                             (int)(R1 / NB), (int)(R1 / NB));
            ws->prof.end(st);
        }
    }
    if (ws->lookahead) {
        (void)hipEventRecord(ws->ev_panel[k], sp);
        (void)hipStreamWaitEvent(st, ws->ev_panel[k], 0);
    }
}

// X = L^-1: diagonal 128-blocks on single CUs (all blocks concurrently), then log2(nt) batched levels.
void trtri_device(hipStream_t st, const double* L, double* X, double* T, long npad, FactorWs* ws) {
    const int nt = (int)(npad / NB);
    ws->prof.begin(st, PF_TRTRI, (double)npad * npad * npad / 3.0);
    launch_inv128(st, L, X, npad, nt, ws->dinv);
    for (int level = 0; (1 << level) < nt; ++level) launch_trtri_level(st, L, X, T, npad, nt, level);
    ws->prof.end(st);
}

void lauum_device(hipStream_t st, const double* X, double* W, long npad, FactorWs* ws) {
    ws->prof.begin(st, PF_LAUUM, (double)npad * npad * npad / 3.0);
    launch_lauum(st, X, W, npad, (int)(npad / NB));
    ws->prof.end(st);
}
