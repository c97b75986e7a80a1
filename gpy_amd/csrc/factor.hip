// factor.hip -- blocked drivers: the dpotrf / dtrtri / dlauum equivalents GPy reaches through
// GPy/util/linalg.py:56-75 (jitchol -> lapack.dpotrf) and :127-145,193-227 (pdinv: dtrtri + dpotri).
//
// Layout: one padded (npad x npad, npad % 128 == 0) row-major fp64 buffer per matrix, lower triangle
// significant; padding rows/cols carry the identity, which factorises and inverts to itself.
#include <cstdlib>
#include <mutex>
#include <vector>

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

// ---- process-wide stream engine ------------------------------------------------------------------------
// The three streams that carry concurrent heavy work -- main (trailing updates, trtri, lauum, everything else), panel
// (the high-priority panel chain) and tri (the CU-masked early-inverse stream) -- exist ONCE per device and process,
// created back to back on first use, and are shared by every context / workspace on that device.  Hardware queues are
// handed to the command processor's pipes in creation order, and two busy queues on one pipe take turns dispatch by
// dispatch: the same schedule ran potrf in 35 or in 45 ms (and the overlapped inverse in 13 or 20 ms) depending on which
// streams a process had created and destroyed before.  One fixed set makes the good case the only case; the calls of
// different contexts are serialised on the main stream, which is what a saturated GPU does to them anyway.
struct FactorEngine {
    hipStream_t main = nullptr, panel = nullptr, tri = nullptr, tri_half = nullptr;
    int tri_pct = 0;
    bool ready = false;
};
static FactorEngine g_engine[16];
static std::mutex g_engine_mutex;
static std::shared_mutex g_engine_gate[16];
std::shared_mutex& engine_gate(int device) { return g_engine_gate[device & 15]; }

int factor_engine(int device, hipStream_t* main, hipStream_t* panel, hipStream_t* tri, hipStream_t* tri_half) {
    if (device < 0 || device >= 16) return -1;
    std::lock_guard<std::mutex> lock(g_engine_mutex);
    FactorEngine& e = g_engine[device];
    if (!e.ready) {
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithFlags(&e.main, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithPriority(&e.panel, hipStreamNonBlocking, greatest));
        const char* envc = DIAG_ENV("TRI_CU_PCT");
        e.tri_pct = (envc && *envc) ? atoi(envc) : 75;
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount, nx = 8;
        if (e.tri_pct > 0 && e.tri_pct < 100 && ncu % nx == 0) {
            // XCD-balanced mask (logical CU i sits on XCD i % 8): the same share of every XCD's CUs
            const int per = ncu / nx, keep = (per * e.tri_pct + 50) / 100;
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            const char* envm = DIAG_ENV("TRI_MASK_MODE");
            const int mode = (envm && *envm) ? atoi(envm) : 0;
            for (int cu = 0; cu < ncu; ++cu) {
                const int idx = cu / nx;                       // CU index inside its XCD
                bool on = idx < keep;                          // mode 0: the first `keep` CUs of every XCD
                if (mode == 1) on = (idx % 4) != 3;            // mode 1: three of every four (75 %), interleaved
                if (mode == 2) on = idx >= per - keep;         // mode 2: the last `keep`
                if (on) mask[cu / 32] |= 1u << (cu % 32);
            }
            if (hipExtStreamCreateWithCUMask(&e.tri, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
                (void)hipGetLastError();
                e.tri = nullptr;
            }
        }
        if (!e.tri) HIP_CHECK(hipStreamCreateWithPriority(&e.tri, hipStreamNonBlocking, least));
        // fourth queue (fourth pipe): the same side stream on two of the four shader engines of every XCD, for
        // factorisations whose early-inverse work is small next to what potrf still has to do
        if (ncu % nx == 0) {
            const int per = ncu / nx, keep = per / 2;
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int cu = 0; cu < ncu; ++cu)
                if (cu / nx < keep) mask[cu / 32] |= 1u << (cu % 32);
            if (hipExtStreamCreateWithCUMask(&e.tri_half, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
                (void)hipGetLastError();
                e.tri_half = nullptr;
            }
        }
        // Hardware queues are allocated on a stream's FIRST submission, not when it is created: a process whose first small
        // evaluation uses main + tri (persistent launch + early inverse) and only later panel would hand the queues to the
        // command processor's pipes in another order than one that starts with a large evaluation (main, panel, tri) -- measured
        // as +0.5 ms per evaluation at N = 16384 behind a parity-gate run at N = 2048.  One tiny submission per stream, in
        // creation order, pins the order for the life of the process.
        {
            void* touch = nullptr;
            if (hipMalloc(&touch, 256) == hipSuccess) {
                hipStream_t order[4] = {e.main, e.panel, e.tri, e.tri_half};
                for (hipStream_t s : order)
                    if (s) {
                        (void)hipMemsetAsync(touch, 0, 256, s);
                        (void)hipStreamSynchronize(s);
                    }
                (void)hipFree(touch);
            }
        }
        e.ready = true;
    }
    if (main) *main = e.main;
    if (panel) *panel = e.panel;
    if (tri) *tri = e.tri;
    if (tri_half) *tri_half = e.tri_half;
    return 0;
}

// ---- workspace --------------------------------------------------------------------------------------
static bool lauum_plan_wanted(const FactorWs* ws, int nt);
static bool lauum_plan_build(FactorWs* ws, int nt);
int factor_ws_alloc(FactorWs* ws, long npad) {
    ws->nblk = npad / NB;
    HIP_CHECK(hipMalloc(&ws->dinv, sizeof(double) * ws->nblk * 8 * 256));
    HIP_CHECK(hipMalloc(&ws->logsum, sizeof(double) * ws->nblk));
    HIP_CHECK(hipMalloc(&ws->info, sizeof(int) * 4));
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (factor_engine(dev, nullptr, &ws->st_panel, &ws->st_tri, &ws->st_tri_half) != 0) return -1;    // shared, never destroyed by a workspace
    {
        const char* envo = PRODUCT_ENV("TRI_OVERLAP");
        if (envo && *envo) ws->tri_overlap = atoi(envo) ? 1 : 0;
        const char* envn2 = DIAG_ENV("TRI_MIN_NT");
        if (envn2 && *envn2) ws->tri_min_nt = atoi(envn2);
        const char* envw2 = DIAG_ENV("TRI_WGS");
        if (envw2 && *envw2) ws->tri_wgs = atoi(envw2);
        ws->tri_cu_pct = g_engine[dev].tri_pct > 0 && g_engine[dev].tri_pct <= 100 ? g_engine[dev].tri_pct : 75;
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_tri, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_tri_lead, hipEventDisableTiming));
        HIP_CHECK(hipMalloc(&ws->tri_counter, sizeof(int) * 4));
        const char* envh = DIAG_ENV("TRI_H");
        if (envh && *envh) ws->tri_h_override = atoi(envh);
    }
    const char* envp1 = DIAG_ENV("PART1_ON_PANEL");
    if (envp1 && *envp1) ws->part1_on_panel = atoi(envp1) ? 1 : 0;
    ws->sched_state = ws->sched_force_steps = ws->persist_auto_off = 0;
    ws->sched_np = ws->sched_ns = 0;
    ws->evals_done = ws->early_pending = 0;
    const char* envpa = PRODUCT_ENV("PERSIST_AUTO");
    if (envpa && *envpa) ws->persist_auto = atoi(envpa) ? 1 : 0;
    const char* envls = DIAG_ENV("LAUUM_SPLIT");
    if (envls && *envls) ws->lauum_split = atoi(envls) > 0 ? atoi(envls) : 0;   // 1: on; n > 1: on up to n tiles per dimension
    const char* envptri = DIAG_ENV("PERSIST_TRI");
    if (envptri && *envptri) ws->persist_tri = atoi(envptri) ? 1 : 0;
    const char* envptn = DIAG_ENV("PERSIST_TRI_MIN_NT");
    if (envptn && *envptn) ws->persist_tri_min_nt = atoi(envptn);
#ifdef MI355GP_DIAG
    const char* envuq = DIAG_ENV("DBG_UPD_QUEUE");
    if (envuq && *envuq && atoi(envuq)) {
        ws->upd_queue_probe = 1;
        HIP_CHECK(hipMalloc(&ws->upd_tasks, sizeof(UpdTask) * 64 + 64));
    }
#endif
    const char* envag = DIAG_ENV("AGG2");
    if (envag && *envag) ws->agg2 = atoi(envag) ? 1 : 0;
    const char* envnbo = DIAG_ENV("NBO");
    if (envnbo && *envnbo) ws->nbo_override = atoi(envnbo);
    const char* envso = DIAG_ENV("SOLVE_OVERLAP");
    if (envso && *envso) ws->solve_overlap = atoi(envso) ? 1 : 0;
    const char* envth = DIAG_ENV("TRI_HALF");
    if (envth && *envth) ws->tri_half_ok = atoi(envth) ? 1 : 0;
    const char* envps = PRODUCT_ENV("PERSIST");
    if (envps && *envps) ws->persist = atoi(envps);
    const char* envpm = DIAG_ENV("PERSIST_MAX_NT");
    if (envpm && *envpm) ws->persist_max_nt = atoi(envpm);
    const char* envpt = DIAG_ENV("PERSIST_TUNE");
    if (envpt && *envpt) ws->persist_tune = atoi(envpt);
    const char* envpk = DIAG_ENV("PERSIST_KCAP");
    if (envpk && *envpk) ws->persist_kcap = atoi(envpk) > 0 ? atoi(envpk) : 1;
    {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        ws->persist_cus = prop.multiProcessorCount;
        const char* envpc = DIAG_ENV("PERSIST_WGS");
        if (envpc && *envpc && atoi(envpc) >= 2 && atoi(envpc) <= ws->persist_cus) ws->persist_cus = atoi(envpc);
        HIP_CHECK(hipMalloc(&ws->persist_sync, sizeof(int) * potrf_persist_sync_ints()));
        HIP_CHECK(hipMalloc(&ws->persist_hs, sizeof(double) * (size_t)(ws->nblk < 64 ? ws->nblk : 64) * NB * NB));
    }
    const char* envx = DIAG_ENV("DIAG_EXCL_FIRST");
    if (envx && *envx) ws->diag_excl_first = atoi(envx) ? 1 : 0;
    HIP_CHECK(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ws->ev_early_pre, hipEventDisableTiming));
    if (lauum_plan_wanted(ws, (int)ws->nblk)) (void)lauum_plan_build(ws, (int)ws->nblk);   // never inside a capture later
    const size_t nouter = (size_t)(npad + NB - 1) / NB + 2;      // enough for the narrowest outer panel (nbo = 128)
    ws->ev_panel.resize(nouter);
    ws->ev_cols.resize(nouter);
    for (size_t i = 0; i < nouter; ++i) {
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_panel[i], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_cols[i], hipEventDisableTiming));
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
    for (hipEvent_t e : ws->ev_cols) (void)hipEventDestroy(e);
    ws->ev_panel.clear();
    ws->ev_cols.clear();
    if (ws->ev_fork) (void)hipEventDestroy(ws->ev_fork);
    ws->ev_fork = nullptr;
    if (ws->ev_early_pre) (void)hipEventDestroy(ws->ev_early_pre);
    ws->ev_early_pre = nullptr;
    ws->st_panel = nullptr;                       // engine streams are shared and never destroyed by a workspace
    ws->st_tri = nullptr;
    ws->st_tri_half = ws->st_tri_cur = nullptr;
    if (ws->ev_tri) (void)hipEventDestroy(ws->ev_tri);
    ws->ev_tri = nullptr;
    if (ws->ev_tri_lead) (void)hipEventDestroy(ws->ev_tri_lead);
    ws->ev_tri_lead = nullptr;
    if (ws->tri_counter) (void)hipFree(ws->tri_counter);
    ws->tri_counter = nullptr;
    if (ws->upd_tasks) (void)hipFree(ws->upd_tasks);
    ws->upd_tasks = nullptr;
    if (ws->lauum_plan_dev) (void)hipFree(ws->lauum_plan_dev);
    ws->lauum_plan_dev = nullptr;
    if (ws->lauum_part) (void)hipFree(ws->lauum_part);
    ws->lauum_part = nullptr;
    ws->lauum_plan_nt = 0;
    if (ws->persist_sync) (void)hipFree(ws->persist_sync);
    ws->persist_sync = nullptr;
    if (ws->persist_hs) (void)hipFree(ws->persist_hs);
    ws->persist_hs = nullptr;
    ws->prof.destroy();
}

// algorithmic flops of a rank-K update of the lower triangle (incl. diagonal) of an (n x n) block / of an (m x n) block
static double syrk_flops(double n, double K) { return K * n * (n + 1.0); }
static double gemm_flops(double m, double n, double K) { return 2.0 * m * n * K; }

// One outer panel: columns [K0, K0+W), rows [K0, npad).  128-column steps, right-looking inside the panel:
//   diag128 (one CU) -> trsm128 on the rows below -> rank-128 update of the remaining columns of the panel.
// (The alternatives that were built and measured -- inverse-based panel solve, split chain / wide streams, one fused launch
//  per panel with flag hand-offs, a resident diagonal-block server, recursive order -- are written up in DESIGN.md 6e;
//  none beat this path and their code is gone.)
static void factor_panel(hipStream_t s, double* A, long npad, long K0, long W, FactorWs* ws) {
    const long ld = npad;
    for (long j = 0; j < W; j += NB) {
        const long c = K0 + j, blk = c / NB;
        double* dv = ws->dinv + blk * 8 * 256;
        ws->prof.begin(s, PF_DIAG, (double)NB * NB * NB / 3.0);
        launch_diag128(s, A, ld, c, dv, ws->logsum + blk, ws->info,
                       ws->diag_excl_first && c == K0 && ws->lookahead == 1 && ws->excl_first_ok);
        ws->prof.end(s);
        const long below = npad - (c + NB);
        if (below <= 0) continue;
        ws->prof.begin(s, PF_TRSM, (double)below * NB * NB);
        launch_trsm128(s, A, ld, c, c + NB, below, dv);
        ws->prof.end(s);
        // rank-128 update of the panel's remaining columns [c + NB, K0 + W) (rows c + NB .. npad)
        const long cd = c + NB, ncols = K0 + W - cd;
        if (ncols <= 0) continue;
        const double* P = A + cd * ld + c;
        ws->prof.begin(s, update_nt_uses_64((int)(below / NB), (int)(ncols / NB), (int)(cd / NB), (int)(cd / NB)) ? PF_UPDATE64 : PF_UPDATE,
                       syrk_flops((double)ncols, (double)NB) + gemm_flops((double)(below - ncols), (double)ncols, (double)NB));
        launch_update_nt(s, A + cd * ld + cd, ld, P, ld, P, ld, NB, (int)(below / NB), (int)(ncols / NB), (int)(cd / NB), (int)(cd / NB));
        ws->prof.end(s);
    }
}

// rank-W update of the trailing columns [c0, c1) (rows c0 .. npad) with the panel at columns [K0, K0+W)
static void update_cols(hipStream_t s, double* A, long npad, long K0, long W, long c0, long c1, FactorWs* ws) {
    if (c1 <= c0) return;
    const long ld = npad, rows = npad - c0, cols = c1 - c0;
    const double* P = A + c0 * ld + K0;
    ws->prof.begin(s, update_nt_uses_64((int)(rows / NB), (int)(cols / NB), (int)(c0 / NB), (int)(c0 / NB)) ? PF_UPDATE64 : PF_UPDATE,
                   syrk_flops((double)cols, (double)W) + gemm_flops((double)(rows - cols), (double)cols, (double)W));
    launch_update_nt(s, A + c0 * ld + c0, ld, P, ld, P, ld, (int)W, (int)(rows / NB), (int)(cols / NB), (int)(c0 / NB),
                     (int)(c0 / NB));
    ws->prof.end(s);
}

// Reference schedule (option LOOKAHEAD = 0): everything in order on one stream
static void potrf_serial(hipStream_t st, double* A, long npad, FactorWs* ws) {
    (void)hipMemsetAsync(ws->info, 0, sizeof(int) * 4, st);
    const long nbo = ws->nbo_for(npad);
    const long P = (npad + nbo - 1) / nbo;
    auto pcol = [&](long p) { return (p * nbo < npad) ? p * nbo : npad; };
    for (long p = 0; p < P; ++p) {
        factor_panel(st, A, npad, pcol(p), pcol(p + 1) - pcol(p), ws);
        update_cols(st, A, npad, pcol(p), pcol(p + 1) - pcol(p), pcol(p + 1), npad, ws);
    }
}

// The part of X = L^-1 that only needs the leading h tile columns of L, on the side stream `sq` while the factorisation goes on:
// the inverse of the leading h x h tiles, then pair 0 of level log2(h), T21 = L21 X11 (rows h .. 2h of the finished columns; the
// tile list is shared with the machine-wide instance that trtri_device launches after potrf).  trtri_device picks up from
// ws->ovl_h.  gated: the factorisation is the PERSISTENT launch -- there is no stream event inside it, so each of the two steps
// is preceded by a one-thread kernel that waits on the launch's progress words (persist.hip).
static void early_inverse(hipStream_t sq, double* A, long npad, int h, FactorWs* ws, bool gated, int gate_gives_up = 0) {
    const int ntl = (int)(npad / NB);
    (void)hipMemsetAsync(ws->tri_counter, 0, sizeof(int) * 4, sq);
    if (gated) launch_wait_persist_rows(sq, ws, 0, h, 0, gate_gives_up);   // rows 0 .. h-1 of L, inverted diagonal tiles: final
    ws->prof.begin(sq, PF_TRTRI_EARLY, 0.0);                    // elapsed on the side stream; the flops are billed to PF_TRTRI
    launch_inv128(sq, A, ws->scratchX, npad, h, ws->dinv);
    int level = 0;
    for (; (1 << level) < h; ++level) launch_trtri_level(sq, A, ws->scratchX, ws->scratchT, npad, h, level);
    (void)hipEventRecord(ws->ev_tri_lead, sq);
    const int nt_pair = ntl < 2 * h ? ntl : 2 * h;
    if (gated) launch_wait_persist_rows(sq, ws, h, nt_pair, h); // columns 0 .. h-1 of rows h .. 2h-1 are final
    launch_trtri_stage1_steal(sq, A, ws->scratchX, ws->scratchT, npad, nt_pair, level, ws->tri_counter,
                              ws->tri_wgs > 0 ? ws->tri_wgs : 512 * ws->tri_cur_pct / 100);
    ws->prof.end(sq);
    (void)hipEventRecord(ws->ev_tri, sq);
    ws->early_pending = 1;
}

// Leading tiles whose inverse (+ T21) goes on the side stream underneath the PERSISTENT launch: the largest power of two <= 2 nt / 3,
// 0 = not this time.  (The hipGraph replay of the factorisation region is not used then -- api.hip asks here: measured at
// N = 4096, replay + early inverse 3.73 ms per evaluation, replay alone 3.35, plain launches + early inverse 3.05.)
int persist_early_h(long npad, const FactorWs* ws) {
    const int ntl = (int)(npad / NB);
    // (not in a workspace's FIRST evaluation: with the one-time set-up of a dozen kernels in between, the side stream's launches
    //  reached the GPU spread out enough that the persistent launch was called off at its gate once per context)
    if (!potrf_persist_eligible(npad, ws) || !ws->tri_overlap || !ws->persist_tri || !ws->st_tri || !ws->scratchX || !ws->scratchT ||
        ntl < ws->persist_tri_min_nt || ws->evals_done < 1)
        return 0;
    int h = 1;
    // the largest power of two <= 2 nt / 3 (round 6; up to round 5: <= nt / 2, which a tile count between two powers of two turns into a
    // third of the rows and 4 % of the inverse): whole evaluations N = 3072 1.83 -> 1.79 ms, N = 3584 2.30 -> 2.15, N = 7168 8.01 -> 7.77,
    // powers of two unchanged (profiles/r6_early_h_ab.txt); MI355GP_PERSIST_TRI_H23=0 (diagnostics build) brings nt / 2 back
    static const int two_thirds = diag_env_int(DIAG_ENV("PERSIST_TRI_H23"), 1);
    if (two_thirds) while (6 * h <= 2 * ntl) h *= 2;
    else while (4 * h <= ntl) h *= 2;
    return h;
}

// Two-level right-looking Cholesky with one panel of look-ahead (default).  Outer panels of NBO = 512 columns keep
// the big trailing update at K = 512 (64 flop per byte of C traffic).  The update of outer step p is split into the
// next panel's 512 columns (part 1) and the rest (part 2); panel p+1 is factored on a second, high-priority stream
// while part 2 of step p still runs, so the latency-bound diag/trsm chain hides behind MFMA-bound work.  All
// trailing updates stay in order on `st`: their launch durations are not inflated by overlapping each other.
void potrf_device(hipStream_t st, double* A, long npad, FactorWs* ws) {
    if (ws->early_pending) {                                    // early-inverse work of a factorisation whose inverse was never taken
        (void)hipStreamWaitEvent(st, ws->ev_tri, 0);            // (a called-off persistent launch that is being redone): it writes
        ws->early_pending = 0;                                  // the scratch buffers this run is about to use
    }
    ws->ovl_h = 0;
    ws->persist_used = 0;
    if (potrf_persist_eligible(npad, ws)) {                     // small factorisation: one persistent dataflow launch
        // Its near-tile owners leave as the chain passes their rows (three CUs per step), so the part of the inverse that
        // needs only the leading half of L starts on the side stream as soon as that half is final: the leading h x h
        // inverse and T21 = L21 X11 on the CUs the launch has given back, keyed on its progress words.
        const int h = persist_early_h(npad, ws);
        const int gate_gives_up = (ws->persist_test == 3) ? 1 : 0;       // fault injection: the first gate times out at once
        hipEvent_t pre_saved = ws->ev_persist_pre;
        if (h > 0 && !ws->ev_persist_pre) ws->ev_persist_pre = ws->ev_early_pre;   // (its own event: ev_fork belongs to the look-ahead schedule)
        const bool ok = launch_potrf_persist(st, A, npad, ws);
        hipEvent_t pre = ws->ev_persist_pre;
        ws->ev_persist_pre = pre_saved;
        if (ok) {
            ws->persist_used = 1;
            if (h > 0) {
                hipStream_t sq = ws->st_tri;
                ws->tri_cur_pct = ws->tri_cu_pct;
                (void)hipStreamWaitEvent(sq, pre, 0);           // the progress words of THIS launch are zeroed
                launch_wait_persist_resident(sq, ws, 50);       // nothing wide may reach the CUs before the launch is in place
                early_inverse(sq, A, npad, h, ws, true, gate_gives_up);
                ws->ovl_h = h;
            }
            return;
        }
        ws->persist = 0;                                        // the launch cannot be made on this device: never try again
    } else if (ws->persist_skip > 0 && ws->persist_skip != 0x7fffffff) {
        --ws->persist_skip;                                     // a called-off launch is retried after a number of evaluations
    }
    if (ws->sched_force_steps > 0) --ws->sched_force_steps;     // (one of the calibration's evaluations on launches is this one)
    if (ws->lookahead != 1) {
        potrf_serial(st, A, npad, ws);
        return;
    }
    (void)hipMemsetAsync(ws->info, 0, sizeof(int) * 4, st);
    const long nbo = ws->nbo_for(npad);
    const long P = (npad + nbo - 1) / nbo;
    auto pcol = [&](long p) { return (p * nbo < npad) ? p * nbo : npad; };
    hipStream_t sp = ws->st_panel;
    hipStream_t su = st;                                        // trailing updates stay in order on the caller's stream
    (void)hipEventRecord(ws->ev_fork, st);                      // panel 0 follows everything queued on st so far
    (void)hipStreamWaitEvent(sp, ws->ev_fork, 0);
    factor_panel(sp, A, npad, 0, pcol(1), ws);
    // Inverse of a leading block early (trtri_device picks up from ws->ovl_h): h tiles, a power of two <= nt/2 or the
    // largest power of two below nt, whichever still fits the time model: inverting the leading block (the part that
    // has to run at the masked stream's CU share) must not take longer than potrf needs for the rest of the matrix.
    const int ntl = (int)(npad / NB);
    int ovl_h = 0;
    if (ws->tri_overlap && ws->st_tri && ws->scratchX && ws->scratchT && ntl >= ws->tri_min_nt) {
        int h = 1;
        while (2 * h < ntl) h *= 2;
        for (; h >= 8; h /= 2) {
            const double lead = (double)h * NB, rest = (double)npad - lead;
            const double t_lead = lead * lead * lead / 3.0 / 45e12;
            const double t_tail = rest * rest * rest / 3.0 / 50e12 + rest / (double)nbo * 0.45e-3 * ((double)nbo / NBO);
            if (t_lead <= t_tail) break;
        }
        if (ws->tri_h_override > 0) h = ws->tri_h_override;
        if (h >= 8 && h < ntl && (h & (h - 1)) == 0) ovl_h = h;
        // which side stream: three shader engines per XCD when the early work (leading inverse + top-level L21 X11) is
        // about as long as the rest of potrf, two when it is much shorter (potrf is disturbed less; N=8192 -2 %,
        // N=20480 -2 %, but N=16384 +1.6 % with two)
        if (ovl_h > 0) {
            const double lead = (double)ovl_h * NB, rest = (double)npad - lead, right = rest < lead ? rest : lead;
            const double t_early = (lead * lead * lead / 3.0 + lead * lead * right) / 60e12;
            const double t_tail = rest * rest * rest / 3.0 / 50e12 + rest / (double)nbo * 0.45e-3 * ((double)nbo / NBO);
            ws->st_tri_cur = (ws->st_tri_half && ws->tri_half_ok && t_early < 0.8 * t_tail) ? ws->st_tri_half : ws->st_tri;
            ws->tri_cur_pct = (ws->st_tri_cur == ws->st_tri_half) ? 50 : ws->tri_cu_pct;
        }
    }
    ws->ovl_h = 0;
    ws->excl_first_ok = (ovl_h == 0 && ntl < ws->tri_min_nt) ? 1 : 0;   // small factorisations only (measured: N >= 8192 loses)
#ifdef MI355GP_DIAG
    const bool uq = ws->upd_queue_probe && ws->upd_tasks && ws->part1_on_panel && ntl >= ws->tri_min_nt && !ws->agg2 && P <= 64;
#else
    const bool uq = false;                                      // (the bounding experiment of the diagnostics build)
#endif
#ifdef MI355GP_DIAG
    if (uq) {
        // bounding experiment (wrong results by construction): all part-2 updates as ONE resident launch with every dependence
        // ignored, on the main stream from the start; the panel stream runs chain + part 1 as always, minus its waits for part 2
        std::vector<UpdTask> tasks;
        for (long p = 0; p + 2 < P; ++p) {
            const long c0 = pcol(p + 2), ntr = (npad - c0) / NB;
            tasks.push_back(UpdTask{c0 * npad + c0, c0 * npad + pcol(p), ntr * (ntr + 1) / 2, (int)(pcol(p + 1) - pcol(p)), (int)ntr});
        }
        int* counter = reinterpret_cast<int*>(ws->upd_tasks + 64);
        (void)hipMemcpyAsync(ws->upd_tasks, tasks.data(), sizeof(UpdTask) * tasks.size(), hipMemcpyHostToDevice, su);
        (void)hipMemsetAsync(counter, 0, sizeof(int), su);
        ws->prof.begin(su, PF_UPDATE, 0.0);
        launch_update_nt_queue(su, A, npad, ws->upd_tasks, (int)tasks.size(), counter, 512);
        ws->prof.end(su);
    }
#endif
    for (long p = 0; p + 1 < P; ++p) {
        const long K0 = pcol(p), W = pcol(p + 1) - K0;
        (void)hipEventRecord(ws->ev_panel[p], sp);
        if (ovl_h > 0 && pcol(p + 1) == (long)ovl_h * NB) {       // columns < 128 h are final: start on their inverse
            hipStream_t sq = ws->st_tri_cur ? ws->st_tri_cur : ws->st_tri;
            (void)hipStreamWaitEvent(sq, ws->ev_panel[p], 0);
            early_inverse(sq, A, npad, ovl_h, ws, false);
            ws->ovl_h = ovl_h;
        }
        if (ws->part1_on_panel && ntl >= ws->tri_min_nt) {   // measured: N=16384 potrf 35.5 -> 33.3 ms (evaluation -0.9 %), N=4096 +1.4..3.5 %
            // part 1 (the next panel's columns) on the panel stream itself: chain(p) -> part 1 -> chain(p+1) then runs in one
            // stream's order, no cross-stream hand-off on the critical path of a chain-bound factorisation.  Its tiles were
            // last written by part 2 of step p-1 (main stream): that event is long signalled when the chain is the bottleneck.
            if (p > 0 && !uq) (void)hipStreamWaitEvent(sp, ws->ev_cols[p], 0);
            update_cols(sp, A, npad, K0, W, pcol(p + 1), pcol(p + 2), ws);
            factor_panel(sp, A, npad, pcol(p + 1), pcol(p + 2) - pcol(p + 1), ws);
            if (uq) continue;
            (void)hipStreamWaitEvent(su, ws->ev_panel[p], 0);
            if (!ws->agg2) {
                update_cols(su, A, npad, K0, W, pcol(p + 2), npad, ws);      // part 2: everything to the right
                (void)hipEventRecord(ws->ev_cols[p + 1], su);
                continue;
            }
            // Part 2 in PAIRS of panels (e, e+1), e even: the far columns receive both panels in ONE pass over C with K = 2 nbo
            // (the two panels are adjacent columns of L), C is read and written once per two panels.  ev_cols[p+1] keeps its
            // meaning "the columns of panel p+2 are up to date through panel p":
            //   p even: only panel p+2's columns now (K = nbo); everything further waits for the partner panel
            //   p odd : panel p+2's columns with K = 2 nbo first (event), then all columns from panel p+3 on with K = 2 nbo,
            //           underneath which part 1 / chain of the next two steps run
            // Every tile still receives its panels in ascending order with the accumulator carried in fp64 through C: the
            // factor is bit-identical to the one-panel-per-pass schedule.
            if ((p & 1) == 0) {
                if (p + 2 < P) {                                             // a partner panel follows inside the loop
                    update_cols(su, A, npad, K0, W, pcol(p + 2), pcol(p + 3), ws);
                } else {
                    update_cols(su, A, npad, K0, W, pcol(p + 2), npad, ws);
                }
                (void)hipEventRecord(ws->ev_cols[p + 1], su);
            } else {
                const long K1 = pcol(p - 1), W2 = pcol(p + 1) - K1;
                update_cols(su, A, npad, K1, W2, pcol(p + 2), pcol(p + 3), ws);
                (void)hipEventRecord(ws->ev_cols[p + 1], su);
                update_cols(su, A, npad, K1, W2, pcol(p + 3), npad, ws);
            }
            continue;
        }
        (void)hipStreamWaitEvent(su, ws->ev_panel[p], 0);
        update_cols(su, A, npad, K0, W, pcol(p + 1), pcol(p + 2), ws);       // part 1: the next panel's columns
        (void)hipEventRecord(ws->ev_cols[p + 1], su);
        (void)hipStreamWaitEvent(sp, ws->ev_cols[p + 1], 0);
        factor_panel(sp, A, npad, pcol(p + 1), pcol(p + 2) - pcol(p + 1), ws);
        // part 2: everything to the right
        update_cols(su, A, npad, K0, W, pcol(p + 2), npad, ws);
    }
    (void)hipEventRecord(ws->ev_panel[P], sp);
    (void)hipStreamWaitEvent(st, ws->ev_panel[P], 0);
}

// X = L^-1: diagonal 128-blocks on single CUs (all blocks concurrently), then log2(nt) batched levels.  If the preceding
// potrf_device already put the leading ovl_h tiles and the top-level T21 on st_tri (same X / T buffers), only the
// trailing block and the top-level X21 = -X22 T21 are left.
void trtri_device(hipStream_t st, const double* L, double* X, double* T, long npad, FactorWs* ws) {
    const int nt = (int)(npad / NB);
    ws->prof.begin(st, PF_TRTRI, (double)npad * npad * npad / 3.0);
    const int h = ws->ovl_h;
    ws->ovl_h = 0;
    ws->evals_done += 1;
    ws->early_pending = 0;                                      // every branch below joins the side stream (ev_tri) if h > 0
    if (h > 0 && X == ws->scratchX && T == ws->scratchT && h < nt) {
        int lev = 0;
        while ((1 << lev) < h) ++lev;
        // (a) tiles [h, nt): diagonal blocks and the levels below lev
        const long off = (long)h * NB * npad + (long)h * NB;
        launch_inv128(st, L + off, X + off, npad, nt - h, ws->dinv + (long)h * 8 * 256);
        for (int level = 0; level < lev; ++level) launch_trtri_level(st, L + off, X + off, T + off, npad, nt - h, level);
        // (b) level lev, pairs >= 1 (tiles from 2h on)
        if (nt > 2 * h) {
            const long off2 = (long)2 * h * NB * npad + (long)2 * h * NB;
            launch_trtri_level(st, L + off2, X + off2, T + off2, npad, nt - 2 * h, lev);
        }
        // (c) level lev, pair 0: help drain the T21 tile list, then X21 = -X22 T21
        const int nt_pair = nt < 2 * h ? nt : 2 * h;
        (void)hipStreamWaitEvent(st, ws->ev_tri_lead, 0);
        launch_trtri_stage1_steal(st, L, X, T, npad, nt_pair, lev, ws->tri_counter, 512);
        (void)hipStreamWaitEvent(st, ws->ev_tri, 0);
        launch_trtri_level(st, L, X, T, npad, nt_pair, lev, 2);
        // (d) the levels above
        for (int level = lev + 1; (1 << level) < nt; ++level) launch_trtri_level(st, L, X, T, npad, nt, level);
    } else {
        if (h > 0) (void)hipStreamWaitEvent(st, ws->ev_tri, 0);   // early work went to other buffers: let it drain, redo all
        launch_inv128(st, L, X, npad, nt, ws->dinv);
        for (int level = 0; (1 << level) < nt; ++level) launch_trtri_level(st, L, X, T, npad, nt, level);
    }
    ws->prof.end(st);
}

// The split X^T X of a small matrix runs from a work list that lives on the device; made once per size.  hipFree / hipMalloc /
// synchronous copies: never inside a stream capture (factor_ws_alloc builds it ahead for the size it is given; a miss while a
// capture is open falls back to the single launch for that call -- ADVICE r5).
static bool lauum_plan_wanted(const FactorWs* ws, int nt) {
    return ws->lauum_split && nt <= (ws->lauum_split > 1 ? ws->lauum_split : LAUUM_SPLIT_MAX_NT) && (long)nt * NB > 1024;
}
static bool lauum_plan_build(FactorWs* ws, int nt) {
    std::vector<LauumItem> items;
    std::vector<LauumSum> sums;
    int nparts = 0;
    lauum_split_plan(nt, items, sums, &nparts);
    if (ws->lauum_plan_dev) (void)hipFree(ws->lauum_plan_dev);
    if (ws->lauum_part) (void)hipFree(ws->lauum_part);
    ws->lauum_plan_dev = nullptr;
    ws->lauum_part = nullptr;
    ws->lauum_plan_nt = 0;
    const size_t bi = sizeof(LauumItem) * items.size(), bs = sizeof(LauumSum) * sums.size();
    if (hipMalloc(&ws->lauum_plan_dev, bi + bs + 16) == hipSuccess &&
        hipMalloc(&ws->lauum_part, sizeof(double) * (size_t)lauum_split_tile(nt) * lauum_split_tile(nt) * (size_t)(nparts + 1)) ==
            hipSuccess) {
        // (pageable copies: done before they return; the plan is made once per size)
        (void)hipMemcpy(ws->lauum_plan_dev, items.data(), bi, hipMemcpyHostToDevice);
        if (bs) (void)hipMemcpy((char*)ws->lauum_plan_dev + bi, sums.data(), bs, hipMemcpyHostToDevice);
        ws->lauum_plan_nt = nt;
        ws->lauum_nitems = (int)items.size();
        ws->lauum_nsums = (int)sums.size();
        return true;
    }
    (void)hipGetLastError();
    return false;
}

void lauum_device(hipStream_t st, const double* X, double* W, long npad, FactorWs* ws) {
    const int nt = (int)(npad / NB);
    ws->prof.begin(st, PF_LAUUM, (double)npad * npad * npad / 3.0);
    // small matrix, long k ranges: the split work list (gemm.hip); the plan is made once per size and lives on the device
    bool split = lauum_plan_wanted(ws, nt);
    if (split && ws->lauum_plan_nt != nt) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
        split = (cap == hipStreamCaptureStatusNone) && lauum_plan_build(ws, nt);
    }
    if (split && ws->lauum_plan_nt == nt)
        launch_lauum_split(st, X, W, npad, nt, (const LauumItem*)ws->lauum_plan_dev, ws->lauum_nitems,
                           (const LauumSum*)((const char*)ws->lauum_plan_dev + sizeof(LauumItem) * ws->lauum_nitems), ws->lauum_nsums,
                           ws->lauum_part);
    else
        launch_lauum(st, X, W, npad, nt);
    ws->prof.end(st);
}
