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

int factor_engine(int device, hipStream_t* main, hipStream_t* panel, hipStream_t* tri, hipStream_t* tri_half) {
    if (device < 0 || device >= 16) return -1;
    std::lock_guard<std::mutex> lock(g_engine_mutex);
    FactorEngine& e = g_engine[device];
    if (!e.ready) {
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithFlags(&e.main, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithPriority(&e.panel, hipStreamNonBlocking, greatest));
        const char* envc = getenv("MI355GP_TRI_CU_PCT");
        e.tri_pct = (envc && *envc) ? atoi(envc) : 75;
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount, nx = 8;
        if (e.tri_pct > 0 && e.tri_pct < 100 && ncu % nx == 0) {
            // XCD-balanced mask (logical CU i sits on XCD i % 8): the same share of every XCD's CUs
            const int per = ncu / nx, keep = (per * e.tri_pct + 50) / 100;
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            const char* envm = getenv("MI355GP_TRI_MASK_MODE");
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
        e.ready = true;
    }
    if (main) *main = e.main;
    if (panel) *panel = e.panel;
    if (tri) *tri = e.tri;
    if (tri_half) *tri_half = e.tri_half;
    return 0;
}

// ---- workspace --------------------------------------------------------------------------------------
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
        const char* envo = getenv("MI355GP_TRI_OVERLAP");
        if (envo && *envo) ws->tri_overlap = atoi(envo) ? 1 : 0;
        const char* envn2 = getenv("MI355GP_TRI_MIN_NT");
        if (envn2 && *envn2) ws->tri_min_nt = atoi(envn2);
        const char* envw2 = getenv("MI355GP_TRI_WGS");
        if (envw2 && *envw2) ws->tri_wgs = atoi(envw2);
        ws->tri_cu_pct = g_engine[dev].tri_pct > 0 && g_engine[dev].tri_pct <= 100 ? g_engine[dev].tri_pct : 75;
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_tri, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_tri_lead, hipEventDisableTiming));
        HIP_CHECK(hipMalloc(&ws->tri_counter, sizeof(int) * 4));
        const char* envh = getenv("MI355GP_TRI_H");
        if (envh && *envh) ws->tri_h_override = atoi(envh);
    }
    {
        // Express lane for the panel chain: the big trailing updates run on a stream whose CU mask leaves `reserve_cus`
        // CUs out, so k_diag128 (one workgroup, dependent fp64 VALU chain) never shares a CU with an fp64-MFMA-saturating
        // update workgroup (measured: 42 us alone, 125-300 us when co-resident).  MI355GP_RESERVE_CUS=0 disables it.
        const char* envr = getenv("MI355GP_RESERVE_CUS");
        ws->reserve_cus = (envr && *envr) ? atoi(envr) : FACTOR_DEFAULT_RESERVE_CUS;
        hipDeviceProp_t prop;
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        const int ncu = prop.multiProcessorCount;
        if (ws->reserve_cus > 0 && ws->reserve_cus < ncu / 2) {
            // Workgroups are dealt round-robin to the 8 XCDs, so the mask must take the SAME number of CUs from every XCD
            // (an XCD with fewer CUs becomes the straggler of every launch: measured 2x slower).  Logical CU i sits on
            // XCD i % 8, so the first 8*r indices are r CUs per XCD.
            const int nx = 8, r = (ws->reserve_cus + nx - 1) / nx;
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int cu = 0; cu < ncu; ++cu)
                if (cu >= nx * r) mask[cu / 32] |= 1u << (cu % 32);
            hipError_t e = hipExtStreamCreateWithCUMask(&ws->st_bulk, (uint32_t)mask.size(), mask.data());
            if (e != hipSuccess) {
                (void)hipGetLastError();
                ws->st_bulk = nullptr;
            }
        }
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_bulk, hipEventDisableTiming));
        const char* enve = getenv("MI355GP_DIAG_EXCL");
        if (enve && *enve) ws->diag_excl_opt = atoi(enve) ? 1 : 0;
    }
    {
        const char* envs = getenv("MI355GP_PANEL_SPLIT");
        if (envs && *envs) ws->panel_split = atoi(envs) ? 1 : 0;
        if (ws->panel_split) HIP_CHECK(hipStreamCreateWithPriority(&ws->st_rest, hipStreamNonBlocking, greatest));
        for (int i = 0; i < 4; ++i) {
            HIP_CHECK(hipEventCreateWithFlags(&ws->ev_d[i], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&ws->ev_t[i], hipEventDisableTiming));
        }
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_rest, hipEventDisableTiming));
    }
    const char* envnbo = getenv("MI355GP_NBO");
    if (envnbo && *envnbo) ws->nbo_override = atoi(envnbo);
    const char* envw = getenv("MI355GP_PART2_WGS");
    if (envw && *envw) ws->part2_wgs = atoi(envw);
    const char* envt = getenv("MI355GP_PART2_TILES");
    if (envt && *envt) ws->part2_tiles = atol(envt);
    const char* envf = getenv("MI355GP_PANEL_FUSED");
    if (envf && *envf) ws->panel_fused = atoi(envf) ? 1 : 0;
    {
        const char* envs2 = getenv("MI355GP_DIAG_SERVER");
        if (envs2 && *envs2) ws->diag_server = atoi(envs2) ? 1 : 0;
        HIP_CHECK(hipMalloc(&ws->diag_flags, sizeof(int) * 2 * ws->nblk));
        HIP_CHECK(hipMemset(ws->diag_flags, 0, sizeof(int) * 2 * ws->nblk));
        if (ws->diag_server) HIP_CHECK(hipStreamCreateWithPriority(&ws->st_diag, hipStreamNonBlocking, greatest));
        HIP_CHECK(hipEventCreateWithFlags(&ws->ev_diag, hipEventDisableTiming));
    }
    const char* envt2 = getenv("MI355GP_TRSM_LDS");
    if (envt2 && *envt2) ws->trsm_lds = atoi(envt2);           // 0: operands from L2, 1: LDS-staged, 2: LDS-staged, two strips per wave
    const char* envso = getenv("MI355GP_SOLVE_OVERLAP");
    if (envso && *envso) ws->solve_overlap = atoi(envso) ? 1 : 0;
    const char* envth = getenv("MI355GP_TRI_HALF");
    if (envth && *envth) ws->tri_half_ok = atoi(envth) ? 1 : 0;
    const char* envx = getenv("MI355GP_DIAG_EXCL_FIRST");
    if (envx && *envx) ws->diag_excl_first = atoi(envx) ? 1 : 0;
    const char* envr = getenv("MI355GP_PANEL_REC");
    if (envr && *envr) ws->panel_rec = atoi(envr) ? 1 : 0;
    const char* envn = getenv("MI355GP_PANEL_FUSED_MAX_NRB");
    if (envn && *envn) ws->panel_fused_max_nrb = atoi(envn);
    const char* envm = getenv("MI355GP_PANEL_FUSED_MIN_NRB");
    if (envm && *envm) ws->panel_fused_min_nrb = atoi(envm);
    const char* envg = getenv("MI355GP_PANEL_WGS");
    if (envg && *envg) ws->panel_max_wgs = atoi(envg);
    {
        const char* envd = getenv("MI355GP_PANEL_DBG");
        if (envd && atoi(envd)) {
            HIP_CHECK(hipMalloc(&ws->panel_dbg, sizeof(long long) * 256 * 16));
            HIP_CHECK(hipMemset(ws->panel_dbg, 0, sizeof(long long) * 256 * 16));
        }
    }
    HIP_CHECK(hipMalloc(&ws->panel_flags, sizeof(int) * 32));
    HIP_CHECK(hipMemset(ws->panel_flags, 0, sizeof(int) * 32));
    ws->panel_gen = 0;
    const char* envp = getenv("MI355GP_PANEL_INV");
    if (envp && *envp) ws->panel_inv = atoi(envp) ? 1 : 0;
    const char* env = getenv("MI355GP_UPD_STREAMS");
    if (env && *env) ws->n_upd = atoi(env);
    if (ws->n_upd < 1) ws->n_upd = 1;
    if (ws->n_upd > FactorWs::MAX_UPD) ws->n_upd = FactorWs::MAX_UPD;
    for (int i = 0; i < ws->n_upd; ++i) HIP_CHECK(hipEventCreateWithFlags(&ws->ev_join[i], hipEventDisableTiming));
    // the chunk-update streams themselves are created by potrf_chunked on first use (option LOOKAHEAD = 2 only)
    HIP_CHECK(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
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
    if (ws->panel_dbg) (void)hipFree(ws->panel_dbg);
    ws->panel_dbg = nullptr;
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
    for (int i = 0; i < FactorWs::MAX_UPD; ++i) {
        if (ws->ev_join[i]) (void)hipEventDestroy(ws->ev_join[i]);
        if (ws->st_upd[i]) (void)hipStreamDestroy(ws->st_upd[i]);
        ws->ev_join[i] = nullptr;
        ws->st_upd[i] = nullptr;
    }
    ws->st_panel = nullptr;
    if (ws->st_rest) (void)hipStreamDestroy(ws->st_rest);
    ws->st_rest = nullptr;
    for (int i = 0; i < 4; ++i) {
        if (ws->ev_d[i]) (void)hipEventDestroy(ws->ev_d[i]);
        if (ws->ev_t[i]) (void)hipEventDestroy(ws->ev_t[i]);
        ws->ev_d[i] = ws->ev_t[i] = nullptr;
    }
    if (ws->ev_rest) (void)hipEventDestroy(ws->ev_rest);
    ws->ev_rest = nullptr;
    if (ws->st_bulk) (void)hipStreamDestroy(ws->st_bulk);
    ws->st_bulk = nullptr;
    if (ws->panel_flags) (void)hipFree(ws->panel_flags);
    ws->panel_flags = nullptr;
    if (ws->diag_flags) (void)hipFree(ws->diag_flags);
    ws->diag_flags = nullptr;
    ws->st_tri = nullptr;
    ws->st_tri_half = ws->st_tri_cur = nullptr;
    if (ws->ev_tri) (void)hipEventDestroy(ws->ev_tri);
    ws->ev_tri = nullptr;
    if (ws->ev_tri_lead) (void)hipEventDestroy(ws->ev_tri_lead);
    ws->ev_tri_lead = nullptr;
    if (ws->tri_counter) (void)hipFree(ws->tri_counter);
    ws->tri_counter = nullptr;
    if (ws->st_diag) (void)hipStreamDestroy(ws->st_diag);
    ws->st_diag = nullptr;
    if (ws->ev_diag) (void)hipEventDestroy(ws->ev_diag);
    ws->ev_diag = nullptr;
    if (ws->ev_bulk) (void)hipEventDestroy(ws->ev_bulk);
    ws->ev_bulk = nullptr;
    ws->prof.destroy();
}

// algorithmic flops of a rank-K update of the lower triangle (incl. diagonal) of an (n x n) block / of an (m x n) block
static double syrk_flops(double n, double K) { return K * n * (n + 1.0); }
static double gemm_flops(double m, double n, double K) { return 2.0 * m * n * K; }

// One outer panel: columns [K0, K0+W), rows [K0, npad).  128-column steps:
//   diag128 (one CU) -> trsm128 on the rows below -> rank-128 update of the remaining columns of the panel.
static void factor_panel_inv(hipStream_t s, double* A, long npad, long K0, long W, FactorWs* ws);
static void factor_panel_split(hipStream_t s, double* A, long npad, long K0, long W, FactorWs* ws);
static void factor_panel(hipStream_t s, double* A, long npad, long K0, long W, FactorWs* ws) {
    if (ws->panel_inv && ws->scratchX && ws->scratchT) {
        factor_panel_inv(s, A, npad, K0, W, ws);
        return;
    }
    if (ws->panel_split && ws->st_rest && ws->lookahead == 1 && W <= 4 * NB) {
        factor_panel_split(s, A, npad, K0, W, ws);
        return;
    }
    const long ld = npad;
    if (ws->panel_fused && ws->panel_flags && W <= 4 * NB && (npad - K0) / NB <= ws->panel_fused_max_nrb &&
        (npad - K0) / NB >= ws->panel_fused_min_nrb) {
        const int ns = (int)(W / NB), nrb = (int)((npad - K0) / NB);
        ws->prof.begin(s, PF_DIAG, (double)ns * NB * NB * NB / 3.0);
        const int rc = launch_panel_fused(s, A, ld, K0, ns, nrb, ws->dinv + (K0 / NB) * 8 * 256, ws->logsum + K0 / NB,
                                          ws->info, ws->panel_flags, ++ws->panel_gen, ws->panel_max_wgs,
                                          K0 == 0 ? ws->panel_dbg : nullptr);
        ws->prof.end(s);
        if (rc == 0 && K0 == 0 && ws->panel_dbg) {          // diagnostics: print the hand-off timeline of the first panel
            (void)hipStreamSynchronize(s);
            const int G = nrb < ws->panel_max_wgs ? nrb : ws->panel_max_wgs;
            std::vector<long long> h((size_t)G * 16);
            (void)hipMemcpy(h.data(), ws->panel_dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
            long long t0 = h[0];
            for (int g = 0; g < G; ++g) t0 = (h[(size_t)g * 16] && h[(size_t)g * 16] < t0) ? h[(size_t)g * 16] : t0;
            for (int g = 0; g < G; g += (g < 6 ? 1 : (G / 6 > 0 ? G / 6 : 1))) {
                fprintf(stderr, "[panel dbg] npad=%ld wg %3d:", npad, g);
                for (int k = 0; k < ns; ++k) {
                    fprintf(stderr, " |");
                    for (int i = 0; i < 4; ++i) fprintf(stderr, " %7.1f", (double)(h[((size_t)g * 4 + k) * 4 + i] - t0) / 100.0);
                }
                fprintf(stderr, "\n");
            }
        }
        if (rc == 0) return;
    }
    // one 128-column step: diagonal block, then the rows below it
    auto leaf = [&](long c) {
        const long blk = c / NB;
        double* dv = ws->dinv + blk * 8 * 256;
        ws->prof.begin(s, PF_DIAG, (double)NB * NB * NB / 3.0);
        if (ws->diag_server_on)
            launch_diag_call(s, ws->diag_flags, ws->diag_flags + ws->nblk, (int)blk, ws->diag_gen, ws->info);
        else
            launch_diag128(s, A, ld, c, dv, ws->logsum + blk, ws->info,
                           ws->diag_excl || (ws->diag_excl_first && c == K0 && ws->lookahead == 1 && ws->excl_first_ok));
        ws->prof.end(s);
        const long below = npad - (c + NB);
        if (below <= 0) return;
        ws->prof.begin(s, PF_TRSM, (double)below * NB * NB);
        launch_trsm128(s, A, ld, c, c + NB, below, dv, ws->trsm_lds);
        ws->prof.end(s);
    };
    // rank-K update of the panel's columns [cd, cend) (rows cd .. npad) with its columns [c0, cd)
    auto inpanel_update = [&](long c0, long cd, long cend) {
        const long below = npad - cd, ncols = cend - cd, K = cd - c0;
        if (below <= 0 || ncols <= 0) return;
        const double* P = A + cd * ld + c0;
        ws->prof.begin(s, PF_UPDATE, syrk_flops((double)ncols, (double)K) + gemm_flops((double)(below - ncols), (double)ncols, (double)K));
        launch_update_nt(s, A + cd * ld + cd, ld, P, ld, P, ld, (int)K, (int)(below / NB), (int)(ncols / NB), (int)(cd / NB),
                         (int)(cd / NB));
        ws->prof.end(s);
    };
    if (!ws->panel_rec) {                    // right-looking inside the panel: every step updates all remaining columns, K = 128
        for (long j = 0; j < W; j += NB) {
            leaf(K0 + j);
            inpanel_update(K0 + j, K0 + j + NB, K0 + W);
        }
        return;
    }
    // Recursive inside the panel (default): [left half] -> update of the right half with K = width of the left half ->
    // [right half].  Same launches and flops as the right-looking loop, but the K=128 read-modify-write passes over the
    // panel's C tiles shrink from 384+256+128 to 128+256+128 columns (the 256-column one at K=256).
    struct Rec {
        decltype(leaf)& lf;
        decltype(inpanel_update)& up;
        void run(long c0, int ns) {
            if (ns == 1) {
                lf(c0);
                return;
            }
            int h = 1;
            while (2 * h < ns) h *= 2;
            run(c0, h);
            up(c0, c0 + (long)h * NB, c0 + (long)ns * NB);
            run(c0 + (long)h * NB, ns - h);
        }
    } rec{leaf, inpanel_update};
    rec.run(K0, (int)(W / NB));
}

// Inverse-based panel (needs ws->scratchX / scratchT): the chain kernels that must win workgroup slots against a
// machine-filling trailing update shrink from 4 x (trsm128 + in-panel update) over the whole panel height to ONE GEMM.
//   D phase : the W x W diagonal block is factored with the 128-step loop restricted to its own rows (<= 6-tile kernels)
//   I phase : XD = L_D^-1 (inv128 + log2(W/128) batched levels) into scratchX at the block's own position
//   R phase : rows below:  L_R = R * XD^T  (k_panel_trmm) into scratchT, copied back into A
// rocprof: the trsm128-based chain takes 0.75 ms per panel on an idle GPU but ~1.7 ms under a trailing update.
static void factor_panel_inv(hipStream_t s, double* A, long npad, long K0, long W, FactorWs* ws) {
    const long ld = npad, end = K0 + W;
    for (long j = 0; j < W; j += NB) {
        const long c = K0 + j, blk = c / NB;
        double* dv = ws->dinv + blk * 8 * 256;
        ws->prof.begin(s, PF_DIAG, (double)NB * NB * NB / 3.0);
        launch_diag128(s, A, ld, c, dv, ws->logsum + blk, ws->info, ws->diag_excl);
        ws->prof.end(s);
        const long below = end - (c + NB);                      // rows of the diagonal block still to do
        if (below <= 0) continue;
        ws->prof.begin(s, PF_TRSM, (double)below * NB * NB);
        launch_trsm128(s, A, ld, c, c + NB, below, dv, ws->trsm_lds);
        ws->prof.end(s);
        double* C = A + (c + NB) * ld + (c + NB);
        const double* P = A + (c + NB) * ld + c;
        ws->prof.begin(s, PF_UPDATE, syrk_flops((double)below, NB));
        launch_update_nt(s, C, ld, P, ld, P, ld, NB, (int)(below / NB), (int)(below / NB), (int)((c + NB) / NB),
                         (int)((c + NB) / NB));
        ws->prof.end(s);
    }
    const long rows = npad - end;
    if (rows <= 0) return;
    double* XD = ws->scratchX + K0 * ld + K0;
    double* TD = ws->scratchT + K0 * ld + K0;
    const int nt = (int)(W / NB);
    (void)hipMemset2DAsync(XD, sizeof(double) * ld, 0, sizeof(double) * W, W, s);      // upper blocks of XD must be zero
    ws->prof.begin(s, PF_TRTRI, (double)W * W * W / 3.0);
    launch_inv128(s, A + K0 * ld + K0, XD, ld, nt, ws->dinv + (K0 / NB) * 8 * 256);
    for (int level = 0; (1 << level) < nt; ++level) launch_trtri_level(s, A + K0 * ld + K0, XD, TD, ld, nt, level);
    ws->prof.end(s);
    double* R = A + end * ld + K0;
    double* Rt = ws->scratchT + end * ld + K0;
    ws->prof.begin(s, PF_TRSM, (double)rows * W * W);
    launch_panel_trmm(s, R, XD, Rt, ld, (int)(rows / NB), nt);
    ws->prof.end(s);
    (void)hipMemcpy2DAsync(R, sizeof(double) * ld, Rt, sizeof(double) * ld, sizeof(double) * W, rows,
                           hipMemcpyDeviceToDevice, s);
}

// Split panel (experiment, MI355GP_PANEL_SPLIT=1; measured slower: potrf 33.1 -> 33.6 ms at N=16384, 3.85 -> 4.19 ms at
// N=4096, the cross-stream event waits cost more than the pipelining hides): the dependency chain of a panel only runs
// through its W x W diagonal block
//   chain stream `s` : for each 128-column step  diag128 -> trsm of the diagonal block's remaining rows -> K=128 update
//                      of the diagonal block's remaining tiles          (kernels of <= 6 workgroups)
//   rest stream      : trsm of all rows BELOW the diagonal block (after that step's diag128) -> K=128 update of those
//                      rows' remaining panel columns (after that step's small trsm), pipelined behind the chain
// so diag128(j+1) no longer waits for the wide kernels of step j.  `s` waits for the rest stream before returning.
static void factor_panel_split(hipStream_t s, double* A, long npad, long K0, long W, FactorWs* ws) {
    const long ld = npad, end = K0 + W, rows_below = npad - end;
    hipStream_t sr = ws->st_rest;
    int nsteps = 0;
    for (long j = 0; j < W; j += NB, ++nsteps) {
        const long c = K0 + j, blk = c / NB;
        double* dv = ws->dinv + blk * 8 * 256;
        ws->prof.begin(s, PF_DIAG, (double)NB * NB * NB / 3.0);
        launch_diag128(s, A, ld, c, dv, ws->logsum + blk, ws->info, ws->diag_excl);
        ws->prof.end(s);
        const long in_block = end - (c + NB);                  // rows of the diagonal block below this step
        if (rows_below > 0) {
            (void)hipEventRecord(ws->ev_d[nsteps], s);
            (void)hipStreamWaitEvent(sr, ws->ev_d[nsteps], 0);
            launch_trsm128(sr, A, ld, c, end, rows_below, dv);
        }
        if (in_block <= 0) continue;
        ws->prof.begin(s, PF_TRSM, (double)in_block * NB * NB);
        launch_trsm128(s, A, ld, c, c + NB, in_block, dv, ws->trsm_lds);
        ws->prof.end(s);
        const double* Pd = A + (c + NB) * ld + c;               // freshly solved rows of the diagonal block
        if (rows_below > 0) {
            (void)hipEventRecord(ws->ev_t[nsteps], s);
            (void)hipStreamWaitEvent(sr, ws->ev_t[nsteps], 0);
            launch_update_nt(sr, A + end * ld + (c + NB), ld, A + end * ld + c, ld, Pd, ld, NB, (int)(rows_below / NB),
                             (int)(in_block / NB), (int)(end / NB), (int)((c + NB) / NB));
        }
        ws->prof.begin(s, PF_UPDATE, syrk_flops((double)in_block, NB));
        launch_update_nt(s, A + (c + NB) * ld + (c + NB), ld, Pd, ld, Pd, ld, NB, (int)(in_block / NB),
                         (int)(in_block / NB), (int)((c + NB) / NB), (int)((c + NB) / NB));
        ws->prof.end(s);
    }
    if (rows_below > 0) {
        (void)hipEventRecord(ws->ev_rest, sr);
        (void)hipStreamWaitEvent(s, ws->ev_rest, 0);
    }
}

// rank-W update of the trailing columns [c0, c1) (rows c0 .. npad) with the panel at columns [K0, K0+W)
static void update_cols(hipStream_t s, double* A, long npad, long K0, long W, long c0, long c1, FactorWs* ws,
                        int max_wgs = 0) {
    if (c1 <= c0) return;
    const long ld = npad, rows = npad - c0, cols = c1 - c0;
    const double* P = A + c0 * ld + K0;
    ws->prof.begin(s, PF_UPDATE, syrk_flops((double)cols, (double)W) + gemm_flops((double)(rows - cols), (double)cols, (double)W));
    launch_update_nt(s, A + c0 * ld + c0, ld, P, ld, P, ld, (int)W, (int)(rows / NB), (int)(cols / NB), (int)(c0 / NB),
                     (int)(c0 / NB), max_wgs);
    ws->prof.end(s);
}

// Alternative schedule (option LOOKAHEAD = 2; measured equal to the default on MI355X, kept as an experiment switch):
// the true dependencies at column-chunk granularity instead of whole steps on one stream:
//   - panel p is factored on a high-priority stream as soon as the updates of ITS columns are done (look-ahead);
//   - the trailing columns are cut into chunks of ~512 tiles; chunk c is always updated on stream c % n_upd, so
//     step p+1's update of a chunk only waits for panel p+1 and for the same chunk's step-p update.  The tail of
//     one launch (fewer tiles left than CU slots) therefore overlaps the head of the next chunk's launch, and
//     the latency-bound diag/trsm chain hides behind MFMA-bound work.
static void potrf_chunked(hipStream_t st, double* A, long npad, FactorWs* ws) {
    (void)hipMemsetAsync(ws->info, 0, sizeof(int) * 4, st);
    const long nbo = ws->nbo_for(npad);
    const long P = (npad + nbo - 1) / nbo;                       // outer panels
    auto pcol = [&](long p) { return (p * nbo < npad) ? p * nbo : npad; };
    if (!ws->lookahead) {                                        // reference schedule: everything in order on st
        for (long p = 0; p < P; ++p) {
            factor_panel(st, A, npad, pcol(p), pcol(p + 1) - pcol(p), ws);
            update_cols(st, A, npad, pcol(p), pcol(p + 1) - pcol(p), pcol(p + 1), npad, ws);
        }
        return;
    }
    // chunk boundaries (in panels): chunk i = panels [cb[i], cb[i+1]); at least ~448 tiles of 128x128 each
    std::vector<long> cb;
    {
        const long nt = npad / NB;
        long tiles = 0;
        cb.push_back(1);
        for (long p = 1; p < P; ++p) {
            const long t0 = pcol(p) / NB, t1 = pcol(p + 1) / NB;
            for (long t = t0; t < t1; ++t) tiles += nt - t;
            if (tiles >= 448 && p + 1 < P) { cb.push_back(p + 1); tiles = 0; }
        }
        cb.push_back(P);
    }
    const int nchunk = (int)cb.size() - 1;
    for (int i = 0; i < ws->n_upd; ++i)
        if (!ws->st_upd[i]) (void)hipStreamCreateWithFlags(&ws->st_upd[i], hipStreamNonBlocking);
    auto chunk_stream = [&](int c) { return ws->st_upd[c % ws->n_upd]; };
    hipStream_t sp = ws->st_panel;
    (void)hipEventRecord(ws->ev_fork, st);                      // everything queued on st so far precedes the factorisation
    (void)hipStreamWaitEvent(sp, ws->ev_fork, 0);
    for (int i = 0; i < ws->n_upd; ++i) (void)hipStreamWaitEvent(ws->st_upd[i], ws->ev_fork, 0);
    for (long p = 0; p < P; ++p) {
        const long K0 = pcol(p), W = pcol(p + 1) - K0;
        if (p > 0) (void)hipStreamWaitEvent(sp, ws->ev_cols[p], 0);
        factor_panel(sp, A, npad, K0, W, ws);
        if (p + 1 >= P) break;
        (void)hipEventRecord(ws->ev_panel[p], sp);
        for (int i = 0; i < ws->n_upd; ++i) (void)hipStreamWaitEvent(ws->st_upd[i], ws->ev_panel[p], 0);
        for (int c = 0; c < nchunk; ++c) {
            if (cb[c + 1] <= p + 1) continue;                   // chunk already factored
            hipStream_t su = chunk_stream(c);
            long first = (cb[c] > p + 1) ? cb[c] : p + 1;
            if (first == p + 1) {                               // the next panel's columns first: they gate panel p+1
                update_cols(su, A, npad, K0, W, pcol(p + 1), pcol(p + 2), ws);
                (void)hipEventRecord(ws->ev_cols[p + 1], su);
                first = p + 2;
            }
            update_cols(su, A, npad, K0, W, pcol(first), pcol(cb[c + 1]), ws);
        }
    }
    for (int i = 0; i < ws->n_upd; ++i) {
        (void)hipEventRecord(ws->ev_join[i], ws->st_upd[i]);
        (void)hipStreamWaitEvent(st, ws->ev_join[i], 0);
    }
    (void)hipEventRecord(ws->ev_panel[P], sp);
    (void)hipStreamWaitEvent(st, ws->ev_panel[P], 0);
}

// Two-level right-looking Cholesky with one panel of look-ahead (default).  Outer panels of NBO = 512 columns keep
// the big trailing update at K = 512 (64 flop per byte of C traffic).  The update of outer step p is split into the
// next panel's 512 columns (part 1) and the rest (part 2); panel p+1 is factored on a second, high-priority stream
// while part 2 of step p still runs, so the latency-bound diag/trsm chain hides behind MFMA-bound work.  All
// trailing updates stay in order on `st`: their launch durations are not inflated by overlapping each other.
void potrf_device(hipStream_t st, double* A, long npad, FactorWs* ws) {
    if (ws->lookahead != 1) {
        ws->diag_excl = 0;
        ws->diag_server_on = 0;
        potrf_chunked(st, A, npad, ws);                          // 0: serial reference schedule, 2: chunk streams
        return;
    }
    ws->diag_excl = (ws->st_bulk != nullptr && ws->diag_excl_opt) ? 1 : 0;
    ws->diag_server_on = (ws->diag_server && ws->diag_flags && ws->st_diag && !ws->panel_inv && !ws->panel_split &&
                          !ws->panel_fused && npad / NB == ws->nblk) ? 1 : 0;
    (void)hipMemsetAsync(ws->info, 0, sizeof(int) * 4, st);
    const long nbo = ws->nbo_for(npad);
    const long P = (npad + nbo - 1) / nbo;
    auto pcol = [&](long p) { return (p * nbo < npad) ? p * nbo : npad; };
    hipStream_t sp = ws->st_panel;
    hipStream_t su = ws->st_bulk ? ws->st_bulk : st;            // trailing updates (CU-masked when an express lane is set)
    (void)hipEventRecord(ws->ev_fork, st);                      // panel 0 follows everything queued on st so far
    (void)hipStreamWaitEvent(sp, ws->ev_fork, 0);
    if (ws->diag_server_on) {                                   // resident diagonal-block server for this factorisation
        ++ws->diag_gen;
        (void)hipStreamWaitEvent(ws->st_diag, ws->ev_fork, 0);
        launch_diag_server(ws->st_diag, A, npad, (int)(npad / NB), ws->dinv, ws->logsum, ws->info, ws->diag_flags,
                           ws->diag_flags + ws->nblk, ws->diag_gen);
        (void)hipEventRecord(ws->ev_diag, ws->st_diag);
    }
    if (su != st) (void)hipStreamWaitEvent(su, ws->ev_fork, 0);
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
    for (long p = 0; p + 1 < P; ++p) {
        const long K0 = pcol(p), W = pcol(p + 1) - K0;
        (void)hipEventRecord(ws->ev_panel[p], sp);
        if (ovl_h > 0 && pcol(p + 1) == (long)ovl_h * NB) {       // columns < 128 h are final: start on their inverse
            hipStream_t sq = ws->st_tri_cur ? ws->st_tri_cur : ws->st_tri;
            (void)hipStreamWaitEvent(sq, ws->ev_panel[p], 0);
            (void)hipMemsetAsync(ws->tri_counter, 0, sizeof(int) * 4, sq);
            launch_inv128(sq, A, ws->scratchX, npad, ovl_h, ws->dinv);
            int level = 0;
            for (; (1 << level) < ovl_h; ++level) launch_trtri_level(sq, A, ws->scratchX, ws->scratchT, npad, ovl_h, level);
            (void)hipEventRecord(ws->ev_tri_lead, sq);
            // pair 0 of level log2(h): T21 = L21 X11 (rows h .. 2h of the finished columns), tile list shared with the
            // machine-wide instance that trtri_device launches after potrf
            const int nt_pair = ntl < 2 * ovl_h ? ntl : 2 * ovl_h;
            launch_trtri_stage1_steal(sq, A, ws->scratchX, ws->scratchT, npad, nt_pair, level, ws->tri_counter,
                                      ws->tri_wgs > 0 ? ws->tri_wgs : 512 * ws->tri_cur_pct / 100);
            (void)hipEventRecord(ws->ev_tri, sq);
            ws->ovl_h = ovl_h;
        }
        (void)hipStreamWaitEvent(su, ws->ev_panel[p], 0);
        update_cols(su, A, npad, K0, W, pcol(p + 1), pcol(p + 2), ws);       // part 1: the next panel's columns
        (void)hipEventRecord(ws->ev_cols[p + 1], su);
        (void)hipStreamWaitEvent(sp, ws->ev_cols[p + 1], 0);
        factor_panel(sp, A, npad, pcol(p + 1), pcol(p + 2) - pcol(p + 1), ws);
        // part 2: everything to the right (optionally with a bounded number of resident workgroups, see part2_wgs)
        const long t2 = (npad - pcol(p + 2)) / NB, tiles2 = t2 * (t2 + 1) / 2;
        update_cols(su, A, npad, K0, W, pcol(p + 2), npad, ws,
                    (ws->part2_wgs > 0 && tiles2 < ws->part2_tiles) ? ws->part2_wgs : 0);
    }
    (void)hipEventRecord(ws->ev_panel[P], sp);
    (void)hipStreamWaitEvent(st, ws->ev_panel[P], 0);
    if (ws->diag_server_on) (void)hipStreamWaitEvent(st, ws->ev_diag, 0);
    if (su != st) {
        (void)hipEventRecord(ws->ev_bulk, su);
        (void)hipStreamWaitEvent(st, ws->ev_bulk, 0);
    }
}

// X = L^-1: diagonal 128-blocks on single CUs (all blocks concurrently), then log2(nt) batched levels.  If the preceding
// potrf_device already put the leading ovl_h tiles and the top-level T21 on st_tri (same X / T buffers), only the
// trailing block and the top-level X21 = -X22 T21 are left.
void trtri_device(hipStream_t st, const double* L, double* X, double* T, long npad, FactorWs* ws) {
    const int nt = (int)(npad / NB);
    ws->prof.begin(st, PF_TRTRI, (double)npad * npad * npad / 3.0);
    const int h = ws->ovl_h;
    ws->ovl_h = 0;
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

void lauum_device(hipStream_t st, const double* X, double* W, long npad, FactorWs* ws) {
    ws->prof.begin(st, PF_LAUUM, (double)npad * npad * npad / 3.0);
    launch_lauum(st, X, W, npad, (int)(npad / NB));
    ws->prof.end(st);
}
