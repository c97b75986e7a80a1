// internal.h -- host-side launcher declarations shared by the translation units of libmi355gp.so.
#pragma once
#include <shared_mutex>
#include <vector>

#include "common.h"

// ---- gemm.hip : tiled fp64 MFMA GEMM family (all dimensions multiples of 128) -------------------
// C[ti,tj] -= A[ti,:] * B[tj,:]^T over a (ntr x ntc)-tile region; tiles with (col0t+tj) > (row0t+ti) skipped.
bool update_nt_uses_64(int ntr, int ntc, int row0t, int col0t);   // which kernel (and profile family) a launch of this shape takes
// one "part 2" update of the blocked Cholesky as a task of the bounding experiment k_update_nt_queue: the lower triangle of
// ntr x ntr tiles at A + c_off, panel P at A + p_off (K columns), ntiles = ntr (ntr + 1) / 2
struct UpdTask { long c_off, p_off, ntiles; int K, ntr; };
#ifdef MI355GP_DIAG
void launch_update_nt_queue(hipStream_t st, double* A, long ld, const UpdTask* tasks_dev, int ntasks, int* counter, int wgs);
#endif
void launch_update_nt(hipStream_t st, double* C, long ldc, const double* A, long lda, const double* B, long ldb,
                      int K, int ntr, int ntc, int row0t, int col0t);
// one bottom-up level of the batched triangular inverse: X21 = -X22 * (L21 * X11) for every block pair
void launch_trtri_level(hipStream_t st, const double* L, double* X, double* T, long ld, int nt, int level,
                        int stages = 3);
// W (lower tiles) = X^T X for lower-triangular X
void launch_lauum(hipStream_t st, const double* X, double* W, long ld, int nt);
// X^T X of a SMALL matrix with the long k ranges cut into chunks (gemm.hip, "k_lauum64_items"): one quadrant chunk per item
struct LauumItem { int ti, tj, q, k0, klen, part; };       // part: index of the 64 x 64 partial it stores, -1: stores W itself
struct LauumSum { int ti, tj, q, first, n; };              // W quadrant = sum of partials first .. first + n - 1
void lauum_split_plan(int nt, std::vector<LauumItem>& items, std::vector<LauumSum>& sums, int* nparts);
int lauum_split_tile(int nt);                               // edge of the items' output tile: 64 (quadrants) or 128
void launch_lauum_split(hipStream_t st, const double* X, double* W, long ld, int nt, const LauumItem* items_dev, int nitems,
                        const LauumSum* sums_dev, int nsums, double* part);
bool lauum_uses_64(int nt);
void launch_trmm_lower(hipStream_t st, const double* X, long ldx, const double* B, long ldb, double* Out, long ldo,
                       int ntr, int ntc);
// Out = X^T B (X lower triangular npad x npad, B npad x mpad)
void launch_trmm_lower_T(hipStream_t st, const double* X, long ldx, const double* B, long ldb, double* Out, long ldo,
                         int ntr, int ntc);
// triangular products on 64 x 64 quadrants (X lower triangular, nt x nt tiles; the other dimension of B / Out: ntother tiles):
// mode 0 Out = alpha X B, 1 alpha X^T B, 2 alpha B X^T, 3 alpha B X -- only the non-zero k range of X is walked
void launch_trmm64(hipStream_t st, int mode, const double* X, long ldx, const double* B, long ldb, double* Out, long ldo, int nt,
                   int ntother, double alpha);
void launch_gemm_tn_sq(hipStream_t st, const double* A, long lda, long K, double* C, long ldc, int nt, double alpha,
                       double beta);
void launch_dbg_gemm(hipStream_t st, int a_mcontig, int b_ncontig, long M, long N, long K, const double* A,
                     const double* B, double* C, double alpha, double beta);

// ---- small.hip : single-CU MFMA kernels on 128x128 diagonal blocks -------------------------------
// in-place Cholesky of the 128x128 block at A[c0,c0]; writes the 8 inverses of its 16x16 diagonal tiles
// to dinv (8*256 doubles), sum(log diag) to logsum[0], first failing 1-based global column to *info.
void launch_diag128(hipStream_t st, double* A, long ld, long c0, double* dinv, double* logsum, int* info,
                    int exclusive = 0);
// rows [r0, r0+mrows) of the 128-wide panel at column c0:  P <- P * L_cc^{-T}   (mrows % 16 == 0)
void launch_trsm128(hipStream_t st, double* A, long ld, long c0, long r0, long mrows, const double* dinv);
// X_cc = L_cc^{-1} for all nblk diagonal blocks (upper tiles of the diagonal blocks of X zeroed)
// stage 1 of one level with a shared tile counter (zeroed by the caller): see k_trtri_stage1_steal
void launch_trtri_stage1_steal(hipStream_t st, const double* L, double* X, double* T, long ld, int nt, int level,
                               int* counter, int grid);
void launch_inv128(hipStream_t st, const double* L, double* X, long ld, int nblk, const double* dinv_all);
void launch_dbg_mfma(hipStream_t st, const double* a, const double* b, double* d);

// ---- factor.hip : blocked drivers -------------------------------------------------------------------
// per-kernel-family device timing (hipEvent pairs on the launching stream) + algorithmic flop counts
enum { PF_UPDATE = 0, PF_TRTRI = 1, PF_LAUUM = 2, PF_DIAG = 3, PF_TRSM = 4, PF_UPDATE64 = 5, PF_PERSIST = 6, PF_TRTRI_EARLY = 7, PF_NUM = 8 };
struct KernelProf {
    unsigned mask = 0;          // bit f set: family f is timed
    bool on = false;
    std::vector<hipEvent_t> pool;
    struct Rec { int fam; double flops; size_t e0; };
    std::vector<Rec> recs;
    bool open = false;
    void begin(hipStream_t st, int fam, double flops);
    void end(hipStream_t st);
    void reset() { recs.clear(); }
    // after the streams are synchronised: per family total ms, total flops, launch count
    int collect(double* ms, double* flops, int* launches);
    void destroy();
};

struct FactorWs {
    double* dinv = nullptr;     // nblk * 8 * 256 doubles: inverses of the 16x16 diagonal tiles
    double* logsum = nullptr;   // nblk doubles: sum(log diag L) per 128-block
    int* info = nullptr;        // device int: 0 or first failing column (1-based)
    long nblk = 0;
    hipStream_t st_panel = nullptr;          // high-priority stream of the look-ahead panel factorisation (shared engine stream)
    std::vector<hipEvent_t> ev_panel;        // [p]: outer panel p is factored
    std::vector<hipEvent_t> ev_cols;         // [p]: every update of panel p's columns has been issued (-> its factorisation)
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_early_pre = nullptr;       // "the progress words of this persistent launch are zeroed" for the early inverse's gates
    int agg2 = 0;               // MI355GP_AGG2: part 2 of the look-ahead schedule in pairs of panels (K = 2 nbo far updates; N >= 6144); measured: no gain
    // Which schedule a small factorisation takes is decided BY MEASUREMENT per workspace (MI355GP_PERSIST_AUTO=0: always the
    // persistent launch): on most boxes the persistent launch + early inverse wins at N = 4096 (2.85 against 3.3 ms per evaluation),
    // on some the very same binary runs the persistent launch at half speed (3.2-3.6 ms against 1.75 for the factorisation alone,
    // the launch-per-step schedule only 10 % slower than elsewhere).  The third evaluation of a workspace is timed on the
    // persistent schedule, the fourth on launches; the faster one stays (both give the same bits).
    int persist_auto = 1, sched_state = 0, sched_force_steps = 0, persist_auto_off = 0;
    int sched_np = 0, sched_ns = 0;  // samples taken so far: persistent schedule (two, warm) / launches (one untimed, then two)
    int can_calibrate = 0;      // set by the owner of a workspace that times its evaluations (the exact-inference contexts)
    float sched_ms_persist = 0.f, sched_ms_steps = 0.f;
    int evals_done = 0;         // inverses taken through this workspace (the early inverse under the persistent launch starts with the second)
    int early_pending = 0;      // early-inverse kernels are in flight on the side stream and nobody has joined them yet (ev_tri)
    // split lauum of a small matrix (lauum_device): the plan of the last nt it was made for, on the device
    int lauum_split = 1, lauum_plan_nt = 0, lauum_nitems = 0, lauum_nsums = 0;     // MI355GP_LAUUM_SPLIT=0 switches it off
    void* lauum_plan_dev = nullptr;  // [items | sums | partials]
    double* lauum_part = nullptr;
    int persist_tri = 1;        // MI355GP_PERSIST_TRI: leading-block inverse + T21 on the side stream UNDERNEATH the persistent launch
    int persist_tri_min_nt = 16;    // ... for factorisations of at least this many tiles (MI355GP_PERSIST_TRI_MIN_NT)
    int upd_queue_probe = 0;    // diagnostics build, MI355GP_DBG_UPD_QUEUE=1 (WRONG RESULTS): every part-2 update of the look-ahead schedule from ONE
                                // resident launch with all dependences ignored -- the bounding experiment of DESIGN.md 6f
    UpdTask* upd_tasks = nullptr;   // device copy of the task list (64 entries) + the queue counter behind it
    int part1_on_panel = 1;     // MI355GP_PART1_ON_PANEL: part 1 of a step on the panel stream (no cross-stream hop before the next chain)
    int lookahead = 1;          // 1: panel p+1 factored on st_panel while the big update of step p runs; 0: serial reference schedule
    // outer panel width of the two-level right-looking Cholesky: NBO (512) keeps the big trailing update at 64 flop per
    // byte of C traffic (measured round 2: 128 / 256 are slower at every N from 2048 to 8192; MI355GP_NBO overrides)
    int nbo_override = 0;
    long nbo_for(long npad) const {
        long w = nbo_override > 0 ? nbo_override : (npad <= FACTOR_NBO_SMALL_N ? FACTOR_NBO_SMALL : NBO);
        w = (w / NB) * NB;
        return w < NB ? NB : w;
    }
    // Scratch for the overlapped inverse: two npad x npad buffers with the same leading dimension as A that are free during
    // the factorisation (the context's X = L^-1 and T buffers of the trtri_device call that follows); nullptr: no overlap.
    double* scratchX = nullptr;
    double* scratchT = nullptr;
    // trtri of the finished leading block + the top-level T21 = L21 X11 run on st_tri while potrf's chain-bound second
    // half leaves the GPU mostly idle
    int tri_overlap = FACTOR_DEFAULT_TRI_OVERLAP, tri_cu_pct = 75, tri_min_nt = 48, tri_wgs = 0, ovl_h = 0;
    hipStream_t st_tri = nullptr, st_tri_half = nullptr, st_tri_cur = nullptr;   // 75 % / 50 % of every XCD / the one in use
    int tri_half_ok = 1, tri_cur_pct = 75;
    hipEvent_t ev_tri = nullptr, ev_tri_lead = nullptr;
    int* tri_counter = nullptr;
    int tri_h_override = 0;          // MI355GP_TRI_H: leading tiles inverted early (0 = time model)
    // the first k_diag128 of a panel starts when part 1 has just drained the GPU: with the exclusive LDS request it takes a
    // CU that no part-2 workgroup can join afterwards (27 us instead of 85-250 us next to one); small factorisations only
    int diag_excl_first = FACTOR_DEFAULT_DIAG_EXCL_FIRST, excl_first_ok = 0;
    int solve_overlap = 1;           // MI355GP_SOLVE_OVERLAP: alpha = X^T (X R) on st_tri underneath lauum
    // MI355GP_PERSIST: the single-launch dataflow Cholesky of persist.hip for factorisations of at most persist_max_nt tiles
    int persist = FACTOR_DEFAULT_PERSIST, persist_max_nt = FACTOR_PERSIST_MAX_NT, persist_kcap = 2, persist_cus = 0;
    // persist_skip: calls of potrf_device that stay on the launch-per-step schedule (set after a called-off / aborted persistent
    // launch: PS_SKIP_AFTER_CLEAN, or INT_MAX after a dirty abort); persist_aborts counts them (MI355GP_OPT_PERSIST_ABORTS)
    int persist_skip = 0, persist_aborts = 0, persist_used = 0;
    hipEvent_t ev_persist_pre = nullptr;   // optional (not owned): recorded on the launching stream once the progress words are zeroed
    int persist_grid_last = 0;             // workgroups of the last persistent launch
    int persist_tune = 0;            // MI355GP_PERSIST_TUNE: A/B bits of the persistent launch (4: no split hand-over, 64 / 128: near
                                     // ownership of 3 / 4 block diagonals; bits 8..15: share of near owners = workers / that number)
    int persist_test = 0;            // MI355GP_OPT_PERSIST_TEST: fault injection for the NEXT persistent launch (1 clean, 2 dirty)
    int* persist_sync = nullptr;     // progress words of the persistent launch (zeroed before every launch)
    double* persist_hs = nullptr;    // [min(nblk, 64)][128 x 128]: sub-diagonal tile of every row, handed to the chain in ITS load order
    KernelProf prof;
};
// Gate of a device's shared engine streams.  Entry points that enqueue on them hold it SHARED for the duration of the call
// (any number of host threads at once -- the loopback transports rendezvous inside such calls); a hipGraph capture window
// on the shared streams holds it EXCLUSIVELY, so that no other thread's launch is recorded into the graph instead of run.
std::shared_mutex& engine_gate(int device);
struct EngineShared {
    std::shared_mutex* m;
    bool shared = true;
    explicit EngineShared(int device) : m(&engine_gate(device)) { m->lock_shared(); }
    EngineShared(const EngineShared&) = delete;
    EngineShared& operator=(const EngineShared&) = delete;
    ~EngineShared() {
        if (shared) m->unlock_shared();
        else m->unlock();
    }
    // Upgrade for a capture window WITHOUT blocking: other shared holders may be loopback ranks parked in a rendezvous whose
    // peer thread still has to take lock_shared() -- behind a pending writer (a writer-preferring shared_mutex is allowed by
    // the standard) that would never be granted.  false: somebody else is inside; the caller skips the capture this time.
    bool try_exclusive() {
        m->unlock_shared();
        if (m->try_lock()) {
            shared = false;
            return true;
        }
        m->lock_shared();
        return false;
    }
    void share() {
        m->unlock();
        m->lock_shared();
        shared = true;
    }
};
// the process-wide main / panel / tri streams of a device (created on first use, shared, never destroyed)
int factor_engine(int device, hipStream_t* main, hipStream_t* panel, hipStream_t* tri, hipStream_t* tri_half = nullptr);
int factor_ws_alloc(FactorWs* ws, long npad);
void factor_ws_free(FactorWs* ws);
// A (npad x npad, ld = npad, lower) -> L in place.  Asynchronous; on return all work is ordered before later work on `st`.
void potrf_device(hipStream_t st, double* A, long npad, FactorWs* ws);
// X = L^-1 (into X, using T as scratch), asynchronous.
void trtri_device(hipStream_t st, const double* L, double* X, double* T, long npad, FactorWs* ws);
// W = X^T X (lower tiles)
void lauum_device(hipStream_t st, const double* X, double* W, long npad, FactorWs* ws);

// ---- persist.hip : the whole factorisation of a small matrix as one persistent dataflow launch ------------------------
bool potrf_persist_eligible(long npad, const FactorWs* ws);
// What the calibrated workspaces of this process found for machine-filling factorisations (nt >= 21): -1 nothing measured yet,
// 0 the persistent launch wins on this box, 1 launches win.  Workspaces that never calibrate (the one-shot pdinv / jitchol
// entry points, the M x M factorisations of the sparse path) follow it.
int persist_box_verdict(int set = -2);
// bookkeeping after the host has read info[0] of a factorisation: true if it is one of the PS_ABORT codes (then persist_skip /
// persist_aborts are updated and the caller redoes the factorisation -- on the untouched matrix if *clean, after
// rebuilding it otherwise)
bool potrf_persist_aborted(int info, FactorWs* ws, bool* clean);
int potrf_persist_sync_ints();
// dbg (optional, 8 * nt wall-clock stamps): per chain step [factor start, factor end, sub tile seen, solve end, diag tile seen, update end]
// false: the launch could not be made (no large-LDS opt-in on this device, launch error): nothing was enqueued that writes A
bool launch_potrf_persist(hipStream_t st, double* A, long npad, FactorWs* ws, long long* dbg = nullptr);
// on `st`: wait (<= 2 ms) until the persistent launch announced by ws->ev_persist_pre has all its workgroups resident
void launch_wait_persist_resident(hipStream_t st, const FactorWs* ws, int timeout_ms = 2);
int persist_early_h(long npad, const FactorWs* ws);
// on `st`: one thread that returns once rows r0 .. r1-1 of L are final in their first `cols` tile columns (cols = 0: the whole
// row up to and including the diagonal block and its inverted diagonal tiles), the launch was called off / aborted, or 20 ms passed
void launch_wait_persist_rows(hipStream_t st, const FactorWs* ws, int r0, int r1, int cols, int give_up = 0);

// ---- kern.hip : covariance assembly, reductions, solves, fetch helpers ----------------------------
struct KernParams {
    int kind;
    int ard;
    int D;
    double variance;
};
// Xt: scaled, transposed inputs [D][ldx] (x_q / l_q); builds lower tiles of Ky = K + diag(noise + jit) into A
// (npad x npad); rows/cols >= n get the identity.
void launch_scale_inputs(hipStream_t st, const double* X, long n, int D, const double* inv_ls, int ard,
                         double* Xt, long ldx);
// accumulate != 0: add this kernel's covariance to what A / Kout already hold (sum kernels);
// mul (same layout as the output, may alias it): element-wise multiplier applied first (product kernels)
void launch_kbuild_sym(hipStream_t st, KernParams kp, const double* Xt, long ldx, long n, long npad, double* A,
                       const double* noise, long noise_len, double jit, int lower_only, int add_diag,
                       int accumulate = 0, const double* mul = nullptr);
void launch_kbuild_cross(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2,
                         long ld2, long m, double* Kout, long ldk, int accumulate = 0, int diag_same = 0,
                         const double* mul = nullptr);
// K(X1, X2) into Kout AND the column partials of psi1^T V = sum_i K[i][j] V[i][d] in one pass (one plain stationary part);
// returns the number of row splits in colpart ([split][mcols][Dy]; combine with launch_sum_splits), 0 = not applicable
int launch_kbuild_cols(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2, long ld2, long m,
                       long mcols, double* Kout, long ldk, const double* V, int Dy, double* colpart);
// y = X r (lower-triangular X, n x n within npad), then a = X^T y   (Dy right-hand sides, row-major n x Dy)
void launch_tri_matvec(hipStream_t st, const double* X, long ld, long n, const double* R, int Dy, double* tmp,
                       double* alpha, double* partials);
// fused gradient reduction over the lower tiles of W; results (per block partials) reduced by launch_finalize
struct GradOut {
    double* partials;   // [nblocks][stride]
    int stride;
    int nblocks;
};
int grad_num_blocks(long n);
// aa_scale (optional, device): factor on the alpha alpha^T term of dL_dK (Student-t process)
void launch_grad_fused(hipStream_t st, KernParams kp, const double* Xt, long ldx, long n, const double* W,
                       long ldw, const double* alpha, int Dy, double* partials, int stride,
                       const double* aa_scale = nullptr, const double* Mul = nullptr, long ldm = 0);
// optional on-the-fly form of the weight matrix read by launch_grad_generic:
//   g = rowscale[i] * (gscale * G + beta * sum_d Y[i][d] V[j][d])        (rowscale == NULL: 1)
struct RankTerm {
    const double* Y;
    const double* V;
    int Dy;
    double beta, gscale;
    const double* rowscale = nullptr;
};
// sparse pass 2: theta partials + H^T [x~ | 1] column partials in one pass over the weights (k_grad_cols); 0 = not applicable
int launch_grad_cols(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2, long ld2,
                     long m, long mcols, const double* G, long ldg, RankTerm rk, double* partials, double* colpart,
                     int* nblocks_out);
void launch_studentt_scale(hipStream_t st, const double* scal, double nu, long n, double* out);
// Hout (optional, may alias G): H = dL_dK * (dK/dr)/r, the weights of the gradients_X reductions
void launch_grad_generic(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2,
                         long ld2, long m, int symmetric, const double* G, long ldg, double* partials,
                         int stride, double* Hout = nullptr, long ldh = 0,
                         RankTerm rk = RankTerm{nullptr, nullptr, 0, 0.0, 1.0, nullptr});
// part[split][cols][nv] = sum over a row range of M[i][j] * V(i, c); V(i, c) = V[i*sr + c*sc] plus an optional
// all-ones column; returns the number of row splits (sum them with launch_sum_splits)
int launch_colreduce_multi(hipStream_t st, const double* M, long ld, long rows, long cols, const double* V, long sr,
                           long sc, int nvt, int ones, double* part);
void launch_sum_splits(hipStream_t st, const double* src, long cnt, int nsplit, int accumulate, double* dst);
int trmv_chunk_rows(long n);        // rows per partial-sum chunk of launch_trmv_lower_T / launch_tri_matvec (partials: ceil(n / rows) * n * Dy)
void launch_trmv_lower(hipStream_t st, const double* X, long ld, long n, const double* R, int Dy, double* y);
void launch_trmv_lower_T(hipStream_t st, const double* X, long ld, long n, const double* y, int Dy, double* out,
                         double* partials);
int gemm_last_clock(double* mhz, double* cycles);
// general tiled GEMM C = alpha*op(A) op(B) + beta*C; M, N % 128 == 0, K % 16 == 0 (see k_gemm_full)
void launch_gemm(hipStream_t st, int a_mcontig, int b_ncontig, long M, long N, long K, const double* A, long lda,
                 const double* B, long ldb, double* C, long ldc, double alpha, double beta);
// split-K Gram matrix of a tall panel: part[s] (lower 128-tiles of an mp x mp matrix, ld = mp) (+)= P_s^T P_s, where P_s
// are rows [s*rows/S, (s+1)*rows/S) of the (rows x mp) row-major panel P; rows % (16*S) == 0
void launch_gram_splitk(hipStream_t st, const double* P, long ldp, long rows, long mp, int S, int accumulate,
                        double* part);
int grad_generic_num_blocks(long n, long m);
// sums `nblocks` rows of `stride` doubles in a fixed order into out[stride]
void launch_reduce_partials(hipStream_t st, const double* partials, int nblocks, int stride, double* out);
// out4[0]=sum alpha*R ; [1]=sum alpha^2 ; [2]=trace W ; [3]=2*sum(logsum) ; [6]=info[0] ; diag_out (n) = 0.5*(|alpha_i|^2 - Dy*W_ii)
void launch_scalars(hipStream_t st, const double* alpha, const double* R, const double* W, long ldw, long n,
                    int Dy, const double* logsum, long nblk, double* out4, double* diag_out, const int* info = nullptr);
// dense n x n host-shaped outputs from padded device matrices
//   mode 0: lower triangle of A, strict upper zero;  1: symmetric mirror of lower(A);
//   2: 0.5*(s * alpha alpha^T - Dy * sym(A)), s = aa_scale[0] (device) or 1;  transpose != 0 writes the transpose
void launch_extract(hipStream_t st, const double* A, long ld, long n, int mode, const double* alpha, int Dy,
                    double* out, int transpose, const double* aa_scale = nullptr);
void launch_pad_from_dense(hipStream_t st, const double* src, long n, double* A, long npad, const double* noise,
                           long noise_len, double jit);
// column reductions over a (rows x ld) matrix: mode 0: out[j*Dy+d] = sum_i M[i][j]*v[i*Dy+d]; mode 1: out[j] = c0 - sum_i M[i][j]^2
void launch_col_reduce(hipStream_t st, const double* M, long ld, long rows, long cols, const double* v, int Dy,
                       double c0, int mode, double* out);

// out_s[i][d] = sum_j K[i][j] v[j][d]  and (t_out != NULL)  t_out[i] = sum_j K[i][j] T[i][j]   over j < m, one wave per row
void launch_rowdots(hipStream_t st, const double* K, const double* T, long ld, long rows, long m, const double* v, int Dy,
                    double* out_s, double* t_out);
// M[i][j] *= sqrt(w[i]) into Out (may alias M), rows x cols (ld shared)
void launch_rowscale_sqrt(hipStream_t st, const double* M, long ld, long rows, long cols, const double* w, double* Out);

// ---- grid.hip : RCCL (dlopen'ed) world communicator for row-sharded paths ---------------------------------------
int rccl_comm_create(int rank, int world, const void* id128, void** comm);
int rccl_allreduce_sum(void* comm, double* buf, size_t count, hipStream_t st);
void rccl_comm_destroy(void* comm);
