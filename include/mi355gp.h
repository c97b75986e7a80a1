/* mi355gp.h -- C-ABI of libmi355gp.so: the MI355X-native (gfx950, hand-written HIP) exact-GP backend.
 *
 * GPy has no FFI: its hot path is three Python call signatures.  This header is the
 * boundary a maintainer binds with ctypes; every entry point names the
 * reference function(s) it replaces (paths relative to the GPy checkout).
 *
 * Conventions
 *   - all matrices fp64; host arrays are C-contiguous (row-major) unless a `fortran_order` flag says otherwise
 *   - return value: 0 = OK; >0 = LAPACK-style `info` (1-based index of the first non-positive pivot of the
 *     Cholesky factorisation); <0 = argument / HIP error, text via mi355gp_last_error()
 *   - `kind`: covariance function, `ard`: 0 = one lengthscale, 1 = one per input dimension
 *   - `theta` = [variance, lengthscale (1 or D values)]  (the order GPy links them: kern/src/stationary.py:78-81)
 *   - gradient outputs use the same order; values/gradients are untransformed (paramz applies Logexp outside)
 *   - one context = one device + one data set (X, Y); contexts are independent, not thread-safe individually
 */
#ifndef MI355GP_H
#define MI355GP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355gp_ctx mi355gp_ctx;

enum { MI355GP_RBF = 0, MI355GP_MATERN52 = 1, MI355GP_MATERN32 = 2, MI355GP_EXPONENTIAL = 3,
       MI355GP_WHITE = 4, MI355GP_BIAS = 5 /* static kernels, only as parts of a sum (kern/src/static.py:63-98,151-173) */ };

/* One part of a sum-of-products kernel expression (GPy.kern.Add, kern/src/add.py:58-84; GPy.kern.Prod,
 * kern/src/prod.py:58-99).  theta = [variance, lengthscale (1, or n_active if ard)] (static kinds: [variance]);
 * active_dims: the input columns this part acts on (kern/src/kern.py:49-53,112-117), NULL / n_active == 0 = all columns.
 * term: 0 = the part is a summand of its own; parts sharing the same NONZERO term id are multiplied element-wise and
 * their product is one summand: K = sum_t prod_{f in t} k_f.  Gradients come back per part, in part order. */
typedef struct {
    int kind;
    int ard;
    int n_active;
    const int* active_dims;
    const double* theta;
    int term;
} mi355gp_part;

/* which device-resident matrix mi355gp_fetch() materialises on the host (all N x N) */
enum {
    MI355GP_FETCH_L = 0,      /* Cholesky factor of Ky, strict upper triangle zero  (Posterior.woodbury_chol) */
    MI355GP_FETCH_KINV = 1,   /* Ky^-1, symmetric                                    (Posterior.woodbury_inv)  */
    MI355GP_FETCH_DLDK = 2,   /* dL_dK = 0.5*(alpha alpha^T - Dy*Ky^-1), symmetric   (grad_dict['dL_dK'])      */
    MI355GP_FETCH_K = 3       /* K(X,X) without noise/jitter, symmetric              (Posterior._K)            */
};

/* out_scalars[] layout of mi355gp_exact_inference / mi355gp_inference_given_K */
enum {
    MI355GP_OUT_LML = 0,        /* log marginal likelihood (without Z_tilde)            */
    MI355GP_OUT_LOGDET = 1,     /* log det Ky                                           */
    MI355GP_OUT_DATAFIT = 2,    /* sum(alpha * R)                                       */
    MI355GP_OUT_DNOISE = 3,     /* dL/d sigma_n^2 = trace(dL_dK)                        */
    MI355GP_OUT_TRKINV = 4,     /* trace(Ky^-1)                                         */
    MI355GP_NUM_OUT = 8
};

/* stage_ms[] layout (hipEvent timings of the last inference call, milliseconds).  Passing stage_ms makes the call run plain
 * launches: below N = 6144 a context otherwise replays its factorisation region (potrf, trtri, alpha solve, lauum) from a hipGraph
 * captured at its second evaluation (MI355GP_GRAPH=0 disables it), and graph nodes carry no timing events. */
enum {
    MI355GP_T_KBUILD = 0, MI355GP_T_POTRF = 1, MI355GP_T_TRTRI = 2, MI355GP_T_LAUUM = 3,
    MI355GP_T_SOLVE = 4, MI355GP_T_GRAD = 5, MI355GP_T_TOTAL = 6, MI355GP_NUM_T = 8
};

/* ---- library / device ------------------------------------------------------------------------- */
const char* mi355gp_last_error(void);
const char* mi355gp_version(void);
int mi355gp_device_count(int* count);
/* hipDeviceSynchronize on `device`: everything the library has enqueued there is complete on return.  (Every compute entry
   point already returns after its own stream has drained; this is the device-wide fence a timing harness brackets a region
   with -- bench.py -- without importing a second HIP runtime user into the process.) */
int mi355gp_device_synchronize(int device);

/* ---- context ------------------------------------------------------------------------------------ */
int mi355gp_create(int device, mi355gp_ctx** ctx);
int mi355gp_destroy(mi355gp_ctx* ctx);

/* Uploads the training set once (X is constant across optimiser iterations: core/gp.py:44-60).
 * X: N x D, R: N x Dy (= Y - mean_function.f(X), exact_gaussian_inference.py:42-50). */
int mi355gp_set_data(mi355gp_ctx* ctx, const double* X, int64_t N, int D, const double* R, int Dy);
/* replaces only the targets (same N, Dy) */
int mi355gp_set_targets(mi355gp_ctx* ctx, const double* R, int Dy);

/* ---- kernel functions (replace Stationary.K / Kdiag / update_gradients_full) ------------------------ */
/* K(X, X2) -> K_out (N x M, row-major).  X2 == NULL: symmetric case (M = N, exact `variance` on the diagonal).
 * Replaces kern/src/stationary.py:105-168 (+ K_of_r: rbf.py:51-52, stationary.py:382-383,488-489,585-586). */
int mi355gp_kern_K(int device, int kind, int ard, const double* theta, const double* X, int64_t N,
                   const double* X2, int64_t M, int D, double* K_out);
/* Kdiag(X) (kern/src/stationary.py:170-173) */
int mi355gp_kern_Kdiag(int kind, const double* theta, int64_t N, double* out);
/* dL/dtheta from a caller-supplied dL_dK (N x M, row-major, need not be symmetric).
 * Replaces Stationary.update_gradients_full (kern/src/stationary.py:193-243) incl. the native
 * lengthscale_grads loop (kern/src/stationary_cython.pyx:53-62).  dtheta_out: 1 + (ard ? D : 1). */
int mi355gp_update_gradients_full(int device, int kind, int ard, const double* theta, const double* dL_dK,
                                  const double* X, int64_t N, const double* X2, int64_t M, int D,
                                  double* dtheta_out);

/* dL/dX (N x D) from dL_dK (N x M): Stationary.gradients_X (kern/src/stationary.py:245-252,330-358, native loop
 * kern/src/stationary_utils.c:1-14).  X2 == NULL: the symmetric form (tmp + tmp^T against X itself).  Any D (the reductions run in groups of 32 dimensions). */
int mi355gp_gradients_X(int device, int kind, int ard, const double* theta, const double* dL_dK, const double* X,
                        int64_t N, const double* X2, int64_t M, int D, double* out);

/* ---- the fused hot path ---------------------------------------------------------------------------- */
/* One GP.parameters_changed (core/gp.py:278-280) with everything N x N resident in HBM:
 *   K = kern.K(X); Ky = K + (noise + jitter + extra_jitter) I; L = chol(Ky); alpha = Ky^-1 R;
 *   LML; Ky^-1; dL_dK (never materialised); dL/dtheta; dL/dnoise; diag(dL_dK).
 * Replaces ExactGaussianInference.inference (inference/latent_function_inference/exact_gaussian_inference.py:37-74),
 * pdinv/jitchol/dpotrs/dpotri/dtrtri/tdot/symmetrify (util/linalg.py:56-75,116-145,193-227,299-379),
 * Gaussian.exact_inference_gradients (likelihoods/gaussian.py:78-79) and
 * Stationary.update_gradients_full (kern/src/stationary.py:193-243) applied to that dL_dK.
 *   noise: noise_len == 1 (homoscedastic) or N values; jitter: the reference's 1e-8 (exact_gaussian_inference.py:56);
 *   extra_jitter: the jitchol ladder term (util/linalg.py:66-72), 0 on the first attempt;
 *   out_scalars[MI355GP_NUM_OUT]; alpha_out (N x Dy) / dtheta_out / diag_dLdK_out (N) / stage_ms[MI355GP_NUM_T] may be NULL.
 * Returns info > 0 when Ky is not positive definite (the caller runs the ladder). */
int mi355gp_exact_inference(mi355gp_ctx* ctx, int kind, int ard, const double* theta,
                            const double* noise, int64_t noise_len, double jitter, double extra_jitter,
                            double* out_scalars, double* alpha_out, double* dtheta_out,
                            double* diag_dLdK_out, double* stage_ms);

/* The same evaluation for a SUM of kernels: Ky = sum_p K_p(X) + (noise + jitter) I.  dtheta_out is the concatenation of the
 * parts' gradients, each [variance, lengthscale(s)] ([variance] for White / Bias), in part order -- the order
 * GPy's Add links its parts (kern/src/add.py:24-45). */
int mi355gp_exact_inference_sum(mi355gp_ctx* ctx, int nparts, const mi355gp_part* parts, const double* noise,
                                int64_t noise_len, double jitter, double extra_jitter, double* out_scalars,
                                double* alpha_out, double* dtheta_out, double* diag_dLdK_out, double* stage_ms);

/* Student-t PROCESS inference (ExactStudentTInference.inference, exact_studentt_inference.py:20-52) for a sum of parts:
 * Ky = K + jitter I (the reference's 1e-8), dL_dK = 0.5 ((nu+N)/(nu+beta-2) alpha alpha^T - Dy Ky^-1), beta = sum(alpha*R).
 * out_scalars: [LML] Student-t log marginal, [LOGDET], [DATAFIT] = beta, [5] = (nu+N)/(nu+beta-2) (dL_dm = that * alpha). */
int mi355gp_exact_studentt_sum(mi355gp_ctx* ctx, int nparts, const mi355gp_part* parts, double nu, double jitter,
                               double extra_jitter, double* out_scalars, double* alpha_out, double* dtheta_out,
                               double* stage_ms);

/* Same with a caller-supplied covariance matrix (the `K=` argument of ExactGaussianInference.inference,
 * exact_gaussian_inference.py:52-53; used by EP and by foreign kernels).  K_host: N x N row-major. No dtheta. */
int mi355gp_inference_given_K(mi355gp_ctx* ctx, const double* K_host, const double* noise, int64_t noise_len,
                              double jitter, double extra_jitter, double* out_scalars, double* alpha_out,
                              double* diag_dLdK_out, double* stage_ms);

/* Lazy materialisation of a device-resident N x N result of the last inference call. */
int mi355gp_fetch(mi355gp_ctx* ctx, int which, double* out, int fortran_order);

/* Posterior prediction on device (PosteriorExact._raw_predict, inference/latent_function_inference/posterior.py:273-302):
 * mu (M x Dy) = K(Xnew,X) alpha; full_cov == 0: var (M) = Kdiag - sum((L^-1 Kx)^2, 0); else var (M x M). */
int mi355gp_predict(mi355gp_ctx* ctx, int kind, int ard, const double* theta, const double* Xnew, int64_t M,
                    double* mu_out, double* var_out, int full_cov);

/* the same for a sum kernel (White parts contribute to Kdiag only, as GPy's White.K(X, X2) = 0: static.py:77-81) */
int mi355gp_predict_sum(mi355gp_ctx* ctx, int nparts, const mi355gp_part* parts, const double* Xnew, int64_t M,
                        double* mu_out, double* var_out, int full_cov);

/* GP.predictive_gradients (core/gp.py:418-474) for a sum of stationary (+ White / Bias) parts, no product terms, any D:
 *   dmu_out  (M x D x Dy, row-major)  d mean[m][d] / d Xnew[m][q] = kern.gradients_X(woodbury_vector[:, d]^T, Xnew, X)  (:448-451)
 *   dvar_out (M x D)                  d var[m]    / d Xnew[m][q] = gradients_X_diag (0 for stationary kernels,
 *                                     stationary.py:360-361) + kern.gradients_X(-2 K(Xnew, X) Ky^-1, Xnew, X)           (:454,462-465)
 * Ky^-1 K(X, Xnew) is formed on the device as X^T (X K(X, Xnew)) with X = L^-1 from the last inference call; nothing N x N or
 * N x M crosses PCIe.  Either output may be NULL. */
int mi355gp_predictive_gradients_sum(mi355gp_ctx* ctx, int nparts, const mi355gp_part* parts, const double* Xnew, int64_t M,
                                     double* dmu_out, double* dvar_out);

/* Posterior covariance between two point sets (Posterior.covariance_between_points, posterior.py:109-130) */
int mi355gp_covariance_between_points(mi355gp_ctx* ctx, int nparts, const mi355gp_part* parts, const double* X1,
                                      int64_t M1, const double* X2, int64_t M2, double* out);

/* ---- standalone dense routines (dpotrf / dpotri equivalents; also what bench.py times in isolation) --- */
/* In-place lower Cholesky of a host matrix (row-major N x N, lower triangle read), strict upper zeroed on return.
 * Replaces lapack.dpotrf(A, lower=1) (util/linalg.py:58).  ms (optional): device time of the factorisation. */
int mi355gp_potrf(int device, double* A, int64_t N, double* ms);
/* Ainv (symmetric, full) from A; replaces pdinv's dpotrf+dpotri+symmetrify (util/linalg.py:193-214). */
int mi355gp_pdinv(int device, const double* A, int64_t N, double* Ainv, double* L_out, double* logdet, double* ms);
/* the same with the third member of GPy's pdinv tuple, Li = L^-1 (dtrtri, linalg.py:207): (Ai, L, Li, logdet) */
int mi355gp_pdinv_full(int device, const double* A, int64_t N, double* Ainv, double* L_out, double* Li_out, double* logdet,
                       double* ms);

/* Options of ONE context, set through the ABI (the MI355GP_* environment variables of DESIGN.md 6e only provide the
 * process-wide DEFAULTS that a context starts from; an option set here survives mi355gp_set_data).
 * PROFILE: bracket launches of the factorisation kernels with hipEvents on the launching stream (adds ~2 us per timed
 *   launch).  value 0 = off, 1 = all families, otherwise (bitmask of 1 << MI355GP_PF_*) << 1;
 * LOOKAHEAD: 1 (default) factor panel k+1 on a second, high-priority stream while the trailing update of step k still
 *   runs; 0 = everything in order on one stream (reference schedule);
 * the schedule switches (value -1 = back to the process default): TRI_OVERLAP 0/1 inverse of the leading block underneath
 *   potrf; TRI_MIN_NT smallest number of 128-tiles for that overlap (and for PART1_ON_PANEL); TRI_H leading tiles inverted
 *   early (0 = time model); TRI_WGS workgroups of the early T21 instance (0 = CU share); TRI_HALF 0/1 allow the
 *   two-shader-engine side stream; PART1_ON_PANEL 0/1; NBO outer panel width (multiple of 128, 0 = default 512);
 *   SOLVE_OVERLAP 0/1 alpha underneath lauum; DIAG_EXCL_FIRST 0/1; GRAPH 0/1 hipGraph replay of the factorisation region
 *   below the overlap threshold; PERSIST 0/1 the single-launch dataflow Cholesky for small N. */
enum { MI355GP_OPT_PROFILE = 0, MI355GP_OPT_LOOKAHEAD = 1, MI355GP_OPT_TRI_OVERLAP = 2, MI355GP_OPT_TRI_MIN_NT = 3,
       MI355GP_OPT_TRI_H = 4, MI355GP_OPT_TRI_WGS = 5, MI355GP_OPT_TRI_HALF = 6, MI355GP_OPT_PART1_ON_PANEL = 7,
       MI355GP_OPT_NBO = 8, MI355GP_OPT_SOLVE_OVERLAP = 9, MI355GP_OPT_DIAG_EXCL_FIRST = 10, MI355GP_OPT_GRAPH = 11,
       MI355GP_OPT_PERSIST = 12, MI355GP_OPT_AGG2 = 13, MI355GP_OPT_PERSIST_TEST = 14, MI355GP_OPT_PERSIST_ABORTS = 15,
       MI355GP_OPT_PERSIST_SKIP = 16, MI355GP_OPT_PERSIST_SCHED = 17, MI355GP_OPT_NUM = 18 };
/* MI355GP_OPT_PERSIST_TEST: test hook, consumed by the NEXT persistent launch of the context: 1 = the launch waits for one
   workgroup more than it has (called off at the co-residency gate, matrix untouched), 2 = the chain workgroup aborts after the
   gate (dirty abort, matrix rebuilt), 3 = the first gate kernel of the early inverse underneath the launch gives up at once (what
   its 20 ms limit does on a GPU shared with something heavy: the evaluation is marked aborted because the kernels behind the gate
   read unfinished rows); either way the evaluation is redone on the launch-per-step schedule inside the same call.
   MI355GP_OPT_PERSIST_ABORTS / _SKIP: read-only (mi355gp_get_option): persistent launches of this context that did not
   complete / evaluations that still stay on the launch-per-step schedule because of the last one.
   MI355GP_OPT_PERSIST_SCHED: read-only: the schedule of a small factorisation this context has settled on, by its own timing of
   the two or by an explicit PERSIST option: 0 = not decided yet (the first evaluations),
   1 = the persistent launch, 2 = launches (this box runs them faster, or PERSIST = 0). */
int mi355gp_set_option(mi355gp_ctx* ctx, int option, int value);
/* the value in effect for a schedule switch (options above MI355GP_OPT_PROFILE) */
int mi355gp_get_option(mi355gp_ctx* ctx, int option, int* value);
/* kernel families of mi355gp_get_profile */
enum { MI355GP_PF_UPDATE = 0 /* k_update_nt<4,true>: trailing update of potrf on 128 x 128 tiles */, MI355GP_PF_TRTRI = 1, MI355GP_PF_LAUUM = 2,
       MI355GP_PF_DIAG = 3 /* k_diag128 */, MI355GP_PF_TRSM = 4 /* k_trsm128 */,
       MI355GP_PF_UPDATE64 = 5 /* k_update_nt64: the same update on 64 x 64 tiles (launches of few tiles) */,
       MI355GP_PF_PERSIST = 6 /* k_potrf_persist: the whole factorisation of a small matrix as one persistent launch */,
       MI355GP_PF_TRTRI_EARLY = 7 /* the part of the triangular inverse that runs on the CU-masked side stream UNDERNEATH potrf
                                     (inverse of the leading block + its share of T21 = L21 X11): elapsed time on that stream;
                                     MI355GP_PF_TRTRI is the part exposed after potrf, and carries the flops of both */,
       MI355GP_PF_NUM = 8 };
/* Per family, for the last inference call made with PROFILE on: summed launch durations (ms), summed ALGORITHMIC
 * flops of those launches, launch count.  Arrays of MI355GP_PF_NUM. */
int mi355gp_get_profile(mi355gp_ctx* ctx, double* ms, double* flops, int* launches);

/* ---- optional multi-GPU mode: 2D block-cyclic factorisation on a Pr x Pc process grid ---------------------------
 * (north_star config 4; no GPy counterpart -- GPy's only parallel code is the mpi4py sparse GP,
 *  inference/latent_function_inference/var_dtc_parallel.py).  One process per GPU; rank = pr*Pc + pc.
 * Rank 0 obtains a 128-byte RCCL id with mi355gp_grid_unique_id and ships it to the others by any host channel
 * (bench.py / gpy_amd/grid.py use torch.distributed or a file); every rank then calls mi355gp_grid_create with it.
 * id128 == NULL selects the LOOPBACK transport: all Pr*Pc logical ranks live in this process on `device`
 * (correctness testing on one GPU); `rank` is ignored then.  nb: tile edge, multiple of 128. */
typedef struct mi355gp_grid mi355gp_grid;
int mi355gp_grid_unique_id(void* id128);
int mi355gp_grid_create(int device, int rank, int world, int Pr, int Pc, int nb, const void* id128, mi355gp_grid** out);
int mi355gp_grid_destroy(mi355gp_grid* g);
/* every rank passes the full X (N x D) and R (N x Dy) */
int mi355gp_grid_set_data(mi355gp_grid* g, const double* X, int64_t N, int D, const double* R, int Dy);
/* same contract and outputs as mi355gp_exact_inference; results are replicated on every rank.  Collective.
 * stage_ms: KBUILD, POTRF (= the whole one-pass factorisation + inversion), SOLVE, GRAD, TOTAL. */
int mi355gp_grid_exact_inference(mi355gp_grid* g, int kind, int ard, const double* theta, const double* noise,
                                 int64_t noise_len, double jitter, double extra_jitter, double* out_scalars,
                                 double* alpha_out, double* dtheta_out, double* diag_dLdK_out, double* stage_ms);
/* tiles owned by this process's ranks, written at their global position of an N x N row-major array (lower triangle);
 * which = MI355GP_FETCH_L or MI355GP_FETCH_LINV */
#define MI355GP_FETCH_LINV 100
int mi355gp_grid_fetch(mi355gp_grid* g, int which, double* out);
/* Schedule options of one grid (every rank must set the same values; takes effect at the next inference call).
 *   MI355GP_GRID_OPT_LOOKAHEAD: 1 = critical path one group ahead on a second stream (default), 0 = in order on one stream
 *   MI355GP_GRID_OPT_G        : steps per group of the two-level blocked factorisation: tiles beyond the group receive the
 *                               group's panels in ONE pass with K = G * nb (default 1 = one rank-nb update per step; measured best on the loopback grid)
 *   MI355GP_GRID_OPT_GW       : steps per update of Kinv = X^T X; default 4; 0 = one deep-K pass after the last step
 *   MI355GP_GRID_OPT_CHECK_SEQ: 1 = after every evaluation compare, inside every communicator, the members' logs of the
 *                               collectives they enqueued (operation, root, size, in order): RCCL matches collectives by order,
 *                               so a divergence is a dead-lock or a wrong panel in waiting; error -7 names the ranks
 * value -1 restores the process default (environment MI355GP_GRID_LOOKAHEAD / _G / _GW / _CHECK_SEQ, else the built-in one). */
enum { MI355GP_GRID_OPT_LOOKAHEAD = 0, MI355GP_GRID_OPT_G = 1, MI355GP_GRID_OPT_GW = 2, MI355GP_GRID_OPT_CHECK_SEQ = 3,
       MI355GP_GRID_OPT_NUM = 4 };
int mi355gp_grid_set_option(mi355gp_grid* g, int option, int value);
int mi355gp_grid_get_option(mi355gp_grid* g, int option, int* value);
/* The collective log of the LAST evaluation of logical rank `rank` (loopback transport: any rank of the grid; one rank per
 * process: the caller's own): out9 = [collectives on the world / row / column communicator, then per communicator an FNV-1a hash
 * over (operation, root, doubles) as (high 32 bits, low 32 bits)]. */
int mi355gp_grid_coll_log(mi355gp_grid* g, int rank, double* out9);
/* ---- sparse GP (VarDTC) path: BASELINE config 5 -----------------------------------------------------------------
 * One SparseGP.parameters_changed (core/sparse_gp.py:76-119) for certain inputs and a homoscedastic Gaussian likelihood:
 * VarDTC.inference (inference/latent_function_inference/var_dtc.py:66-215, helpers :217-276) + the kernel and
 * inducing-input gradient assembly of SparseGP._update_gradients (sparse_gp.py:108-118), streamed over row chunks of X
 * like the reference's own VarDTC_minibatch (var_dtc_parallel.py:72-133).  X: N x D, Y: N x Dy, Z: M x D.
 *   out_scalars: [0] log marginal likelihood, [1] dL/d(noise variance), [2] trace(A), [3] data_fit,
 *                [4] sum(log diag LB), [5] beta
 *   dtheta_out: 1 + (ard ? D : 1) (variance, lengthscales; diag + Knm + Kmm terms summed as sparse_gp.py:110-115)
 *   dZ_out: M x D = gradients_X(dL_dKmm, Z) + gradients_X(dL_dKnm^T, Z, X) (sparse_gp.py:116-118)
 *   wv_out (optional): woodbury_vector M x Dy;  stage_ms (optional, 4): pass 1, M x M algebra, pass 2, total
 * Returns info > 0 if Kmm or B is not positive definite (the caller runs the jitchol ladder via extra_jitter). */
typedef struct mi355gp_sparse mi355gp_sparse;
int mi355gp_sparse_create(int device, mi355gp_sparse** out);
int mi355gp_sparse_destroy(mi355gp_sparse* s);
int mi355gp_sparse_set_data(mi355gp_sparse* s, const double* X, int64_t N, int D, const double* Y, int Dy);
int mi355gp_vardtc_inference(mi355gp_sparse* s, int kind, int ard, const double* theta, const double* Z, int64_t M,
                             double noise_var, double extra_jitter, double* out_scalars, double* dtheta_out,
                             double* dZ_out, double* wv_out, double* stage_ms);
/* The same evaluation for a SUM of kernel parts (GPy.kern.Add of stationary / White / Bias kernels with active_dims,
 * kern/src/add.py:58-88, static.py:63-98,151-173; product terms are not accepted here), scalar OR per-point noise
 * variances (heteroscedastic precision: var_dtc.py:78-86,126-129,224-226,240-256,267-269; any Dy) and targets
 * R = Y - mean_function.f(X) uploaded by mi355gp_sparse_set_data (var_dtc.py:73-76,88-89).
 *   noise: noise_len == 1 or N (the rows of this context) noise VARIANCES
 *   out_scalars: as above; [1] (dL/d noise variance) only for noise_len == 1
 *   dtheta_out: concatenation over the parts, each [variance, lengthscale(s)] ([variance] for White / Bias)
 *   dnoise_rows_out (N x Dy, required for noise_len == N): dL_dR per point and output column = what
 *                                likelihood.exact_inference_gradients receives (var_dtc.py:240-256 as written there)
 *   dLdm_out (optional, N x Dy): dL_dm = beta R - K(X, Z) woodbury_vector (var_dtc.py:148; SparseGP hands it to
 *                                mean_function.update_gradients, sparse_gp.py:84-85) */
int mi355gp_vardtc_inference_sum(mi355gp_sparse* s, int nparts, const mi355gp_part* parts, const double* Z, int64_t M,
                                 const double* noise, int64_t noise_len, double extra_jitter, double* out_scalars,
                                 double* dtheta_out, double* dZ_out, double* wv_out, double* dnoise_rows_out,
                                 double* dLdm_out, double* stage_ms);
/* Sparse posterior prediction on the device (Posterior._raw_predict, inference/latent_function_inference/posterior.py:198-262,
 * for the (woodbury_inv, woodbury_vector) posterior of var_dtc.py:213): mu (Mn x Dy) = K(X*, Z) woodbury_vector;
 * full_cov == 0: var (Mn) = Kdiag - sum(Kx * (woodbury_inv Kx), 0), clipped at 1e-15 (posterior.py:248); else Mn x Mn.
 * `parts` = the kernel of the last inference call. */
int mi355gp_sparse_predict(mi355gp_sparse* s, int nparts, const mi355gp_part* parts, const double* Xnew, int64_t Mn,
                           double* mu_out, double* var_out, int full_cov);
/* Rows [row0, row0 + nrows) of dL_dKnm (nrows x M, row-major) of the last call -- the matrix SparseGP._update_gradients
 * hands to a foreign kernel's update_gradients_full / gradients_X (core/sparse_gp.py:108-118).  It is never resident as a
 * whole (3.3 GB at config 5): the caller walks it in row blocks. */
int mi355gp_sparse_fetch_dLdKnm(mi355gp_sparse* s, int64_t row0, int64_t nrows, double* out);
/* Row-sharded multi-GPU mode (the reference's MPI design, var_dtc_parallel.py:121-130,387-394): every rank holds a
 * slice of the rows of (X, Y); psi2/psi1Y after pass 1 and the gradient sums after pass 2 are all-reduced over RCCL,
 * the M x M algebra is replicated.  Call on every rank before set_data (id128: mi355gp_grid_unique_id on rank 0). */
int mi355gp_sparse_attach_comm(mi355gp_sparse* s, int rank, int world, const void* id128);
/* The same mode over the LOOPBACK transport: `world` contexts of ONE process (one host thread each) that name the same
 * group_key rendezvous for every exchange step and are summed in rank order -- world > 1 on a single GPU (tests). */
int mi355gp_sparse_attach_loopback(mi355gp_sparse* s, int rank, int world, int group_key);
/* M x M results of the last call: 0 dL_dKmm, 1 woodbury_inv (var_dtc.py:206-210), 2 Lm, 3 Kmm (+1e-8 I), 4 psi2 */
int mi355gp_sparse_fetch(mi355gp_sparse* s, int which, double* out);
/* Launch timing of the two MFMA kernels of the last mi355gp_vardtc_inference_sum call (hipEvent pairs on the launching
 * stream, recorded on every call): out6 = [T = Kfu dL_dpsi2 GEMM: summed ms, algorithmic flops (2 rows M^2), launches,
 * split-K Gram psi2: summed ms, algorithmic flops (rows M^2, lower half), launches]. */
int mi355gp_sparse_get_profile(mi355gp_sparse* s, double* out6);

#ifdef __cplusplus
}
#endif
#endif
