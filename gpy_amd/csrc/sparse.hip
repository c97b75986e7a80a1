// sparse.hip -- the variational sparse GP (VarDTC) path on device: BASELINE config 5 / SURVEY.md a18.
// Replaces, for certain inputs and a homoscedastic Gaussian likelihood,
//   VarDTC.inference                 GPy/inference/latent_function_inference/var_dtc.py:66-215 (+ helpers :217-276)
//   SparseGP._update_gradients       GPy/core/sparse_gp.py:108-118  (kernel and inducing-input gradients)
//   Stationary.gradients_X           GPy/kern/src/stationary.py:245-252,330-358 (C kernel stationary_utils.c)
// in the streaming (two-pass) form the reference itself uses for its MPI variant
// (VarDTC_minibatch.gatherPsiStat, var_dtc_parallel.py:72-133):
//   pass 1 over row chunks of X:  Kfu chunk -> psi2 += Kuf Kfu (split-K MFMA Gram), psi1Y += Kuf Y
//   M x M algebra (M <= a few thousand): Lm, Lm^-1, A, LB, LB^-1, B^-1, dL_dKmm, dL_dpsi2, woodbury_inv
//   pass 2 over the same chunks:   T = Kfu dL_dpsi2 (MFMA), dL_dKnm = beta Y v^T + 2 T formed in place,
//                                  theta reductions + H = dL_dKnm * dK/dr / r, then H^T [X~ | 1] for dL/dZ
// Larger N streams through bounded chunk buffers (<= 262144 rows: 2 x 4.3 GB at M = 2048); up to that size the Kfu
// chunk of pass 1 is still resident in pass 2 and is not rebuilt.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mi355gp.h"
#include "internal.h"

#define GP_STRIDE 34
#define SPLITK_MAX 16
#define CHUNK_MAX 262144                    // rows per chunk: 2 x (chunk x Mp) doubles of HBM (8.6 GB at M = 2048)
#define ARGCHK(cond, msg)                 \
    do {                                  \
        if (!(cond)) {                    \
            mi355gp_set_error("%s", msg); \
            return -1;                    \
        }                                 \
    } while (0)

// ---- elementwise M x M helpers (mp x mp row-major, ld = mp) --------------------------------------------------
// mirror the lower triangle onto the upper one; optionally sum `nsplit` partial matrices first
__global__ void k_sym_from_lower(const double* __restrict__ part, long mp, int nsplit, double* __restrict__ out) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= mp) return;
    const long a = (i >= j) ? i : j, b = (i >= j) ? j : i;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += part[(long)k * mp * mp + a * mp + b];
    out[i * mp + j] = s;
}
// out = ca * A + cb * B + ci * I   (A or B may be NULL)
__global__ void k_mm_axpby(const double* __restrict__ A, double ca, const double* __restrict__ B, double cb, double ci,
                           long mp, double* __restrict__ out) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= mp) return;
    double v = (i == j) ? ci : 0.0;
    if (A) v += ca * A[i * mp + j];
    if (B) v += cb * B[i * mp + j];
    out[i * mp + j] = v;
}
// P = Dy * sym(Wlow) + w w^T   (Wlow: lower tiles of B^-1 from lauum; w: mp x Dy)
__global__ void k_form_P(const double* __restrict__ Wlow, const double* __restrict__ w, int Dy, long mp, long m,
                         double* __restrict__ P) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= mp) return;
    const long a = (i >= j) ? i : j, b = (i >= j) ? j : i;
    double v = (double)Dy * Wlow[a * mp + b];
    if (i < m && j < m)
        for (int d = 0; d < Dy; ++d) v = fma(w[i * Dy + d], w[j * Dy + d], v);
    P[i * mp + j] = v;
}
// out[0] = trace(A) ; out[1] = sum(A * P) ; out[2] = sum_i log(LB_ii) ; out[3] = sum(c^2)   over the leading m x m / m x Dy
// stage 1: one block per row i: rowpart[i] = {A_ii, sum_j A_ij P_ij, log LB_ii, sum_d c_id^2}
__global__ __launch_bounds__(256) void k_sparse_scalars_rows(const double* __restrict__ A, const double* __restrict__ P,
                                                             const double* __restrict__ LB,
                                                             const double* __restrict__ c, int Dy, long mp, long m,
                                                             double* __restrict__ rowpart) {
    __shared__ double red[256];
    const int t = threadIdx.x;
    const long i = blockIdx.x;
    double s1 = 0.0;
    for (long j = t; j < m; j += 256) s1 = fma(A[i * mp + j], P[i * mp + j], s1);
    red[t] = s1;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (t < k) red[t] += red[t + k];
        __syncthreads();
    }
    if (t == 0) {
        double c2 = 0.0;
        for (int d = 0; d < Dy; ++d) c2 = fma(c[i * Dy + d], c[i * Dy + d], c2);
        rowpart[i * 4 + 0] = A[i * mp + i];
        rowpart[i * 4 + 1] = red[0];
        rowpart[i * 4 + 2] = log(LB[i * mp + i]);
        rowpart[i * 4 + 3] = c2;
    }
}
// stage 2 (fixed order): out[q] = sum_i rowpart[i][q]
__global__ __launch_bounds__(256) void k_sparse_scalars(const double* __restrict__ rowpart, long m,
                                                        double* __restrict__ out) {
    __shared__ double red[4][256];
    const int t = threadIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (long i = t; i < m; i += 256)
        for (int q = 0; q < 4; ++q) s[q] += rowpart[i * 4 + q];
    for (int q = 0; q < 4; ++q) red[q][t] = s[q];
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (t < k)
            for (int q = 0; q < 4; ++q) red[q][t] += red[q][t + k];
        __syncthreads();
    }
    if (t < 4) out[t] = red[t][0];
}
// G[i][j] = beta * sum_d Y[i][d] v[j][d] + 2 G[i][j] for i < rows, j < m; 0 in the padding
__global__ void k_form_dLdKnm(double* __restrict__ G, long ld, long rows, long rows_pad, long m,
                              const double* __restrict__ Y, const double* __restrict__ v, int Dy, double beta) {
    const long j = (long)blockIdx.y * blockDim.x + threadIdx.x, i = blockIdx.x;   // rows on x: no 65535 limit
    if (j >= ld || i >= rows_pad) return;
    double g = 0.0;
    if (i < rows && j < m) {
        double yv = 0.0;
        for (int d = 0; d < Dy; ++d) yv = fma(Y[i * Dy + d], v[j * Dy + d], yv);
        g = fma(2.0, G[i * ld + j], beta * yv);
    }
    G[i * ld + j] = g;
}
__global__ void k_vec_axpy(double* __restrict__ dst, const double* __restrict__ src, long cnt, double a) {
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l < cnt) dst[l] = fma(a, src[l], dst[l]);
}

static dim3 grid2d(long cols, long rows) { return dim3((unsigned)((cols + 255) / 256), (unsigned)rows); }

struct mi355gp_sparse {
    int fuse_cols = 1;            // MI355GP_SPARSE_FUSE_COLS: k_grad_cols (gradient pass + column reductions in one)
    int device = 0;
    hipStream_t st = nullptr;
    long n = 0, chunk = 0;
    int D = 0, Dy = 0, splitk = 8;
    double trYYT = 0.0;
    // row-sharded multi-GPU mode (the reference's MPI design: var_dtc_parallel.py:121-130,387-394):
    // this rank holds n of n_global rows; psi2 / psi1Y and the pass-2 sums are all-reduced, M x M algebra is replicated
    void* comm = nullptr;
    int world = 1, rank = 0;
    long n_global = 0;
    double *dX = nullptr, *dY = nullptr, *XtC = nullptr, *Kfu = nullptr, *T = nullptr;
    // M-dependent
    long m = 0, mp = 0;
    double *dZ = nullptr, *XtZ = nullptr, *invls = nullptr, *zero1 = nullptr;
    double *Lm = nullptr, *Xm = nullptr, *Tm = nullptr, *psi2part = nullptr, *psi2 = nullptr, *Amat = nullptr,
           *LB = nullptr, *XB = nullptr, *Bi = nullptr, *P = nullptr, *E = nullptr, *T1 = nullptr, *Q2 = nullptr,
           *dLdKmm = nullptr, *Winv = nullptr;
    double *psi1Y = nullptr, *vecA = nullptr, *vecB = nullptr, *cvec = nullptr, *wvec = nullptr, *vvec = nullptr,
           *trmvPart = nullptr, *colPart = nullptr, *HX = nullptr, *HZ = nullptr, *gradPart = nullptr,
           *gradChunk = nullptr, *gradNM = nullptr, *gradMM = nullptr, *scal = nullptr;
    FactorWs ws;
    bool ws_ok = false, have_result = false;
    hipEvent_t ev[6] = {};
    KernParams kp = {0, 0, 0, 1.0};
    std::vector<double> theta;
};

static void free_m(mi355gp_sparse* s) {
    double** ptrs[] = {&s->dZ, &s->XtZ, &s->invls, &s->zero1, &s->Lm, &s->Xm, &s->Tm, &s->psi2part, &s->psi2, &s->Amat,
                       &s->LB, &s->XB, &s->Bi, &s->P, &s->E, &s->T1, &s->Q2, &s->dLdKmm, &s->Winv, &s->psi1Y, &s->vecA,
                       &s->vecB, &s->cvec, &s->wvec, &s->vvec, &s->trmvPart, &s->colPart, &s->HX, &s->HZ, &s->gradPart,
                       &s->gradChunk, &s->gradNM, &s->gradMM, &s->scal, &s->Kfu, &s->T};
    for (auto p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (s->ws_ok) factor_ws_free(&s->ws);
    s->ws_ok = false;
    s->m = s->mp = 0;
    s->have_result = false;
}

static int alloc_m(mi355gp_sparse* s, long M) {
    free_m(s);
    if (s->XtC) (void)hipFree(s->XtC);
    s->XtC = nullptr;
    s->m = M;
    s->mp = round_up(M, NB);
    {
        // The Gram matrix has only ntl = (mp/128)(mp/128+1)/2 output tiles (136 at M = 2048) against 512 workgroup slots:
        // split K into S partial matrices, S chosen so that ntl*S fills whole rounds of the machine.
        const long nt = s->mp / NB, ntl = nt * (nt + 1) / 2;
        int best = 1;
        double beff = 0.0;
        for (int S = 1; S <= SPLITK_MAX; ++S) {
            const double wg = (double)ntl * S, eff = wg / (ceil(wg / 512.0) * 512.0);
            if (eff > beff + 1e-9 || (eff > beff - 0.02 && S < best)) { best = S; beff = eff; }
        }
        if (ntl >= 2048) best = 1;
        s->splitk = best;
        const long gran = 128L * best;                        // every split a multiple of 128 rows
        const long cmax = (CHUNK_MAX / gran) * gran;
        const long nchunks = (s->n + cmax - 1) / cmax;        // balanced chunks: no nearly-empty last chunk
        s->chunk = round_up((s->n + nchunks - 1) / nchunks, gran);
    }
    HIP_CHECK(hipMalloc(&s->XtC, sizeof(double) * s->D * s->chunk));
    const long mp = s->mp, D = s->D, Dy = s->Dy;
    const size_t mm = sizeof(double) * mp * mp;
    const int groups = (int)((D + 31) / 32);
    HIP_CHECK(hipMalloc(&s->dZ, sizeof(double) * M * D));
    HIP_CHECK(hipMalloc(&s->XtZ, sizeof(double) * D * mp));
    HIP_CHECK(hipMalloc(&s->invls, sizeof(double) * D));
    HIP_CHECK(hipMalloc(&s->zero1, sizeof(double) * 8));
    HIP_CHECK(hipMemset(s->zero1, 0, sizeof(double) * 8));
    double** mats[] = {&s->Lm, &s->Xm, &s->Tm, &s->psi2, &s->Amat, &s->LB, &s->XB, &s->Bi, &s->P, &s->E, &s->T1, &s->Q2,
                       &s->dLdKmm, &s->Winv};
    for (auto p : mats) HIP_CHECK(hipMalloc(p, mm));
    HIP_CHECK(hipMalloc(&s->psi2part, mm * SPLITK_MAX));
    HIP_CHECK(hipMalloc(&s->Kfu, sizeof(double) * s->chunk * mp));
    HIP_CHECK(hipMalloc(&s->T, sizeof(double) * s->chunk * mp));
    double** vecs[] = {&s->psi1Y, &s->vecA, &s->vecB, &s->cvec, &s->wvec, &s->vvec};
    for (auto p : vecs) HIP_CHECK(hipMalloc(p, sizeof(double) * mp * Dy));
    const long nchunks = (mp + 255) / 256;
    HIP_CHECK(hipMalloc(&s->trmvPart, sizeof(double) * nchunks * mp * Dy));
    const long nvmax = (D + 1 > Dy ? D + 1 : Dy);
    HIP_CHECK(hipMalloc(&s->colPart, sizeof(double) * 64 * mp * nvmax));
    HIP_CHECK(hipMalloc(&s->HX, sizeof(double) * mp * (D + 1)));
    HIP_CHECK(hipMalloc(&s->HZ, sizeof(double) * mp * (D + 1)));
    HIP_CHECK(hipMalloc(&s->gradPart, sizeof(double) * groups * 2048 * GP_STRIDE));
    HIP_CHECK(hipMalloc(&s->gradChunk, sizeof(double) * groups * GP_STRIDE));
    HIP_CHECK(hipMalloc(&s->gradNM, sizeof(double) * groups * GP_STRIDE));
    HIP_CHECK(hipMalloc(&s->gradMM, sizeof(double) * groups * GP_STRIDE));
    HIP_CHECK(hipMalloc(&s->scal, sizeof(double) * 8));
    if (factor_ws_alloc(&s->ws, mp) != 0) return -3;
    s->ws_ok = true;
    return 0;
}

extern "C" {

int mi355gp_sparse_create(int device, mi355gp_sparse** out) {
    int n = 0;
    mi355gp_device_count(&n);
    if (device < 0 || device >= n) {
        mi355gp_set_error("mi355gp_sparse_create: device %d not available (%d HIP devices visible)", device, n);
        return -2;
    }
    HIP_CHECK(hipSetDevice(device));
    mi355gp_sparse* s = new mi355gp_sparse();
    s->device = device;
    if (factor_engine(device, &s->st, nullptr, nullptr) != 0) return -2;    // the device's shared main stream
    {
        const char* e = getenv("MI355GP_SPARSE_FUSE_COLS");
        if (e && *e) s->fuse_cols = atoi(e) ? 1 : 0;
    }
    for (auto& e : s->ev) HIP_CHECK(hipEventCreate(&e));
    *out = s;
    return 0;
}

int mi355gp_sparse_destroy(mi355gp_sparse* s) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->st);
    free_m(s);
    if (s->dX) (void)hipFree(s->dX);
    if (s->dY) (void)hipFree(s->dY);
    if (s->XtC) (void)hipFree(s->XtC);
    if (s->comm) rccl_comm_destroy(s->comm);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->st) (void)hipStreamSynchronize(s->st);
    delete s;
    return 0;
}

int mi355gp_sparse_set_data(mi355gp_sparse* s, const double* X, int64_t N, int D, const double* Y, int Dy) {
    ARGCHK(s && X && Y && N > 0 && D > 0 && Dy > 0, "mi355gp_sparse_set_data: bad arguments");
    ARGCHK(D <= 32, "mi355gp_sparse_set_data: D <= 32 in this version (one LDS group of input dimensions)");
    HIP_CHECK(hipSetDevice(s->device));
    HIP_CHECK(hipStreamSynchronize(s->st));
    free_m(s);
    if (s->dX) (void)hipFree(s->dX);
    if (s->dY) (void)hipFree(s->dY);
    if (s->XtC) (void)hipFree(s->XtC);
    s->n = N;
    s->D = D;
    s->Dy = Dy;
    HIP_CHECK(hipMalloc(&s->dX, sizeof(double) * N * D));
    HIP_CHECK(hipMalloc(&s->dY, sizeof(double) * N * Dy));
    s->XtC = nullptr;                                         // sized with the chunk (alloc_m)
    HIP_CHECK(hipMemcpy(s->dX, X, sizeof(double) * N * D, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(s->dY, Y, sizeof(double) * N * Dy, hipMemcpyHostToDevice));
    double t = 0.0;
    for (int64_t i = 0; i < N * Dy; ++i) t += Y[i] * Y[i];     // get_trYYT (var_dtc.py:48-54)
    s->trYYT = t;
    s->n_global = N;
    if (s->comm) {                                             // global N and tr(Y Y^T) over the shards
        double h[2] = {(double)N, t}, *d = nullptr;
        HIP_CHECK(hipMalloc(&d, sizeof(h)));
        HIP_CHECK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
        if (int rc = rccl_allreduce_sum(s->comm, d, 2, s->st)) return rc;
        HIP_CHECK(hipStreamSynchronize(s->st));
        HIP_CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        (void)hipFree(d);
        s->n_global = (long)(h[0] + 0.5);
        s->trYYT = h[1];
    }
    return 0;
}

// Row-sharded mode: call once, before set_data, on every rank (id128 from mi355gp_grid_unique_id on rank 0).
// Each rank then passes ITS rows to mi355gp_sparse_set_data; Z and theta are replicated; results are identical on all ranks.
int mi355gp_sparse_attach_comm(mi355gp_sparse* s, int rank, int world, const void* id128) {
    ARGCHK(s && id128 && world >= 1 && rank >= 0 && rank < world, "mi355gp_sparse_attach_comm: bad arguments");
    HIP_CHECK(hipSetDevice(s->device));
    if (s->comm) rccl_comm_destroy(s->comm);
    s->comm = nullptr;
    if (int rc = rccl_comm_create(rank, world, id128, &s->comm)) return rc;
    s->rank = rank;
    s->world = world;
    return 0;
}

// out_scalars: [0] log marginal likelihood, [1] dL/d(noise variance) (= dL_dthetaL), [2] trace(A), [3] data_fit,
//              [4] sum(log diag LB), [5] beta
// dtheta_out: 1 + (ard ? D : 1) ; dZ_out: M x D ; wv_out (optional): woodbury_vector M x Dy
// stage_ms (optional): [0] pass 1 (Kfu, psi2, psi1Y), [1] M x M algebra, [2] pass 2 (gradients), [3] total
int mi355gp_vardtc_inference(mi355gp_sparse* s, int kind, int ard, const double* theta, const double* Z, int64_t M,
                             double noise_var, double extra_jitter, double* out_scalars, double* dtheta_out,
                             double* dZ_out, double* wv_out, double* stage_ms) {
    ARGCHK(s && s->n > 0, "mi355gp_vardtc_inference: set_data first");
    ARGCHK(theta && Z && M > 0 && out_scalars, "mi355gp_vardtc_inference: bad arguments");
    ARGCHK(kind >= 0 && kind <= 3 && theta[0] > 0.0, "mi355gp_vardtc_inference: bad kernel parameters");
    HIP_CHECK(hipSetDevice(s->device));
    const int D = s->D, Dy = s->Dy, nl = ard ? D : 1;
    std::vector<double> inv_ls((size_t)D, 0.0);
    for (int q = 0; q < nl; ++q) {
        ARGCHK(theta[1 + q] > 0.0, "lengthscales must be positive");
        inv_ls[q] = 1.0 / theta[1 + q];
    }
    if (M != s->m)
        if (int rc = alloc_m(s, M)) return rc;
    hipStream_t st = s->st;
    const long n = s->n, m = s->m, mp = s->mp, chunk = s->chunk;
    const int groups = (D + 31) / 32;
    const double beta = 1.0 / fmax(noise_var, 1e-8);                                    // var_dtc.py:78-80
    KernParams kp{kind, ard ? 1 : 0, D, theta[0]};
    s->kp = kp;
    s->theta.assign(theta, theta + 1 + nl);
    s->have_result = false;
    HIP_CHECK(hipMemcpyAsync(s->invls, inv_ls.data(), sizeof(double) * D, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(s->dZ, Z, sizeof(double) * m * D, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipEventRecord(s->ev[0], st));
    launch_scale_inputs(st, s->dZ, m, D, s->invls, kp.ard, s->XtZ, mp);
    // Kmm + 1e-8 I (var_dtc.py:93-94), Lm = chol (jitchol, :95), Xm = Lm^-1
    launch_kbuild_sym(st, kp, s->XtZ, mp, m, mp, s->Lm, s->zero1, 1, 1e-8 + extra_jitter, /*lower_only=*/1, 1);
    potrf_device(st, s->Lm, mp, &s->ws);
    int info_m = 0;
    HIP_CHECK(hipMemcpyAsync(&info_m, s->ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemsetAsync(s->Xm, 0, sizeof(double) * mp * mp, st));
    trtri_device(st, s->Lm, s->Xm, s->Tm, mp, &s->ws);
    // ---- pass 1: psi2 = Kuf Kfu, psi1Y = Kuf Y -----------------------------------------------------------
    HIP_CHECK(hipMemsetAsync(s->psi1Y, 0, sizeof(double) * mp * Dy, st));
    int nch = 0;
    for (long r0 = 0; r0 < n; r0 += chunk, ++nch) {
        const long rc = (n - r0 < chunk) ? (n - r0) : chunk;
        launch_scale_inputs(st, s->dX + r0 * D, rc, D, s->invls, kp.ard, s->XtC, chunk);
        // the cross-covariance kernel writes rows < rc, columns < m: zero only what it leaves out
        if (m < mp) {
            if (rc < chunk || nch == 0) HIP_CHECK(hipMemsetAsync(s->Kfu, 0, sizeof(double) * chunk * mp, st));
        } else if (rc < chunk) {
            HIP_CHECK(hipMemsetAsync(s->Kfu + rc * mp, 0, sizeof(double) * (chunk - rc) * mp, st));
        }
        launch_kbuild_cross(st, kp, s->XtC, chunk, rc, s->XtZ, mp, m, s->Kfu, mp);
        launch_gram_splitk(st, s->Kfu, mp, round_up(rc, 16L * s->splitk), mp, s->splitk, nch > 0, s->psi2part);   // rows >= rc are zero
        const int ns = launch_colreduce_multi(st, s->Kfu, mp, rc, mp, s->dY + r0 * Dy, Dy, 1, Dy, 0, s->colPart);
        launch_sum_splits(st, s->colPart, mp * Dy, ns, 1, s->psi1Y);               // psi1Y += Kuf Y_chunk
    }
    hipLaunchKernelGGL(k_sym_from_lower, grid2d(mp, mp), dim3(256), 0, st, s->psi2part, mp, s->splitk, s->psi2);
    if (s->comm) {                                              // the one exchange step of pass 1
        if (int rc = rccl_allreduce_sum(s->comm, s->psi2, (size_t)mp * mp, st)) return rc;
        if (int rc = rccl_allreduce_sum(s->comm, s->psi1Y, (size_t)mp * Dy, st)) return rc;
    }
    HIP_CHECK(hipEventRecord(s->ev[1], st));
    // ---- M x M algebra ----------------------------------------------------------------------------------------
    // A = beta * Lm^-1 psi2 Lm^-T (var_dtc.py:129-134), B = I + A (:137), LB = chol(B) (:138), XB = LB^-1
    launch_gemm(st, 0, 1, mp, mp, mp, s->Xm, mp, s->psi2, mp, s->T1, mp, 1.0, 0.0);
    launch_gemm(st, 0, 0, mp, mp, mp, s->T1, mp, s->Xm, mp, s->Amat, mp, beta, 0.0);
    hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->Amat, 1.0, (const double*)nullptr, 0.0, 1.0, mp,
                       s->LB);
    potrf_device(st, s->LB, mp, &s->ws);
    int info_b = 0;
    HIP_CHECK(hipMemcpyAsync(&info_b, s->ws.info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemsetAsync(s->XB, 0, sizeof(double) * mp * mp, st));
    trtri_device(st, s->LB, s->XB, s->Tm, mp, &s->ws);
    // c = LB^-1 Lm^-1 psi1 (beta Y) (:141-143), w = LB^-T c (:144), v = Lm^-T w = woodbury_vector (:145)
    hipLaunchKernelGGL(k_vec_axpy, dim3((unsigned)((mp * Dy + 255) / 256)), dim3(256), 0, st, s->psi1Y, s->psi1Y, mp * Dy,
                       beta - 1.0);                                                    // psi1Y *= beta
    launch_trmv_lower(st, s->Xm, mp, mp, s->psi1Y, Dy, s->vecA);
    launch_trmv_lower(st, s->XB, mp, mp, s->vecA, Dy, s->cvec);
    launch_trmv_lower_T(st, s->XB, mp, mp, s->cvec, Dy, s->wvec, s->trmvPart);
    launch_trmv_lower_T(st, s->Xm, mp, mp, s->wvec, Dy, s->vvec, s->trmvPart);
    // B^-1 = XB^T XB (lower tiles), P = Dy B^-1 + w w^T = DBi_plus_BiPBi (:150-152)
    lauum_device(st, s->XB, s->Bi, mp, &s->ws);
    hipLaunchKernelGGL(k_form_P, grid2d(mp, mp), dim3(256), 0, st, s->Bi, s->wvec, Dy, mp, m, s->P);
    // dL_dKmm = Lm^-T (-0.5 P - 0.5 Dy B + Dy I) Lm^-1 (:153-158);  -0.5 Dy (I + A) + Dy I = -0.5 Dy A + 0.5 Dy I
    hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->P, -0.5, s->Amat, -0.5 * Dy, 0.5 * Dy, mp, s->E);
    launch_gemm(st, 1, 1, mp, mp, mp, s->Xm, mp, s->E, mp, s->T1, mp, 1.0, 0.0);        // Xm^T E
    launch_gemm(st, 0, 1, mp, mp, mp, s->T1, mp, s->Xm, mp, s->dLdKmm, mp, 1.0, 0.0);   // (Xm^T E) Xm
    // dL_dpsi2 = beta * 0.5 * Lm^-T (Dy I - P) Lm^-1 (:220,231)
    hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->P, -0.5 * beta, (const double*)nullptr, 0.0,
                       0.5 * beta * Dy, mp, s->E);
    launch_gemm(st, 1, 1, mp, mp, mp, s->Xm, mp, s->E, mp, s->T1, mp, 1.0, 0.0);
    launch_gemm(st, 0, 1, mp, mp, mp, s->T1, mp, s->Xm, mp, s->Q2, mp, 1.0, 0.0);
    hipLaunchKernelGGL(k_sparse_scalars_rows, dim3((unsigned)m), dim3(256), 0, st, s->Amat, s->P, s->LB, s->cvec, Dy, mp, m,
                       s->colPart);
    hipLaunchKernelGGL(k_sparse_scalars, dim3(1), dim3(256), 0, st, s->colPart, m, s->scal);
    HIP_CHECK(hipEventRecord(s->ev[2], st));
    // ---- pass 2: dL_dKnm = beta Y v^T + 2 Kfu dL_dpsi2 (:219,233), its theta reductions and H^T [X~ | 1] ---------
    HIP_CHECK(hipMemsetAsync(s->gradNM, 0, sizeof(double) * groups * GP_STRIDE, st));
    HIP_CHECK(hipMemsetAsync(s->HX, 0, sizeof(double) * mp * (D + 1), st));
    nch = 0;
    const int one_chunk = (n <= chunk);
    for (long r0 = 0; r0 < n; r0 += chunk, ++nch) {
        const long rc = (n - r0 < chunk) ? (n - r0) : chunk;
        const long rcp = round_up(rc, NB);
        if (!one_chunk) {                                    // a single chunk is still resident from pass 1
            launch_scale_inputs(st, s->dX + r0 * D, rc, D, s->invls, kp.ard, s->XtC, chunk);
            if (rc < chunk) HIP_CHECK(hipMemsetAsync(s->Kfu + rc * mp, 0, sizeof(double) * (chunk - rc) * mp, st));
            launch_kbuild_cross(st, kp, s->XtC, chunk, rc, s->XtZ, mp, m, s->Kfu, mp);
        }
        launch_gemm(st, 0, 1, rcp, mp, mp, s->Kfu, mp, s->Q2, mp, s->T, mp, 1.0, 0.0);
        // dL_dKnm = 2 T + beta Y v^T is formed inside the gradient pass (no separate read-modify-write of the chunk).
        // D <= 16: the same pass also accumulates H^T [X~ | 1] (H = dL_dKnm * (dK/dr)/r stays on chip); otherwise H
        // overwrites T in place and a second pass reduces it.  Padding rows / columns of T are zero from the GEMM.
        const RankTerm rk{s->dY + r0 * Dy, s->vvec, Dy, beta, 2.0};
        int nbk = 0;
        int ns = s->fuse_cols ? launch_grad_cols(st, kp, s->XtC, chunk, rc, s->XtZ, mp, m, mp, s->T, mp, rk, s->gradPart,
                                                 s->colPart, &nbk) : 0;
        if (ns == 0) {
            nbk = grad_generic_num_blocks(rc, m);
            launch_grad_generic(st, kp, s->XtC, chunk, rc, s->XtZ, mp, m, 0, s->T, mp, s->gradPart, GP_STRIDE, s->T, mp, rk);
            ns = launch_colreduce_multi(st, s->T, mp, rc, mp, s->XtC, 1, chunk, D, 1, s->colPart);
        }
        for (int g = 0; g < (kp.ard ? groups : 1); ++g)
            launch_reduce_partials(st, s->gradPart + (long)g * nbk * GP_STRIDE, nbk, GP_STRIDE,
                                   s->gradChunk + (long)g * GP_STRIDE);
        hipLaunchKernelGGL(k_vec_axpy, dim3(1), dim3(256), 0, st, s->gradNM, s->gradChunk, (long)groups * GP_STRIDE, 1.0);
        launch_sum_splits(st, s->colPart, mp * (D + 1), ns, 1, s->HX);
    }
    if (s->comm) {                                              // the one exchange step of pass 2
        if (int rc = rccl_allreduce_sum(s->comm, s->gradNM, (size_t)groups * GP_STRIDE, st)) return rc;
        if (int rc = rccl_allreduce_sum(s->comm, s->HX, (size_t)mp * (D + 1), st)) return rc;
    }
    // the M x M part: update_gradients_full(dL_dKmm, Z) and gradients_X(dL_dKmm, Z) (sparse_gp.py:114-117)
    {
        const int nbk = grad_generic_num_blocks(m, m);
        launch_grad_generic(st, kp, s->XtZ, mp, m, s->XtZ, mp, m, 1, s->dLdKmm, mp, s->gradPart, GP_STRIDE, s->T1, mp);
        for (int g = 0; g < (kp.ard ? groups : 1); ++g)
            launch_reduce_partials(st, s->gradPart + (long)g * nbk * GP_STRIDE, nbk, GP_STRIDE,
                                   s->gradMM + (long)g * GP_STRIDE);
        const int ns = launch_colreduce_multi(st, s->T1, mp, m, mp, s->XtZ, 1, mp, D, 1, s->colPart);
        launch_sum_splits(st, s->colPart, mp * (D + 1), ns, 0, s->HZ);
    }
    HIP_CHECK(hipEventRecord(s->ev[3], st));
    // ---- small results to the host -------------------------------------------------------------------------------------
    std::vector<double> gnm((size_t)groups * GP_STRIDE), gmm((size_t)groups * GP_STRIDE), HX((size_t)mp * (D + 1)),
        HZ((size_t)mp * (D + 1)), Zs((size_t)D * mp);
    double scal[8];
    HIP_CHECK(hipMemcpyAsync(gnm.data(), s->gradNM, sizeof(double) * gnm.size(), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(gmm.data(), s->gradMM, sizeof(double) * gmm.size(), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(HX.data(), s->HX, sizeof(double) * HX.size(), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(HZ.data(), s->HZ, sizeof(double) * HZ.size(), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(Zs.data(), s->XtZ, sizeof(double) * Zs.size(), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(scal, s->scal, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
    if (wv_out) HIP_CHECK(hipMemcpyAsync(wv_out, s->vvec, sizeof(double) * m * Dy, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipGetLastError());
    if (stage_ms) {
        float ms;
        for (int i = 0; i < 3; ++i) {
            HIP_CHECK(hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]));
            stage_ms[i] = ms;
        }
        HIP_CHECK(hipEventElapsedTime(&ms, s->ev[0], s->ev[3]));
        stage_ms[3] = ms;
    }
    if (info_m < 0 || info_b < 0) {
        mi355gp_set_error("panel factorisation: workgroup hand-off timed out (k_panel_fused)");
        return -7;
    }
    if (info_m > 0) return info_m > m ? (int)m : info_m;                 // Kmm not positive definite: caller adds jitter
    if (info_b > 0) return info_b > m ? (int)m : info_b;
    const double trA = scal[0], sumAP = scal[1], logLB = scal[2], data_fit = scal[3];
    const double ng = (double)s->n_global;                       // all shards
    const double variance = theta[0], psi0sum = ng * variance, nd = ng * Dy;
    // _compute_log_marginal_likelihood (var_dtc.py:264-276)
    const double lik_1 = -0.5 * nd * (log(2.0 * M_PI) - log(beta)) - 0.5 * beta * s->trYYT;
    const double lik_2 = -0.5 * Dy * (beta * psi0sum - trA);
    const double lik_3 = -(double)Dy * logLB;
    // _compute_dL_dR (var_dtc.py:258-261)
    double dL_dR = -0.5 * nd * beta + 0.5 * s->trYYT * beta * beta;
    dL_dR += 0.5 * Dy * (psi0sum * beta * beta - trA * beta);
    dL_dR += beta * (0.5 * sumAP - data_fit);
    for (int i = 0; i < MI355GP_NUM_OUT; ++i) out_scalars[i] = 0.0;
    out_scalars[0] = lik_1 + lik_2 + lik_3 + 0.5 * data_fit;
    out_scalars[1] = dL_dR;
    out_scalars[2] = trA;
    out_scalars[3] = data_fit;
    out_scalars[4] = logLB;
    out_scalars[5] = beta;
    if (dtheta_out) {
        // update_gradients_diag(dL_dKdiag = -0.5 Dy beta) (sparse_gp.py:110, stationary.py:175-184): variance only
        dtheta_out[0] = -0.5 * Dy * beta * ng + (gnm[0] + gmm[0]) / variance;
        if (!kp.ard) dtheta_out[1] = -(gnm[1] + gmm[1]) / theta[1];
        else
            for (int q = 0; q < D; ++q) {
                const int o = (q / 32) * GP_STRIDE + 2 + (q % 32);
                dtheta_out[1 + q] = -(gnm[o] + gmm[o]) / theta[1 + q];
            }
    }
    if (dZ_out) {
        // gradients_X(dL_dKnm^T, Z, X) + gradients_X(dL_dKmm, Z) (sparse_gp.py:116-118):
        //   sum_n H[n,m] (z~_mq - x~_nq) / l_q  +  2 sum_j Hmm[j,m] (z~_mq - z~_jq) / l_q
        for (long j = 0; j < m; ++j)
            for (int q = 0; q < D; ++q) {
                const double zs = Zs[(size_t)q * mp + j], il = inv_ls[ard ? q : 0];
                const double a = zs * HX[j * (D + 1) + D] - HX[j * (D + 1) + q];
                const double b = zs * HZ[j * (D + 1) + D] - HZ[j * (D + 1) + q];
                dZ_out[j * D + q] = (a + 2.0 * b) * il;
            }
    }
    s->have_result = true;
    return 0;
}

// M x M results of the last call: 0 = dL_dKmm, 1 = woodbury_inv = Lm^-T (I - B^-1) Lm^-1 (var_dtc.py:206-210),
// 2 = Lm (lower, strict upper zero), 3 = Kmm (with the 1e-8 jitter), 4 = psi2
int mi355gp_sparse_fetch(mi355gp_sparse* s, int which, double* out) {
    ARGCHK(s && out && s->have_result, "mi355gp_sparse_fetch: run mi355gp_vardtc_inference first");
    HIP_CHECK(hipSetDevice(s->device));
    hipStream_t st = s->st;
    const long m = s->m, mp = s->mp;
    const double* src = nullptr;
    if (which == 0) src = s->dLdKmm;
    else if (which == 1) {
        hipLaunchKernelGGL(k_sym_from_lower, grid2d(mp, mp), dim3(256), 0, st, s->Bi, mp, 1, s->E);
        hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->E, -1.0, (const double*)nullptr, 0.0, 1.0, mp,
                           s->E);
        launch_gemm(st, 1, 1, mp, mp, mp, s->Xm, mp, s->E, mp, s->T1, mp, 1.0, 0.0);
        launch_gemm(st, 0, 1, mp, mp, mp, s->T1, mp, s->Xm, mp, s->Winv, mp, 1.0, 0.0);
        src = s->Winv;
    } else if (which == 2) {
        launch_extract(st, s->Lm, mp, mp, 0, nullptr, 0, s->E, 0);
        src = s->E;
    } else if (which == 3) {
        launch_kbuild_sym(st, s->kp, s->XtZ, mp, m, mp, s->E, s->zero1, 1, 1e-8, 0, 1);
        src = s->E;
    } else if (which == 4) src = s->psi2;
    else {
        mi355gp_set_error("mi355gp_sparse_fetch: unknown matrix id %d", which);
        return -1;
    }
    HIP_CHECK(hipMemcpy2DAsync(out, sizeof(double) * m, src, sizeof(double) * mp, sizeof(double) * m, m,
                               hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
