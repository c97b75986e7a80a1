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
#include <functional>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mi355gp.h"
#include "../../include/mi355gp_debug.h"
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
// G[i][j] = beta_i * (sum_d R[i][d] v[j][d] + 2 G[i][j]) for i < rows, j < m; 0 in the padding   (dL_dKnm, var_dtc.py:219-233)
__global__ void k_form_dLdKnm(double* __restrict__ G, long ld, long rows, long rows_pad, long m,
                              const double* __restrict__ Y, const double* __restrict__ v, int Dy,
                              const double* __restrict__ beta) {
    const long j = (long)blockIdx.y * blockDim.x + threadIdx.x, i = blockIdx.x;   // rows on x: no 65535 limit
    if (j >= ld || i >= rows_pad) return;
    double g = 0.0;
    if (i < rows && j < m) {
        double yv = 0.0;
        for (int d = 0; d < Dy; ++d) yv = fma(Y[i * Dy + d], v[j * Dy + d], yv);
        g = beta[i] * fma(2.0, G[i * ld + j], yv);
    }
    G[i * ld + j] = g;
}
// out[j] = c0 - sum_i A[i][j] * B[i][j]   (64 columns per block, 4 row groups, fixed-order combine)
__global__ __launch_bounds__(256) void k_col_dot(const double* __restrict__ A, const double* __restrict__ B, long ld,
                                                 long rows, long cols, double c0, double* __restrict__ out) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + tx;
    double acc = 0.0;
    if (j < cols)
        for (long i = g; i < rows; i += 4) acc = fma(A[i * ld + j], B[i * ld + j], acc);
    red[g][tx] = acc;
    __syncthreads();
    if (g == 0 && j < cols) out[j] = c0 - ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]));
}
__global__ void k_vec_axpy(double* __restrict__ dst, const double* __restrict__ src, long cnt, double a) {
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l < cnt) dst[l] = fma(a, src[l], dst[l]);
}

// out[i][d] = w[i] * in[i][d]
__global__ void k_scale_rows(const double* __restrict__ in, const double* __restrict__ w, long cnt, int Dy,
                             double* __restrict__ out) {
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l < cnt) out[l] = w[l / Dy] * in[l];
}

__global__ void k_fill_const(double* __restrict__ out, long cnt, double v) {
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l < cnt) out[l] = v;
}

static dim3 grid2d(long cols, long rows) { return dim3((unsigned)((cols + 255) / 256), (unsigned)rows); }

// ---- communicators of the row-sharded mode ------------------------------------------------------------------------
// RCCL (one rank per process / GPU) or LOOPBACK: `world` contexts of ONE process on one device, driven by one host thread
// each, meet at a process-local rendezvous and are summed in rank order -- the transport that lets the world > 1 logic
// (global N, tr(YY^T), the two exchange steps, replicated M x M algebra) be parity-tested on a 1-GPU box.
#include <condition_variable>
#include <map>
#include <mutex>
struct LoopGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, arrived = 0;
    long gen = 0;
    std::vector<double*> bufs;
};
static std::map<int, LoopGroup*> g_loop_groups;
static std::mutex g_loop_mu;

__global__ void k_vec_add(double* __restrict__ dst, const double* __restrict__ src, long cnt) {
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l < cnt) dst[l] += src[l];
}

static int loop_allreduce(LoopGroup* G, int rank, double* buf, size_t count, hipStream_t st) {
    HIP_CHECK(hipStreamSynchronize(st));                      // this rank's partial sums are complete
    std::unique_lock<std::mutex> lk(G->mu);
    G->bufs[(size_t)rank] = buf;
    const long gen = G->gen;
    if (++G->arrived == G->world) {                           // the last arrival reduces, in rank order, and redistributes
        const unsigned nblk = (unsigned)((count + 255) / 256);
        for (int r = 1; r < G->world; ++r)
            hipLaunchKernelGGL(k_vec_add, dim3(nblk), dim3(256), 0, st, G->bufs[0], G->bufs[(size_t)r], (long)count);
        for (int r = 1; r < G->world; ++r)
            HIP_CHECK(hipMemcpyAsync(G->bufs[(size_t)r], G->bufs[0], count * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        G->arrived = 0;
        ++G->gen;
        G->cv.notify_all();
    } else {
        G->cv.wait(lk, [&] { return G->gen != gen; });
    }
    return 0;
}

struct SPart {
    KernParams kp = {0, 0, 0, 1.0};
    std::vector<double> theta, inv_ls;
    std::vector<int> dims;
    double *XtZ = nullptr, *XtC = nullptr, *HX = nullptr, *HZ = nullptr, *gradNM = nullptr, *gradMM = nullptr;
    int term = 0;                 // parts with the same non-zero id are the factors of one product (GPy/kern/src/prod.py)
    int tix = 0;                  // index into mi355gp_sparse::terms
    bool stationary() const { return kp.kind <= 3; }
};

struct mi355gp_sparse {
    // MI355GP_SPARSE_KMM_OVERLAP: Kmm's build + Cholesky + inverse (they need only Z) on a side stream UNDERNEATH pass 1 when the
    // factorisation is the single persistent launch (~1.2 ms of latency-bound work that otherwise runs on an idle GPU)
    int kmm_overlap = 1;
    hipStream_t st_kmm = nullptr;
    hipEvent_t ev_z = nullptr, ev_kmm = nullptr;
    int fuse_cols = 1;            // MI355GP_SPARSE_FUSE_COLS: k_grad_cols (gradient pass + column reductions in one)
    int device = 0;
    hipStream_t st = nullptr;
    long n = 0, chunk = 0;
    int D = 0, Dy = 0, splitk = 8;
    double trYYT = 0.0, trYYT_local = 0.0;
    std::vector<double> rowYY;    // host: |R_n|^2 per row (heteroscedastic log likelihood)
    // row-sharded multi-GPU mode (the reference's MPI design: var_dtc_parallel.py:121-130,387-394):
    // this rank holds n of n_global rows; psi2 / psi1Y and the pass-2 sums are all-reduced, M x M algebra is replicated
    void* comm = nullptr;         // RCCL communicator
    LoopGroup* loop = nullptr;    // or the loopback rendezvous
    int world = 1, rank = 0;
    long n_global = 0;
    double *dX = nullptr, *dY = nullptr, *dV = nullptr, *dBeta = nullptr, *dRowS = nullptr, *dRowT = nullptr, *dRowR = nullptr,
           *Kfu = nullptr,
           *T = nullptr;
    // M-dependent
    long m = 0, mp = 0;
    double *dZ = nullptr, *invls = nullptr, *zero1 = nullptr;
    double *Lm = nullptr, *Xm = nullptr, *Tm = nullptr, *psi2part = nullptr, *psi2 = nullptr, *Amat = nullptr,
           *LB = nullptr, *XB = nullptr, *Bi = nullptr, *P = nullptr, *E = nullptr, *T1 = nullptr, *Q2 = nullptr,
           *dLdKmm = nullptr, *Winv = nullptr;
    double *psi1Y = nullptr, *vecA = nullptr, *vecB = nullptr, *cvec = nullptr, *wvec = nullptr, *vvec = nullptr,
           *trmvPart = nullptr, *colPart = nullptr, *gradPart = nullptr, *gradChunk = nullptr, *scal = nullptr,
           *redbuf = nullptr;
    std::vector<SPart> parts;
    std::vector<std::vector<int>> terms;   // part indices per summand, in order of first appearance (one part, or the factors of a Prod)
    FactorWs ws;
    bool ws_ok = false, have_result = false, winv_ok = false;
    hipEvent_t ev[6] = {};
    int h_info[2] = {0, 0};       // LAPACK-style info of the two M x M factorisations (targets of async copies: not on the stack)
    double beta_scalar = 0.0;     // homoscedastic precision of the last call (0: per-point)
    KernelProf mfma_prof;         // launch timing of the two MFMA kernels of a call: family 0 = T = Kfu dL_dpsi2, 1 = split-K Gram
};

static int sparse_allreduce(mi355gp_sparse* s, double* buf, size_t count) {
    if (s->comm) return rccl_allreduce_sum(s->comm, buf, count, s->st);
    if (s->loop) return loop_allreduce(s->loop, s->rank, buf, count, s->st);
    return 0;
}
static bool sharded(const mi355gp_sparse* s) { return s->comm != nullptr || s->loop != nullptr; }

static void free_parts(mi355gp_sparse* s) {
    for (SPart& p : s->parts) {
        double** ptrs[] = {&p.XtZ, &p.XtC, &p.HX, &p.HZ, &p.gradNM, &p.gradMM};
        for (auto q : ptrs) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
    }
    s->parts.clear();
}

static void free_m(mi355gp_sparse* s) {
    free_parts(s);
    double** ptrs[] = {&s->dZ, &s->invls, &s->zero1, &s->Lm, &s->Xm, &s->Tm, &s->psi2part, &s->psi2, &s->Amat,
                       &s->LB, &s->XB, &s->Bi, &s->P, &s->E, &s->T1, &s->Q2, &s->dLdKmm, &s->Winv, &s->psi1Y, &s->vecA,
                       &s->vecB, &s->cvec, &s->wvec, &s->vvec, &s->trmvPart, &s->colPart, &s->gradPart,
                       &s->gradChunk, &s->scal, &s->redbuf, &s->Kfu, &s->T};
    for (auto p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (s->ws_ok) factor_ws_free(&s->ws);
    s->ws_ok = false;
    s->m = s->mp = 0;
    s->have_result = s->winv_ok = false;
}

static int alloc_m(mi355gp_sparse* s, long M) {
    free_m(s);
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
    const long mp = s->mp, D = s->D, Dy = s->Dy;
    const size_t mm = sizeof(double) * mp * mp;
    const int groups = (int)((D + 31) / 32);
    HIP_CHECK(hipMalloc(&s->dZ, sizeof(double) * M * D));
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
    const long nchunks = (mp + trmv_chunk_rows(mp) - 1) / trmv_chunk_rows(mp);
    HIP_CHECK(hipMalloc(&s->trmvPart, sizeof(double) * nchunks * mp * Dy));
    const long nvmax = (D + 1 > Dy ? D + 1 : Dy);
    HIP_CHECK(hipMalloc(&s->colPart, sizeof(double) * 64 * mp * nvmax));
    HIP_CHECK(hipMalloc(&s->gradPart, sizeof(double) * groups * 2048 * GP_STRIDE));
    HIP_CHECK(hipMalloc(&s->gradChunk, sizeof(double) * groups * GP_STRIDE));
    HIP_CHECK(hipMalloc(&s->scal, sizeof(double) * 8));
    if (factor_ws_alloc(&s->ws, mp) != 0) return -3;
    s->ws_ok = true;
    return 0;
}

// (re)builds the part list of a call; the device buffers of a part are kept while the number of parts is unchanged
static int prepare_sparse_parts(mi355gp_sparse* s, int nparts, const mi355gp_part* parts) {
    ARGCHK(nparts >= 1 && nparts <= 16 && parts, "between 1 and 16 kernel parts");
    const long mp = s->mp, D = s->D;
    const int groups = (int)((D + 31) / 32);
    if ((int)s->parts.size() != nparts) {
        free_parts(s);
        s->parts.resize((size_t)nparts);
        if (s->redbuf) (void)hipFree(s->redbuf);
        s->redbuf = nullptr;
        HIP_CHECK(hipMalloc(&s->redbuf, sizeof(double) * nparts * ((size_t)groups * GP_STRIDE + (size_t)mp * (D + 1))));
        for (SPart& p : s->parts) {
            HIP_CHECK(hipMalloc(&p.XtZ, sizeof(double) * D * mp));
            HIP_CHECK(hipMalloc(&p.XtC, sizeof(double) * D * s->chunk));
            HIP_CHECK(hipMalloc(&p.HX, sizeof(double) * mp * (D + 1)));
            HIP_CHECK(hipMalloc(&p.HZ, sizeof(double) * mp * (D + 1)));
            HIP_CHECK(hipMalloc(&p.gradNM, sizeof(double) * groups * GP_STRIDE));
            HIP_CHECK(hipMalloc(&p.gradMM, sizeof(double) * groups * GP_STRIDE));
        }
    }
    for (int i = 0; i < nparts; ++i) {
        const mi355gp_part& in = parts[i];
        SPart& p = s->parts[(size_t)i];
        ARGCHK(in.kind >= 0 && in.kind <= 5 && in.theta, "unknown covariance kind / NULL theta in a kernel part");
        p.term = in.term;
        ARGCHK(in.theta[0] > 0.0, "variance must be positive");
        p.dims.clear();
        if (in.active_dims && in.n_active > 0) {
            for (int a = 0; a < in.n_active; ++a) {
                ARGCHK(in.active_dims[a] >= 0 && in.active_dims[a] < D, "active dimension out of range");
                p.dims.push_back(in.active_dims[a]);
            }
        } else {
            for (int q = 0; q < D; ++q) p.dims.push_back(q);
        }
        const int na = (int)p.dims.size();
        const bool st = in.kind <= 3;
        const int nl = st ? (in.ard ? na : 1) : 0;
        p.kp = KernParams{in.kind, (st && in.ard) ? 1 : 0, (int)D, in.theta[0]};
        p.theta.assign(in.theta, in.theta + 1 + nl);
        p.inv_ls.assign((size_t)D, 0.0);
        for (int a = 0; a < na && st; ++a) {
            const double l = in.theta[1 + (in.ard ? a : 0)];
            ARGCHK(l > 0.0, "lengthscales must be positive");
            p.inv_ls[(size_t)p.dims[a]] = 1.0 / l;
        }
    }
    // summands: term id 0 = a part of its own; parts sharing a non-zero id are multiplied (prod.py:58-72)
    s->terms.clear();
    std::vector<int> ids;
    for (int i = 0; i < nparts; ++i) {
        const int id = s->parts[(size_t)i].term;
        size_t t = ids.size();
        if (id != 0)
            for (t = 0; t < ids.size() && ids[t] != id; ++t) {}
        if (t == ids.size()) {
            ids.push_back(id != 0 ? id : -1 - i);
            s->terms.emplace_back();
        }
        s->terms[t].push_back(i);
        s->parts[(size_t)i].tix = (int)t;
    }
    for (const auto& t : s->terms)
        if (t.size() > 1)
            for (int f : t) ARGCHK(s->parts[(size_t)f].kp.kind != 4, "a White factor inside a product is not supported by the sparse path");
    return 0;
}

// Kdiag of the expression (psi0_n): sum over summands of the product of the factors' variances (add.py:74-79, prod.py:67-71)
static double sparse_kdiag(const mi355gp_sparse* s) {
    double k = 0.0;
    for (const auto& t : s->terms) {
        double v = 1.0;
        for (int f : t) v *= s->parts[(size_t)f].kp.variance;
        k += v;
    }
    return k;
}
// product of the OTHER factors' variances of part p's summand (= dKdiag/dvariance_p; 1 for a plain summand)
static double sparse_other_variances(const mi355gp_sparse* s, size_t p) {
    double v = 1.0;
    for (int f : s->terms[(size_t)s->parts[p].tix])
        if ((size_t)f != p) v *= s->parts[(size_t)f].kp.variance;
    return v;
}
// out (+)= sum over summands of the element-wise product of their factors: emit(part, dst, mul, accumulate) launches one
// factor, dst (+)= K_part * mul.  The leading factors of a product are multiplied up in `scratch` (same shape as out).
// skip_white: cross-covariances (White contributes nothing, static.py:77-81).  Returns false if nothing was emitted.
template <class Emit>
static bool sparse_expression(const mi355gp_sparse* s, double* out, double* scratch, bool skip_white, Emit emit) {
    bool first = true;
    for (const auto& t : s->terms) {
        if (skip_white && t.size() == 1 && s->parts[(size_t)t[0]].kp.kind == 4) continue;
        for (size_t f = 0; f + 1 < t.size(); ++f) emit(t[f], scratch, f > 0 ? scratch : nullptr, 0, false);
        emit(t.back(), out, t.size() > 1 ? scratch : nullptr, first ? 0 : 1, first);
        first = false;
    }
    return !first;
}
// dst = the product of the OTHER factors of part p's summand, evaluated by emit(part, dst, mul, accumulate); false: p stands alone
template <class Emit>
static bool sparse_other_factors(const mi355gp_sparse* s, size_t p, double* dst, Emit emit) {
    const auto& t = s->terms[(size_t)s->parts[p].tix];
    if (t.size() < 2) return false;
    bool first = true;
    for (int f : t) {
        if ((size_t)f == p) continue;
        emit(f, dst, first ? nullptr : dst, 0, first);
        first = false;
    }
    return true;
}

// scaled, dimension-major copies of `rows` points (row-major src) for every part
static int scale_for_parts(mi355gp_sparse* s, const double* src, long rows, long ldt, bool inducing) {
    for (SPart& p : s->parts) {
        HIP_CHECK(hipMemcpyAsync(s->invls, p.inv_ls.data(), sizeof(double) * s->D, hipMemcpyHostToDevice, s->st));
        launch_scale_inputs(s->st, src, rows, s->D, s->invls, 1, inducing ? p.XtZ : p.XtC, ldt);
    }
    return 0;
}

// Kfu chunk = sum over summands of (the product of) K_p(X_chunk, Z)  (add.py:58-72, prod.py:58-65; White contributes nothing
// off the diagonal, static.py:77-81).  Products are multiplied up in `scratch` (chunk x mp, e.g. the T buffer).
static void build_cross_chunk(mi355gp_sparse* s, long rc, double* out, double* scratch) {
    const bool any = sparse_expression(s, out, scratch, true, [&](int p, double* dst, const double* mul, int acc, bool) {
        const SPart& pt = s->parts[(size_t)p];
        launch_kbuild_cross(s->st, pt.kp, pt.XtC, s->chunk, rc, pt.XtZ, s->mp, s->m, dst, s->mp, acc, 0, mul);
    });
    if (!any) (void)hipMemsetAsync(out, 0, sizeof(double) * rc * s->mp, s->st);       // only White parts: K(X, Z) = 0
}
// K(Z) (lower tiles; diag != NULL: + diag on the diagonal) of the expression into out (mp x mp), scratch mp x mp
static void build_kmm(mi355gp_sparse* s, double* out, double* scratch, double jitter, int lower_only, hipStream_t st = nullptr) {
    if (!st) st = s->st;
    sparse_expression(s, out, scratch, false, [&](int p, double* dst, const double* mul, int acc, bool first) {
        const SPart& pt = s->parts[(size_t)p];
        launch_kbuild_sym(st, pt.kp, pt.XtZ, s->mp, s->m, s->mp, dst, s->zero1, 1, jitter, lower_only,
                          /*add_diag=*/(first && dst == out) ? 1 : 0, acc, mul);
    });
}
// W[i][j] = beta_i (sum_d R[i][d] v[j][d] + 2 T[i][j]) * W[i][j] for i < rows, j < m; 0 in the padding: dL_dKnm times the other
// factors' covariance (prod.py:86-99), the weights one factor of a product sees
__global__ void k_form_dLdKnm_times(const double* __restrict__ T, double* __restrict__ W, long ld, long rows, long rows_pad,
                                    long m, const double* __restrict__ Y, const double* __restrict__ v, int Dy,
                                    const double* __restrict__ beta) {
    const long j = (long)blockIdx.y * blockDim.x + threadIdx.x, i = blockIdx.x;
    if (j >= ld || i >= rows_pad) return;
    double g = 0.0;
    if (i < rows && j < m) {
        double yv = 0.0;
        for (int d = 0; d < Dy; ++d) yv = fma(Y[i * Dy + d], v[j * Dy + d], yv);
        g = beta[i] * fma(2.0, T[i * ld + j], yv) * W[i * ld + j];
    }
    W[i * ld + j] = g;
}
// A[i][j] *= B[i][j]  (mp x mp)
__global__ void k_mm_mul(double* __restrict__ A, const double* __restrict__ B, long mp) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < mp) A[i * mp + j] *= B[i * mp + j];
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
        const char* e = DIAG_ENV("SPARSE_FUSE_COLS");
        if (e && *e) s->fuse_cols = atoi(e) ? 1 : 0;
    }
    for (auto& e : s->ev) HIP_CHECK(hipEventCreate(&e));
    {
        const char* e = DIAG_ENV("SPARSE_KMM_OVERLAP");
        if (e && *e) s->kmm_overlap = atoi(e) ? 1 : 0;
        HIP_CHECK(hipStreamCreateWithFlags(&s->st_kmm, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_z, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_kmm, hipEventDisableTiming));
    }
    *out = s;
    return 0;
}

int mi355gp_sparse_get_profile(mi355gp_sparse* s, double* out6) {
    ARGCHK(s && out6, "mi355gp_sparse_get_profile: NULL argument");
    HIP_CHECK(hipSetDevice(s->device));
    HIP_CHECK(hipStreamSynchronize(s->st));
    double ms[PF_NUM], fl[PF_NUM];
    int nl[PF_NUM];
    if (s->mfma_prof.collect(ms, fl, nl) != 0) {
        mi355gp_set_error("mi355gp_sparse_get_profile: event timing failed");
        return -5;
    }
    for (int f = 0; f < 2; ++f) {
        out6[3 * f] = ms[f];
        out6[3 * f + 1] = fl[f];
        out6[3 * f + 2] = (double)nl[f];
    }
    return 0;
}

int mi355gp_sparse_destroy(mi355gp_sparse* s) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->st);
    free_m(s);
    double** ptrs[] = {&s->dX, &s->dY, &s->dV, &s->dBeta, &s->dRowS, &s->dRowT, &s->dRowR};
    for (auto p : ptrs)
        if (*p) (void)hipFree(*p);
    if (s->comm) rccl_comm_destroy(s->comm);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->st_kmm) {
        (void)hipStreamSynchronize(s->st_kmm);
        (void)hipStreamDestroy(s->st_kmm);
    }
    if (s->ev_z) (void)hipEventDestroy(s->ev_z);
    if (s->ev_kmm) (void)hipEventDestroy(s->ev_kmm);
    s->mfma_prof.destroy();
    if (s->st) (void)hipStreamSynchronize(s->st);
    delete s;
    return 0;
}

int mi355gp_sparse_set_data(mi355gp_sparse* s, const double* X, int64_t N, int D, const double* Y, int Dy) {
    ARGCHK(s && X && Y && N > 0 && D > 0 && Dy > 0, "mi355gp_sparse_set_data: bad arguments");
    HIP_CHECK(hipSetDevice(s->device));
    EngineShared gate(s->device);
    HIP_CHECK(hipStreamSynchronize(s->st));
    free_m(s);
    double** ptrs[] = {&s->dX, &s->dY, &s->dV, &s->dBeta, &s->dRowS, &s->dRowT, &s->dRowR};
    for (auto p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    s->n = N;
    s->D = D;
    s->Dy = Dy;
    HIP_CHECK(hipMalloc(&s->dX, sizeof(double) * N * D));
    HIP_CHECK(hipMalloc(&s->dY, sizeof(double) * N * Dy));
    HIP_CHECK(hipMalloc(&s->dV, sizeof(double) * N * Dy));
    HIP_CHECK(hipMalloc(&s->dBeta, sizeof(double) * N));
    HIP_CHECK(hipMalloc(&s->dRowS, sizeof(double) * N * Dy));
    HIP_CHECK(hipMalloc(&s->dRowT, sizeof(double) * N));
    HIP_CHECK(hipMalloc(&s->dRowR, sizeof(double) * N));
    HIP_CHECK(hipMemcpy(s->dX, X, sizeof(double) * N * D, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(s->dY, Y, sizeof(double) * N * Dy, hipMemcpyHostToDevice));
    double t = 0.0;
    s->rowYY.assign((size_t)N, 0.0);
    for (int64_t i = 0; i < N; ++i) {
        double r = 0.0;
        for (int d = 0; d < Dy; ++d) r += Y[i * Dy + d] * Y[i * Dy + d];
        s->rowYY[(size_t)i] = r;
        t += r;                                                 // get_trYYT (var_dtc.py:48-54)
    }
    s->trYYT = t;
    s->trYYT_local = t;                                        // this shard's share (the per-shard sums of the log likelihood)
    s->n_global = N;
    if (sharded(s)) {                                          // global N and tr(Y Y^T) over the shards
        double h[2] = {(double)N, t}, *d = nullptr;
        HIP_CHECK(hipMalloc(&d, sizeof(h)));
        HIP_CHECK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
        if (int rc = sparse_allreduce(s, d, 2)) return rc;
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
    EngineShared gate(s->device);
    if (s->comm) rccl_comm_destroy(s->comm);
    s->comm = nullptr;
    s->loop = nullptr;
    if (int rc = rccl_comm_create(rank, world, id128, &s->comm)) return rc;
    s->rank = rank;
    s->world = world;
    return 0;
}

// The same mode over the LOOPBACK transport: `world` contexts of this process (one host thread each, any devices... the
// 1-GPU box: all on one) that name the same group_key meet at a process-local rendezvous for every exchange step.
int mi355gp_sparse_attach_loopback(mi355gp_sparse* s, int rank, int world, int group_key) {
    ARGCHK(s && world >= 1 && rank >= 0 && rank < world, "mi355gp_sparse_attach_loopback: bad arguments");
    std::lock_guard<std::mutex> lk(g_loop_mu);
    LoopGroup*& G = g_loop_groups[group_key];
    if (!G) {
        G = new LoopGroup();
        G->world = world;
        G->bufs.assign((size_t)world, nullptr);
    }
    ARGCHK(G->world == world, "mi355gp_sparse_attach_loopback: the group exists with a different world size");
    if (s->comm) rccl_comm_destroy(s->comm);
    s->comm = nullptr;
    s->loop = G;
    s->rank = rank;
    s->world = world;
    return 0;
}

// Cholesky of an M x M matrix of the sparse path with the persistent launch's "did not run" outcomes handled in place: when the
// single-launch schedule was taken, the host reads info[0] right away (one stream sync, twice per evaluation: ~0.1 % of
// configuration 5) and, if the launch was called off at its co-residency gate or aborted, redoes the factorisation with the
// launch-per-step schedule -- on the untouched matrix, or after rebuild() if it was partly overwritten.  Doing it HERE (and not
// by repeating the evaluation) keeps a row-sharded run in step: the redo involves no collective.  info_host receives the
// LAPACK-style info (0 or the first non-positive pivot), never an abort code.
static int factor_launch(hipStream_t st, double* A, double* X, double* T, double* W, long mp, FactorWs* ws) {
    // A -> L (in place), X = L^-1 (T: scratch of the launch-per-step inverse), W = X^T X if W != NULL.  X is used as a FULL
    // matrix by the GEMMs of the M x M phase: its strictly upper tiles are zeroed here (no schedule writes them).
    HIP_CHECK(hipMemsetAsync(X, 0, sizeof(double) * mp * mp, st));
    potrf_device(st, A, mp, ws);
    trtri_device(st, A, X, T, mp, ws);
    if (W) lauum_device(st, X, W, mp, ws);
    return 0;
}
// the host side of the same: reads info[0] (one stream sync when the persistent schedule was taken) and redoes the factorisation
// on the launch-per-step schedule if the launch was called off (matrix untouched) or aborted (rebuild() first)
static int factor_check(hipStream_t st, double* A, double* X, double* T, double* W, long mp, FactorWs* ws, int* info_host,
                        const std::function<void()>& rebuild) {
    if (!ws->persist_used) {
        HIP_CHECK(hipMemcpyAsync(info_host, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
        return 0;
    }
    int info = 0;
    HIP_CHECK(hipMemcpyAsync(&info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    bool clean = false;
    if (potrf_persist_aborted(info, ws, &clean)) {
        if (!clean) rebuild();
        HIP_CHECK(hipMemsetAsync(X, 0, sizeof(double) * mp * mp, st));
        potrf_device(st, A, mp, ws);                           // persist_skip > 0: the launch-per-step schedule
        trtri_device(st, A, X, T, mp, ws);
        if (W) lauum_device(st, X, W, mp, ws);
        HIP_CHECK(hipMemcpyAsync(&info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (info >= PS_ABORT_INFO) {
            mi355gp_set_error("sparse path: the M x M factorisation aborted twice (info %d)", info);
            return -6;
        }
    }
    *info_host = info;
    return 0;
}
static int potrf_checked(hipStream_t st, double* A, double* X, double* T, double* W, long mp, FactorWs* ws, int* info_host,
                         const std::function<void()>& rebuild) {
    if (int rc = factor_launch(st, A, X, T, W, mp, ws)) return rc;
    return factor_check(st, A, X, T, W, mp, ws, info_host, rebuild);
}

// One SparseGP.parameters_changed for a SUM of kernels, scalar or per-point noise and R = Y - mean (see mi355gp.h).
// out_scalars: [0] log marginal likelihood, [1] dL/d(noise variance) (homoscedastic; 0 otherwise), [2] trace(A),
//              [3] data_fit, [4] sum(log diag LB), [5] beta (homoscedastic)
int mi355gp_vardtc_inference_sum(mi355gp_sparse* s, int nparts, const mi355gp_part* parts, const double* Z, int64_t M,
                                 const double* noise, int64_t noise_len, double extra_jitter, double* out_scalars,
                                 double* dtheta_out, double* dZ_out, double* wv_out, double* dnoise_rows_out,
                                 double* dLdm_out, double* stage_ms) {
    ARGCHK(s && s->n > 0, "mi355gp_vardtc_inference: set_data first");
    ARGCHK(parts && Z && M > 0 && out_scalars && noise, "mi355gp_vardtc_inference: bad arguments");
    ARGCHK(noise_len == 1 || noise_len == s->n, "noise must have 1 or N entries");
    const bool het = noise_len > 1;
    ARGCHK(!het || dnoise_rows_out, "per-point noise: dnoise_rows_out (N x Dy) is required");
    HIP_CHECK(hipSetDevice(s->device));
    EngineShared gate(s->device);
    const int D = s->D, Dy = s->Dy;
    if (M != s->m)
        if (int rc = alloc_m(s, M)) return rc;
    if (int rc = prepare_sparse_parts(s, nparts, parts)) return rc;
    hipStream_t st = s->st;
    const long n = s->n, m = s->m, mp = s->mp, chunk = s->chunk;
    const int groups = (D + 31) / 32;
    const size_t gsz = (size_t)groups * GP_STRIDE, hsz = (size_t)mp * (D + 1);
    // per-point precision beta_n = 1 / max(noise_n, 1e-8) (var_dtc.py:78-80), V = beta * R (:88)
    // (scalar noise: the three sums are closed forms -- unused by the homoscedastic formulas -- and the precision vector is filled on
    //  the device: the N-element host loop with its logarithms and the 1.6 MB upload cost ~1 ms per evaluation at N = 200000)
    std::vector<double> hbeta(het ? (size_t)n : (size_t)1);
    double sum_beta = 0.0, sum_logbeta = 0.0, sum_bYY = 0.0;
    if (het) {
        for (long i = 0; i < n; ++i) {
            const double b = 1.0 / fmax(noise[i], 1e-8);
            hbeta[(size_t)i] = b;
            sum_beta += b;
            sum_logbeta += log(b);
            sum_bYY += b * s->rowYY[(size_t)i];
        }
    } else {
        hbeta[0] = 1.0 / fmax(noise[0], 1e-8);
        sum_beta = hbeta[0] * (double)n;
        sum_logbeta = log(hbeta[0]) * (double)n;
        sum_bYY = hbeta[0] * s->trYYT_local;
    }
    const double beta = het ? 0.0 : hbeta[0];
    s->beta_scalar = beta;
    s->mfma_prof.on = true;
    s->mfma_prof.mask = 0x3u;
    s->mfma_prof.reset();
    s->have_result = s->winv_ok = false;
    if (het) HIP_CHECK(hipMemcpyAsync(s->dBeta, hbeta.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
    else hipLaunchKernelGGL(k_fill_const, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->dBeta, n, hbeta[0]);
    HIP_CHECK(hipMemcpyAsync(s->dZ, Z, sizeof(double) * m * D, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipEventRecord(s->ev[0], st));
    {   // V = beta * R
        const long cnt = n * Dy;
        hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, s->dY, s->dBeta, cnt, Dy, s->dV);
    }
    if (int rc = scale_for_parts(s, s->dZ, m, mp, true)) return rc;
    // Kmm + 1e-8 I (var_dtc.py:93-94) = sum of the parts' K(Z) (White on the diagonal), Lm = chol (jitchol, :95), Xm = Lm^-1.
    // They need only Z: when the factorisation is the single persistent launch (a latency-bound chain on an otherwise idle
    // GPU), the three go to a side stream and pass 1 is enqueued underneath; the launch is first in line, so its workgroups are
    // resident before pass 1 fills the remaining CUs.  info is read (and a called-off launch redone) after pass 1 is queued.
    s->h_info[0] = s->h_info[1] = 0;
    int inject = 0;                                            // fault injection for the tests: 1 / 2 hit Kmm's launch, 11 / 12 B's
    {
        const char* et = DIAG_ENV("SPARSE_PERSIST_TEST");
        if (et && *et) inject = atoi(et);
    }
    if (inject == 1 || inject == 2) s->ws.persist_test = inject, s->ws.persist_skip = 0;
    const bool overlap_kmm = s->kmm_overlap && s->st_kmm && potrf_persist_eligible(mp, &s->ws);
    hipStream_t sk = overlap_kmm ? s->st_kmm : st;
    auto rebuild_kmm = [&]() { build_kmm(s, s->Lm, s->T1, 1e-8 + extra_jitter, /*lower_only=*/1, sk); };
    if (overlap_kmm) {
        HIP_CHECK(hipEventRecord(s->ev_z, st));
        HIP_CHECK(hipStreamWaitEvent(sk, s->ev_z, 0));
    }
    rebuild_kmm();
    s->ws.ev_persist_pre = overlap_kmm ? s->ev_z : nullptr;     // (ev_z has served its purpose: reused as "progress words zeroed")
    const int rcl = factor_launch(sk, s->Lm, s->Xm, s->Tm, nullptr, mp, &s->ws);
    s->ws.ev_persist_pre = nullptr;
    if (rcl) return rcl;
    if (!overlap_kmm) {
        if (int rc = factor_check(sk, s->Lm, s->Xm, s->Tm, nullptr, mp, &s->ws, &s->h_info[0], rebuild_kmm)) return rc;
    } else if (s->ws.persist_used) {
        // pass 1 must not take the CUs' LDS before the 155 KB workgroups of the persistent launch are in place
        HIP_CHECK(hipStreamWaitEvent(st, s->ev_z, 0));
        launch_wait_persist_resident(st, &s->ws);
    }
    // ---- pass 1: psi2 = sum_n beta_n k_n k_n^T (heteroscedastic) or Kuf Kfu (then A carries beta), psi1V = Kuf V -------
    HIP_CHECK(hipMemsetAsync(s->psi1Y, 0, sizeof(double) * mp * Dy, st));
    int nch = 0;
    for (long r0 = 0; r0 < n; r0 += chunk, ++nch) {
        const long rc = (n - r0 < chunk) ? (n - r0) : chunk;
        if (int e = scale_for_parts(s, s->dX + r0 * D, rc, chunk, false)) return e;
        // the cross-covariance kernel writes rows < rc, columns < m: zero only what it leaves out
        if (m < mp) {
            if (rc < chunk || nch == 0) HIP_CHECK(hipMemsetAsync(s->Kfu, 0, sizeof(double) * chunk * mp, st));
        } else if (rc < chunk) {
            HIP_CHECK(hipMemsetAsync(s->Kfu + rc * mp, 0, sizeof(double) * (chunk - rc) * mp, st));
        }
        // one plain stationary part: K(X_chunk, Z) and the column sums of psi1^T V in ONE pass over the chunk; otherwise the
        // expression is accumulated part by part and reduced by a second pass
        int ns_fused = 0;
        if (s->parts.size() == 1 && s->parts[0].stationary() && s->fuse_cols) {
            const SPart& pt = s->parts[0];
            ns_fused = launch_kbuild_cols(st, pt.kp, pt.XtC, chunk, rc, pt.XtZ, mp, m, mp, s->Kfu, mp, s->dV + r0 * Dy, Dy, s->colPart);
        }
        if (ns_fused == 0) build_cross_chunk(s, rc, s->Kfu, s->T);
        const double* G = s->Kfu;
        if (het) {                                            // rows scaled by sqrt(beta_n) (var_dtc.py:126-129) into T
            HIP_CHECK(hipMemsetAsync(s->T + rc * mp, 0, sizeof(double) * (round_up(rc, 16L * s->splitk) - rc) * mp, st));
            launch_rowscale_sqrt(st, s->Kfu, mp, rc, mp, s->dBeta + r0, s->T);
            G = s->T;
        }
        s->mfma_prof.begin(st, 1, (double)rc * (double)m * (double)m);            // algorithmic: the lower half of psi2
        launch_gram_splitk(st, G, mp, round_up(rc, 16L * s->splitk), mp, s->splitk, nch > 0, s->psi2part);   // rows >= rc are zero
        s->mfma_prof.end(st);
        const int ns = ns_fused > 0 ? ns_fused : launch_colreduce_multi(st, s->Kfu, mp, rc, mp, s->dV + r0 * Dy, Dy, 1, Dy, 0, s->colPart);
        launch_sum_splits(st, s->colPart, mp * Dy, ns, 1, s->psi1Y);               // psi1V += Kuf V_chunk
    }
    if (overlap_kmm) {                                          // Kmm's factorisation: long finished; join the main stream
        if (int rc = factor_check(sk, s->Lm, s->Xm, s->Tm, nullptr, mp, &s->ws, &s->h_info[0], rebuild_kmm)) return rc;
        HIP_CHECK(hipEventRecord(s->ev_kmm, sk));
        HIP_CHECK(hipStreamWaitEvent(st, s->ev_kmm, 0));
    }
    hipLaunchKernelGGL(k_sym_from_lower, grid2d(mp, mp), dim3(256), 0, st, s->psi2part, mp, s->splitk, s->psi2);
    if (sharded(s)) {                                           // the one exchange step of pass 1
        if (int rc = sparse_allreduce(s, s->psi2, (size_t)mp * mp)) return rc;
        if (int rc = sparse_allreduce(s, s->psi1Y, (size_t)mp * Dy)) return rc;
    }
    HIP_CHECK(hipEventRecord(s->ev[1], st));
    // ---- M x M algebra ----------------------------------------------------------------------------------------
    // A = Lm^-1 psi2_beta Lm^-T (var_dtc.py:129-134), B = I + A (:137), LB = chol(B) (:138), XB = LB^-1
    // (Xm = Lm^-1 is lower triangular: the four products below walk only its non-zero k range, half the flops of full GEMMs)
    const int ntm = (int)(mp / NB);
    launch_trmm64(st, 0, s->Xm, mp, s->psi2, mp, s->T1, mp, ntm, ntm, 1.0);
    launch_trmm64(st, 2, s->Xm, mp, s->T1, mp, s->Amat, mp, ntm, ntm, het ? 1.0 : beta);
    auto build_B = [&]() {
        hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->Amat, 1.0, (const double*)nullptr, 0.0, 1.0, mp,
                           s->LB);
    };
    build_B();
    if (inject == 11 || inject == 12) s->ws.persist_test = inject - 10, s->ws.persist_skip = 0;
    // LB = chol(B), XB = LB^-1 and B^-1 = XB^T XB (lower tiles; :150) in one go
    if (int rc = potrf_checked(st, s->LB, s->XB, s->Tm, s->Bi, mp, &s->ws, &s->h_info[1], build_B)) return rc;
    // c = LB^-1 Lm^-1 psi1 V (:141-143), w = LB^-T c (:144), v = Lm^-T w = woodbury_vector (:145)
    launch_trmv_lower(st, s->Xm, mp, mp, s->psi1Y, Dy, s->vecA);
    launch_trmv_lower(st, s->XB, mp, mp, s->vecA, Dy, s->cvec);
    launch_trmv_lower_T(st, s->XB, mp, mp, s->cvec, Dy, s->wvec, s->trmvPart);
    launch_trmv_lower_T(st, s->Xm, mp, mp, s->wvec, Dy, s->vvec, s->trmvPart);
    // B^-1 = XB^T XB (lower tiles: computed with the factorisation above), P = Dy B^-1 + w w^T = DBi_plus_BiPBi (:150-152)
    hipLaunchKernelGGL(k_form_P, grid2d(mp, mp), dim3(256), 0, st, s->Bi, s->wvec, Dy, mp, m, s->P);
    // dL_dKmm = Lm^-T (-0.5 P - 0.5 Dy B + Dy I) Lm^-1 (:153-158);  -0.5 Dy (I + A) + Dy I = -0.5 Dy A + 0.5 Dy I
    hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->P, -0.5, s->Amat, -0.5 * Dy, 0.5 * Dy, mp, s->E);
    launch_trmm64(st, 1, s->Xm, mp, s->E, mp, s->T1, mp, ntm, ntm, 1.0);                 // Xm^T E
    launch_trmm64(st, 3, s->Xm, mp, s->T1, mp, s->dLdKmm, mp, ntm, ntm, 1.0);            // (Xm^T E) Xm
    // Q2 = dL_dpsi2_beta = 0.5 Lm^-T (Dy I - P) Lm^-1 (:220); the precision enters per row in pass 2 (:224-226,231)
    hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->P, -0.5, (const double*)nullptr, 0.0, 0.5 * Dy, mp,
                       s->E);
    launch_trmm64(st, 1, s->Xm, mp, s->E, mp, s->T1, mp, ntm, ntm, 1.0);
    launch_trmm64(st, 3, s->Xm, mp, s->T1, mp, s->Q2, mp, ntm, ntm, 1.0);
    const bool het_multi = het && Dy > 1;
    if (het_multi) {
        // several output columns with per-point noise: dL_dR (var_dtc.py:240-256) needs r_n = |LB^-1 Lm^-1 k_n|^2 on its own
        // (for Dy = 1 it folds into t_n and s_n); r_n = k_n^T Gr k_n with Gr = Lm^-T B^-1 Lm^-1, built in the Winv buffer
        hipLaunchKernelGGL(k_sym_from_lower, grid2d(mp, mp), dim3(256), 0, st, s->Bi, mp, 1, s->E);
        launch_trmm64(st, 1, s->Xm, mp, s->E, mp, s->T1, mp, ntm, ntm, 1.0);
        launch_trmm64(st, 3, s->Xm, mp, s->T1, mp, s->Winv, mp, ntm, ntm, 1.0);
    }
    hipLaunchKernelGGL(k_sparse_scalars_rows, dim3((unsigned)m), dim3(256), 0, st, s->Amat, s->P, s->LB, s->cvec, Dy, mp, m,
                       s->colPart);
    hipLaunchKernelGGL(k_sparse_scalars, dim3(1), dim3(256), 0, st, s->colPart, m, s->scal);
    HIP_CHECK(hipEventRecord(s->ev[2], st));
    // ---- pass 2: dL_dKnm = beta_n (R v^T + 2 Kfu Q2) (:219,224-226,233), its theta reductions and H^T [X~ | 1] per part --
    for (SPart& p : s->parts) {
        HIP_CHECK(hipMemsetAsync(p.gradNM, 0, sizeof(double) * gsz, st));
        HIP_CHECK(hipMemsetAsync(p.HX, 0, sizeof(double) * hsz, st));
    }
    const bool want_rows = het || dLdm_out != nullptr;
    nch = 0;
    const int one_chunk = (n <= chunk);
    for (long r0 = 0; r0 < n; r0 += chunk, ++nch) {
        const long rc = (n - r0 < chunk) ? (n - r0) : chunk;
        const long rcp = round_up(rc, NB);
        if (!one_chunk) {                                    // a single chunk is still resident from pass 1
            if (int e = scale_for_parts(s, s->dX + r0 * D, rc, chunk, false)) return e;
            if (rc < chunk) HIP_CHECK(hipMemsetAsync(s->Kfu + rc * mp, 0, sizeof(double) * (chunk - rc) * mp, st));
            build_cross_chunk(s, rc, s->Kfu, s->T);
        }
        if (het_multi) {                                     // r_n = sum_j (Kfu Gr)_nj Kfu_nj, before T is needed for anything else
            launch_gemm(st, 0, 1, rcp, mp, mp, s->Kfu, mp, s->Winv, mp, s->T, mp, 1.0, 0.0);
            launch_rowdots(st, s->Kfu, s->T, mp, rc, m, s->vvec, Dy, s->dRowS + r0 * Dy, s->dRowR + r0);
        }
        s->mfma_prof.begin(st, 0, 2.0 * (double)rc * (double)m * (double)m);
        launch_gemm(st, 0, 1, rcp, mp, mp, s->Kfu, mp, s->Q2, mp, s->T, mp, 1.0, 0.0);
        s->mfma_prof.end(st);
        // per-row reductions for dL_dm = V - Kfu v (:148) and the per-point noise gradient (t_n = sum_j T_nj Kfu_nj)
        if (want_rows) launch_rowdots(st, s->Kfu, s->T, mp, rc, m, s->vvec, Dy, s->dRowS + r0 * Dy, het ? s->dRowT + r0 : nullptr);
        // dL_dKnm is formed inside the gradient pass (no separate read-modify-write of the chunk).  D <= 16: the same pass
        // also accumulates H^T [X~ | 1] (H = dL_dKnm * (dK/dr)/r stays on chip); otherwise H is written to the Kfu buffer
        // (not needed any more for this chunk: the gradient kernels recompute the covariance from the inputs) and reduced
        // by a second pass -- T itself must survive for the parts that follow.
        const RankTerm rk{s->dY + r0 * Dy, s->vvec, Dy, 1.0, 2.0, s->dBeta + r0};
        const size_t nstat = s->parts.size();
        for (size_t pi = 0; pi < nstat; ++pi) {
            SPart& p = s->parts[pi];
            if (p.kp.kind == 4) continue;                    // White: K(X, Z) = 0, no contribution (static.py:89-93)
            int nbk = 0, ns = 0;
            // a factor of a product sees dL_dKnm TIMES the other factors' covariance (prod.py:86-99): that weight matrix is
            // materialised in the Kfu buffer (free by now) -- other factors multiplied up, then dL_dKnm formed on top
            const bool prod = sparse_other_factors(s, pi, s->Kfu, [&](int f, double* dst, const double* mul, int, bool) {
                const SPart& pf = s->parts[(size_t)f];
                launch_kbuild_cross(st, pf.kp, pf.XtC, chunk, rc, pf.XtZ, mp, m, dst, mp, 0, 0, mul);
            });
            if (prod) {
                hipLaunchKernelGGL(k_form_dLdKnm_times, dim3((unsigned)rcp, (unsigned)((mp + 255) / 256)), dim3(256), 0, st, s->T,
                                   s->Kfu, mp, rc, rcp, m, s->dY + r0 * Dy, s->vvec, Dy, s->dBeta + r0);
                nbk = grad_generic_num_blocks(rc, m);
                launch_grad_generic(st, p.kp, p.XtC, chunk, rc, p.XtZ, mp, m, 0, s->Kfu, mp, s->gradPart, GP_STRIDE,
                                    p.stationary() ? s->Kfu : nullptr, mp);         // H over the weights, in place
                if (p.stationary()) ns = launch_colreduce_multi(st, s->Kfu, mp, rc, mp, p.XtC, 1, chunk, D, 1, s->colPart);
            } else {
                if (p.stationary() && s->fuse_cols)
                    ns = launch_grad_cols(st, p.kp, p.XtC, chunk, rc, p.XtZ, mp, m, mp, s->T, mp, rk, s->gradPart, s->colPart, &nbk);
                if (ns == 0) {
                    nbk = grad_generic_num_blocks(rc, m);
                    double* Hbuf = p.stationary() ? s->Kfu : nullptr;
                    launch_grad_generic(st, p.kp, p.XtC, chunk, rc, p.XtZ, mp, m, 0, s->T, mp, s->gradPart, GP_STRIDE, Hbuf, mp, rk);
                    if (p.stationary()) ns = launch_colreduce_multi(st, s->Kfu, mp, rc, mp, p.XtC, 1, chunk, D, 1, s->colPart);
                }
            }
            for (int g = 0; g < (p.kp.ard ? groups : 1); ++g)
                launch_reduce_partials(st, s->gradPart + (long)g * nbk * GP_STRIDE, nbk, GP_STRIDE,
                                       s->gradChunk + (long)g * GP_STRIDE);
            hipLaunchKernelGGL(k_vec_axpy, dim3(1), dim3(256), 0, st, p.gradNM, s->gradChunk, (long)gsz, 1.0);
            if (p.stationary()) launch_sum_splits(st, s->colPart, mp * (D + 1), ns, 1, p.HX);
        }
    }
    if (sharded(s)) {                                           // the one exchange step of pass 2 (one buffer, one all-reduce)
        double* rb = s->redbuf;
        for (SPart& p : s->parts) {
            HIP_CHECK(hipMemcpyAsync(rb, p.gradNM, sizeof(double) * gsz, hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipMemcpyAsync(rb + gsz, p.HX, sizeof(double) * hsz, hipMemcpyDeviceToDevice, st));
            rb += gsz + hsz;
        }
        if (int rc = sparse_allreduce(s, s->redbuf, s->parts.size() * (gsz + hsz))) return rc;
        rb = s->redbuf;
        for (SPart& p : s->parts) {
            HIP_CHECK(hipMemcpyAsync(p.gradNM, rb, sizeof(double) * gsz, hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipMemcpyAsync(p.HX, rb + gsz, sizeof(double) * hsz, hipMemcpyDeviceToDevice, st));
            rb += gsz + hsz;
        }
    }
    // the M x M part: update_gradients_full(dL_dKmm, Z) and gradients_X(dL_dKmm, Z) (sparse_gp.py:114-117), per part; a factor
    // of a product sees dL_dKmm times the other factors' K(Z) (prod.py:86-99), materialised in T1
    for (size_t pi = 0; pi < s->parts.size(); ++pi) {
        SPart& p = s->parts[pi];
        const int nbk = grad_generic_num_blocks(m, m);
        const bool prod = sparse_other_factors(s, pi, s->T1, [&](int f, double* dst, const double* mul, int, bool) {
            const SPart& pf = s->parts[(size_t)f];
            launch_kbuild_cross(st, pf.kp, pf.XtZ, mp, m, pf.XtZ, mp, m, dst, mp, 0, /*diag_same=*/1, mul);
        });
        if (prod) hipLaunchKernelGGL(k_mm_mul, grid2d(mp, mp), dim3(256), 0, st, s->T1, s->dLdKmm, mp);
        launch_grad_generic(st, p.kp, p.XtZ, mp, m, p.XtZ, mp, m, 1, prod ? s->T1 : s->dLdKmm, mp, s->gradPart, GP_STRIDE,
                            p.stationary() ? s->T1 : nullptr, mp);
        for (int g = 0; g < (p.kp.ard ? groups : 1); ++g)
            launch_reduce_partials(st, s->gradPart + (long)g * nbk * GP_STRIDE, nbk, GP_STRIDE, p.gradMM + (long)g * GP_STRIDE);
        if (p.stationary()) {
            const int ns = launch_colreduce_multi(st, s->T1, mp, m, mp, p.XtZ, 1, mp, D, 1, s->colPart);
            launch_sum_splits(st, s->colPart, mp * (D + 1), ns, 0, p.HZ);
        }
    }
    HIP_CHECK(hipEventRecord(s->ev[3], st));
    // ---- small results to the host -------------------------------------------------------------------------------------
    const size_t np_ = s->parts.size();
    std::vector<double> gnm(np_ * gsz), gmm(np_ * gsz), HX(np_ * hsz), HZ(np_ * hsz), Zs(np_ * (size_t)D * mp);
    std::vector<double> rowS, rowT, rowR;
    double scal[8];
    for (size_t i = 0; i < np_; ++i) {
        SPart& p = s->parts[i];
        HIP_CHECK(hipMemcpyAsync(gnm.data() + i * gsz, p.gradNM, sizeof(double) * gsz, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(gmm.data() + i * gsz, p.gradMM, sizeof(double) * gsz, hipMemcpyDeviceToHost, st));
        if (!p.stationary()) continue;
        HIP_CHECK(hipMemcpyAsync(HX.data() + i * hsz, p.HX, sizeof(double) * hsz, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(HZ.data() + i * hsz, p.HZ, sizeof(double) * hsz, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(Zs.data() + i * (size_t)D * mp, p.XtZ, sizeof(double) * D * mp, hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipMemcpyAsync(scal, s->scal, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
    if (wv_out) HIP_CHECK(hipMemcpyAsync(wv_out, s->vvec, sizeof(double) * m * Dy, hipMemcpyDeviceToHost, st));
    if (want_rows) {
        rowS.resize((size_t)n * Dy);
        HIP_CHECK(hipMemcpyAsync(rowS.data(), s->dRowS, sizeof(double) * n * Dy, hipMemcpyDeviceToHost, st));
        if (het) {
            rowT.resize((size_t)n);
            HIP_CHECK(hipMemcpyAsync(rowT.data(), s->dRowT, sizeof(double) * n, hipMemcpyDeviceToHost, st));
            if (het_multi) {
                rowR.resize((size_t)n);
                HIP_CHECK(hipMemcpyAsync(rowR.data(), s->dRowR, sizeof(double) * n, hipMemcpyDeviceToHost, st));
            }
        }
    }
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
    const int info_m = s->h_info[0], info_b = s->h_info[1];
    if (info_m > 0) return info_m > m ? (int)m : info_m;                 // Kmm not positive definite: caller adds jitter
    if (info_b > 0) return info_b > m ? (int)m : info_b;
    // sums over ALL shards of the per-point quantities
    double glob[3] = {sum_beta, sum_logbeta, sum_bYY};
    if (sharded(s)) {
        HIP_CHECK(hipMemcpy(s->scal + 4, glob, sizeof(glob), hipMemcpyHostToDevice));
        if (int rc = sparse_allreduce(s, s->scal + 4, 3)) return rc;
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipMemcpy(glob, s->scal + 4, sizeof(glob), hipMemcpyDeviceToHost));
    }
    const double trA = scal[0], sumAP = scal[1], logLB = scal[2], data_fit = scal[3];
    const double ng = (double)s->n_global, nd = ng * Dy;
    const double kdiag = sparse_kdiag(s);                        // psi0_n = Kdiag of the expression
    // _compute_log_marginal_likelihood (var_dtc.py:264-276)
    double lik_1, lik_2;
    if (het) {
        lik_1 = -0.5 * nd * log(2.0 * M_PI) + 0.5 * Dy * glob[1] - 0.5 * glob[2];
        lik_2 = -0.5 * Dy * (glob[0] * kdiag - trA);
    } else {
        lik_1 = -0.5 * nd * (log(2.0 * M_PI) - log(beta)) - 0.5 * beta * s->trYYT;
        lik_2 = -0.5 * Dy * (beta * ng * kdiag - trA);
    }
    const double lik_3 = -(double)Dy * logLB;
    for (int i = 0; i < MI355GP_NUM_OUT; ++i) out_scalars[i] = 0.0;
    out_scalars[0] = lik_1 + lik_2 + lik_3 + 0.5 * data_fit;
    out_scalars[2] = trA;
    out_scalars[3] = data_fit;
    out_scalars[4] = logLB;
    out_scalars[5] = beta;
    if (!het) {                                                  // _compute_dL_dR (var_dtc.py:258-261)
        double dL_dR = -0.5 * nd * beta + 0.5 * s->trYYT * beta * beta;
        dL_dR += 0.5 * Dy * (ng * kdiag * beta * beta - trA * beta);
        dL_dR += beta * (0.5 * sumAP - data_fit);
        out_scalars[1] = dL_dR;
    } else {
        // per point and output column (var_dtc.py:240-256 AS WRITTEN there), with s_nd = k_n^T v_d, t_n = sum_j T_nj Kfu_nj,
        // q_n = |Lm^-1 k_n|^2, r_n = |LB^-1 Lm^-1 k_n|^2 and the identity Dy q_n = 2 t_n + Dy r_n + sum_d s_nd^2:
        //   dL_dR_nd = -b/2 + (b R_nd)^2/2 + Dy b^2 psi0/2 - b^2 t_n - (Dy - 1) b^2 r_n/2 - b^2 sum_d' s_nd'^2/2
        //              - b^2 s_nd R_nd + b^2 s_nd^2/2                       (Dy = 1: the r_n and s^2 terms cancel)
        std::vector<double> Rh((size_t)n * Dy);
        HIP_CHECK(hipMemcpy(Rh.data(), s->dY, sizeof(double) * n * Dy, hipMemcpyDeviceToHost));
        for (long i = 0; i < n; ++i) {
            const double b = hbeta[(size_t)i], b2 = b * b;
            double ss = 0.0;
            for (int d = 0; d < Dy; ++d) ss += rowS[(size_t)i * Dy + d] * rowS[(size_t)i * Dy + d];
            const double common = -0.5 * b + 0.5 * Dy * b2 * kdiag - b2 * rowT[(size_t)i] - 0.5 * b2 * ss -
                                  (het_multi ? 0.5 * (Dy - 1) * b2 * rowR[(size_t)i] : 0.0);
            for (int d = 0; d < Dy; ++d) {
                const double R = Rh[(size_t)i * Dy + d], sv = rowS[(size_t)i * Dy + d];
                dnoise_rows_out[i * Dy + d] = common + 0.5 * b2 * R * R - b2 * sv * R + 0.5 * b2 * sv * sv;
            }
        }
    }
    if (dLdm_out) {                                              // dL_dm = V - Kfu v (var_dtc.py:148)
        std::vector<double> Rh((size_t)n * Dy);
        HIP_CHECK(hipMemcpy(Rh.data(), s->dY, sizeof(double) * n * Dy, hipMemcpyDeviceToHost));
        for (long i = 0; i < n; ++i)
            for (int d = 0; d < Dy; ++d) dLdm_out[i * Dy + d] = hbeta[het ? (size_t)i : (size_t)0] * Rh[(size_t)i * Dy + d] - rowS[(size_t)i * Dy + d];
    }
    if (dtheta_out) {
        double* o = dtheta_out;
        for (size_t i = 0; i < np_; ++i) {
            const SPart& p = s->parts[i];
            const double* a = gnm.data() + i * gsz;
            const double* b = gmm.data() + i * gsz;
            // update_gradients_diag(dL_dKdiag = -0.5 Dy beta_n) (sparse_gp.py:110, stationary.py:175-184, static.py:95-96)
            // (a factor of a product: dKdiag / dvariance = the product of the other factors' variances, prod.py:67-71)
            *o++ = -0.5 * Dy * glob[0] * sparse_other_variances(s, i) + (a[0] + b[0]) / p.kp.variance;
            if (!p.stationary()) continue;
            if (!p.kp.ard) *o++ = -(a[1] + b[1]) / p.theta[1];
            else
                for (size_t k = 0; k < p.dims.size(); ++k) {
                    const int q = p.dims[k], off = (q / 32) * GP_STRIDE + 2 + (q % 32);
                    *o++ = -(a[off] + b[off]) / p.theta[1 + k];
                }
        }
    }
    if (dZ_out) {
        // gradients_X(dL_dKnm^T, Z, X) + gradients_X(dL_dKmm, Z) (sparse_gp.py:116-118), summed over the parts (add.py:84-88):
        //   sum_n H[n,m] (z~_mq - x~_nq) / l_q  +  2 sum_j Hmm[j,m] (z~_mq - z~_jq) / l_q
        for (long j = 0; j < m * D; ++j) dZ_out[j] = 0.0;
        for (size_t i = 0; i < np_; ++i) {
            const SPart& p = s->parts[i];
            if (!p.stationary()) continue;
            const double* hx = HX.data() + i * hsz;
            const double* hz = HZ.data() + i * hsz;
            const double* zs = Zs.data() + i * (size_t)D * mp;
            for (long j = 0; j < m; ++j)
                for (int q = 0; q < D; ++q) {
                    const double il = p.inv_ls[(size_t)q];
                    if (il == 0.0) continue;
                    const double z = zs[(size_t)q * mp + j];
                    const double a = z * hx[j * (D + 1) + D] - hx[j * (D + 1) + q];
                    const double b = z * hz[j * (D + 1) + D] - hz[j * (D + 1) + q];
                    dZ_out[j * D + q] += (a + 2.0 * b) * il;
                }
        }
    }
    s->have_result = true;
    return 0;
}

int mi355gp_vardtc_inference(mi355gp_sparse* s, int kind, int ard, const double* theta, const double* Z, int64_t M,
                             double noise_var, double extra_jitter, double* out_scalars, double* dtheta_out,
                             double* dZ_out, double* wv_out, double* stage_ms) {
    ARGCHK(kind >= 0 && kind <= 3 && theta, "mi355gp_vardtc_inference: bad kernel parameters");
    const mi355gp_part part{kind, ard, 0, nullptr, theta, 0};
    return mi355gp_vardtc_inference_sum(s, 1, &part, Z, M, &noise_var, 1, extra_jitter, out_scalars, dtheta_out, dZ_out, wv_out,
                                        nullptr, nullptr, stage_ms);
}

// woodbury_inv = Lm^-T (I - B^-1) Lm^-1 (var_dtc.py:206-210) into s->Winv, once per inference call
static int ensure_winv(mi355gp_sparse* s) {
    if (s->winv_ok) return 0;
    hipStream_t st = s->st;
    const long mp = s->mp;
    hipLaunchKernelGGL(k_sym_from_lower, grid2d(mp, mp), dim3(256), 0, st, s->Bi, mp, 1, s->E);
    hipLaunchKernelGGL(k_mm_axpby, grid2d(mp, mp), dim3(256), 0, st, s->E, -1.0, (const double*)nullptr, 0.0, 1.0, mp, s->E);
    launch_gemm(st, 1, 1, mp, mp, mp, s->Xm, mp, s->E, mp, s->T1, mp, 1.0, 0.0);
    launch_gemm(st, 0, 1, mp, mp, mp, s->T1, mp, s->Xm, mp, s->Winv, mp, 1.0, 0.0);
    s->winv_ok = true;
    return 0;
}

// M x M results of the last call: 0 = dL_dKmm, 1 = woodbury_inv = Lm^-T (I - B^-1) Lm^-1 (var_dtc.py:206-210),
// 2 = Lm (lower, strict upper zero), 3 = Kmm (with the 1e-8 jitter), 4 = psi2 (heteroscedastic: sum_n beta_n k_n k_n^T)
int mi355gp_sparse_fetch(mi355gp_sparse* s, int which, double* out) {
    ARGCHK(s && out && s->have_result, "mi355gp_sparse_fetch: run mi355gp_vardtc_inference first");
    HIP_CHECK(hipSetDevice(s->device));
    EngineShared gate(s->device);
    hipStream_t st = s->st;
    const long m = s->m, mp = s->mp;
    const double* src = nullptr;
    if (which == 0) src = s->dLdKmm;
    else if (which == 1) {
        if (int rc = ensure_winv(s)) return rc;
        src = s->Winv;
    } else if (which == 2) {
        launch_extract(st, s->Lm, mp, mp, 0, nullptr, 0, s->E, 0);
        src = s->E;
    } else if (which == 3) {
        build_kmm(s, s->E, s->T1, 1e-8, /*lower_only=*/0);
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

// Rows [row0, row0 + nrows) of dL_dKnm = beta_n (R_n v^T + 2 k_n^T dL_dpsi2_beta) (var_dtc.py:219-233), nrows x M row-major:
// what SparseGP._update_gradients hands to a FOREIGN kernel's update_gradients_full / gradients_X (sparse_gp.py:108-118).
// The N x M matrix is never resident: the caller walks it in row blocks (nrows <= the context's chunk size).
int mi355gp_sparse_fetch_dLdKnm(mi355gp_sparse* s, int64_t row0, int64_t nrows, double* out) {
    ARGCHK(s && out && s->have_result, "mi355gp_sparse_fetch_dLdKnm: run mi355gp_vardtc_inference first");
    ARGCHK(row0 >= 0 && nrows > 0 && row0 + nrows <= s->n && nrows <= s->chunk, "mi355gp_sparse_fetch_dLdKnm: bad row range");
    HIP_CHECK(hipSetDevice(s->device));
    EngineShared gate(s->device);
    hipStream_t st = s->st;
    const long m = s->m, mp = s->mp, rc = nrows, rcp = round_up(rc, NB);
    if (int e = scale_for_parts(s, s->dX + row0 * s->D, rc, s->chunk, false)) return e;
    HIP_CHECK(hipMemsetAsync(s->Kfu, 0, sizeof(double) * rcp * mp, st));
    build_cross_chunk(s, rc, s->Kfu, s->T);
    launch_gemm(st, 0, 1, rcp, mp, mp, s->Kfu, mp, s->Q2, mp, s->T, mp, 1.0, 0.0);
    hipLaunchKernelGGL(k_form_dLdKnm, dim3((unsigned)rcp, (unsigned)((mp + 255) / 256)), dim3(256), 0, st, s->T, mp, rc, rcp, m,
                       s->dY + row0 * s->Dy, s->vvec, s->Dy, s->dBeta + row0);
    HIP_CHECK(hipMemcpy2DAsync(out, sizeof(double) * m, s->T, sizeof(double) * mp, sizeof(double) * m, rc,
                               hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipGetLastError());
    return 0;
}

// Sparse posterior prediction on the device (Posterior._raw_predict, posterior.py:198-262, for the woodbury_inv /
// woodbury_vector representation VarDTC returns): mu = K(X*, Z) v; var = Kdiag - sum(Kx * (Winv Kx), 0) or the full
// K(X*, X*) - Kx^T Winv Kx.  The kernel (parts) must be the one of the last inference call.
int mi355gp_sparse_predict(mi355gp_sparse* s, int nparts, const mi355gp_part* parts, const double* Xnew, int64_t Mn,
                           double* mu_out, double* var_out, int full_cov) {
    ARGCHK(s && s->have_result, "mi355gp_sparse_predict: run mi355gp_vardtc_inference first");
    ARGCHK(Xnew && Mn > 0 && mu_out, "mi355gp_sparse_predict: bad arguments");
    HIP_CHECK(hipSetDevice(s->device));
    EngineShared gate(s->device);
    if (int rc = prepare_sparse_parts(s, nparts, parts)) return rc;
    hipStream_t st = s->st;
    const long m = s->m, mp = s->mp, D = s->D, Dy = s->Dy, mnp = round_up(Mn, NB), ldn = round_up(Mn, 64);
    if (int rc = scale_for_parts(s, s->dZ, m, mp, true)) return rc;
    if (int rc = ensure_winv(s)) return rc;
    double *dXn = nullptr, *dXt = nullptr, *Kx = nullptr, *Tmp = nullptr, *dMu = nullptr, *dVar = nullptr;
    auto cleanup = [&]() {
        double* ptrs[] = {dXn, dXt, Kx, Tmp, dMu, dVar};
        for (double* p : ptrs)
            if (p) (void)hipFree(p);
    };
    hipError_t e = hipMalloc(&dXn, sizeof(double) * Mn * D);
    if (e == hipSuccess) e = hipMalloc(&dXt, sizeof(double) * D * ldn);
    if (e == hipSuccess) e = hipMalloc(&Kx, sizeof(double) * mp * mnp);
    if (e == hipSuccess) e = hipMalloc(&Tmp, sizeof(double) * mp * mnp);
    if (e == hipSuccess) e = hipMalloc(&dMu, sizeof(double) * Mn * Dy);
    if (e == hipSuccess) e = hipMalloc(&dVar, sizeof(double) * (full_cov ? mnp * mnp : Mn));
    if (e != hipSuccess) {
        cleanup();
        mi355gp_set_error("mi355gp_sparse_predict: %s", hipGetErrorString(e));
        return -(1000 + (int)e);
    }
    (void)hipMemcpyAsync(dXn, Xnew, sizeof(double) * Mn * D, hipMemcpyHostToDevice, st);
    (void)hipMemsetAsync(Kx, 0, sizeof(double) * mp * mnp, st);
    if (full_cov && var_out) (void)hipMemsetAsync(dVar, 0, sizeof(double) * mnp * mnp, st);
    const double kdiag = sparse_kdiag(s);
    // K(Z, X*) and (full_cov) K(X*, X*) of the expression: every factor is evaluated with ITS scaling of the new inputs;
    // products are multiplied up in Tmp / a second M* x M* scratch
    auto scale_new = [&](const SPart& p) {
        (void)hipMemcpyAsync(s->invls, p.inv_ls.data(), sizeof(double) * D, hipMemcpyHostToDevice, st);
        launch_scale_inputs(st, dXn, Mn, (int)D, s->invls, 1, dXt, ldn);
    };
    sparse_expression(s, Kx, Tmp, true, [&](int pi, double* dst, const double* mul, int acc, bool) {
        const SPart& p = s->parts[(size_t)pi];
        scale_new(p);
        launch_kbuild_cross(st, p.kp, p.XtZ, mp, m, dXt, ldn, Mn, dst, mnp, acc, 0, mul);
    });
    if (full_cov && var_out) {
        double* scr = nullptr;
        bool prod = false;
        for (const auto& t : s->terms) prod = prod || t.size() > 1;
        if (prod && hipMalloc(&scr, sizeof(double) * mnp * mnp) != hipSuccess) {
            cleanup();
            mi355gp_set_error("mi355gp_sparse_predict: out of memory for the product scratch");
            return -3;
        }
        sparse_expression(s, dVar, scr, false, [&](int pi, double* dst, const double* mul, int acc, bool) {
            const SPart& p = s->parts[(size_t)pi];
            scale_new(p);
            launch_kbuild_cross(st, p.kp, dXt, ldn, Mn, dXt, ldn, Mn, dst, mnp, acc, /*diag_same=*/1, mul);
        });
        if (scr) {
            (void)hipStreamSynchronize(st);
            (void)hipFree(scr);
        }
    }
    launch_col_reduce(st, Kx, mnp, m, Mn, s->vvec, (int)Dy, 0.0, 0, dMu);                              // mu = Kx^T v
    if (var_out) {
        launch_gemm(st, 0, 1, mp, mnp, mp, s->Winv, mp, Kx, mnp, Tmp, mnp, 1.0, 0.0);                 // Winv Kx
        if (!full_cov)
            hipLaunchKernelGGL(k_col_dot, dim3((unsigned)((Mn + 63) / 64)), dim3(256), 0, st, Kx, Tmp, mnp, m, (long)Mn, kdiag, dVar);
        else
            launch_gemm(st, 1, 1, mnp, mnp, mp, Kx, mnp, Tmp, mnp, dVar, mnp, -1.0, 1.0);             // K** - Kx^T Winv Kx
    }
    hipError_t e2 = hipMemcpyAsync(mu_out, dMu, sizeof(double) * Mn * Dy, hipMemcpyDeviceToHost, st);
    if (var_out && e2 == hipSuccess) {
        if (!full_cov) e2 = hipMemcpyAsync(var_out, dVar, sizeof(double) * Mn, hipMemcpyDeviceToHost, st);
        else e2 = hipMemcpy2DAsync(var_out, sizeof(double) * Mn, dVar, sizeof(double) * mnp, sizeof(double) * Mn, Mn,
                                   hipMemcpyDeviceToHost, st);
    }
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
    if (e2 == hipSuccess) e2 = hipGetLastError();
    cleanup();
    if (e2 != hipSuccess) {
        mi355gp_set_error("mi355gp_sparse_predict: %s", hipGetErrorString(e2));
        return -(1000 + (int)e2);
    }
    if (var_out && !full_cov)
        for (int64_t i = 0; i < Mn; ++i) var_out[i] = var_out[i] < 1e-15 ? 1e-15 : var_out[i];      // posterior.py:248
    return 0;
}

}  // extern "C"
