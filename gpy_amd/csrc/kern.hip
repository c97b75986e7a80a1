// kern.hip -- covariance assembly (Stationary.K), the fused dL/dK -> dL/dtheta reduction
// (Stationary.update_gradients_full), the alpha products and the fetch helpers.
// Reference: GPy/kern/src/stationary.py:105-168,193-243; rbf.py:51-52,177-178; stationary.py:382-386,
// 488-492,585-589; stationary_cython.pyx:53-62; inference/.../exact_gaussian_inference.py:55-72.
//
// These stages are HBM-bound (8 N^2 bytes written by the build, 4 N^2 read by the gradient pass over the
// lower triangle of Ky^-1): inputs are pre-scaled by 1/lengthscale and stored dimension-major so a 64-point
// slab of every dimension is one coalesced 512-byte read, X tiles live in LDS, and K is recomputed from X
// inside the gradient pass instead of being re-read from memory.
#include <cstdint>
#include <cstdlib>

#include "internal.h"

#define KT 64      // covariance tile edge
#define KDC 32     // input dimensions staged per LDS chunk

struct CovVal {
    double k;        // K(r)
    double dk_r;     // dK/dr * r
    double dk_or;    // dK/dr / r   (0 where r == 0 for the exponential kernel)
};

// kinds 4 / 5 are the static kernels of GPy/kern/src/static.py: White (variance on coinciding points of the symmetric
// case, :77-81) and Bias (constant, :165-167); `same` = the entry is a diagonal entry of a symmetric evaluation.
__device__ __forceinline__ double cov_k(int kind, double var, double r2, bool same = false) {
    if (kind == 4) return same ? var : 0.0;
    if (kind == 5) return var;
    if (kind == 0) return var * exp(-0.5 * r2);
    const double r = sqrt(r2);
    if (kind == 1) {
        const double s5r = 2.2360679774997896964 * r;
        return var * (1.0 + s5r + (5.0 / 3.0) * r2) * exp(-s5r);
    }
    if (kind == 2) {
        const double s3r = 1.7320508075688772935 * r;
        return var * (1.0 + s3r) * exp(-s3r);
    }
    return var * exp(-r);
}

__device__ __forceinline__ CovVal cov_all(int kind, double var, double r2, bool same = false) {
    CovVal c;
    if (kind >= 4) {
        c.k = (kind == 5 || same) ? var : 0.0;
        c.dk_r = 0.0;
        c.dk_or = 0.0;
        return c;
    }
    if (kind == 0) {
        c.k = var * exp(-0.5 * r2);
        c.dk_r = -r2 * c.k;
        c.dk_or = -c.k;
        return c;
    }
    const double r = sqrt(r2);
    if (kind == 1) {
        const double s5r = 2.2360679774997896964 * r;
        const double e = var * exp(-s5r);
        c.k = (1.0 + s5r + (5.0 / 3.0) * r2) * e;
        c.dk_or = -(5.0 / 3.0) * (1.0 + s5r) * e;
        c.dk_r = c.dk_or * r2;
        return c;
    }
    if (kind == 2) {
        const double s3r = 1.7320508075688772935 * r;
        const double e = var * exp(-s3r);
        c.k = (1.0 + s3r) * e;
        c.dk_or = -3.0 * e;
        c.dk_r = c.dk_or * r2;
        return c;
    }
    c.k = var * exp(-r);
    c.dk_r = -r * c.k;
    c.dk_or = (r != 0.0) ? -c.k / r : 0.0;
    return c;
}

// ------------------------------------------------------------------------------------------------
// Xt[q][i] = X[i][q] / l_q, zero for i in [n, ldx)
__global__ void k_scale_inputs(const double* __restrict__ X, long n, int D, const double* __restrict__ inv_ls,
                               int ard, double* __restrict__ Xt, long ldx) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)D * ldx) return;
    const int q = (int)(idx / ldx);
    const long i = idx - (long)q * ldx;
    Xt[idx] = (i < n) ? X[i * D + q] * inv_ls[ard ? q : 0] : 0.0;
}

void launch_scale_inputs(hipStream_t st, const double* X, long n, int D, const double* inv_ls, int ard,
                         double* Xt, long ldx) {
    const long total = (long)D * ldx;
    hipLaunchKernelGGL(k_scale_inputs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, n, D, inv_ls,
                       ard, Xt, ldx);
}

// Stage rows [i0, i0+64) of dims [q0, q0+qc) of a dimension-major input into LDS s[q][64].
__device__ __forceinline__ void stage_x(const double* __restrict__ Xt, long ldx, long i0, int q0, int qc,
                                        double* s, int t) {
    for (int idx = t; idx < qc * KT; idx += 256) {
        const int q = idx >> 6, ii = idx & 63;
        s[q * KT + ii] = Xt[(long)(q0 + q) * ldx + i0 + ii];
    }
}

// The COLUMN-side slab is stored permuted with rows of KTJ doubles: element ii = 4 tx + b of dimension q sits at
// q KTJ + 40 (b >> 1) + 2 tx + (b & 1).  A thread's four values are then two 16-byte reads whose 16 lanes of a ds_read_b128
// lane group cover 256 contiguous bytes: conflict-free.  In the natural layout (4 tx + b) the two reads of a lane are 32 bytes
// apart and lanes tx, tx + 8 of a group land on the same banks (2-way on every read of the slab: the 28 % of LDS cycles that
// SQ_LDS_BANK_CONFLICT showed for k_grad / k_kbuild).  The offset of the second half (40, not 32) keeps the 64-bit stores of
// the staging pass conflict-free as well (ds_write_b64: groups of 16 consecutive lanes, banks modulo 32 dwords).
#define KTJ 72
__device__ __forceinline__ int xj_pos(int ii) { return 40 * ((ii >> 1) & 1) + 2 * (ii >> 2) + (ii & 1); }
__device__ __forceinline__ void stage_xj(const double* __restrict__ Xt, long ldx, long i0, int q0, int qc, double* s, int t) {
    for (int idx = t; idx < qc * KT; idx += 256) {
        const int q = idx >> 6, ii = idx & 63;
        s[q * KTJ + xj_pos(ii)] = Xt[(long)(q0 + q) * ldx + i0 + ii];
    }
}
__device__ __forceinline__ d4 ld_xj(const double* sj, int q, int tx) {
    const d2 lo = *reinterpret_cast<const d2*>(sj + q * KTJ + 2 * tx);
    const d2 hi = *reinterpret_cast<const d2*>(sj + q * KTJ + 40 + 2 * tx);
    return d4{lo[0], lo[1], hi[0], hi[1]};
}

// r2[a][b] += sum_q (xi[q][ty*4+a] - xj[q][tx*4+b])^2
__device__ __forceinline__ void accum_r2(const double* si, const double* sj, int qc, int ty, int tx,
                                         double (&r2)[4][4]) {
    for (int q = 0; q < qc; ++q) {
        const d4 xi = *reinterpret_cast<const d4*>(si + q * KT + ty * 4);
        const d4 xj = ld_xj(sj, q, tx);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double d = xi[a] - xj[b];
                r2[a][b] = fma(d, d, r2[a][b]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Covariance assembly.  SYM: Ky = K + diag(noise + jit) into the padded npad x npad buffer (identity in the
// padding), optionally lower 64-tiles only.  !SYM: rectangular K(X1, X2) into a dense n x m buffer.
template <bool SYM>
__global__ __launch_bounds__(256) void k_kbuild(KernParams kp, const double* __restrict__ Xt1, long ld1, long n,
                                                const double* __restrict__ Xt2, long ld2, long m,
                                                double* __restrict__ out, long ldo, long nrows_out,
                                                const double* __restrict__ noise, long noise_len, double jit,
                                                int lower_only, int add_diag, int ntc, int accumulate, int diag_same,
                                                const double* mul) {
    // mul (may alias out): element-wise multiplier with out's layout -- product kernels (GPy/kern/src/prod.py:58-65)
    __shared__ __attribute__((aligned(16))) double si[KDC * KT];
    __shared__ __attribute__((aligned(16))) double sj[KDC * KTJ];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const long ti = blockIdx.x / ntc, tj = blockIdx.x % ntc;
    if (SYM && lower_only && tj > ti) return;
    const long i0 = ti * KT, j0 = tj * KT;
    double r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
    const bool real_tile = (i0 < n) && (j0 < m);
    if (real_tile) {
        for (int q0 = 0; q0 < kp.D; q0 += KDC) {
            const int qc = (kp.D - q0 < KDC) ? (kp.D - q0) : KDC;
            __syncthreads();
            stage_x(Xt1, ld1, i0, q0, qc, si, t);
            stage_xj(Xt2, ld2, j0, q0, qc, sj, t);
            __syncthreads();
            accum_r2(si, sj, qc, ty, tx, r2);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long i = i0 + ty * 4 + a;
        double v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const long j = j0 + tx * 4 + b;
            if (i < n && j < m) v[b] = cov_k(kp.kind, kp.variance, r2[a][b], (SYM || diag_same) && i == j);
            else v[b] = (SYM && i == j && !accumulate) ? 1.0 : 0.0;
        }
        if (SYM) {
            if (i < nrows_out) {
                d4* p = reinterpret_cast<d4*>(out + i * ldo + j0 + tx * 4);
                d4 o = (d4){v[0], v[1], v[2], v[3]};
                if (mul) o *= *reinterpret_cast<const d4*>(mul + i * ldo + j0 + tx * 4);
                if (add_diag && i < n) {                        // noise + jitter enter once, outside any product
                    const long d = i - (j0 + tx * 4);
                    if (d >= 0 && d < 4) o[d] += noise[noise_len > 1 ? i : 0] + jit;
                }
                if (accumulate) o += *p;                       // sum kernels (GPy/kern/src/add.py:58-72): K += K_part
                *p = o;
            }
        } else if (i < n) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long j = j0 + tx * 4 + b;
                if (j < m) {
                    const double w = mul ? v[b] * mul[i * ldo + j] : v[b];
                    out[i * ldo + j] = accumulate ? out[i * ldo + j] + w : w;
                }
            }
        }
    }
}



// K(X_chunk, Z) of the streaming sparse path (var_dtc.py:123) WITH the column reduction psi1^T V = sum_i K[i][j] V[i][d]
// (var_dtc_parallel.py:91-116) fused in: block = (column tile, row split) as in k_grad_cols; the Z slab of the column tile is
// staged once per block, every 64 x 64 tile of K is stored as 32-byte row vectors and its contribution to the column sums stays
// in registers across the block's row tiles.  The separate pass re-read the whole 3.3 GB chunk for it (k_colreduce_multi: 0.58 ms
// at configuration 5).  One plain stationary part only (no accumulation, no product): sums of kernels keep the two-pass form.
// Partials [split][column][d] are combined in fixed order by launch_sum_splits: bit reproducible.
#define KBC_DY 4
__global__ __launch_bounds__(256) void k_kbuild_cols(KernParams kp, const double* __restrict__ Xt1, long ld1, long n,
                                                     const double* __restrict__ Xt2, long ld2, long m, double* __restrict__ out,
                                                     long ldo, const double* __restrict__ V, int Dy, int ntc, int ntr,
                                                     int tiles_per_split, double* __restrict__ colpart, long mcols) {
    __shared__ __attribute__((aligned(16))) double si[KDC * KT];
    __shared__ __attribute__((aligned(16))) double sj[KDC * KTJ];
    __shared__ double sv[KT * KBC_DY];
    __shared__ double red[256];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int tj = blockIdx.x % ntc, split = blockIdx.x / ntc;
    const long j0 = (long)tj * KT;
    const int D = kp.D;                                   // <= KDC
    double csum[4][KBC_DY];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int d = 0; d < KBC_DY; ++d) csum[b][d] = 0.0;
    stage_xj(Xt2, ld2, j0, 0, D, sj, t);
    const bool fullcols = j0 + KT <= m;
    const int ti_end = ((split + 1) * tiles_per_split < ntr) ? (split + 1) * tiles_per_split : ntr;
    for (int ti = split * tiles_per_split; ti < ti_end; ++ti) {
        const long i0 = (long)ti * KT;
        __syncthreads();                                  // the previous tile's reads of si / sv are done
        stage_x(Xt1, ld1, i0, 0, D, si, t);
        if (t < KT * Dy) {
            const long i = i0 + t / Dy;
            sv[(t / Dy) * KBC_DY + t % Dy] = (i < n) ? V[i * Dy + t % Dy] : 0.0;
        }
        __syncthreads();
        double r2[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
        accum_r2(si, sj, D, ty, tx, r2);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long i = i0 + ty * 4 + a;
            d4 o;
#pragma unroll
            for (int b = 0; b < 4; ++b) o[b] = (i < n && j0 + tx * 4 + b < m) ? cov_k(kp.kind, kp.variance, r2[a][b], false) : 0.0;
            if (i < n) {
                if (fullcols) *reinterpret_cast<d4*>(out + i * ldo + j0 + tx * 4) = o;
                else
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (j0 + tx * 4 + b < m) out[i * ldo + j0 + tx * 4 + b] = o[b];
            }
#pragma unroll
            for (int d = 0; d < KBC_DY; ++d) {
                if (d < Dy) {
                    const double v = sv[(ty * 4 + a) * KBC_DY + d];
#pragma unroll
                    for (int b = 0; b < 4; ++b) csum[b][d] = fma(o[b], v, csum[b][d]);
                }
            }
        }
    }
    // column partials of this block: over the 16 row groups (ty) in fixed order
    double* cp = colpart + (long)split * mcols * Dy;
#pragma unroll
    for (int d = 0; d < KBC_DY; ++d) {
        if (d < Dy) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                __syncthreads();
                red[ty * 16 + tx] = csum[b][d];
                __syncthreads();
                if (ty == 0) {
                    double sacc = 0.0;
                    for (int r = 0; r < 16; ++r) sacc += red[r * 16 + tx];
                    const long j = j0 + tx * 4 + b;
                    if (j < mcols) cp[j * Dy + d] = sacc;
                }
            }
        }
    }
}

// returns the number of row splits (colpart: nsplit * mcols * Dy doubles), 0 if the fused form does not apply
int launch_kbuild_cols(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2, long ld2, long m,
                       long mcols, double* Kout, long ldk, const double* V, int Dy, double* colpart) {
    if (kp.D > KDC || kp.kind > 3 || Dy > KBC_DY || Dy < 1 || ldk % 4 != 0 || ((uintptr_t)Kout & 31) != 0) return 0;
    const int ntr = (int)((n + KT - 1) / KT), ntc = (int)((mcols + KT - 1) / KT);
    int nsplit = 2048 / ntc;
    if (nsplit > 64) nsplit = 64;
    if (nsplit > ntr) nsplit = ntr;
    if (nsplit < 1) nsplit = 1;
    const int tps = (ntr + nsplit - 1) / nsplit;
    nsplit = (ntr + tps - 1) / tps;
    hipLaunchKernelGGL(k_kbuild_cols, dim3((unsigned)(ntc * nsplit)), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2, ld2, m, Kout, ldk, V, Dy,
                       ntc, ntr, tps, colpart, mcols);
    return nsplit;
}

void launch_kbuild_sym(hipStream_t st, KernParams kp, const double* Xt, long ldx, long n, long npad, double* A,
                       const double* noise, long noise_len, double jit, int lower_only, int add_diag, int accumulate,
                       const double* mul) {
    const int nt = (int)(npad / KT);
    hipLaunchKernelGGL((k_kbuild<true>), dim3((unsigned)((long)nt * nt)), dim3(256), 0, st, kp, Xt, ldx, n, Xt, ldx, n,
                       A, npad, npad, noise, noise_len, jit, lower_only, add_diag, nt, accumulate, 0, mul);
}

void launch_kbuild_cross(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2,
                         long ld2, long m, double* Kout, long ldk, int accumulate, int diag_same, const double* mul) {
    const int ntr = (int)((n + KT - 1) / KT), ntc = (int)((m + KT - 1) / KT);
    hipLaunchKernelGGL((k_kbuild<false>), dim3((unsigned)((long)ntr * ntc)), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2,
                       ld2, m, Kout, ldk, n, nullptr, 0, 0.0, 0, 0, ntc, accumulate, diag_same, mul);
}

// ------------------------------------------------------------------------------------------------
// Gradient reduction.  For every (i, j): g = weight * dL_dK[i][j];
//   acc_var += g*K ; acc_iso += g*(dK/dr*r) ; acc_q += g*(dK/dr / r)*(x~_iq - x~_jq)^2   (x~ = x / l)
// FUSED: dL_dK = 0.5*(alpha_i . alpha_j - Dy*W_ij) from the lower triangle of W (off-diagonal weight 2).
// else : dL_dK read from G (n x m).
// Per-block partials [2 + 32]: [0] var, [1] iso, [2+q] lengthscale dims q_off..q_off+31.
#define GP_STRIDE 34
template <bool FUSED, bool ARD>
__global__ __launch_bounds__(256) void k_grad(KernParams kp, const double* __restrict__ Xt1, long ld1, long n,
                                              const double* __restrict__ Xt2, long ld2, long m,
                                              const double* __restrict__ G, long ldg,
                                              const double* __restrict__ alpha, int Dy, int q_off,
                                              long ntiles, int ntc, double* __restrict__ partials,
                                              double* __restrict__ Hout = nullptr, long ldh = 0, int diag_same = 0,
                                              const double* __restrict__ aa_scale = nullptr,
                                              const double* __restrict__ Mul = nullptr, long ldm = 0,
                                              RankTerm rk = RankTerm{nullptr, nullptr, 0, 0.0, 1.0, nullptr}) {
    __shared__ __attribute__((aligned(16))) double si[KDC * KT];
    __shared__ __attribute__((aligned(16))) double sj[KDC * KTJ];
    __shared__ double red[256];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double a_var = 0.0, a_iso = 0.0;
    double a_q[KDC];
#pragma unroll
    for (int q = 0; q < KDC; ++q) a_q[q] = 0.0;
    const int qcnt = ARD ? ((kp.D - q_off < KDC) ? (kp.D - q_off) : KDC) : 0;
    // dL_dK = 0.5 (sc * alpha alpha^T - Dy W): sc = 1 for the Gaussian process, (nu+N)/(nu+beta-2) for the Student-t
    // process (exact_studentt_inference.py:46), read from device memory because beta is produced on the device
    const double sc = (FUSED && aa_scale) ? aa_scale[0] : 1.0;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long ti, tj;
        if (FUSED) {   // lower-triangular enumeration
            ti = (long)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
            while (ti * (ti + 1) / 2 > tile) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
            tj = tile - ti * (ti + 1) / 2;
        } else {
            ti = tile / ntc;
            tj = tile - ti * ntc;
        }
        const long i0 = ti * KT, j0 = tj * KT;
        double r2[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
        int last_q0 = -1;
        for (int q0 = 0; q0 < kp.D; q0 += KDC) {
            const int qc = (kp.D - q0 < KDC) ? (kp.D - q0) : KDC;
            __syncthreads();
            stage_x(Xt1, ld1, i0, q0, qc, si, t);
            stage_xj(Xt2, ld2, j0, q0, qc, sj, t);
            __syncthreads();
            accum_r2(si, sj, qc, ty, tx, r2);
            last_q0 = q0;
        }
        // weights * dL_dK, then the covariance factors
        double gT[4][4];   // g * dK/dr / r
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long i = i0 + ty * 4 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long j = j0 + tx * 4 + b;
                double g = 0.0;
                if (i < n && j < m) {
                    if (FUSED) {
                        if (j <= i) {
                            double aa = 0.0;
                            for (int d = 0; d < Dy; ++d) aa = fma(alpha[i * Dy + d], alpha[j * Dy + d], aa);
                            g = 0.5 * (sc * aa - (double)Dy * G[i * ldg + j]);
                            if (j < i) g *= 2.0;
                        }
                    } else {
                        g = G[i * ldg + j];
                        if (rk.Y) {   // dL_dKnm = gscale * G + beta * Y v^T formed on the fly (var_dtc.py:219,233)
                            double yv = 0.0;
                            for (int d = 0; d < rk.Dy; ++d) yv = fma(rk.Y[i * rk.Dy + d], rk.V[j * rk.Dy + d], yv);
                            g = fma(rk.gscale, g, rk.beta * yv);
                            if (rk.rowscale) g *= rk.rowscale[i];   // per-point precision (var_dtc.py:224-226)
                        }
                    }
                    // factor of a product kernel: dL_dK times the other factors' covariances (prod.py:86-99)
                    if (Mul) g *= Mul[i * ldm + j];
                }
                const CovVal c = cov_all(kp.kind, kp.variance, r2[a][b], (FUSED || diag_same) && i == j);
                a_var = fma(g, c.k, a_var);
                if (!ARD) a_iso = fma(g, c.dk_r, a_iso);
                gT[a][b] = g * c.dk_or;
            }
        }
        if (!FUSED && Hout) {   // H = dL_dK * (dK/dr)/r for the gradients_X reductions (stationary.py:330-346)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const long i = i0 + ty * 4 + a;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const long j = j0 + tx * 4 + b;
                    if (i < n && j < m) Hout[i * ldh + j] = gT[a][b];
                }
            }
        }
        if (ARD) {
            if (last_q0 != q_off) {   // D > 32: bring the dims of this launch back into LDS
                __syncthreads();
                stage_x(Xt1, ld1, i0, q_off, qcnt, si, t);
                stage_xj(Xt2, ld2, j0, q_off, qcnt, sj, t);
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < KDC; ++q) {
                if (q < qcnt) {
                    const d4 xi = *reinterpret_cast<const d4*>(si + q * KT + ty * 4);
                    const d4 xj = ld_xj(sj, q, tx);
                    double s = 0.0;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const double d = xi[a] - xj[b];
                            s = fma(gT[a][b], d * d, s);
                        }
                    a_q[q] += s;
                }
            }
        }
    }
    // deterministic block reduction -> partials[blockIdx][...]
    double* out = partials + (long)blockIdx.x * GP_STRIDE;
    auto block_sum = [&](double v) -> double {
        __syncthreads();
        red[t] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (t < s) red[t] += red[t + s];
            __syncthreads();
        }
        return red[0];
    };
    const double sv = block_sum(a_var);
    if (t == 0) out[0] = sv;
    if (!ARD) {
        const double sl = block_sum(a_iso);
        if (t == 0) out[1] = sl;
    } else {
#pragma unroll
        for (int q = 0; q < KDC; ++q) {
            if (q < qcnt) {
                const double sq = block_sum(a_q[q]);
                if (t == 0) out[2 + q] = sq;
            }
        }
    }
}

static int pick_grad_blocks(long ntiles) { return (int)((ntiles < 2048) ? ntiles : 2048); }

int grad_num_blocks(long n) {
    const long nt = (n + KT - 1) / KT;
    return pick_grad_blocks(nt * (nt + 1) / 2);
}

int grad_generic_num_blocks(long n, long m) {
    return pick_grad_blocks(((n + KT - 1) / KT) * ((m + KT - 1) / KT));
}

// one launch per group of 32 lengthscale dimensions (ARD); partials for group gidx at partials + gidx*nblocks*GP_STRIDE
void launch_grad_fused(hipStream_t st, KernParams kp, const double* Xt, long ldx, long n, const double* W,
                       long ldw, const double* alpha, int Dy, double* partials, int stride, const double* aa_scale,
                       const double* Mul, long ldm) {
    (void)stride;
    const long nt = (n + KT - 1) / KT;
    const long ntiles = nt * (nt + 1) / 2;
    const int nb = pick_grad_blocks(ntiles);
    if (!kp.ard) {
        hipLaunchKernelGGL((k_grad<true, false>), dim3(nb), dim3(256), 0, st, kp, Xt, ldx, n, Xt, ldx, n, W, ldw,
                           alpha, Dy, 0, ntiles, (int)nt, partials, nullptr, 0, 0, aa_scale, Mul, ldm);
    } else {
        for (int q_off = 0, gidx = 0; q_off < kp.D; q_off += KDC, ++gidx)
            hipLaunchKernelGGL((k_grad<true, true>), dim3(nb), dim3(256), 0, st, kp, Xt, ldx, n, Xt, ldx, n, W, ldw,
                               alpha, Dy, q_off, ntiles, (int)nt, partials + (long)gidx * nb * GP_STRIDE, nullptr, 0, 0,
                               aa_scale, Mul, ldm);
    }
}

// ------------------------------------------------------------------------------------------------
// Gradient pass of the sparse path with the column reductions fused in (D <= 16): besides the theta partials of k_grad,
// every block accumulates  HX[j][c] = sum_i H[i][j] * x~[i][c]  (c < D) and the column sums of H (c = D) for its 64
// columns over ITS range of row tiles -- H = dL_dKnm * (dK/dr)/r never goes to memory (the separate pass wrote and
// re-read the 3.3 GB chunk).  Block = (column tile, row split); partial sums per split are combined in fixed order by
// launch_sum_splits, so the result stays bit-reproducible.  G: n x m weights, optionally formed on the fly (RankTerm).
#ifdef MI355GP_DIAG   // the VALU-only form (18 % of the issue rate at configuration 5): kept for A/B in the diagnostics build
template <bool ARD>
__global__ __launch_bounds__(256) void k_grad_cols(KernParams kp, const double* __restrict__ Xt1, long ld1, long n,
                                                   const double* __restrict__ Xt2, long ld2, long m,
                                                   const double* __restrict__ G, long ldg, RankTerm rk, int ntc,
                                                   int ntr, int tiles_per_split, double* __restrict__ partials,
                                                   double* __restrict__ colpart, long mcols, int nv) {
    constexpr int NVMAX = 17;
    __shared__ __attribute__((aligned(16))) double si[KDC * KT];
    __shared__ __attribute__((aligned(16))) double sj[KDC * KTJ];
    __shared__ double red[256];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int tj = blockIdx.x % ntc, split = blockIdx.x / ntc;
    const long j0 = (long)tj * KT;
    const int D = kp.D;                                   // <= 16: one staging pass holds every dimension
    double a_var = 0.0, a_iso = 0.0;
    double a_q[NVMAX - 1];
    double hc[4][NVMAX];
#pragma unroll
    for (int q = 0; q < NVMAX - 1; ++q) a_q[q] = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < NVMAX; ++c) hc[b][c] = 0.0;
    const int ti_end = ((split + 1) * tiles_per_split < ntr) ? (split + 1) * tiles_per_split : ntr;
    for (int ti = split * tiles_per_split; ti < ti_end; ++ti) {
        const long i0 = (long)ti * KT;
        double r2[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
        __syncthreads();
        stage_x(Xt1, ld1, i0, 0, D, si, t);
        stage_xj(Xt2, ld2, j0, 0, D, sj, t);
        __syncthreads();
        accum_r2(si, sj, D, ty, tx, r2);
        double gT[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long i = i0 + ty * 4 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long j = j0 + tx * 4 + b;
                double g = 0.0;
                if (i < n && j < m) {
                    g = G[i * ldg + j];
                    if (rk.Y) {
                        double yv = 0.0;
                        for (int d = 0; d < rk.Dy; ++d) yv = fma(rk.Y[i * rk.Dy + d], rk.V[j * rk.Dy + d], yv);
                        g = fma(rk.gscale, g, rk.beta * yv);
                        if (rk.rowscale) g *= rk.rowscale[i];
                    }
                }
                const CovVal c = cov_all(kp.kind, kp.variance, r2[a][b], false);
                a_var = fma(g, c.k, a_var);
                if (!ARD) a_iso = fma(g, c.dk_r, a_iso);
                gT[a][b] = g * c.dk_or;
            }
        }
#pragma unroll
        for (int q = 0; q < NVMAX - 1; ++q) {
            if (q < D) {
                const d4 xi = *reinterpret_cast<const d4*>(si + q * KT + ty * 4);
                if (ARD) {
                    const d4 xj = ld_xj(sj, q, tx);
                    double sacc = 0.0;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const double d = xi[a] - xj[b];
                            sacc = fma(gT[a][b], d * d, sacc);
                        }
                    a_q[q] += sacc;
                }
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int a = 0; a < 4; ++a) hc[b][q] = fma(gT[a][b], xi[a], hc[b][q]);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) hc[b][NVMAX - 1] += gT[a][b];     // column sums (the "ones" column, stored at c = D)
    }
    // theta partials of this block
    double* out = partials + (long)blockIdx.x * GP_STRIDE;
    auto block_sum = [&](double v) -> double {
        __syncthreads();
        red[t] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (t < s) red[t] += red[t + s];
            __syncthreads();
        }
        return red[0];
    };
    const double sv = block_sum(a_var);
    if (t == 0) out[0] = sv;
    if (!ARD) {
        const double sl = block_sum(a_iso);
        if (t == 0) out[1] = sl;
    } else {
#pragma unroll
        for (int q = 0; q < NVMAX - 1; ++q) {
            if (q < D) {
                const double sq = block_sum(a_q[q]);
                if (t == 0) out[2 + q] = sq;
            }
        }
    }
    // column partials: sum over the 16 row groups (ty) in fixed order, one (b, c) pair at a time
    double* cp = colpart + (long)split * mcols * nv;
#pragma unroll
    for (int c = 0; c < NVMAX; ++c) {
        const int cdst = (c == NVMAX - 1) ? D : c;
        if (c < D || c == NVMAX - 1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                __syncthreads();
                red[ty * 16 + tx] = hc[b][c];
                __syncthreads();
                if (ty == 0) {
                    double sacc = 0.0;
                    for (int r = 0; r < 16; ++r) sacc += red[r * 16 + tx];
                    const long j = j0 + tx * 4 + b;
                    if (j < mcols) cp[j * nv + cdst] = (j < m) ? sacc : 0.0;
                }
            }
        }
    }
}
#endif   // MI355GP_DIAG


// The same pass with the column reductions on the MATRIX pipe.  k_grad_cols keeps hc[4][17] per thread (136 VGPRs of accumulators:
// 256 VGPRs in all, ONE workgroup per CU, 18 % of the issue rate at configuration 5).  Here the tile H = dL_dKnm * (dK/dr)/r goes
// to LDS once and  HX[j][c] += sum_i H[i][j] x~[i][c]  is a 64 x 16 x 64 product per tile on v_mfma_f64_16x16x4 (wave w: columns
// 16 w .. 16 w + 15; A operand H^T from the LDS tile, B operand from a row-major copy of the x~ slab): 4 accumulator VGPR pairs
// instead of 68, no cross-thread reduction (every (j, c) lives in one lane) and the same fp64 pipe time as the FMAs it replaces
// (MFMA and VALU fp64 share the pipe) -- the gain is occupancy.  The column SUMS of H (the "ones" column) stay on the
// VALU (4 partial sums per thread, reduced once per block).  Rows in fixed order per block: bit reproducible.
#define GC_DY 4                                     // output columns the rank term keeps in registers (more: per-element loads)
#define GC_HS 80                                    // row stride (doubles) of the LDS H tile: rows 128 B apart in bank space
template <bool ARD>
__global__ __launch_bounds__(256, 2) void k_grad_cols_mfma(KernParams kp, const double* __restrict__ Xt1, long ld1, long n,
                                                           const double* __restrict__ Xt2, long ld2, long m,
                                                           const double* __restrict__ G, long ldg, RankTerm rk, int ntc,
                                                           int ntr, int tiles_per_split, double* __restrict__ partials,
                                                           double* __restrict__ colpart, long mcols, int nv) {
    constexpr int DM = 16;                               // D <= 16
    __shared__ __attribute__((aligned(16))) double si[DM * KT];
    __shared__ __attribute__((aligned(16))) double sj[DM * KTJ];
    __shared__ __attribute__((aligned(16))) double sit[KT * 18];   // x~ slab row-major [i][c], stride 18: the B operand
    __shared__ __attribute__((aligned(16))) double sh[KT * GC_HS];
    __shared__ double red[256];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4, lane = t & 63, w = t >> 6;
    const int tj = blockIdx.x % ntc, split = blockIdx.x / ntc;
    const long j0 = (long)tj * KT;
    const int D = kp.D;
    double a_var = 0.0, a_iso = 0.0;
    double a_q[DM];
    double csum[4] = {0.0, 0.0, 0.0, 0.0};
    d4 hx = {0.0, 0.0, 0.0, 0.0};                        // HX[j = 16 w + (lane >> 4) + 4 r][c = lane & 15]
#pragma unroll
    for (int q = 0; q < DM; ++q) a_q[q] = 0.0;
    for (int idx = t; idx < KT * 18; idx += 256) sit[idx] = 0.0;      // columns c >= D stay zero
    // this thread's four columns: are all of them there (vector loads of the weights), and the rank term's column values
    const bool vec4 = (j0 + tx * 4 + 3 < m) && (ldg % 4 == 0) && ((reinterpret_cast<unsigned long long>(G) & 31ull) == 0);
    const bool rkfast = rk.Y != nullptr && rk.Dy <= GC_DY;
    double vcol[4][GC_DY];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int d = 0; d < GC_DY; ++d)
            vcol[b][d] = (rkfast && d < rk.Dy && j0 + tx * 4 + b < m) ? rk.V[(j0 + tx * 4 + b) * rk.Dy + d] : 0.0;
    const int ti_end = ((split + 1) * tiles_per_split < ntr) ? (split + 1) * tiles_per_split : ntr;
    for (int ti = split * tiles_per_split; ti < ti_end; ++ti) {
        const long i0 = (long)ti * KT;
        double r2[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
        __syncthreads();                                 // the previous tile's MFMA reads of sh / sit are done
        for (int idx = t; idx < D * KT; idx += 256) {
            const int q = idx >> 6, ii = idx & 63;
            const double v = Xt1[(long)q * ld1 + i0 + ii];
            si[q * KT + ii] = v;
            sit[ii * 18 + q] = v;
        }
        stage_xj(Xt2, ld2, j0, 0, D, sj, t);
        __syncthreads();
        accum_r2(si, sj, D, ty, tx, r2);
        double gT[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long i = i0 + ty * 4 + a;
            // the weights of this row: one 32-byte load when the four columns exist (the scalar loads it replaces touched every
            // cache line four times), the rank term's row values once per row (they used to be fetched per element)
            d4 gv = {0.0, 0.0, 0.0, 0.0};
            double yrow[GC_DY], rs = 1.0;
            if (i < n) {
                if (vec4) gv = *reinterpret_cast<const d4*>(G + i * ldg + j0 + tx * 4);
                else
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (j0 + tx * 4 + b < m) gv[b] = G[i * ldg + j0 + tx * 4 + b];
                if (rk.Y && rkfast) {
#pragma unroll
                    for (int d = 0; d < GC_DY; ++d) yrow[d] = (d < rk.Dy) ? rk.Y[i * rk.Dy + d] : 0.0;
                    if (rk.rowscale) rs = rk.rowscale[i];
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long j = j0 + tx * 4 + b;
                double g = 0.0;
                if (i < n && j < m) {
                    g = gv[b];
                    if (rk.Y) {
                        double yv = 0.0;
                        if (rkfast) {
#pragma unroll
                            for (int d = 0; d < GC_DY; ++d)
                                if (d < rk.Dy) yv = fma(yrow[d], vcol[b][d], yv);
                        } else {
                            for (int d = 0; d < rk.Dy; ++d) yv = fma(rk.Y[i * rk.Dy + d], rk.V[j * rk.Dy + d], yv);
                        }
                        g = fma(rk.gscale, g, rk.beta * yv);
                        if (rk.rowscale) g *= rkfast ? rs : rk.rowscale[i];
                    }
                }
                const CovVal c = cov_all(kp.kind, kp.variance, r2[a][b], false);
                a_var = fma(g, c.k, a_var);
                if (!ARD) a_iso = fma(g, c.dk_r, a_iso);
                gT[a][b] = g * c.dk_or;
                csum[b] += gT[a][b];
            }
            *reinterpret_cast<d4*>(sh + (ty * 4 + a) * GC_HS + tx * 4) = (d4){gT[a][0], gT[a][1], gT[a][2], gT[a][3]};
        }
        if (ARD) {
#pragma unroll
            for (int q = 0; q < DM; ++q) {
                if (q < D) {
                    const d4 xi = *reinterpret_cast<const d4*>(si + q * KT + ty * 4);
                    const d4 xj = ld_xj(sj, q, tx);
                    double sacc = 0.0;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const double d = xi[a] - xj[b];
                            sacc = fma(gT[a][b], d * d, sacc);
                        }
                    a_q[q] += sacc;
                }
            }
        }
        __syncthreads();                                 // the H tile is complete
        // HX[16 w .., :] += H[:, 16 w ..]^T x~ : k = 4 s + (lane >> 4) over the 64 rows of the tile
        const double* ha = sh + (lane >> 4) * GC_HS + 16 * w + (lane & 15);
        const double* xb = sit + (lane >> 4) * 18 + (lane & 15);
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) hx = mfma_f64(ha[4 * s4 * GC_HS], xb[4 * s4 * 18], hx);
    }
    // theta partials of this block
    double* out = partials + (long)blockIdx.x * GP_STRIDE;
    auto block_sum = [&](double v) -> double {
        __syncthreads();
        red[t] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (t < s) red[t] += red[t + s];
            __syncthreads();
        }
        return red[0];
    };
    const double sv = block_sum(a_var);
    if (t == 0) out[0] = sv;
    if (!ARD) {
        const double sl = block_sum(a_iso);
        if (t == 0) out[1] = sl;
    } else {
#pragma unroll
        for (int q = 0; q < DM; ++q) {
            if (q < D) {
                const double sq = block_sum(a_q[q]);
                if (t == 0) out[2 + q] = sq;
            }
        }
    }
    double* cp = colpart + (long)split * mcols * nv;
    // HX: every (j, c) is one accumulator register
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long j = j0 + 16 * w + (lane >> 4) + 4 * r;
        const int c = lane & 15;
        if (c < D && j < mcols) cp[j * nv + c] = (j < m) ? hx[r] : 0.0;
    }
    // column sums of H: over the 16 row groups (ty) in fixed order
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        __syncthreads();
        red[ty * 16 + tx] = csum[b];
        __syncthreads();
        if (ty == 0) {
            double sacc = 0.0;
            for (int r = 0; r < 16; ++r) sacc += red[r * 16 + tx];
            const long j = j0 + tx * 4 + b;
            if (j < mcols) cp[j * nv + D] = (j < m) ? sacc : 0.0;
        }
    }
}

// returns the number of row splits (colpart holds nsplit * mcols * (D+1) doubles, partials ntc*nsplit blocks); 0 if the
// fused form does not apply (D > 16)
int launch_grad_cols(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2, long ld2,
                     long m, long mcols, const double* G, long ldg, RankTerm rk, double* partials, double* colpart,
                     int* nblocks_out) {
    if (kp.D > 16) return 0;
    const int ntr = (int)((n + KT - 1) / KT), ntc = (int)((mcols + KT - 1) / KT);
    int nsplit = 2048 / ntc;
    if (nsplit > 64) nsplit = 64;
    if (nsplit > ntr) nsplit = ntr;
    if (nsplit < 1) nsplit = 1;
    const int tps = (ntr + nsplit - 1) / nsplit;
    nsplit = (ntr + tps - 1) / tps;
    const int nb = ntc * nsplit;
    static const int use_mfma = [] { const char* e = DIAG_ENV("GRAD_COLS_MFMA"); return (e && *e) ? atoi(e) : 1; }();
    if (use_mfma) {
        if (kp.ard)
            hipLaunchKernelGGL((k_grad_cols_mfma<true>), dim3(nb), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2, ld2, m, G, ldg, rk, ntc,
                               ntr, tps, partials, colpart, mcols, kp.D + 1);
        else
            hipLaunchKernelGGL((k_grad_cols_mfma<false>), dim3(nb), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2, ld2, m, G, ldg, rk, ntc,
                               ntr, tps, partials, colpart, mcols, kp.D + 1);
        *nblocks_out = nb;
        return nsplit;
    }
#ifdef MI355GP_DIAG
    if (kp.ard)
        hipLaunchKernelGGL((k_grad_cols<true>), dim3(nb), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2, ld2, m, G, ldg, rk, ntc, ntr,
                           tps, partials, colpart, mcols, kp.D + 1);
    else
        hipLaunchKernelGGL((k_grad_cols<false>), dim3(nb), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2, ld2, m, G, ldg, rk, ntc, ntr,
                           tps, partials, colpart, mcols, kp.D + 1);
#endif
    *nblocks_out = nb;
    return nsplit;
}

// out[0] = (nu + n) / (nu + beta - 2) with beta = scal[0] = sum(alpha * R)   (exact_studentt_inference.py:46,51)
__global__ void k_studentt_scale(const double* __restrict__ scal, double nu, double n, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (nu + n) / (nu + scal[0] - 2.0);
}
void launch_studentt_scale(hipStream_t st, const double* scal, double nu, long n, double* out) {
    hipLaunchKernelGGL(k_studentt_scale, dim3(1), dim3(64), 0, st, scal, nu, (double)n, out);
}

void launch_grad_generic(hipStream_t st, KernParams kp, const double* Xt1, long ld1, long n, const double* Xt2,
                         long ld2, long m, int symmetric, const double* G, long ldg, double* partials,
                         int stride, double* Hout, long ldh, RankTerm rk) {
    (void)stride;
    const long ntr = (n + KT - 1) / KT, ntc = (m + KT - 1) / KT;
    const long ntiles = ntr * ntc;
    const int nb = pick_grad_blocks(ntiles);
    if (!kp.ard) {
        hipLaunchKernelGGL((k_grad<false, false>), dim3(nb), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2, ld2, m, G, ldg,
                           nullptr, 0, 0, ntiles, (int)ntc, partials, Hout, ldh, symmetric, nullptr, nullptr, 0, rk);
    } else {
        // Hout may alias G (in place): only the LAST group launch writes it, every launch reads G
        for (int q_off = 0, gidx = 0; q_off < kp.D; q_off += KDC, ++gidx)
            hipLaunchKernelGGL((k_grad<false, true>), dim3(nb), dim3(256), 0, st, kp, Xt1, ld1, n, Xt2, ld2, m, G, ldg,
                               nullptr, 0, q_off, ntiles, (int)ntc, partials + (long)gidx * nb * GP_STRIDE,
                               (q_off + KDC >= kp.D) ? Hout : nullptr, ldh, symmetric, nullptr, nullptr, 0, rk);
    }
}

// out[j][c] = sum_i M[i][j] * V(i, c), c < nv <= 33.  V(i, c) = V[i*sr + c*sc] for c < nvt, 1 for c == nvt (the column
// sums): sr = 1, sc = ld for a dimension-major input, sr = Dy, sc = 1 for a row-major (rows x Dy) one.  One thread per column, 4 row groups per block, optional row split over blockIdx.y with a
// fixed-order combine by the caller (partials [split][cols][nv]).
template <int NV>
__global__ __launch_bounds__(256) void k_colreduce_multi(const double* __restrict__ M, long ld, long rows, long cols,
                                                         const double* __restrict__ V, long sr, long sc, int nvt,
                                                         int nv, double* __restrict__ part, int nv_total, int c_off) {
    __shared__ double sv[NV][256];
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + tx;
    const long rs = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * rs, r1 = (r0 + rs < rows) ? r0 + rs : rows;
    double acc[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) acc[c] = 0.0;
    for (long ib = r0; ib < r1; ib += 256) {
        __syncthreads();
        for (int c = 0; c < nv; ++c) {
            const long i = ib + threadIdx.x;
            sv[c][threadIdx.x] = (i < r1) ? ((c < nvt) ? V[i * sr + (long)c * sc] : 1.0) : 0.0;
        }
        __syncthreads();
        const long iend = (ib + 256 < r1) ? 256 : (r1 - ib);
        if (j < cols) {
            for (long ii = g; ii < iend; ii += 4) {
                const double x = M[(ib + ii) * ld + j];
#pragma unroll
                for (int c = 0; c < NV; ++c)
                    if (c < nv) acc[c] = fma(x, sv[c][ii], acc[c]);
            }
        }
    }
    double* out = part + ((long)blockIdx.y * cols) * nv_total + c_off;   // this launch fills columns [c_off, c_off + nv) of nv_total
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        if (c >= nv) break;
        __syncthreads();
        red[g][tx] = acc[c];
        __syncthreads();
        if (g == 0 && j < cols) out[j * nv_total + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    }
}

// part must hold nsplit * cols * nv doubles (nv = nvt + ones); returns nsplit.  More than 32 V columns (input dimension
// D > 32 in gradients_X / the sparse path's dL/dZ, stationary.py:330-358 has no such limit) run as several launches of at
// most 32 columns each that fill their own column range of the same [split][cols][nv] partials; the all-ones column rides
// with the last one.
int launch_colreduce_multi(hipStream_t st, const double* M, long ld, long rows, long cols, const double* V, long sr,
                           long sc, int nvt, int ones, double* part) {
    const int nv_total = nvt + (ones ? 1 : 0);
    // 64 columns per block: the row split supplies the parallelism (>= 2048 blocks for a 32k x 2k panel)
    int nsplit = (int)((rows + 511) / 512);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 64) nsplit = 64;
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)nsplit);
    int c0 = 0;
    do {
        const int nvc = nvt - c0 < 32 ? nvt - c0 : 32;                 // V columns of this launch
        const bool lastc = c0 + nvc >= nvt;
        const int nv = nvc + ((lastc && ones) ? 1 : 0);                // <= 33
        const double* Vc = V + (long)c0 * sc;
        if (nv <= 9)
            hipLaunchKernelGGL((k_colreduce_multi<9>), grid, dim3(256), 0, st, M, ld, rows, cols, Vc, sr, sc, nvc, nv, part, nv_total, c0);
        else if (nv <= 17)
            hipLaunchKernelGGL((k_colreduce_multi<17>), grid, dim3(256), 0, st, M, ld, rows, cols, Vc, sr, sc, nvc, nv, part, nv_total, c0);
        else
            hipLaunchKernelGGL((k_colreduce_multi<33>), grid, dim3(256), 0, st, M, ld, rows, cols, Vc, sr, sc, nvc, nv, part, nv_total, c0);
        c0 += nvc;
    } while (c0 < nvt);
    return nsplit;
}

// dst[i] (+)= sum_s src[s*cnt + i]  (fixed order)
__global__ void k_sum_splits(const double* __restrict__ src, long cnt, int nsplit, int accumulate,
                             double* __restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    double s = accumulate ? dst[i] : 0.0;
    for (int k = 0; k < nsplit; ++k) s += src[(long)k * cnt + i];
    dst[i] = s;
}
void launch_sum_splits(hipStream_t st, const double* src, long cnt, int nsplit, int accumulate, double* dst) {
    hipLaunchKernelGGL(k_sum_splits, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, src, cnt, nsplit, accumulate,
                       dst);
}



// out[c] = sum_b partials[b][c], fixed order (bit-reproducible run to run)
__global__ void k_reduce_partials(const double* __restrict__ partials, int nblocks, int stride,
                                  double* __restrict__ out) {
    __shared__ double red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int b = t; b < nblocks; b += 256) s += partials[(long)b * stride + c];
    red[t] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (t < k) red[t] += red[t + k];
        __syncthreads();
    }
    if (t == 0) out[c] = red[0];
}

void launch_reduce_partials(hipStream_t st, const double* partials, int nblocks, int stride, double* out) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(stride), dim3(256), 0, st, partials, nblocks, stride, out);
}

// ------------------------------------------------------------------------------------------------
// alpha = X^T (X R) with X = L^-1 lower triangular (replaces lapack.dpotrs, GPy/util/linalg.py:116-125).
// pass 1: one wave per row, y_i = sum_{j<=i} X_ij R_j
template <int DC>
__global__ __launch_bounds__(256) void k_trmv_rows(const double* __restrict__ X, long ld, long n,
                                                   const double* __restrict__ R, int Dy, int d0,
                                                   double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double acc[DC];
#pragma unroll
    for (int d = 0; d < DC; ++d) acc[d] = 0.0;
    const double* xr = X + i * ld;
    // eight 512-byte row segments in flight per wave (one load per lane and segment was one load in flight: 0.4-0.8 TB/s, and the
    // pass sits on the CUs X^T X wants); the sums run in the same order as before: the same bits
    long j = lane;
    for (; j + 7 * 64 <= i; j += 8 * 64) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = xr[j + u * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int d = 0; d < DC; ++d)
                if (d0 + d < Dy) acc[d] = fma(x[u], R[(j + u * 64) * Dy + d0 + d], acc[d]);
    }
    for (; j <= i; j += 64) {
        const double x = xr[j];
#pragma unroll
        for (int d = 0; d < DC; ++d)
            if (d0 + d < Dy) acc[d] = fma(x, R[j * Dy + d0 + d], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        double v = acc[d];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && d0 + d < Dy) y[i * Dy + d0 + d] = v;
    }
}

// pass 2: partial column sums over chunks of `crows` rows: part[chunk][j][d] = sum_{i in chunk, i>=j} X_ij y_i
// (crows = 64 up to N = 16384: four times as many, four times shorter blocks than with 256 -- at N = 4096 the pass took
//  300 us on 256 blocks of 256 sequential row steps and was as long as the X^T X it is meant to hide under)
int trmv_chunk_rows(long n) { return n <= 16384 ? 64 : 256; }
template <int DC>
__global__ __launch_bounds__(256) void k_trmv_cols(const double* __restrict__ X, long ld, long n,
                                                   const double* __restrict__ y, int Dy, int d0,
                                                   double* __restrict__ part, int crows) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    const long c = blockIdx.y;
    const long ibeg = c * crows, iend = (ibeg + crows < n) ? ibeg + crows : n;
    if ((long)blockIdx.x * 256 >= iend) return;   // whole block above the diagonal band: no contribution
    double acc[DC];
#pragma unroll
    for (int d = 0; d < DC; ++d) acc[d] = 0.0;
    if (j < n) {
        // the rows of the chunk eight at a time (eight independent loads in flight per thread instead of one); rows above the
        // diagonal are skipped by predicate, the sums run in the same order as before: the same bits
        long i = ibeg;
        for (; i + 8 <= iend; i += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (i + u >= j) ? X[(i + u) * ld + j] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + u >= j) {
#pragma unroll
                    for (int d = 0; d < DC; ++d)
                        if (d0 + d < Dy) acc[d] = fma(x[u], y[(i + u) * Dy + d0 + d], acc[d]);
                }
        }
        for (i = (i > j ? i : j); i < iend; ++i) {
            const double x = X[i * ld + j];
#pragma unroll
            for (int d = 0; d < DC; ++d)
                if (d0 + d < Dy) acc[d] = fma(x, y[i * Dy + d0 + d], acc[d]);
        }
#pragma unroll
        for (int d = 0; d < DC; ++d)
            if (d0 + d < Dy) part[(c * n + j) * Dy + d0 + d] = acc[d];
    }
}

__global__ void k_trmv_finish(const double* __restrict__ part, long n, int Dy, long nchunks,
                              double* __restrict__ alpha, int crows) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * Dy) return;
    const long j = idx / Dy;
    double s = 0.0;
    long c = j / crows;
    for (; c + 8 <= nchunks; c += 8) {                       // eight partials in flight (was: one dependent load per chunk)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(c + u) * n * Dy + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < nchunks; ++c) s += part[c * n * Dy + idx];
    alpha[idx] = s;
}

void launch_tri_matvec(hipStream_t st, const double* X, long ld, long n, const double* R, int Dy, double* tmp,
                       double* alpha, double* partials) {
    const int crows = trmv_chunk_rows(n);
    const long nchunks = (n + crows - 1) / crows, ncolb = (n + 255) / 256;
    for (int d0 = 0; d0 < Dy; d0 += 4)
        hipLaunchKernelGGL((k_trmv_rows<4>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, X, ld, n, R, Dy, d0, tmp);
    for (int d0 = 0; d0 < Dy; d0 += 4)
        hipLaunchKernelGGL((k_trmv_cols<4>), dim3((unsigned)ncolb, (unsigned)nchunks), dim3(256), 0, st, X, ld, n,
                           tmp, Dy, d0, partials, crows);
    hipLaunchKernelGGL(k_trmv_finish, dim3((unsigned)((n * Dy + 255) / 256)), dim3(256), 0, st, partials, n, Dy,
                       nchunks, alpha, crows);
}

// ------------------------------------------------------------------------------------------------
// out3[0] = sum alpha*R ; out3[1] = sum alpha^2 ; out3[2] = trace(W) ; out3[3] = 2*sum(logsum)
// diag_out[i] = 0.5*(sum_d alpha_id^2 - Dy*W_ii)   (= diag(dL_dK), exact_gaussian_inference.py:70-72)
__global__ __launch_bounds__(1024) void k_scalars(const double* __restrict__ alpha, const double* __restrict__ R,
                                                  const double* __restrict__ W, long ldw, long n, int Dy,
                                                  const double* __restrict__ logsum, long nblk,
                                                  double* __restrict__ out4, double* __restrict__ diag_out,
                                                  const int* __restrict__ info) {
    __shared__ double red[4][1024];
    const int t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (long i = t; i < n; i += 1024) {
        double a2 = 0.0;
        for (int d = 0; d < Dy; ++d) {
            const double a = alpha[i * Dy + d];
            s0 = fma(a, R[i * Dy + d], s0);
            a2 = fma(a, a, a2);
        }
        s1 += a2;
        const double w = W ? W[i * ldw + i] : 0.0;
        s2 += w;
        if (diag_out) diag_out[i] = 0.5 * (a2 - (double)Dy * w);
    }
    for (long b = t; b < nblk; b += 1024) s3 += logsum[b];
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2; red[3][t] = s3;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if (t < k) {
            red[0][t] += red[0][t + k]; red[1][t] += red[1][t + k];
            red[2][t] += red[2][t + k]; red[3][t] += red[3][t + k];
        }
        __syncthreads();
    }
    if (t == 0) {
        out4[0] = red[0][0]; out4[1] = red[1][0]; out4[2] = red[2][0]; out4[3] = 2.0 * red[3][0];
        if (info) out4[6] = (double)info[0];      // the factorisation's LAPACK-style info rides along with the scalars
    }
}

void launch_scalars(hipStream_t st, const double* alpha, const double* R, const double* W, long ldw, long n,
                       int Dy, const double* logsum, long nblk, double* out4, double* diag_out, const int* info) {
    hipLaunchKernelGGL(k_scalars, dim3(1), dim3(1024), 0, st, alpha, R, W, ldw, n, Dy, logsum, nblk, out4, diag_out, info);
}

// ------------------------------------------------------------------------------------------------
// dense host-shaped views of padded device matrices (lazy fetch path; PCIe-bound, not on the hot loop)
__global__ void k_extract(const double* __restrict__ A, long ld, long n, int mode, const double* __restrict__ alpha,
                          int Dy, double* __restrict__ out, int transpose, const double* __restrict__ aa_scale) {
    const long nbx = (n + 255) / 256;
    const long i = blockIdx.x / nbx;
    const long j = (blockIdx.x - i * nbx) * 256 + threadIdx.x;
    if (j >= n) return;
    double v;
    if (mode == 0) {
        v = (j <= i) ? A[i * ld + j] : 0.0;
    } else {
        const long hi = i > j ? i : j, lo = i > j ? j : i;
        v = A[hi * ld + lo];
        if (mode == 2) {
            double aa = 0.0;
            for (int d = 0; d < Dy; ++d) aa = fma(alpha[i * Dy + d], alpha[j * Dy + d], aa);
            // Student-t process: the alpha alpha^T term carries (nu+N)/(nu+beta-2) (exact_studentt_inference.py:46)
            v = 0.5 * ((aa_scale ? aa_scale[0] : 1.0) * aa - (double)Dy * v);
        }
    }
    if (transpose) out[j * n + i] = v; else out[i * n + j] = v;
}

void launch_extract(hipStream_t st, const double* A, long ld, long n, int mode, const double* alpha, int Dy,
                    double* out, int transpose, const double* aa_scale) {
    hipLaunchKernelGGL(k_extract, dim3((unsigned)(((n + 255) / 256) * n)), dim3(256), 0, st, A, ld, n, mode, alpha, Dy,
                       out, transpose, aa_scale);
}

// A (npad x npad) <- dense src (n x n) + diag(noise + jit); identity in the padding
__global__ void k_pad_from_dense(const double* __restrict__ src, long n, double* __restrict__ A, long npad,
                                 const double* __restrict__ noise, long noise_len, double jit) {
    const long nbx = (npad + 255) / 256;
    const long i = blockIdx.x / nbx;
    const long j = (blockIdx.x - i * nbx) * 256 + threadIdx.x;
    if (j >= npad) return;
    double v;
    if (i < n && j < n) {
        v = src[i * n + j];
        if (i == j) v += (noise ? noise[noise_len > 1 ? i : 0] : 0.0) + jit;
    } else {
        v = (i == j) ? 1.0 : 0.0;
    }
    A[i * npad + j] = v;
}

void launch_pad_from_dense(hipStream_t st, const double* src, long n, double* A, long npad, const double* noise,
                           long noise_len, double jit) {
    hipLaunchKernelGGL(k_pad_from_dense, dim3((unsigned)(((npad + 255) / 256) * npad)), dim3(256), 0, st, src, n, A,
                       npad, noise, noise_len, jit);
}

// ------------------------------------------------------------------------------------------------
// Column reductions for prediction: 64 columns per block, 4 row groups per column, fixed-order combine.
__global__ __launch_bounds__(256) void k_col_reduce(const double* __restrict__ M, long ld, long rows, long cols,
                                                    const double* __restrict__ v, int Dy, int d, double c0, int mode,
                                                    double* __restrict__ out) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + tx;
    double s = 0.0;
    if (j < cols) {
        for (long i = g; i < rows; i += 4) {
            const double x = M[i * ld + j];
            s = (mode == 0) ? fma(x, v[i * Dy + d], s) : fma(x, x, s);
        }
    }
    red[g][tx] = s;
    __syncthreads();
    if (g == 0 && j < cols) {
        const double tot = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        if (mode == 0) out[j * Dy + d] = tot; else out[j] = c0 - tot;
    }
}

void launch_col_reduce(hipStream_t st, const double* M, long ld, long rows, long cols, const double* v, int Dy,
                       double c0, int mode, double* out) {
    const unsigned nb = (unsigned)((cols + 63) / 64);
    if (mode == 0) {
        for (int d = 0; d < Dy; ++d)
            hipLaunchKernelGGL(k_col_reduce, dim3(nb), dim3(256), 0, st, M, ld, rows, cols, v, Dy, d, c0, 0, out);
    } else {
        hipLaunchKernelGGL(k_col_reduce, dim3(nb), dim3(256), 0, st, M, ld, rows, cols, v, 1, 0, c0, 1, out);
    }
}

// y = X r (lower-triangular X) and a = X^T y as separate products (sparse path: LB^-1 Lm^-1 psi1Y etc.)
void launch_trmv_lower(hipStream_t st, const double* X, long ld, long n, const double* R, int Dy, double* y) {
    for (int d0 = 0; d0 < Dy; d0 += 4)
        hipLaunchKernelGGL((k_trmv_rows<4>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, X, ld, n, R, Dy, d0, y);
}
void launch_trmv_lower_T(hipStream_t st, const double* X, long ld, long n, const double* y, int Dy, double* out,
                         double* partials) {
    const int crows = trmv_chunk_rows(n);
    const long nchunks = (n + crows - 1) / crows, ncolb = (n + 255) / 256;
    for (int d0 = 0; d0 < Dy; d0 += 4)
        hipLaunchKernelGGL((k_trmv_cols<4>), dim3((unsigned)ncolb, (unsigned)nchunks), dim3(256), 0, st, X, ld, n, y,
                           Dy, d0, partials, crows);
    hipLaunchKernelGGL(k_trmv_finish, dim3((unsigned)((n * Dy + 255) / 256)), dim3(256), 0, st, partials, n, Dy,
                       nchunks, out, crows);
}

// ------------------------------------------------------------------------------------------------
// Row reductions of the sparse path over a resident chunk: s[i][d] = sum_j K[i][j] v[j][d] (= K(X, Z) woodbury_vector:
// dL_dm = V - s, var_dtc.py:148) and t[i] = sum_j K[i][j] T[i][j] (the per-point noise gradient, var_dtc.py:240-256).
__global__ __launch_bounds__(256) void k_rowdots(const double* __restrict__ K, const double* __restrict__ T, long ld,
                                                 long rows, long m, const double* __restrict__ v, int Dy,
                                                 double* __restrict__ out_s, double* __restrict__ t_out) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const double* kr = K + i * ld;
    double t = 0.0;
    if (t_out) {
        const double* tr = T + i * ld;
        for (long j = lane; j < m; j += 64) t = fma(kr[j], tr[j], t);
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
        if (lane == 0) t_out[i] = t;
    }
    if (out_s) {
        for (int d = 0; d < Dy; ++d) {
            double a = 0.0;
            for (long j = lane; j < m; j += 64) a = fma(kr[j], v[j * Dy + d], a);
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if (lane == 0) out_s[i * Dy + d] = a;
        }
    }
}
void launch_rowdots(hipStream_t st, const double* K, const double* T, long ld, long rows, long m, const double* v, int Dy,
                    double* out_s, double* t_out) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_rowdots, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, K, T, ld, rows, m, v, Dy, out_s, t_out);
}

__global__ void k_rowscale_sqrt(const double* __restrict__ M, long ld, long rows, long cols, const double* __restrict__ w,
                                double* __restrict__ Out) {
    const long j = (long)blockIdx.y * blockDim.x + threadIdx.x, i = blockIdx.x;
    if (j >= cols || i >= rows) return;
    Out[i * ld + j] = M[i * ld + j] * sqrt(w[i]);
}
void launch_rowscale_sqrt(hipStream_t st, const double* M, long ld, long rows, long cols, const double* w, double* Out) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_rowscale_sqrt, dim3((unsigned)rows, (unsigned)((cols + 255) / 256)), dim3(256), 0, st, M, ld, rows,
                       cols, w, Out);
}
