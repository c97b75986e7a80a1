// potf2_probe.hip -- hardware probe for the generated 16 x 16 potf2 (tools/gen_potf2.py): does gfx950 execute the DP-ALU DPP
// forms the assembler accepts (v_fmac_f64_dpp, v_rsq_f64_dpp with row_newbcast), which wait states do they need, and how
// many cycles does each variant of the whole factorisation take on one wave.
//   hipcc --offload-arch=gfx950 -O3 -I gpy_amd/csrc tools/hwprobe/potf2_probe.hip -o tools/hwprobe/potf2_probe && tools/hwprobe/potf2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
void mi355gp_set_error(const char*, ...) {}
#define TS 18
#define BC16_CASE(J) case J: return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true);
__device__ __forceinline__ double bcast16(double v, int j) {
    switch (j) {
        BC16_CASE(0) BC16_CASE(1) BC16_CASE(2) BC16_CASE(3) BC16_CASE(4) BC16_CASE(5) BC16_CASE(6) BC16_CASE(7)
        BC16_CASE(8) BC16_CASE(9) BC16_CASE(10) BC16_CASE(11) BC16_CASE(12) BC16_CASE(13) BC16_CASE(14)
        default: return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + 15, 0xf, 0xf, true);
    }
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// the compiler-scheduled form (round 1-4)
__device__ __forceinline__ int potf2_v_cpp(double (&a)[16], double (&xs)[4], int lane) {
    const int i = lane & 15, g = lane >> 4;
    int fail = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[k] = (4 * k + g == i) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double piv = readlane_d(a[j], j);
        if (!(piv > 0.0)) {
            if (fail == 0) fail = j + 1;
            piv = 1.0;
        }
        const double rd = rsqrt(piv);
        a[j] *= rd;
        const double lm = (i > j) ? a[j] : 0.0;
        const double sc = (i == j) ? rd : 1.0;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = fma(-lm, bcast16(a[j], c), a[c]);
#pragma unroll
        for (int k = 0; k <= (j >> 2); ++k) {
            xs[k] *= sc;
            xs[k] = fma(-lm, bcast16(xs[k], j), xs[k]);
        }
    }
    return fail;
}
#include "pv_fused.h"
#include "pv_pad.h"
#include "pv_prsq.h"
#include "pv_pfmac.h"
#include "pv_plain.h"
#include "pv_plainpad.h"

// ---- single instructions ----------------------------------------------------------------------------------------------------
// out[0][l] = acc + bcast(src, K) * mul through v_fmac_f64_dpp; out[1][l] = rsq(bcast(src, K)) through v_rsq_f64_dpp
template <int K>
__global__ void k_single(const double* in, double* out) {
    const int l = threadIdx.x;
    double acc = in[l], src = in[64 + l], mul = in[128 + l], y = 0.0;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_rsq_f64_dpp %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\ts_nop 4"
                 : "+v"(acc), "=&v"(y) : "v"(src), "v"(mul), "n"(K));
    out[l] = acc;
    out[64 + l] = y;
}

template <int V>
__global__ void k_potf2(const double* A, double* Lout, double* Xout, int* info, long long* cycles, int reps) {
    const int lane = threadIdx.x, i = lane & 15;
    double a0[16], a[16], xs[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) a0[c] = A[i * 16 + c];
    int fail = 0;
    const long long t0 = clock64(), w0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = a0[c];
        asm volatile("" ::: "memory");
        if (V == 0) fail = potf2_v_cpp(a, xs, lane);
        if (V == 1) potf2_v_fused(a, xs, lane);   // (generated streams: no pivot test, a bad pivot turns into NaN)
        if (V == 2) potf2_v_pad(a, xs, lane);   // (generated streams: no pivot test, a bad pivot turns into NaN)
        if (V == 3) potf2_v_prsq(a, xs, lane);   // (generated streams: no pivot test, a bad pivot turns into NaN)
        if (V == 4) potf2_v_pfmac(a, xs, lane);   // (generated streams: no pivot test, a bad pivot turns into NaN)
        if (V == 5) potf2_v_plain(a, xs, lane);   // (generated streams: no pivot test, a bad pivot turns into NaN)
        if (V == 6) potf2_v_plainpad(a, xs, lane);   // (generated streams: no pivot test, a bad pivot turns into NaN)
#pragma unroll
        for (int c = 0; c < 16; ++c) a0[c] += 1e-300 * a[c];          // keep every iteration live
    }
    const long long t1 = clock64(), w1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Lout[i * 16 + c] = (c <= i) ? a[c] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) Xout[i * 16 + 4 * k + (lane >> 4)] = xs[k];
    if (lane == 0) {
        info[0] = fail;
        cycles[0] = (t1 - t0) / reps;
        cycles[1] = (w1 - w0) * 10 / reps;              // ns (100 MHz constant clock)
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V>
static int run_variant(const char* name, const double* dA, const std::vector<double>& Lref, const std::vector<double>& Xref, int expect_info) {
    double *dL, *dX;
    int* dinfo;
    long long* dcyc;
    CK(hipMalloc(&dL, 256 * 8)); CK(hipMalloc(&dX, 256 * 8)); CK(hipMalloc(&dinfo, 4)); CK(hipMalloc(&dcyc, 16));
    hipLaunchKernelGGL(k_potf2<V>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dinfo, dcyc, 1);
    CK(hipDeviceSynchronize());
    std::vector<double> Lg(256), Xg(256);
    int info;
    CK(hipMemcpy(Lg.data(), dL, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Xg.data(), dX, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    double eL = 0, eX = 0, mL = 0, mX = 0;
    for (int q = 0; q < 256; ++q) {
        eL = fmax(eL, fabs(Lg[q] - Lref[q])); mL = fmax(mL, fabs(Lref[q]));
        eX = fmax(eX, fabs(Xg[q] - Xref[q])); mX = fmax(mX, fabs(Xref[q]));
    }
    hipLaunchKernelGGL(k_potf2<V>, dim3(1), dim3(64), 0, 0, dA, dL, dX, dinfo, dcyc, 2000);
    CK(hipDeviceSynchronize());
    long long cyc[2];
    CK(hipMemcpy(cyc, dcyc, 16, hipMemcpyDeviceToHost));
    printf("%-10s info %2d (expect %d)  |dL|/|L| %.2e  |dX|/|X| %.2e  %lld clock64 ticks = %lld ns per potf2 (incl. 16 copies + 16 fmas)\n", name,
           info, expect_info, eL / mL, eX / mX, cyc[0], cyc[1]);
    (void)hipFree(dL); (void)hipFree(dX); (void)hipFree(dinfo); (void)hipFree(dcyc);
    return 0;
}

int main() {
    // ---- single-instruction semantics
    std::vector<double> in(192), out(128);
    for (int l = 0; l < 64; ++l) { in[l] = 0.25 * l; in[64 + l] = 1.0 + l; in[128 + l] = 3.0 + 0.5 * l; }
    double *din, *dout;
    CK(hipMalloc(&din, 192 * 8)); CK(hipMalloc(&dout, 128 * 8));
    CK(hipMemcpy(din, in.data(), 192 * 8, hipMemcpyHostToDevice));
    for (int K : {0, 5, 15}) {
        if (K == 0) hipLaunchKernelGGL(k_single<0>, dim3(1), dim3(64), 0, 0, din, dout);
        if (K == 5) hipLaunchKernelGGL(k_single<5>, dim3(1), dim3(64), 0, 0, din, dout);
        if (K == 15) hipLaunchKernelGGL(k_single<15>, dim3(1), dim3(64), 0, 0, din, dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dout, 128 * 8, hipMemcpyDeviceToHost));
        int badf = 0, badr = 0;
        for (int l = 0; l < 64; ++l) {
            const double s = in[64 + (l & ~15) + K];
            if (out[l] != fma(s, in[128 + l], in[l])) ++badf;
            if (fabs(out[64 + l] - 1.0 / sqrt(s)) > 1e-6 / sqrt(s)) ++badr;
        }
        printf("row_newbcast:%-2d  v_fmac_f64_dpp wrong lanes %d (lane 17: got %.4f want %.4f)   v_rsq_f64_dpp wrong lanes %d (lane 17: got %.6f want %.6f)\n",
               K, badf, out[17], fma(in[64 + 16 + K], in[128 + 17], in[17]), badr, out[64 + 17], 1.0 / sqrt(in[64 + 16 + K]));
    }
    // ---- the factorisation: SPD tile, host reference
    std::vector<double> A(256), Lr(256, 0.0), Xr(256, 0.0);
    srand(7);
    std::vector<double> B(256);
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = (i == j) ? 2.0 : 0.0;
            for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k];
            A[i * 16 + j] = s;
        }
    for (int j = 0; j < 16; ++j) {
        double s = A[j * 16 + j];
        for (int k = 0; k < j; ++k) s -= Lr[j * 16 + k] * Lr[j * 16 + k];
        Lr[j * 16 + j] = sqrt(s);
        for (int i = j + 1; i < 16; ++i) {
            double t = A[i * 16 + j];
            for (int k = 0; k < j; ++k) t -= Lr[i * 16 + k] * Lr[j * 16 + k];
            Lr[i * 16 + j] = t / Lr[j * 16 + j];
        }
    }
    for (int c = 0; c < 16; ++c)
        for (int i = 0; i < 16; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s -= Lr[i * 16 + k] * Xr[k * 16 + c];
            Xr[i * 16 + c] = s / Lr[i * 16 + i];
        }
    double* dA;
    CK(hipMalloc(&dA, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice));
    run_variant<0>("cpp", dA, Lr, Xr, 0);
    run_variant<1>("fused", dA, Lr, Xr, 0);
    run_variant<2>("fused+pad6", dA, Lr, Xr, 0);
    run_variant<3>("plain-rsq", dA, Lr, Xr, 0);
    run_variant<4>("plain-fmac", dA, Lr, Xr, 0);
    run_variant<5>("plain", dA, Lr, Xr, 0);
    run_variant<6>("plain+pad6", dA, Lr, Xr, 0);
    // not positive definite at pivot 10 (1-based)
    A[9 * 16 + 9] = -3.0;
    CK(hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice));
    printf("pivot 10 made negative (L / X differences are meaningless here, info must be 10):\n");
    run_variant<0>("cpp", dA, Lr, Xr, 10);
    run_variant<1>("fused", dA, Lr, Xr, 10);
    run_variant<5>("plain", dA, Lr, Xr, 10);
    return 0;
}
