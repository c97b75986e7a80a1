// diag128_probe.hip -- timeline of the single-CU 128 x 128 Cholesky (chain_dev.h: diag128_factor) per 16-column step and per
// wave, and its result against a host Cholesky.  Stamps (shader cycles, clock64) of wave 0 and wave 1:
//   0 step starts | 1 potf2 done (wave 0) | 2 before barrier 1 | 3 after it | 4 before barrier 2 | 5 after it | 6 step ends
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gpy_amd/csrc tools/hwprobe/diag128_probe.hip -o tools/hwprobe/diag128_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "chain_dev.h"
void mi355gp_set_error(const char*, ...) {}

struct Stamp {
    long long* p;
    __device__ __forceinline__ void operator()(int jb, int id) const {
        const int t = threadIdx.x;
        if ((t & 63) == 0 && t < 128) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            p[((t >> 6) * 8 + jb) * 8 + id] = clock64();
        }
    }
};

__global__ __launch_bounds__(256) void k_probe(double* A, long ld, double* dinv, double* logsum, int* info, long long* st, int timed) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Tt = sm;
    double* Dv = sm + NTILE * TSZ;
    diag128_load<false>(A, ld, Tt);
    __syncthreads();
    const long long t0 = clock64();
    if (timed) diag128_factor<false, 0>(Tt, Dv, 0, dinv, info, Stamp{st});
    else diag128_factor<false, 0>(Tt, Dv, 0, dinv, info);
    const long long t1 = clock64();
    diag128_store<false>(A, ld, Tt, logsum);
    if (threadIdx.x == 0) st[128] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int n = 128;
    std::vector<double> A(n * n), B(n * n), L(n * n, 0.0);
    srand(3);
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = (i == j) ? 4.0 : 0.0;
            for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k];
            A[i * n + j] = s;
        }
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        L[j * n + j] = sqrt(s);
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / L[j * n + j];
        }
    }
    double *dA, *ddinv, *dls;
    int* dinfo;
    long long* dst;
    CK(hipMalloc(&dA, n * n * 8)); CK(hipMalloc(&ddinv, 8 * 256 * 8)); CK(hipMalloc(&dls, 8)); CK(hipMalloc(&dinfo, 4));
    CK(hipMalloc(&dst, 129 * 8));
    const int lds = (NTILE * TSZ + TSZ) * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int timed = 1; timed >= 0; --timed) {
        long long best = 1LL << 60;
        std::vector<long long> st(129);
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
            CK(hipMemset(dinfo, 0, 4));
            CK(hipMemset(dst, 0, 129 * 8));
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), lds, 0, dA, (long)n, ddinv, dls, dinfo, dst, timed);
            CK(hipDeviceSynchronize());
            std::vector<long long> s2(129);
            CK(hipMemcpy(s2.data(), dst, 129 * 8, hipMemcpyDeviceToHost));
            if (s2[128] < best) { best = s2[128]; st = s2; }
        }
        std::vector<double> Lg(n * n);
        int info;
        CK(hipMemcpy(Lg.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
        double e = 0, m = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) { e = fmax(e, fabs(Lg[i * n + j] - L[i * n + j])); m = fmax(m, fabs(L[i * n + j])); }
        printf("%s: diag128_factor %lld cycles = %.2f us at 2.4 GHz (best of 5), info %d, |L - chol| / |L| = %.2e\n",
               timed ? "with stamps" : "plain", best, best / 2400.0, info, e / m);
        if (timed) {
            printf("  jb | wave 0: potf2  ->B1   wait1  P-chain->B2  wait2  syrk+rows | wave 1: trailing  wait1  solve  wait2 | step\n");
            for (int jb = 0; jb < 8; ++jb) {
                const long long* a = &st[(0 * 8 + jb) * 8];
                const long long* b = &st[(1 * 8 + jb) * 8];
                printf("  %d  | %6lld %6lld %6lld %6lld %6lld %6lld | %6lld %6lld %6lld %6lld | %6lld\n", jb, a[1] - a[0], a[2] - a[1], a[3] - a[2],
                       a[4] - a[3], a[5] - a[4], a[6] - a[5], b[2] - b[0], b[3] - b[2], b[4] - b[3], b[5] - b[4], a[6] - a[0]);
            }
        }
    }
    return 0;
}
