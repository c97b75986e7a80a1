// valu_probe.hip -- issue cost and dependent latency (shader cycles) of the fp64 instructions the 16 x 16 potf2 is made of, one wave alone
// on its SIMD.   hipcc --offload-arch=gfx950 -O3 tools/hwprobe/valu_probe.hip -o tools/hwprobe/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))
template <int V>
__global__ void k(long long* out, double* sink) {
    double a = 1.0 + threadIdx.x, b = 0.5, c = 0.25, d = 2.0, e = 3.0, f = 4.0, g = 5.0, h = 6.0;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0}, m3 = {0, 0, 0, 0};
    const long long t0 = clock64();
    if (V == 0) asm volatile(R64("v_fma_f64 %0, %1, %2, %0\n") : "+v"(a) : "v"(b), "v"(c));                        // dependent fma
    if (V == 1) asm volatile(R64("v_fma_f64 %0, %4, %5, %0\nv_fma_f64 %1, %4, %5, %1\nv_fma_f64 %2, %4, %5, %2\nv_fma_f64 %3, %4, %5, %3\n")
                             : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));                                  // 4 independent chains
    if (V == 2) asm volatile(R64("s_nop 1\nv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a) : "v"(b));   // dependent fused
    if (V == 3) asm volatile(R64("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\nv_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                                 "v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\nv_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n")
                             : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));                                  // 4 independent fused
    if (V == 4) asm volatile(R64("v_rsq_f64 %0, %0\ns_nop 0\n") : "+v"(a));                                           // dependent rsq
    if (V == 5) asm volatile(R64("v_rsq_f64 %0, %4\nv_rsq_f64 %1, %4\nv_rsq_f64 %2, %4\nv_rsq_f64 %3, %4\n")
                             : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));                                          // independent rsq
    if (V == 6) asm volatile(R64("s_nop 1\nv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a));    // dependent mov dpp
    if (V == 7) asm volatile(R64("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\nv_mov_b64_dpp %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                                 "v_mov_b64_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\nv_mov_b64_dpp %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                             : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));
    if (V == 8) asm volatile(R64("v_mul_f64 %0, %0, %1\n") : "+v"(a) : "v"(b));                                       // dependent mul
    if (V == 9) asm volatile(R64("v_cndmask_b32 %0, %1, %2, vcc\nv_cndmask_b32 %0, %1, %2, vcc\nv_cndmask_b32 %0, %1, %2, vcc\nv_cndmask_b32 %0, %1, %2, vcc\n")
                             : "+v"(*(int*)&g) : "v"(1), "v"(2) : "vcc");
    if (V == 10) asm volatile(R64("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n") : "+v"(m0) : "v"(b), "v"(c));            // dependent fp64 MFMA
    if (V == 11) asm volatile(R64("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\nv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\nv_mfma_f64_16x16x4_f64 %2, %4, %5, %2\nv_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n")
                              : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(b), "v"(c));                             // 4 independent MFMA chains
    if (V == 12) asm volatile(R64("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\nv_fma_f64 %1, %4, %5, %1\nv_fma_f64 %2, %4, %5, %2\nv_fma_f64 %3, %4, %5, %3\n"
                                  "v_fma_f64 %1, %4, %5, %1\nv_fma_f64 %2, %4, %5, %2\nv_fma_f64 %3, %4, %5, %3\nv_fma_f64 %1, %4, %5, %1\nv_fma_f64 %2, %4, %5, %2\n")
                              : "+v"(m0), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));                                // 1 dependent MFMA + 8 independent fmas per round
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[V] = t1 - t0;
    sink[threadIdx.x] = a + d + e + f + g + h + m0[0] + m1[1] + m2[2] + m3[3];
}
int main() {
    long long* d; double* s;
    hipMalloc(&d, 16 * 8); hipMalloc(&s, 64 * 8);
#define RUN(V) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d, s);
    for (int rep = 0; rep < 2; ++rep) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) }
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, d, 16 * 8, hipMemcpyDeviceToHost);
    const char* nm[] = {"v_fma_f64 dependent (64)", "v_fma_f64 4 chains (256)", "s_nop 1 + v_fmac_f64_dpp dependent (64)", "v_fmac_f64_dpp 4 chains (256)",
                        "v_rsq_f64 + s_nop 0 dependent (64)", "v_rsq_f64 independent (256)", "s_nop 1 + v_mov_b64_dpp dependent (64)", "v_mov_b64_dpp independent (256)",
                        "v_mul_f64 dependent (64)", "v_cndmask_b32 (256)", "mfma f64 16x16x4 dependent (64)", "mfma f64 4 chains (256)",
                        "per round: 1 dependent mfma + 8 v_fma_f64 (64 rounds)"};
    const int cnt[] = {64, 256, 64, 256, 64, 256, 64, 256, 64, 256, 64, 256, 64};
    for (int v = 0; v < 13; ++v) printf("%-58s %7lld cycles  = %6.1f per instruction / round\n", nm[v], h[v], (double)h[v] / cnt[v]);
    return 0;
}
