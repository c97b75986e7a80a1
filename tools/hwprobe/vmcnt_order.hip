// Hardware probe (gfx950): does s_waitcnt vmcnt(N) count a wave's vector-memory LOADS and STORES in issue order?
// A slow load (cold line, agent scope) is followed by a fast store; s_waitcnt vmcnt(1) must then guarantee that the load
// (the older operation) has returned.  The kernel copies the load's destination register right after that wait and again after
// vmcnt(0); a difference means the store was retired first and the counted wait let the wave read a register whose load was
// still in flight.     hipcc --offload-arch=gfx950 -O2 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(const unsigned* __restrict__ far, unsigned* __restrict__ sink, unsigned* __restrict__ bad, long stride,
                        int iters) {
    const long lane = blockIdx.x * (long)blockDim.x + threadIdx.x;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned* p = far + (lane + (long)it * gridDim.x * blockDim.x) * stride;
        unsigned* q = sink + lane;
        unsigned early, late, poison = 0xdeadbeefu;
        asm volatile(
            "v_mov_b32 %0, %4\n\t"
            "global_load_dword %0, %2, off sc1\n\t"
            "global_store_dword %3, %4, off\n\t"
            "s_waitcnt vmcnt(1)\n\t"
            "v_mov_b32 %1, %0\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(late), "=&v"(early)
            : "v"(p), "v"(q), "v"(poison)
            : "memory");
        if (early != late) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const long stride = 64;                       // dwords: one lane per 256-byte line
    const int blocks = 256, threads = 64, iters = 64;
    const long n = (long)blocks * threads * iters * stride;
    unsigned *far, *sink, *bad;
    hipMalloc(&far, n * 4);
    hipMalloc(&sink, (long)blocks * threads * 4);
    hipMalloc(&bad, 4);
    std::vector<unsigned> h((size_t)n);
    for (long i = 0; i < n; ++i) h[(size_t)i] = (unsigned)(i * 2654435761u) | 1u;
    hipMemcpy(far, h.data(), n * 4, hipMemcpyHostToDevice);
    unsigned total = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), 0, 0, far, sink, bad, stride, iters);
        unsigned b = 0;
        hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        total += b;
    }
    printf("vmcnt order probe: %u of %ld (load, store, vmcnt(1)) sequences read the load's register before it had returned\n",
           total, (long)blocks * threads * iters * 20);
    return 0;
}
