// Layout of a b96 / b128 LDS-DMA (global_load_lds_dwordx3 / x4) on gfx950: where do lane l's dwords land?
// hipcc --offload-arch=gfx950 -O2 tools/probe_lds_dma.hip -o tools/probe_lds_dma && tools/probe_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SIZE>
__global__ void probe(const uint32_t* src, uint32_t* out) {
    __shared__ uint32_t lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // lane l reads SIZE bytes at src + l * 16 bytes (values: dword index = 4 l + k)
    if constexpr (SIZE == 12)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * threadIdx.x), (lptr_t)&lds[0], 12, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * threadIdx.x), (lptr_t)&lds[0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<uint32_t> h(256);
    for (int i = 0; i < 256; i++) h[i] = (uint32_t)i;  // dword i: lane = i / 4, k = i % 4
    uint32_t *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (int size : {12, 16}) {
        if (size == 12) probe<12><<<1, 64>>>(d, o); else probe<16><<<1, 64>>>(d, o);
        std::vector<uint32_t> r(512);
        hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        printf("size %d: LDS dwords 0..23:", size);
        for (int i = 0; i < 24; i++) printf(" %x", r[i]);
        printf("\n   dwords 60..70:");
        for (int i = 60; i < 70; i++) printf(" %x", r[i]);
        printf("\n   dwords 188..200:");
        for (int i = 188; i < 200; i++) printf(" %x", r[i]);
        printf("\n");
    }
    return 0;
}
