// ubench_lds.hip -- LDS reads at byte-unaligned addresses (gfx950): are they correct, and what do they cost?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_lds.hip -o tools/ubench_lds && tools/ubench_lds
// Each lane reads 8 (ds_read_b64) or 12 (ds_read_b96) bytes at byte address base + stride * lane + shift from a 16 KiB
// LDS image of known bytes; the aligned variant reads the enclosing 4-byte-aligned 12-byte window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* ticks, int stride, int shift, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (uint8_t)(i * 7 + (i >> 8));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t addr = (uint32_t)(uintptr_t)(lds) + (uint32_t)(lane * stride + shift);
    uint32_t acc0 = 0, acc1 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint32_t a = (addr + (it & 7) * 192) ;
        if (MODE == 0) {  // unaligned ds_read_b64
            uint64_t v;
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
            acc0 += (uint32_t)v; acc1 += (uint32_t)(v >> 32);
        } else if (MODE == 1) {  // aligned ds_read_b96 window + funnel shift
            uint32_t w0, w1, w2;
            typedef uint32_t u3 __attribute__((ext_vector_type(3)));
            u3 v;
            asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a & ~3u));
            w0 = v.x; w1 = v.y; w2 = v.z;
            acc0 += __builtin_amdgcn_alignbit(w1, w0, a << 3); acc1 += __builtin_amdgcn_alignbit(w2, w1, a << 3);
        } else {  // two unaligned ds_read_b32
            uint32_t v0, v1;
            asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:4\n s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1) : "v"(a));
            acc0 += v0; acc1 += v1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * 256 + threadIdx.x) * 2] = acc0;
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = acc1;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256 * 4, iters = 512;
    uint32_t* out; unsigned long long* ticks;
    CHECK(hipMalloc(&out, blocks * 256 * 8)); CHECK(hipMalloc(&ticks, blocks * 8));
    std::vector<uint32_t> h(blocks * 256 * 2); std::vector<unsigned long long> ht(blocks);
    auto img = [](int i) { return (uint8_t)(i * 7 + (i >> 8)); };
    for (int mode = 0; mode < 3; mode++)
        for (int stride : {3, 5, 8, 12}) for (int shift : {0, 1, 2, 3}) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, ticks, stride, shift, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, ticks, stride, shift, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, ticks, stride, shift, iters);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ht.data(), ticks, blocks * 8, hipMemcpyDeviceToHost));
            // expected sums for thread 5 of block 0
            int lane = 5; uint32_t e0 = 0, e1 = 0;
            for (int it = 0; it < iters; it++) {
                int a = lane * stride + shift + (it & 7) * 192;
                uint32_t v0 = 0, v1 = 0;
                for (int b = 0; b < 4; b++) { v0 |= (uint32_t)img(a + b) << (8 * b); v1 |= (uint32_t)img(a + 4 + b) << (8 * b); }
                e0 += v0; e1 += v1;
            }
            const bool ok = h[5 * 2] == e0 && h[5 * 2 + 1] == e1;
            double mean = 0; for (auto t : ht) mean += (double)t; mean /= blocks;
            printf("mode %d (%s) stride %2d shift %d : %s  %.1f ticks per read+wait (4 waves per CU-slot)\n", mode,
                   mode == 0 ? "ds_read_b64 unaligned" : mode == 1 ? "ds_read_b96 aligned+funnel" : "2 x ds_read_b32 unaligned", stride, shift, ok ? "correct" : "WRONG", mean / iters);
        }
    return 0;
}
