// ubench_launch.hip -- what a grid of short workgroups costs on gfx950: launch + drain time of N workgroups of
// 256 threads that each run `work` dependent VALU instructions per wave, with ~124 VGPRs and 4 KiB of LDS (the
// warp kernel's footprint: 4 workgroups per CU).  Separates dispatcher-bound from issue-bound time.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BIGV>
__global__ __launch_bounds__(256) void spin(unsigned* out, int work, int sload) {
    __shared__ unsigned lds[1024];
    unsigned a = threadIdx.x + blockIdx.x;
    if (BIGV) asm volatile("v_mov_b32 v123, 0" ::: "v123");
    for (int i = 0; i < work; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
    lds[threadIdx.x] = a;
    if (a == 0x12345678u) out[threadIdx.x] = lds[(threadIdx.x + 1) & 255];
}

// persistent form: grid = slots, each workgroup loops over its share of `items` (same total work)
__global__ __launch_bounds__(256) void spin_persistent(unsigned* out, int work, int items) {
    __shared__ unsigned lds[1024];
    asm volatile("v_mov_b32 v123, 0" ::: "v123");
    unsigned a = threadIdx.x + blockIdx.x;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        for (int i = 0; i < work; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
        a += it;
    }
    lds[threadIdx.x] = a;
    if (a == 0x12345678u) out[threadIdx.x] = lds[(threadIdx.x + 1) & 255];
}

template <typename F>
float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    unsigned* out; CHECK(hipMalloc(&out, 4096));
    const int grids[] = {256, 1024, 2048, 8192, 32768};
    const int works[] = {0, 256, 1024, 4096};
    printf("%-28s", "workgroups \\ VALU instr/wave");
    for (int w : works) printf(" %9d", w);
    printf("   (us per launch, back-to-back)\n");
    for (int big = 0; big < 2; big++)
        for (int g : grids) {
            printf("%-6d WGs, %-3s VGPRs       ", g, big ? "124" : "few");
            for (int w : works) {
                float us = big ? time_us([&] { hipLaunchKernelGGL(spin<1>, dim3(g), dim3(256), 0, 0, out, w, 0); }, 20)
                               : time_us([&] { hipLaunchKernelGGL(spin<0>, dim3(g), dim3(256), 0, 0, out, w, 0); }, 20);
                printf(" %9.1f", us);
            }
            printf("\n");
        }
    for (int items : {8192, 32768}) {
        printf("persistent 1024 WGs, %-5d items", items);
        for (int w : works) printf(" %9.1f", time_us([&] { hipLaunchKernelGGL(spin_persistent, dim3(1024), dim3(256), 0, 0, out, w, items); }, 20));
        printf("\n");
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
