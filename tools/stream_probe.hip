// stream_probe.hip -- the streaming ceiling for the float32 configs[1] launch's byte counts on this box: read R bytes with
// coalesced 16-byte loads, write W bytes with 16-byte (non-temporal) stores, one launch, 256-thread workgroups, every CU busy.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe
// R = 637 MB (the keystone footprint of 32 float32 1080p frames), W = 403 MB (32 x 1024^2 x 3 floats); buffers rotate over
// > 1 GB so that no launch finds its bytes in the 256 MB Infinity Cache.  The sum keeps every load alive.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void stream(const f4* __restrict__ src, long nr, f4* __restrict__ dst, long nw) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    // interleave: every thread alternates ~1.58 loads per store, like the warp (reads 637 MB, writes 403 MB)
    long ir = tid, iw = tid;
    while (ir < nr || iw < nw) {
        for (int k = 0; k < 3 && ir < nr; k++, ir += nth) acc += NT_LOAD ? __builtin_nontemporal_load(src + ir) : src[ir];
        for (int k = 0; k < 2 && iw < nw; k++, iw += nth) {
            f4 v = acc + (float)iw;
            if (NT_STORE) __builtin_nontemporal_store(v, dst + iw); else dst[iw] = v;
        }
    }
}

int main() {
    const long R = 637167744, W = 402653184;  // bytes
    const int sets = 3;
    std::vector<f4*> s(sets), d(sets);
    for (int i = 0; i < sets; i++) { hipMalloc(&s[i], R); hipMalloc(&d[i], W); hipMemset(s[i], 1, R); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 4; variant++) {
        for (int grid : {4096, 8192, 16384}) {
            std::vector<float> ts;
            for (int it = 0; it < 24; it++) {
                const int k = it % sets;
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL((stream<false, false>), dim3(grid), dim3(256), 0, 0, s[k], R / 16, d[k], W / 16);
                if (variant == 1) hipLaunchKernelGGL((stream<false, true>), dim3(grid), dim3(256), 0, 0, s[k], R / 16, d[k], W / 16);
                if (variant == 2) hipLaunchKernelGGL((stream<true, false>), dim3(grid), dim3(256), 0, 0, s[k], R / 16, d[k], W / 16);
                if (variant == 3) hipLaunchKernelGGL((stream<true, true>), dim3(grid), dim3(256), 0, 0, s[k], R / 16, d[k], W / 16);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 4) ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            const double med = ts[ts.size() / 2] * 1e-3;
            printf("loads %s stores %s grid %5d : %7.1f us  %.2f TB/s (read 637 MB + write 403 MB)\n", (variant & 2) ? "nt   " : "plain", (variant & 1) ? "nt   " : "plain",
                   grid, med * 1e6, (R + W) / med / 1e12);
        }
    }
    return 0;
}
