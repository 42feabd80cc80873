// streamprobe.hip -- the streaming ceiling of THIS box for a given byte mix, callable from bench.py in the same process and
// on the same buffers as the headline ("the harness must also measure an on-box streaming-copy ceiling and
// report both").  Measurement aid, not product code: libbevwarp.so does not contain it.
//
//   probe_stream(src, read_bytes, dst, write_bytes, nt_loads, nt_stores, grid, stream, clk)
//
// One launch reads `read_bytes` with coalesced 16-byte loads and writes `write_bytes` with 16-byte stores, every thread
// alternating ~3 loads per 2 stores (the warp's 637 MB : 403 MB mix), 256-thread workgroups.  `clk` (device, 3 x u64,
// may be NULL): one workgroup in 64 adds its lifetime in shader-clock ticks (s_memtime) and in 100 MHz reference ticks
// (s_memrealtime) and 1 -- ratio x 100 MHz = the shader clock the chip held (MI355X_MICROARCH.md, DVFS give-back item 6).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/streamprobe.hip -o tools/libstreamprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ src, long nr, f4* __restrict__ dst, long nw, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    long ir = tid, iw = tid;
    while (ir < nr || iw < nw) {
        for (int k = 0; k < 3 && ir < nr; k++, ir += nth) acc += NT_LOAD ? __builtin_nontemporal_load(src + ir) : src[ir];
        for (int k = 0; k < 2 && iw < nw; k++, iw += nth) {
            f4 v = acc + (float)iw;  // (keeps every load alive)
            if (NT_STORE)
                __builtin_nontemporal_store(v, dst + iw);
            else
                dst[iw] = v;
        }
    }
    if (clk && threadIdx.x == 0 && (blockIdx.x & 63) == 0) {  // one workgroup in 64: thousands of atomics on one line would be the kernel's time
        atomicAdd(&clk[0], __builtin_amdgcn_s_memtime() - t0);
        atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - r0);
        atomicAdd(&clk[2], 1ull);
    }
}

extern "C" int probe_stream(const void* src, long read_bytes, void* dst, long write_bytes, int nt_loads, int nt_stores, int grid, void* stream,
                            unsigned long long* clk) {
    if (!src || !dst || read_bytes < 0 || write_bytes < 0 || grid <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return -1;
    const f4* s = (const f4*)src;
    f4* d = (f4*)dst;
    const long nr = read_bytes / 16, nw = write_bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    if (nt_loads && nt_stores)
        hipLaunchKernelGGL((stream_kernel<true, true>), dim3(grid), dim3(256), 0, st, s, nr, d, nw, clk);
    else if (nt_loads)
        hipLaunchKernelGGL((stream_kernel<true, false>), dim3(grid), dim3(256), 0, st, s, nr, d, nw, clk);
    else if (nt_stores)
        hipLaunchKernelGGL((stream_kernel<false, true>), dim3(grid), dim3(256), 0, st, s, nr, d, nw, clk);
    else
        hipLaunchKernelGGL((stream_kernel<false, false>), dim3(grid), dim3(256), 0, st, s, nr, d, nw, clk);
    return (int)hipGetLastError();
}
