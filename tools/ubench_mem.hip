// ubench_mem.hip -- per-CU cost of vector-memory instructions by access shape (gfx950).  Every CU runs
// 8 waves that loop over an L2-resident window; reports shader cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 256;
struct U3 { uint32_t x, y, z; };

// MODE: 0 load x4 coalesced (16 B/lane contiguous)   1 load x2 contiguous (8 B/lane)   2 load x2, 24 B lane stride
//       3 load x2 scattered (lane stride 1000 B)       4 load x2 unaligned (23 B stride, +1)  5 load x1 contiguous
//       6 store x4 contiguous  7 store x2 contiguous  8 store x2 24 B stride  9 store x3 contiguous (12 B/lane)
//       10 store x4 48 B stride   11 load x4 48 B stride  12 load x3 contiguous (12 B/lane) 13 store x1 contiguous
//       14 load x4: 16 lanes contiguous per row, 4 rows 6 KB apart (tile shape)
template <int MODE>
__global__ __launch_bounds__(512) void k(uint8_t* buf, unsigned long long* cyc, uint32_t* sink, size_t win) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* base = buf + (size_t)blockIdx.x * win;
    size_t off;
    if (MODE == 0 || MODE == 6) off = lane * 16; else if (MODE == 1 || MODE == 7) off = lane * 8; else if (MODE == 2 || MODE == 8) off = lane * 24;
    else if (MODE == 3) off = lane * 1000; else if (MODE == 4) off = lane * 23 + 1; else if (MODE == 5 || MODE == 13) off = lane * 4;
    else if (MODE == 9 || MODE == 12) off = lane * 12; else if (MODE == 10 || MODE == 11) off = lane * 48;
    else off = (lane & 15) * 16 + (lane >> 4) * 6144;
    uint32_t acc = 0;
    uint4 v4 = make_uint4(lane, 1, 2, 3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
        uint8_t* p = base + off + (size_t)((it * 8 + wave) & 15) * 2048 % (win - 70000);
        if (MODE == 0 || MODE == 11 || MODE == 14) { uint4 v = *(const uint4*)p; acc += v.x ^ v.w; }
        else if (MODE == 1 || MODE == 2 || MODE == 3) { uint2 v = *(const uint2*)p; acc += v.x ^ v.y; }
        else if (MODE == 4) { uint2 v; __builtin_memcpy(&v, p, 8); acc += v.x ^ v.y; }
        else if (MODE == 5) { acc += *(const uint32_t*)p; }
        else if (MODE == 12) { U3 v = *(const U3*)p; acc += v.x ^ v.z; }
        else if (MODE == 6 || MODE == 10) { *(uint4*)p = v4; }
        else if (MODE == 7 || MODE == 8) { *(uint2*)p = make_uint2(v4.x, it); }
        else if (MODE == 9) { U3 v = {v4.x, (uint32_t)it, 3}; *(U3*)p = v; }
        else if (MODE == 13) { *(uint32_t*)p = it; }
    }
    if (MODE <= 5 || MODE == 11 || MODE == 12 || MODE == 14) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345678) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(uint8_t*, unsigned long long*, uint32_t*, size_t);
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t win = 256 * 1024;  // per-CU window, stays in L2
    uint8_t* buf; unsigned long long* cyc; uint32_t* sink;
    CHECK(hipMalloc(&buf, win * cus)); CHECK(hipMemset(buf, 1, win * cus)); CHECK(hipMalloc(&cyc, 8 * cus)); CHECK(hipMalloc(&sink, 4));
    struct { const char* name; kern_t f; int bytes; } ks[] = {
        {"load  x4 contiguous (1 KiB/instr)", k<0>, 1024}, {"load  x2 contiguous (512 B)", k<1>, 512}, {"load  x1 contiguous (256 B)", k<5>, 256},
        {"load  x3 contiguous (768 B)", k<12>, 768}, {"load  x2 24-B lane stride", k<2>, 512}, {"load  x4 48-B lane stride", k<11>, 1024},
        {"load  x2 unaligned 23-B stride", k<4>, 512}, {"load  x2 scattered 1000-B stride", k<3>, 512}, {"load  x4 tile 16 lanes x 4 rows", k<14>, 1024},
        {"store x4 contiguous", k<6>, 1024}, {"store x2 contiguous", k<7>, 512}, {"store x1 contiguous", k<13>, 256}, {"store x3 contiguous (12 B/lane)", k<9>, 768},
        {"store x2 24-B lane stride", k<8>, 512}, {"store x4 48-B lane stride", k<10>, 1024},
    };
    std::vector<unsigned long long> h(cus);
    for (auto& e : ks) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(e.f, dim3(cus), dim3(512), 0, 0, buf, cyc, sink, win);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), cyc, 8 * cus, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        double per = (double)h[cus / 2] / (ITER * 8);  // 8 waves per CU each issue ITER instructions
        printf("%-36s %7.1f ticks per wave-instruction per CU   (%.1f B/tick/CU)\n", e.name, per, e.bytes / per);
    }
    return 0;
}
