// ubench_stream.hip -- cost of a vector-memory LOAD instruction when its data comes from HBM (no reuse), by shape.
// 4 workgroups x 4 waves per CU (the warp kernel's occupancy); every wave walks its own rows of a 2 GiB buffer and
// keeps one row of loads in flight while it consumes the previous one (the warp kernel's pipeline).  Reports
// device time per launch, load instructions per CU, cycles per load instruction per CU and the HBM read rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: 8 x dwordx2 per row, byte-unaligned, 3 B between lanes, 4 blocks of 192 B on each of 2 source rows (u8 bilinear gather)
// MODE 1: 4 x dword   per row, 3 B between lanes, 4 blocks on 1 source row (u8 nearest gather)
// MODE 2: 2 x dwordx4 per row, 16 B per lane contiguous (the same 768 B per source row, fetched coalesced; 256 B slack)
// MODE 3: 8 x (dwordx4 + dwordx2), 12 B between lanes (f32 bilinear gather, 128 px)
// MODE 4: 8 x dwordx4 per row contiguous (8 KiB per row)
// MODE 5: 8 x dwordx2 per row, 8 B per lane contiguous aligned (4 KiB per row)
// MODE 6: as MODE 0 with the address rounded down to 4 B and 12 B loaded (dwordx3): aligned windows that cover the 6 bytes
// MODE 7: as MODE 0 with the address rounded down to 4 B, 8 B loaded (dwordx2 aligned; not enough bytes, timing only)
// MODE 8: as MODE 0 with the address rounded down to 8 B and 16 B loaded (dwordx4... aligned to 8)
// MODE 9: as MODE 3 with the 24 bytes fetched as dwordx3 + dwordx3
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ buf, uint32_t* sink, int rows_per_wave, int64_t row_stride, int64_t wave_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint8_t* p = buf + gw * wave_stride;
    constexpr int NL = MODE == 1 ? 4 : MODE == 2 ? 2 : 8;
    uint4 cur[NL], nxt[NL];
    uint2 cur2[NL], nxt2[NL];
    uint32_t acc = 0;
    auto issue = [&](const uint8_t* r, uint4 (&v)[NL], uint2 (&w)[NL]) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            v[i] = make_uint4(0, 0, 0, 0);
            w[i] = make_uint2(0, 0);
            if (MODE == 0) {
                const uint8_t* q = r + (i >> 1) * 192 + (i & 1) * row_stride + lane * 3 + 1;
                __builtin_memcpy(&w[i], q, 8);
            } else if (MODE == 1) {
                const uint8_t* q = r + i * 192 + lane * 3 + 1;
                __builtin_memcpy(&v[i].x, q, 4);
            } else if (MODE == 2) {
                const uint8_t* q = r + i * row_stride + lane * 16;
                v[i] = *reinterpret_cast<const uint4*>(q);
            } else if (MODE == 3) {
                const uint8_t* q = r + (i >> 1) * 768 + (i & 1) * row_stride + lane * 12 + 4;
                __builtin_memcpy(&v[i], q, 16);
                __builtin_memcpy(&w[i], q + 16, 8);
            } else if (MODE == 6) {
                const uint8_t* q = r + (i >> 1) * 192 + (i & 1) * row_stride + ((lane * 3 + 1) & ~3);
                __builtin_memcpy(&v[i], q, 12);
            } else if (MODE == 7) {
                const uint8_t* q = r + (i >> 1) * 192 + (i & 1) * row_stride + ((lane * 3 + 1) & ~3);
                w[i] = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(q, 4));
            } else if (MODE == 8) {
                const uint8_t* q = r + (i >> 1) * 192 + (i & 1) * row_stride + ((lane * 3 + 1) & ~7);
                __builtin_memcpy(&v[i], q, 16);
            } else if (MODE == 9) {
                const uint8_t* q = r + (i >> 1) * 768 + (i & 1) * row_stride + lane * 12 + 4;
                __builtin_memcpy(&v[i], q, 12);
                uint32_t t3[3];
                __builtin_memcpy(t3, q + 12, 12);
                w[i] = make_uint2(t3[0] ^ t3[2], t3[1]);
            } else if (MODE == 4) {
                v[i] = *reinterpret_cast<const uint4*>(r + i * 1024 + lane * 16);
            } else {
                w[i] = *reinterpret_cast<const uint2*>(r + i * 512 + lane * 8);
            }
        }
    };
    issue(p, cur, cur2);
    for (int y = 1; y <= rows_per_wave; y++) {
        if (y < rows_per_wave) issue(p + (int64_t)y * 2 * row_stride, nxt, nxt2);
#pragma unroll
        for (int i = 0; i < NL; i++) acc += cur[i].x ^ cur[i].w ^ cur2[i].x ^ cur2[i].y;
#pragma unroll
        for (int i = 0; i < NL; i++) cur[i] = nxt[i], cur2[i] = nxt2[i];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
int run(const char* name, const uint8_t* buf, uint32_t* sink, int cus, double useful_bytes_per_row, int instr_per_row) {
    const int rows = 64;
    const int64_t row_stride = 8192, wave_stride = (int64_t)rows * 2 * row_stride + 4096;  // 2 MiB per wave, nothing shared
    const int blocks = cus * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, sink, rows, row_stride, wave_stride);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double us = best * 1e3, instr_per_cu = 16.0 * rows * instr_per_row;
    printf("%-58s %8.1f us  %6.0f load instr/CU  %6.1f cycles/instr/CU @2.4GHz  useful %6.2f TB/s\n", name, us, instr_per_cu,
           us * 2400.0 / instr_per_cu, useful_bytes_per_row * rows * 16.0 * cus / (us * 1e6));
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t)cus * 16 * (64 * 2 * 8192 + 4096) + (1 << 20);
    uint8_t* buf; uint32_t* sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMalloc(&sink, 4));
    printf("%d CUs, %.2f GiB walked once per launch (no reuse), 16 waves per CU, one row of loads in flight per wave\n", cus, bytes / 1073741824.0);
    run<0>("8 x dwordx2 unaligned, 3 B lane stride (u8 bilinear gather)", buf, sink, cus, 2 * 776.0, 8);
    run<1>("4 x dword unaligned, 3 B lane stride (u8 nearest gather)", buf, sink, cus, 772.0, 4);
    run<2>("2 x dwordx4 coalesced (same bytes as the u8 bilinear row)", buf, sink, cus, 2048.0, 2);
    run<3>("8 x (dwordx4+dwordx2), 12 B lane stride (f32 bilinear gather)", buf, sink, cus, 2 * 3096.0, 16);
    run<4>("8 x dwordx4 coalesced (8 KiB per row)", buf, sink, cus, 8192.0, 8);
    run<5>("8 x dwordx2 coalesced aligned (4 KiB per row)", buf, sink, cus, 4096.0, 8);
    run<9>("8 x (dwordx3+dwordx3), 12 B lane stride (f32 bilinear gather)", buf, sink, cus, 2 * 3096.0, 16);
    run<6>("8 x dwordx3 4-B aligned windows, ~3 B lane stride", buf, sink, cus, 2 * 776.0, 8);
    run<7>("8 x dwordx2 4-B aligned, ~3 B lane stride (timing only)", buf, sink, cus, 2 * 776.0, 8);
    run<8>("8 x dwordx4 8-B aligned windows, ~3 B lane stride", buf, sink, cus, 2 * 776.0, 8);
    return 0;
}
