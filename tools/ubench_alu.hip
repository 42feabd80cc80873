// ubench_alu.hip -- vector-ALU cost of the two arithmetic blocks of the 8-bit bilinear row (no memory): the float64
// fixed-point coordinate chain of 4 pixels per lane, and the 4-pixel RGB blend (funnel shift + dot4 / dot2).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Ibev_amd/csrc -Iinclude tools/ubench_alu.hip -o tools/ubench_alu
// Prints shader ticks per row segment (256 px) per wave for 1..5 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../bev_amd/csrc/warp_kernels.hip"

using namespace bevwarp;

template <int WHAT>  // 1 = chain, 2 = blend, 3 = both
__global__ __launch_bounds__(256) void alu(const double* __restrict__ Min, uint32_t* __restrict__ out, unsigned long long* ticks, int iters, uint32_t seed) {
    constexpr int PPL = 4;
    using F = Fix<kLinear>;
    const int lane = threadIdx.x & 63;
    const double m0 = Min[0], m1 = Min[1], m2 = Min[2], m3 = Min[3], m4 = Min[4], m5 = Min[5], m6 = Min[6], m7 = Min[7], m8 = Min[8];
    const double CX = m2 * kTwo32, CY = m5 * kTwo32, CW = m8, RX = m1 * kTwo32, RY = m4 * kTwo32, RW = m7;
    const double DX = m0 * 64.0 * kTwo32, DY = m3 * 64.0 * kTwo32, DW = m6 * 64.0;
    const double cx0 = m0 * kTwo32 * lane, cy0 = m3 * kTwo32 * lane, cw0 = m6 * lane;
    uint32_t acc = seed + threadIdx.x;
    uint32_t t[4][3], u[4][3];
    for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) { t[j][k] = seed * (7 * j + k + 1) + lane; u[j][k] = seed * (13 * j + k + 3) ^ lane; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint32_t hx[PPL], lx[PPL], hy[PPL], ly[PPL];
        if (WHAT & 1) {
            const double dy = (double)(it + (int)(acc & 1));
            const double UX = __builtin_fma(RX, dy, CX), UY = __builtin_fma(RY, dy, CY), UW = __builtin_fma(RW, dy, CW);
            double W[PPL], r[PPL];
            W[0] = UW + cw0;
            for (int j = 1; j < PPL; j++) W[j] = W[j - 1] + DW;
            const double p01 = W[0] * W[1], p23 = W[2] * W[3];
            const double inv = rcp_newton(p01 * p23);
            const double i01 = inv * p23, i23 = inv * p01;
            r[0] = i01 * W[1]; r[1] = i01 * W[0]; r[2] = i23 * W[3]; r[3] = i23 * W[2];
            uint32_t tie = 0xffffffffu;
            double Xn = UX + cx0, Yn = UY + cy0;
            for (int j = 0; j < PPL; j++) {
                const double tx_ = __builtin_fma(Xn, r[j], F::kMagic), ty_ = __builtin_fma(Yn, r[j], F::kMagic);
                hx[j] = (uint32_t)__double2hiint(tx_), lx[j] = (uint32_t)__double2loint(tx_);
                hy[j] = (uint32_t)__double2hiint(ty_), ly[j] = (uint32_t)__double2loint(ty_);
                tie = min(tie, min(lx[j] & F::kTieMask, ly[j] & F::kTieMask));
                if (j + 1 < PPL) { Xn += DX; Yn += DY; }
            }
            if (tie == 0) acc ^= 0x55u;
            for (int j = 0; j < PPL; j++) acc += __umul24(hy[j], 5760u) + (__umul24(hx[j], 3u) + 77u);
        } else {
            for (int j = 0; j < PPL; j++) { lx[j] = acc * (j + 3); ly[j] = acc * (j + 11); hx[j] = hy[j] = 0; }
        }
        if (WHAT & 2) {
            for (int j = 0; j < PPL; j++) {
                const uint32_t sh = (acc + j) << 3;
                const uint32_t a0 = __builtin_amdgcn_alignbit(t[j][1], t[j][0], sh), a1 = __builtin_amdgcn_alignbit(t[j][2], t[j][1], sh);
                const uint32_t b0 = __builtin_amdgcn_alignbit(u[j][1], u[j][0], sh), b1 = __builtin_amdgcn_alignbit(u[j][2], u[j][1], sh);
                const uint32_t px = blend_u8_rgb_window(a0, a1, b0, b1, lx[j] >> 27, ly[j] >> 27);
                acc += px;
                t[j][0] ^= px;  // keep the taps changing
            }
        } else {
            for (int j = 0; j < PPL; j++) acc ^= lx[j] + ly[j];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 2000;
    double hM[9] = {1.3, 0.01, 5.0, 0.02, 1.1, 7.0, 1e-5, 2e-5, 1.0};
    double* dM; uint32_t* out; unsigned long long* ticks;
    hipMalloc(&dM, 72); hipMemcpy(dM, hM, 72, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 5; wps++) {
        const int blocks = 256 * wps;  // 256-thread blocks: one wave per SIMD each; wps blocks per CU
        hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ticks, blocks * 8);
        std::vector<unsigned long long> ht(blocks);
        for (int what = 1; what <= 3; what++) {
            for (int rep = 0; rep < 2; rep++) {
                if (what == 1) hipLaunchKernelGGL(alu<1>, dim3(blocks), dim3(256), 0, 0, dM, out, ticks, iters, 12345u);
                if (what == 2) hipLaunchKernelGGL(alu<2>, dim3(blocks), dim3(256), 0, 0, dM, out, ticks, iters, 12345u);
                if (what == 3) hipLaunchKernelGGL(alu<3>, dim3(blocks), dim3(256), 0, 0, dM, out, ticks, iters, 12345u);
                hipDeviceSynchronize();
            }
            hipMemcpy(ht.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto t : ht) mean += (double)t; mean /= blocks;
            printf("%d wave(s) per SIMD  %-6s  %8.1f ticks per row per wave   -> %7.1f ticks of SIMD time per row\n", wps,
                   what == 1 ? "chain" : what == 2 ? "blend" : "both", mean / iters, mean / iters / wps);
        }
        hipFree(out); hipFree(ticks);
    }
    return 0;
}
