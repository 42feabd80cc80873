// proto_lean.hip -- PROTOTYPE (measurement aid, not product code; libbevwarp.so does not contain it).
//
// Question: the 8-bit bilinear kernel's vector ALU is busy 0.6 of the time and every wave of it waits most of its
// life; it is compiled for four waves per SIMD because ONE kernel carries every tile class (edge blocks, patches, the exact chain) and its
// interior loop holds four pixels per lane and two tap sets.  What does the interior row-affine pipeline run at when it is ALL the kernel
// holds -- fewer pixels per lane, more waves per SIMD?
//
// warp_lean<PPL, ROWS, AHEAD, WPE>: 8-bit RGB bilinear, row-affine interior tiles ONLY (every tile is treated as one: the caller pads the
// source batch with a frame on either side and compares against the product on a footprint that lies inside the frame), 4-byte aligned
// frames and row strides.  Workgroup = 4 waves = a tile of 64 PPL x 4 ROWS pixels, rows dealt round-robin; per pass one reciprocal and
// one Y per wave, X = one add + one FMA per pixel, aligned 12-byte tap windows, the product's exact integer blend (sample.h), one LDS
// transposition row per pass, 12-byte stores after the last pass; straight-line code with AHEAD tap sets in flight.  Pixels in a tie
// window (coords.h) are NOT redone: they are counted (`ties`), so that the comparison can say how many pixels may differ.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -shared -fPIC -Ibev_amd/csrc -Iinclude tools/proto_lean.hip -o tools/libproto_lean.so
#include "sample.h"

using namespace bevwarp;

struct LeanArgs {
    const uint8_t* src;
    uint8_t* dst;
    const double* minv;
    int64_t src_fs, src_rs, dst_fs, dst_rs;
    int batch, src_h, src_w, dst_h, dst_w;
    int tiles_x, tiles_per_frame, chunk;
    unsigned int total;
    unsigned long long* ties;
};

template <int PPL, int ROWS, int AHEAD, int WPE, int MODE = 0, bool PAIR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, (MODE & 32) ? WPE : 8))) void warp_lean(const LeanArgs a) {
    constexpr int TW = 64 * PPL, TH = 4 * ROWS;
    using F = Fix<kLinear>;
    __shared__ __attribute__((aligned(16))) uint32_t s_tr[4][ROWS][TW];
    const uint32_t item = (blockIdx.x & 7u) * (uint32_t)a.chunk + (blockIdx.x >> 3);
    if (item >= a.total) return;
    const uint32_t frame_idx = item / (uint32_t)a.tiles_per_frame, t = item - frame_idx * (uint32_t)a.tiles_per_frame;
    const uint32_t ty = t / (uint32_t)a.tiles_x, tx = t - ty * (uint32_t)a.tiles_x;
    const int x0 = (int)tx * TW, y0 = (int)ty * TH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t* __restrict__ frame = a.src + (int64_t)frame_idx * a.src_fs;
    uint8_t* __restrict__ dframe = a.dst + (int64_t)frame_idx * a.dst_fs;
    const double* __restrict__ M = a.minv + (int64_t)frame_idx * 9;
    auto uniform_f64 = [](double v) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    };
    const double m0 = M[0], m1 = M[1], m2 = M[2], m4 = M[4], m5 = M[5], m7 = M[7], m8 = M[8];  // (row-affine: M3 = M6 = 0)
    const double x0d = (double)x0;
    const double CX = uniform_f64((m0 * x0d + m2) * kTwo32), CY = uniform_f64(m5 * kTwo32), CW = uniform_f64(m8);
    const double RX = uniform_f64(m1 * kTwo32), RY = uniform_f64(m4 * kTwo32), RW = uniform_f64(m7);
    const double DX = uniform_f64(m0 * (64.0 * kTwo32));
    // PAIR: a lane owns PAIRS of adjacent pixels -- pixels 128 p + 2 l and + 1 of the segment -- and fetches both pixels' taps of a row with ONE
    // aligned 16-byte load (legal while the two left taps are at most 2 source pixels apart: 12 bytes + 3 of alignment): half the gather
    // instructions, each spanning twice the bytes.
    const double cx0 = (m0 * kTwo32) * (double)(PAIR ? 2 * lane : lane);
    [[maybe_unused]] const double DX1 = uniform_f64(m0 * kTwo32);
    const uint32_t rs32 = (uint32_t)a.src_rs;
    const uint32_t kOff = 0u - 0x380000u * (rs32 + 3u);
    const uint8_t* b0 = frame;
    const uint8_t* b1 = frame + rs32;
    const double yf = (double)(y0 + wave);
    double UX = __builtin_fma(RX, yf, CX), UY = __builtin_fma(RY, yf, CY), UW = __builtin_fma(RW, yf, CW);
    const double SX = uniform_f64(RX * 4.0), SY = uniform_f64(RY * 4.0), SW = uniform_f64(RW * 4.0);

    uint32_t R0[AHEAD][PPL], R1[AHEAD][PPL], RYl[AHEAD];
    Bytes<PAIR ? 16 : 12> r0[AHEAD][PAIR ? PPL / 2 : PPL], r1[AHEAD][PAIR ? PPL / 2 : PPL];
    uint32_t tie_acc = 0xffffffffu;
    auto coords = [&](int d) __attribute__((always_inline)) {
        const double r = rcp_newton(UW);
        const double ty_ = __builtin_fma(UY, r, F::kMagic);
        const uint32_t hyu = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(ty_)), lyu = (uint32_t)__builtin_amdgcn_readfirstlane(__double2loint(ty_));
        const uint32_t row_off = (hyu & 0xffffffu) * rs32 + kOff;
        double Xn = UX + cx0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const double tx_ = __builtin_fma(Xn, r, F::kMagic);
            R0[d][j] = __umul24((uint32_t)__double2hiint(tx_), 3u) + row_off;
            R1[d][j] = (uint32_t)__double2loint(tx_);
            tie_acc = min(tie_acc, R1[d][j] & F::kTieMask);
            if (j + 1 < PPL) Xn += PAIR ? ((j & 1) ? DX + DX - DX1 : DX1) : DX;
        }
        RYl[d] = lyu;
        tie_acc = min(tie_acc, lyu & F::kTieMask);
        UX += SX;
        UY += SY;
        UW += SW;
    };
    auto issue = [&](int d) __attribute__((always_inline)) {
        if constexpr (PAIR) {
#pragma unroll
            for (int p = 0; p < PPL / 2; p++) {
                const uint32_t offa = R0[d][2 * p] & ~3u;
                __builtin_memcpy(&r0[d][p], b0 + offa, 16);
                __builtin_memcpy(&r1[d][p], b1 + offa, 16);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            uint32_t offa = R0[d][j] & ~3u;
            if constexpr (MODE & 8) offa &= 0xfffcu;  // ablation: every tap from the frame's first 64 KB (cache hits)
            if constexpr (MODE & 2) {                 // ablation: no tap loads
#pragma unroll
                for (int q = 0; q < 3; q++) r0[d][j].w[q] = offa * (q + 3), r1[d][j].w[q] = offa ^ (0x9e3779b9u * (q + 1));
            } else {
                __builtin_memcpy(&r0[d][j], b0 + offa, 12);
                __builtin_memcpy(&r1[d][j], b1 + offa, 12);
            }
        }
    };
    auto put = [&](int k, int idx, uint32_t a0, uint32_t a1, uint32_t c0, uint32_t c1, uint32_t fx, uint32_t fy) __attribute__((always_inline)) {
        if constexpr (MODE & 4)  // ablation: no blend
            s_tr[wave][k][idx] = (a0 ^ a1 ^ c0 ^ c1) + fx + fy;
        else
            s_tr[wave][k][idx] = blend_u8_rgb_window(a0, a1, c0, c1, fx, fy);
    };
    auto finish = [&](int d, int k) __attribute__((always_inline)) {
        const uint32_t fy = RYl[d] >> 27;
        if constexpr (PAIR) {
#pragma unroll
            for (int p = 0; p < PPL / 2; p++) {
                // window = 16 bytes from the aligned address below the first pixel's left tap; the first pixel's taps start at byte sh & 3, the
                // second's at b = (sh & 3) + 3 (ix1 - ix0) <= 9: dword b >> 2 of the window, same byte phase modulo 4 as `b` itself
                const uint32_t sh = R0[d][2 * p], b = R0[d][2 * p + 1] - (sh & ~3u);
                const uint32_t (&w)[4] = r0[d][p].w;
                const uint32_t (&v)[4] = r1[d][p].w;
                const uint32_t A0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh), A1 = __builtin_amdgcn_alignbyte(w[2], w[1], sh);
                const uint32_t C0 = __builtin_amdgcn_alignbyte(v[1], v[0], sh), C1 = __builtin_amdgcn_alignbyte(v[2], v[1], sh);
                put(k, 128 * p + 2 * lane, A0, A1, C0, C1, R1[d][2 * p] >> 27, fy);
                const uint32_t B0 = __builtin_amdgcn_alignbyte(w[1], w[0], b), B1 = __builtin_amdgcn_alignbyte(w[2], w[1], b), B2 = __builtin_amdgcn_alignbyte(w[3], w[2], b),
                               B3 = __builtin_amdgcn_alignbyte(w[3], w[3], b);
                const uint32_t D0 = __builtin_amdgcn_alignbyte(v[1], v[0], b), D1 = __builtin_amdgcn_alignbyte(v[2], v[1], b), D2 = __builtin_amdgcn_alignbyte(v[3], v[2], b),
                               D3 = __builtin_amdgcn_alignbyte(v[3], v[3], b);
                const bool s1 = b >= 4u, s2 = b >= 8u;
                const uint32_t a0 = s2 ? B2 : (s1 ? B1 : B0), a1 = s2 ? B3 : (s1 ? B2 : B1);
                const uint32_t c0 = s2 ? D2 : (s1 ? D1 : D0), c1 = s2 ? D3 : (s1 ? D2 : D1);
                put(k, 128 * p + 2 * lane + 1, a0, a1, c0, c1, R1[d][2 * p + 1] >> 27, fy);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint32_t fx = R1[d][j] >> 27, sh = R0[d][j];
            const uint32_t a0 = __builtin_amdgcn_alignbyte(r0[d][j].w[1], r0[d][j].w[0], sh), a1 = __builtin_amdgcn_alignbyte(r0[d][j].w[2], r0[d][j].w[1], sh);
            const uint32_t c0 = __builtin_amdgcn_alignbyte(r1[d][j].w[1], r1[d][j].w[0], sh), c1 = __builtin_amdgcn_alignbyte(r1[d][j].w[2], r1[d][j].w[1], sh);
            put(k, 64 * j + lane, a0, a1, c0, c1, fx, fy);
        }
    };
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {
        coords(k);
        issue(k);
    }
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int d = k % AHEAD;
        finish(d, k);
        if (k + AHEAD < ROWS) {
            coords(d);
            issue(d);
        }
    }
    if (__ballot(tie_acc == 0) != 0ull && lane == 0) atomicAdd(a.ties, 1ull);  // (passes of this wave with a pixel in a tie window: >= 1)
    asm volatile("" ::: "memory");
    // -- LDS rows -> memory: a lane stores 4 consecutive pixels (12 bytes); 64 / (TW / 4) rows per instruction
    constexpr int LPR = TW / 4, RPS = 64 / LPR;
    static_assert(ROWS % RPS == 0, "rows per store instruction");
#pragma unroll
    for (int kk = 0; kk < ROWS; kk += RPS) {
        const int k = kk + lane / LPR, u = lane % LPR;
        const uint4 o = *reinterpret_cast<const uint4*>(&s_tr[wave][k][4 * u]);
        const int y = y0 + wave + 4 * k, x = x0 + 4 * u;
        if constexpr (MODE & 1) {  // ablation: no stores
            asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
            if (y != 12345678) continue;
        }
        if constexpr (MODE & 16) {  // ablation: stores land in the first rows of the frame (cache-resident destination)
            const u32x3 v = {o.x | (o.y << 24), (o.y >> 8) | (o.z << 16), (o.z >> 16) | (o.w << 8)};
            wide_store(reinterpret_cast<u32x3*>(dframe + (int64_t)(y & 7) * a.dst_rs + (int64_t)x * 3), v);
            continue;
        }
        if (y < a.dst_h && x + 3 < a.dst_w) {
            const u32x3 v = {o.x | (o.y << 24), (o.y >> 8) | (o.z << 16), (o.z >> 16) | (o.w << 8)};
            wide_store(reinterpret_cast<u32x3*>(dframe + (int64_t)y * a.dst_rs + (int64_t)x * 3), v);
        }
    }
}

template <int PPL, int ROWS, int AHEAD, int WPE, int MODE = 0, bool PAIR = false>
static int launch(const LeanArgs& a0, hipStream_t st) {
    LeanArgs a = a0;
    constexpr int TW = 64 * PPL, TH = 4 * ROWS;
    a.tiles_x = (a.dst_w + TW - 1) / TW;
    a.tiles_per_frame = a.tiles_x * ((a.dst_h + TH - 1) / TH);
    const long total = (long)a.batch * a.tiles_per_frame;
    a.total = (unsigned)total;
    a.chunk = (int)((total + 7) / 8);
    (void)hipGetLastError();
    hipLaunchKernelGGL((warp_lean<PPL, ROWS, AHEAD, WPE, MODE, PAIR>), dim3(8 * a.chunk), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

// variant = PPL * 1000 + ROWS * 100 + AHEAD * 10 + WPE (+ 10000 * MODE: ablations of 2628 -- 1 no stores, 2 no tap loads, 4 no blend,
// (+ 1000000: PAIR -- one 16-byte load per pixel pair and row)
// 32 exactly WPE waves per SIMD (the register allocator may otherwise reach more);
// 8 taps from the first 64 KB of the frame, 16 stores into the first 8 rows of the frame; sums combine)
extern "C" int proto_lean(int variant, const void* src, void* dst, const double* minv, int batch, int src_h, int src_w, int dst_h, int dst_w, long src_fs,
                          long src_rs, long dst_fs, long dst_rs, unsigned long long* ties, void* stream) {
    if (((uintptr_t)src & 3) || (src_rs & 3) || (src_fs & 3) || ((uintptr_t)dst & 3) || (dst_rs & 3) || (dst_fs & 3) || (dst_w & 3)) return -1;
    LeanArgs a = {(const uint8_t*)src, (uint8_t*)dst, minv, src_fs, src_rs, dst_fs, dst_rs, batch, src_h, src_w, dst_h, dst_w, 0, 0, 0, 0u, ties};
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
#define V(P, R, A, W) \
    case P * 1000 + R * 100 + A * 10 + W: \
        return launch<P, R, A, W>(a, st);
        V(4, 6, 2, 4)  // the product's shape
        V(4, 6, 1, 5)
        V(4, 6, 2, 5)
        V(4, 6, 3, 4)
        V(4, 6, 3, 5)
        V(4, 4, 2, 5)
        V(2, 6, 2, 4)
        V(2, 6, 2, 6)
        V(2, 6, 2, 8)
        V(2, 6, 3, 6)
        V(2, 6, 1, 8)
        V(2, 8, 2, 6)
        V(2, 8, 2, 8)
        V(2, 4, 2, 8)
        V(1, 8, 2, 8)
        V(1, 8, 4, 8)
        V(1, 4, 2, 8)
        V(1, 4, 4, 8)
#undef V
#define P(PP, R, A_, W, MODE) \
    case 1000000 + PP * 1000 + R * 100 + A_ * 10 + W + 10000 * MODE: \
        return launch<PP, R, A_, W, MODE, true>(a, st);
        P(4, 6, 2, 4, 32) P(4, 6, 2, 3, 32) P(4, 6, 2, 5, 32) P(4, 6, 2, 4, 0) P(4, 6, 2, 5, 0) P(4, 6, 3, 4, 0) P(2, 6, 2, 8, 0) P(2, 6, 3, 8, 0) P(2, 6, 2, 6, 0) P(4, 6, 2, 5, 4) P(4, 6, 2, 5, 1) P(4, 6, 2, 5, 5) P(2, 6, 2, 8, 4) P(2, 6, 2, 8, 5)
#undef P
#define A(MODE) \
    case 2628 + 10000 * MODE: \
        return launch<2, 6, 2, 8, MODE>(a, st);
        A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(32) A(16) A(18) A(24) A(20) A(22) A(12)
#undef A
    }
    return -2;
}
