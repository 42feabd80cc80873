// ubench_isa.hip -- issue cost of the instructions the warp's coordinate chain is made of, on gfx950.
// Each kernel runs N independent copies of one instruction per loop iteration; 2 waves per SIMD on every CU.
// Prints cycles per wave-instruction per SIMD (shader clock from s_memtime).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdint>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 512;
constexpr int UNROLL = 8;

#define KERNEL_F64_1(name, ASM)                                                                         \
__global__ __launch_bounds__(256) void name(double* out, double seed, unsigned long long* cyc) {           \
    double a[UNROLL];                                                                                   \
    for (int i = 0; i < UNROLL; i++) a[i] = seed + threadIdx.x * 1e-3 + i;                              \
    double b = seed * 0.999 + 1.0, c = seed * 1.0001 + 2.0;                                            \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; it++) {                                                                 \
        _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                                   \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    double s = 0; for (int i = 0; i < UNROLL; i++) s += a[i];                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                     \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                    \
}

KERNEL_F64_1(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_F64_1(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_F64_1(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_F64_1(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_F64_1(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL_F64_1(k_min_f64, "v_min_f64 %0, %0, %1")
KERNEL_F64_1(k_max_f64, "v_max_f64 %0, %0, %1")
KERNEL_F64_1(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
KERNEL_F64_1(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %2")
KERNEL_F64_1(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
KERNEL_F64_1(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL_F64_1(k_sqrt_f64, "v_sqrt_f64 %0, %0")

__global__ __launch_bounds__(256) void k_cvt_i32_f64(double* out, double seed, unsigned long long* cyc) {
    double a[UNROLL]; int r[UNROLL];
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + threadIdx.x * 1e-3 + i; r[i] = 0; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < UNROLL; i++) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_cvt_f64_i32(double* out, double seed, unsigned long long* cyc) {
    double a[UNROLL]; int r[UNROLL];
    for (int i = 0; i < UNROLL; i++) { a[i] = 0; r[i] = (int)seed + threadIdx.x + i; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(r[i]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < UNROLL; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// full IEEE division as hipcc emits it
__global__ __launch_bounds__(256) void k_div_ieee(double* out, double seed, unsigned long long* cyc) {
    double a[UNROLL];
    for (int i = 0; i < UNROLL; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    double b = seed * 0.999 + 1.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) { a[i] = b / a[i]; asm volatile("" : "+v"(a[i])); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < UNROLL; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define KERNEL_I32(name, ASM)                                                                           \
__global__ __launch_bounds__(256) void name(double* out, double seed, unsigned long long* cyc) {           \
    unsigned a[UNROLL];                                                                                 \
    for (int i = 0; i < UNROLL; i++) a[i] = (unsigned)seed + threadIdx.x + i;                           \
    unsigned b = (unsigned)seed * 3 + 1, c = (unsigned)seed + 7;                                        \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; it++) {                                                                 \
        _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                                   \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    unsigned s = 0; for (int i = 0; i < UNROLL; i++) s += a[i];                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                     \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                    \
}
KERNEL_I32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_I32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL_I32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_I32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_I32(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL_I32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL_I32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_I32(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL_I32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_I32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_I32(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %0")
KERNEL_I32(k_dot2_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
KERNEL_I32(k_dot4_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL_I32(k_mul_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL_I32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL_I32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")

__global__ __launch_bounds__(256) void k_lds_read2(double* out, double seed, unsigned long long* cyc) {
    __shared__ unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    unsigned addr = ((threadIdx.x * 29) & 8191) * 4 & ~7u;
    unsigned acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            uint2 v;
            asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((addr + 64 * i) & 32767u));
            acc += v.x ^ v.y;
        }
        addr = (addr + 4 * (acc & 1)) & 32760u;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// accuracy of v_rcp_f64 and of one / two Newton steps on top of it (max relative error in ulps of 2^-52)
__global__ void k_rcp_accuracy(const double* x, double* e0, double* e1, double* e2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double w = x[i], r0;
    asm volatile("v_rcp_f64 %0, %1" : "=v"(r0) : "v"(w));
    double r1 = fma(fma(-w, r0, 1.0), r0, r0);
    double r2 = fma(fma(-w, r1, 1.0), r1, r1);
    double ex = 1.0 / w;
    e0[i] = fabs(r0 - ex) / ex; e1[i] = fabs(r1 - ex) / ex; e2[i] = fabs(r2 - ex) / ex;
}

typedef void (*kern_t)(double*, double, unsigned long long*);

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz  LDS/block %zu\n", prop.gcnArchName, cus, prop.clockRate, prop.sharedMemPerBlock);
    int blocks = cus * 2;  // 2 blocks of 4 waves per CU -> 2 waves per SIMD
    double* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 256)); CHECK(hipMalloc(&cyc, 8 * blocks));
    std::vector<unsigned long long> h(blocks);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_rcp_f64", k_rcp_f64},
        {"v_rndne_f64", k_rndne_f64}, {"v_min_f64", k_min_f64}, {"v_max_f64", k_max_f64}, {"v_cmp_lt_f64", k_cmp_f64},
        {"v_div_scale_f64", k_div_scale_f64}, {"v_div_fmas_f64", k_div_fmas_f64}, {"v_div_fixup_f64", k_div_fixup_f64},
        {"v_sqrt_f64", k_sqrt_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"ieee b/a (f64)", k_div_ieee},
        {"v_add_u32", k_add_u32}, {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo_u32},
        {"v_alignbit_b32", k_alignbit}, {"v_perm_b32", k_perm}, {"v_and_or_b32", k_and_or}, {"v_bfe_u32", k_bfe},
        {"v_fma_f32", k_fma_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_cvt_f32_ubyte1", k_cvt_ubyte}, {"v_dot2_u32_u16", k_dot2_u16},
        {"v_dot4_u32_u8", k_dot4_u8}, {"v_mul_u32_u24_sdwa", k_mul_sdwa}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16},
        {"v_pk_mad_u16", k_pk_mad_u16}, {"ds_read2_b32+wait", k_lds_read2},
    };
    for (auto& k : ks) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1.25, cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        double med = (double)h[blocks / 2];
        // s_memtime counts at 100 MHz on gfx9? report both raw ticks and per-instr
        double per = med / (ITER * UNROLL);
        printf("%-22s ticks/instr/wave %.3f   (x2 waves per SIMD -> %.3f ticks per wave-instr per SIMD)\n", k.name, per, per / 2);
    }
    // clock calibration: s_memtime vs wall clock
    {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_fma_f64, dim3(blocks), dim3(256), 0, 0, out, 1.25, cyc);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        CHECK(hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("fma_f64 kernel: %.3f us wall (incl launch), median %llu ticks -> >= %.1f MHz tick rate\n", ms * 1e3, h[blocks / 2], h[blocks / 2] / (ms * 1e3));
    }
    // rcp accuracy
    {
        int n = 1 << 22; std::vector<double> x(n);
        uint64_t s = 88172645463325252ull;
        for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = 0.5 + (double)(s >> 11) * (1.0 / 9007199254740992.0) * 1.5; if (i & 1) x[i] *= 1e3; if (i & 2) x[i] = -x[i]; }
        double *dx, *e0, *e1, *e2; CHECK(hipMalloc(&dx, 8 * n)); CHECK(hipMalloc(&e0, 8 * n)); CHECK(hipMalloc(&e1, 8 * n)); CHECK(hipMalloc(&e2, 8 * n));
        CHECK(hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rcp_accuracy, dim3(n / 256), dim3(256), 0, 0, dx, e0, e1, e2, n);
        std::vector<double> h0(n), h1(n), h2(n);
        CHECK(hipMemcpy(h0.data(), e0, 8 * n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1.data(), e1, 8 * n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), e2, 8 * n, hipMemcpyDeviceToHost));
        double m0 = 0, m1 = 0, m2 = 0; for (int i = 0; i < n; i++) { m0 = std::max(m0, fabs(h0[i])); m1 = std::max(m1, fabs(h1[i])); m2 = std::max(m2, fabs(h2[i])); }
        printf("v_rcp_f64 max rel err: raw %.3e (2^%.1f)  +1 Newton %.3e (2^%.1f)  +2 Newton %.3e (2^%.1f)\n", m0, log2(m0), m1, m1 > 0 ? log2(m1) : -99.0, m2, m2 > 0 ? log2(m2) : -99.0);
    }
    return 0;
}
