// ubench_lat.hip -- dependent-issue latency of float64 / integer VALU ops on gfx950: ONE wave per SIMD, one
// dependency chain (each instruction consumes the previous result).  Ticks of s_memtime per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2048;
#define CHAIN_F64(name, ASM, NCH)                                                                      \
__global__ __launch_bounds__(256) void name(double* out, double seed, unsigned long long* cyc) {          \
    double a[NCH]; for (int i = 0; i < NCH; i++) a[i] = seed + threadIdx.x * 1e-3 + i;                  \
    double b = seed * 0.999 + 1.0, c = seed * 1.0001 + 2.0;                                            \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < NCH; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    double s = 0; for (int i = 0; i < NCH; i++) s += a[i];                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }
CHAIN_F64(add1, "v_add_f64 %0, %0, %1", 1)
CHAIN_F64(add2, "v_add_f64 %0, %0, %1", 2)
CHAIN_F64(add4, "v_add_f64 %0, %0, %1", 4)
CHAIN_F64(fma1, "v_fma_f64 %0, %0, %1, %2", 1)
CHAIN_F64(mul1, "v_mul_f64 %0, %0, %1", 1)
CHAIN_F64(rcp1, "v_rcp_f64 %0, %0", 1)
CHAIN_F64(rcp2, "v_rcp_f64 %0, %0", 2)
#define CHAIN_I32(name, ASM, NCH)                                                                      \
__global__ __launch_bounds__(256) void name(double* out, double seed, unsigned long long* cyc) {          \
    unsigned a[NCH]; for (int i = 0; i < NCH; i++) a[i] = (unsigned)seed + threadIdx.x + i;              \
    unsigned b = (unsigned)seed * 3 + 1, c = (unsigned)seed + 7;                                        \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < NCH; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    unsigned s = 0; for (int i = 0; i < NCH; i++) s += a[i];                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }
CHAIN_I32(iadd1, "v_add_u32 %0, %0, %1", 1)
CHAIN_I32(iadd4, "v_add_u32 %0, %0, %1", 4)
CHAIN_I32(perm1, "v_perm_b32 %0, %0, %1, %2", 1)
CHAIN_I32(perm4, "v_perm_b32 %0, %0, %1, %2", 4)
CHAIN_I32(fma32_1, "v_fma_f32 %0, %0, %1, %2", 1)
typedef void (*kern_t)(double*, double, unsigned long long*);
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount;  // one 4-wave block per CU -> one wave per SIMD
    double* out; unsigned long long* cyc; CHECK(hipMalloc(&out, 8 * blocks * 256)); CHECK(hipMalloc(&cyc, 8 * blocks));
    std::vector<unsigned long long> h(blocks);
    struct { const char* n; kern_t k; int nch; } ks[] = {{"v_add_f64 x1 chain", add1, 1}, {"v_add_f64 x2 chains", add2, 2}, {"v_add_f64 x4 chains", add4, 4},
        {"v_fma_f64 x1 chain", fma1, 1}, {"v_mul_f64 x1 chain", mul1, 1}, {"v_rcp_f64 x1 chain", rcp1, 1}, {"v_rcp_f64 x2 chains", rcp2, 2},
        {"v_add_u32 x1 chain", iadd1, 1}, {"v_add_u32 x4 chains", iadd4, 4}, {"v_perm_b32 x1 chain", perm1, 1}, {"v_perm_b32 x4 chains", perm4, 4}, {"v_fma_f32 x1 chain", fma32_1, 1}};
    for (auto& e : ks) {
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.25, cyc);
        CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost)); std::sort(h.begin(), h.end());
        printf("%-24s %.2f ticks per instruction (one wave per SIMD)\n", e.n, (double)h[blocks / 2] / (ITER * e.nch));
    }
    return 0;
}
