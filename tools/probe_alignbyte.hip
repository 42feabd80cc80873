// probe_alignbyte.hip -- which bits of S2 does v_alignbyte_b32 use on gfx950?  (ISA manuals disagree: [4:0] vs [1:0].)
//   hipcc -O3 --offload-arch=gfx950 tools/probe_alignbyte.hip -o tools/probe_alignbyte && tools/probe_alignbyte
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned hi = 0x77665544u, lo = 0x33221100u, s = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_alignbyte(hi, lo, s);
}
int main() {
    unsigned* d; unsigned h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    bool low2 = true;
    for (int s = 0; s < 64; s++) {
        const unsigned long long v = 0x7766554433221100ull;
        const unsigned exp2 = (unsigned)(v >> (8 * (s & 3)));
        if (h[s] != exp2) low2 = false;
        if (s < 9 || s == 35) printf("s=%2d -> %08x (uses [1:0] would give %08x)\n", s, h[s], exp2);
    }
    printf("v_alignbyte_b32 uses S2[1:0] only: %s\n", low2 ? "YES" : "NO");
    return 0;
}
