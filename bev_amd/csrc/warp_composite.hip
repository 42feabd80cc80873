// warp_composite.hip -- the composite instances of warp_rows (NSRC = 3: bev/tool/compo.py:26-49 in one launch; warp_rows.h).
#include "warp_rows.h"

namespace bevwarp {

hipError_t launch_warp_composite(const WarpArgs& a, int channels, hipStream_t stream) {
    (void)hipGetLastError();
    const dim3 grid((unsigned)(8 * a.chunk)), block(kWG * 3);
    switch (channels) {
        case 1: hipLaunchKernelGGL((warp_rows<uint8_t, 1, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((warp_rows<uint8_t, 2, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL((warp_rows<uint8_t, 3, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((warp_rows<uint8_t, 4, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
    }
    return hipGetLastError();
}
int composite_max_rows() { return kCompositeRows; }
#ifdef BEVWARP_CLOCK
hipError_t launch_composite_clock(unsigned long long* out4, int reset) { return read_clock_of_this_unit(out4, reset); }
#endif

}  // namespace bevwarp
