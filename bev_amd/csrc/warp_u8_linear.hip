// warp_u8_linear.hip -- the 8-bit bilinear instances of warp_rows (warp_rows.h); one translation unit per pixel type and interpolation so that
// the formats compile side by side.
#include "warp_rows.h"

namespace bevwarp {

void launch_u8_linear(const WarpArgs& a, int channels, dim3 grid, hipStream_t stream) { launch_channels<uint8_t, kLinear>(a, channels, grid, stream); }
#ifdef BEVWARP_CLOCK
hipError_t launch_u8_linear_clock(unsigned long long* out4, int reset) { return read_clock_of_this_unit(out4, reset); }
#endif

}  // namespace bevwarp
