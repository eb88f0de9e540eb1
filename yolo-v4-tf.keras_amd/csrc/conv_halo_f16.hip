// conv_halo_f16.hip -- instantiates conv_halo_kernel (conv_halo_kernel.h) for Y4_F16.
#include "conv_halo_kernel.h"

namespace y4 {
int conv_halo_launch_f16(int bm, int bn, const ConvK& k, hipStream_t s) { return launch_halo<Y4_F16>(bm, bn, k, s); }
}  // namespace y4
