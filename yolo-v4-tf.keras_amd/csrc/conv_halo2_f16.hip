// conv_halo2_f16.hip -- instantiates conv_halo2_kernel (conv_halo2_kernel.h) for Y4_F16.
#include "conv_halo2_kernel.h"

namespace y4 {
int conv_halo2_launch_f16(int tile, const ConvK& k, hipStream_t s) { return launch_halo2<Y4_F16>(tile, k, s); }
}  // namespace y4
