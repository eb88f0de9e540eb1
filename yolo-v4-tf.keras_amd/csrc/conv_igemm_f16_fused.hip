// conv_igemm_f16_fused.hip -- instantiates conv_igemm_kernel's fused launches for Y4_F16 (split per dtype so the library builds in
// parallel; the kernel itself is conv_igemm_kernel.h).
#include "conv_igemm_kernel.h"

namespace y4 {
int conv_launch_f16_fused(int tile, const ConvK& k, hipStream_t s) { return launch_fused<Y4_F16>(tile, k, s); }
}  // namespace y4
