// conv_igemm_f32.hip -- instantiates conv_igemm_kernel's plain launches for Y4_F32 (split per dtype so the library builds in
// parallel; the kernel itself is conv_igemm_kernel.h).
#include "conv_igemm_kernel.h"

namespace y4 {
int conv_launch_f32(int tile, const ConvK& k, hipStream_t s) { return launch_plain<Y4_F32>(tile, k, s); }
}  // namespace y4
