// conv_halo_bf16.hip -- instantiates conv_halo_kernel (conv_halo_kernel.h) for Y4_BF16 (one translation unit per dtype: the library
// builds in parallel).
#include "conv_halo_kernel.h"

namespace y4 {
int conv_halo_launch_bf16(int bm, int bn, const ConvK& k, hipStream_t s) { return launch_halo<Y4_BF16>(bm, bn, k, s); }
}  // namespace y4
