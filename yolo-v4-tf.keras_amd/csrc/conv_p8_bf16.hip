// conv_p8_bf16.hip -- instantiates the phased conv kernels (conv_p8_kernel.h, conv_l12_kernel.h) for Y4_BF16.
#include "conv_l12_kernel.h"

namespace y4 {
int conv_p8_launch_bf16(int bm, int nst, const ConvK& k, hipStream_t s) {
    if (nst == 10 && bm == 192) return launch_l12<Y4_BF16>(k, s);
    return launch_p8<Y4_BF16>(bm, nst, k, s);
}
}  // namespace y4
