// conv_halo2_bf16.hip -- instantiates conv_halo2_kernel (conv_halo2_kernel.h) for Y4_BF16 (one translation unit per dtype: the library
// builds in parallel).
#include "conv_halo2_kernel.h"

namespace y4 {
int conv_halo2_launch_bf16(int tile, const ConvK& k, hipStream_t s) { return launch_halo2<Y4_BF16>(tile, k, s); }
}  // namespace y4

#ifdef H2_TRACE
extern "C" int y4_h2_trace_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::h2_trace_buf), sizeof(unsigned long long) * 16 * 4 * 16);
}
extern "C" int y4_h2_trace_blocks(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::h2_trace_blocks), sizeof(unsigned long long) * 4096 * 4);
}
#endif
