// conv_igemm_bf16_fused.hip -- instantiates conv_igemm_kernel's fused launches for Y4_BF16 (split per dtype so the library builds in
// parallel; the kernel itself is conv_igemm_kernel.h).
#include "conv_igemm_kernel.h"

namespace y4 {
int conv_launch_bf16_fused(int tile, const ConvK& k, hipStream_t s) { return launch_fused<Y4_BF16>(tile, k, s); }
}  // namespace y4

#ifdef Y4_TRACE
// experiments only (scripts/pair_trace.py): the life trace of an LDS-pair head (Y4_TRACE_PAIR=1)
extern "C" int y4_trace_read_life_fused(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::y4_trace_life), sizeof(unsigned long long) * 8 * 8);
}
#endif
