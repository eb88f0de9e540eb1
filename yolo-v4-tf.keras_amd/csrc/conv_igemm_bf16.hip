// conv_igemm_bf16.hip -- instantiates conv_igemm_kernel's plain launches for Y4_BF16 (split per dtype so the library builds in
// parallel; the kernel itself is conv_igemm_kernel.h).
#include "conv_igemm_kernel.h"

namespace y4 {
int conv_launch_bf16(int tile, const ConvK& k, hipStream_t s) { return launch_plain<Y4_BF16>(tile, k, s); }
}  // namespace y4

#ifdef Y4_TRACE
// experiments only (scripts/trace_read.py): copies the in-kernel phase trace out; not part of include/yolo4hip.h
extern "C" int y4_trace_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::y4_trace_buf), sizeof(unsigned long long) * 4 * 8 * 8);
}
extern "C" int y4_trace_read_life(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::y4_trace_life), sizeof(unsigned long long) * 8 * 8);
}
#endif
