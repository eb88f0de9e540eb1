#!/bin/bash
# Host-side sanitizers: every translation unit's HOST pass (plan, workspace layout and aliasing, schedule
# validation, tuner bookkeeping, argument checks -- csrc/runtime.hip is 1 700 lines of it) built with AddressSanitizer +
# UndefinedBehaviorSanitizer (-Xarch_host: the device pass is compiled as always) into scratch/libyolo4hip_hostasan.so.
# (GPU ASan / XNACK are not available on this pool.)  Used by tests/test_host_sanitizers.py:
#     LD_PRELOAD=<libclang_rt.asan> YOLO4HIP_LIB=scratch/libyolo4hip_hostasan.so python -m pytest tests/test_lib_abi.py tests/test_plan.py
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/yolo-v4-tf.keras_amd/csrc
out=$here/scratch/hostasan; mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -g -w"
pids=""
# (four compiles at a time: the device passes are the expensive part)
for u in conv_igemm_bf16 conv_igemm_f16 conv_p8_bf16 conv_p8_f16 conv_halo_bf16 conv_halo_f16 conv_igemm_bf16_fused conv_igemm_f16_fused conv_igemm_f32 resblock csp_stage stem_down misc_kernels decode_nms runtime conv_igemm; do
  ( cd $src && hipcc $FLAGS -c $u.hip -o $out/$u.o ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
hipcc -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $here/scratch/libyolo4hip_hostasan.so $out/*.o
echo "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)" > $here/scratch/hostasan_runtime.txt
ls -la $here/scratch/libyolo4hip_hostasan.so
