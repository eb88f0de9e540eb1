#!/bin/bash
# The evidence of the final tree.  Run on the GPU box from the repo root AFTER the last source commit:
#     gpurun --timeout 2400 -- 'bash scripts/final_gate.sh <label>'
# writes gpurun_out/final_gate/pytest_gpu_<label>.txt and pytest_cpu_<label>.txt, each starting with the tree's identity: the commit
# hash stamped into the snapshot by the caller (HEAD.txt, written locally by `git rev-parse HEAD > HEAD.txt` before the call -- the
# box has no .git) and a sha256 over every tracked source file, so that a log can be matched to a tree without trusting either.
# Copy the logs to profiles/r05/.  No kernel commit after it.
set -u
label=${1:-run}
out=gpurun_out/final_gate; mkdir -p $out
ident() {
  echo "HEAD $(cat HEAD.txt 2>/dev/null || echo unknown)"
  echo "tree sha256 $( (find yolo-v4-tf.keras_amd/csrc yolo-v4-tf.keras_amd/yolo4hip include oracle tests bench.py __graft_entry__.py -type f \( -name '*.py' -o -name '*.hip' -o -name '*.h' -o -name '*.json' \) -not -path '*/build/*' -not -path '*/__pycache__/*' | LC_ALL=C sort | xargs sha256sum | sha256sum | cut -d' ' -f1) )"
  echo "box $(hostname) $(date -u +%Y-%m-%dT%H:%M:%SZ) $(rocm-smi --showclocks 2>/dev/null | grep -E 'sclk' | head -1)"
}
{ ident; python -m pytest tests/ -x -q -m gpu 2>&1; echo "rc=$?"; } > $out/pytest_gpu_$label.txt
{ ident; python -m pytest tests/ -x -q -m "not gpu" 2>&1; echo "rc=$?"; } > $out/pytest_cpu_$label.txt
tail -4 $out/pytest_gpu_$label.txt; tail -3 $out/pytest_cpu_$label.txt
