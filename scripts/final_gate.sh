#!/bin/bash
# The evidence of the final tree.  Run on the GPU box from the repo root AFTER the last source commit:
#     gpurun --timeout 2400 -- 'bash scripts/final_gate.sh <label>'
# writes gpurun_out/final_gate/pytest_gpu_<label>.txt, pytest_cpu_<label>.txt and determinism_<label>.txt, each starting with the tree's identity: the commit
# hash stamped into the snapshot by the caller (HEAD.txt, written locally by `git rev-parse HEAD > HEAD.txt` before the call -- the
# box has no .git) and a sha256 over every tracked source file, so that a log can be matched to a tree without trusting either.
# Copy the logs to profiles/r06/.  No kernel commit after it.
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
# round 6: the determinism hunt and the halo2 race screens on the same tree
{ ident; timeout 900 python scripts/determinism_hunt.py --sweep --iters 200 2>&1 | tail -30; for dt in bf16 f16; do timeout 600 python scripts/h2_stress.py --reps 100 --dtype $dt --act 1 2>&1 | grep -E "TOTAL|[1-9][0-9]*/100"; timeout 600 python scripts/h2_det.py --dtype $dt --forwards 100 2>&1 | tail -2; done; } > $out/determinism_$label.txt 2>&1
tail -4 $out/pytest_gpu_$label.txt; tail -3 $out/pytest_cpu_$label.txt; tail -3 $out/determinism_$label.txt
