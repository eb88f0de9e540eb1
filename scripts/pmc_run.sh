#!/bin/bash
# usage: pmc_run.sh <outdir> <tiles.json> "<COUNTER COUNTER ...>" ["<second pass counters>" ...]
# One rocprofv3 --pmc pass per counter group (kernel trace only, program straight after `--`, as the pool requires).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; T=$2; shift 2
mkdir -p $O
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pass$i -- python3 bench.py --steps 2 --blocks 1 --warmup 1 --no-cpu-baseline --load-tiles $T > /dev/null 2> $O/pass$i.err
  find $O/pass$i -name "*kernel_trace.csv" -delete; find $O/pass$i -name "*agent_info.csv" -delete
done
python scripts/pmc_kernels.py $O/summary.json $O/pass* > $O/summary.txt
