#!/bin/bash
# usage: pmc_any.sh <outdir> <kernel-name-substring> <script.py> "<COUNTERS>" ["<COUNTERS>" ...]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; K=$2; S=$3; shift 3
mkdir -p $O
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pass$i -- python3 $S > /dev/null 2> $O/pass$i.err
  find $O/pass$i -name "*kernel_trace.csv" -delete; find $O/pass$i -name "*agent_info.csv" -delete
done
python scripts/pmc_kernels.py $O/summary.json $O/pass* > $O/summary.txt
grep "$K" $O/summary.txt
