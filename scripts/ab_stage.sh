#!/bin/bash
# A/B of the stage kernel on one box: per-op event tables + rocprofv3 kernel stats for both variants (same tuned tiles)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_stage; rm -rf $O; mkdir -p $O
python bench.py --no-cpu-baseline --save-tiles $O/tiles.json > $O/tune.json 2> $O/tune.err
python bench.py --no-cpu-baseline --load-tiles $O/tiles.json --per-op > $O/on.json 2> $O/on.err
python bench.py --no-cpu-baseline --load-tiles $O/tiles.json --per-op --no-stage-fusion > $O/off.json 2> $O/off.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_on -- python3 bench.py --steps 5 --blocks 1 --warmup 2 --no-cpu-baseline --load-tiles $O/tiles.json > $O/stats_on.json 2> $O/stats_on.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_off -- python3 bench.py --steps 5 --blocks 1 --warmup 2 --no-cpu-baseline --load-tiles $O/tiles.json --no-stage-fusion > $O/stats_off.json 2> $O/stats_off.err
find $O -name "*kernel_trace.csv" -size +30M -delete
find $O -name "*_agent_info.csv" -delete
ls $O/stats_on/*/ | head
