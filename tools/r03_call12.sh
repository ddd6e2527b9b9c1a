#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03l; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python tools/rollout_bench.py 1 512 > $O/rollout_bench.txt 2>&1; cat $O/rollout_bench.txt
timeout 600 python tools/config4_rollout.py 1 8 32 > $O/config4_rollout.txt 2>&1; cat $O/config4_rollout.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_per -- python $R/tools/rollout_bench_per.py > $O/stats_per.log 2>&1
cp $(ls $O/stats_per/*/*kernel_stats.csv | head -1) $O/kernel_stats_per_rollout.csv; rm -rf $O/stats_per
head -12 $O/kernel_stats_per_rollout.csv | cut -c1-160
