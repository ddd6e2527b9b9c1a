#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03p; rm -rf $O; mkdir -p $O; cd $R
python tools/single_bench.py 2000 2>&1 | tee $O/single_rowchunk.txt
FRL_CRITIC_V2=1 python tools/single_bench.py 2000 2>&1 | tee $O/single_chained.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o single -- python $R/tools/single_bench.py 500 > $O/prof.log 2>&1
f=$(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1); head -30 $f
