#!/bin/bash
# round 3, GPU call 2: stage 1 of the headline kernel's rework (fragment-image parameters in HBM, linear staging, Adam from the accumulators)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --headline-only --steps 40 --warmup 4 > $O/bench_headline.json 2> $O/bench_headline.err
timeout 300 python tools/critic2_timing.py 512 > $O/critic2_timing.txt 2>&1
timeout 300 python tools/rollout_bench.py 512 > $O/rollout_bench.txt 2>&1
timeout 300 python tools/config_bench.py 512 > $O/config_bench.txt 2>&1
tail -5 $O/pytest.log; cat $O/critic2_timing.txt
