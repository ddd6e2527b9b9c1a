#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/r03_ab.py > $O/ab.txt 2>&1
FRL_CRITIC_PERSIST=0 timeout 600 python -m pytest tests/test_gpu_longrun.py -q -k "chained" > $O/pytest_long_v2.log 2>&1
FRL_CRITIC_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_longrun.py -q -k "chained" > $O/pytest_long_v3.log 2>&1
# the background task's cost when HBM is NOT saturated: 64 learners on 32 workgroups (2 each), then 512 on 256
FRL_CRITIC_PERSIST=1 FRL_CRITIC_GRID=32 timeout 300 python tools/critic2_timing.py 64 > $O/critic3_timing_P64_grid32.txt 2>&1
FRL_CRITIC_PERSIST=1 FRL_CRITIC_GRID=32 timeout 300 python tools/critic2_timing.py 32 > $O/critic3_timing_P32_grid32.txt 2>&1
cat $O/ab.txt; tail -4 $O/pytest_long_v2.log $O/pytest_long_v3.log; cat $O/critic3_timing_P64_grid32.txt $O/critic3_timing_P32_grid32.txt
