#!/bin/bash
# round 3, GPU call 1: the full -m gpu suite, the default bench line, the population sweep of the headline kernel, round 2's section timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
rm -rf $O; mkdir -p $O
cd $R
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for P in 768 1024; do
  timeout 600 python bench.py --learners $P --headline-only --steps 20 --warmup 2 > $O/bench_P$P.json 2> $O/bench_P$P.err
done
timeout 300 python bench.py --spawn --headline-only --steps 10 --warmup 2 > $O/bench_spawn.json 2> $O/bench_spawn.err
timeout 300 python tools/critic2_timing.py 512 > $O/critic2_timing.txt 2>&1
timeout 300 python tools/dqn_bench.py 512 2048 4096 > $O/dqn_bench.txt 2>&1
timeout 300 python tools/rollout_bench.py > $O/rollout_bench.txt 2>&1
ls -la $O
tail -3 $O/pytest.log
