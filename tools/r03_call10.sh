#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03j; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "ppo or PPO" > $O/pytest_ppo.log 2>&1; tail -n 3 $O/pytest_ppo.log
timeout 600 python tools/ppo_bench.py 1 64 256 > $O/ppo_bench.txt 2>&1; cat $O/ppo_bench.txt
timeout 300 python tools/ppo_timing.py > $O/ppo_timing.txt 2>&1; tail -n 18 $O/ppo_timing.txt
timeout 300 python tools/actor2_timing.py 512 sac > $O/actor2_timing_sac.txt 2>&1; cat $O/actor2_timing_sac.txt
timeout 300 python tools/critic2_timing.py 512 > $O/critic2_timing.txt 2>&1; cat $O/critic2_timing.txt
