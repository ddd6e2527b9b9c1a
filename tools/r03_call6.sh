#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03f; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python tools/r03_dbg.py > $O/dbg.txt 2>&1
grep -E "calls=|elements off: [1-9]" $O/dbg.txt | head -40
timeout 600 python tools/r03_ab.py > $O/ab.txt 2>&1; cat $O/ab.txt
FRL_CRITIC_PERSIST=0 timeout 600 python -m pytest tests/test_gpu_longrun.py -q -k "chained" > $O/pytest_long_v2.log 2>&1; tail -n 3 $O/pytest_long_v2.log
FRL_CRITIC_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_longrun.py -q -k "chained" > $O/pytest_long_v3.log 2>&1; tail -n 3 $O/pytest_long_v3.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
