#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03q; rm -rf $O; mkdir -p $O; cd $R
timeout 200 python tools/single_bench.py 2000 2>&1 | tee $O/single_coop.txt
FRL_COOP_TAIL=0 timeout 200 python tools/single_bench.py 2000 2>&1 | tee $O/single_nocoop.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -n 2; grep -E "^FAILED|Error" $O/pytest.log | head
timeout 300 python tools/config_bench.py 1 8 < /dev/null 2>&1 | tee $O/config_bench_small.txt
