#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03n; rm -rf $O; mkdir -p $O; cd $R
tools/_bin/chain_bench > $O/chain_bench.txt 2>&1; cat $O/chain_bench.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -n 2
timeout 300 python bench.py --headline-only --steps 40 --warmup 4 > $O/bench_headline.json 2> $O/bench_headline.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03n/bench_headline.json')); r=d['roofline']
print('value %.0f frac %.4f launch %.4f ms step_tf %.1f'%(d['value'], r['frac'], r['avg_launch_ms'], r['step_tflops']), {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()})
PY
timeout 300 python tools/actor2_timing.py 512 td3 > $O/actor2_timing_td3.txt 2>&1; cat $O/actor2_timing_td3.txt
timeout 300 python tools/critic2_timing.py 512 > $O/critic2_timing.txt 2>&1; cat $O/critic2_timing.txt
