#!/bin/bash
# Developer loop on the GPU box: the GPU test tier, then the headline numbers of the bench workload.
#   gpurun --timeout 1800 -- 'bash tools/dev_check.sh [extra command ...]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dev; rm -rf $O; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.log | tail -n 2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
for i in 1 2; do
timeout 300 python bench.py --headline-only --steps 40 --warmup 4 > $O/bench_headline.json 2> $O/bench_headline.err < /dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/dev/bench_headline.json')); r=d['roofline']
print('value %.0f frac %.4f launch %.4f ms step_tf %.1f'%(d['value'], r['frac'], r['avg_launch_ms'], r['step_tflops']), {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()})
PY
done
for c in "$@"; do echo "== $c"; timeout 600 bash -c "$c" < /dev/null 2>&1 | tail -n 40; done
