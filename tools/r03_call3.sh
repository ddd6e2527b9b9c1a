#!/bin/bash
# round 3, GPU call 3: the persistent critic kernel (kernels_critic3.hip) against round 2's launch shape, correctness first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for P in 512 1024; do
  for PE in 0 1; do
    FRL_CRITIC_PERSIST=$PE timeout 600 python bench.py --learners $P --headline-only --steps 40 --warmup 4 > $O/bench_P${P}_persist$PE.json 2> $O/bench_P${P}_persist$PE.err
  done
done
FRL_CRITIC_PERSIST=1 timeout 300 python tools/critic2_timing.py 512 > $O/critic3_timing.txt 2>&1
FRL_CRITIC_PERSIST=0 timeout 300 python tools/critic2_timing.py 512 > $O/critic2_timing.txt 2>&1
tail -5 $O/pytest.log; cat $O/critic3_timing.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/bench_P*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'value %.0f frac %.4f launch %.4f ms'%(d['value'], r['frac'], r['avg_launch_ms']), {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()})
    except Exception as ex: print(f, 'ERR', ex)
PY
