#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03h; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python tools/r03_dbg2.py > $O/dbg2.txt 2>&1; grep -c "differ" $O/dbg2.txt
timeout 600 python tools/r03_ab.py > $O/ab.txt 2>&1; cat $O/ab.txt
for P in 512 1024; do
  for PE in 0 1; do
    FRL_CRITIC_PERSIST=$PE timeout 600 python bench.py --learners $P --headline-only --steps 40 --warmup 4 > $O/bench_P${P}_persist$PE.json 2> $O/bench_P${P}_persist$PE.err
  done
done
FRL_CRITIC_PERSIST=1 timeout 300 python tools/critic2_timing.py 512 > $O/critic3_timing.txt 2>&1; cat $O/critic3_timing.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03h/bench_P*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'value %.0f frac %.4f launch %.4f ms'%(d['value'], r['frac'], r['avg_launch_ms']), {k:round(v['avg_ms'],4) for k,v in r['kernels'].items()})
    except Exception as ex: print(f, 'ERR', ex)
PY
