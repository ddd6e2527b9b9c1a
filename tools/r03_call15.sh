#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03o; rm -rf $O; mkdir -p $O; cd $R
for v in nt a3 ld st sn nt; do
  FRL_HIP_VARIANT=$v timeout 300 python bench.py --headline-only --steps 40 --warmup 4 > $O/bench_$v.json 2> $O/bench_$v.err
  V=$v python - <<'PY'
import json, os
v=os.environ['V']
d=json.load(open('gpurun_out/r03o/bench_%s.json'%v)); r=d['roofline']
print('variant [%s] value %.0f frac %.4f launch %.4f ms'%(v, d['value'], r['frac'], r['avg_launch_ms']), {k:round(x['avg_ms'],4) for k,x in r['kernels'].items()})
PY
done
