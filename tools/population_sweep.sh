#!/bin/bash
# updates/s and the critic stage's fraction of the fp32 MFMA peak against the number of resident learners (bench.py's workload)
#   gpurun --timeout 900 -- 'bash tools/population_sweep.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/sweep; mkdir -p $O; cd $R
for P in 64 128 256 384 512 768 1024 1536; do
  timeout 300 python bench.py --headline-only --steps 30 --warmup 4 --learners $P > $O/b_$P.json 2> $O/b_$P.err < /dev/null
  P=$P python - <<'PY'
import json, os
P = os.environ['P']
d = json.load(open('gpurun_out/sweep/b_%s.json' % P)); r = d['roofline']
print('P=%5s  %-9s %8.0f updates/s  %.3f ms per step  critic launch %.4f ms = %.3f of the fp32 MFMA peak, step %.1f TFLOP/s' % (
    P, d['config']['kernel_family'].split()[0], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['step_tflops']))
PY
done
