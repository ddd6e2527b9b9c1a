#!/bin/bash
# Same-box A/B of library variants on the headline bench (run on the GPU box):  bash tools/ab_bench.sh "" w8half "" w8half
# ("" = the product library; anything else = FRL_HIP_VARIANT).  Prints updates/s, ms per step, the critic stage's fraction, per-kernel ms.
R=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  FRL_HIP_VARIANT=$v python $R/bench.py --headline-only --steps ${STEPS:-20} --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-10s %9.0f updates/s  %.4f ms/step  critic %.4f ms  frac %.4f  %s' % ('$v' or 'product', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], {k: round(v['avg_ms'], 4) for k, v in r.get('kernels', {}).items()}))"
done
