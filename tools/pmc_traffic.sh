#!/bin/bash
# The two --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) behind profiles/traffic.json, alone: after a change to one of the
# files its source digest covers that does not touch the headline kernel.   gpurun -- bash tools/pmc_traffic.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 2 --headline-only"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- $BENCH > $O/pmc$i.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc$i $O/pmc$i.json
done
python $R/tools/pmc_summary.py --traffic $O/traffic.json $O/pmc1.json $O/pmc2.json ac_critic_v2_twin_nv_kernel
rm -rf $O/pmc1 $O/pmc2
cat $O/traffic.json
