#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03r; rm -rf $O; mkdir -p $O; cd $R
timeout 200 python tools/single_bench.py 2000 2>&1 | tee $O/single_coop.txt
FRL_COOP_TAIL=0 timeout 200 python tools/single_bench.py 2000 2>&1 | grep asyn | tee $O/single_nocoop.txt
FRL_HIP_VARIANT=phase FRL_HIPCC_FLAGS=-DFRL_PHASE_TIMING timeout 200 python tools/phase_timing.py 1 < /dev/null 2>&1 | tee $O/phase_critic_p1.txt
FRL_HIP_VARIANT=phase FRL_HIPCC_FLAGS=-DFRL_PHASE_TIMING timeout 200 python tools/phase_timing.py 1 actor < /dev/null 2>&1 | tee $O/phase_actor_p1.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o single -- python $R/tools/single_bench.py 300 > $O/prof.log 2>&1 < /dev/null
f=$(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-140
