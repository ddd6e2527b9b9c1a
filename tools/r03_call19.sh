#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03s; rm -rf $O; mkdir -p $O; cd $R
export FRL_HIP_VARIANT=phase FRL_HIPCC_FLAGS=-DFRL_PHASE_TIMING
FRL_RAW_MARKS=1 timeout 200 python tools/phase_timing.py 1 < /dev/null 2>&1 | tee $O/marks_critic_p1.txt
FRL_RAW_MARKS=1 timeout 200 python tools/phase_timing.py 1 actor < /dev/null 2>&1 | tee $O/marks_actor_p1.txt
timeout 200 python tools/phase_timing.py 1 < /dev/null 2>&1 | head -3
unset FRL_HIP_VARIANT FRL_HIPCC_FLAGS
for rc in 16 32; do echo "FRL_RC=$rc"; FRL_COOP_TAIL=0 FRL_RC=$rc timeout 200 python tools/single_bench.py 2000 2>&1 | grep asyn | tee $O/single_rc$rc.txt; done
