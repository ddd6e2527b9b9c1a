#!/bin/bash
# The kernels_solow.hip part of tools/profile_round.sh (also on its own: gpurun -- 'bash tools/profile_solow.sh gpurun_out/prof_solow'):
# rocprofv3 kernel trace + stats and three PMC passes of one SAC learner at Humanoid-v4's dims, the section stamps, the A/Bs.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-$R/gpurun_out/prof_solow}
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
export TMPDIR=/tmp
CMD="python $R/tools/config_bench.py 1 C4"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_solow -- $CMD > $O/stats_solow.log 2>&1
cp $(ls $O/stats_solow/*/*kernel_stats.csv | head -1) $O/kernel_stats_solow.csv
j=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"; do
  j=$((j+1))
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_solow$j -- $CMD > $O/pmc_solow$j.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_solow$j $O/pmc_solow$j.json
done
rm -rf $O/stats_solow $O/pmc_solow[0-9]
cd $R
# (the stamps: the two-launch form — in the fused policy step the actor half's stamps overwrite the critic half's)
{ for a in "sac 376 17" "td3 376 17" "td3 17 6" "maddpg"; do echo "== $a"; FRL_SOLOW_FUSE=0 FRL_HIP_VARIANT=solowt timeout 120 python tools/solow_timing.py $a; done; } 2>&1 < /dev/null | grep -v amdgpu.ids > $O/solow_timing.txt
{ echo "== default (kernels_solow.hip)"; timeout 300 python tools/config_bench.py 1 2 4 5 8 16 17 C4
  echo "== FRL_SOLOW_FUSE=0 (critic and actor stage as two launches)"; FRL_SOLOW_FUSE=0 timeout 300 python tools/config_bench.py 1 2 4 8 C4
  echo "== FRL_SOLOW_HELPERS=0 (sixteen workgroups per learner only: no helpers, no pre-draw, two launches)"; FRL_SOLOW_HELPERS=0 timeout 300 python tools/config_bench.py 1 4 C4
  echo "== FRL_SOLO_PREDRAW=0"; FRL_SOLO_PREDRAW=0 timeout 300 python tools/config_bench.py 1 C4
  echo "== FRL_SOLOW=0 (the row-chunk chain)"; FRL_SOLOW=0 timeout 300 python tools/config_bench.py 1 2 4 8 16 C4
  echo "== FRL_CRITIC_V2=1 (the K-sliced chained family: one workgroup per learner)"; FRL_CRITIC_V2=1 timeout 300 python tools/config_bench.py 1 16 C4
  echo "== config 5 (MADDPG, three agents, batch 1024): default (kernels_solow.hip at one learner: 3 units x 64 workgroups)"; timeout 300 python tools/config_bench.py 1 2 4 C5
  echo "== FRL_SOLOW=0"; FRL_SOLOW=0 timeout 300 python tools/config_bench.py 1 C5
  echo "== FRL_SOLOW=0 FRL_DRAW_SCAN=1 (round 5's draw_kernel: the scan of every entry's predecessors)"; FRL_SOLOW=0 FRL_DRAW_SCAN=1 timeout 300 python tools/config_bench.py 1 C5; } 2>&1 | grep -v amdgpu.ids > $O/config_bench_solow.txt
timeout 60 $R/tools/_bin/mfma_chain > $O/mfma_chain.txt 2>&1       # hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o tools/_bin/mfma_chain
