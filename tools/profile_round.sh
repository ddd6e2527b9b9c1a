#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh'
# kernel trace + stats first, then one --pmc pass per counter group (never combined with other traces).
# The section-stamp tools need developer variants of the library under tools/_bin/ (not shipped with the product snapshot: build them
# right before this call and delete them after it):
#   for v in ppot:-DFRL_PPO_TIMING widet:-DFRL_WIDE_TIMING phase:-DFRL_PHASE_TIMING; do FRL_HIP_VARIANT=${v%%:*} FRL_HIPCC_FLAGS=${v#*:} \
#       python -c "from freerl_amd import _native as N; N.build(force=True)"; done;  bash tools/build_solo_timing.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 2 --headline-only"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- $BENCH > $O/pmc$i.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc$i $O/pmc$i.json
done
python $R/tools/pmc_summary.py --traffic $O/traffic.json $O/pmc1.json $O/pmc2.json ac_critic_v2_twin_nv_kernel
# ... and for the other families' dominant kernels: FETCH / WRITE / MFMA counters of the config shapes, the DQN launch and the Categorical update
for w in "cfg:python $R/tools/config_bench.py C4 C5 h256 512" "dqn:python $R/tools/dqn_bench.py 512"; do
  tag=${w%%:*}; cmd=${w#*:}; j=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"; do
    j=$((j+1))
    cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${tag}$j -- $cmd > $O/pmc_${tag}$j.log 2>&1
    python $R/tools/pmc_summary.py $O/pmc_${tag}$j $O/pmc_${tag}$j.json
  done
done
rm -rf $O/pmc_cfg[0-9] $O/pmc_dqn[0-9]
# the single-learner kernels (kernels_solo.hip): trace + section stamps + the loops
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_solo -- python $R/tools/single_bench.py 1000 > $O/stats_solo.log 2>&1
cp $(ls $O/stats_solo/*/*kernel_stats.csv | head -1) $O/kernel_stats_single.csv
cd $R
timeout 300 python tools/single_bench.py 2000 > $O/single_bench.txt 2>&1 < /dev/null        # (the numbers: without the profiler attached)
timeout 300 python tools/small_pop_bench.py 1 2 4 8 12 16 17 32 64 128 129 256 512 > $O/small_pop_bench.txt 2>&1 < /dev/null
for a in td3 ddpg sac; do FRL_HIP_VARIANT=solot timeout 120 python tools/solo_timing.py $a; done > $O/solo_timing.txt 2>&1 < /dev/null
timeout 300 python tools/critic2_timing.py 512 > $O/critic2_timing.txt 2>&1
FRL_CHAIN_WAVES=4 timeout 300 python tools/critic2_timing.py 512 > $O/critic2_timing_w4.txt 2>&1        # (round 5's four-wave kernel, same box)
FRL_HIP_VARIANT=bwdt timeout 300 python tools/bwd_timing.py 512 > $O/bwd_timing.txt 2>&1 < /dev/null   # FRL_UNIT_OUT=bwdt FRL_UNIT_FLAGS=-DFRL_BWD_TIMING bash tools/build_unit_timing.sh kernels_critic2
FRL_CHAIN_WAVES=4 timeout 300 python bench.py --headline-only --steps 60 --warmup 3 > $O/bench_headline_w4.json 2>/dev/null   # the A/B of the round: four-wave kernels ...
timeout 300 python bench.py --headline-only --steps 60 --warmup 3 > $O/bench_headline_w8.json 2>/dev/null                      # ... against the eight-wave ones
timeout 120 $R/tools/_bin/lds_put > $O/lds_put.txt 2>&1
timeout 300 python tools/actor2_timing.py 512 td3 > $O/actor2_timing.txt 2>&1
timeout 300 python tools/actor2_timing.py 512 sac >> $O/actor2_timing.txt 2>&1
timeout 300 python tools/ppo_timing.py 256 > $O/ppo_timing.txt 2>&1 < /dev/null
timeout 300 python tools/dqn2_timing.py 512 > $O/dqn2_timing.txt 2>&1 < /dev/null
timeout 300 $R/tools/_bin/chain_bench > $O/chain_bench.txt 2>&1
timeout 60 $R/tools/_bin/lane_xor_test > $O/lane_xor_test.txt 2>&1        # hipcc --offload-arch=gfx950 -O3 -I freerl_amd/csrc tools/lane_xor_test.hip -o tools/_bin/lane_xor_test
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ppo -- python $R/tools/ppo_bench.py 256 > $O/stats_ppo.log 2>&1
cp $(ls $O/stats_ppo/*/*kernel_stats.csv | head -1) $O/kernel_stats_ppo.csv
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dqn -- python $R/tools/dqn_bench.py 512 > $O/stats_dqn.log 2>&1
cp $(ls $O/stats_dqn/*/*kernel_stats.csv | head -1) $O/kernel_stats_dqn.csv
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg -- python $R/tools/config_bench.py C4 C5 h256 512 > $O/stats_cfg.log 2>&1
cp $(ls $O/stats_cfg/*/*kernel_stats.csv | head -1) $O/kernel_stats_configs.csv
cd $R && timeout 300 python tools/config_bench.py 1 512 > $O/config_bench.txt 2>&1
FRL_CRITIC_V2=0 timeout 300 python tools/config_bench.py C4 C5 h256 512 > $O/config_bench_rowchunk.txt 2>&1
# one learner with a wide first layer (kernels_solow.hip): trace, PMC, section stamps (variant: bash tools/build_unit_variant.sh kernels_solow solowt -DFRL_SOLO_TIMING),
# the sixteen-workgroup family against the row-chunk chain on the same box, with and without its helper workgroups / pre-draw
bash $R/tools/profile_solow.sh $O
# the K-sliced chained families' sections (library variant `widet`: -DFRL_WIDE_TIMING)
for c in sac_c4 maddpg_c5 td3_h256; do timeout 300 python tools/wide_timing.py $c 256 > $O/wide_timing_$c.txt 2>&1 < /dev/null; done
timeout 300 python tools/dqn_bench.py 1 512 2048 4096 > $O/dqn_bench.txt 2>&1
timeout 300 python tools/ppo_bench.py 1 64 256 > $O/ppo_bench.txt 2>&1
timeout 600 python tools/rollout_bench.py 1 512 > $O/rollout_bench.txt 2>&1
timeout 300 python tools/config4_rollout.py 1 8 32 > $O/config4_rollout.txt 2>&1
FRL_CRITIC_V2=0 FRL_HIP_VARIANT=phase FRL_HIPCC_FLAGS=-DFRL_PHASE_TIMING timeout 200 python tools/phase_timing.py 1 > $O/phase_critic_p1.txt 2>&1 < /dev/null
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --spawn --headline-only --steps 10 --warmup 2 > $O/bench_spawn1.json 2> $O/bench_spawn1.err
rm -rf $O/stats $O/stats_solo $O/stats_ppo $O/stats_dqn $O/stats_cfg $O/pmc[0-9]     # keep the summaries only (gpurun_out is size-capped)
ls -la $O
