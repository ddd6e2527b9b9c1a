#!/bin/bash
# tools/_bin/libfreerl_hip_ppot.so, the quick way: the product objects with kernels_ppo2.hip recompiled under -DFRL_PPO_TIMING and
# frl_api.hip under -DFRL_PPO_TIMING_SPLIT (tools/ppo_timing.py reads the stamps through frl_debug_ppo_clocks).  ~1.5 minutes instead
# of the unity variant's 8.  Run after the product build.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/freerl_amd/_lib/obj; mkdir -p $R/tools/_bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
hipcc $F -DFRL_PPO_TIMING -c $R/freerl_amd/csrc/kernels_ppo2.hip -o /tmp/kernels_ppo2_t.o &
hipcc $F -DFRL_PPO_TIMING_SPLIT -c $R/freerl_amd/csrc/frl_api.hip -o /tmp/frl_api_t.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libfreerl_hip_ppot.so $(ls $O/*.o | grep -v -e kernels_ppo2.o -e frl_api.o) /tmp/kernels_ppo2_t.o /tmp/frl_api_t.o
echo built $R/tools/_bin/libfreerl_hip_ppot.so
