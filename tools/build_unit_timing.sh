#!/bin/bash
# tools/_bin/libfreerl_hip_ppot.so, the quick way: the product objects with ONE unit (default kernels_ppo2; or kernels_dqn2,
# kernels_critic2, kernels_actor2) recompiled under -DFRL_PPO_TIMING -DFRL_CLK_COPY=1 and frl_api.hip under -DFRL_PPO_TIMING_SPLIT
# (tools/ppo_timing.py, dqn2_timing.py, critic2_timing.py, actor2_timing.py read the stamps through frl_debug_ppo_clocks).
# ~1.5 minutes instead of the unity variant's 8.  Run after the product build.
#     bash tools/build_unit_timing.sh [kernels_ppo2]        (FRL_UNIT_FLAGS: extra flags of the unit; FRL_UNIT_OUT: variant name, default ppot)
set -e
U=${1:-kernels_ppo2}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/freerl_amd/_lib/obj; mkdir -p $R/tools/_bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
hipcc $F -DFRL_PPO_TIMING -DFRL_CLK_COPY=1 $FRL_UNIT_FLAGS -c $R/freerl_amd/csrc/$U.hip -o /tmp/${U}_t.o &
hipcc $F -DFRL_PPO_TIMING_SPLIT -c $R/freerl_amd/csrc/frl_api.hip -o /tmp/frl_api_t.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libfreerl_hip_${FRL_UNIT_OUT:-ppot}.so $(ls $O/*.o | grep -v -e $U.o -e frl_api.o) /tmp/${U}_t.o /tmp/frl_api_t.o
echo built $R/tools/_bin/libfreerl_hip_${FRL_UNIT_OUT:-ppot}.so with the stamps of $U
