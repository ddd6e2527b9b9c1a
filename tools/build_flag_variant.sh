#!/bin/bash
# A whole-library variant under extra hipcc flags (every unit recompiled in parallel, split objects as the product build):
# tools/_bin/libfreerl_hip_<name>.so, for FRL_HIP_VARIANT=<name> A/Bs of compiler options on one box.
#     bash tools/build_flag_variant.sh ilp -mllvm -amdgpu-sched-strategy=iterative-ilp
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); T=/tmp/frl_var_$NAME; mkdir -p $T $R/tools/_bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
for src in $R/freerl_amd/csrc/frl_api.hip $R/freerl_amd/csrc/kernels_*.hip; do
  u=$(basename $src .hip)
  ( cd $T && hipcc $F "$@" -save-temps=obj -c $src -o $T/$u.o > $T/$u.log 2>&1 || { echo "FAILED $u"; tail -5 $T/$u.log; } ) &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libfreerl_hip_$NAME.so $(ls $T/*.o | grep -v -e -hip-)
for s in $T/kernels_*gfx950.s; do grep "vgpr_spill_count" $s | awk -v f=$(basename $s | cut -d- -f1) '{n+=$2} END{if (n) printf "%s: %d spilled VGPRs in all\n", f, n}'; done
echo built $R/tools/_bin/libfreerl_hip_$NAME.so
