#!/bin/bash
# A developer variant of the library in about a minute: tools/_bin/libfreerl_hip_<name>.so = the product's objects with ONE unit (or a
# comma-separated list of units) recompiled under extra flags (FRL_HIP_VARIANT=<name> selects it; same-box A/B runs of a kernel knob).
# Run after the product build.
#     bash tools/build_variant.sh <name> <unit[,unit...], e.g. kernels_critic2> [hipcc flags ...]
set -e
NAME=$1; US=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/freerl_amd/_lib/obj; mkdir -p $R/tools/_bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
EXCL=""; NEW=""
for U in ${US//,/ }; do
  hipcc $F "$@" -c $R/freerl_amd/csrc/$U.hip -o /tmp/${U}_${NAME}.o &
  EXCL="$EXCL -e /$U.o"; NEW="$NEW /tmp/${U}_${NAME}.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libfreerl_hip_${NAME}.so $(ls $O/*.o | grep -v $EXCL) $NEW
echo built $R/tools/_bin/libfreerl_hip_${NAME}.so: $US with "$@"
