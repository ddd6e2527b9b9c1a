#!/bin/bash
# A developer variant of the library in ~1 min: the product objects with ONE unit (or a comma-separated list of units) recompiled under
# extra flags, linked as tools/_bin/libfreerl_hip_<name>.so (load it with FRL_HIP_VARIANT=<name>; A/B two versions of a kernel on the
# same box: tools/ab_bench.sh).  Run after the product build.
#     bash tools/build_unit_variant.sh kernels_c51 c51skip2 -DFRL_C51_SKIP=2        [SRC=/path/to/other/kernels_c51.hip, single unit only]
#     bash tools/build_unit_variant.sh kernels_critic2,kernels_actor2 w8half -DFRL_FW_HALF=1
set -e
US=$1; NAME=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/freerl_amd/_lib/obj; mkdir -p $R/tools/_bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
EXCL=""; NEW=""
for U in ${US//,/ }; do
  S=$R/freerl_amd/csrc/$U.hip
  if [ -n "${SRC:-}" ] && [ "$US" = "$U" ]; then S=$SRC; fi
  hipcc $F -I $R/freerl_amd/csrc "$@" -c $S -o /tmp/${U}_$NAME.o &
  EXCL="$EXCL -e /$U.o"; NEW="$NEW /tmp/${U}_$NAME.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libfreerl_hip_$NAME.so $(ls $O/*.o | grep -v $EXCL) $NEW
echo built $R/tools/_bin/libfreerl_hip_$NAME.so: $US with "$@"
