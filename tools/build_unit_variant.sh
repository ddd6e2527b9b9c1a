#!/bin/bash
# A developer variant of the library in ~20 s: the product objects with ONE unit recompiled under extra flags, linked as
# tools/_bin/libfreerl_hip_<name>.so (load it with FRL_HIP_VARIANT=<name>; A/B two versions of a kernel on the same box).  Run after the product build.
#     bash tools/build_unit_variant.sh kernels_c51 c51skip2 -DFRL_C51_SKIP=2        [SRC=/path/to/other/kernels_c51.hip]
set -e
U=$1; NAME=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/freerl_amd/_lib/obj; mkdir -p $R/tools/_bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
SRC=${SRC:-$R/freerl_amd/csrc/$U.hip}
hipcc $F -I $R/freerl_amd/csrc "$@" -c $SRC -o /tmp/${U}_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libfreerl_hip_$NAME.so $(ls $O/*.o | grep -v -e /$U.o) /tmp/${U}_$NAME.o
echo built $R/tools/_bin/libfreerl_hip_$NAME.so
