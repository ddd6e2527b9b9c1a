#!/bin/bash
# tools/_bin/libfreerl_hip_solot.so = the product library with kernels_solo.hip recompiled under -DFRL_SOLO_TIMING (tools/solo_timing.py).
# Run after the product build (python -c 'import __graft_entry__ as g; g.build()'); seconds.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/freerl_amd/_lib/obj; mkdir -p $R/tools/_bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -DFRL_SOLO_TIMING -c $R/freerl_amd/csrc/kernels_solo.hip -o /tmp/kernels_solo_t.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libfreerl_hip_solot.so $(ls $O/*.o | grep -v kernels_solo.o) /tmp/kernels_solo_t.o
echo built $R/tools/_bin/libfreerl_hip_solot.so
