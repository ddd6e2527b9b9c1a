#!/bin/bash
# generic round-5 GPU call: bash tools/gpu_call.sh <tag> '<command>' ['<command>' ...]; each command's output -> gpurun_out/<tag>/<n>.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$1; shift; O=$R/gpurun_out/$T; rm -rf $O; mkdir -p $O; cd $R
i=0
for c in "$@"; do i=$((i+1)); echo "== [$i] $c"; timeout 900 bash -c "$c" > $O/$i.log 2>&1 < /dev/null; echo "rc=$?"; tail -n 25 $O/$i.log; done
