#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03e; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python tools/r03_dbg.py > $O/dbg.txt 2>&1
cat $O/dbg.txt | head -150
