#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03g; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python tools/r03_dbg2.py > $O/dbg2.txt 2>&1; cat $O/dbg2.txt
