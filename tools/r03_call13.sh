#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03m; rm -rf $O; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_per -- python $R/tools/rollout_bench_per.py > $O/trace_per.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$O/trace_per/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = open("$O/per_rollout_kernels.txt", "w")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1][-100:])):
    last = v[-100:]
    line = "%-40s calls %5d   last %3d calls: mean %8.1f us  min %8.1f  max %8.1f" % (k[:40], len(v), len(last), sum(last) / len(last) / 1e3, min(last) / 1e3, max(last) / 1e3)
    print(line); out.write(line + "\n")
PY
rm -rf $O/trace_per
