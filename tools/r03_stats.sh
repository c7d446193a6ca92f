#!/bin/bash
# usage: r03_stats.sh <tag> ; rocprofv3 kernel stats of the default workload (3 steps)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plugin > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-70s calls %6s avg %10.1f us  %5s %%" % (r["Name"].replace("void h264::","")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
