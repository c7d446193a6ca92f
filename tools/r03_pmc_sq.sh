#!/bin/bash
# usage: r03_pmc_sq.sh <tag> ; SQ counters of the default workload (1 step), one rocprofv3 --pmc pass, summed per kernel and launch shape
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-plugin > $O/pmc_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc_sq/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].replace("void h264::", "")[:40], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in sorted(acc):
    if "intra" in k[0] or "k_me" in k[0]:
        print(k, n[k], {c: round(v / max(1, n[k])) for c, v in acc[k].items()})
PY
