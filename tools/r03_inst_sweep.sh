#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for cfg in "64 2 1" "64 2 0" "96 3 1" "64 4 1" "128 4 1" "128 2 1" "96 2 1" "64 2 1" "64 2 0"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 --gops-in-flight $1 --instances $2 --phase-lock $3 > $O/inst_$1_$2_$3.json 2> $O/inst_$1_$2_$3.err
  python - <<PY
import json
try:
    d = json.load(open("$O/inst_$1_$2_$3.json"))
    print("G=$1 I=$2 lock=$3", d["value"], d["ms_per_step"], d["config"]["selfcheck_batch_equals_single"], {k: x["ms_per_launch"] for k, x in d["kernels"].items()})
except Exception as ex:
    print("G=$1 I=$2 lock=$3 failed", ex)
PY
done
