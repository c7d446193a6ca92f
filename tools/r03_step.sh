#!/bin/bash
# usage: r03_step.sh <tag> ; GPU parity suite + the default line (no plugin / CPU legs) for one build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests_$1.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests_$1.log
timeout -k 10 600 python bench.py --no-plugin --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_$1.json"))
print({k: d[k] for k in ("value", "ms_per_step", "single_gop_in_flight_fps")}, d["roofline"]["frac"], d["roofline_exclusive"]["frac"], d["config"]["bytes_per_gop"], {k: v["ms_per_launch"] for k, v in d["kernels"].items()})
PY
