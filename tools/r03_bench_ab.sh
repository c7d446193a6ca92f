#!/bin/bash
# usage: r03_bench_ab.sh <tag> ; the default workload and three other contents with the seeded and the exhaustive integer search
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --no-plugin --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --no-plugin --no-cpu-baseline --search exhaustive > $O/bench_exh_$1.json 2> $O/bench_exh_$1.err; echo "bench exhaustive rc=$?"
for c in scroll s2 s3; do timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 4 --content $c > $O/bench_${c}_$1.json 2> /dev/null; timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 4 --content $c --search exhaustive > $O/bench_${c}_exh_$1.json 2> /dev/null; done
python - <<PY
import json
for n in ("bench_$1", "bench_exh_$1", "bench_scroll_$1", "bench_scroll_exh_$1", "bench_s2_$1", "bench_s2_exh_$1", "bench_s3_$1", "bench_s3_exh_$1"):
    try:
        d = json.load(open("$O/%s.json" % n))
        print(n, {k: d[k] for k in ("value", "ms_per_step", "single_gop_in_flight_fps")}, d["roofline"]["frac"], d["config"]["bytes_per_gop"], {k: v["ms_per_launch"] for k, v in d["kernels"].items()})
    except Exception as ex:
        print(n, "unreadable", ex)
PY
