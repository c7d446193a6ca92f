#!/bin/bash
# usage: r03_ab.sh <tag> [bench args] ; the default line with the shipped library (A) and media_amd/lib/libmi355x_h264_ab.so (B), alternating, same box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
tag=$1; shift
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 6 "$@" > $O/ab_${tag}_A$rep.json 2> /dev/null
  MI355X_H264_LIB=$R/media_amd/lib/libmi355x_h264_ab.so timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 6 "$@" > $O/ab_${tag}_B$rep.json 2> /dev/null
done
python - <<PY
import json
for v in "AB":
    for rep in (1, 2, 3):
        try:
            d = json.load(open("$O/ab_${tag}_%s%d.json" % (v, rep)))
            print(v, rep, d["value"], d["single_gop_in_flight_fps"], {k: x["ms_per_launch"] for k, x in d["kernels"].items()}, {k: x["ms_per_launch"] for k, x in d.get("kernels_exclusive", {}).items()})
        except Exception as ex:
            print(v, rep, "unreadable", ex)
PY
