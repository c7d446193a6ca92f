#!/bin/bash
# usage: r03_prio_sweep.sh ; the default line and three other contents: shipped library (adaptive k_pintra_rows grid), the grid fixed at 1 and at 32 pictures, library B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for c in s1 s2 s3 scroll; do
for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 --content $c > $O/pc_${c}_A_$rep.json 2> /dev/null
  MI355X_H264_PINTRA_SLOTS=1 timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 --content $c > $O/pc_${c}_K1_$rep.json 2> /dev/null
  MI355X_H264_PINTRA_SLOTS=32 timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 --content $c > $O/pc_${c}_K32_$rep.json 2> /dev/null
  MI355X_H264_LIB=$R/media_amd/lib/libmi355x_h264_ab.so timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 --content $c > $O/pc_${c}_B_$rep.json 2> /dev/null
done
done
python - <<PY
import json
for c in ("s1", "s2", "s3", "scroll"):
  for v in ["A", "K1", "K32", "B"]:
    out = []
    for rep in (1, 2):
        try:
            d = json.load(open("$O/pc_%s_%s_%d.json" % (c, v, rep)))
            k = d.get("kernels") or {}
            out.append("%.0f/%.0f %s" % (d["value"], d["single_gop_in_flight_fps"], {a: round(b["ms_per_launch"], 3) for a, b in k.items()}))
        except Exception as ex:
            out.append("unreadable %s" % ex)
    print(c, v, " ".join(out))
PY
