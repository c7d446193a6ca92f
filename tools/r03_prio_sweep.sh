#!/bin/bash
# usage: r03_prio_sweep.sh ; the default line with the IDR row wavefront holding 16 / 24 / 32 pictures at a time (MI355X_H264_INTRA_SLOTS)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for rep in 1 2 3 4; do
  for p in 16 24 32; do
  MI355X_H264_INTRA_SLOTS=$p timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 > $O/sl_${p}_$rep.json 2> /dev/null
  done
done
python - <<PY
import json
for v in ("16", "24", "32"):
    out = []
    for rep in (1, 2, 3, 4):
        try:
            d = json.load(open("$O/sl_%s_%d.json" % (v, rep)))
            out.append("%.0f/%.0f (intra %.2f alone %.2f)" % (d["value"], d["single_gop_in_flight_fps"], d["kernels"]["intra"]["ms_per_launch"], d["kernels_exclusive"]["intra"]["ms_per_launch"]))
        except Exception as ex:
            out.append("unreadable")
    print(v, " ".join(out))
PY
