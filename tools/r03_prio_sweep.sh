#!/bin/bash
# usage: r03_prio_sweep.sh ; the default line of the shipped library, five times
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for rep in 1 2 3 4 5; do
  timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 > $O/cur_$rep.json 2> /dev/null
done
python - <<PY
import json
out = []
for rep in (1, 2, 3, 4, 5):
    try:
        d = json.load(open("$O/cur_%d.json" % rep)); k = d["kernels"]
        out.append("%.0f/%.0f (me %.2f tq %.3f cavlc %.3f db %.2f in %.2f)" % (d["value"], d["single_gop_in_flight_fps"], k["me"]["ms_per_launch"], k["tq"]["ms_per_launch"], k["cavlc"]["ms_per_launch"], k["deblock"]["ms_per_launch"], k["intra"]["ms_per_launch"]))
    except Exception as ex:
        out.append("unreadable")
print(" ".join(out))
PY
