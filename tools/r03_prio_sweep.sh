#!/bin/bash
# usage: r03_prio_sweep.sh ; the default line (10 steps) with instance 1 started 0 / 17 / 34 ms after instance 0: 34 ms = half a GOP, the instances' IDR steps then never coincide
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for sg in 0 17 34; do
  timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 10 --stagger-ms $sg > $O/hg_${sg}_$rep.json 2> /dev/null
  done
done
python - <<PY
import json
for v in ("0", "17", "34"):
    out = []
    for rep in (1, 2, 3):
        try:
            d = json.load(open("$O/hg_%s_%d.json" % (v, rep))); k = d["kernels"]
            out.append("%.0f (me %.2f tq %.3f cavlc %.3f db %.2f in %.2f)" % (d["value"], k["me"]["ms_per_launch"], k["tq"]["ms_per_launch"], k["cavlc"]["ms_per_launch"], k["deblock"]["ms_per_launch"], k["intra"]["ms_per_launch"]))
        except Exception as ex:
            out.append("unreadable")
    print("stagger", v, " ".join(out))
PY
