#!/bin/bash
# usage: r03_prio_sweep.sh ; the default line with the motion-search lock on (default) / off (MI355X_H264_ME_TURNS=0), instance 1 started 0 / 1.6 ms after instance 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for rep in 1 2 3 4; do
 for sg in 0 1.6; do
  timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 --stagger-ms $sg > $O/turn_on_${sg}_$rep.json 2> /dev/null
  MI355X_H264_ME_TURNS=0 timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 5 --stagger-ms $sg > $O/turn_off_${sg}_$rep.json 2> /dev/null
 done
done
python - <<PY
import json
for sg in ("0", "1.6"):
  for v in ("on", "off"):
    out = []
    for rep in (1, 2, 3, 4):
        try:
            d = json.load(open("$O/turn_%s_%s_%d.json" % (v, sg, rep)))
            k = d["kernels"]
            out.append("%.0f (me %.2f tq %.3f cavlc %.3f db %.2f)" % (d["value"], k["me"]["ms_per_launch"], k["tq"]["ms_per_launch"], k["cavlc"]["ms_per_launch"], k["deblock"]["ms_per_launch"]))
        except Exception as ex:
            out.append("unreadable")
    print("stagger", sg, "lock", v, " ".join(out))
PY
