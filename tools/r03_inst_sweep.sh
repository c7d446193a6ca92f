#!/bin/bash
# usage: r03_inst_sweep.sh ; the default workload with 2 instances of 24 / 32 / 48 / 64 GOPs (final build)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for rep in 1 2; do
for cfg in "2 48" "2 64" "2 96" "2 128"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --no-plugin --no-cpu-baseline --steps 4 --instances $1 --gops-in-flight $2 > $O/isw_$1_$2_$rep.json 2> /dev/null
done
done
python - <<PY
import json
for cfg in ("2_48", "2_64", "2_96", "2_128"):
    out = []
    for rep in (1, 2):
        try:
            d = json.load(open("$O/isw_%s_%d.json" % (cfg, rep))); k = d["kernels"]
            out.append("%.0f (me %.2f tq %.3f cavlc %.3f db %.2f in %.2f)" % (d["value"], k["me"]["ms_per_launch"], k["tq"]["ms_per_launch"], k["cavlc"]["ms_per_launch"], k["deblock"]["ms_per_launch"], k["intra"]["ms_per_launch"]))
        except Exception as ex:
            out.append("unreadable")
    print(cfg, " ".join(out))
PY
