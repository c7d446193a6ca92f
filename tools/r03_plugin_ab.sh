#!/bin/bash
# usage: r03_plugin_ab.sh ; the plugin measurement (4 and 16 streams) with the shipped library and with media_amd/lib/libmi355x_h264_ab.so put in its place, alternating
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
cp media_amd/lib/libmi355x_h264.so /tmp/lib_A.so
cp media_amd/lib/libmi355x_h264_ab.so /tmp/lib_B.so
for rep in 1 2 3 4 5 6; do
  for v in A B; do
    cp /tmp/lib_$v.so media_amd/lib/libmi355x_h264.so
    timeout -k 10 300 python bench.py --mode plugin --streams 16 --plugin-frames 300 > $O/pl_${v}_$rep.json 2> /dev/null
  done
done
cp /tmp/lib_A.so media_amd/lib/libmi355x_h264.so
python - <<PY
import json
for v in "AB":
    out = []
    for rep in (1, 2, 3, 4, 5, 6):
        try:
            d = json.loads(open("$O/pl_%s_%d.json" % (v, rep)).read().strip().splitlines()[-1])
            out.append(" ".join("S=%d %.0f fps p50 %.2f p99 %.2f" % (r["streams"], r["fps_aggregate"], r["latency_ms_p50"], r["latency_ms_p99"]) for r in d["plugin"]["results"]))
        except Exception as ex:
            out.append("unreadable %s" % ex)
    print(v, " | ".join(out))
PY
