#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/prof.sh) into profiles/<tag>_*.{csv,json}."""
import csv, collections, glob, json, os, sys
tag = sys.argv[1]
src = "gpurun_out/prof_%s" % tag
os.makedirs("profiles", exist_ok=True)
# 1. kernel stats (rocprofv3 --kernel-trace --stats), as produced
ks = glob.glob(src + "/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(ks)))
with open("profiles/%s_kernel_stats.csv" % tag, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
# split the launches by grid size (batched main region vs single-GOP phase) from the trace
tr = glob.glob(src + "/stats/*/*kernel_trace.csv")[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    by[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_Y"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
split = {"%s grid.y=%d" % (k[0].replace("void ", "").replace("h264::", ""), k[1] // 1): {"calls": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2)}
         for k, v in sorted(by.items()) if k[0].replace("void ", "").startswith("h264::")}
# 2. PMC passes
def pmc(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(path)[0])):
        if r["Counter_Name"] == name:
            d[(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("h264::", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return d
out = {"tag": tag, "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (stats); --steps 1 (each --pmc pass)",
       "kernel_avg_by_batch": split, "pmc": {}}
fetch = pmc(src + "/pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE")
write = pmc(src + "/pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
for k in sorted(set(fetch) | set(write)):
    f_kb = sum(fetch[k]) / len(fetch[k]) if k in fetch else None
    w_kb = sum(write[k]) / len(write[k]) if k in write else None
    out["pmc"]["%s grid=%d threads" % k] = {"FETCH_SIZE_KB_raw": None if f_kb is None else round(f_kb, 1),
                                      "WRITE_SIZE_KB": None if w_kb is None else round(w_kb, 1)}
sq = glob.glob(src + "/pmc_sq/*/*counter_collection.csv")
if sq:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sq[0])):
        d[(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("h264::", ""), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["sq"] = {"%s grid=%d threads" % k: {c: round(sum(v) / len(v)) for c, v in cs.items()} for k, cs in sorted(d.items())}
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in out["pmc"].items() if "k_tq" in k or "k_me" in k}, indent=1))
print(json.dumps({k: v for k, v in split.items() if "k_tq" in k or "k_me" in k or "deblock" in k}, indent=1))
# HBM traffic of the roofline kernel (k_tq).  MI355X_MICROARCH.md: FETCH_SIZE reads half the bytes of a wide coalesced stream and
# "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  tools/ubench_fetch.hip did
# (profiles/r03b_ubench_fetch.json): WRITE_SIZE / bytes = 1.0 for every store shape of the kernel; FETCH_SIZE / bytes = 1.0 for loads that
# touch 64-byte row segments (the chroma passes, MbInfo; the luma passes of rounds 1-3: rd4_tq) and 0.5 for loads that touch 128-byte
# segments (the luma passes since the end of round 3: rd4_tq8).  So the luma bytes the kernel reads - 512 per coded macroblock,
# source + prediction - are in the counter at half: traffic = WRITE_SIZE + FETCH_SIZE + 256 x coded macroblocks, the coded count from
# the bench line of the same build (profiles/r03_bench_default.json).  The lockstep launch = the largest grid.
tq = [(k, v) for k, v in out["pmc"].items() if k.startswith("k_tq") and not k.startswith("k_tq8") and not k.startswith("k_tq_list") and v["FETCH_SIZE_KB_raw"] and v["WRITE_SIZE_KB"]]
if tq:
    k, v = max(tq, key=lambda kv: int(kv[0].split("grid=")[1].split()[0]))
    threads = int(k.split("grid=")[1].split()[0])
    mbs = threads // 64 * 8
    try:
        coded = json.loads(open("profiles/r03_bench_default.json").read().strip().splitlines()[-1])["roofline"]["coded_fraction"] * mbs
    except Exception:
        coded = 0.658 * mbs
    half_counted = int(256 * coded)
    tr_ = {"kernel": "k_tq", "lockstep_batch": mbs // 8160, "macroblocks_per_launch": mbs, "coded_macroblocks_per_launch": round(coded, 1),
           "FETCH_SIZE_bytes": int(v["FETCH_SIZE_KB_raw"] * 1024), "WRITE_SIZE_bytes": int(v["WRITE_SIZE_KB"] * 1024),
           "luma_read_bytes_counted_at_half": 2 * half_counted, "correction_bytes": half_counted,
           "fetch_ratio_calibrated": "1.0 for 64-byte segments, 0.5 for the luma passes' 128-byte segments", "write_ratio_calibrated": 1.0,
           "calibration": "profiles/r03b_ubench_fetch.json (rd4_tq 1.0, rd4_tq8 0.5, wr4_tq / wr4_tq8 / wr32_lv 1.0)",
           "traffic_bytes_per_launch": int(v["FETCH_SIZE_KB_raw"] * 1024 + v["WRITE_SIZE_KB"] * 1024) + half_counted,
           "note": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-plugin` "
                   "(profiles/%s_summary.json).  traffic = WRITE_SIZE + FETCH_SIZE + the half of the luma reads the counter leaves out (measured "
                   "ratio 0.5 for that access shape).  bench.py compares this with coded x 1952 + settled x 2 bytes of the same launch" % tag}
    json.dump(tr_, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
    print(json.dumps(tr_, indent=1))
