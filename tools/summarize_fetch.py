#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/ubench_fetch.bin (tools/prof_fetch.sh) into profiles/<tag>_ubench_fetch.json:
per access shape, counter value / bytes really moved.  usage: summarize_fetch.py <tag>"""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
src = "gpurun_out/fetch_%s" % tag
known = json.loads([l for l in open(src + "/run.log") if l.startswith("{")][-1])["bytes"]


def pmc(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(path)[0])):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d


fetch = pmc(src + "/pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE")
write = pmc(src + "/pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(src + "/stats/*/*kernel_trace.csv")[0])):
    dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"tag": tag, "what": "tools/ubench_fetch.hip on MI355X: every kernel moves a known number of bytes once (401 MB buffers, caches flushed in "
                           "between); counters from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (values are KB); "
                           "ratio = counter bytes / bytes really moved", "kernels": {}}
for k in ("rd16", "rd4_row", "rd4_tq", "rd4_tq8", "rd4_win", "wr4_tq", "wr4_tq8", "wr32_lv", "wr16", "flush_caches"):
    f = sum(fetch[k]) / len(fetch[k]) * 1024 if k in fetch else None
    w = sum(write[k]) / len(write[k]) * 1024 if k in write else None
    rd = known.get(k) if k.startswith("rd") else (known.get("rd4_win_requested") if k == "rd4_win" else known.get("flush_caches_read") if k == "flush_caches" else 0)
    if k == "rd4_win":
        rd = known["rd4_win_requested"]
    wr = known.get(k) if k.startswith("wr") else (known.get("flush_caches_written") if k == "flush_caches" else 0)
    e = {"bytes_read": rd, "bytes_written": wr, "FETCH_SIZE_bytes": None if f is None else int(f), "WRITE_SIZE_bytes": None if w is None else int(w),
         "avg_us": round(sum(dur[k]) / len(dur[k]) / 1e3, 1) if k in dur else None}
    if rd and f is not None:
        e["fetch_ratio"] = round(f / rd, 4)
    if k == "rd4_win" and f is not None:
        e["fetch_ratio_vs_unique_bytes"] = round(f / known["rd4_win_unique_upper"], 4)
    if wr and w is not None:
        e["write_ratio"] = round(w / wr, 4)
    if e["avg_us"] and (rd or wr):
        e["GBps_moved"] = round(((rd or 0) + (wr or 0)) / (e["avg_us"] * 1e-6) / 1e9, 1)
    out["kernels"][k] = e
json.dump(out, open("profiles/%s_ubench_fetch.json" % tag, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
