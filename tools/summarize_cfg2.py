#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ of tools/prof_cfg2.sh (BASELINE.json configs[2]: 1080p60 NV12, main profile) into
profiles/r03_cfg2_nv12_1080p60_main.json: per-kernel calls / average duration of the lockstep run, FETCH_SIZE / WRITE_SIZE per launch
for the kernels whose launches all have one shape.  usage: summarize_cfg2.py <tag> [fps of the unprofiled two-instance run]"""
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1]
src = "gpurun_out/prof_%s" % tag
fps_line = [l for l in open(src + "/stats.log") if l.startswith("{")]
fps = json.loads(fps_line[-1])["value"] if fps_line else None


def short(n):
    return n.replace("void ", "").replace("h264::", "").split("(")[0]


# per kernel only its LARGEST launches: the lockstep steps of 32 pictures (the line's one-GOP-in-flight part launches single pictures)
rows = [r for r in csv.DictReader(open(max(glob.glob(src + "/stats/*/*kernel_trace.csv"), key=os.path.getmtime))) if "h264::" in r["Kernel_Name"]]
size = lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
big = collections.defaultdict(int)
for r in rows:
    big[short(r["Kernel_Name"])] = max(big[short(r["Kernel_Name"])], size(r))
dur = collections.defaultdict(list); shapes = collections.defaultdict(set)
for r in rows:
    k = short(r["Kernel_Name"])
    if size(r) == big[k]:
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        shapes[k].add((r["Grid_Size_X"], r["Grid_Size_Y"]))


def pmc(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(max(glob.glob(path), key=os.path.getmtime))):
        if r["Counter_Name"] == name and "h264::" in r["Kernel_Name"] and int(r["Grid_Size"]) == big[short(r["Kernel_Name"])]:
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024)
    return d


fetch, write = pmc(src + "/pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE"), pmc(src + "/pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
out = {"what": "BASELINE.json configs[2]: 1080p60 NV12, main profile, CAVLC (CABAC off), one instance x 32 GOPs in lockstep; rocprofv3 --kernel-trace --stats and "
               "separate --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --input nv12 --profile main --fps 60 --instances 1 --gops-in-flight 32 "
               "--no-cpu-baseline --no-plugin` (tools/prof_cfg2.sh %s, tools/summarize_cfg2.py)" % tag,
       "fps_under_profiler": fps, "fps_unprofiled_two_instances": float(sys.argv[2]) if len(sys.argv) > 2 else None, "kernels": {}}
for k in sorted(dur):
    e = {"calls": len(dur[k]), "avg_us": round(sum(dur[k]) / len(dur[k]) / 1e3, 1)}
    if len(shapes[k]) == 1 and k in fetch and k in write:
        f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
        e.update({"FETCH_SIZE_bytes_per_launch": int(f), "WRITE_SIZE_bytes_per_launch": int(w), "counter_GBps": round((f + w) / (sum(dur[k]) / len(dur[k])), 1)})
    out["kernels"][k] = e
out["note"] = ("FETCH_SIZE counts 64-byte requests at 64 B and larger ones at half their bytes (profiles/r03b_ubench_fetch.json): since the re-layout of "
               "k_tq's luma passes its luma reads are counted at half too (DESIGN.md section 6), so counter_GBps is a lower bound for every kernel; no ingest "
               "kernel exists - every kernel reads the interleaved chroma plane itself; bytes and GB/s are given for the kernels whose launches all have one shape")
json.dump(out, open("profiles/r03_cfg2_nv12_1080p60_main.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1)[:600])
for k in ("k_me<false>", "k_tq<false>", "k_intra_rows<false>", "k_deblock_pairs<false, false>"):
    print(k, out["kernels"].get(k))
