#!/usr/bin/env python3
"""Durations of the P-step kernels of a rocprofv3 kernel trace (lockstep launches only), split by whether the launch overlaps an IDR row
wavefront (k_intra_rows of the other instance): what the IDR step costs the instance running beside it.
usage: idr_overlap.py <dir with stats/*/..._kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/*/*kernel_trace.csv")[-1]
rows = list(csv.DictReader(open(f)))
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_intra_rows" in r["Kernel_Name"] and r["Grid_Size_Y"] != "1"]
print("IDR launches", len(iv), "avg %.1f us" % (sum(b - a for a, b in iv) / max(1, len(iv)) / 1e3))
for name in ("k_me", "k_tq<", "k_mvpred", "k_cavlc<false", "k_cavlc<true", "k_deblock_pairs<false", "k_deblock_rows<false", "k_pintra_rows", "k_skip_scan", "k_bit_scan"):
    a, b = [], []
    for r in rows:
        if name in r["Kernel_Name"] and r["Grid_Size_Y"] != "1":
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            (a if any(s < y and e > x for x, y in iv) else b).append((e - s) / 1e3)
    if b:
        print("%-24s beside IDR n=%3d avg %7.1f | otherwise n=%4d avg %7.1f us" % (name, len(a), sum(a) / max(1, len(a)), len(b), sum(b) / len(b)))
