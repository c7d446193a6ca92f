#!/usr/bin/env python3
"""k_intra_rows launch durations of a rocprofv3 kernel trace, split by the number of pictures in the launch (grid.y)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/stats/*/*kernel_trace.csv")[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_intra_rows" in n:
        d[r["Grid_Size_Y"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print("k_intra_rows grid.y=%s calls %d avg %.1f us min %.1f max %.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
