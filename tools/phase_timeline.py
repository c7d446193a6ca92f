#!/usr/bin/env python3
"""Phase between the instances of a rocprofv3 kernel trace of bench.py: for every lockstep k_me launch, which kernels of the OTHER
queue run during it (share of its duration), and the offset between the two queues' k_me starts.
usage: phase_timeline.py <dir with stats/*/..._kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/stats/*/*kernel_trace.csv")[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Grid_Size_Y"] not in ("1",) or "k_intra_rows" in r["Kernel_Name"]]
def nm(r): return r["Kernel_Name"].replace("void h264::", "").split("(")[0].split("<")[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], nm(r)) for r in rows if r["Kernel_Name"].startswith(("void h264::", "h264::"))]
qs = collections.Counter(q for _, _, q, n in ev if n == "k_me")
main = [q for q, _ in qs.most_common(2)]
share = collections.defaultdict(float); tot = 0.0
for s, e, q, n in ev:
    if n != "k_me" or q not in main: continue
    tot += e - s
    for s2, e2, q2, n2 in ev:
        if q2 != q and q2 in main and s2 < e and e2 > s:
            share[n2] += min(e, e2) - max(s, s2)
print("during a k_me of one instance the other instance runs:", {k: round(v / tot, 3) for k, v in sorted(share.items(), key=lambda kv: -kv[1])})
me = {q: sorted(s for s, e, qq, n in ev if qq == q and n == "k_me") for q in main}
import bisect
offs = []
for s in me[main[0]]:
    i = bisect.bisect_left(me[main[1]], s)
    c = [abs(me[main[1]][j] - s) for j in (i - 1, i) if 0 <= j < len(me[main[1]])]
    if c: offs.append(min(c) / 1e3)
offs.sort()
print("offset between the instances' k_me starts (us): median %.0f, quartiles %.0f / %.0f" % (offs[len(offs) // 2], offs[len(offs) // 4], offs[3 * len(offs) // 4]))
per = [(b - a) / 1e3 for a, b in zip(me[main[0]], me[main[0]][1:])]
per.sort(); print("k_me period of one instance (us): median %.0f" % per[len(per) // 2])
if len(sys.argv) > 2:   # a stretch of the timeline, both queues side by side
    t0 = me[main[0]][len(me[main[0]]) // 2]
    for s, e, q, n in sorted(ev):
        if q in main and t0 <= s < t0 + int(float(sys.argv[2]) * 1e6):
            print("%s%-18s %8.1f -> %8.1f (%6.1f)" % ("" if q == main[0] else " " * 44, n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
