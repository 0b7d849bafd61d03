#!/usr/bin/env python3
"""Per-bounce durations of the traversal launches from a rocprofv3 --kernel-trace csv (one frame at a time, so that the
dispatch order is the bounce order).  usage: python tools/trace_bounces.py <kernel_trace.csv> <bounces + 1>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
trav = [r for r in rows if "k_traverse<false" in r["Kernel_Name"]]
shade = [r for r in rows if "k_shade" in r["Kernel_Name"]]
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
n = len(trav) // per * per
skip = n // 2 // per * per  # second half: warmed up
by = collections.defaultdict(list)
for i, r in enumerate(trav[skip:n]): by[i % per].append(dur(r))
bs = collections.defaultdict(list)
for i, r in enumerate(shade[skip:n]): bs[i % per].append(dur(r))
for b in range(per):
    print("bounce %d: traverse %.1f us (min %.1f max %.1f, grid %s)  shade %.1f us" % (b, sum(by[b]) / len(by[b]), min(by[b]), max(by[b]), trav[skip + b].get("Grid_Size", "?"), sum(bs[b]) / max(1, len(bs[b]))))
# idle time between consecutive kernels
allk = rows[len(rows) // 2:]
gap = sum(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(allk, allk[1:])) / 1e3
span = (int(allk[-1]["End_Timestamp"]) - int(allk[0]["Start_Timestamp"])) / 1e3
print("second half: span %.0f us, idle between kernels %.0f us (%.1f %%)" % (span, gap, 100 * gap / span))
