#!/usr/bin/env python3
"""How the two lanes of a context share the GPU: from a `rocprofv3 --kernel-trace --output-format csv` of a bench.py run,
per hardware queue the time its kernels run and the gaps between them, and the share of the span during which 0 / 1 / 2+
kernels are running.  usage: trace_overlap.py <kernel_trace.csv> [skip_fraction]  (the first part of the trace -- scene
build, warm-up -- is skipped: default the first half)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
count_runs = [e[0] for e in ev if "k_traverse<true" in e[3]]   # bench.py's counting frames come after the timed ones
frames = [e for e in ev if "k_shade" in e[3] and (not count_runs or e[0] < count_runs[0])]
t0, t1 = frames[0][0], frames[-1][1]
cut = t0 + int((t1 - t0) * skip)
ev = [e for e in ev if e[0] >= cut and e[1] <= t1]
t0, t1 = ev[0][0], max(e[1] for e in ev)
span = t1 - t0
print("span %.3f ms, %d kernels" % (span / 1e6, len(ev)))
byq = collections.defaultdict(list)
for e in ev: byq[e[2]].append(e)
for q, es in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in es)
    gaps = [max(0, es[i + 1][0] - es[i][1]) for i in range(len(es) - 1)]
    gaps.sort()
    print("queue %s: %d kernels, running %.1f %% of the span, mean kernel %.1f us, gap between consecutive kernels: mean %.1f us median %.1f us p90 %.1f us" % (
        q, len(es), 100.0 * busy / span, busy / len(es) / 1e3, sum(gaps) / max(1, len(gaps)) / 1e3, gaps[len(gaps) // 2] / 1e3 if gaps else 0, gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0))
pts = []
for e in ev: pts.append((e[0], 1)); pts.append((e[1], -1))
pts.sort()
level, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[min(level, 3)] += t - last; last = t; level += d
print("kernels running at once: " + ", ".join("%d%s: %.1f %%" % (k, "+" if k == 3 else "", 100.0 * v / span) for k, v in sorted(hist.items())))
names = collections.Counter()
for e in ev: names[e[3].split("(")[0][:60]] += e[1] - e[0]
for n, v in names.most_common(6): print("  %-60s %.2f ms" % (n, v / 1e6))
