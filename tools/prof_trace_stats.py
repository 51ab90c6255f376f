#!/usr/bin/env python3
"""Steady-state kernel stats from a rocprofv3 kernel trace of bench.py: drops everything before the
first MSDA forward launch of the first TIMED step (warm-up, MIOpen find, allocator growth).
usage: prof_trace_stats.py <kernel_trace.csv> <warmup_steps> <out.csv>"""
import csv, sys, collections
trace, warmup, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
fwd = [s for s, e, n in rows if "msda_fwd" in n]
per_step = 6
t0 = fwd[warmup * per_step]
steps = (len(fwd) - warmup * per_step) // per_step
agg = collections.OrderedDict()
for s, e, n in rows:
    if s < t0:
        continue
    d = agg.setdefault(n, [0, 0, 10**18, 0])
    d[0] += 1; d[1] += e - s; d[2] = min(d[2], e - s); d[3] = max(d[3], e - s)
tot = sum(d[1] for d in agg.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for n, d in sorted(agg.items(), key=lambda x: -x[1][1]):
        w.writerow([n, d[0], d[1], d[1] / d[0], 100.0 * d[1] / tot, d[2], d[3], 0])
span = max(e for s, e, n in rows) - t0
print(f"timed steps={steps} kernels busy={tot/1e6/steps:.2f} ms/step, span={span/1e6/steps:.2f} ms/step, launches/step={sum(d[0] for d in agg.values())/steps:.0f}")
