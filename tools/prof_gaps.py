#!/usr/bin/env python3
"""GPU idle gaps in the steady-state part of a rocprofv3 kernel trace of bench.py: where the device
waits for the host.  usage: prof_gaps.py <kernel_trace.csv> <warmup_steps> [min_gap_us]"""
import csv, sys, collections
trace, warmup = sys.argv[1], int(sys.argv[2])
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
fwd = [s for s, e, n in rows if "msda_fwd" in n]
t0 = fwd[warmup * 6]
steps = (len(fwd) - warmup * 6) // 6
rows = [r for r in rows if r[0] >= t0]
def short(n):
    n = n.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    return n[:70]
gaps = []
end = rows[0][1]
prev = rows[0][2]
hist = collections.Counter()
tot_idle = 0
for s, e, n in rows[1:]:
    g = (s - end) / 1e3
    if g > 0:
        tot_idle += g
        b = 5 if g < 5 else 20 if g < 20 else 100 if g < 100 else 1000 if g < 1000 else 10**6
        hist[b] += g
    if g >= min_gap:
        gaps.append((g, prev, n, (s - t0) / 1e6))
    if e > end:
        end, prev = e, n
print(f"steps={steps} idle={tot_idle/1e3/steps:.2f} ms/step; idle by gap size (ms/step): " +
      ", ".join(f"<{b}us:{v/1e3/steps:.2f}" for b, v in sorted(hist.items())))
agg = collections.defaultdict(lambda: [0, 0.0])
for g, a, b, t in gaps:
    k = (short(a), short(b))
    agg[k][0] += 1; agg[k][1] += g
print(f"gaps >= {min_gap} us grouped by (kernel before -> kernel after), ms/step:")
for k, (c, g) in sorted(agg.items(), key=lambda x: -x[1][1])[:25]:
    print(f"  {g/1e3/steps:6.2f} ms  x{c/steps:5.1f}  {k[0]}  ->  {k[1]}")
