#!/usr/bin/env python3
"""Ordered kernel sequence of ONE steady-state step from a rocprofv3 kernel trace of bench.py.
usage: prof_sequence.py <kernel_trace.csv> <warmup_steps> <step_index> <out.txt>"""
import csv, sys
trace, warmup, step, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step starts at the first kernel after the previous step's optimizer kernel (opt_adamw_kernel)
ends = [i for i, (s, e, n) in enumerate(rows) if "opt_adamw_kernel" in n]
lo, hi = ends[warmup + step - 1] + 1, ends[warmup + step] + 1
with open(out, "w") as f:
    t0 = rows[lo][0]
    prev_end = t0
    for s, e, n in rows[lo:hi]:
        f.write(f"{(s - t0) / 1e3:10.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {n[:150]}\n")
        prev_end = e
print(f"step {step}: {hi - lo} launches, span {(rows[hi - 1][1] - rows[lo][0]) / 1e6:.2f} ms -> {out}")
