#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace into runs of consecutive launches of the same kernel: name, count, avg / min us.
usage: trace_runs.py <dir with *kernel_trace.csv> [min_run_length]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel trace under " + sys.argv[1])
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
runs, prev = [], None
for r in rows:
    n = r["Kernel_Name"][:90]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if prev and prev[0] == n:
        prev[1].append(d)
    else:
        prev = [n, [d]]
        runs.append(prev)
for n, ds in runs:
    if len(ds) >= minlen:
        print(f"{n:90s} x{len(ds):3d} avg {sum(ds) / len(ds):7.2f} us min {min(ds):7.2f}")
