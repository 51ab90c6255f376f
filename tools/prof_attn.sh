#!/bin/bash
# per-launch durations of the attention kernels in one steady step of bench.py, grouped by (kernel, grid size)
# usage: tools/prof_attn.sh [MPF_OPTIONS value]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pa
MPF_OPTIONS=$1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pa -o b -- python3 bench.py --steps 4 --warmup 3 --profile-steps 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pa/**/b_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "attn_" in r["Kernel_Name"]]
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("((")[0].split("(float")[0].split("(__hip")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in agg.items():
    v = v[len(v) // 2:]
    print(f"{sum(v)/len(v)/1e3:8.1f} us x{len(v):3d}  {k[0][:48]:48s} grid {k[1]}x{k[2]}x{k[3]} wg {k[4]}")
    tot += sum(v)
print("total attention us per step:", tot / 1e3 / 3.5)
PY
