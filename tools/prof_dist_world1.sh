#!/bin/bash
# kernel-time difference between the plain step and the step through FlatGradSync + RCCL at world size 1 (same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pd0 /tmp/pd1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd0 -o b -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --no-cpu-baseline > /dev/null 2>&1
MPF_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29733 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd1 -o b -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(d + "/**/b_kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("/tmp/pd0"), load("/tmp/pd1")
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f"total kernel time over 9 steps: plain {ta/1e6:.2f} ms, dist {tb/1e6:.2f} ms, diff per step {(tb-ta)/9e6:.3f} ms")
d = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k) for k in set(a) | set(b)), reverse=True)
for dv, k in d[:12]:
    print(f"{dv/9e3:9.1f} us/step  calls {a.get(k,(0,0))[0]:5d} -> {b.get(k,(0,0))[0]:5d}  {k[:120]}")
for dv, k in d[-5:]:
    print(f"{dv/9e3:9.1f} us/step  calls {a.get(k,(0,0))[0]:5d} -> {b.get(k,(0,0))[0]:5d}  {k[:120]}")
PY
