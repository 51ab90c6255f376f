#!/bin/bash
# per-kernel time difference between two source trees' bench.py on the same box: tools/prof_diff_trees.sh <treeA> <treeB>
A=$1; B=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in A B; do
  d=$([ $t = A ] && echo $A || echo $B)
  rm -rf /tmp/pt$t
  (cd $d && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt$t -o b -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --no-cpu-baseline > /dev/null 2>&1)
done
python3 - <<'PY'
import csv, glob, re
def load(d):
    f = glob.glob(d + "/**/b_kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")[:90]
        c, t = out.get(n, (0, 0))
        out[n] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
    return out
a, b = load("/tmp/ptA"), load("/tmp/ptB")
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f"total kernel time per step: A {ta/9e6:.3f} ms, B {tb/9e6:.3f} ms, diff {(tb-ta)/9e6:.3f} ms")
d = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k) for k in set(a) | set(b)), reverse=True)
for dv, k in d[:22]:
    print(f"{dv/9e3:9.1f} us/step  calls {a.get(k,(0,0))[0]/9:6.1f} -> {b.get(k,(0,0))[0]/9:6.1f}  {k}")
print("...")
for dv, k in d[-14:]:
    print(f"{dv/9e3:9.1f} us/step  calls {a.get(k,(0,0))[0]/9:6.1f} -> {b.get(k,(0,0))[0]/9:6.1f}  {k}")
PY
