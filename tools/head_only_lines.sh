#!/bin/bash
# BASELINE configs B, C, D, E as head-only bench lines + rocprofv3 kernel statistics of the same command.
#   bash tools/head_only_lines.sh TAG     ->  gpurun_out/TAG/cfg{B,C,D,E}_head_only.json, cfg*_kernel_stats.csv
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
for W in B C D E; do
  python3 bench.py --workload $W --head-only --steps 10 --warmup 3 --profile-steps 3 --trained-steps 5 > gpurun_out/$TAG/cfg${W}_head_only.json 2> gpurun_out/$TAG/cfg${W}_head_only.err
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$W -o p -- python3 bench.py --workload $W --head-only --steps 6 --warmup 2 --profile-steps 0 --trained-steps 0 > /dev/null 2>&1
  f=$(find /tmp/prof_$W -name "p_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/$TAG/cfg${W}_kernel_stats.csv
  python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/$TAG/cfg${W}_head_only.json"))
    r = d["roofline"]
    print("$W", d["value"], "img/s", d["ms_per_step"], "ms/step | gemm3", r["ms_per_step"], "| " + " | ".join(f'{e["kernel"][:22]} {e.get("avg_us", e.get("ms_per_step"))} frac {e.get("frac", e.get("hbm_frac"))}' for e in r["also"][5:10]))
except Exception as e:
    print("$W failed:", e, open("gpurun_out/$TAG/cfg${W}_head_only.err").read()[-1500:])
PY
done
