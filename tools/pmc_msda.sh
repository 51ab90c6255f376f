#!/bin/bash
# HBM traffic of the MSDA backward kernels from rocprofv3 PMC counters (separate passes, as the guide
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 tools/bench_msda_breakdown.py init > /dev/null 2>&1
done
python3 - <<PY
import csv, collections, json, re, glob
res = collections.defaultdict(dict)
for d in glob.glob("/tmp/pmc_*"):
    rows = list(csv.DictReader(open(d + "/p_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        m = re.search(r"(msda_\w+|tile_scan_kernel)", r["Kernel_Name"])
        if m:
            agg[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        res[k][c] = sum(v) / len(v); res[k]["launches"] = len(v)
json.dump(res, open("gpurun_out/msda_bwd_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
