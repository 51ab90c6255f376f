#!/bin/bash
# SQ counters of the weight-stationary GEMM kernel (two passes of 8 SQ slots).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/pmc_ws -o p -- python3 tools/bench_gemm3_ws.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pmc_wsb -o p -- python3 tools/bench_gemm3_ws.py > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("/tmp/pmc_ws", "/tmp/pmc_wsb"):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemm3_ws" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:70], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, g, c), v in sorted(agg.items()):
        print(f"{k:72s} grid {g:8s} {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
