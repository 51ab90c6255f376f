#!/bin/bash
# SQ counters of the two-pass TN kernel in both forms (bf16 x 3 / six products vs fp16 x 2 / three products) on the three encoder
# shapes of tools/probe_f16x2.py (two passes of 8 SQ counters; --kernel-trace only).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_h2a /tmp/pmc_h2b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/pmc_h2a -o p -- python3 tools/probe_f16x2.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pmc_h2b -o p -- python3 tools/probe_f16x2.py > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("/tmp/pmc_h2a", "/tmp/pmc_h2b"):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "gemm3_tn2_kernel" in k:
            form = "fp16x2 (3 products)" if k.rstrip(">(anoymus:G3) ").endswith("true") or ", true>" in k else "bf16x3 (6 products)"
            rows = "96" if "<96" in k else "128"
            agg[(form, rows, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (form, rows, grid, c), v in sorted(agg.items()):
        print(f"{form:22s} rows {rows:4s} grid {grid:>8s} {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
