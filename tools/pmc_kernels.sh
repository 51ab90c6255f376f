#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass; no tracing domains besides --kernel-trace) over a python
# script; prints per-kernel averages as JSON and writes gpurun_out/pmc_<tag>.json.
# usage: tools/pmc_kernels.sh <tag> <kernel-name-regex> <python script + args...>
tag=$1; shift
regex=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
CGROUPS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"
)
for c in "${CGROUPS[@]}"; do
  d=/tmp/pmc_${tag}_$i; rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 "$@" > $d.log 2>&1 || tail -3 $d.log
  i=$((i+1))
done
python3 - "$tag" "$regex" <<'PY'
import csv, collections, json, re, glob, sys
tag, regex = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(f"/tmp/pmc_{tag}_*")):
    fs = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        m = re.search(regex, r["Kernel_Name"])
        if m:
            agg[(m.group(0), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        v = v[len(v) // 3:]          # skip warm-up launches
        res[k][c] = sum(v) / len(v); res[k]["launches"] = len(v)
json.dump(res, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
