#!/bin/bash
# HBM traffic of the MSDA kernels at config B, N = 2 from rocprofv3 PMC counters (separate passes, as the guide
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass), with FETCH_SIZE / WRITE_SIZE calibrated on kernels that
# move a known byte count in the same access shapes (tools/ubench/fetch_calib.hip).  Run on the GPU box from the repo
# root; writes gpurun_out/${TAG}_msda_bwd_pmc_configB_N2.json (TAG defaults to r04; copy the file to profiles/).
TAG=${1:-r05}
export PMC_TAG=$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# the calibration binary is built here (it is not tracked): without it every factor would silently be 1.0 and the
# "corrected" traffic off by ~2x
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib || { echo "pmc_msda: cannot build fetch_calib" >&2; exit 1; }
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  t=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pmcm_$t /tmp/pmcc_$t
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcm_$t -o p -- python3 tools/bench_msda_breakdown.py init > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcc_$t -o p -- tools/ubench/fetch_calib > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections, glob, hashlib, json, re
def load(prefix, regex):
    res = collections.defaultdict(dict)
    for d in glob.glob(f"/tmp/{prefix}_*"):
        fs = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            m = re.search(regex, r["Kernel_Name"])
            if m:
                agg[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            v = v[len(v) // 3:]
            res[k][c] = sum(v) / len(v); res[k]["launches"] = len(v)
    return res
cal = load("pmcc", r"(calib_\w+)")
msda = load("pmcm", r"(msda_\w+_kernel)")
EXPECT = 512 * 1024 * 1024
factors = {}
need = {"calib_stream16": "FETCH_SIZE", "calib_rows16": "FETCH_SIZE", "calib_rows8": "FETCH_SIZE", "calib_write16": "WRITE_SIZE"}
missing = [k for k, c in need.items() if k not in cal or c not in cal[k] or cal[k][c] <= 0]
if missing:
    raise SystemExit(f"pmc_msda: calibration kernels / counters missing ({missing}): refusing to write uncalibrated traffic")
for k, v in cal.items():
    if k == "calib_write16":
        factors[k] = EXPECT / (v["WRITE_SIZE"] * 1024.0)
    else:
        factors[k] = EXPECT / (v["FETCH_SIZE"] * 1024.0)
# dominant read shape per kernel: 128-B rows fetched as 8 lanes x 16 B (forward, push) or 16 lanes x 8 B (pull)
shape = {"msda_fwd_block_kernel": "calib_rows16", 
         "msda_bwd_bin_kernel": "calib_stream16", "msda_bwd_tile_kernel": "calib_rows16"}
import os
out = {"config": "B (1024x1024: S = 21504), N = 2, init-like offsets (tools/bench_msda_breakdown.py init)",
       "calibration": "tools/ubench/fetch_calib.hip built and run in this call: kernels that move a known 512 MiB in the access shapes of the MSDA kernels",
       "source_sha256": hashlib.sha256(open("mp_former_amd/csrc/msda_block.hip", "rb").read()).hexdigest(),
       "calibration_bytes_per_counter_byte": factors, "kernels": {}}
total = 0.0
for k, v in msda.items():
    f = factors[shape.get(k, "calib_rows16")]
    fw = factors["calib_write16"]
    rd = v.get("FETCH_SIZE", 0.0) * 1024.0 * f
    wr = v.get("WRITE_SIZE", 0.0) * 1024.0 * fw
    e = dict(v)
    e.update({"read_bytes_corrected": rd, "write_bytes_corrected": wr, "read_factor": f, "write_factor": fw})
    if "TCC_HIT_sum" in v:
        e["l2_hit_rate"] = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1.0)
    if "TCP_TCC_READ_REQ_sum" in v:
        e["l1_hit_rate"] = 1.0 - v["TCP_TCC_READ_REQ_sum"] / max(v["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0)
    out["kernels"][k] = e
    if "bwd" in k:
        total += rd + wr
out["hbm_bytes_per_call"] = round(total)
out["algorithmic_bytes_per_call"] = 1344 * 4 * 21504 * 2
out["note"] = ("FETCH_SIZE x read_factor + WRITE_SIZE x write_factor of the backward kernels (bin + tile + spill3; push + pull before round 4); factors from tools/ubench/fetch_calib "
               "(known 512 MiB per launch in the kernels' access shapes)")
json.dump(out, open("gpurun_out/%s_msda_bwd_pmc_configB_N2.json" % os.environ.get("PMC_TAG", "r04"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
