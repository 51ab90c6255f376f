#!/bin/bash
# HBM traffic of the encoder FFN products with the ReLU gate as saved activation / as bit mask (tools/probe_gate_bits.py) from
# rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in separate passes, calibrated on kernels that move a known 512 MiB
# (tools/ubench/fetch_calib.hip), as tools/pmc_gemm3_traffic.sh does.  Run on the GPU box from the repo root; writes
# gpurun_out/${TAG}_gate_bits_traffic.json.
TAG=${1:-r04}
export PMC_TAG=$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib || { echo "cannot build fetch_calib" >&2; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pbc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pbc_$c -o p -- tools/ubench/fetch_calib > /dev/null 2>&1
  for s in 0 1 2 3; do
    rm -rf /tmp/pb${s}_$c
    PROBE_GATE=$s rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pb${s}_$c -o p -- python3 tools/probe_gate_bits.py > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, collections, glob, json, os, re
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
                agg[(m.group(0), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            v = v[-8:]                       # the last launches = the probed variant (the set-up runs linear1 once before)
            res[k][c] = sum(v) / len(v)
    return res
cal = load("pbc", r"calib_\w+")
EXPECT = 512 * 1024 * 1024
if "calib_stream16" not in cal or "calib_write16" not in cal:
    raise SystemExit("calibration kernels missing: refusing to write uncalibrated traffic")
fr = EXPECT / (cal["calib_stream16"]["FETCH_SIZE"] * 1024.0)
fw = EXPECT / (cal["calib_write16"]["WRITE_SIZE"] * 1024.0)
M, C, F = 43008, 256, 1024
names = ["linear1 + ReLU", "linear1 + ReLU + gate-mask output", "dh, activation as gate", "dh, bit-mask gate"]
alg = [4.0 * (M * C + M * F) + 4 * F * C, 4.0 * (M * C + M * F) + 4 * F * C + M * F / 8,
       4.0 * (M * C + 2 * M * F) + 4 * F * C, 4.0 * (M * C + M * F) + 4 * F * C + M * F / 8]
out = {"calibration_bytes_per_counter_byte": {"read (16-byte lanes, streaming)": fr, "write": fw}, "M": M, "variants": {}}
for s in range(4):
    r = load(f"pb{s}", r"gemm3_tn2_kernel<128, false, true>")
    for kern, v in r.items():
        rd, wr = v.get("FETCH_SIZE", 0.0) * 1024.0 * fr, v.get("WRITE_SIZE", 0.0) * 1024.0 * fw
        out["variants"][names[s]] = {"kernel": kern, "read_bytes": round(rd), "write_bytes": round(wr), "algorithmic_bytes": round(alg[s]),
                                     "traffic_over_algorithmic": round((rd + wr) / alg[s], 3)}
json.dump(out, open("gpurun_out/%s_gate_bits_traffic.json" % os.environ.get("PMC_TAG", "r04"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
