#!/bin/bash
# HBM traffic of the TN kernels the default routing picks (fp16 x 2 form: gemm3_tn3_kernel for the 256-column products, gemm3_ws_kernel
# for N = 1024; bf16 x 3 form: the two-pass gemm3_tn2_kernel) on the three encoder shapes from rocprofv3 PMC counters: FETCH_SIZE and
# WRITE_SIZE in separate passes, calibrated on kernels that move a known 512 MiB (tools/ubench/fetch_calib.hip), as
# tools/pmc_msda.sh does.  Run on the GPU box from the repo root; writes gpurun_out/${TAG}_gemm3_traffic.json.
TAG=${1:-r05}
export PMC_TAG=$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib || { echo "cannot build fetch_calib" >&2; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pgc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pgc_$c -o p -- tools/ubench/fetch_calib > /dev/null 2>&1
  for s in 0 1 2; do
    rm -rf /tmp/pg${s}_$c
    PROBE_SHAPE=$s rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg${s}_$c -o p -- python3 tools/probe_f16x2.py > /dev/null 2>&1
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
            v = v[len(v) // 3:]
            res[k][c] = sum(v) / len(v)
    return res
cal = load("pgc", r"calib_\w+")
EXPECT = 512 * 1024 * 1024
if "calib_stream16" not in cal or "calib_write16" not in cal:
    raise SystemExit("calibration kernels missing: refusing to write uncalibrated traffic")
fr = EXPECT / (cal["calib_stream16"]["FETCH_SIZE"] * 1024.0)
fw = EXPECT / (cal["calib_write16"]["WRITE_SIZE"] * 1024.0)
M = 43008
out = {"calibration_bytes_per_counter_byte": {"read (16-byte lanes, streaming)": fr, "write": fw}, "M": M, "shapes": {}}
for s, (n, k) in enumerate(((256, 256), (1024, 256), (256, 1024))):
    r = load(f"pg{s}", r"gemm3_tn3_kernel|gemm3_ws_kernel<0, false, false, false>|gemm3_tn2_kernel<\d+, false, false>")
    e = {}
    want16 = "gemm3_ws_kernel" if n == 1024 else "gemm3_tn3_kernel"     # (the probe also runs the input-gradient GEMM of the shape: the other kernel)
    rows = "128" if n == 1024 else "96"
    for kern, v in r.items():
        if kern.startswith("gemm3_tn2_kernel"):
            if not kern.startswith("gemm3_tn2_kernel<" + rows):
                continue
            form, planes = "bf16x3", 6
        elif kern.startswith(want16):
            form, planes = "fp16x2", 4
        else:
            continue
        alg = 4.0 * (M * k + M * n) + planes * n * k + (4.0 * n)
        rd, wr = v.get("FETCH_SIZE", 0.0) * 1024.0 * fr, v.get("WRITE_SIZE", 0.0) * 1024.0 * fw
        e[form] = {"kernel": kern, "read_bytes": round(rd), "write_bytes": round(wr), "algorithmic_bytes": round(alg),
                   "traffic_over_algorithmic": round((rd + wr) / alg, 3)}
    out["shapes"][f"N={n},K={k}"] = e
json.dump(out, open("gpurun_out/%s_gemm3_traffic.json" % os.environ.get("PMC_TAG", "r05"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
