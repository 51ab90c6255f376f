#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in one csrc/*.hip file (cross-compiles to gfx950 in a temp dir)."""
import os
import re
import subprocess
import sys
import tempfile

src = os.path.abspath(sys.argv[1])
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                           "-Wno-unused-function", "-c", src, "-o", os.path.join(d, "x.o"), "-save-temps=obj",
                           "-I", os.path.dirname(src)] + sys.argv[2:], cwd=d)
    s = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    txt = open(os.path.join(d, s)).read()
    if "--keep" in os.environ.get("KR", ""):
        open("/tmp/last_kernel.s", "w").write(txt)
md = txt[txt.index("amdhsa.kernels"):]
for blk in md.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)  # noqa: E731
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:90]:90s} agpr {blk.split(chr(10))[0].strip(': '):>3s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} "
          f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s} spill {g('vgpr_spill_count')}")
