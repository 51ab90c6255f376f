#!/bin/bash
# tools/bench_gemm3_ws.py with gemm3.hip rebuilt under extra -D flags (one variant per argument), on the GPU box:
#   bash tools/ab_ws_def.sh "-DWS_STAGGER=0" "-DWS_STAGGER=30"
set -e
cd "$(dirname "$0")/../mp_former_amd/csrc"
OBJS=$(ls *.o | grep -v '^gemm3.o$')
i=0
for defs in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $defs -c gemm3.hip -o /tmp/gemm3_v$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libmpf_v$i.so $OBJS /tmp/gemm3_v$i.o
  echo "== $defs"
  MPF_LIB_PATH=/tmp/libmpf_v$i.so python ../../tools/bench_gemm3_ws.py 43008 2>&1 | grep -E "value_proj|output_proj|linear1|dh" | sed 's/.*| ws=1/ws=1/'
done
