#!/bin/bash
# Builds libmpformer_hip variants with gemm3_ws.h's WS_ABL timing ablations (results wrong, timing only) into /tmp and prints
# tools/bench_gemm3_ws.py's ws=1 column for each.  Run ON the GPU box:  bash tools/ab_ws_ablate.sh "0 1 2 4 8 16 3 7"
set -e
cd "$(dirname "$0")/../mp_former_amd/csrc"
OBJS=$(ls *.o | grep -v '^gemm3.o$')
for abl in ${1:-0 1 2 4 8 16}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DWS_ABL=$abl -c gemm3.hip -o /tmp/gemm3_abl$abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libmpf_abl$abl.so $OBJS /tmp/gemm3_abl$abl.o
  echo "== WS_ABL=$abl"
  MPF_LIB_PATH=/tmp/libmpf_abl$abl.so python ../../tools/bench_gemm3_ws.py ${2:-43008} 2>&1 | grep -E "value_proj|output_proj|linear1" | sed 's/.*| ws=1/ws=1/'
done
