#!/bin/bash
# A/B of two builds of the library on one GPU box, interleaved: tools/experiments/ab_lib2.sh <libA.so> <libB.so>
A=$1; B=$2
for i in 1 2 3; do
  MPF_LIB_PATH=$A python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A', d['ms_per_step'], [round(x['ms_per_step'],3) for x in d['roofline']['also'] if 'gemm3' in x['kernel']])"
  MPF_LIB_PATH=$B python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B', d['ms_per_step'], [round(x['ms_per_step'],3) for x in d['roofline']['also'] if 'gemm3' in x['kernel']])"
done
