#!/bin/bash
# Regenerate mp_former_amd/miopen_db (run on the GPU box from the repo root, ~4 min):
# MIOpen times its solvers for every convolution of the training step and records the winners.
#   gpurun -- 'bash tools/miopen_find.sh'   then copy gpurun_out/miopen_db/*.txt into mp_former_amd/miopen_db/
mkdir -p gpurun_out/miopen_db
cp mp_former_amd/miopen_db/*.txt gpurun_out/miopen_db/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
MPF_CONV_FIND=1 python bench.py --steps 5 --warmup 6 --no-cpu-baseline | tail -1 | cut -c1-160
ls -la gpurun_out/miopen_db
