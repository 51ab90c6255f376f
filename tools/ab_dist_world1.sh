#!/bin/bash
# world-size-1 overhead of the gradient exchange on ONE GPU: bench.py plain vs through a real RCCL process group
# (MPF_FORCE_DIST=1: flat buckets + hooks + all-reduce of 176 MB at world size 1), interleaved on the same box.
p() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['process_group'])"; }
for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 30 2>/dev/null | p plain
  MPF_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29533 + i)) python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 30 2>/dev/null | p flat3
  MPF_GRAD_SYNC=ddp MPF_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29633 + i)) python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 30 2>/dev/null | p ddp
done
