#!/bin/bash
# A/B of an mpf_set_option switch: tools/experiments/ab_opt.sh "key=a" "key=b"
for i in 1 2 3; do
  MPF_OPTIONS=$1 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A $1', d['ms_per_step'])"
  MPF_OPTIONS=$2 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B $2', d['ms_per_step'])"
done
