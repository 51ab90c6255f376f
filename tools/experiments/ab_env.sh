#!/bin/bash
# A/B of an environment switch on one GPU box, interleaved: tools/experiments/ab_env.sh "VAR=a" "VAR=b" [bench args]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  env $A python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A $A', d['ms_per_step'])"
  env $B python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B $B', d['ms_per_step'])"
done
