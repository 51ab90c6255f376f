# same-box A/B of an environment switch: tools/ab_env.sh VAR [pairs] -> bench.py --steps 20 with VAR=0 / VAR=1, interleaved
var=$1; pairs=${2:-3}
for r in $(seq $pairs); do
  for m in 0 1; do
    env $var=$m python bench.py --steps 20 --warmup 8 --trained-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$m', d['ms_per_step'], d['roofline']['ms_per_step'])"
  done
done
