p() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  (cd ab/r02 && python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 30 2>/dev/null | p r02)
  python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 30 2>/dev/null | p r03
done
