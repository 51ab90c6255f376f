# same-box A/B: the tree under ab/base (git archive of a baseline commit + its built library) against the working tree
p() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  (cd ab/base && python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 30 2>/dev/null | p base)
  python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 30 2>/dev/null | p work
done
