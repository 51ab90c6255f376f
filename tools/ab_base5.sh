# same-box A/B, five interleaved pairs of 60 steps: ab/base (git archive of a baseline commit + its library) vs the working tree
p() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3 4 5; do
  (cd ab/base && python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 60 2>/dev/null | p base)
  python3 bench.py --no-cpu-baseline --profile-steps 0 --steps 60 2>/dev/null | p work
done | tee /tmp/ab5.txt
python3 - <<'PY'
import statistics as st
b=[float(l.split()[1]) for l in open('/tmp/ab5.txt') if l.startswith('base')]
w=[float(l.split()[1]) for l in open('/tmp/ab5.txt') if l.startswith('work')]
print("median base %.2f work %.2f  (min %.2f / %.2f)" % (st.median(b), st.median(w), min(b), min(w)))
PY
