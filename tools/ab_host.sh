# same-box A/B of the launch thread: tools/host_step_time.py in ab/base (git archive of HEAD + the built library) against the working tree
for i in 1 2 3; do
  (cd ab/base && python3 tools/host_step_time.py 2>&1 | tail -1 | sed 's/^/base: /')
  python3 tools/host_step_time.py 2>&1 | tail -1 | sed 's/^/work: /'
done
