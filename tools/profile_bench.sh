#!/bin/bash
# Steady-state kernel statistics of bench.py (run on the GPU box from the repo root):
#   tools/profile_bench.sh <tag>   ->  gpurun_out/prof_<tag>/steady_kernel_stats.csv (+ summary on stdout)
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --trained-steps 0 --head-steps 0 --no-cpu-baseline > gpurun_out/bench_prof_$tag.log 2>&1
mkdir -p gpurun_out/prof_$tag
trace=$(find /tmp/prof_$tag -name 'bench_kernel_trace.csv' | head -1)
python3 tools/prof_trace_stats.py "$trace" 3 gpurun_out/prof_$tag/steady_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/prof_$tag/steady_kernel_stats.csv 6
python3 tools/prof_gaps.py "$trace" 3 20 | head -3 | tee gpurun_out/prof_$tag/gaps.txt
python3 tools/prof_sequence.py "$trace" 3 2 gpurun_out/prof_$tag/sequence_step2.txt
tail -1 gpurun_out/bench_prof_$tag.log | cut -c1-160
