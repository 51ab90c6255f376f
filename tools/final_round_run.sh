cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/t_all_${1:-run}.log
bash tools/profile_bench.sh ${1:-run} > gpurun_out/prof_${1:-run}_summary.txt 2>&1
python bench.py > gpurun_out/bench_${1:-run}.json 2> gpurun_out/bench_${1:-run}.err
tail -3 gpurun_out/t_all_${1:-run}.log; head -16 gpurun_out/prof_${1:-run}_summary.txt; cut -c1-400 gpurun_out/bench_${1:-run}.json
