#!/bin/bash
# full GPU suite + default bench line + rocprofv3 kernel statistics of the same command (tag = $1)
tag=${1:-run}
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_tests.log
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
bash tools/profile_bench.sh $tag > gpurun_out/${tag}_prof.log 2>&1
cat gpurun_out/${tag}_tests.log; python -c "
import json;d=json.load(open('gpurun_out/${tag}_bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'])"
head -20 gpurun_out/${tag}_prof.log
