timeout 600 python -m pytest tests/test_gemm3_gpu.py -q -x 2>&1 | tail -5
timeout 300 python tools/bench_gemm3_ws.py 2>&1 | grep -v amdgpu.ids | sed 's/ | ws=1.*//'
