timeout 600 python -m pytest tests/test_gemm3_gpu.py tests/test_encoder_fused_gpu.py -q -x 2>&1 | tail -2
for k in 1 2 3; do
python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'], d['roofline']['ms_per_step'])"
done
