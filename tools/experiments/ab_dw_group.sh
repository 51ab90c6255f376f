python -m pytest tests/test_small_linear_gpu.py tests/test_attn_gpu.py tests/test_decoder_layer_gpu.py tests/test_resln_gpu.py tests/test_encoder_fused_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/ab1_tests.log
for i in 1 2 3; do
  MPF_OPTIONS=decoder_dw_group=0 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A dw_group=0', d['ms_per_step'])" >> gpurun_out/ab1.log
  python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B dw_group=1', d['ms_per_step'])" >> gpurun_out/ab1.log
done
cat gpurun_out/ab1_tests.log gpurun_out/ab1.log
