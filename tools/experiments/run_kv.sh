python -m pytest tests/test_attn_gpu.py tests/test_decoder_layer_gpu.py tests/test_head_gpu.py tests/test_rccl_gpu.py -x -q 2>&1 | tail -8
bash tools/experiments/ab_env.sh MPF_KV_BATCH=0 MPF_KV_BATCH=1
