python -m pytest tests/test_head_gpu.py tests/test_loss_kernels_gpu.py tests/test_decoder_layer_gpu.py tests/test_rccl_gpu.py -x -q 2>&1 | tail -8
bash tools/experiments/ab_env.sh MPF_NATIVE_LOSS_TAIL=0 MPF_NATIVE_LOSS_TAIL=1
