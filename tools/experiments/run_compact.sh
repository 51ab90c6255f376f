python -m pytest tests/test_head_gpu.py tests/test_loss_kernels_gpu.py tests/test_lsa_gpu.py tests/test_rccl_gpu.py -x -q 2>&1 | tail -8
bash tools/experiments/ab_env.sh MPF_COMPACT_MASK_GRAD=0 MPF_COMPACT_MASK_GRAD=1
