python -m pytest tests/test_rccl_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
  MPF_FORCE_DIST=1 MPF_GRAD_SYNC=ddp python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ddp  ', d['ms_per_step'])"
  MPF_FORCE_DIST=1 MPF_GRAD_SYNC=flat python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$i bench.py --gpus 1 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flat ', d['ms_per_step'])"
done
