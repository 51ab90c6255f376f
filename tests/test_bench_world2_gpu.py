"""bench.py's N > 1 code path inside a one-GPU lease: two ranks started by torch.distributed.run exactly as the driver
starts them, both mapped to cuda:0 and talking over gloo (tests/_bench_world2_child.py patches exactly those two things
around an unchanged bench.py: RCCL refuses two ranks on one device).  Everything else is the real thing — rank-dependent synthetic batches (different
numbers of ground-truth masks per rank, hence different Qtot), the three-bucket gradient exchange whose first two buckets
are launched from tensor hooks while the backbone back-propagates (a gradient arriving after its bucket's launch raises), the device-side num_masks all-reduce, ClipAdamW on the bucket views,
barrier + max-over-ranks timing, one JSON line from rank 0.  The ranks' parameters must still be identical after the
averaged updates.  RCCL itself is covered at world size 1 by tests/test_rccl_gpu.py; no scaling number follows from this."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_two_ranks(grad_sync, extra_args=(), **extra_env):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MPF_FORCE_DIST"):
        env.pop(k, None)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MPF_CHECK_SYNC="1", MPF_GRAD_SYNC=grad_sync, **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_bench_world2_child.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--profile-steps", "1", "--trained-steps", "1", "--size", "512", "--no-cpu-baseline"] + list(extra_args)
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1400)


@pytest.mark.timeout(1500)
def test_late_gradient_after_an_early_launch_raises():
    """the three-bucket route under MPF_CHECK_SYNC with a gradient injected after the head bucket's launch (from the res5
    hook): FlatGradSync.finish must raise — a dropped gradient would otherwise only show as slowly diverging replicas"""
    r = _run_two_ranks("flat", MPF_TEST_LATE_GRAD="1")
    assert r.returncode != 0, "a late gradient went unnoticed"
    assert "arrived after its all-reduce was launched" in r.stderr, r.stderr[-3000:]


@pytest.mark.timeout(1500)
def test_missing_res3_hook_is_caught_up_by_finish():
    """the second early launch never happens: finish() exchanges that bucket itself; replicas identical afterwards"""
    r = _run_two_ranks("flat", MPF_TEST_NO_RES3_HOOK="1")
    assert r.returncode == 0, f"bench.py --gpus 2 failed:\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["param_sync_spread"] <= 1e-7, out["config"]["param_sync_spread"]


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("grad_sync", ["flat", "ddp", "flat-bf16"])
def test_bench_two_ranks_one_device(grad_sync):
    wire = "bf16" if grad_sync.endswith("-bf16") else "fp32"
    grad_sync = grad_sync.split("-")[0]
    r = _run_two_ranks(grad_sync, ("--grad-wire", wire))
    assert r.returncode == 0, f"bench.py --gpus 2 failed:\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    pg = out["config"]["process_group"]           # proves what the process group saw: backend, ranks, (RCCL version when nccl)
    assert pg["backend"] == "gloo" and pg["world_size"] == 2 and pg["grad_sync"] == grad_sync and "rccl_version" not in pg
    assert out["config"]["trained_like_offsets"]["ms_per_step"] > 0
    assert out["cpu_baseline"] is None                      # rank 0 at N = 1 only
    assert out["value"] > 0 and out["ms_per_step"] > 0
    loss = out["config"]["final_loss"]
    assert loss == loss and abs(loss) < 1e6, loss
    # identical parameters on both ranks after 4 averaged updates: per parameter tensor |sum_rank0 - sum_rank1| / sum|p|
    # (a tensor left out of the exchange would differ by >= 1e-5 after four AdamW steps)
    assert out["config"]["param_sync_spread"] <= 1e-7, out["config"]["param_sync_spread"]
    gs = out["config"].get("grad_sync")
    if grad_sync == "flat":                                  # the N > 1 line explains its own exchange
        assert gs["wire_dtype"] == wire and len(gs["buckets_mb"]) == 3 and len(gs["allreduce_ms"]) == 3 and len(gs["in_flight_ms"]) == 3
        assert all(t > 0 for t in gs["allreduce_ms"]) and gs["exposed_wait_ms"] >= 0
        assert gs["launched_under_backbone"] is True and gs["launched_early"] == [True, True, False]
        mb = {"fp32": (80.0, 88.0), "bf16": (40.0, 44.0)}[wire]               # head | res5 + res4 buckets on the wire
        assert abs(gs["buckets_mb"][0] - mb[0]) < 0.1 * mb[0] and abs(gs["buckets_mb"][1] - mb[1]) < 0.1 * mb[1], gs["buckets_mb"]
    else:
        assert gs is None
    assert out["roofline"]["launches_per_step"] > 0         # the split-bf16 GEMMs ran on the profiled step ...
    msda = [e for e in out["roofline"]["also"] if e["kernel"].startswith("MSDA backward")]
    assert msda and msda[0]["launches"] > 0                 # ... and so did the native MSDA backward


@pytest.mark.timeout(1500)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the reference's train_net.py:399-412 starts its own
    process per GPU): bench.py spawns torch.distributed.run itself before touching the GPU, relays rank 0's single JSON line and
    returns the children's status.  MPF_BENCH_LAUNCH_SCRIPT points the ranks at the one-GPU child wrapper (both ranks on cuda:0,
    gloo), everything else is the product's launch path."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MPF_FORCE_DIST"):
        env.pop(k, None)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MPF_CHECK_SYNC="1", MPF_GRAD_SYNC="flat",
               MPF_BENCH_LAUNCH_SCRIPT=os.path.join(ROOT, "tests", "_bench_world2_child.py"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--profile-steps", "1",
           "--trained-steps", "0", "--size", "512", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, f"self-launched bench.py --gpus 2 failed:\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["process_group"]["world_size"] == 2
    assert out["config"]["param_sync_spread"] <= 1e-7
