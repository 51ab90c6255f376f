"""The natively sequenced decoder layer (csrc/decoder_layer.hip) against the op-by-op path: same kernels,
same rounding points — losses and gradients must agree to bf16 accumulation-order noise."""
import os

import pytest
import torch

from conftest import fifo_to_tags, load_head_fixture
from test_head_gpu import _build

pytestmark = pytest.mark.gpu


def _run(h, feats, targets, replay, cfg, fused):
    from mp_former_amd import _rng
    os.environ["MPF_FUSED_DECODER"] = "1" if fused else "0"
    h.zero_grad(set_to_none=True)
    _rng.install_replay(fifo_to_tags(replay, cfg, True))
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses, _ = h(feats, targets)
            total = sum(losses.values())
        total.backward()
    finally:
        _rng.install_replay(None)
        os.environ.pop("MPF_FUSED_DECODER", None)
    grads = {n: p.grad.detach().clone() for n, p in h.named_parameters() if p.grad is not None}
    return {k: float(v) for k, v in losses.items()}, grads


def test_fused_decoder_layer_matches_op_by_op_path():
    from mp_former_amd import _lib
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_small")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    l_ops, g_ops = _run(h, feats, targets, replay, cfg, fused=False)
    _lib.profile_enable(True)
    l_fused, g_fused = _run(h, feats, targets, replay, cfg, fused=True)
    torch.cuda.synchronize()
    n_small, _, _ = _lib.profile_get("small_gemm_kernel")
    _lib.profile_enable(False)
    assert n_small > 0
    for k in l_ops:
        assert abs(l_ops[k] - l_fused[k]) <= 2e-3 * max(1.0, abs(l_ops[k])), (k, l_ops[k], l_fused[k])
    assert set(g_ops) == set(g_fused)
    worst = ("", 0.0)
    for n in g_ops:
        a, b = g_ops[n].double().flatten(), g_fused[n].double().flatten()
        rel = (a - b).norm().item() / (a.norm().item() + 1e-12)
        if rel > worst[1]:
            worst = (n, rel)
    # identical kernels; differences are the order of bf16 gradient accumulation and of the LayerNorm atomics
    assert worst[1] < 2e-2, worst


def test_decoder_layer_rejects_bad_struct():
    import ctypes
    from mp_former_amd import _lib
    from mp_former_amd.decoder_layer import MpfDecoderLayer
    L = MpfDecoderLayer()
    L.Qt, L.N, L.H, L.S, L.ffn_dim = 10, 1, 8, 64, 2048
    assert _lib.lib().mpf_decoder_layer_forward(ctypes.byref(L), None) == -3      # MPF_E_NULL
    L.H = 4
    assert _lib.lib().mpf_decoder_layer_forward(ctypes.byref(L), None) == -2      # MPF_E_SHAPE
