"""Host-side logic of the decoder / criterion mirrors that needs no GPU: the row-gather of the per-level key / value
weights, the column-block views with their packed gradient, the shared stacked ground-truth masks."""
import numpy as np
import torch


def test_gather_rows_concatenates_and_routes_gradients_as_views():
    from mp_former_amd.transformer_decoder import _GatherRows
    torch.manual_seed(0)
    ws = [torch.randn(4, 6, requires_grad=True) for _ in range(5)] + [torch.randn(4, requires_grad=True) for _ in range(5)]
    sizes = [3, 2, 3, 2]
    outs = _GatherRows.apply(sizes, *ws)
    assert [tuple(o.shape) for o in outs] == [(12, 6), (8, 6), (12,), (8,)]
    assert torch.equal(outs[0], torch.cat(ws[0:3])) and torch.equal(outs[1], torch.cat(ws[3:5]))
    assert torch.equal(outs[2], torch.cat(ws[5:8])) and torch.equal(outs[3], torch.cat(ws[8:10]))
    gs = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, gs)
    assert torch.equal(ws[1].grad, gs[0][4:8]) and torch.equal(ws[4].grad, gs[1][4:8]) and torch.equal(ws[9].grad, gs[3][4:8])


def test_split_cols_packs_the_gradient_in_place():
    """split_cols: consumers that write their block of the packed gradient in place (what DecoderLayerFn.backward does) and
    consumers that return an ordinary gradient both end up in ONE [S, N, 3E] gradient; an unused block is zero."""
    from mp_former_amd.decoder_layer import split_cols
    torch.manual_seed(1)
    S, N, E = 5, 2, 8
    y = torch.randn(S, N, 3 * E, requires_grad=True)
    (k0, h0), (k1, h1), (k2, h2) = split_cols(y * 1.0, 3)
    assert torch.equal(k1, y[..., E:2 * E]) and h1[1] == 1 and h0[0] is h1[0] is h2[0]

    class InPlace(torch.autograd.Function):           # a consumer that writes its block of the pack, like the decoder layer
        @staticmethod
        def forward(ctx, x, handle):
            ctx.handle = handle
            return x.sum()

        @staticmethod
        def backward(ctx, g):
            pack, j = ctx.handle
            view = pack.block_view(j)
            view.fill_(float(j + 1))
            return view, None

    loss = InPlace.apply(k0, h0) + (k1 * 2.0).sum()   # k2 unused
    loss.backward()
    want = torch.zeros(S, N, 3 * E)
    want[..., :E] = 1.0
    want[..., E:2 * E] = 2.0
    assert torch.equal(y.grad, want)
    with torch.no_grad():                             # no graph: plain views, no handles
        views = split_cols(y.detach(), 3)
    assert all(h is None for _, h in views)


def test_stacked_masks_is_shared_within_a_step_and_follows_inplace_changes():
    from mp_former_amd._targets import stacked_masks
    a = torch.zeros(3, 4, 4, dtype=torch.bool)
    b = torch.ones(2, 4, 4, dtype=torch.bool)
    s1 = stacked_masks([a, b])
    s2 = stacked_masks([a, b])
    assert s1 is s2 and s1.shape == (5, 4, 4) and bool(s1[3:].all()) and not bool(s1[:3].any())
    a[0, 0, 0] = True                                  # version counter changes -> rebuilt
    s3 = stacked_masks([a, b])
    assert s3 is not s1 and bool(s3[0, 0, 0])
    assert stacked_masks([b, a]).shape == (5, 4, 4) and bool(stacked_masks([b, a])[:2].all())


def test_slot_layout_orders_pairs_by_image():
    """criterion._slot_layout: the logits planes of the step's pairs back to back, image by image (no padding)."""
    from mp_former_amd.criterion import SetCriterion
    bi = np.array([0, 1, 1, 0, 1, 0, 0])
    slot, first, count = SetCriterion._slot_layout(bi, 3)
    assert list(slot) == [0, 4, 5, 1, 6, 2, 3]
    assert list(first) == [0, 4, 7] and list(count) == [4, 3, 0]


def test_factored_masks_views_and_row_offsets():
    """mask_fused.FactoredMasks: dim-1 slices stay factored (incl. the decoder's [:, :-nq] / [:, -nq:] split) and address
    the embedding rows through the strides of the sequence-first tensor the heads produce."""
    from mp_former_amd.mask_fused import FactoredMasks
    L, Qt, N, C, H, W = 3, 7, 2, 256, 8, 16
    me = torch.zeros(L * Qt, N, C).transpose(0, 1)              # [N, L*Qt, C] view of [L*Qt, N, C]
    mf = torch.zeros(N, H, W, C).permute(0, 3, 1, 2)
    root = FactoredMasks(me, mf)
    assert root.shape == (N, L * Qt, H, W) and root.dim() == 4
    layer1 = root[:, Qt:2 * Qt]
    dn, main = layer1[:, :-5], layer1[:, -5:]
    assert (dn.q0, dn.q1, main.q0, main.q1) == (7, 9, 9, 14) and main.shape == (N, 5, H, W)
    assert main.same_factors(root) and not main.same_factors(FactoredMasks(me.clone(), mf))
    offs = main.row_offsets(np.array([0, 1]), np.array([0, 4]))
    assert list(offs) == [9 * N * C, C + 13 * N * C]
    assert dn.detach().q0 == 7


def test_prepare_targets_pads_masks_and_normalises_boxes():
    """head.prepare_targets == maskformer_model.py:281-299 (masks zero-padded to the batch size, cxcywh / (w, h, w, h))."""
    from types import SimpleNamespace
    from mp_former_amd.head import prepare_targets
    m0 = torch.rand(3, 20, 30) > 0.5
    m1 = torch.rand(0, 16, 16) > 0.5
    inst = [SimpleNamespace(image_size=(20, 30), gt_classes=torch.tensor([1, 5, 2]), gt_masks=m0,
                            gt_boxes=SimpleNamespace(tensor=torch.tensor([[3.0, 2.0, 9.0, 12.0], [0.0, 0.0, 30.0, 20.0], [10.0, 5.0, 20.0, 15.0]]))),
            SimpleNamespace(image_size=(16, 16), gt_classes=torch.zeros(0, dtype=torch.long), gt_masks=SimpleNamespace(tensor=m1))]
    t = prepare_targets(inst, (32, 32))
    assert t[0]["masks"].shape == (3, 32, 32) and t[0]["masks"].dtype == torch.bool
    assert torch.equal(t[0]["masks"][:, :20, :30], m0) and not bool(t[0]["masks"][:, 20:].any()) and not bool(t[0]["masks"][:, :, 30:].any())
    want = torch.tensor([[6.0 / 30, 7.0 / 20, 6.0 / 30, 10.0 / 20], [0.5, 0.5, 1.0, 1.0], [0.5, 0.5, 10.0 / 30, 0.5]])
    assert torch.allclose(t[0]["boxes"], want) and torch.equal(t[0]["labels"], torch.tensor([1, 5, 2]))
    assert t[1]["masks"].shape == (0, 32, 32) and t[1]["boxes"] is None
    import pytest
    with pytest.raises(ValueError):
        prepare_targets(inst, (16, 16))


def test_decoder_weight_cache_follows_the_module_tree():
    """transformer_decoder._weights caches (name, owner, key) triples; a replaced parameter, a swapped submodule and a wrapped
    submodule must all be picked up on the next call (the cache may not outlive the tree it was built from)."""
    import torch.nn as nn
    from mp_former_amd.transformer_decoder import MLP, MultiScaleMaskedTransformerDecoderMaskDN
    dec = MultiScaleMaskedTransformerDecoderMaskDN(256, True, num_classes=5, hidden_dim=256, num_queries=4, nheads=8,
                                                   dim_feedforward=64, dec_layers=2, pre_norm=False, mask_dim=256,
                                                   enforce_input_project=False, dn_mode="points")
    w0 = dec._weights()
    assert w0["class_embed.weight"] is dec.class_embed.weight
    assert "decoder_norm.weight" not in w0 and "query_feat.weight" not in w0
    assert dec._weights()["mask_embed.layers.0.weight"] is dec.mask_embed.layers[0].weight          # cached path
    dec.class_embed.weight = nn.Parameter(torch.zeros_like(dec.class_embed.weight))                 # replaced parameter
    assert dec._weights()["class_embed.weight"] is dec.class_embed.weight
    dec.mask_embed = MLP(256, 256, 256, 3)                                                          # swapped submodule
    assert dec._weights()["mask_embed.layers.2.weight"] is dec.mask_embed.layers[2].weight
    dec.transformer_ffn_layers[1].linear1 = nn.Linear(256, 64)                                      # swapped leaf, two levels down
    assert dec._weights()["transformer_ffn_layers.1.linear1.weight"] is dec.transformer_ffn_layers[1].linear1.weight
    dec.class_embed.register_parameter("extra", nn.Parameter(torch.zeros(3)))                       # added parameter
    assert dec._weights()["class_embed.extra"] is dec.class_embed.extra


def test_configure_training_process_moves_backward_to_the_calling_thread():
    """dropin.configure_training_process: backward() runs on the calling thread afterwards (a python autograd node sees the
    caller's thread id), gradients unchanged; the previous setting is returned so a caller can restore it."""
    import threading
    from mp_former_amd import dropin
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen.append(threading.get_ident())
            return g * 2

    prev = dropin.configure_training_process(single_thread_autograd=True)
    try:
        assert not torch.autograd.is_multithreading_enabled()
        x = torch.ones(3, requires_grad=True)
        Probe.apply(x).sum().backward()
        assert seen == [threading.get_ident()] and torch.equal(x.grad, torch.full((3,), 2.0))
    finally:
        torch.autograd.set_multithreading_enabled(prev)
    assert dropin.configure_training_process(single_thread_autograd=False) == prev


def test_cached_module_lookups_follow_the_module_tree():
    """The per-step lookups that bypass Module.__getattr__ (backbone.Bottleneck.pairs, encoder_fused.layer_params,
    FrozenBatchNorm2d.scale_bias) are caches over the module tree: a swapped submodule, a re-assigned parameter, an in-place
    change or a replacement of a frozen statistic must all be picked up."""
    from torch import nn
    from mp_former_amd import encoder_fused as EF, pixel_decoder as PD
    from mp_former_amd.backbone import Bottleneck, FrozenBatchNorm2d
    b = Bottleneck(64, 64, 256, 1)
    p1 = b.pairs()
    assert b.pairs() is p1 and len(p1) == 4 and p1[3][0] is b.shortcut
    b.conv1 = nn.Conv2d(64, 64, 1, bias=False)
    assert b.pairs() is not p1 and b.pairs()[0][0] is b.conv1
    assert len(Bottleneck(256, 64, 256, 1).pairs()) == 3
    enc = PD.MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=1, dim_feedforward=1024, dropout=0.0,
                                                num_feature_levels=3)
    layer = enc.encoder.layers[0]
    q1 = EF.layer_params(layer)
    a = layer.self_attn
    want = [a.sampling_offsets.weight, a.sampling_offsets.bias, a.attention_weights.weight, a.attention_weights.bias,
            a.value_proj.weight, a.value_proj.bias, a.output_proj.weight, a.output_proj.bias, layer.norm1.weight, layer.norm1.bias,
            layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias]
    assert len(q1) == 16 and all(x is y for x, y in zip(q1, want))
    layer.linear1 = nn.Linear(256, 1024)
    assert EF.layer_params(layer)[10] is layer.linear1.weight
    layer.norm1.weight = nn.Parameter(torch.ones(256))
    assert EF.layer_params(layer)[8] is layer.norm1.weight
    bn = FrozenBatchNorm2d(8)
    bn.running_var.fill_(4.0)
    s1, _ = bn.scale_bias()
    assert bn.scale_bias()[0] is s1 and torch.allclose(s1, torch.full((8,), 0.5))
    bn.weight.mul_(2.0)                                        # in place: the version counter moves
    assert torch.allclose(bn.scale_bias()[0], torch.ones(8))
    bn = bn.double()                                           # .to(): the buffers are replaced
    assert bn.scale_bias()[0].dtype == torch.float64


def test_bench_self_launch_starts_one_fresh_process_per_rank(tmp_path):
    """bench.py --gpus N without a launcher (train_net.py:399-412 starts its own ranks): the parent spawns torch.distributed.run
    with the same arguments before anything touches the GPU, relays the ranks' stdout and returns their status.  The ranks here
    are a stub script (MPF_BENCH_LAUNCH_SCRIPT) that reports what the launcher gave it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / "rank.py"
    stub.write_text("import json, os, sys\n"
                    "if os.environ['RANK'] == '0':\n"
                    "    print(json.dumps({'world': os.environ['WORLD_SIZE'], 'addr': os.environ['MASTER_ADDR'], 'argv': sys.argv[1:]}), flush=True)\n"
                    "sys.exit(3 if '--fail' in sys.argv else 0)\n")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["MPF_BENCH_LAUNCH_SCRIPT"] = str(stub)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got == {"world": "2", "addr": "127.0.0.1", "argv": ["--gpus", "2", "--steps", "3", "--warmup", "1"]}
    # a failing rank's status reaches the caller
    env["MPF_BENCH_LAUNCH_SCRIPT"] = str(stub)
    stub.write_text(stub.read_text().replace("if '--fail' in sys.argv", "if True"))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_registry_plugin_builds_both_decoders_by_name():
    """mp_former_amd.d2_plugin against stub registries of detectron2's interface (this image has no detectron2): the reference's
    build functions (pixel_decoder/fpn.py:21-34, transformer_decoder/maskformer_transformer_decoder.py:21-28) look a class up
    by the configured NAME and call it with (cfg, ...) — both HIP classes must come back from the registries and construct
    themselves from a config tree with the reference's keys, with the reference's state-dict keys."""
    from types import SimpleNamespace as NS
    from mp_former_amd import d2_plugin

    class Registry:                       # detectron2.utils.registry.Registry (fvcore): register(obj) as call or decorator, get(name)
        def __init__(self, name):
            self._name, self._map = name, {}

        def register(self, obj=None):
            if obj is None:
                return lambda o: self.register(o) or o
            assert obj.__name__ not in self._map, f"{obj.__name__} already registered in {self._name}"
            self._map[obj.__name__] = obj
            return obj

        def get(self, name):
            if name not in self._map:
                raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
            return self._map[name]

    assert d2_plugin.registered() is False                      # no detectron2 here: the import-time registration was a no-op
    sem, trf = Registry("SEM_SEG_HEADS"), Registry("TRANSFORMER_MODULE")
    assert d2_plugin.register(sem, trf)
    cfg = NS(MODEL=NS(
        SEM_SEG_HEAD=NS(IN_FEATURES=["res2", "res3", "res4", "res5"], CONVS_DIM=256, MASK_DIM=256, NORM="GN", TRANSFORMER_ENC_LAYERS=2,
                        DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES=["res3", "res4", "res5"], COMMON_STRIDE=4, NUM_CLASSES=80,
                        PIXEL_DECODER_NAME=d2_plugin.PIXEL_DECODER_NAME),
        MASK_FORMER=NS(DROPOUT=0.0, NHEADS=8, HIDDEN_DIM=256, NUM_OBJECT_QUERIES=100, DIM_FEEDFORWARD=2048, DEC_LAYERS=4, PRE_NORM=False,
                       ENFORCE_INPUT_PROJ=False, DN_MODE="points", HEAD_DN=False, ALL_LY_DN=True, DN_RATIO=0.5, LB_NOISE_RATIO=0.2,
                       TRANSFORMER_DECODER_NAME=d2_plugin.TRANSFORMER_DECODER_NAME)))
    shape = {k: NS(channels=c, stride=s) for k, (c, s) in {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}.items()}
    # build_pixel_decoder / build_transformer_decoder, restated
    pix = sem.get(cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME)(cfg, shape)
    dec = trf.get(cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME)(cfg, 256, True)
    from mp_former_amd.pixel_decoder import MSDeformAttnPixelDecoder
    from mp_former_amd.transformer_decoder import MultiScaleMaskedTransformerDecoderMaskDN
    assert isinstance(pix, MSDeformAttnPixelDecoder) and isinstance(dec, MultiScaleMaskedTransformerDecoderMaskDN)
    assert len(pix.transformer.encoder.layers) == 2 and dec.num_layers == 3 and dec.num_heads == 8
    assert hasattr(pix, "forward_features") and callable(pix.forward_features)
    keys = set(pix.state_dict()) | set(dec.state_dict())
    for k in ("transformer.encoder.layers.0.self_attn.sampling_offsets.weight", "input_proj.0.1.weight", "mask_features.weight",
              "adapter_1.norm.weight", "layer_1.weight", "query_feat.weight", "label_enc.weight",
              "transformer_cross_attention_layers.0.multihead_attn.in_proj_weight", "mask_embed.layers.2.bias", "class_embed.weight"):
        assert k in keys, k
    # keyword construction (what from_config returns) still works on the registered class, and a second registration of the same
    # name is refused by the registry as detectron2's does
    again = sem.get(d2_plugin.PIXEL_DECODER_NAME)(**type(pix).from_config(cfg, shape))
    assert list(again.state_dict()) == list(pix.state_dict())
    import pytest
    with pytest.raises(AssertionError):
        d2_plugin.register(sem, trf)


def test_encoder_arena_layout_is_aligned_and_disjoint():
    """_arena_layout (encoder_fused.py): the per-layer results of the native encoder forward as slices of ONE allocation — every
    tensor starts on a 16-byte boundary, none overlaps the next, the bit mask of the FFN gate has a bit per hidden element."""
    from mp_former_amd.encoder_fused import _arena_layout
    for (R, F) in ((43008, 1024), (35, 128), (1, 2048)):
        M, L, P = 8, 3, 4
        offs, tot = _arena_layout(R, 256, F, M * L * P * 3, M, L, P)
        spans = sorted(offs.values())
        for (o, n), (o2, _) in zip(spans, spans[1:] + [(tot, 0)]):
            assert o % 4 == 0 and o + n <= o2
        assert offs["hbits"][1] * 32 >= R * F and offs["loc"][1] == R * M * L * P * 2 and offs["x2"][1] == R * 256
