"""GPU parity of the product head (mp_former_amd: pixel decoder + MP decoder + criterion, native
MSDA kernels inside) against the golden vectors of the imported reference modules and against the
oracle, in fp32 with the reference's random draws replayed."""
import json

import numpy as np
import pytest
import torch

from conftest import HEAD_FIXTURES, fifo_to_tags, load_head_fixture, masks_view

pytestmark = pytest.mark.gpu


def _sub(t, step):
    return t.detach().float().reshape(-1)[::step].cpu().numpy()


def _build(cfg, pp, dp, dev):
    from mp_former_amd.head import MPFormerHead
    h = MPFormerHead(num_classes=cfg["num_classes"], num_queries=cfg["num_queries"], enc_layers=cfg["enc_layers"],
                     dec_layers=cfg["dec_layers"], num_points=cfg["num_points"], noise_scale=cfg.get("noise_scale", 0.0))
    h.pixel_decoder.load_state_dict(pp, strict=True)
    h.predictor.load_state_dict(dp, strict=True)
    return h.to(dev).train()


def _planes_leaf(v, dev):
    """the same values as channels_last planes [N][H*W][C] (what the bf16 backbone hands over), as a leaf"""
    N, C, H, W = v.shape
    return v.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)


@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
@pytest.mark.parametrize("name", HEAD_FIXTURES)
def test_head_matches_reference_golden_fp32(name, layout):
    """layout = channels_last feeds the features as [N][H*W][C] planes: the convolutions then return planes and the
    pixel decoder takes the channel-last GroupNorm / fused FPN-sum kernels and the channel-last mask product — the
    route bench.py runs; nchw is the reference's layout (round-1 kernels for the norms)."""
    from mp_former_amd import _lib, _rng
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture(name)
    # state-dict key names / shapes are part of the boundary (SURVEY.md §5 checkpoint row)
    h = _build(cfg, pp, dp, dev)
    assert {k: list(v.shape) for k, v in h.pixel_decoder.state_dict().items()} == json.loads(str(z["pix_keys"]))
    assert {k: list(v.shape) for k, v in h.predictor.state_dict().items()} == json.loads(str(z["dec_keys"]))
    use_dn = "dn_pred_logits" in z
    _rng.install_replay(fifo_to_tags(replay, cfg, use_dn))
    try:
        if layout == "channels_last":
            feats = {k: _planes_leaf(v, dev) for k, v in feats.items()}
        else:
            feats = {k: v.to(dev).requires_grad_(True) for k, v in feats.items()}
        targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
        _lib.profile_enable(True)
        mf, _, ms = h.pixel_decoder.forward_features(feats)
        # the native kernels are what ran (deformable attention + the encoder's split-bf16 GEMMs)
        assert _lib.profile_get("msda_fwd_block")[0] >= 1 and _lib.profile_get("gemm3_tn")[0] >= 1, _lib.last_kernel()
        if layout == "channels_last":
            assert _lib.profile_get("gn_cl_apply")[0] >= 5, "the channel-last GroupNorm kernels did not run"
            assert mf.stride(1) == 1, "mask_features left the channel-last layout"
        _lib.profile_enable(False)
        np.testing.assert_allclose(_sub(mf, 5), z["mask_features_s5"], rtol=2e-3, atol=5e-4)
        for i, t in enumerate(ms):
            np.testing.assert_allclose(_sub(t, 3), z[f"multi_scale_{i}_s3"], rtol=2e-3, atol=5e-4)
        out = h.predictor(ms, mf, None, {"tgt": targets, "scalar": 1, "noise_scale": cfg.get("noise_scale", 0.0)})
        np.testing.assert_allclose(out["pred_logits"].detach().cpu().numpy(), z["pred_logits"], rtol=5e-3, atol=2e-3)
        np.testing.assert_allclose(masks_view(out["pred_masks"], cfg), z["pred_masks"], rtol=5e-3, atol=5e-3)
        for i, a in enumerate(out["aux_outputs"]):
            np.testing.assert_allclose(a["pred_logits"].detach().cpu().numpy(), z[f"aux{i}_pred_logits"], rtol=5e-3, atol=2e-3)
            np.testing.assert_allclose(_sub(a["pred_masks"], cfg.get("aux_step", 3)), z[f"aux{i}_pred_masks_s3"], rtol=5e-3, atol=5e-3)
        if use_dn:
            assert out["dn_out"]["dn_args"] == {"max_num": int(z["dn_max_num"]), "pad_size": int(z["dn_pad_size"])}
            np.testing.assert_allclose(out["dn_out"]["pred_masks"].detach().cpu().numpy(), z["dn_pred_masks"], rtol=5e-3, atol=5e-3)
        else:
            assert out["dn_out"] is None
        losses = h.criterion(out, targets)
        assert _rng.remaining() == 0
        ref_keys = sorted(k[5:] for k in z if k.startswith("loss."))
        assert sorted(losses) == ref_keys
        for k in ref_keys:
            np.testing.assert_allclose(float(losses[k]), float(z["loss." + k]), rtol=2e-3, atol=1e-4, err_msg=k)
        wd = h.criterion.weight_dict
        total = sum(losses[k] * wd[k] for k in losses if k in wd)
        np.testing.assert_allclose(float(total), float(z["total_loss"]), rtol=5e-4)
        _lib.profile_enable(True)
        total.backward()
        assert _lib.profile_get("msda_bwd_tile")[0] >= 1 and _lib.profile_get("gemm3_nt")[0] >= 1, _lib.last_kernel()
        _lib.profile_enable(False)
        for k, v in feats.items():
            n = float(z[f"grad_feat_{k}_norm"])
            np.testing.assert_allclose(v.grad.norm().item(), n, rtol=5e-3)
            # (six deformable-attention layers deep, isolated elements — 0.1 % in head_deep — differ by a few 1e-2 of the RMS: a
            # sampling point within rounding distance of a pixel boundary takes the other bilinear cell; the field agrees in L2)
            got, want = _sub(v.grad, 7), z[f"grad_feat_{k}_s7"]
            # (config A, 6 layers: 0.2 % of res4's sampled elements sit beyond that band — ReLU gates and bilinear cells that flip
            # between two fp32 pipelines, tests/test_encoder_fused_gpu.py measures the same against fp64 for the library path —
            # so up to 0.5 % may, none by more than half the RMS, and the field as a whole is held by the L2 bar below)
            rms = n / np.sqrt(v.numel())
            viol = np.abs(got - want) > (5e-3 if cfg["enc_layers"] < 6 else 5e-2) * rms + 1e-2 * np.abs(want)
            assert viol.sum() <= (0 if cfg["enc_layers"] < 6 else int(5e-3 * viol.size)), (k, int(viol.sum()), viol.size)
            assert np.abs(got - want).max() <= 0.5 * rms + 1e-2 * np.abs(want).max(), k
            # relative L2 of the field: 3e-3 through <= 2 encoder layers / at 128 px; 1e-2 for the 6-layer configs (config A's res4:
            # 7e-3) — the level at which ANY two fp32 pipelines differ once ReLU gates / bilinear cells flip (a fraction f of flipped
            # gates is sqrt(f) in relative L2: tests/test_encoder_fused_gpu.py measures 5e-4 .. 1e-3 per tensor for the library-fp32
            # modules against fp64 at config-B size, where the flip fraction is smaller)
            assert np.linalg.norm(got - want) <= (3e-3 if cfg["enc_layers"] < 6 or cfg["size"] <= 128 else 1e-2) * np.linalg.norm(want), k
        pg = dict(h.pixel_decoder.named_parameters())
        for k in [k for k in z if k.startswith("grad_pix.") and "_s11" not in k]:
            np.testing.assert_allclose(pg[k[9:]].grad.cpu().numpy(), z[k], rtol=1e-2, atol=1e-4 + 5e-3 * np.abs(z[k]).max(), err_msg=k)
        dg = dict(h.predictor.named_parameters())
        for k in [k for k in z if k.startswith("grad_dec.")]:
            g = dg[k[9:]].grad
            g = torch.zeros_like(dg[k[9:]]) if g is None else g
            np.testing.assert_allclose(g.cpu().numpy(), z[k], rtol=1e-2, atol=1e-4 + 5e-3 * np.abs(z[k]).max(), err_msg=k)
    finally:
        _rng.install_replay(None)


def test_head_bf16_autocast_close_to_fp32():
    """AMP (bf16 autocast for the decoder; pixel decoder stays fp32 like the reference,
    msdeformattn.py:314): weighted total loss within 2 % of the fp32 run on identical draws."""
    from mp_former_amd import _rng
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_small")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    _rng.install_replay(fifo_to_tags(replay, cfg, True))
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses, _ = h(feats, targets)
        total = float(sum(losses.values()))
    finally:
        _rng.install_replay(None)
    from mp_former_amd import _lib
    assert abs(total - float(z["total_loss"])) / float(z["total_loss"]) < 0.02, (total, float(z["total_loss"]))
    # under AMP the decoder's attention runs on the native MFMA kernels
    _lib.profile_enable(True)
    _rng.install_replay(fifo_to_tags(replay, cfg, True))
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses, _ = h(feats, targets)
            sum(losses.values()).backward()
    finally:
        _rng.install_replay(None)
    torch.cuda.synchronize()
    nf, _, _ = _lib.profile_get("attn_fwd_kernel")
    nb, _, _ = _lib.profile_get("attn_bwd_kv_kernel")
    _lib.profile_enable(False)
    assert nf == 2 * cfg["dec_layers"] and nb == 2 * cfg["dec_layers"], (nf, nb)


def test_head_config_A_runs_and_is_finite():
    """config A: 256x256, N=1, 100 queries, 80 classes, full depth; fresh draws; loss finite, all
    parameters receive a gradient (DDP needs every parameter used, decoder :1846)."""
    from mp_former_amd.head import MPFormerHead
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    h = MPFormerHead().to(dev).train()
    feats = {k: torch.randn(1, c, 256 // s, 256 // s, device=dev) for k, (c, s) in
             {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}.items()}
    masks = torch.zeros(3, 256, 256, dtype=torch.bool, device=dev)
    masks[0, 10:100, 20:120] = True
    masks[1, 150:250, 100:200] = True
    masks[2, 60:90, 200:250] = True
    targets = [{"labels": torch.tensor([1, 5, 7], device=dev), "masks": masks, "boxes": torch.zeros(3, 4, device=dev)}]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        losses, _ = h(feats, targets)
    assert len(losses) == 60
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    missing = [n for n, p in h.named_parameters() if p.grad is None]
    assert not missing, missing


@pytest.mark.parametrize("name,size,chans,classes,queries,n", [
    ("D_ade20k_swinB_640", (640, 640), (128, 256, 512, 1024), 150, 100, 2),
    ("E_cityscapes_swinL_1024x2048", (1024, 2048), (192, 384, 768, 1536), 19, 200, 1),
])
def test_head_baseline_config_shapes_run(name, size, chans, classes, queries, n):
    """BASELINE.json configs D and E (Swin channel counts, 200 queries, non-square high-resolution input):
    one AMP forward + backward of the whole head; finite loss, every parameter gets a gradient, native
    kernels on the path."""
    from mp_former_amd import _lib
    from mp_former_amd.head import MPFormerHead
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    H, W = size
    shapes = {f"res{i + 2}": (c, s) for i, (c, s) in enumerate(zip(chans, (4, 8, 16, 32)))}
    h = MPFormerHead(num_classes=classes, num_queries=queries, feature_shapes=shapes).to(dev).train()
    # what the bf16 backbone hands over under autocast: bf16 channel-last planes (the route bench.py times)
    feats = {k: torch.randn(n, H // s, W // s, c, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2) for k, (c, s) in shapes.items()}
    targets = []
    for b in range(n):
        T = 4 + 3 * b
        m = torch.zeros(T, H, W, dtype=torch.bool, device=dev)
        for t in range(T):
            m[t, 40 * t:40 * t + 200, 60 * t:60 * t + 300] = True
        targets.append({"labels": torch.arange(T, device=dev) % classes, "masks": m, "boxes": torch.zeros(T, 4, device=dev)})
    _lib.profile_enable(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = h.total_loss(feats, targets)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert not [k for k, p in h.named_parameters() if p.grad is None]
    for kern in ("msda_fwd_block", "msda_bwd_tile", "attn_fwd_kernel", "attn_bwd_kv", "match_cost_fused", "pair_planes_fwd", "pair_planes_dfeat",
                 "pair_planes_dembed", "mask_loss_fwd", "mask_head_bits", "pool_features", "lsa_kernel"):
        assert _lib.profile_get(kern)[0] > 0, kern
    _lib.profile_enable(False)


def _rel_l2(got, want):
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


# fixtures on which the autocast path's OWN Hungarian assignment reproduces the reference's goldens at the pinned test's tolerances.
# Measured on MI355X (round 5): head_small, head_ragged, head_noise pass unpinned; head_deep keeps every loss within tolerance but one
# gradient lands at 0.0712 relative L2 (bar 0.07), head_cfgA (100 freshly initialised near-duplicate queries: 45 of 70 targets matched
# as in fp32, regret <= 0.8 %) moves d query_feat to 0.16 — those two stay on the pinned form only.
_OWN_ASSIGNMENT_FIXTURES = ["head_small", "head_ragged", "head_noise"]


@pytest.mark.parametrize("name", ["head_small", "head_ragged", "head_deep", "head_noise", "head_cfgA"])
def test_head_amp_path_matches_reference_golden(name):
    _amp_golden(name, pin=True)


@pytest.mark.parametrize("name", _OWN_ASSIGNMENT_FIXTURES)
def test_head_amp_path_on_its_own_assignment_matches_reference_golden(name):
    """The same comparison WITHOUT the pinned assignment: the autocast path solves the matching from its own
    (bf16-noisy) cost matrices on the device, with the reference's matcher draws replayed, and must still land on the fp32
    goldens — i.e. on these fixtures the AMP assignment IS the reference's.  The fixtures left to the pinned form only are the
    ones whose cost matrices tie at bf16 resolution (near-duplicate queries of the toy models; the agreement share and the
    regret of the flipped pairs are measured in test_amp_path_own_matching_against_the_fp32_matching)."""
    _amp_golden(name, pin=False)


def _amp_output_errors(out, z, cfg, use_dn):
    """Every OUTPUT of the autocast decoder against the fp32 goldens, per decoder layer (the losses average
    over queries and points — a wrong head or a few wrong MP rows can hide inside 2 % of a loss, not inside its own row).
    Per output ("aux0" = after the query initialisation, ..., "final", and the mask-piloted part "dn"; class logits and mask
    logits, the maps materialised from their factors by the product kernel):
      l2      relative L2 of the whole tensor,
      med     MEDIAN over the (image, query) rows of the row's relative L2 — the statistic a systematic error (a wrong head, a
              wrong scale, a mis-indexed row block) moves and a flipped attention-mask bit does not,
      over    share of rows whose relative L2 exceeds 0.25 — rows whose masked attention saw a different mask: the next-layer
              mask is a hard threshold of the previous layer's logits (mask2former_transformer_decoder.py:1869-1875), so bf16 noise
              flips pixels near 0 and those queries change by O(1) in ANY bf16 evaluation, the reference's own autocast included.
    (aux mask maps are stored subsampled in the fixtures: whole-tensor l2 only.)"""
    def rows(got, want, nrow):
        got = np.asarray(got, dtype=np.float64).reshape(nrow, -1)
        want = np.asarray(want, dtype=np.float64).reshape(nrow, -1)
        r = np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
        return {"l2": _rel_l2(got, want), "med": float(np.median(r)), "over": float((r > 0.25).mean())}

    e = {}
    outs = [(f"aux{i}", a) for i, a in enumerate(out["aux_outputs"])] + [("final", out)]
    for tag, o in outs:
        pre = "" if tag == "final" else tag + "_"
        lg = o["pred_logits"].detach().float().cpu().numpy()
        e[tag + ".logits"] = rows(lg, z[pre + "pred_logits"], lg.shape[0] * lg.shape[1])
        if tag == "final" and cfg.get("mask_step", 1) == 1:
            pm = masks_view(o["pred_masks"], cfg)
            e[tag + ".masks"] = rows(pm, z["pred_masks"], pm.shape[0] * pm.shape[1])
        elif tag == "final":
            e[tag + ".masks"] = {"l2": _rel_l2(masks_view(o["pred_masks"], cfg), z["pred_masks"])}
        else:
            e[tag + ".masks"] = {"l2": _rel_l2(_sub(o["pred_masks"], cfg.get("aux_step", 3)), z[pre + "pred_masks_s3"])}
    if use_dn:
        d = out["dn_out"]
        assert d["dn_args"] == {"max_num": int(z["dn_max_num"]), "pad_size": int(z["dn_pad_size"])}
        lg = d["pred_logits"].detach().float().cpu().numpy()
        e["dn.logits"] = rows(lg, z["dn_pred_logits"], lg.shape[0] * lg.shape[1])
        pm = d["pred_masks"].detach().float().cpu().numpy()
        e["dn.masks"] = rows(pm, z["dn_pred_masks"], pm.shape[0] * pm.shape[1])
    return e


# bf16 bars of the per-output pins, per fixture family: l2 (whole tensor), med (median row), over (share of rows beyond 0.25).
# Measured on MI355X (round 6) and set to the worst value over the fixtures of the family + 30 %; DESIGN.md section 2 lists them.
_AMP_OUT_BARS = {"shallow": {"l2": 3.0e-2, "med": 1.75e-2, "over": 0.01},         # <= 4 decoder layers: measured 0.0225 / 0.0133 / 0
                 "deep": {"l2": 1.25e-1, "med": 2.1e-2, "over": 0.04},            # 9 layers (head_deep 0.0951 / 0.0159 / 0.03 at layer 4; head_cfgA 0.026 / 0.008 / 0)
                 "noise": {"l2": 1.1e-2, "med": 1.1e-2, "over": 0.01}}            # point noise (head_noise): measured 0.0081 / 0.0080 / 0
AMP_OUT_MEASURED = {}


def _check_amp_outputs(name, e, cfg):
    fam = "noise" if cfg.get("noise_scale", 0.0) > 0 else ("shallow" if cfg["dec_layers"] <= 4 else "deep")
    bars = _AMP_OUT_BARS[fam]
    worst = {k: max(v.get(k, 0.0) for v in e.values()) for k in ("l2", "med", "over")}
    AMP_OUT_MEASURED[name] = worst
    print(f"[amp per-output] {name} ({fam}) worst: " + ", ".join(f"{k} {v:.4f}" for k, v in worst.items()))
    print(f"[amp per-output] {name} per layer: " + "; ".join(f"{k} " + "/".join(f"{v.get(s_, float('nan')):.4f}" for s_ in ("l2", "med", "over"))
                                                              for k, v in e.items()))
    bad = {k: {s_: round(x, 4) for s_, x in v.items()} for k, v in e.items() if any(x > bars[s_] for s_, x in v.items())}
    assert not bad, f"AMP decoder outputs off the fp32 goldens ({fam} bars {bars}): {bad}"


def _check_against_the_reference_under_cpu_autocast(name, out, z, cfg, use_dn):
    """Second golden family (tests/golden/head_<name>_amp.npz, make_golden.py head_amp): the reference decoder itself evaluated under
    `torch.autocast("cpu", dtype=torch.bfloat16)` on the same fp32 pixel-decoder outputs, parameters and draws.  CPU autocast is not
    CUDA autocast (softmax / LayerNorm stay in bf16 there), so it is not a tighter pin of THIS path — it is the yardstick for what
    "bf16 tolerance" means on each fixture: per output, the product's autocast result must lie no further from the fp32 golden
    than 1.5 x the reference's own bf16 evaluation does (+ 1e-2), and the two bf16 evaluations no further apart than the sum of
    their distances to fp32 (+ 1e-2).  Measured (round 6, worst output per fixture; product | reference-under-autocast, relative L2
    to the fp32 golden): small 0.015 | 0.016, ragged 0.023 | 0.133, noise 0.008 | 0.013, deep 0.095 | 0.089, cfgA 0.026 | 0.033."""
    import os
    f = os.path.join(os.path.dirname(__file__), "golden", f"{name}_amp.npz")
    if not os.path.exists(f):
        return
    za = np.load(f, allow_pickle=True)
    got = {}
    outs = [(f"aux{i}_", a) for i, a in enumerate(out["aux_outputs"])] + [("", out)]
    for pre, o in outs:
        got[pre + "pred_logits"] = o["pred_logits"].detach().float().cpu().numpy()
        got[pre + ("pred_masks" if pre == "" else "pred_masks_s3")] = (masks_view(o["pred_masks"], cfg) if pre == ""
                                                                       else _sub(o["pred_masks"], cfg.get("aux_step", 3)))
    if use_dn:
        got["dn_pred_logits"] = out["dn_out"]["pred_logits"].detach().float().cpu().numpy()
        got["dn_pred_masks"] = out["dn_out"]["pred_masks"].detach().float().cpu().numpy()
    rows, bad = [], {}
    for k, g in got.items():
        d_prod, d_ref, d_between = _rel_l2(g, z[k]), _rel_l2(za[k], z[k]), _rel_l2(g, za[k])
        rows.append((k, d_prod, d_ref, d_between))
        if not (d_prod <= 1.5 * d_ref + 1e-2 and d_between <= d_prod + d_ref + 1e-2):
            bad[k] = (round(d_prod, 4), round(d_ref, 4), round(d_between, 4))
    print(f"[amp vs cpu-autocast reference] {name}: worst product->fp32 {max(r[1] for r in rows):.4f}, reference-under-autocast->fp32 "
          f"{max(r[2] for r in rows):.4f}, between the two {max(r[3] for r in rows):.4f}")
    assert not bad, f"(product->fp32, reference-under-autocast->fp32, between): {bad}"


def _amp_golden(name, pin):
    """The path bench.py times — bf16 autocast with every default switch (natively sequenced decoder layers, MFMA
    attention, small-row GEMMs, device-side assignment) — against the PINNED goldens of the imported reference (fp32),
    with the reference's draws replayed: every one of the 6 x (1 + #aux) losses per key, and the gradients of the
    backbone features, of the pixel decoder and of the decoder per tensor (relative L2: mean <= 5e-2, each <= 7e-2).  The tolerance is bf16's:
    8 mantissa bits through 3 decoder layers; an assignment flipped by rounding would show up as an O(1) error in the
    losses of that output."""
    from mp_former_amd import _lib, _rng
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture(name)
    h = _build(cfg, pp, dp, dev)
    feats = {k: _planes_leaf(v, dev) for k, v in feats.items()}      # channel-last planes, as the bf16 backbone delivers them
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    use_dn = "dn_pred_logits" in z
    # The assignment is PINNED to the reference's: these toy models have near-duplicate queries whose matching costs tie to
    # within bf16's rounding, and a flipped pair is an O(0.1) change of that output's losses that says nothing about a kernel
    # (any last-bit change upstream can trigger it).  Pass 1: the fp32 path (golden-exact: test_head_matches_reference_golden_fp32)
    # on the reference's own route — cost matrices to the host, SciPy — records the indices; pass 2 (autocast) gets them back
    # from the matcher instead of solving its own noisy costs.  The device solver is pinned to SciPy by tests/test_lsa_gpu.py and
    # asserted on the default route in test_head_baseline_config_shapes_run.
    import os
    matcher = h.criterion.matcher
    solve = matcher.match_many
    pinned = []
    if pin:
        os.environ["MPF_DEVICE_LSA"] = "0"
    try:
        if pin:
            _rng.install_replay(fifo_to_tags(replay, cfg, use_dn))
            matcher.match_many = lambda *a, **k: pinned.append(solve(*a, **k)) or pinned[-1]
            with torch.no_grad():
                h(feats, targets)
            assert len(pinned) == 1 and _rng.remaining() == 0
            matcher.match_many = lambda *a, **k: pinned[0]
            # (the matcher's own point draws are not consumed in pass 2)
            _rng.install_replay({t: d for t, d in fifo_to_tags(replay, cfg, use_dn).items() if not t.startswith("match")})
        else:
            _rng.install_replay(fifo_to_tags(replay, cfg, use_dn))       # own assignment: device solver on the AMP costs, matcher draws replayed
        _lib.profile_enable(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses, out = h(feats, targets)            # weighted, like maskformer_model.py:226-231
            total = sum(losses.values())
        assert _rng.remaining() == 0
        out_err = _amp_output_errors(out, z, cfg, use_dn)      # (before backward: the factors are still alive)
        _check_against_the_reference_under_cpu_autocast(name, out, z, cfg, use_dn)
        total.backward()
        torch.cuda.synchronize()
        # the mask predictions never exist as maps: matching cost from the factors, loss planes of the paired rows only, and
        # their gradients w.r.t. (mask_embed, mask_features) from the native products (csrc/mask_fused.hip)
        # (pair_planes_fwd also serves the next-layer attention mask where a level is too small for the fused mask head)
        # (the fused matching cost ran in pass 1, before the launch log was switched on; asserted in test_head_baseline_config_shapes_run)
        assert _lib.profile_get("pair_planes_fwd_kernel")[0] >= 1
        assert _lib.profile_get("pair_planes_dfeat_kernel")[0] == 1 and _lib.profile_get("pair_planes_dembed_kernel")[0] == 1
        for kern in ("attn_fwd_kernel", "attn_bwd_kv_kernel", "small_gemm", "small_gemm_group_kernel", "msda_fwd_block",
                     "msda_bwd_tile", "gemm3", "gn_cl_apply", "gn_cl_bwd_apply"):
            assert _lib.profile_get(kern)[0] > 0, f"{kern} did not run on the AMP path"
    finally:
        _lib.profile_enable(False)
        _rng.install_replay(None)
        matcher.match_many = solve
        os.environ.pop("MPF_DEVICE_LSA", None)
    _check_amp_outputs(name, out_err, cfg)
    wd = h.criterion.weight_dict
    ref_keys = sorted(k[5:] for k in z if k.startswith("loss."))
    assert sorted(losses) == ref_keys
    bad = {}
    # 2 % per loss through 3-4 bf16 decoder layers; 3 % through the 9 of head_deep (the importance sampling of the mask losses
    # picks its points by the bf16 logits' uncertainty, so a few of the 3 x 112 / 224 points per pair differ from the reference's)
    rtol = 2e-2 if cfg["dec_layers"] <= 4 else 3e-2
    for k in ref_keys:
        want = float(z["loss." + k]) * wd[k]
        got = float(losses[k])
        if abs(got - want) > rtol * abs(want) + 2e-3:
            bad[k] = (got, want)
    assert not bad, f"AMP losses off the fp32 goldens: {bad}"
    np.testing.assert_allclose(float(total), float(z["total_loss"]), rtol=1e-2)
    errs = {}
    for k, v in feats.items():
        errs["grad_feat_" + k] = _rel_l2(_sub(v.grad, 7), z[f"grad_feat_{k}_s7"])
    pg = dict(h.pixel_decoder.named_parameters())
    for k in [k for k in z if k.startswith("grad_pix.") and "_s11" not in k]:
        errs[k] = _rel_l2(pg[k[9:]].grad.cpu().numpy(), z[k])
    dg = dict(h.predictor.named_parameters())
    for k in [k for k in z if k.startswith("grad_dec.")]:
        g = dg[k[9:]].grad
        g = torch.zeros_like(dg[k[9:]]) if g is None else g
        errs[k] = _rel_l2(g.float().cpu().numpy(), z[k])
    # bf16 noise through the decoder puts the sampled tensors at 0.02-0.05: 5e-2 on the mean, 7e-2 on any single tensor.
    # head_noise (point noise: the MP queries attend a few isolated pixels, which bf16 attention resolves worse) sits at
    # 0.06-0.15 — the deviation is autocast's, not a kernel's: on that fixture the fp32 path matches the goldens to < 1e-4 and
    # the natively sequenced and the per-op AMP decoders agree with each other to 1e-4 (tools/amp_noise_check.py)
    each, avg = (7e-2, 5e-2) if cfg.get("noise_scale", 0.0) == 0 else (1.6e-1, 1.0e-1)
    bad = {k: round(e, 4) for k, e in errs.items() if not e <= each}
    assert not bad, f"AMP gradients, relative L2 vs the reference goldens: {bad}\nall: { {k: round(e, 4) for k, e in errs.items()} }"
    mean = float(np.mean(list(errs.values())))
    assert mean <= avg, f"AMP gradients, mean relative L2 {mean:.4f}: { {k: round(e, 4) for k, e in errs.items()} }"


@pytest.mark.parametrize("name,classes,n", [("B_coco_instance_R50_1024", 80, 2), ("C_coco_panoptic_R50_1024", 133, 2)])
def test_head_full_size_configs_B_C(name, classes, n):
    """BASELINE.json configs B and C at full size (1024x1024, R50 channel counts, 100 queries, per-GPU batch 2; C = the
    panoptic class count, its DDP part is tests/test_dist_*): one AMP forward + backward of the whole head.  Size-
    independent properties: finite losses, all 60 keys, every parameter gets a finite gradient, the production
    kernels are the ones that ran, and a second run on the same draws gives bit-identical losses."""
    from mp_former_amd import _lib, _rng
    from mp_former_amd.head import MPFormerHead
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    H = W = 1024
    shapes = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}
    h = MPFormerHead(num_classes=classes, num_queries=100, feature_shapes=shapes).to(dev).train()
    # what the bf16 backbone hands over under autocast: bf16 channel-last planes (the route bench.py times)
    feats = {k: torch.randn(n, H // s, W // s, c, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2) for k, (c, s) in shapes.items()}
    targets = []
    for b in range(n):
        T = 5 + 9 * b
        m = torch.zeros(T, H, W, dtype=torch.bool, device=dev)
        for t in range(T):
            m[t, 60 * t:60 * t + 180, 50 * t:50 * t + 260] = True
        targets.append({"labels": (torch.arange(T, device=dev) * 7) % classes, "masks": m, "boxes": torch.zeros(T, 4, device=dev)})

    def run(seed):
        torch.manual_seed(seed)
        h.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses, _ = h(feats, targets)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        return {k: float(v) for k, v in losses.items()}

    _lib.profile_enable(True)
    l0 = run(11)
    for kern in ("msda_fwd_block", "msda_bwd_bin", "msda_bwd_tile", "attn_fwd_kernel", "attn_bwd_kv", "match_cost_fused",
                 "pair_planes_fwd", "pair_planes_dfeat", "pair_planes_dembed", "mask_loss_fwd", "mask_head_bits", "pool_features", "lsa_kernel", "gemm3", "gemm3_conv_kernel",
                 "gemm3_nt_kernel<conv3x3", "gemm3_tn_kernel<a16>", "gemm3_nt_kernel<b16>", "gn_cl_apply", "gn_cl_bwd_apply"):
        assert _lib.profile_get(kern)[0] > 0, kern
    _lib.profile_enable(False)
    assert len(l0) == 60 and all(np.isfinite(v) for v in l0.values()), l0
    bad = [k for k, p in h.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    assert not bad, bad
    # the stride-4 FPN branch only feeds mask_features: its gradients are those of the mask losses through the mask product
    for k, p in h.named_parameters():
        if any(s in k for s in ("adapter_1", "layer_1", "mask_features")):
            assert float(p.grad.abs().sum()) > 0, f"{k}: zero gradient — the mask losses did not reach the pixel decoder"
    # run-to-run: the forward has no float atomics and no library split reductions left — a second run on the same draws gives
    # the same 60 losses BIT FOR BIT (and therefore the same assignments)
    l1 = run(11)
    assert l1 == l0, [(k, l0[k], l1[k]) for k in l0 if l0[k] != l1[k]][:5]


@pytest.mark.parametrize("name", ["head_deep", "head_cfgA"])
def test_amp_path_own_matching_against_the_fp32_matching(name):
    """The AMP path's OWN Hungarian matching (the other AMP tests pin the assignment to the fp32 pass): cost matrices of the bf16
    autocast forward against those of the fp32 forward (= the reference's, golden-exact) on the same replayed draws.  Reported and
    bounded: the share of (output, image, target) assignments that agree, and the REGRET of the AMP assignment — its cost under
    the fp32 cost matrix relative to the fp32 optimum.  A flipped pair is a near-tie (small regret), never a different matching
    regime; the reference under fp16 autocast has the same property (its matcher also reads autocast logits, matcher.py:105-147)."""
    from scipy.optimize import linear_sum_assignment
    from mp_former_amd import _rng
    from mp_former_amd.matcher import GTMasks
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture(name)
    h = _build(cfg, pp, dp, dev)
    feats = {k: _planes_leaf(v, dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    use_dn = "dn_pred_logits" in z
    matcher = h.criterion.matcher
    orig = matcher.cost_matrices
    got = []
    import os
    os.environ["MPF_DEVICE_LSA"] = "0"
    try:
        matcher.cost_matrices = lambda *a, **k: got.append(orig(*a, **k)) or got[-1]
        for amp in (False, True):
            _rng.install_replay(fifo_to_tags(replay, cfg, use_dn))
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                h(feats, targets)
    finally:
        matcher.cost_matrices = orig
        _rng.install_replay(None)
        os.environ.pop("MPF_DEVICE_LSA", None)
    assert len(got) == 2 and got[0] is not None
    c32, camp = (c.float().cpu().numpy() for c in got)
    counts = GTMasks(targets).counts
    agree, total, regrets = 0, 0, []
    for l in range(c32.shape[0]):
        for b, t in enumerate(counts):
            if t == 0:
                continue
            a32, aamp = c32[l, b, :, :t], camp[l, b, :, :t]
            i1, j1 = linear_sum_assignment(a32)
            i2, j2 = linear_sum_assignment(aamp)
            q1, q2 = np.empty(t, np.int64), np.empty(t, np.int64)
            q1[j1], q2[j2] = i1, i2
            agree += int((q1 == q2).sum())
            total += t
            best = a32[i1, j1].sum()
            regrets.append(float((a32[i2, j2].sum() - best) / max(abs(best), 1e-6)))
    share = agree / max(total, 1)
    print(f"{name}: AMP matching agrees with the fp32 matching on {agree}/{total} targets ({share:.3f}); regret max {max(regrets):.4f} "
          f"mean {np.mean(regrets):.5f}")
    # measured (round 4): head_deep 54 / 60 targets agree, regret max 3.9e-2, mean 3.7e-3; head_cfgA (100 random-init queries, near
    # duplicates of each other) 45 / 70 agree with regret max 8e-3, mean 1.7e-3 — the disagreements are ties at bf16 resolution,
    # so the bound is on the regret, the agreement share is only required not to collapse
    assert share >= 0.5, (share, regrets)
    assert max(regrets) <= 6e-2 and np.mean(regrets) <= 8e-3, regrets


@pytest.mark.parametrize("name", ["head_deep", "head_cfgA", "head_ragged"])
def test_amp_matching_on_bf16_rounded_maps_as_the_reference_samples_them(name):
    """``HungarianMatcher.reference_amp_rounding``: under autocast the reference's matcher point-samples the
    bf16-ROUNDED prediction maps (matcher.py:120-132 after the autocast einsum, mask2former_transformer_decoder.py:1865); the
    default here samples fp32-class logits from the factors.  With the switch on, the cost matrices come from the materialised
    bf16 maps.  Same autocast forward, same replayed draws, both cost matrices: the share of (output, image, target) assignments
    that agree and the regret of one assignment under the other's costs are reported and bounded; the costs themselves agree to
    bf16's resolution."""
    from scipy.optimize import linear_sum_assignment
    from mp_former_amd import _lib, _rng
    from mp_former_amd.matcher import GTMasks
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture(name)
    h = _build(cfg, pp, dp, dev)
    feats = {k: _planes_leaf(v, dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    use_dn = "dn_pred_logits" in z
    matcher = h.criterion.matcher
    orig = matcher.cost_matrices
    got, kern = [], []
    import os
    os.environ["MPF_DEVICE_LSA"] = "0"
    try:
        matcher.cost_matrices = lambda *a, **k: got.append(orig(*a, **k)) or kern.append(_lib.last_kernel()) or got[-1]
        for rounding in (False, True):
            matcher.reference_amp_rounding = rounding
            _rng.install_replay(fifo_to_tags(replay, cfg, use_dn))
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                h(feats, targets)
    finally:
        matcher.cost_matrices = orig
        matcher.reference_amp_rounding = False
        _rng.install_replay(None)
        os.environ.pop("MPF_DEVICE_LSA", None)
    assert len(got) == 2 and got[0] is not None
    cf, cr = (c.float().cpu().numpy() for c in got)
    counts = GTMasks(targets).counts
    agree, total, regrets, rel = 0, 0, [], []
    for l in range(cf.shape[0]):
        for b, t in enumerate(counts):
            if t == 0:
                continue
            a1, a2 = cf[l, b, :, :t], cr[l, b, :, :t]
            rel.append(float(np.abs(a1 - a2).max() / max(np.abs(a1).max(), 1e-6)))
            i1, j1 = linear_sum_assignment(a1)
            i2, j2 = linear_sum_assignment(a2)
            q1, q2 = np.empty(t, np.int64), np.empty(t, np.int64)
            q1[j1], q2[j2] = i1, i2
            agree += int((q1 == q2).sum())
            total += t
            best = a2[i2, j2].sum()
            regrets.append(float((a2[i1, j1].sum() - best) / max(abs(best), 1e-6)))
    share = agree / max(total, 1)
    print(f"{name}: factor-sampled vs bf16-map-sampled matching agree on {agree}/{total} targets ({share:.3f}); cost difference max "
          f"{max(rel):.2e} of the matrix maximum; regret of the default assignment under the reference-rounded costs: max {max(regrets):.4f} "
          f"mean {np.mean(regrets):.5f}")
    assert max(rel) <= 2e-2, rel                    # bf16 rounding of the logits: 2^-9 relative per logit, accumulated over 12 544 points
    assert share >= 0.5 and max(regrets) <= 6e-2, (share, regrets)


def test_amp_forward_is_bit_reproducible_run_to_run():
    """The AMP head forward (pixel decoder outputs and all 6 x (1 + #aux) losses) twice in one process, draws replayed and the
    assignment pinned, with unrelated allocations in between: bit-identical.  (Until the end of round 4 the NCHW route fell back
    to the library's fp32 3x3 convolution, whose result moved run to run — 2e-5 relative on the losses, rarely 3e-3.)"""
    from mp_former_amd import _rng
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_ragged")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}                  # NCHW maps: the route that used the library convolution
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    matcher = h.criterion.matcher
    solve = matcher.match_many
    pin = []
    import os
    os.environ["MPF_DEVICE_LSA"] = "0"

    def run(first):
        tags = fifo_to_tags(replay, cfg, "dn_pred_logits" in z)
        if first:
            matcher.match_many = lambda *a, **k: pin.append(solve(*a, **k)) or pin[-1]
        else:
            matcher.match_many = lambda *a, **k: pin[0]
            tags = {t: d for t, d in tags.items() if not t.startswith("match")}
        _rng.install_replay(tags)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            pdo = h.pixel_decoder.forward_features(feats)
            losses, _ = h(feats, targets)
        out = {"mask_features": pdo[0].float().clone()}
        out.update({f"ms{i}": t.float().clone() for i, t in enumerate(pdo[2])})
        out.update({k: v.float().clone() for k, v in losses.items()})
        return out

    try:
        ref = run(True)
        for it in range(3):
            junk = [torch.full((1 << 20,), 1e3 * (it + 1), device=dev) for _ in range(8)]
            del junk
            cur = run(False)
            diff = [k for k in ref if not torch.equal(ref[k], cur[k])]
            assert not diff, (it, diff[:5])
    finally:
        matcher.match_many = solve
        _rng.install_replay(None)
        os.environ.pop("MPF_DEVICE_LSA", None)
