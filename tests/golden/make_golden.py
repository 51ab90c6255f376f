#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (this container
only; /root/reference is absent on the GPU box).  Usage:

    python tests/golden/make_golden.py [msda] [head] ...

Every fixture is data only: inputs (or the closed-form recipe that regenerates them) and the
outputs / gradients the reference's own python path produced for them.

MSDA fixtures come from the reference's `ms_deform_attn_core_pytorch`
(mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72) and autograd
through it; the known-answer problem is the one of the reference's only test,
mask2former/modeling/pixel_decoder/ops/test.py:24-40,66-89 (same seed, same draw order; the
test draws on the CPU generator and then moves to the device, so the inputs are reproducible
here bit for bit).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def _level_start(shapes):
    hw = shapes[:, 0] * shapes[:, 1]
    return torch.cat((hw.new_zeros((1,)), hw.cumsum(0)[:-1]))


def _ref_fwd_bwd(core, value, shapes, loc, attn, grad_out):
    """Run the reference python path in the dtype of `value`; returns out and the three grads."""
    v = value.clone().requires_grad_(True)
    lo = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    out = core(v, shapes, lo, a)
    out.backward(grad_out)
    return out.detach(), v.grad, lo.grad, a.grad


def closed_form(n, salt):
    """Deterministic pseudo-random numbers in [0,1) that need no stored inputs (numpy only)."""
    i = np.arange(n, dtype=np.float64)
    x = np.sin(i * 12.9898 + salt * 78.233) * 43758.5453
    return x - np.floor(x)


def gen_msda():
    core = R.msda_func().ms_deform_attn_core_pytorch

    # ---- (1) the reference's own known-answer problem: ops/test.py ----------------------------
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = _level_start(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    torch.manual_seed(3)

    def draw(channels):
        value = torch.rand(N, S, M, channels) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        return value, loc, attn

    # check_forward_equal_with_pytorch_double (test.py:35-47)
    value, loc, attn = draw(D)
    g = torch.from_numpy(closed_form(N * Lq * M * D, 1).reshape(N, Lq, M * D))
    out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g)
    _save("msda_testpy_double", value=value.numpy(), shapes=shapes.numpy(), level_start=lsi.numpy(),
          loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(),
          out=out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
    # check_forward_equal_with_pytorch_float (test.py:51-63): next draws of the same generator
    value, loc, attn = draw(D)
    out32 = core(value, shapes, loc, attn)
    out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g)
    _save("msda_testpy_float", value=value.numpy(), shapes=shapes.numpy(), level_start=lsi.numpy(),
          loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(), out_f32=out32.numpy(),
          out=out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
    # check_gradient_numerical (test.py:66-89): channel sizes chosen to hit every backward branch
    for channels in [30, 32, 64, 71, 1025, 2048, 3096]:
        value, loc, attn = draw(channels)
        g = torch.from_numpy(closed_form(N * Lq * M * channels, channels).reshape(N, Lq, M * channels))
        out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g)
        if channels <= 71:
            _save(f"msda_testpy_grad_D{channels}", value=value.numpy(), shapes=shapes.numpy(),
                  level_start=lsi.numpy(), loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(),
                  out=out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
        else:
            # big-D: `value` is replaced by a closed-form recipe so it need not be stored; the
            # scatter result is stored as a strided subsample + its total.
            value = torch.from_numpy(closed_form(N * S * M * channels, 7).reshape(N, S, M, channels) * 0.01)
            out, gv, gl, ga = _ref_fwd_bwd(core, value, shapes, loc.double(), attn.double(), g)
            _save(f"msda_testpy_grad_D{channels}", value_recipe=np.array([7, 0.01]), channels=np.array(channels),
                  shapes=shapes.numpy(), level_start=lsi.numpy(), loc=loc.numpy(), attn=attn.numpy(),
                  grad_out_recipe=np.array([channels]),
                  out=out.numpy(), grad_value_stride7=gv.numpy().reshape(-1)[::7].copy(),
                  grad_value_sum=np.array(gv.sum().item()),
                  grad_loc=gl.numpy(), grad_attn=ga.numpy())

    # ---- (2) scaled-down versions of the BASELINE configs (M=8, D=32, L=3, P=4) ---------------
    torch.manual_seed(20260001)
    cases = {
        # name: (N, level shapes)            -- same aspect / level ratios as configs A-E
        "A_square": (2, [(2, 2), (4, 4), (6, 6)]),
        "E_wide": (1, [(1, 2), (2, 4), (4, 8)]),
        "D_odd": (1, [(1, 3), (3, 5), (5, 9)]),          # odd sizes, S*M not a multiple of 32
    }
    for name, (n, lv) in cases.items():
        M, D, L, P = 8, 32, 3, 4
        shapes = torch.as_tensor(lv, dtype=torch.long)
        lsi = _level_start(shapes)
        S = int((shapes[:, 0] * shapes[:, 1]).sum())
        Lq = S
        value = torch.randn(n, S, M, D)
        # locations: mostly in range, ~15 % outside [0,1] (incl. beyond the -1 / H cut-offs),
        # a few exactly on pixel centres and on the borders
        loc = torch.rand(n, Lq, M, L, P, 2) * 1.5 - 0.25
        loc[0, 0, 0, :, :, :] = 0.0
        loc[0, 0, 1, :, :, :] = 1.0
        loc[0, 1, 0, 0, :, 0] = (torch.arange(P, dtype=torch.float32) + 0.5) / lv[0][1]
        loc[0, 1, 0, 0, :, 1] = 0.5 / lv[0][0]
        loc[0, 1, 1, :, :, :] = -2.0
        loc[0, 1, 2, :, :, :] = 3.0
        attn = torch.softmax(torch.randn(n, Lq, M, L * P), -1).view(n, Lq, M, L, P)
        g = torch.randn(n, Lq, M * D)
        out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g.double())
        _save(f"msda_cfg_{name}", value=value.numpy(), shapes=shapes.numpy(), level_start=lsi.numpy(),
              loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(),
              out=out.numpy(), grad_value=gv.numpy().astype(np.float32), grad_loc=gl.numpy(),
              grad_attn=ga.numpy())


# =================================================================================================
# Head fixtures (F2-F9 of SURVEY.md §8(c)): pixel decoder, MP decoder, matcher, criterion.
# Parameters and inputs are closed-form (det_params.py), so only reference OUTPUTS are stored.
# =================================================================================================
HEAD_CFGS = {
    # name: dict(size, N, counts (GT per image), queries, classes, enc_layers, dec_layers, points)
    "small": dict(size=64, N=2, counts=[3, 1], num_queries=10, num_classes=7, enc_layers=2, dec_layers=3,
                  num_points=112),
    # one image without GT and one with many (padding rows that become all-False; T > pad of others)
    "ragged": dict(size=64, N=3, counts=[0, 5, 2], num_queries=6, num_classes=4, enc_layers=1, dec_layers=4,
                   num_points=64),
    # no GT anywhere -> prepare_for_normal path, dn losses are zeros
    "nogt": dict(size=64, N=1, counts=[0], num_queries=5, num_classes=3, enc_layers=1, dec_layers=2,
                 num_points=32),
    # the shipped depth and class count (6 encoder / 9 decoder layers, 80 classes; 10 outputs -> 60 losses) at 128 x 128
    # point noise on the mask-piloted rows (NOISE_SCALE 0.2: run_50ep_noise scripts of the reference)
    "noise": dict(size=64, N=2, counts=[3, 2], num_queries=8, num_classes=5, enc_layers=1, dec_layers=3,
                  num_points=64, noise_scale=0.2),
    "deep": dict(size=128, N=2, counts=[4, 2], num_queries=50, num_classes=80, enc_layers=6, dec_layers=9,
                 num_points=224, aux_step=11),
    # BASELINE.json config A itself: 256 x 256 crop, batch 1, 100 queries, 80 classes, 12 544 points, 6 + 9 layers.  The
    # ~40 MB of random draws (12 544 matcher points x 10 outputs, 3 x 12 544 + 3 136 loss points per pair) are NOT stored:
    # the generator is seeded with draw_seed right before the decoder runs and the fixture keeps only the call list
    # (function, shape, dtype) + a checksum per draw; the tests regenerate them (conftest.regenerate_draws)
    "cfgA": dict(size=256, N=1, counts=[7], num_queries=100, num_classes=80, enc_layers=6, dec_layers=9,
                 num_points=12544, aux_step=211, mask_step=5, draw_seed=20261002),
}


def build_reference_head(cfg):
    import det_params as DP
    pd = R.pixel_decoder()
    dec = R.decoder()
    shapes = {"res2": R.ShapeSpec(256, stride=4), "res3": R.ShapeSpec(512, stride=8),
              "res4": R.ShapeSpec(1024, stride=16), "res5": R.ShapeSpec(2048, stride=32)}
    pix = pd.MSDeformAttnPixelDecoder(
        shapes, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
        transformer_enc_layers=cfg["enc_layers"], conv_dim=256, mask_dim=256, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    d = dec.MultiScaleMaskedTransformerDecoderMaskDN(
        256, True, num_classes=cfg["num_classes"], hidden_dim=256, num_queries=cfg["num_queries"], nheads=8,
        dim_feedforward=2048, dec_layers=cfg["dec_layers"], pre_norm=False, mask_dim=256,
        enforce_input_project=False, dn_mode="points", head_dn=False, all_lys=True, dn_ratio=0.5,
        dn_label_noise_ratio=0.2)
    pix.load_state_dict(DP.det_state_dict({k: v.shape for k, v in pix.state_dict().items()}, "pix."), strict=False)
    d.load_state_dict(DP.det_state_dict({k: v.shape for k, v in d.state_dict().items()}, "dec."), strict=False)
    m = R.matcher().HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=cfg["num_points"])
    wd = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 5.0}
    wd.update({k + "_dn": v for k, v in list(wd.items())})
    aux = {}
    for i in range(cfg["dec_layers"]):
        aux.update({k + f"_{i}": v for k, v in wd.items()})
    wd.update(aux)
    crit = R.criterion().SetCriterion(cfg["num_classes"], matcher=m, weight_dict=wd, eos_coef=0.1,
                                      losses=["labels", "masks"], num_points=cfg["num_points"],
                                      oversample_ratio=3.0, importance_sample_ratio=0.75)
    crit.train()
    return pix, d, crit, wd


def _sub(t, step):
    return t.detach().reshape(-1)[::step].clone().numpy()


def gen_head_amp(only=None):
    """Second golden family: the reference DECODER run under `torch.autocast("cpu", dtype=torch.bfloat16)` on the
    fp32 pixel-decoder outputs (on the GPU the reference's pixel decoder is fp32 under AMP as well: msdeformattn.py:314,320), same
    parameters, same replayed draws -> tests/golden/head_<name>_amp.npz with the outputs of every decoder layer as float32.  CPU
    autocast is not CUDA autocast (its softmax and LayerNorm stay in bf16 where CUDA's run in fp32), so this is A bf16 evaluation
    of the reference, not THE one a GPU run of the reference would give: the product's autocast path and this family are two
    bf16-noisy evaluations of the same function (tests/test_head_gpu.py reports the distance of each to the fp32 goldens and to
    each other)."""
    gen_head(only=only, amp_family=True)


def gen_head(only=None, amp_family=False):
    import det_params as DP
    for name, cfg in HEAD_CFGS.items():
        if only and name not in only:
            continue
        if amp_family and name == "nogt":
            continue
        torch.manual_seed(1234)
        pix, d, crit, wd = build_reference_head(cfg)
        feats = DP.det_features(cfg["N"], cfg["size"])
        for v in feats.values():
            v.requires_grad_(True)
        targets = DP.det_targets(cfg["N"], cfg["size"], cfg["counts"], cfg["num_classes"])
        out = {}
        import json
        out["pix_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in pix.state_dict().items()}))
        out["dec_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in d.state_dict().items()}))
        out["cfg"] = np.array(json.dumps(cfg))
        # ---- pixel decoder (F3) ----------------------------------------------------------------
        mf, o0, ms = pix.forward_features(feats)
        out["mask_features_s5"] = _sub(mf, 5)
        out["mask_features_absmean"] = np.array(mf.abs().mean().item())
        for i, z in enumerate(ms):
            out[f"multi_scale_{i}_s3"] = _sub(z, 3)
        # ---- decoder with MP queries (F4-F6, F9) -------------------------------------------------
        captured_masks = []
        orig_heads = d.forward_prediction_heads
        if "draw_seed" in cfg:
            torch.manual_seed(cfg["draw_seed"])
        rng_before_decoder = torch.get_rng_state()
        if amp_family:
            dn_args = {"tgt": targets, "scalar": 1, "noise_scale": cfg.get("noise_scale", 0.0)}
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
                dout = d([z.detach() for z in ms], mf.detach(), None, dn_args)
            ms_ = cfg.get("mask_step", 1)
            f32 = lambda t: t.detach().float()          # noqa: E731
            amp = {"cfg": out["cfg"], "pred_logits": f32(dout["pred_logits"]).numpy(),
                   "pred_masks": f32(dout["pred_masks"]).numpy() if ms_ == 1 else _sub(f32(dout["pred_masks"]), ms_)}
            for i, a in enumerate(dout["aux_outputs"]):
                amp[f"aux{i}_pred_logits"] = f32(a["pred_logits"]).numpy()
                amp[f"aux{i}_pred_masks_s3"] = _sub(f32(a["pred_masks"]), cfg.get("aux_step", 3))
            if dout["dn_out"] is not None:
                amp["dn_pred_logits"] = f32(dout["dn_out"]["pred_logits"]).numpy()
                amp["dn_pred_masks"] = f32(dout["dn_out"]["pred_masks"]).numpy()
            amp["dtypes"] = np.array(json.dumps({"pred_logits": str(dout["pred_logits"].dtype), "pred_masks": str(dout["pred_masks"].dtype)}))
            _save(f"head_{name}_amp", **amp)
            print(f"    [amp family] {name}: pred_logits {dout['pred_logits'].dtype}, pred_masks {dout['pred_masks'].dtype}")
            continue
        del rng_before_decoder
        with R.RandCapture() as cap:
            dn_args = {"tgt": targets, "scalar": 1, "noise_scale": cfg.get("noise_scale", 0.0)}
            dout = d(ms, mf, None, dn_args)
            n_dec_draws = len(cap.log)
            losses = crit(dout, targets)
        draws = cap.log
        # drop the mask-noise draws (rand_like of the bool GT rows; NOISE_SCALE = 0 makes them no-ops)
        keep = []
        for i, (fn, t) in enumerate(draws):
            is_mask_noise = (i < n_dec_draws and fn == "rand_like" and t.dim() == 2)
            if not is_mask_noise or cfg.get("noise_scale", 0.0) > 0:
                keep.append((fn, t))
        if "draw_seed" in cfg:
            # the draws are regenerated, not stored: call list of ALL draws (the dropped mask-noise draws advance the generator
            # too), which of them the product replays, and a checksum each; regeneration is verified here, bit for bit
            sys.path.insert(0, os.path.join(HERE, ".."))
            from conftest import regenerate_draws
            again = regenerate_draws(cfg["draw_seed"], cap.spec)
            assert len(again) == len(draws) and all(torch.equal(a, t) for a, (_, t) in zip(again, draws)), "draws do not regenerate"
            kept_idx = [i for i, (fn, t) in enumerate(draws)
                        if not (i < n_dec_draws and fn == "rand_like" and t.dim() == 2) or cfg.get("noise_scale", 0.0) > 0]
            out["rng_spec"] = np.array(json.dumps(cap.spec))
            out["rng_kept"] = np.array(kept_idx, dtype=np.int64)
            out["rng_checksum"] = np.array([float(t.double().sum()) for _, t in draws])
        else:
            for i, (fn, t) in enumerate(keep):
                out[f"rng_{i:03d}_{fn}"] = t.numpy()
        out["n_rng"] = np.array(len(keep))
        ms_ = cfg.get("mask_step", 1)
        out["pred_logits"] = dout["pred_logits"].detach().numpy()
        out["pred_masks"] = dout["pred_masks"].detach().numpy() if ms_ == 1 else _sub(dout["pred_masks"], ms_)
        for i, a in enumerate(dout["aux_outputs"]):
            out[f"aux{i}_pred_logits"] = a["pred_logits"].detach().numpy()
            out[f"aux{i}_pred_masks_s3"] = _sub(a["pred_masks"], cfg.get("aux_step", 3))
        if dout["dn_out"] is not None:
            out["dn_pred_logits"] = dout["dn_out"]["pred_logits"].detach().numpy()
            out["dn_pred_masks"] = dout["dn_out"]["pred_masks"].detach().numpy()
            out["dn_pad_size"] = np.array(dout["dn_out"]["dn_args"]["pad_size"])
            out["dn_max_num"] = np.array(dout["dn_out"]["dn_args"]["max_num"])
        # ---- criterion (F7, F8) ------------------------------------------------------------------
        for k, v in losses.items():
            out["loss." + k] = np.array(float(v))
        total = sum(losses[k] * wd[k] for k in losses if k in wd)
        out["total_loss"] = np.array(float(total))
        total.backward()
        for k, v in feats.items():
            out[f"grad_feat_{k}_s7"] = _sub(v.grad, 7)
            out[f"grad_feat_{k}_norm"] = np.array(v.grad.norm().item())
        sd_p = dict(pix.named_parameters())
        sd_d = dict(d.named_parameters())
        for k in ["transformer.level_embed", "transformer.encoder.layers.0.self_attn.sampling_offsets.bias",
                  "transformer.encoder.layers.0.self_attn.attention_weights.bias", "mask_features.bias",
                  "transformer.encoder.layers.0.norm2.weight"]:
            out["grad_pix." + k] = sd_p[k].grad.numpy()
        out["grad_pix.value_proj0_s11"] = _sub(sd_p["transformer.encoder.layers.0.self_attn.value_proj.weight"].grad, 11)
        for k in ["query_feat.weight", "level_embed.weight", "class_embed.bias", "decoder_norm.weight",
                  "label_enc.weight", "mask_embed.layers.2.bias",
                  "transformer_cross_attention_layers.0.multihead_attn.in_proj_bias"]:
            g = sd_d[k].grad
            out["grad_dec." + k] = (g if g is not None else torch.zeros_like(sd_d[k])).numpy()
        _save(f"head_{name}", **out)
        print(f"    total_loss={float(total):.6f}  rng draws kept={len(keep)} of {len(draws)}")


def main():
    what = sys.argv[1:] or ["msda", "head"]
    torch.set_num_threads(4)
    for w in what:
        print(f"[{w}]")
        if w.startswith("head_amp:"):
            gen_head_amp(only=w.split(":", 1)[1].split(","))
        elif w.startswith("head:"):          # e.g. head:deep — one head fixture only
            gen_head(only=w.split(":", 1)[1].split(","))
        else:
            globals()["gen_" + w]()


if __name__ == "__main__":
    main()
