"""CPU: pin oracle/head_ref.py (pixel decoder, MP decoder, matcher, criterion, and their backward)
against golden vectors produced by the imported reference modules (tests/golden/make_golden.py).
Parameters/inputs are closed-form (tests/golden/det_params.py); random draws are replayed."""
import numpy as np
import pytest
import torch

from conftest import HEAD_FIXTURES, load_head_fixture, masks_view
from oracle import head_ref as O


def _sub(t, step):
    return t.detach().reshape(-1)[::step].numpy()


@pytest.mark.parametrize("name", HEAD_FIXTURES)
def test_head_forward_backward_matches_reference(name):
    torch.set_num_threads(4)
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture(name)
    for v in feats.values():
        v.requires_grad_(True)
    for d in (pp, dp):
        for v in d.values():
            v.requires_grad_(True)
    rng = O.Rng(replay)
    mf, o0, ms, shapes = O.pixel_decoder_forward(pp, feats, enc_layers=cfg["enc_layers"])
    np.testing.assert_allclose(_sub(mf, 5), z["mask_features_s5"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(mf.abs().mean().item(), z["mask_features_absmean"], rtol=1e-4)
    for i, t in enumerate(ms):
        np.testing.assert_allclose(_sub(t, 3), z[f"multi_scale_{i}_s3"], rtol=1e-3, atol=2e-4)

    out = O.decoder_forward(dp, ms, mf, targets, num_queries=cfg["num_queries"], num_classes=cfg["num_classes"],
                            dec_layers=cfg["dec_layers"], scalar=1, noise_scale=cfg.get("noise_scale", 0.0), label_noise_ratio=0.2, rng=rng)
    np.testing.assert_allclose(out["pred_logits"].detach().numpy(), z["pred_logits"], rtol=2e-3, atol=5e-4)
    np.testing.assert_allclose(masks_view(out["pred_masks"], cfg), z["pred_masks"], rtol=2e-3, atol=2e-3)
    for i, a in enumerate(out["aux_outputs"]):
        np.testing.assert_allclose(a["pred_logits"].detach().numpy(), z[f"aux{i}_pred_logits"], rtol=2e-3, atol=5e-4)
        np.testing.assert_allclose(_sub(a["pred_masks"], cfg.get("aux_step", 3)), z[f"aux{i}_pred_masks_s3"], rtol=2e-3, atol=2e-3)
    if "dn_pred_logits" in z:
        assert out["dn_out"]["dn_args"] == {"max_num": int(z["dn_max_num"]), "pad_size": int(z["dn_pad_size"])}
        np.testing.assert_allclose(out["dn_out"]["pred_logits"].detach().numpy(), z["dn_pred_logits"], rtol=2e-3, atol=5e-4)
        np.testing.assert_allclose(out["dn_out"]["pred_masks"].detach().numpy(), z["dn_pred_masks"], rtol=2e-3, atol=2e-3)
    else:
        assert out["dn_out"] is None

    losses, info = O.criterion_forward(out, targets, num_classes=cfg["num_classes"], num_points=cfg["num_points"], rng=rng)
    assert not rng.replay, "oracle drew fewer random tensors than the reference"
    ref_keys = sorted(k[5:] for k in z if k.startswith("loss."))
    assert sorted(losses) == ref_keys
    for k in ref_keys:
        np.testing.assert_allclose(float(losses[k]), float(z["loss." + k]), rtol=2e-4, atol=1e-5, err_msg=k)
    wd = O.weight_dict(cfg["dec_layers"])
    total = sum(v * wd[k] for k, v in losses.items() if k in wd)
    np.testing.assert_allclose(float(total), float(z["total_loss"]), rtol=1e-4)

    total.backward()
    for k, v in feats.items():
        np.testing.assert_allclose(v.grad.norm().item(), z[f"grad_feat_{k}_norm"], rtol=2e-3)
        # six deformable-attention layers deep, a sampling point within rounding distance of a pixel boundary takes the other
        # bilinear cell in one of the two fp32 evaluation orders: isolated elements move by a few 1e-2 of the RMS (0.15 % of
        # them in head_deep), the field as a whole does not (relative L2 below)
        rms = float(z[f"grad_feat_{k}_norm"]) / np.sqrt(v.numel())
        got, want = _sub(v.grad, 7), z[f"grad_feat_{k}_s7"]
        np.testing.assert_allclose(got, want, rtol=5e-3, atol=(2e-3 if cfg["enc_layers"] < 6 else 5e-2) * rms)
        assert np.linalg.norm(got - want) <= 2e-3 * np.linalg.norm(want), k
    for k in [k for k in z if k.startswith("grad_pix.") and "_s11" not in k]:
        g = pp[k[9:]].grad
        np.testing.assert_allclose(g.numpy(), z[k], rtol=5e-3, atol=1e-4 + 2e-3 * np.abs(z[k]).max(), err_msg=k)
    g = pp["transformer.encoder.layers.0.self_attn.value_proj.weight"].grad
    np.testing.assert_allclose(_sub(g, 11), z["grad_pix.value_proj0_s11"], rtol=5e-3,
                               atol=2e-3 * np.abs(z["grad_pix.value_proj0_s11"]).max())
    for k in [k for k in z if k.startswith("grad_dec.")]:
        g = dp[k[9:]].grad
        g = torch.zeros_like(dp[k[9:]]) if g is None else g
        np.testing.assert_allclose(g.numpy(), z[k], rtol=5e-3, atol=1e-4 + 2e-3 * np.abs(z[k]).max(), err_msg=k)


def test_msda_core_torch_matches_c_oracle(oracle_msda):
    """the grid_sample restatement (reference's CPU path) and the plain-C definition agree."""
    from conftest import load_msda_fixture
    z = load_msda_fixture("msda_cfg_A_square")
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    out = O.msda_core(t("value").double(), t("shapes"), t("loc").double(), t("attn").double())
    ref = oracle_msda.msda_forward(z["value"].astype(np.float64), z["shapes"], z["level_start"],
                                   z["loc"].astype(np.float64), z["attn"].astype(np.float64))
    np.testing.assert_allclose(out.numpy(), ref, rtol=1e-10, atol=1e-12)


def test_amp_family_fixtures_are_bf16_evaluations_of_the_same_reference():
    """tests/golden/head_*_amp.npz (make_golden.py head_amp: the reference decoder under CPU bf16 autocast): same outputs, same shapes
    as the fp32 family, produced in bf16, and within bf16 noise of it — the yardstick the GPU tests hold the product's autocast path
    against.  The reference's OWN bf16 evaluation sits 1-13 % (relative L2, worst output) from its fp32 evaluation on these
    fixtures: masked attention thresholds the previous layer's mask logits, so rounding flips attention-mask bits."""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN
    worst = {}
    for name in ("small", "ragged", "noise", "deep", "cfgA"):
        z = np.load(os.path.join(GOLDEN, f"head_{name}.npz"), allow_pickle=True)
        a = np.load(os.path.join(GOLDEN, f"head_{name}_amp.npz"), allow_pickle=True)
        assert json.loads(str(a["dtypes"])) == {"pred_logits": "torch.bfloat16", "pred_masks": "torch.bfloat16"}
        assert str(a["cfg"]) == str(z["cfg"])
        keys = [k for k in a.files if k not in ("cfg", "dtypes")]
        assert "pred_logits" in keys and "pred_masks" in keys and "dn_pred_masks" in keys and "aux0_pred_logits" in keys
        w = 0.0
        for k in keys:
            assert a[k].shape == z[k].shape and a[k].dtype == np.float32, k
            w = max(w, float(np.linalg.norm(a[k].astype(np.float64) - z[k]) / np.linalg.norm(z[k])))
        worst[name] = w
    assert all(1e-3 < w < 0.2 for w in worst.values()), worst
