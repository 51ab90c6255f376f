import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def closed_form(n, salt):
    """Same closed-form generator as tests/golden/make_golden.py (inputs that are not stored)."""
    i = np.arange(n, dtype=np.float64)
    x = np.sin(i * 12.9898 + salt * 78.233) * 43758.5453
    return x - np.floor(x)


def load_msda_fixture(name):
    """-> dict with inputs (value, shapes, level_start, loc, attn, grad_out) and reference results."""
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "value_recipe" in z:  # big-D test.py cases: regenerate value / grad_out from the recipe
        D = int(z["channels"])
        shapes = z["shapes"]
        S = int((shapes[:, 0] * shapes[:, 1]).sum())
        N, Lq, M = z["loc"].shape[:3]
        salt, scale = z["value_recipe"]
        z["value"] = closed_form(N * S * M * D, salt).reshape(N, S, M, D) * scale
        z["grad_out"] = closed_form(N * Lq * M * D, int(z["grad_out_recipe"][0])).reshape(N, Lq, M * D)
    return z


MSDA_TESTPY = ["msda_testpy_double", "msda_testpy_float"] + [
    f"msda_testpy_grad_D{d}" for d in (30, 32, 64, 71, 1025, 2048, 3096)]
MSDA_CFG = ["msda_cfg_A_square", "msda_cfg_E_wide", "msda_cfg_D_odd"]


@pytest.fixture(scope="session")
def oracle_msda():
    from oracle import msda_oracle
    msda_oracle.build()
    return msda_oracle


def smooth_points(z, eps=1e-4):
    """Boolean [N,Lq,M,L,P] mask of sampling points that are NOT within eps of an integer pixel
    coordinate.  d(out)/d(loc) is discontinuous there (floor()), so a reduced-precision run may
    legitimately land on the other side; grad_loc comparisons in fp32 skip those points."""
    loc = z["loc"].astype(np.float64)
    shapes = z["shapes"].astype(np.float64)
    x = loc[..., 0] * shapes[None, None, None, :, None, 1] - 0.5
    y = loc[..., 1] * shapes[None, None, None, :, None, 0] - 0.5
    H = shapes[None, None, None, :, None, 0]
    W = shapes[None, None, None, :, None, 1]
    near = (np.abs(x - np.round(x)) < eps) | (np.abs(y - np.round(y)) < eps)
    # the in/out-of-range cut-offs (-1, H, W) are integers too, covered by `near`
    del H, W
    return ~near


# ---- head fixtures (tests/golden/head_*.npz) ----------------------------------------------------
HEAD_FIXTURES = ["head_small", "head_ragged", "head_nogt", "head_deep", "head_noise", "head_cfgA"]


def regenerate_draws(seed, spec):
    """The reference's random draws of a fixture that stores a seed instead of the tensors (head_cfgA: ~40 MB of point
    coordinates): torch's CPU generator is seeded and the same calls are made in the same order with the same shapes —
    rand / rand_like are uniform_() on an empty tensor of that shape, randint_like(t, lo, hi) is random_(lo, hi).  The CPU
    generator's stream is a function of (seed, call sequence) for a given torch build; the fixture's per-draw checksums
    catch a build that draws differently."""
    import torch
    state = torch.get_rng_state()
    try:
        torch.manual_seed(int(seed))
        out = []
        for item in spec:
            fn, shape, dtype = item[0], tuple(item[1]), getattr(torch, item[2])
            if fn == "randint_like":
                out.append(torch.empty(shape, dtype=dtype).random_(int(item[3]), int(item[4])))
            else:
                out.append(torch.empty(shape, dtype=dtype).uniform_())
        return out
    finally:
        torch.set_rng_state(state)


def load_head_fixture(name):
    """-> (z, cfg, pix_params, dec_params, features, targets, rng_replay)"""
    import json
    import torch
    sys.path.insert(0, GOLDEN)
    import det_params as DP
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    cfg = json.loads(str(z["cfg"]))
    pix_shapes = json.loads(str(z["pix_keys"]))
    dec_shapes = json.loads(str(z["dec_keys"]))
    pp = DP.det_state_dict(pix_shapes, "pix.")
    dp = DP.det_state_dict(dec_shapes, "dec.")
    feats = DP.det_features(cfg["N"], cfg["size"])
    targets = DP.det_targets(cfg["N"], cfg["size"], cfg["counts"], cfg["num_classes"])
    replay = []
    if "rng_spec" in z:
        draws = regenerate_draws(cfg["draw_seed"], json.loads(str(z["rng_spec"])))
        got = np.array([float(t.double().sum()) for t in draws])
        np.testing.assert_allclose(got, z["rng_checksum"], rtol=0, atol=0,
                                   err_msg="this torch build does not regenerate the fixture's random draws")
        replay = [draws[i] for i in z["rng_kept"]]
    else:
        for i in range(int(z["n_rng"])):
            key = [k for k in z if k.startswith(f"rng_{i:03d}_")][0]
            replay.append(torch.from_numpy(z[key]))
    return z, cfg, pp, dp, feats, targets, replay


def masks_view(t, cfg):
    """pred_masks as the fixture stores them: whole, or every mask_step-th element (head_cfgA)"""
    step = cfg.get("mask_step", 1)
    a = t.detach().float().cpu()
    return a.numpy() if step == 1 else a.reshape(-1)[::step].numpy()


def fifo_to_tags(replay, cfg, use_dn, label_noise=True):
    """Map the reference's FIFO draw order (see make_golden.gen_head) onto the tagged draws of
    mp_former_amd._rng: decoder label noise first, then per output (final, aux 0..): N matcher point
    sets, the loss's oversampled + random points, and the same pair for the MP (`_dn`) loss."""
    q = list(replay)
    tags = {}

    def put(tag):
        tags.setdefault(tag, []).append(q.pop(0))

    noise = use_dn and cfg.get("noise_scale", 0.0) > 0
    if noise:
        put("mp_noise")                       # prepare_for_dn_v5: the first level's rows, before the label noise
    if use_dn and label_noise:
        put("label_prob")
        put("label_new")
    if noise:
        for _ in range(cfg["dec_layers"]):    # gen_mask_dn after every layer (the last one's mask is never used)
            put("mp_noise")
    for suffix in [""] + [f"_{i}" for i in range(cfg["dec_layers"])]:
        for _ in range(cfg["N"]):
            put("match" + suffix)
        put("loss" + suffix + "_over")
        put("loss" + suffix + "_rand")
        if use_dn:
            put("loss_dn" + suffix + "_over")
            put("loss_dn" + suffix + "_rand")
    assert not q, f"{len(q)} unconsumed reference draws"
    return tags
