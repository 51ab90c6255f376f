"""Single-node encoder (mp_former_amd/encoder_fused.py: split-bf16 GEMMs with fused prologues /
epilogues, analytic level_embed gradient) against the layer-by-layer modules it replaces
(msdeformattn.py:92-161 mirror): forward and every gradient, fp32 round-off tolerance."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(enc, srcs, pos, go, fused):
    os.environ["MPF_FUSED_ENCODER"] = "1" if fused else "0"
    try:
        for p in enc.parameters():
            p.grad = None
        xs = [s.detach().clone().requires_grad_(True) for s in srcs]
        mem, _, _ = enc(xs, pos)
        mem.backward(go)
        grads = {n: p.grad.detach().clone() for n, p in enc.named_parameters()}
        return mem.detach(), [x.grad.detach() for x in xs], grads
    finally:
        os.environ.pop("MPF_FUSED_ENCODER", None)


@pytest.mark.parametrize("shapes,batch", [(((4, 4), (8, 8), (16, 16)), 2), (((5, 7), (10, 14), (20, 28)), 1),
                                          (((8, 8), (16, 16), (32, 32)), 3)])
def test_fused_encoder_matches_modules(shapes, batch):
    from mp_former_amd import pixel_decoder as PD
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    enc = PD.MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=3, dim_feedforward=1024,
                                                dropout=0.0, num_feature_levels=3).to(dev).train()
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "sampling_offsets.weight" in n or "attention_weights" in n:
                p.normal_(0, 0.05)
            if "norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
    srcs = [torch.randn(batch, 256, h, w, device=dev) for h, w in shapes]
    pe = PD.PositionEmbeddingSine(128, normalize=True)
    pos = [pe(s) for s in srcs]
    go = torch.randn(batch, sum(h * w for h, w in shapes), 256, device=dev)
    m0, gx0, gp0 = _run(enc, srcs, pos, go, fused=False)
    m1, gx1, gp1 = _run(enc, srcs, pos, go, fused=True)

    errs = {}

    # Two fp32 pipelines are compared across ReLU gates and bilinear-cell boundaries: a pre-activation
    # within round-off of 0 (or a sampling point within round-off of a pixel edge) flips a whole
    # gradient contribution, so single entries may differ at the 1e-2 level while everything else
    # agrees to 1e-6.  Hence: relative L2 error tight, max error loose.
    def close(a, b, what):
        errs[what] = (float((a - b).norm() / (a.norm() + 1e-20)), float((a - b).abs().max()) / (float(a.abs().max()) + 1e-20))

    close(m0, m1, "memory")
    for i, (a, b) in enumerate(zip(gx0, gx1)):
        close(a, b, f"grad src[{i}]")
    assert set(gp0) == set(gp1)
    for n in gp0:
        close(gp0[n], gp1[n], f"grad {n}")
    bad = {k: v for k, v in errs.items() if not (v[0] < 2e-3 and v[1] < 5e-2)}
    assert not bad, f"(relative L2, relative max) errors too large: {bad}"
    assert errs["memory"][1] < 1e-5, errs["memory"]


def test_fused_encoder_is_the_default_path():
    from mp_former_amd import _lib, pixel_decoder as PD
    dev = torch.device("cuda:0")
    enc = PD.MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=1, dim_feedforward=1024,
                                                dropout=0.0, num_feature_levels=3).to(dev)
    srcs = [torch.randn(1, 256, s, s, device=dev) for s in (4, 8, 16)]
    pe = PD.PositionEmbeddingSine(128, normalize=True)
    _lib.lib().mpf_profile_enable(1)
    try:
        enc(srcs, [pe(s) for s in srcs])
        torch.cuda.synchronize()
        n_gemm3 = _lib.profile_get("gemm3")[0]
    finally:
        _lib.lib().mpf_profile_enable(0)
    assert n_gemm3 > 0, "the default encoder path did not launch the split-bf16 GEMM"
    os.environ["MPF_FUSED_ENCODER"] = "1"
    try:
        assert enc._fused_ok(srcs, [pe(s) for s in srcs])
    finally:
        os.environ.pop("MPF_FUSED_ENCODER", None)


def _encoder_at_size(offset_weights):
    """6-layer encoder at config B (levels 32^2 / 64^2 / 128^2, N = 2: R = 43 008 rows) with trained-like parameters: every
    weight jittered away from its initial value, LayerNorm gamma x 8 on layer 2 (the amax slot of a LayerNorm output is an
    upper bound from (gamma, beta): its slack costs the fp16 x 2 split mantissa bits exactly there)."""
    from mp_former_amd import pixel_decoder as PD
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    enc = PD.MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024,
                                                dropout=0.0, num_feature_levels=3).to(dev).train()
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "sampling_offsets.weight" in n:
                p.normal_(0, 0.02) if offset_weights else p.zero_()
            elif "sampling_offsets.bias" in n:
                p.add_(torch.randn_like(p) * 0.5)
            elif "attention_weights" in n:
                p.normal_(0, 0.05)
            elif "norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
                if ".layers.2.norm" in n and n.endswith("weight"):
                    p.mul_(8.0)
            else:
                p.add_(torch.randn_like(p) * 0.3 * float(p.std() if p.dim() > 1 else 0.02))
    shapes, batch = ((32, 32), (64, 64), (128, 128)), 2
    srcs = [torch.randn(batch, 256, h, w, device=dev) for h, w in shapes]
    pe = PD.PositionEmbeddingSine(128, normalize=True)
    pos = [pe(s) for s in srcs]
    go = torch.randn(batch, sum(h * w for h, w in shapes), 256, device=dev)
    return enc, srcs, pos, go


def _run64(enc, srcs, pos, go):
    import copy
    enc64 = copy.deepcopy(enc).double()
    enc64._shape_cache = {}
    xs = [s.detach().double().requires_grad_(True) for s in srcs]
    mem, _, _ = enc64(xs, [p.double() for p in pos])
    mem.backward(go.double())
    return mem.detach(), [x.grad for x in xs], {n: p.grad for n, p in enc64.named_parameters()}


def _errs(a, ref):
    a = a.double()
    return float((a - ref).norm() / (ref.norm() + 1e-300)), float((a - ref).abs().max() / (ref.abs().max() + 1e-300))


def test_fp16x2_encoder_at_config_B_size_against_library_fp32_and_fp64():
    """the fp16 x 2 form of every encoder GEMM (forward, input gradients, weight gradients) on REAL
    activation / gradient tensors at config-B size, next to the library-fp32 run of the layer-by-layer modules on the same
    inputs, with an fp64 run of those modules as the referee.  The sampling geometry is pinned (sampling_offsets.weight = 0:
    the offsets are the jittered biases), so the comparison measures GEMM arithmetic; every other parameter is trained-like
    (jittered, gamma x 8 on one layer)."""
    enc, srcs, pos, go = _encoder_at_size(offset_weights=False)
    mL, gxL, gpL = _run(enc, srcs, pos, go, fused=False)
    mH, gxH, gpH = _run(enc, srcs, pos, go, fused=True)
    m64, gx64, gp64 = _run64(enc, srcs, pos, go)
    rows = {"memory": (_errs(mH, m64), _errs(mL, m64), _errs(mH, mL.double()))}
    for i in range(len(srcs)):
        rows[f"grad src[{i}]"] = (_errs(gxH[i], gx64[i]), _errs(gxL[i], gx64[i]), _errs(gxH[i], gxL[i].double()))
    for n in gp64:
        rows["grad " + n] = (_errs(gpH[n], gp64[n]), _errs(gpL[n], gp64[n]), _errs(gpH[n], gpL[n].double()))
    print()
    for k, (h, l, hl) in rows.items():
        print(f"{k:62s} h2-vs-fp64 {h[0]:.2e} / {h[1]:.2e}   lib-vs-fp64 {l[0]:.2e} / {l[1]:.2e}   h2-vs-lib {hl[0]:.2e}")
    bad = {k: v for k, v in rows.items() if not (v[0][0] <= 2 * v[1][0] + 2e-6 and v[0][1] <= 2 * v[1][1] + 1e-5)}
    assert not bad, f"fp16 x 2 encoder further from the fp64 run than 2 x the library-fp32 modules + (2e-6, 1e-5): {bad}"
    # where no ReLU gate / bilinear cell can flip between the runs (the forward result and the gradients of the last layer's
    # output stage) the distance to fp64 is GEMM arithmetic alone: an absolute bar there.  Everything upstream of a ReLU gate
    # carries ~1e-3 in BOTH fp32 pipelines (a fraction f of flipped gates is a relative L2 error of sqrt(f)): measured
    # 5e-4 .. 1e-3 for the library path as for the fp16 x 2 path, which is why the bar above is relative to the library's.
    for k in ("memory", "grad encoder.layers.5.linear2.weight", "grad encoder.layers.5.linear2.bias",
              "grad encoder.layers.5.norm2.weight", "grad encoder.layers.5.norm2.bias"):
        assert rows[k][0][0] <= 4e-6 and rows[k][0][1] <= 1e-5, (k, rows[k])


def test_fp16x2_encoder_at_config_B_size_with_learned_offsets():
    """Same problem with non-zero sampling_offsets weights: now a sampling point within round-off of a pixel edge may take
    the neighbouring bilinear cell in one pipeline (the value is continuous there, its location gradient is not), so single
    contributions flip in ANY two fp32 pipelines; the bar is the library-fp32 pipeline's own distance from fp64."""
    enc, srcs, pos, go = _encoder_at_size(offset_weights=True)
    m64, gx64, gp64 = _run64(enc, srcs, pos, go)
    mL, gxL, gpL = _run(enc, srcs, pos, go, fused=False)
    mH, gxH, gpH = _run(enc, srcs, pos, go, fused=True)
    assert _errs(mH, m64)[0] <= 4e-6, _errs(mH, m64)
    rows = {}
    for i in range(len(srcs)):
        rows[f"grad src[{i}]"] = (_errs(gxH[i], gx64[i])[0], _errs(gxL[i], gx64[i])[0])
    for n in gp64:
        rows["grad " + n] = (_errs(gpH[n], gp64[n])[0], _errs(gpL[n], gp64[n])[0])
    print("\nlearned offsets, rel-L2 vs fp64 (h2 | library): worst h2 %.2e, worst library %.2e" %
          (max(v[0] for v in rows.values()), max(v[1] for v in rows.values())))
    bad = {k: v for k, v in rows.items() if not v[0] <= 3 * v[1] + 2e-6}
    assert not bad, f"fp16 x 2 encoder further from fp64 than 3 x the library-fp32 pipeline: {bad}"


@pytest.mark.parametrize("shapes,batch,layers", [(((8, 8), (16, 16), (32, 32)), 2, 3), (((5, 7), (10, 14), (20, 28)), 1, 1), (((32, 32), (64, 64), (128, 128)), 2, 6)])
def test_native_encoder_calls_match_the_python_sequenced_ones(shapes, batch, layers):
    """``mpf_encoder_forward`` (csrc/encoder_layer.hip: all layers of the forward as ONE native call writing into one arena)
    against the python-sequenced forward it replaces (one ctypes call and one allocation per kernel): the same kernels with the
    same arguments in the same order — the output and every gradient (the backward reads the arena views) bit for bit, incl.
    config B's size and a single layer (no next-layer q) — the forward bit for bit, the gradients to round-off."""
    from mp_former_amd import _lib, encoder_fused
    from mp_former_amd import pixel_decoder as PD
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    enc = PD.MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=layers, dim_feedforward=1024, dropout=0.0,
                                                num_feature_levels=3).to(dev).train()
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "sampling_offsets.weight" in n or "attention_weights" in n:
                p.normal_(0, 0.05)
    srcs = [torch.randn(batch, 256, h, w, device=dev) for h, w in shapes]
    pe = PD.PositionEmbeddingSine(128, normalize=True)
    pos = [pe(s) for s in srcs]
    go = torch.randn(batch, sum(h * w for h, w in shapes), 256, device=dev)
    res = {}
    _lib.set_option("msda_bwd_sorted", 1)              # (the MSDA backward's grad_value depends on the arrival order of the tile entries otherwise)
    for native in ((False, False), (True, False), (True, True)):
        encoder_fused._NATIVE_FWD, encoder_fused._NATIVE_BWD = native
        try:
            _lib.profile_enable(True)
            res[native] = _run(enc, srcs, pos, go, fused=True)
            n_calls = _lib.profile_get("gemm3_tn_kernel")[0]
            _lib.profile_enable(False)
            assert n_calls > 0
        finally:
            encoder_fused._NATIVE_FWD = encoder_fused._NATIVE_BWD = True
    _lib.set_option("msda_bwd_sorted", 0)
    (m0, gx0, gp0) = res[(False, False)]
    for key in ((True, False), (True, True)):
        m1, gx1, gp1 = res[key]
        assert torch.equal(m0, m1)                       # the forward: bit for bit
        # the backward runs the same kernels with the same arguments on the same saved values (mpf_encoder_backward: one native
        # call, shared temporaries); it is not bit-reproducible run to run by itself (fp32 reassociation in the weight-gradient
        # reductions), so: equal to round-off
        for a, b in zip(gx0, gx1):
            assert float((a - b).norm() / a.norm()) < 1e-6, key
        for n in gp0:
            assert float((gp0[n] - gp1[n]).norm() / (gp0[n].norm() + 1e-30)) < 1e-5, (key, n)


def test_range_guard_flags_a_skewed_operand_and_stays_quiet_otherwise():
    """Run-time range guard of the fp16 x 2 GEMMs (encoder_fused.RANGE_GUARD_EVERY; msdeformattn.py:314,320 promise fp32): on a
    guarded call every GEMM operand of the encoder, forward and backward, is counted by mpf_h2_range_stats.  Ordinary inputs: no
    operand has rows below 2^-18 of its slot beyond the bar, no warning.  One input level scaled by 2^-22 (its rows of the
    value_proj operand x then sit far below the other levels'): the report shows it and the process warns once."""
    import warnings
    from mp_former_amd import encoder_fused as EF
    from mp_former_amd import pixel_decoder as PD
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes, batch = ((8, 8), (16, 16), (32, 32)), 2
    enc = PD.MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=2, dim_feedforward=1024,
                                                dropout=0.0, num_feature_levels=3).to(dev).train()
    pe = PD.PositionEmbeddingSine(128, normalize=True)
    go = torch.randn(batch, sum(h * w for h, w in shapes), 256, device=dev)
    every = EF.RANGE_GUARD_EVERY
    EF.RANGE_GUARD_EVERY = 1
    try:
        def run(scale_level0):
            EF.range_guard_report(sync=True, reset=True)
            srcs = [torch.randn(batch, 256, h, w, device=dev) for h, w in shapes]
            srcs[2] = srcs[2] * scale_level0                       # (levels are flattened coarsest-last: the 32 x 32 map = most rows)
            xs = [s_.requires_grad_(True) for s_ in srcs]
            mem, _, _ = enc(xs, [pe(s_) for s_ in srcs])
            mem.backward(go)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                rep = EF.range_guard_report(sync=True)
            return rep, [str(x.message) for x in w if "fp16 x 2" in str(x.message)]

        rep, warned = run(1.0)
        assert len(rep) == 10, sorted(rep)                          # 5 forward + 5 backward operands
        assert all(v["rows"] > 0 for v in rep.values())
        assert max(v["share"] for v in rep.values()) <= EF.RANGE_GUARD_SHARE and not warned, (rep, warned)
        rep, warned = run(2.0 ** -22)
        x = rep["fwd x (value_proj)"]
        # layer 0's operand x IS the flattened input: the 2 * 1024 rows of the scaled level lie below the line (of 2 layers' rows)
        assert x["below"] == batch * 32 * 32 and x["rows"] == 2 * batch * (64 + 256 + 1024), x
        assert len(warned) == 1 and "value_proj" in warned[0], warned
    finally:
        EF.RANGE_GUARD_EVERY = every
        EF.range_guard_report(sync=True, reset=True)
