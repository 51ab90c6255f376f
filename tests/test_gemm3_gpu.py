"""Split-bf16 fp32 GEMM (csrc/gemm3.hip) against fp64: the claim of DESIGN.md section 4 — three bf16 planes per
operand, six products, error NOT above the library's fp32 GEMM — as a test, on ragged shapes with every epilogue,
on the mixed 128/64-column tiling, and for the weight-gradient (NT) form."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_tn(a, a2, w, b, cin, cin2, relu, gate):
    x = a.double()
    if a2 is not None:
        x = x + a2.double()[torch.arange(a.shape[0], device=a.device) % a2.shape[0]]
    y = x @ w.double().t()
    if b is not None:
        y = y + b.double()
    if cin is not None:
        y = y + cin.double()
    if cin2 is not None:
        y = y + cin2.double()
    if relu:
        y = y.relu()
    if gate is not None:
        y = torch.where(gate > 0, y, torch.zeros_like(y))
    return y


@pytest.mark.parametrize("M,N,K,opts", [
    (300, 288, 64, "bias a2 cin relu"), (129, 100, 32, "bias cin cin2"), (128, 256, 256, "bias"),
    (1000, 1024, 256, "bias relu"), (777, 256, 1024, "cin gate"), (5, 4, 32, ""),
])
def test_gemm3_tn_fp32_accuracy(M, N, K, opts):
    from mp_former_amd.gemm3 import gemm3, split_weight
    dev = torch.device("cuda:0")
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev) if "bias" in opts else None
    a2 = torch.randn(7, K, device=dev) if "a2" in opts else None
    cin = torch.randn(M, N, device=dev) if "cin" in opts else None
    cin2 = torch.randn(M, N, device=dev) if "cin2" in opts else None
    gate = torch.randn(M, N, device=dev) if "gate" in opts else None
    relu = "relu" in opts
    ref = _ref_tn(a, a2, w, b, cin, cin2, relu, gate)
    got = gemm3(a, split_weight(w), b, a2=a2, cin=cin, cin2=cin2, gate=gate, relu=relu)
    # the library's fp32 path on the same problem
    x32 = a if a2 is None else a + a2[torch.arange(M, device=dev) % 7]
    lib = x32 @ w.t()
    for t in (b, cin, cin2):
        if t is not None:
            lib = lib + t
    if relu:
        lib = lib.relu()
    if gate is not None:
        lib = torch.where(gate > 0, lib, torch.zeros_like(lib))
    e3 = (got.double() - ref).abs()
    el = (lib.double() - ref).abs()
    scale = float(ref.abs().max()) + 1.0
    assert float(e3.max()) <= 4e-6 * scale, (float(e3.max()), scale)                      # fp32-accurate outright ...
    assert float(e3.mean()) <= 1.25 * float(el.mean()) + 1e-9, (float(e3.mean()), float(el.mean()))   # ... and not worse than the library


@pytest.mark.parametrize("N,K", [(256, 64), (1024, 32), (256, 1024)])
def test_gemm3_tn_mixed_tiles_bit_equal(N, K):
    """The 128 x 64 tiles of the last partial round (gemm3_tn_mixed_kernel) accumulate every output element in the same
    order as the 128 x 128 tiles: identical bits."""
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import gemm3, split_weight
    dev = torch.device("cuda:0")
    torch.manual_seed(N + K)
    slots = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    tiles_n = N // 128
    M = 128 * (slots // tiles_n + max(1, slots // (4 * tiles_n))) - 37        # one full round + a quarter round, ragged last block
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    cin = torch.randn(M, N, device=dev)
    planes = split_weight(w)
    _lib.set_option("gemm3_two_pass", 0)          # (N % 256 == 0 takes the 128 x 256 tiles by default)
    try:
        got = gemm3(a, planes, b, cin=cin, relu=True)
        assert _lib.last_kernel() == "gemm3_tn_kernel<128+64>", _lib.last_kernel()
        _lib.set_option("gemm3_mixed_tiles", 0)
        try:
            want = gemm3(a, planes, b, cin=cin, relu=True)
            assert _lib.last_kernel() == "gemm3_tn_kernel<128>"
        finally:
            _lib.set_option("gemm3_mixed_tiles", 1)
    finally:
        _lib.set_option("gemm3_two_pass", 256)
    assert torch.equal(got, want)
    ref = (a.double() @ w.double().t() + b.double() + cin.double()).relu()
    assert float((got.double() - ref).abs().max()) <= 4e-6 * (float(ref.abs().max()) + 1.0) * max(1.0, (K / 256) ** 0.5)


@pytest.mark.parametrize("M,N,K", [(1000, 256, 64), (43008, 1024, 256), (4099, 512, 1024), (128, 256, 32)])
def test_gemm3_tn_two_pass_tiles_bit_equal(M, N, K):
    """The 128 x 256 tile (two 128-column passes over one A image per K step, gemm3_tn2_kernel) does the same products in
    the same order per output element as the 128 x 128 tile: identical bits, with every epilogue operand and a ragged
    last row block."""
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import gemm3, split_weight
    dev = torch.device("cuda:0")
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    cin = torch.randn(M, N, device=dev)
    gate = torch.randn(M, N, device=dev)
    planes = split_weight(w)
    _lib.set_option("gemm3_two_pass", 0)
    want = gemm3(a, planes, b, cin=cin, gate=gate, relu=True)
    assert "x256" not in _lib.last_kernel()
    _lib.set_option("gemm3_two_pass", 256)
    try:
        for rows in (128, 96):          # 96-row blocks: the first two waves stage rows 64-95 of the A tile
            _lib.set_option("gemm3_two_pass_rows", rows)
            got = gemm3(a, planes, b, cin=cin, gate=gate, relu=True)
            assert _lib.last_kernel() == f"gemm3_tn_kernel<{rows}x256>", _lib.last_kernel()
            assert torch.equal(got, want), rows
    finally:
        _lib.set_option("gemm3_two_pass_rows", 0)
        _lib.set_option("gemm3_two_pass", 256)
    ref = torch.where(gate > 0, (a.double() @ w.double().t() + b.double() + cin.double()).relu(), torch.zeros((), dtype=torch.float64, device=dev))
    assert float((got.double() - ref).abs().max()) <= 4e-6 * (float(ref.abs().max()) + 1.0) * max(1.0, (K / 256) ** 0.5)


@pytest.mark.parametrize("R,M,N,rps", [(4096, 256, 256, 512), (3000, 256, 1024, 512), (2048, 288, 256, 256), (1000, 100, 36, 128)])
def test_gemm3_nt_weight_gradient_accuracy(R, M, N, rps):
    from mp_former_amd.gemm3 import gemm3_nt, nt_reduce
    dev = torch.device("cuda:0")
    torch.manual_seed(R + M + N)
    g = torch.randn(R, M, device=dev)
    x = torch.randn(R, N, device=dev)
    c, ca, _ = gemm3_nt(g, x, rps, want_csum_a=True)
    dw, db = nt_reduce(c, ca)
    ref = g.double().t() @ x.double()
    lib = g.t() @ x
    e3, el = (dw.double() - ref).abs(), (lib.double() - ref).abs()
    assert float(e3.max()) <= 1e-5 * (float(ref.abs().max()) + 1.0)
    assert float(e3.mean()) <= 1.25 * float(el.mean()) + 1e-9, (float(e3.mean()), float(el.mean()))
    torch.testing.assert_close(db.double(), g.double().sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("R,rps,strided", [(4096, 512, False), (3000, 352, True), (43008, 3584, False)])
def test_gemm3_nt_grouped_bit_equal_to_single_problems(R, rps, strided):
    """The four weight gradients of an encoder layer in one launch: same tiles, same split boundaries, same order of products and
    of the split sum as four ``gemm3_nt`` + ``nt_reduce`` calls -> bit-identical; ragged sizes and row strides included."""
    from mp_former_amd.gemm3 import gemm3_nt, gemm3_nt_grouped, nt_reduce
    dev = torch.device("cuda:0")
    torch.manual_seed(R)
    shapes = [(256, 1024), (1024, 256), (256, 256), (100, 36)] if R < 10000 else [(256, 1024), (1024, 256), (256, 256), (256, 256)]
    pairs = []
    for (M, N) in shapes:
        if strided:
            g = torch.randn(R, M + 8, device=dev)[:, 4:4 + M]
            x = torch.randn(R, N + 4, device=dev)[:, :N]
        else:
            g, x = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev)
        pairs.append((g, x))
    got = gemm3_nt_grouped(pairs, rps)
    for (g, x), (dw, db) in zip(pairs, got):
        c, ca, _ = gemm3_nt(g, x, rps, want_csum_a=True)
        N = x.shape[1]
        same_tile = -N % 128 <= -N % 96           # the single-problem entry takes 96-column tiles (another MFMA shape) when they waste less
        if same_tile and c[0].numel() % 4 == 0 and ca[0].numel() % 4 == 0:
            wdw, wdb = nt_reduce(c, ca)
            assert torch.equal(dw, wdw) and torch.equal(db, wdb), (g.shape, x.shape)
        ref = g.double().t() @ x.double()
        assert float((dw.double() - ref).abs().max()) <= 1e-5 * (float(ref.abs().max()) + 1.0) * max(1.0, (R / 4096) ** 0.5)
        torch.testing.assert_close(db.double(), g.double().sum(0), rtol=1e-5, atol=1e-4 * max(1.0, (R / 4096) ** 0.5))


@pytest.mark.parametrize("M,N,K,adt,cdt", [(300, 288, 64, "bf16", "f32"), (4099, 256, 512, "bf16", "f32"), (1000, 256, 256, "f32", "bf16"),
                                            (129, 128, 2048, "bf16", "bf16"), (640, 96, 96, "bf16", "f32")])
def test_gemm3_tn_bf16_operand_and_result(M, N, K, adt, cdt):
    """mpf_gemm3_tn_ex: a bf16 A is its own first plane (three products), a bf16 result is the fp32 result rounded once."""
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import gemm3, gemm3_ex, split_weight
    dev = torch.device("cuda:0")
    torch.manual_seed(M + K)
    T = {"bf16": torch.bfloat16, "f32": torch.float32}
    a = torch.randn(M, K, device=dev).to(T[adt])
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    cin = torch.randn(M, N, device=dev)
    planes = split_weight(w)
    got = gemm3_ex(a, planes, b, cin=cin, relu=True, out_dtype=T[cdt])
    assert _lib.last_kernel() in (("gemm3_tn_kernel<a16>",) if adt == "bf16" else ("gemm3_tn_kernel<c16>", "gemm3_tn_kernel<96x256>", "gemm3_tn_kernel<128x256>")), _lib.last_kernel()
    ref = (a.double() @ w.double().t() + b.double() + cin.double()).relu()
    full = gemm3(a.float(), planes, b, cin=cin, relu=True)          # the six-product kernel on the same values
    scale = float(ref.abs().max()) + 1.0
    if cdt == "f32":
        assert float((got.double() - ref).abs().max()) <= 4e-6 * scale * max(1.0, (K / 256) ** 0.5)
        assert float((got - full).abs().max()) <= 4e-6 * scale * max(1.0, (K / 256) ** 0.5)
    else:
        assert got.dtype == torch.bfloat16
        assert torch.equal(got, full.to(torch.bfloat16)) or float((got.float() - full).abs().max()) <= 2.0 ** -8 * scale


@pytest.mark.parametrize("R,M,N,rps,which", [(4096, 256, 256, 512, "a"), (3000, 256, 1024, 352, "b"), (2048, 100, 128, 256, "b"),
                                              (1000, 288, 2048, 128, "a")])
def test_gemm3_nt_one_bf16_operand(R, M, N, rps, which):
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import gemm3_nt_ex, nt_reduce
    dev = torch.device("cuda:0")
    torch.manual_seed(R + N)
    g = torch.randn(R, M, device=dev)
    x = torch.randn(R, N, device=dev)
    if which == "a":
        g = g.to(torch.bfloat16)
    else:
        x = x.to(torch.bfloat16)
    c, ca = gemm3_nt_ex(g, x, rps, want_csum_a=True)
    assert _lib.last_kernel() == f"gemm3_nt_kernel<{which}16>", _lib.last_kernel()
    dw, db = nt_reduce(c, ca)
    ref = g.double().t() @ x.double()
    assert float((dw.double() - ref).abs().max()) <= 1e-5 * (float(ref.abs().max()) + 1.0)
    torch.testing.assert_close(db.double(), g.double().sum(0), rtol=1e-5, atol=1e-4)


# ---- the fp16 x 2 form (two pieces per operand, three products, amax-scaled) --------------------------------------------------
def _heavy(M, K, dev, small_rows=1.0):
    """activations with a 10x spread of row scales, 30x outliers, and (optionally) half of the rows `small_rows` times smaller"""
    a = torch.randn(M, K, device=dev) * (1 + 9 * torch.rand(M, 1, device=dev))
    a[::7, ::13] *= 30
    if small_rows != 1.0:
        a[M // 2:] *= small_rows
    return a


@pytest.mark.parametrize("M,N,K", [(4096, 256, 256), (2000, 1024, 256), (3000, 256, 1024), (700, 256, 288), (1500, 288, 256), (333, 128, 64),
                                   (257, 100, 96)])
def test_gemm3_tn_h2_accuracy_not_worse_than_library_fp32(M, N, K):
    """Error against fp64, relative to sum |a||b| per output element: the fp16 x 2 form is held to the library fp32 GEMM's
    (max and mean), with bias / addends / ReLU gate in the epilogue and the recorded output amax exact."""
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import amax, amax_slots, amax_value, gemm3_h2, split_weights_grouped_h2
    dev = torch.device("cuda:0")
    torch.manual_seed(M + N + K)
    a = _heavy(M, K, dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b, cin, gate = torch.randn(N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, N, device=dev)
    (pl, wam), = split_weights_grouped_h2([([w], False)])
    assert float(amax_value(wam)) == float(w.abs().max())
    am = amax(a)
    assert float(amax_value(am)) == float(a.abs().max())
    oam = amax_slots(1, dev)[0]
    got = gemm3_h2(a, am, pl, wam, b, cin=cin, gate=gate, out_amax=oam)
    assert "h2" in _lib.last_kernel()
    assert float(amax_value(oam)) == float(got.abs().max())
    ref = torch.where(gate > 0, a.double() @ w.double().t() + b.double() + cin.double(), torch.zeros((), dtype=torch.float64, device=dev))
    lib = torch.where(gate > 0, torch.addmm(b, a, w.t()) + cin, torch.zeros((), device=dev))
    den = a.double().abs() @ w.double().abs().t() + b.double().abs() + cin.double().abs()
    e2, el = (got.double() - ref).abs() / den, (lib.double() - ref).abs() / den
    assert float(e2.max()) <= 1.25 * float(el.max()) + 1e-9, (float(e2.max()), float(el.max()))
    assert float(e2.mean()) <= 1.1 * float(el.mean()) + 1e-12, (float(e2.mean()), float(el.mean()))


def test_gemm3_tn_h2_dynamic_range_and_special_values():
    """Rows 2^-13 below the operand's largest magnitude keep the library's accuracy (the second piece stays a normal fp16
    number down to 2^-18); rows 2^-20 below are still far better than one fp16 / bf16 piece; an all-zero operand gives
    exactly the bias; a NaN in the operand reaches the output (and its amax slot)."""
    from mp_former_amd.gemm3 import amax, amax_slots, amax_value, gemm3_h2, split_weights_grouped_h2
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    M, N, K = 2048, 256, 256
    w = torch.randn(N, K, device=dev) / 16
    b = torch.randn(N, device=dev)
    (pl, wam), = split_weights_grouped_h2([([w], False)])
    for small, bound in ((2.0 ** -13, 5e-7), (2.0 ** -20, 6e-6)):        # measured 1.6e-7 / 2.1e-6 (library fp32: 2.9e-7)
        a = torch.randn(M, K, device=dev)
        a[M // 2:] *= small
        got = gemm3_h2(a, amax(a), pl, wam)
        ref = a.double() @ w.double().t()
        den = a.double().abs() @ w.double().abs().t()
        e = ((got.double() - ref).abs() / den)[M // 2:]
        assert float(e.max()) <= bound, (small, float(e.max()))
    z = torch.zeros(M, K, device=dev)
    got = gemm3_h2(z, amax(z), pl, wam, b)
    assert torch.equal(got, b.expand(M, N))
    a = torch.randn(M, K, device=dev)
    a[5, 7] = float("nan")
    oam = amax_slots(1, dev)[0]
    got = gemm3_h2(a, amax(a), pl, wam, b, out_amax=oam)
    assert bool(torch.isnan(got[5]).all()) and not bool(torch.isnan(got[6]).any())


@pytest.mark.parametrize("M,F,C", [(3000, 1024, 256), (1000, 128, 256), (43008, 1024, 256), (130, 384, 64)])
def test_gemm3_tn_h2_bit_mask_gate_is_the_activation_gate(M, F, C):
    """``mpf_gemm3_tn_h2_bits`` (the encoder's linear1 -> ReLU -> linear2 backward, msdeformattn.py:123-127): the product with
    ReLU also returns the mask of (C > 0), one bit per element in numpy's little bit order; the input-gradient product gated by
    that mask is bit-identical to the one gated by the saved activation (two-pass and one-pass tiles, ragged last row tile)."""
    import numpy as np
    from mp_former_amd.gemm3 import amax, amax_slots, gemm3_h2, gemm3_h2_bits, split_weights_grouped_h2
    dev = torch.device("cuda:0")
    torch.manual_seed(M + F)
    w1 = torch.randn(F, C, device=dev) / 16
    b1 = torch.randn(F, device=dev) * 0.1
    x = torch.randn(M, C, device=dev)
    g = torch.randn(M, C, device=dev)
    add = torch.randn(M, F, device=dev)
    (p1, a1), (p1t, a1t) = split_weights_grouped_h2([([w1], False), ([w1], True)])       # linear1; and W1 as the operand of g . W1
    h_ref = gemm3_h2(x, amax(x), p1, a1, b1, relu=True)
    oam = amax_slots(2, dev)
    h, bits = gemm3_h2_bits(x, amax(x), p1, a1, b1, relu=True, out_amax=oam[0], want_bits=True)
    assert torch.equal(h, h_ref)
    assert bits.shape == (M, F // 8) and bits.dtype == torch.uint8
    want = np.packbits((h_ref > 0).cpu().numpy(), axis=1, bitorder="little")
    assert np.array_equal(bits.cpu().numpy(), want)
    assert 0.2 < float((h_ref > 0).float().mean()) < 0.8
    # dh = (g . W1^T ... here: any product with N = F) gated: activation gate against bit gate, with an addend in the epilogue
    (p2t, a2t), = split_weights_grouped_h2([([torch.randn(C, F, device=dev) / 16], True)])  # [F, C] planes: g [M, C] -> [M, F]
    d_ref = gemm3_h2(g, amax(g), p2t, a2t, cin=add, gate=h_ref, out_amax=oam[1])
    d_bits = gemm3_h2_bits(g, amax(g), p2t, a2t, cin=add, gate_bits=bits)
    assert torch.equal(d_bits, d_ref)
    assert bool(((d_ref == 0) == (h_ref <= 0)).all())
    del p1t, a1t


@pytest.mark.parametrize("R,M,N,rps", [(4096, 256, 256, 512), (3000, 256, 1024, 512), (2048, 256, 288, 256), (1000, 100, 36, 128)])
def test_gemm3_nt_h2_weight_gradient_accuracy(R, M, N, rps):
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import amax, gemm3_nt
    dev = torch.device("cuda:0")
    torch.manual_seed(R + M + N)
    g = _heavy(R, M, dev) * 1e-4
    x = _heavy(R, N, dev)
    c, ca, cb = gemm3_nt(g, x, rps, want_csum_a=True, want_csum_b=True, amax_ab=(amax(g), amax(x)))
    assert "h2" in _lib.last_kernel()
    dw = c.sum(0)
    ref = g.double().t() @ x.double()
    lib = g.t() @ x
    den = g.double().abs().t() @ x.double().abs()
    e2, el = (dw.double() - ref).abs() / den, (lib.double() - ref).abs() / den
    assert float(e2.max()) <= 1.25 * float(el.max()) + 1e-9, (float(e2.max()), float(el.max()))
    assert float(e2.mean()) <= 1.1 * float(el.mean()) + 1e-12, (float(e2.mean()), float(el.mean()))
    torch.testing.assert_close(ca.sum(0).double(), g.double().sum(0), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(cb.sum(0).double(), x.double().sum(0), rtol=1e-5, atol=1e-3)
    # transposed output (the 288-wide gradient of the encoder)
    ct, _, _ = gemm3_nt(g, x, rps, transpose_out=True, amax_ab=(amax(g), amax(x)))
    assert torch.equal(ct.transpose(1, 2), c)


def test_gemm3_nt_grouped_h2_bit_equal_to_single_problems():
    from mp_former_amd.gemm3 import amax, gemm3_nt, gemm3_nt_grouped, nt_reduce
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    R, rps = 6000, 512
    pairs = [(_heavy(R, M, dev) * 1e-3, _heavy(R, N, dev)) for (M, N) in [(256, 1024), (1024, 256), (256, 256), (256, 256)]]
    ams = [(amax(g), amax(x)) for g, x in pairs]
    got = gemm3_nt_grouped(pairs, rps, ams)
    for (g, x), am, (dw, db) in zip(pairs, ams, got):
        c, ca, _ = gemm3_nt(g, x, rps, want_csum_a=True, amax_ab=am)
        wdw, wdb = nt_reduce(c, ca)
        assert torch.equal(dw, wdw) and torch.equal(db, wdb)


@pytest.mark.parametrize("M,N", [(43008, 256), (43008, 1024), (1000, 256), (77, 512), (16, 256), (2731, 1024), (16800, 256)])
def test_gemm3_ws_kernel_bit_identical_to_the_tiled_kernel(M, N):
    """The weight-stationary K = 256 kernel (csrc/gemm3_ws.h: weight fragments in registers, A tiles by DMA, one barrier per 64
    rows) against the two-pass tiled kernel it replaces (`gemm3_ws=0`): same split, same product order per output element —
    every variant of the epilogue (bias, two addends, ReLU, fp32 gate, bit gate, bit-mask output, output amax) bit for bit,
    incl. row counts that are no multiple of the 64-row tile / the 16-row granule and a strided A."""
    import numpy as np
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import amax, amax_slots, amax_value, gemm3_h2, gemm3_h2_bits, split_weights_grouped_h2
    dev = torch.device("cuda:0")
    torch.manual_seed(M + N)
    K = 256
    a_full = _heavy(M, K + 64, dev)
    a = a_full[:, :K]                                   # row stride 320 floats (16-byte aligned rows)
    w = torch.randn(N, K, device=dev) / 16
    b, cin, cin2, gate = torch.randn(N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, N, device=dev)
    (pl, wam), = split_weights_grouped_h2([([w], False)])
    am = amax(a.contiguous())
    res = {}
    for ws in (0, 1):
        _lib.set_option("gemm3_ws", 256 if ws else 0)       # (1: every N % 256 == 0; the default takes it from N = 512)
        try:
            oam = amax_slots(3, dev)
            r = [gemm3_h2(a, am, pl, wam)]
            assert ("ws" in _lib.last_kernel()) == bool(ws), _lib.last_kernel()
            r.append(gemm3_h2(a, am, pl, wam, b, cin=cin, cin2=cin2, out_amax=oam[0]))          # (two addends: always the tiled kernel)
            r.insert(2, gemm3_h2(a, am, pl, wam, b, relu=True, gate=gate, out_amax=oam[1]))     # (an fp32 gate: always the tiled kernel)
            assert "ws" not in _lib.last_kernel()
            r.append(gemm3_h2(a, am, pl, wam, b, cin=cin, relu=True))
            h, bits = gemm3_h2_bits(a, am, pl, wam, b, relu=True, out_amax=oam[2], want_bits=True)
            r += [h, bits, gemm3_h2_bits(a, am, pl, wam, cin=cin, gate_bits=bits)]
            assert ("ws" in _lib.last_kernel()) == bool(ws), _lib.last_kernel()
            r += [amax_value(oam[i]).clone() for i in range(3)]
            res[ws] = r
        finally:
            _lib.set_option("gemm3_ws", 512)
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    assert float(res[1][7]) == float(res[1][1].abs().max())
    want = np.packbits((res[1][4] > 0).cpu().numpy(), axis=1, bitorder="little")
    assert np.array_equal(res[1][5].cpu().numpy(), want)
    ref = a.double() @ w.double().t()
    den = a.double().abs() @ w.double().abs().t()
    assert float(((res[1][0].double() - ref).abs() / den).max()) < 1e-6


@pytest.mark.parametrize("M,N,K", [(43008, 256, 256), (43008, 256, 1024), (5000, 256, 288), (2731, 512, 256), (2048, 1024, 64), (21511, 256, 32)])
def test_gemm3_tn3_kernel_bit_identical_to_the_two_pass_kernel(M, N, K):
    """The 192 x 256-tile kernel (csrc/gemm3_tn3.h: one 8-wave workgroup per CU, double-buffered plane images, B planes by DMA one K
    step ahead, one barrier per K step) against the two-pass tiled kernel it replaces (`gemm3_tn3=0`): same split, same product
    order per output element — every variant of the epilogue bit for bit, incl. row counts that are no multiple of 192 / 16,
    a strided A, K = 32 (one step) and K = 288."""
    import numpy as np
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import amax, amax_slots, amax_value, gemm3_h2, gemm3_h2_bits, split_weights_grouped_h2
    dev = torch.device("cuda:0")
    torch.manual_seed(M + N + K)
    a_full = _heavy(M, K + 64, dev)
    a = a_full[:, :K]
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b, cin, cin2, gate = torch.randn(N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(M, N, device=dev)
    (pl, wam), = split_weights_grouped_h2([([w], False)])
    am = amax(a.contiguous())
    res = {}
    _lib.set_option("gemm3_ws", 0)
    try:
        for t3 in (0, 1, 2):                                # two-pass kernel | 192-row tiles | 176-row tiles where they cost no extra round
            _lib.set_option("gemm3_tn3", 2 * min(t3, 1))    # (2: every eligible shape, not only those whose tiles fill the chip)
            _lib.set_option("gemm3_tn3_176", int(t3 == 2))
            oam = amax_slots(3, dev)
            r = [gemm3_h2(a, am, pl, wam)]
            assert ("192x256" in _lib.last_kernel()) == bool(t3), _lib.last_kernel()
            r.append(gemm3_h2(a, am, pl, wam, b, cin=cin, cin2=cin2, out_amax=oam[0]))
            r.append(gemm3_h2(a, am, pl, wam, b, relu=True, gate=gate, out_amax=oam[1]))
            r.append(gemm3_h2(a, am, pl, wam, b, cin=cin, relu=True))
            h, bits = gemm3_h2_bits(a, am, pl, wam, b, relu=True, out_amax=oam[2], want_bits=True)
            r += [h, bits, gemm3_h2_bits(a, am, pl, wam, cin=cin, gate_bits=bits)]
            assert ("192x256" in _lib.last_kernel()) == bool(t3), _lib.last_kernel()
            r += [amax_value(oam[i]).clone() for i in range(3)]
            res[t3] = r
            if t3 == 2 and M == 43008:
                assert "192x256:176" in _lib.last_kernel(), _lib.last_kernel()      # 245 tiles of 176 rows: one round, like 224 of 192
    finally:
        _lib.set_option("gemm3_ws", 512)
        _lib.set_option("gemm3_tn3", 1)
        _lib.set_option("gemm3_tn3_176", 1)
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    for x, y in zip(res[0], res[2]):
        assert torch.equal(x, y)
    assert float(res[1][7]) == float(res[1][1].abs().max())
    want = np.packbits((res[1][4] > 0).cpu().numpy(), axis=1, bitorder="little")
    assert np.array_equal(res[1][5].cpu().numpy(), want)
    ref = a.double() @ w.double().t()
    den = a.double().abs() @ w.double().abs().t()
    assert float(((res[1][0].double() - ref).abs() / den).max()) < 1e-6


def test_h2_range_stats_counts_rows_below_the_slot():
    """mpf_h2_range_stats (the run-time guard of the fp16 x 2 form) on a constructed skewed operand: rows
    scaled 2^-10 / 2^-19 / 2^-25 below the largest and all-zero rows -> counters = (non-zero rows, rows below 2^-18 of the slot),
    exactly; accumulates over calls; a row stride larger than the row is honoured."""
    from mp_former_amd import _lib
    from mp_former_amd.gemm3 import amax
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    rows, cols = 5000, 256
    a = torch.randn(rows, 320, device=dev)[:, :cols]                 # lda = 320
    a[0, 0] = 8.0                                                     # the operand's largest magnitude
    scale = torch.ones(rows, device=dev)
    scale[100:400] = 2.0 ** -10
    scale[400:1000] = 2.0 ** -19
    scale[1000:1100] = 2.0 ** -25
    scale[1100:1500] = 0.0
    a = (a * scale[:, None])
    a = torch.empty(rows, 320, device=dev).copy_(torch.nn.functional.pad(a, (0, 64)))[:, :cols]
    slot = amax(a.contiguous())
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)

    def run():
        with _lib.device_guard(dev):
            _lib.check(_lib.lib().mpf_h2_range_stats(a.data_ptr(), rows, cols, a.stride(0), slot.data_ptr(), 18, cnt.data_ptr(),
                                                     _lib.stream_ptr(dev)), "mpf_h2_range_stats")
    run()
    rmax = a.abs().amax(1)
    ref = float(a.abs().max())
    want = (int((rmax > 0).sum()), int(((rmax > 0) & (rmax < ref * 2.0 ** -18)).sum()))
    assert want[0] == rows - 400 and want[1] >= 690            # (the 2^-19 and 2^-25 rows; a few 2^-19 rows' maxima may sit above the line)
    assert tuple(cnt.tolist()) == want
    run()
    assert tuple(cnt.tolist()) == (2 * want[0], 2 * want[1])
    assert _lib.lib().mpf_h2_range_stats(a.data_ptr(), rows, 255, a.stride(0), slot.data_ptr(), 18, cnt.data_ptr(), _lib.stream_ptr(dev)) != 0


@pytest.mark.parametrize("cout", [72, 100])
def test_linear_tall_widths_outside_the_kernels_take_the_library_in_both_directions(cout):
    """out_features % 4 == 0 but % 32 != 0 (heads * levels * points = 72, any 4k width) passed the forward's shape
    check and then failed in backward, whose dx = g @ W contracts over out_features.  Such widths — and non-contiguous
    weights / non-fp32 biases — must take F.linear for the whole op: forward AND backward run and match fp64."""
    from mp_former_amd.linear import linear_tall, native_ok
    dev = torch.device("cuda:0")
    torch.manual_seed(cout)
    x = torch.randn(2, 2048, 256, device=dev, requires_grad=True)
    w = (torch.randn(cout, 256, device=dev) / 16).requires_grad_(True)
    b = torch.randn(cout, device=dev, requires_grad=True)
    assert not native_ok(x, w, b)
    y = linear_tall(x, w, b)
    g = torch.randn_like(y)
    y.backward(g)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64, b64)
    y64.backward(g.double())
    for a_, r_ in ((y, y64), (x.grad, x64.grad), (w.grad, w64.grad), (b.grad, b64.grad)):
        assert float((a_.double() - r_).abs().max()) <= 1e-5 * float(r_.abs().max()) + 1e-6
    # a transposed (non-contiguous) weight view and a half-precision bias are refused by the native route as well
    wt = torch.randn(256, 256, device=dev).t()
    assert not native_ok(x, wt) and native_ok(x, wt.contiguous())
    assert not native_ok(x, wt.contiguous(), torch.zeros(256, device=dev, dtype=torch.float16))
    y2 = linear_tall(x.detach(), wt)
    assert torch.allclose(y2, torch.nn.functional.linear(x.detach(), wt), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,cin,cout", [(43008, 256, 288), (5000, 256, 1024), (2048, 1024, 256)])
def test_linear_tall_native_forward_and_gradients(rows, cin, cout):
    """mp_former_amd.linear.linear_tall — the Linear of the per-layer encoder route and of the MSDeformAttn module
    (ops/modules/ms_deform_attn.py:95-124, msdeformattn.py:116-131) — runs the native fp32 GEMM forward, for the input
    gradient and (split over rows) for the weight / bias gradients: all four against fp64, no further from it than 2 x the
    library's fp32 result + 1e-6 relative."""
    from mp_former_amd import _lib
    from mp_former_amd.linear import linear_tall
    dev = torch.device("cuda:0")
    torch.manual_seed(rows + cout)
    x = torch.randn(2, rows // 2, cin, device=dev, requires_grad=True)
    w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).requires_grad_(True)
    b = torch.randn(cout, device=dev, requires_grad=True)
    g = torch.randn(2, rows // 2, cout, device=dev)
    y = linear_tall(x, w, b)
    assert "gemm3" in _lib.last_kernel(), _lib.last_kernel()
    y.backward(g)
    got = [y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64, b64)
    y64.backward(g.double())
    ref = [y64.detach(), x64.grad, w64.grad, b64.grad]
    x.grad = w.grad = b.grad = None
    yl = torch.nn.functional.linear(x, w, b)
    yl.backward(g)
    lib = [yl.detach(), x.grad, w.grad, b.grad]
    for name, a_, r_, l_ in zip(("y", "dx", "dw", "db"), got, ref, lib):
        den = float(r_.abs().max())
        e_got, e_lib = float((a_.double() - r_).abs().max()) / den, float((l_.double() - r_).abs().max()) / den
        assert e_got <= 2.0 * e_lib + 1e-6, (name, e_got, e_lib)
    # few rows / other dtypes keep F.linear
    small = linear_tall(torch.randn(4, 10, cin, device=dev), w.detach(), b.detach())
    assert small.shape == (4, 10, cout)
