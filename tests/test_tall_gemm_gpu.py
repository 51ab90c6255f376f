"""mpf_tall_gemm_bf16 (csrc/small_gemm.hip): the forward / input-gradient GEMM of the many-row bf16 Linear layers (key / value
in-projections, batched prediction heads) against an fp32 reference of the same op, and tall_linear / the padded class_embed
against F.linear with autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ws", [1, 0])
@pytest.mark.parametrize("M,N,K,bias", [(2048, 768, 256, True), (32768, 768, 256, True), (8192, 256, 768, False), (2400, 256, 256, True),
                                        (2280, 96, 256, True), (1090, 260, 768, True), (130, 100, 64, False), (1, 4, 32, True),
                                        (257, 260, 96, False)])
def test_tall_gemm_matches_fp32_reference(M, N, K, bias, ws):
    """both kernels behind mpf_tall_gemm_bf16: the weight-stationary one (K = 256 / 768, >= 1024 rows: W fragments in registers, X
    row tiles through LDS) and the fragment-per-wave one (every other shape; all shapes with mpf_set_option("tall_ws", 0))"""
    from mp_former_amd import _lib
    from mp_former_amd.small_linear import tall_gemm
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    bz = (torch.randn(N, generator=g) * 0.2).to(torch.bfloat16).to(dev) if bias else None
    _lib.set_option("tall_ws", ws)
    try:
        c = tall_gemm(a, b, bz)
        want = "tall_ws_bf16_kernel" if (ws and M >= 1024 and K in (256, 768)) else "tall_gemm_bf16_kernel"
        assert _lib.last_kernel() == want
    finally:
        _lib.set_option("tall_ws", 1)
    ref = a.float() @ b.float().t() + (bz.float() if bias else 0.0)
    # one bf16 rounding of an fp32-accumulated result: half an ulp = 2^-9 relative
    torch.testing.assert_close(c.float(), ref, rtol=4e-3, atol=4e-3 * float(ref.abs().max()) / 64)
    # bit-equal to the rounded fp32 reference almost everywhere (accumulation order differs in the last fp32 bits only)
    same = (c == ref.to(torch.bfloat16)).float().mean()
    assert float(same) > 0.98, float(same)


def test_tall_gemm_strided_operands_and_errors():
    from mp_former_amd import _lib
    from mp_former_amd.small_linear import tall_gemm
    dev = torch.device("cuda:0")
    big = torch.randn(4096, 768, device=dev).to(torch.bfloat16)
    a = big[:, 256:512]                                   # a column block of a wider matrix: row stride 768
    b = torch.randn(256, 256, device=dev).to(torch.bfloat16)
    torch.testing.assert_close(tall_gemm(a, b).float(), a.float() @ b.float().t(), rtol=4e-3, atol=2e-2)
    with pytest.raises(Exception):                       # K not a multiple of 32
        tall_gemm(torch.zeros(8, 40, device=dev, dtype=torch.bfloat16), torch.zeros(8, 40, device=dev, dtype=torch.bfloat16))


@pytest.mark.parametrize("rows,N,K", [(4096, 768, 256), (2280, 256, 256)])
def test_tall_linear_forward_backward_vs_f_linear(rows, N, K):
    from mp_former_amd.small_linear import tall_linear
    dev = torch.device("cuda:0")
    torch.manual_seed(rows)
    x = torch.randn(rows // 2, 2, K, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16).requires_grad_(True)
    b = (torch.randn(N, device=dev) * 0.1).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(rows // 2, 2, N, device=dev).to(torch.bfloat16)
    y = tall_linear(x, w, b)
    y.backward(go)
    x32, w32, b32 = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    y32 = F.linear(x32, w32, b32)
    y32.backward(go.float())
    torch.testing.assert_close(y.float(), y32, rtol=8e-3, atol=2e-2)
    for got, want, name in ((x.grad, x32.grad, "dx"), (w.grad, w32.grad, "dw"), (b.grad, b32.grad, "db")):
        err = (got.float() - want).norm() / want.norm()
        assert float(err) < 6e-3, (name, float(err))


def test_padded_class_embed_equals_f_linear():
    """81 classes: weight / bias padded to 96 rows for the native kernels, pad sliced off; values and all three gradients"""
    from mp_former_amd import _lib
    from mp_former_amd.transformer_decoder import _class_linear
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(1200, 2, 256, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(81, 256, device=dev) * 0.05).to(torch.bfloat16).requires_grad_(True)
    b = (torch.randn(81, device=dev) * 0.1).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(1200, 2, 81, device=dev).to(torch.bfloat16)
    _lib.profile_enable(True)
    y = _class_linear(x, w, b)
    y.backward(go)
    torch.cuda.synchronize()
    assert _lib.profile_get("tall_gemm_bf16_kernel")[0] == 2, "forward and input gradient did not take the native kernel"
    _lib.profile_enable(False)
    assert y.shape == (1200, 2, 81)
    x32, w32, b32 = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    y32 = F.linear(x32, w32, b32)
    y32.backward(go.float())
    torch.testing.assert_close(y.float(), y32, rtol=8e-3, atol=2e-2)
    for got, want, name in ((x.grad, x32.grad, "dx"), (w.grad, w32.grad, "dw"), (b.grad, b32.grad, "db")):
        assert got.shape == want.shape
        err = (got.float() - want).norm() / want.norm()
        assert float(err) < 6e-3, (name, float(err))
