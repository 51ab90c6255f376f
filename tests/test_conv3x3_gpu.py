"""3x3 convolution on the split-bf16 GEMM (mpf_gemm3_conv3x3) against F.conv2d in fp64: forward, input gradient,
weight / bias gradients; odd sizes, batch > 1 (taps must not leak across image or row boundaries)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,C,Co,H,W,bias", [(2, 64, 64, 9, 7, True), (1, 32, 96, 5, 12, False), (3, 256, 256, 16, 16, True),
                                               (2, 256, 256, 37, 21, False), (2, 128, 128, 10, 24, True), (2, 256, 256, 64, 64, True)])
def test_conv3x3_matches_conv2d(N, C, Co, H, W, bias):
    from mp_former_amd import _lib
    from mp_former_amd.conv3x3 import conv3x3, supported
    dev = torch.device("cuda:0")
    torch.manual_seed(N + C + H)
    x = torch.randn(N, H, W, C, device=dev).permute(0, 3, 1, 2).requires_grad_(True)
    w = (torch.randn(Co, C, 3, 3, device=dev) / (3 * C ** 0.5)).requires_grad_(True)
    b = torch.randn(Co, device=dev).requires_grad_(True) if bias else None
    assert supported(x, w)
    y = conv3x3(x, w, b)
    # channel counts that are multiples of 256 take the fp16 x 2 form (two-pass tiles both ways), the others the bf16 x 3 one
    assert _lib.last_kernel() == ("gemm3_conv_kernel<h2>" if C % 256 == 0 and Co % 256 == 0 else "gemm3_conv_kernel")
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, 1, 1)
    lib = F.conv2d(x.detach(), w.detach(), b.detach() if bias else None, 1, 1)
    e3, el = (y.double() - yr).abs(), (lib.double() - yr).abs()
    assert float(e3.max()) <= 4e-6 * (float(yr.abs().max()) + 1.0)
    assert float(e3.mean()) <= 4.0 * float(el.mean()) + 1e-8        # same order as the library fp32 convolution (K = 9 Cin terms)
    g = torch.randn_like(y)
    y.backward(g)
    yr.backward(g.double())
    assert float((x.grad.double() - xr.grad).abs().max()) <= 4e-6 * (float(xr.grad.abs().max()) + 1.0)
    torch.testing.assert_close(w.grad.double(), wr.grad, rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()))
    if bias:
        torch.testing.assert_close(b.grad.double(), br.grad, rtol=1e-4, atol=1e-4 * float(br.grad.abs().max()))


@pytest.mark.parametrize("N,C,Co,H,W,bias", [(2, 256, 256, 24, 20, True), (1, 512, 256, 9, 7, False), (2, 64, 96, 5, 5, True)])
def test_conv1x1_matches_conv2d(N, C, Co, H, W, bias):
    from mp_former_amd import _lib
    from mp_former_amd.conv3x3 import conv1x1, supported_1x1
    dev = torch.device("cuda:0")
    torch.manual_seed(C + H)
    x = torch.randn(N, H, W, C, device=dev).permute(0, 3, 1, 2).requires_grad_(True)
    w = (torch.randn(Co, C, 1, 1, device=dev) / C ** 0.5).requires_grad_(True)
    b = torch.randn(Co, device=dev).requires_grad_(True) if bias else None
    assert supported_1x1(x, w)
    y = conv1x1(x, w, b)
    assert "gemm3_tn" in _lib.last_kernel()
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br)
    assert float((y.double() - yr).abs().max()) <= 4e-6 * (float(yr.abs().max()) + 1.0)
    g = torch.randn(N, Co, H, W, device=dev)          # NCHW-contiguous gradient: the backward relayouts it
    y.backward(g)
    yr.backward(g.double())
    assert float((x.grad.double() - xr.grad).abs().max()) <= 4e-6 * (float(xr.grad.abs().max()) + 1.0)
    torch.testing.assert_close(w.grad.double(), wr.grad, rtol=1e-4, atol=1e-5 * float(wr.grad.abs().max()) + 1e-6)
    if bias:
        torch.testing.assert_close(b.grad.double(), br.grad, rtol=1e-4, atol=1e-5 * float(br.grad.abs().max()) + 1e-6)


@pytest.mark.parametrize("N,C,Co,H,W,mode", [(2, 256, 256, 24, 20, "bf16_in"), (1, 512, 256, 9, 8, "bf16_in"), (2, 256, 256, 16, 12, "bf16_out"),
                                              (2, 96, 64, 6, 6, "bf16_in")])
def test_conv1x1_bf16_input_or_output(N, C, Co, H, W, mode):
    """bf16 feature map in (no cast pass, bf16 input gradient out of the GEMM epilogue) / bf16 result out (its gradient taken
    in bf16): the values of the fp32 convolution on the same numbers, rounded once."""
    from mp_former_amd.conv3x3 import conv1x1, supported_1x1
    dev = torch.device("cuda:0")
    torch.manual_seed(C + W)
    xdt = torch.bfloat16 if mode == "bf16_in" else torch.float32
    odt = torch.bfloat16 if mode == "bf16_out" else torch.float32
    x = torch.randn(N, H, W, C, device=dev).to(xdt).permute(0, 3, 1, 2).requires_grad_(True)
    w = (torch.randn(Co, C, 1, 1, device=dev) / C ** 0.5).requires_grad_(True)
    b = torch.randn(Co, device=dev).requires_grad_(True)
    assert supported_1x1(x, w)
    y = conv1x1(x, w, b, odt)
    assert y.dtype == odt
    xr, wr, br = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    tol = 2.0 ** -8 if odt == torch.bfloat16 else 4e-6
    assert float((y.double() - yr).abs().max()) <= tol * (float(yr.abs().max()) + 1.0)
    g = torch.randn(N, Co, H, W, device=dev).to(odt)
    y.backward(g)
    yr.backward(g.double())
    assert x.grad.dtype == xdt
    tolx = 2.0 ** -8 if xdt == torch.bfloat16 else 4e-6
    assert float((x.grad.double() - xr.grad).abs().max()) <= tolx * (float(xr.grad.abs().max()) + 1.0)
    torch.testing.assert_close(w.grad.double(), wr.grad, rtol=1e-4, atol=1e-5 * float(wr.grad.abs().max()) + 1e-6)
    torch.testing.assert_close(b.grad.double(), br.grad, rtol=1e-4, atol=1e-5 * float(br.grad.abs().max()) + 1e-6)
