"""Small-row bf16 Linear (csrc/small_gemm.hip) against torch: forward, dX, dW, db, ReLU gate, ragged sizes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, b, relu, gy):
    """bf16 operands, fp64 accumulation: what both the library and the kernel approximate."""
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if b is not None else None
    y = F.linear(xd, wd, bd)
    if relu:
        y = y.relu()
    y.backward(gy.double())
    return y.detach(), xd.grad, wd.grad, (bd.grad if b is not None else None)


@pytest.mark.parametrize("M,K,N,relu,bias", [
    (228, 256, 256, False, True), (228, 256, 2048, True, True), (228, 2048, 256, False, True),
    (202, 256, 256, True, False), (17, 64, 40, True, True), (300, 256, 768, False, True), (1, 32, 8, False, True),
])
def test_small_linear_matches_reference(M, K, N, relu, bias):
    from mp_former_amd.small_linear import small_linear
    torch.manual_seed(M + K + N)
    dev = torch.device("cuda:0")
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16() if bias else None
    gy = torch.randn(M, N, device=dev).bfloat16()
    y_ref, dx_ref, dw_ref, db_ref = _ref(x, w, b, relu, gy)

    xs, ws = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bs = b.clone().requires_grad_(True) if bias else None
    y = small_linear(xs, ws, bs, relu)
    assert y.dtype == torch.bfloat16 and y.shape == (M, N)
    y.backward(gy)

    xl, wl = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bl = b.clone().requires_grad_(True) if bias else None
    yl = F.linear(xl, wl, bl)
    if relu:
        yl = yl.relu()
    yl.backward(gy)

    def close(got, lib, ref, name):
        e_got = (got.double() - ref).abs().max().item()
        e_lib = (lib.double() - ref).abs().max().item()
        scale = ref.abs().max().item() + 1e-6
        # one bf16 rounding of the result (2^-8 relative to the largest entry), and no worse than 2x the library
        assert e_got <= max(2 * e_lib, 2 ** -8 * scale), f"{name}: err {e_got:.3e} lib {e_lib:.3e} scale {scale:.3e}"

    # the ReLU gate of the reference uses the fp64 pre-activation sign; entries that round to 0 in bf16
    # are gated off by both the kernel and the library path, so compare those against the library only
    close(y, yl, y_ref, "y")
    if relu:
        assert torch.equal(y > 0, yl > 0) or ((y > 0) != (yl > 0)).float().mean().item() < 1e-3
        dx_ref, dw_ref = xl.grad.double(), wl.grad.double()
        db_ref = bl.grad.double() if bias else None
        tol = lambda r: 2 ** -6 * (r.abs().max().item() + 1e-6)     # two bf16-rounded results compared
        assert (xs.grad.double() - dx_ref).abs().max().item() <= tol(dx_ref)
        assert (ws.grad.double() - dw_ref).abs().max().item() <= tol(dw_ref)
        if bias:
            assert (bs.grad.double() - db_ref).abs().max().item() <= tol(db_ref)
    else:
        close(xs.grad, xl.grad, dx_ref, "dx")
        close(ws.grad, wl.grad, dw_ref, "dw")
        if bias:
            close(bs.grad, bl.grad, db_ref, "db")


def test_small_linear_3d_input_and_strided_grad():
    from mp_former_amd.small_linear import small_linear
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    x = torch.randn(114, 2, 256, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(256, 256, device=dev) / 16).bfloat16().requires_grad_(True)
    b = torch.randn(256, device=dev).bfloat16().requires_grad_(True)
    y = small_linear(x, w, b)
    g = torch.randn(2, 114, 256, device=dev).bfloat16().transpose(0, 1)       # non-contiguous gradient
    y.backward(g)
    xl, wl, bl = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    F.linear(xl, wl, bl).backward(g)
    for got, lib in ((x.grad, xl.grad), (w.grad, wl.grad), (b.grad, bl.grad)):
        assert (got.float() - lib.float()).abs().max().item() <= 2 ** -6 * lib.float().abs().max().item()


def test_small_gemm_rejects_bad_strides():
    from mp_former_amd import _lib
    dev = torch.device("cuda:0")
    a = torch.zeros(16, 32, device=dev, dtype=torch.bfloat16)
    c = torch.zeros(16, 16, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    code = _lib.lib().mpf_small_gemm_bf16(a.data_ptr(), 32, 2, None, a.data_ptr(), 32, 1, None, None, 0, c.data_ptr(), 16, None, 16, 16, 32, 0, s)
    assert code == -2      # MPF_E_SHAPE
    code = _lib.lib().mpf_small_gemm_bf16(a.data_ptr(), 32, 1, None, a.data_ptr(), 32, 1, None, None, 0, c.data_ptr(), 16, None, 16, 14, 32, 0, s)
    assert code < 0


@pytest.mark.parametrize("R,M,N", [(2048, 256, 256), (8192, 256, 256), (32768, 256, 256), (5000, 256, 128), (3001, 100, 36)])
def test_gemm_nt_bf16_matches_reference(R, M, N):
    from mp_former_amd.small_linear import gemm_nt_bf16
    torch.manual_seed(R + M)
    dev = torch.device("cuda:0")
    a = torch.randn(R, M, device=dev).bfloat16()
    b = torch.randn(R, N, device=dev).bfloat16()
    c, cs = gemm_nt_bf16(a, b)
    ref = a.double().t() @ b.double()
    lib = (a.t() @ b)
    e_got = (c.double() - ref).abs().max().item()
    e_lib = (lib.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert e_got <= max(2 * e_lib, 2 ** -8 * scale), (e_got, e_lib, scale)
    sref = a.double().sum(0)
    assert (cs.double() - sref).abs().max().item() <= 2 ** -8 * sref.abs().max().item() + 1e-3


def test_tall_linear_grads():
    from mp_former_amd.small_linear import tall_linear
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    x = torch.randn(4096, 2, 256, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(256, 256, device=dev) / 16).bfloat16().requires_grad_(True)
    b = torch.randn(256, device=dev).bfloat16().requires_grad_(True)
    g = torch.randn(4096, 2, 256, device=dev).bfloat16()
    y = tall_linear(x, w, b)
    y.backward(g)
    xl, wl, bl = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    yl = F.linear(xl, wl, bl)
    yl.backward(g)
    assert torch.equal(y, yl)
    for got, lib in ((x.grad, xl.grad), (w.grad, wl.grad), (b.grad, bl.grad)):
        assert (got.float() - lib.float()).abs().max().item() <= 2 ** -6 * lib.float().abs().max().item()


def test_grouped_weight_gradients_equal_the_single_launches():
    """mpf_small_gemm_bf16_group: the dW problems of a decoder layer in one launch — bit-identical to one launch each."""
    from mp_former_amd.small_linear import small_gemm, weight_grads_grouped
    torch.manual_seed(11)
    dev = torch.device("cuda:0")
    R = 230
    shapes = [(256, 2048, False), (2048, 256, True), (256, 256, False), (768, 256, False), (256, 256, False), (40, 64, True)]
    probs = []
    for J, K, gated in shapes:
        dy = torch.randn(R, J, device=dev).bfloat16()
        x = torch.randn(R, K, device=dev).bfloat16()
        gate = torch.randn(R, J, device=dev).bfloat16() if gated else None
        probs.append((dy, x, gate))
    outs = weight_grads_grouped(probs)
    for (dy, x, gate), (dw, db) in zip(probs, outs):
        J, K = dy.shape[1], x.shape[1]
        dw1, db1 = small_gemm(dy, 1, dy.stride(0), x, 1, x.stride(0), J, K, R, gate=gate, rowsum=True)
        assert torch.equal(dw, dw1) and torch.equal(db, db1)
        g = dy.double() * (gate.double() > 0) if gate is not None else dy.double()
        ref = g.t() @ x.double()
        assert (dw.double() - ref).abs().max() <= 2e-2 * ref.abs().max()
    with pytest.raises(RuntimeError):
        weight_grads_grouped(probs + probs)          # more than 8 problems
