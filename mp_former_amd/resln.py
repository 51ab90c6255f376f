"""Post-norm residual block ``LayerNorm(x + t)`` of the decoder layers as one native pass
(csrc/elementwise.hip, mpf_res_ln256_*): the fp32 residual stream ``x`` plus the (bf16 under AMP)
branch output ``t``, normalised, written as the next fp32 residual and/or as the bf16 operand of the
next GEMMs — instead of add + LayerNorm + cast kernels forward and five kernels backward.
Reference: mask2former_transformer_decoder.py:42-52, :100-112, :165-169 (forward_post), :1861
(decoder_norm).  256 channels, CUDA; other cases use the torch ops."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib

_DT = {torch.float32: _lib.MPF_F32, torch.bfloat16: _lib.MPF_BF16}


def _stream(t):
    return _lib.stream_ptr(t.device)


class _ResLN(Function):
    @staticmethod
    def forward(ctx, x, t, gamma, beta, eps, want32, want16):
        rows = x.numel() // 256
        s = torch.empty_like(x) if t is not None else x
        y32 = torch.empty_like(x) if want32 else None
        y16 = torch.empty_like(x, dtype=torch.bfloat16) if want16 else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        with _lib.device_guard(x.device):
            code = _lib.lib().mpf_res_ln256_forward(
                x.data_ptr(), t.data_ptr() if t is not None else None, _DT[t.dtype] if t is not None else 0,
                gamma.data_ptr(), beta.data_ptr(), s.data_ptr() if t is not None else None,
                y32.data_ptr() if want32 else None, y16.data_ptr() if want16 else None, mean.data_ptr(), rstd.data_ptr(),
                rows, float(eps), None, 0, None, _stream(x))
        _lib.check(code, "mpf_res_ln256_forward")
        ctx.save_for_backward(s, mean, rstd, gamma)
        ctx.t_dtype = t.dtype if t is not None else None
        ctx.rows = rows
        return y32, y16

    @staticmethod
    def backward(ctx, g32, g16):
        s, mean, rstd, gamma = ctx.saved_tensors
        need_x, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.t_dtype is not None
        t16 = need_t and ctx.t_dtype == torch.bfloat16
        want32 = need_x or (need_t and not t16) or not t16
        ds32 = torch.empty_like(s) if want32 else None
        ds16 = torch.empty_like(s, dtype=torch.bfloat16) if t16 else None
        dgb = torch.empty((2, 256), dtype=torch.float32, device=s.device)
        if g32 is not None:
            g32 = g32.contiguous()
        if g16 is not None:
            g16 = g16.contiguous()
        lib = _lib.lib()
        args = (s.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                g32.data_ptr() if g32 is not None else None, g16.data_ptr() if g16 is not None else None, None,
                ds32.data_ptr() if ds32 is not None else None, ds16.data_ptr() if ds16 is not None else None)
        with _lib.device_guard(s.device):
            # parameter gradients without float atomics (bit-reproducible): per-workgroup partials summed in a fixed order —
            # by the workgroup that arrives last (one launch) for a few hundred rows, by a parallel second launch beyond
            if ctx.rows <= 1024:
                ws = _det_ws(s.device, lib.mpf_res_ln256_backward_det_workspace_bytes(ctx.rows))
                code = lib.mpf_res_ln256_backward_det(*args, dgb.data_ptr(), ctx.rows, ws.data_ptr(), ws.numel(), _stream(s))
            else:
                ws = _det_ws(s.device, lib.mpf_res_ln256_backward_workspace_bytes(ctx.rows) + 256)
                code = lib.mpf_res_ln256_backward_ws(*args, dgb[0].data_ptr(), dgb[1].data_ptr(), ctx.rows, ws.data_ptr() + 256,
                                                     ws.numel() - 256, _stream(s))
        _lib.check(code, "mpf_res_ln256_backward")
        dt = None
        if need_t:
            dt = ds16 if t16 else ds32
        return (ds32 if need_x else None), dt, dgb[0], dgb[1], None, None, None


def res_ln(norm, x, t=None, want32=True, want16=False):
    """(y32, y16) = LayerNorm(x + t) with the parameters of ``norm`` (an nn.LayerNorm).  x: fp32
    residual stream; t: branch output (fp32 / bf16) or None.  y32 (fp32) / y16 (bf16) are None unless
    requested."""
    C = x.shape[-1]
    ok = (x.is_cuda and C == 256 and x.dtype == torch.float32 and x.is_contiguous() and norm.elementwise_affine
          and norm.bias is not None and (t is None or (t.shape == x.shape and t.dtype in _DT and t.is_contiguous())))
    if ok:
        return _ResLN.apply(x, t, norm.weight, norm.bias, norm.eps, want32, want16)
    s = x if t is None else x + t
    y = F.layer_norm(s.float(), (C,), norm.weight, norm.bias, norm.eps)
    return (y if want32 else None), (y.to(torch.bfloat16) if want16 else None)


def ln256_forward(x, gamma, beta, eps, padd=None, y_bound=None, padd_amax=None, yplus_bound=None):
    """Raw forward for hand-scheduled callers (encoder_fused): x fp32 [rows, 256] -> (y, mean, rstd[, y + padd[row % len]]).
    y_bound / yplus_bound: zeroed amax slots that receive an upper bound of |y| / |y + padd| (padd_amax: the slot of padd)."""
    rows = x.shape[0]
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    yp = torch.empty_like(x) if padd is not None else None
    p_ = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with _lib.device_guard(x.device):
        if y_bound is not None:
            code = _lib.lib().mpf_res_ln256_forward_b(
                x.data_ptr(), None, 0, gamma.data_ptr(), beta.data_ptr(), None, y.data_ptr(), None, mean.data_ptr(), rstd.data_ptr(),
                rows, float(eps), p_(padd), padd.shape[0] if padd is not None else 0, p_(yp), y_bound.data_ptr(), p_(padd_amax),
                p_(yplus_bound) if yp is not None else None, _stream(x))
        else:
            code = _lib.lib().mpf_res_ln256_forward(
                x.data_ptr(), None, 0, gamma.data_ptr(), beta.data_ptr(), None, y.data_ptr(), None, mean.data_ptr(), rstd.data_ptr(),
                rows, float(eps), p_(padd), padd.shape[0] if padd is not None else 0, p_(yp), _stream(x))
    _lib.check(code, "mpf_res_ln256_forward")
    return y, mean, rstd, yp


def ln256_backward(s, mean, rstd, gamma, gy, gy_plus=None, ds_amax=None):
    """Raw backward: -> (ds fp32, dgamma, dbeta) with g = gy (+ gy_plus); ds_amax: a zeroed amax slot that receives max |ds|."""
    ds = torch.empty_like(s)
    dgb = torch.empty((2, 256), dtype=torch.float32, device=s.device)
    lib = _lib.lib()
    nbytes = lib.mpf_res_ln256_backward_workspace_bytes(s.shape[0])
    wkey = (s.device, _lib.ws_scope())          # (a graph capture has its own buffers: _lib.workspace_scope)
    ws = _ln_ws.get(wkey)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes) + 1024, dtype=torch.uint8, device=s.device)
        _ln_ws[wkey] = ws
    with _lib.device_guard(s.device):
        # parameter gradients through per-workgroup partials, fixed order (no atomics, no zero-fill)
        args = (s.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), gy.data_ptr(), None,
                gy_plus.data_ptr() if gy_plus is not None else None, ds.data_ptr(), None, dgb[0].data_ptr(), dgb[1].data_ptr(),
                s.shape[0], ws.data_ptr(), ws.numel())
        if ds_amax is not None:
            code = lib.mpf_res_ln256_backward_ws_amax(*args, ds_amax.data_ptr(), _stream(s))
        else:
            code = lib.mpf_res_ln256_backward_ws(*args, _stream(s))
    _lib.check(code, "mpf_res_ln256_backward_ws")
    return ds, dgb[0], dgb[1]


class LnGradGroup:
    """Parameter gradients of SEVERAL 256-channel LayerNorm backwards over the same number of rows with ONE reduce launch: every
    backward leaves its per-workgroup partial sums in its own slot (`backward`), `finish()` sums all slots in a fixed order ->
    [n, 2, 256] (dgamma, dbeta) per LayerNorm, in the order of the `backward` calls."""

    def __init__(self, n, rows, device):
        lib = _lib.lib()
        self.n, self.rows, self.used = n, rows, 0
        self.stride = (int(lib.mpf_res_ln256_backward_workspace_bytes(rows)) + 255) & ~255
        self.parts = torch.empty(n * self.stride, dtype=torch.uint8, device=device)

    def backward(self, s, mean, rstd, gamma, gy, gy_plus=None, ds_amax=None):
        """-> ds fp32 (the parameter gradients come from finish())"""
        assert self.used < self.n and s.shape[0] == self.rows
        ds = torch.empty_like(s)
        with _lib.device_guard(s.device):
            code = _lib.lib().mpf_res_ln256_backward_partial_amax(
                s.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), gy.data_ptr(), None,
                gy_plus.data_ptr() if gy_plus is not None else None, ds.data_ptr(), None, self.rows,
                self.parts.data_ptr() + self.used * self.stride, self.stride, ds_amax.data_ptr() if ds_amax is not None else None, _stream(s))
        _lib.check(code, "mpf_res_ln256_backward_partial_amax")
        self.used += 1
        return ds

    def finish(self):
        out = torch.empty((self.used, 2, 256), dtype=torch.float32, device=self.parts.device)
        with _lib.device_guard(out.device):
            code = _lib.lib().mpf_ln_partial_reduce(self.parts.data_ptr(), self.stride, self.rows, self.used, out.data_ptr(), _stream(out))
        _lib.check(code, "mpf_ln_partial_reduce")
        return out


_ln_ws = {}
_det_ws_cache = {}


def _det_ws(dev, nbytes):
    """workspace of mpf_res_ln256_backward_det per (device, stream): its ticket word is zero between calls (zero-initialised
    here, reset by every launch), so it is shared by all LayerNorms that run on that stream"""
    key = (dev, _lib.stream_ptr(dev), _lib.ws_scope())
    ws = _det_ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(int(nbytes) + 4096, dtype=torch.uint8, device=dev)
        _det_ws_cache[key] = ws
    return ws
