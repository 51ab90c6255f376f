"""One masked-attention decoder layer (cross-attention, self-attention, FFN, post-norm residuals) as ONE
autograd node whose forward / backward are single native calls (csrc/decoder_layer.hip;
include/mpformer_hip.h MpfDecoderLayer).  Reference: mask2former_transformer_decoder.py:1784-1800.

Same kernels and rounding points as the op-by-op path of transformer_decoder.py (small_linear, attention
core, res_ln) — what changes is who issues them: ~20 / ~45 launches per layer from C++ instead of one
Python autograd node per launch, which left the GPU idle for most of the decoder.  bf16 autocast, 256
channels, 8 heads, GPU only; the caller projects the cross-attention keys / values (library GEMMs).
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib
from .attention import _workspace as _attn_workspace

_vp, _u64, _i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32

PARAM_NAMES = ("ca_wq", "ca_bq", "ca_wo", "ca_bo", "ca_gamma", "ca_beta",
               "sa_wq", "sa_bq", "sa_wk", "sa_bk", "sa_wv", "sa_bv", "sa_wo", "sa_bo", "sa_gamma", "sa_beta",
               "ff_w1", "ff_b1", "ff_w2", "ff_b2", "ff_gamma", "ff_beta")
_LN = ("ca_gamma", "ca_beta", "sa_gamma", "sa_beta", "ff_gamma", "ff_beta")
_SAVED = ("q_c", "kT_c", "o_c", "lse_c", "s1", "mean1", "rstd1", "xb1", "q_s", "k_s", "v_s", "kT_s", "o_s", "lse_s",
          "s2", "mean2", "rstd2", "xb2", "h", "s3", "mean3", "rstd3")


class MpfDecoderLayer(ctypes.Structure):
    _fields_ = ([(n, _vp) for n in PARAM_NAMES] + [(n, _vp) for n in ("x0", "xb0", "k_c", "v_c", "mask_c", "mask_s")]
                + [(n, _vp) for n in _SAVED] + [("x3", _vp), ("xb3", _vp), ("scratch", _vp), ("attn_ws", _vp),
                                                ("scratch_bytes", _u64), ("attn_ws_bytes", _u64)]
                + [(n, _i32) for n in ("Qt", "N", "H", "S", "ffn_dim")] + [("eps", ctypes.c_float)]
                + [("kv_row_stride", ctypes.c_int64), ("kv_img_stride", ctypes.c_int64)])


_WGRADS = tuple("d_" + n for n in PARAM_NAMES if n not in _LN)


class MpfDecoderLayerGrad(ctypes.Structure):
    _fields_ = ([(n, _vp) for n in ("g_x3", "g_xb3", "d_x0", "d_xb0", "d_k_c", "d_v_c")] + [(n, _vp) for n in _WGRADS]
                + [("d_ln", _vp), ("dkv_row_stride", ctypes.c_int64), ("dkv_img_stride", ctypes.c_int64), ("g_x3_plus", _vp)])


_checked = False
_scratch = {}
_PARAM_DTYPES = tuple(torch.float32 if n in _LN else torch.bfloat16 for n in PARAM_NAMES)
_layouts = {}


def _arena_layout(R, E, N, S, Qt, H, F_):
    """byte offsets of the saved tensors (_SAVED order) inside a layer's arena, and its size"""
    key = (R, E, N, S, Qt, H, F_)
    hit = _layouts.get(key)
    if hit is None:
        sizes = {"q_c": R * E * 2, "kT_c": N * E * S * 2, "o_c": R * E * 2, "lse_c": N * H * Qt * 4,
                 "s1": R * E * 4, "mean1": R * 4, "rstd1": R * 4, "xb1": R * E * 2,
                 "q_s": R * E * 2, "k_s": R * E * 2, "v_s": R * E * 2, "kT_s": N * E * Qt * 2, "o_s": R * E * 2,
                 "lse_s": N * H * Qt * 4, "s2": R * E * 4, "mean2": R * 4, "rstd2": R * 4, "xb2": R * E * 2,
                 "h": R * F_ * 2, "s3": R * E * 4, "mean3": R * 4, "rstd3": R * 4}
        offs, tot = [], 0
        for n in _SAVED:
            offs.append(tot)
            tot += _al(sizes[n])
        hit = _layouts[key] = (tuple(offs), tot)
    return hit


_scratch_sizes = {}


def _scratch_bytes(lib, Qt, N, H, S, F_):
    """(layer scratch, attention workspace) bytes of a shape.  The attention workspace depends on run-time options (the key
    split: attn_nw / attn_kpw), so the cache is keyed on the option generation `_lib.set_option` advances."""
    key = (Qt, N, H, S, F_, _lib.option_generation())
    hit = _scratch_sizes.get(key)
    if hit is None:
        if len(_scratch_sizes) > 256:
            _scratch_sizes.clear()
        hit = _scratch_sizes[key] = (max(lib.mpf_decoder_layer_scratch_bytes(Qt, N, H, S, F_, 0),
                                         lib.mpf_decoder_layer_scratch_bytes(Qt, N, H, S, F_, 1)),
                                     max(lib.mpf_attn_workspace_bytes(Qt, S, N, H), lib.mpf_attn_workspace_bytes(Qt, Qt, N, H)))
    return hit


def _lib_checked():
    global _checked
    lib = _lib.lib()
    if not _checked:
        if (lib.mpf_decoder_layer_struct_bytes(0) != ctypes.sizeof(MpfDecoderLayer)
                or lib.mpf_decoder_layer_struct_bytes(1) != ctypes.sizeof(MpfDecoderLayerGrad)):
            raise RuntimeError("MpfDecoderLayer layout differs between decoder_layer.py and libmpformer_hip.so")
        _checked = True
    return lib


def _scratch_buf(device, nbytes):
    key = (device, _lib.ws_scope())
    w = _scratch.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _scratch[key] = w
    return w


def _al(n):
    return (n + 255) & ~255


class ColumnPack:
    """The gradient of a [S, N, n * E] matrix whose n column blocks went to n different consumers (the key / value
    projections of the decoder layers that share a feature level): ONE buffer, allocated when the first consumer's
    backward asks for its block; each consumer writes its block in place, so no zero-padded slice gradients are summed."""

    def __init__(self, shape, dtype, device, block):
        self.shape, self.dtype, self.device, self.block = tuple(shape), dtype, device, block
        self.buf = None

    def block_view(self, j):
        if self.buf is None:
            self.buf = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        return self.buf[..., j * self.block:(j + 1) * self.block]


class _SplitColsFn(Function):
    @staticmethod
    def forward(ctx, y, pack, n):
        ctx.pack, ctx.n = pack, n
        E = pack.block
        return tuple(y[..., j * E:(j + 1) * E] for j in range(n))

    @staticmethod
    def backward(ctx, *gs):
        pack = ctx.pack
        for j, g in enumerate(gs):
            want = pack.block_view(j)
            if g is None:
                want.zero_()
            elif not (g.data_ptr() == want.data_ptr() and g.stride() == want.stride() and g.dtype == want.dtype):
                want.copy_(g)                      # a consumer that did not write in place
        buf, pack.buf = pack.buf, None
        return buf, None, None


def split_cols(y, n):
    """y [S, N, n * E] -> [(view_j [S, N, E], (pack, j))]: column-block views for n consumers plus the handle through which
    a consumer's backward (DecoderLayerFn) writes its block of the gradient in place."""
    pack = ColumnPack(y.shape, y.dtype, y.device, y.shape[-1] // n)
    if not y.requires_grad:
        E = pack.block
        return [(y[..., j * E:(j + 1) * E], None) for j in range(n)]
    views = _SplitColsFn.apply(y, pack, n)
    return [(v, (pack, j)) for j, v in enumerate(views)]


_IS_LN = tuple(n in _LN for n in PARAM_NAMES)
_WNAMES = tuple(n for n in PARAM_NAMES if n not in _LN)
_wgrad_layouts = {}


def _wgrad_layout(wshapes):
    """element offsets of the weight / bias gradients inside a layer's gradient arena: the q / k / v weight gradients side by side,
    then their bias gradients (the native layer then produces them with one GEMM), the rest in parameter order"""
    hit = _wgrad_layouts.get(wshapes)
    if hit is None:
        order = sorted(range(len(_WNAMES)), key=lambda i: ({"sa_wq": 0, "sa_wk": 1, "sa_wv": 2, "sa_bq": 3, "sa_bk": 4, "sa_bv": 5}
                                                            .get(_WNAMES[i], 6 + i)))
        offs, tot = [0] * len(_WNAMES), 0
        for i in order:
            offs[i] = tot
            tot += (wshapes[i].numel() + 127) & ~127
        hit = _wgrad_layouts[wshapes] = (tuple(offs), tot)
    return hit


def _contig_strides(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= d
    return tuple(reversed(st))


class DecoderLayerFn(Function):
    @staticmethod
    def forward(ctx, x0, xb0, k_c, v_c, mask_c, mask_s, nheads, eps, kv_pack, *params):
        lib = _lib_checked()
        Qt, N, E = x0.shape
        S = k_c.shape[0]
        F_ = params[16].shape[0]
        R = Qt * N
        dev = x0.device
        assert E == 256 and nheads * 32 == E and len(params) == len(PARAM_NAMES)
        assert x0.dtype == torch.float32 and x0.is_contiguous() and xb0.dtype == torch.bfloat16 and xb0.is_contiguous()
        # k_c / v_c: dense [S, N, E] or column blocks of a wider projection (same strides for both, rows 16-byte aligned)
        assert k_c.dtype == v_c.dtype == torch.bfloat16 and k_c.shape == v_c.shape == (S, N, E) and k_c.stride() == v_c.stride()
        assert k_c.stride(2) == 1 and k_c.stride(0) % 8 == 0 and k_c.stride(1) % 8 == 0
        assert k_c.data_ptr() % 16 == 0 and v_c.data_ptr() % 16 == 0
        assert mask_c.dtype == torch.bool and mask_c.is_contiguous() and mask_c.shape == (N, Qt, S)
        if mask_s is not None:
            assert mask_s.dtype == torch.bool and mask_s.shape == (Qt, Qt)
            mask_s = mask_s.contiguous()
        # (this function runs 9 times per step on the launch thread: one pass over the parameters, shape-derived tables cached)
        ptrs = []
        fixed = None
        for i, p in enumerate(params):
            if p.dtype is not _PARAM_DTYPES[i]:
                raise AssertionError(PARAM_NAMES[i])
            if not p.is_contiguous():
                if fixed is None:
                    fixed = list(params)
                fixed[i] = p = p.contiguous()
            ptrs.append(p.data_ptr())
        if fixed is not None:
            params = tuple(fixed)
        # one arena for everything the backward reads (offsets per shape cached)
        offs, total = _arena_layout(R, E, N, S, Qt, nheads, F_)
        arena = torch.empty(total, dtype=torch.uint8, device=dev)
        x3 = torch.empty_like(x0)
        xb3 = torch.empty_like(xb0)
        L = MpfDecoderLayer()
        base = arena.data_ptr()
        for n, o in zip(_SAVED, offs):
            setattr(L, n, base + o)
        for n, v in zip(PARAM_NAMES, ptrs):
            setattr(L, n, v)
        L.x0, L.xb0, L.k_c, L.v_c = x0.data_ptr(), xb0.data_ptr(), k_c.data_ptr(), v_c.data_ptr()
        L.mask_c = mask_c.data_ptr()
        L.mask_s = mask_s.data_ptr() if mask_s is not None else None
        L.x3, L.xb3 = x3.data_ptr(), xb3.data_ptr()
        L.Qt, L.N, L.H, L.S, L.ffn_dim, L.eps = Qt, N, nheads, S, F_, float(eps)
        if not k_c.is_contiguous():
            L.kv_row_stride, L.kv_img_stride = k_c.stride(0), k_c.stride(1)
        sc_bytes, ws_bytes = _scratch_bytes(lib, Qt, N, nheads, S, F_)
        sc = _scratch_buf(dev, sc_bytes)
        ws = _attn_workspace(dev, ws_bytes)
        L.scratch, L.scratch_bytes, L.attn_ws, L.attn_ws_bytes = sc.data_ptr(), sc.numel(), ws.data_ptr(), ws.numel()
        stream = _lib.stream_ptr(dev)
        with _lib.device_guard(dev):
            code = lib.mpf_decoder_layer_forward(ctypes.byref(L), stream)
        _lib.check(code, "mpf_decoder_layer_forward")
        ctx.save_for_backward(xb0, k_c, v_c, mask_c, mask_s, arena, *params)
        ctx.layer = L
        ctx.kv_pack = kv_pack
        ctx.dims = (Qt, N, E, S, F_, nheads)
        # x3 leaves twice (two aliases of one tensor): the next layer and the prediction heads both consume it, and their
        # gradients then arrive as two arguments — summed by the first LayerNorm backward pass of the native call (its third
        # gradient operand) instead of by an add kernel and an accumulation step of the autograd engine per layer
        ctx.set_materialize_grads(False)
        return x3, xb3, x3.view(x3.shape)

    @staticmethod
    def backward(ctx, g_x3, g_xb3, g_x3h=None):
        if g_x3 is None:
            g_x3, g_x3h = g_x3h, None
        if g_x3 is None and g_xb3 is None:
            return (None,) * (9 + len(PARAM_NAMES))
        lib = _lib_checked()
        saved = ctx.saved_tensors
        xb0, k_c = saved[0], saved[1]
        params = saved[6:]
        Qt, N, E, S, F_, H = ctx.dims
        dev = xb0.device
        L = ctx.layer
        if g_x3 is not None:
            g_x3 = g_x3.to(torch.float32).contiguous()
        if g_xb3 is not None:
            g_xb3 = g_xb3.to(torch.bfloat16).contiguous()
        if g_x3h is not None:
            g_x3h = g_x3h.to(torch.float32).contiguous()
        d_x0 = torch.empty((Qt, N, E), dtype=torch.float32, device=dev)
        d_xb0 = torch.empty((Qt, N, E), dtype=torch.bfloat16, device=dev)
        G = MpfDecoderLayerGrad()
        if ctx.kv_pack is not None:       # our blocks of the packed dK / dV of the level, written in place
            (pk, jk), (pv, jv) = ctx.kv_pack
            d_k, d_v = pk.block_view(jk), pv.block_view(jv)
            assert d_k.stride() == d_v.stride()
            G.dkv_row_stride, G.dkv_img_stride = d_k.stride(0), d_k.stride(1)
        else:
            d_kv = torch.empty((2,) + tuple(k_c.shape), dtype=torch.bfloat16, device=dev)
            d_k, d_v = d_kv[0], d_kv[1]
        wshapes = [p.shape for p, ln in zip(params, _IS_LN) if not ln]
        offs, tot = _wgrad_layout(tuple(wshapes))
        wg = torch.empty(tot, dtype=torch.bfloat16, device=dev)
        d_ln = torch.empty((6, 256), dtype=torch.float32, device=dev)
        G.g_x3 = g_x3.data_ptr() if g_x3 is not None else None
        G.g_xb3 = g_xb3.data_ptr() if g_xb3 is not None else None
        G.g_x3_plus = g_x3h.data_ptr() if g_x3h is not None else None
        G.d_x0, G.d_xb0, G.d_k_c, G.d_v_c = d_x0.data_ptr(), d_xb0.data_ptr(), d_k.data_ptr(), d_v.data_ptr()
        wbase = wg.data_ptr()
        for n, o in zip(_WGRADS, offs):
            setattr(G, n, wbase + 2 * o)
        G.d_ln = d_ln.data_ptr()
        sc_bytes, ws_bytes = _scratch_bytes(lib, Qt, N, H, S, F_)
        sc = _scratch_buf(dev, sc_bytes)
        ws = _attn_workspace(dev, ws_bytes)
        L.scratch, L.scratch_bytes, L.attn_ws, L.attn_ws_bytes = sc.data_ptr(), sc.numel(), ws.data_ptr(), ws.numel()
        stream = _lib.stream_ptr(dev)
        with _lib.device_guard(dev):
            code = lib.mpf_decoder_layer_backward(ctypes.byref(L), ctypes.byref(G), stream)
        _lib.check(code, "mpf_decoder_layer_backward")
        grads, wi, li = [], 0, 0
        lns = d_ln.unbind(0)
        for ln in _IS_LN:
            if ln:
                grads.append(lns[li])
                li += 1
            else:
                s_ = wshapes[wi]
                grads.append(wg.as_strided(s_, _contig_strides(s_), offs[wi]))      # (one view op instead of slice + view)
                wi += 1
        return (d_x0, d_xb0, d_k, d_v, None, None, None, None, None, *grads)


def decoder_layer(x0, xb0, k_c, v_c, mask_c, mask_s, nheads, eps, params, kv_pack=None):
    """(x3 fp32, xb3 bf16, x3 again) = one decoder layer; ``params``: the 22 tensors of PARAM_NAMES.  The third result is an
    alias of the first for a second consumer (see DecoderLayerFn.forward).  ``kv_pack`` = the
    ``split_cols`` handles of k_c and v_c when they are column blocks of a level's packed projection."""
    if not x0.is_cuda:
        raise RuntimeError("mp_former_amd decoder layer runs on the GPU only (no CPU fallback)")
    return DecoderLayerFn.apply(x0, xb0, k_c, v_c, mask_c, mask_s, nheads, eps, kv_pack, *params)
