"""ResNet-50 feature extractor used ONLY to complete the training step in bench.py (res2..res5 at
strides 4/8/16/32, channels 256/512/1024/2048; detectron2 `build_resnet_backbone` layout with
STRIDE_IN_1X1 False and FrozenBN, configs/coco/instance-segmentation/Base-COCO-InstanceSegmentation.yaml:2-15).
Stock PyTorch-ROCm ops (MIOpen convolutions) — the backbone is outside the native hot path
('backbone stays stock PyTorch-ROCm')."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._h2d import upload


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm with fixed statistics and affine parameters (detectron2 FrozenBatchNorm2d)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def scale_bias(self):
        """(scale, shift) of the fixed affine; cached — the statistics are buffers that only change through load_state_dict
        (in place: the version counters move) or .to() (the tensor objects are replaced).  The check reads the buffer dict
        directly (identity + version of the four tensors): ``self.weight`` on a Module goes through ``__getattr__``, and this
        runs once per convolution and step."""
        b = self._buffers
        w, bb, rm, rv = b["weight"], b["bias"], b["running_mean"], b["running_var"]
        c = self.__dict__.get("_sb_cache")
        if (c is None or c[0] is not w or c[1] is not bb or c[2] is not rm or c[3] is not rv
                or c[4] != (w._version, bb._version, rm._version, rv._version)):
            with torch.no_grad():
                scale = w * (rv + self.eps).rsqrt()
                c = (w, bb, rm, rv, (w._version, bb._version, rm._version, rv._version), scale,
                     (bb - rm * scale).float().contiguous())
            self.__dict__["_sb_cache"] = c
        return c[5], c[6]

    def forward(self, x):
        scale, bias = self.scale_bias()
        return x * scale.to(x.dtype).view(1, -1, 1, 1) + bias.to(x.dtype).view(1, -1, 1, 1)


class _BiasAct(torch.autograd.Function):
    """relu?(x + shift[c] + res) over a channel-last activation in one native pass (csrc/elementwise.hip);
    the shift comes from frozen buffers (no gradient)."""

    @staticmethod
    def forward(ctx, x, shift, res, relu):
        y = torch.empty_like(x)
        C = x.shape[1]
        code = _lib.lib().mpf_bias_act(x.data_ptr(), shift.data_ptr(), res.data_ptr() if res is not None else None,
                                       y.data_ptr(), x.numel(), C, _lib.MPF_BF16 if x.dtype == torch.bfloat16 else _lib.MPF_F32,
                                       1 if relu else 0, _lib.stream_ptr(x.device))
        _lib.check(code, "mpf_bias_act")
        ctx.relu, ctx.has_res = relu, res is not None
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        if ctx.relu:
            (y,) = ctx.saved_tensors
            g = torch.ops.aten.threshold_backward(g, y, 0)
        return g, None, (g if ctx.has_res else None), None


def _relu_bwd_add(ga, gb, y):
    """y > 0 ? ga + gb : 0 in one native pass (csrc/elementwise.hip: aten's add + threshold_backward, same rounding points);
    gb may be None.  Falls back to the two aten ops for layouts / dtypes the kernel does not take."""
    same = (ga.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and ga.shape == y.shape and ga.stride() == y.stride() and _dense(y)
            and ga.numel() % 8 == 0 and (gb is None or (gb.dtype == ga.dtype and gb.shape == ga.shape and gb.stride() == ga.stride())))
    if not (ga.is_cuda and same):
        return torch.ops.aten.threshold_backward(ga if gb is None else ga + gb, y, 0)
    out = torch.empty_like(y)
    with _lib.device_guard(y.device):
        code = _lib.lib().mpf_relu_bwd_add(ga.data_ptr(), gb.data_ptr() if gb is not None else None, y.data_ptr(), out.data_ptr(),
                                           y.numel(), _lib.MPF_BF16, _lib.stream_ptr(y.device))
    _lib.check(code, "mpf_relu_bwd_add")
    return out


class _BiasActFork(torch.autograd.Function):
    """relu(x + shift[c] + res) returned TWICE (two aliases of one tensor): the output of a residual block has two consumers in
    the next block (conv1 / shortcut convolution, and the identity skip or the other of the two), whose gradients autograd would
    add with a kernel of its own before the ReLU backward — as separate outputs they arrive separately and are summed inside
    the ReLU-backward pass.  A third consumer (the pixel decoder on a stage's last block) adds into the first alias as usual."""

    @staticmethod
    def forward(ctx, x, shift, res):
        y = torch.empty_like(x)
        code = _lib.lib().mpf_bias_act(x.data_ptr(), shift.data_ptr(), res.data_ptr(), y.data_ptr(), x.numel(), x.shape[1],
                                       _lib.MPF_BF16 if x.dtype == torch.bfloat16 else _lib.MPF_F32, 1,
                                       _lib.stream_ptr(x.device))
        _lib.check(code, "mpf_bias_act")
        ctx.save_for_backward(y)
        ctx.set_materialize_grads(False)
        return y, y.view(y.shape)

    @staticmethod
    def backward(ctx, ga, gb):
        (y,) = ctx.saved_tensors
        if ga is None and gb is None:
            return None, None, None
        if ga is None:
            ga, gb = gb, None
        g = _relu_bwd_add(ga, gb, y)
        return g, None, g


def _bias_act_ok(x, res):
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last)
            and (res is None or (res.dtype == x.dtype and res.shape == x.shape
                                 and res.is_contiguous(memory_format=torch.channels_last))))


def bias_act(x, shift, res=None, relu=True):
    """FrozenBN shift (+ residual) (+ ReLU) after a bias-free folded convolution."""
    if _bias_act_ok(x, res):
        return _BiasAct.apply(x, shift, res, relu)
    y = x + shift.to(x.dtype).view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


def bias_act_fork(x, shift, res):
    """relu(x + shift + res) as two aliases (see _BiasActFork); (y, y) on the paths the native kernel does not take"""
    if _bias_act_ok(x, res) and x.dtype == torch.bfloat16:
        return _BiasActFork.apply(x, shift, res)
    y = bias_act(x, shift, res, True)
    return y, y


class _MaxPool3x3s2(torch.autograd.Function):
    """F.max_pool2d(x, 3, 2, 1) of the stem on a channel-last bf16 activation in native passes (csrc/elementwise.hip): the forward
    keeps one byte per output element (the winner's window position, aten's tie rule), the backward is a gather over the <= 4
    windows of an input pixel — aten's nhwc backward at this size is 112 us for a 17 + 67 MB pass."""

    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        code = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        with _lib.device_guard(x.device):
            rc = _lib.lib().mpf_maxpool3x3s2_forward(x.data_ptr(), y.data_ptr(), code.data_ptr(), N, H, W, C,
                                                     _lib.stream_ptr(x.device))
        _lib.check(rc, "mpf_maxpool3x3s2_forward")
        ctx.save_for_backward(code)
        ctx.in_shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        (code,) = ctx.saved_tensors
        N, C, H, W = ctx.in_shape
        if gy.dtype != torch.bfloat16 or not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx = torch.empty((N, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        with _lib.device_guard(gy.device):
            rc = _lib.lib().mpf_maxpool3x3s2_backward(gy.data_ptr(), code.data_ptr(), gx.data_ptr(), N, H, W, C,
                                                      _lib.stream_ptr(gy.device))
        _lib.check(rc, "mpf_maxpool3x3s2_backward")
        return gx


def max_pool_3x3_s2(x):
    if (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and 256 % (x.shape[1] // 8) == 0
            and x.is_contiguous(memory_format=torch.channels_last)):
        return _MaxPool3x3s2.apply(x)
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def conv_bn(conv, bn, x, folded=None, res=None, relu=True, fork=False):
    """relu?(FrozenBN(conv(x)) + res) with the fixed per-channel scale folded into the convolution
    weight (w' = w * scale: same function, same gradient wrt w) and shift / residual / ReLU as ONE
    element-wise pass over the activation instead of MIOpen's bias kernel + add + ReLU."""
    if folded is not None:
        w, shift = folded
    else:
        scale, shift = bn.scale_bias()
        w = conv.weight * scale.view(-1, 1, 1, 1)
    y = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return bias_act_fork(y, shift, res) if fork else bias_act(y, shift, res, relu)


def _dense(t):
    """memory is one dense block with dim 0 outermost (contiguous or channels_last)"""
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


_gsc_layouts = {}


def _grouped_scale_cast(srcs, scales, out_dtype):
    """[cast(src_i * scale_i[c])] for a list of [C, ...] tensors (dense, dim 0 outermost) in ONE native
    launch (csrc/elementwise.hip); the results are views of one buffer with the sources' strides.
    The layout of a group (sizes, offsets, strides, scale pointers: functions of the shapes and of the frozen scales) is
    computed once per group; per call only the source pointers and the new buffer's base enter the table."""
    dev = srcs[0].device
    key = tuple([(t.shape, t.stride()) for t in srcs] + [id(sc) for sc in scales] + [out_dtype])
    lay = _gsc_layouts.get(key)
    if lay is None:
        numels = [t.numel() for t in srcs]
        offs, tot, blk = [], 0, 0
        table = np.zeros((len(srcs), 6), dtype=np.int64)
        for i, (t, sc) in enumerate(zip(srcs, scales)):
            assert _dense(t) and sc.dtype == torch.float32 and sc.is_contiguous() and sc.numel() == t.shape[0]
            offs.append(tot)
            table[i, 2:] = (sc.data_ptr(), numels[i], numels[i] // t.shape[0], blk)
            tot += (numels[i] + 63) & ~63
            blk += (numels[i] + 2047) // 2048
        esz = torch.empty(0, dtype=out_dtype).element_size()
        lay = (table, np.asarray(offs, dtype=np.int64) * esz, tot, blk, [(o, n, t.shape, t.stride()) for o, n, t in zip(offs, numels, srcs)],
               list(scales))                       # (the scales are kept referenced: their ids are part of the key)
        if len(_gsc_layouts) > 64:
            _gsc_layouts.clear()
        _gsc_layouts[key] = lay
    table, offs_b, tot, blk, views, _ = lay
    out = torch.empty(tot, dtype=out_dtype, device=dev)
    table[:, 0] = [t.data_ptr() for t in srcs]
    table[:, 1] = offs_b + out.data_ptr()
    items = upload(table.reshape(-1), dev)
    dt = {torch.float32: _lib.MPF_F32, torch.bfloat16: _lib.MPF_BF16}
    with _lib.device_guard(dev):
        code = _lib.lib().mpf_grouped_scale_cast(items.data_ptr(), len(srcs), blk, dt[srcs[0].dtype], dt[out_dtype],
                                                 _lib.stream_ptr(dev))
    _lib.check(code, "mpf_grouped_scale_cast")
    return [out[o:o + n].as_strided(shp, std) for o, n, shp, std in views]


class _FoldCast(torch.autograd.Function):
    """All folded conv weights of a stage in ONE launch: w'_i = w_i * scale_i, cast to the autocast dtype;
    the backward casts the gradients back to fp32 and multiplies by scale_i, also in one launch."""

    @staticmethod
    def forward(ctx, dtype, scales, *weights):
        ctx.scales = scales
        ws = [p.detach() for p in weights]
        native = (ws[0].is_cuda and all(w.dtype == torch.float32 and _dense(w) for w in ws)
                  and dtype in (None, torch.float32, torch.bfloat16))
        ctx.native = native
        if native:
            return tuple(_grouped_scale_cast(ws, scales, dtype or torch.float32))
        w = torch._foreach_mul(ws, [s_.view(-1, 1, 1, 1) for s_ in scales])
        if dtype is None:
            return tuple(w)
        outs = [torch.empty_like(t, dtype=dtype) for t in w]
        torch._foreach_copy_(outs, w)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        if ctx.native and all(g is not None and g.dtype == grads[0].dtype for g in grads) \
                and grads[0].dtype in (torch.float32, torch.bfloat16):
            gs = [g if _dense(g) else g.contiguous() for g in grads]
            return (None, None, *_grouped_scale_cast(gs, ctx.scales, torch.float32))
        g32 = [torch.empty_like(g, dtype=torch.float32) for g in grads]
        torch._foreach_copy_(g32, list(grads))
        torch._foreach_mul_(g32, [s_.view(-1, 1, 1, 1) for s_ in ctx.scales])
        return (None, None, *g32)


_PAIR_NAMES = (("conv1", "norm1"), ("conv2", "norm2"), ("conv3", "norm3"), ("shortcut", "shortcut_norm"))


class Bottleneck(nn.Module):
    def __init__(self, cin, cmid, cout, stride):
        super().__init__()
        self.shortcut = None
        if cin != cout or stride != 1:
            self.shortcut = nn.Conv2d(cin, cout, 1, stride=stride, bias=False)
            self.shortcut_norm = FrozenBatchNorm2d(cout)
        self.conv1 = nn.Conv2d(cin, cmid, 1, bias=False)
        self.norm1 = FrozenBatchNorm2d(cmid)
        self.conv2 = nn.Conv2d(cmid, cmid, 3, stride=stride, padding=1, bias=False)   # stride in the 3x3
        self.norm2 = FrozenBatchNorm2d(cmid)
        self.conv3 = nn.Conv2d(cmid, cout, 1, bias=False)
        self.norm3 = FrozenBatchNorm2d(cout)

    def pairs(self):
        """[(conv, norm)]: conv1, conv2, conv3 (, shortcut) — built once per instance (submodules are not replaced after
        construction; ``self.conv1`` on a Module is a ``__getattr__`` round trip, ~430 of them per step came from here)"""
        m = self._modules
        p = self.__dict__.get("_pairs")
        if p is not None:          # still the children this module holds (a swapped submodule rebuilds the list)
            names = _PAIR_NAMES if len(p) == 4 else _PAIR_NAMES[:3]
            if (m.get("shortcut") is None) != (len(p) == 3) or any(c is not m.get(cn) or n is not m.get(nn_) for (c, n), (cn, nn_) in zip(p, names)):
                p = None
        if p is None:
            p = [(self.conv1, self.norm1), (self.conv2, self.norm2), (self.conv3, self.norm3)]
            if self.shortcut is not None:
                p.append((self.shortcut, self.shortcut_norm))
            self.__dict__["_pairs"] = p
        return p

    def forward(self, x, fw=None):
        return self.forward_fork((x, x), fw)[0]

    def forward_fork(self, xs, fw=None):
        """xs = two aliases of the block input (the previous block's forward_fork result, or (x, x)): conv1 reads the first, the
        shortcut convolution / identity skip the second, so their gradients reach the previous block's ReLU backward as two
        arguments (summed there) instead of through an add kernel.  Returns the output as two aliases."""
        f = fw if fw is not None else [None] * 4
        x_main, x_skip = xs
        p = self.pairs()
        out = conv_bn(p[0][0], p[0][1], x_main, f[0])
        out = conv_bn(p[1][0], p[1][1], out, f[1])
        sc = x_skip if len(p) == 3 else conv_bn(p[3][0], p[3][1], x_skip, f[3], relu=False)
        return conv_bn(p[2][0], p[2][1], out, f[2], res=sc, fork=True)


class ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem_conv = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.stem_norm = FrozenBatchNorm2d(64)
        cfg = [("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2)]
        cin = 64
        self.stage_names = []
        for name, n, cmid, cout, stride in cfg:
            blocks = []
            for i in range(n):
                blocks.append(Bottleneck(cin, cmid, cout, stride if i == 0 else 1))
                cin = cout
            setattr(self, name, nn.Sequential(*blocks))
            self.stage_names.append(name)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    @staticmethod
    def _fold_group(pairs, dtype):
        """Fold + cast the conv weights of one stage in grouped launches.  One group per stage (not one for
        the whole network) so that, in backward, the stage's weight gradients become ready as soon as
        the stage has been back-propagated and DDP can start all-reducing them under the earlier stages."""
        sb = [bn.scale_bias() for _, bn in pairs]
        scales = [s_ for s_, _ in sb]
        ws = _FoldCast.apply(dtype, scales, *[c.weight for c, _ in pairs])
        return [(w, b_) for w, (_, b_) in zip(ws, sb)]

    def forward(self, x):
        return self.forward_stages(x, self.stage_names, stem=True)

    def forward_stages(self, x, names, stem=False):
        """The stem (optional) and the named consecutive stages: {name: feature map}.  forward() is the whole chain."""
        return run_stages(x, [(n, getattr(self, n)) for n in names], (self.stem_conv, self.stem_norm) if stem else None)


def run_stages(x, stages, stem=None):
    """stem = (conv, norm) or None; stages = [(name, nn.Sequential of Bottlenecks)] -> {name: feature map}"""
    dtype = torch.get_autocast_dtype("cuda") if (torch.is_autocast_enabled() and x.is_cuda) else None
    if stem is not None:
        st = ResNet50._fold_group([stem], dtype)
        x = conv_bn(stem[0], stem[1], x, st[0])
        x = max_pool_3x3_s2(x)
    out = {}
    for name, stage in stages:
        blocks = list(stage)
        folded = ResNet50._fold_group([p for b in blocks for p in b.pairs()], dtype)
        k = 0
        xs = x if isinstance(x, tuple) else (x, x)
        for b in blocks:
            n = len(b.pairs())
            xs = b.forward_fork(xs, folded[k:k + n])
            k += n
        x = xs
        out[name] = xs[0]
    return out
