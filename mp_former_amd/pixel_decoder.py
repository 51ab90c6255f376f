"""MSDeformAttn pixel decoder — host-side mirror of
mask2former/modeling/pixel_decoder/msdeformattn.py (:23-89 EncoderOnly, :92-131 layer,
:134-161 encoder, :164-358 MSDeformAttnPixelDecoder) with the deformable attention running on the
native HIP kernels (mp_former_amd.msda).  Parameter names are the reference's, so its checkpoints
load with load_state_dict; the constructor takes the reference's explicit (``@configurable``)
keyword arguments.

GPU only: MSDeformAttn raises on CPU tensors (no fallback).
"""
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import normal_

from . import encoder_fused
from . import conv3x3
from .groupnorm import group_norm_flatten, GroupNorm, to_nchw
from .linear import linear_tall
from .msda import MSDeformAttn, attach_host_shapes


class ShapeSpec:
    """Minimal stand-in for detectron2.layers.ShapeSpec (channels / stride are all that is read)."""

    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class PositionEmbeddingSine(nn.Module):
    """transformer_decoder/position_encoding.py:12-52 with mask=None; the embedding only depends
    on (H, W), so it is computed once per shape and cached (the reference recomputes it 6x/step)."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def forward(self, x, mask=None):
        assert mask is None, "only the mask-free path of the hot path is implemented"
        N, _, H, W = x.shape
        key = (H, W, x.device)
        pos = self._cache.get(key)
        if pos is None:
            y = torch.arange(1, H + 1, dtype=torch.float32, device=x.device)[:, None].expand(H, W)
            xx = torch.arange(1, W + 1, dtype=torch.float32, device=x.device)[None, :].expand(H, W)
            if self.normalize:
                eps = 1e-6
                y = y / (H + eps) * self.scale
                xx = xx / (W + eps) * self.scale
            dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
            dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / self.num_pos_feats)
            px = xx[:, :, None] / dim_t
            py = y[:, :, None] / dim_t
            px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
            py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
            pos = torch.cat((py, px), dim=2).permute(2, 0, 1).contiguous()
            self._cache[key] = pos
        return pos[None].expand(N, -1, -1, -1)


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        assert activation == "relu"
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        src2 = self.self_attn(src if pos is None else src + pos, reference_points, src, spatial_shapes,
                              level_start_index, padding_mask)
        src = self.norm1(src + self.dropout1(src2))
        src2 = linear_tall(self.dropout2(F.relu(linear_tall(src, self.linear1.weight, self.linear1.bias))),
                           self.linear2.weight, self.linear2.bias)
        return self.norm2(src + self.dropout3(src2))


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, layer_args, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([MSDeformAttnTransformerEncoderLayer(*layer_args) for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes_list, n, device):
        """msdeformattn.py:141-153 with valid_ratios == 1: pixel centres, broadcast over levels."""
        refs = []
        for H, W in spatial_shapes_list:
            ys = torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device) / H
            xs = torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device) / W
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            refs.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), -1))
        ref = torch.cat(refs, 0)
        return ref[None, :, None, :].expand(n, -1, len(spatial_shapes_list), -1)

    def forward(self, src, spatial_shapes, level_start_index, shapes_list, pos=None, padding_mask=None):
        output = src
        ref = self.get_reference_points(shapes_list, src.shape[0], src.device)
        for layer in self.layers:
            output = layer(output, pos, ref, spatial_shapes, level_start_index, padding_mask)
        return output


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = MSDeformAttnTransformerEncoder(
            (d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, enc_n_points),
            num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._reset_parameters()
        self._shape_cache = {}

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    def _fused_ok(self, srcs, pos_embeds):
        """The single-node encoder (encoder_fused.py) applies: fp32 CUDA inputs, inactive dropout, a
        batch-invariant positional embedding, 32 channels per head."""
        if os.environ.get("MPF_FUSED_ENCODER", "1") == "0":
            return False
        lay = self.encoder.layers[0]
        if self.training and (lay.dropout1.p > 0 or lay.dropout2.p > 0 or lay.dropout3.p > 0):
            return False
        if self.d_model != 256 or self.nhead != 8:       # the native LayerNorm / MSDA kernels of this path
            return False
        for s_, p_ in zip(srcs, pos_embeds):
            if not (s_.is_cuda and s_.dtype == torch.float32 and p_.dtype == torch.float32):
                return False
            if p_.shape[0] != 1 and p_.stride(0) != 0:
                return False
        return True

    def _fused_meta(self, shapes_list, spatial_shapes, level_start_index, device):
        key = ("meta", tuple(shapes_list), device)
        meta = self._shape_cache.get(key)
        if meta is None:
            lay = self.encoder.layers[0].self_attn
            ref = MSDeformAttnTransformerEncoder.get_reference_points(shapes_list, 1, device)[0, :, 0, :].contiguous()
            sizes = [h * w for h, w in shapes_list]
            meta = dict(
                n_heads=lay.n_heads, n_levels=lay.n_levels, n_points=lay.n_points, shapes=spatial_shapes,
                lsi=level_start_index, ref=ref, sizes=sizes,
                normalizer=torch.tensor([[w, h] for h, w in shapes_list], dtype=torch.float32, device=device),
                level_idx=torch.repeat_interleave(torch.arange(len(sizes), device=device),
                                                  torch.tensor(sizes, device=device)))
            self._shape_cache[key] = meta
        return meta

    def forward(self, srcs, pos_embeds, src_flatten=None):
        """srcs: the projected levels [N, C, H_l, W_l]; src_flatten: their flatten(2).transpose(1, 2) concatenation if the
        caller already has it in one buffer (then srcs only give the shapes)."""
        shapes_list = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
        key = (tuple(shapes_list), srcs[0].device)
        cached = self._shape_cache.get(key)
        if cached is None:
            ss = torch.as_tensor(shapes_list, dtype=torch.long, device=srcs[0].device)
            lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
            # host copy of the geometry + the level starts it vouches for (lsi is the running sum by construction: no sync)
            attach_host_shapes(ss, shapes_list)
            ss._mpf_lsi = lsi
            cached = (ss, lsi)
            self._shape_cache[key] = cached
        spatial_shapes, level_start_index = cached
        if src_flatten is None:
            src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        if self._fused_ok(srcs, pos_embeds):
            meta = self._fused_meta(shapes_list, spatial_shapes, level_start_index, srcs[0].device)
            # [S, C] positional rows: a function of the (cached) embeddings only — rebuilt when they change, not per step
            pkey = tuple((p.data_ptr(), tuple(p.shape), p._version) for p in pos_embeds)     # (the held views pin the storage)
            hit = self._shape_cache.get("pos_const")
            if hit is None or hit[0] != pkey:
                hit = (pkey, torch.cat([p[0].flatten(1).t() for p in pos_embeds], 0).contiguous(), list(pos_embeds))
                self._shape_cache["pos_const"] = hit
            pos_const = hit[1]
            params = [t for layer in self.encoder.layers for t in encoder_fused.layer_params(layer)]
            memory = encoder_fused.EncoderFn.apply(src_flatten, pos_const, self.level_embed, meta, *params)
            return memory, spatial_shapes, level_start_index
        lvl_pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[l].view(1, 1, -1)
                             for l, p in enumerate(pos_embeds)], 1)
        memory = self.encoder(src_flatten, spatial_shapes, level_start_index, shapes_list, lvl_pos, None)
        return memory, spatial_shapes, level_start_index


def _conv(m, x, out_dtype=torch.float32):
    """m(x) for an nn.Conv2d-like module, on the split-bf16 GEMM when the input is channel-last planes and the shape is one
    the GEMM forms take (3x3 / stride 1 / pad 1, or 1x1), else the library convolution.  x may be a bf16 feature map (1x1
    only; anything else gets the ``.float()`` of msdeformattn.py:320); out_dtype = bf16 asks the 1x1 form for a bf16 result."""
    if m.groups == 1 and m.dilation == (1, 1) and m.stride == (1, 1):
        if m.kernel_size == (1, 1) and m.padding == (0, 0):
            if not conv3x3.supported_1x1(x, m.weight) and x.is_cuda and x.dim() == 4:
                # an NCHW map: one relayout to channel-last planes, then the native GEMM (the library's convolution is not run-to-run
                # reproducible for every solver MIOpen may pick — see the 3x3 case below)
                xp = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
                if conv3x3.supported_1x1(xp, m.weight):
                    x = xp
            if conv3x3.supported_1x1(x, m.weight):
                return conv3x3.conv1x1(x, m.weight, m.bias, out_dtype)
        x = x.float()
        if m.kernel_size == (3, 3) and m.padding == (1, 1):
            if not conv3x3.supported(x, m.weight) and x.is_cuda and x.dim() == 4:
                # an NCHW map: one relayout to channel-last planes, then the native kernel.  (The library's fp32 3x3 convolution
                # that this route used to fall back to is not run-to-run reproducible — MIOpen picks split-K / Winograd solvers by
                # the state of its caches — which showed as a rare 3e-3 deviation of a whole forward pass in the route tests.)
                xp = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
                if conv3x3.supported(xp, m.weight):
                    x = xp
            if conv3x3.supported(x, m.weight):
                return conv3x3.conv3x3(x, m.weight, m.bias)
    y = F.conv2d(x.float(), m.weight, m.bias, m.stride, m.padding, m.dilation, m.groups)
    return y.to(out_dtype) if out_dtype != torch.float32 else y


class _ConvNorm(nn.Conv2d):
    """detectron2.layers.Conv2d: conv -> optional norm -> optional activation; the norm lives under
    the attribute ``norm`` (state-dict keys ``<name>.norm.weight``)."""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = _conv(self, x)
        if isinstance(self.norm, GroupNorm) and self.activation in (None, F.relu) and self.norm.cl_ok(x):
            return self.norm.forward_cl(x, relu=self.activation is F.relu)     # norm (+ ReLU) in one pass, channel-last
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, channels):
    if norm is None or norm == "":
        return None
    if norm != "GN":
        raise ValueError(f"only GN / no norm are supported by this mirror, got {norm!r}")
    return GroupNorm(32, channels)


def _c2_xavier_fill(m):
    nn.init.kaiming_uniform_(m.weight, a=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


class MSDeformAttnPixelDecoder(nn.Module):
    """msdeformattn.py:164-358.  ``forward_features(features) -> (mask_features, out[0],
    multi_scale_features[3])`` (the SEM_SEG_HEADS_REGISTRY pixel-decoder contract)."""

    def __init__(self, input_shape: Dict[str, ShapeSpec], *, transformer_dropout: float, transformer_nheads: int,
                 transformer_dim_feedforward: int, transformer_enc_layers: int, conv_dim: int, mask_dim: int,
                 norm: Optional[str] = None, transformer_in_features: List[str], common_stride: int):
        super().__init__()
        tr_shape = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.feature_strides = [v.stride for k, v in input_shape]
        self.feature_channels = [v.channels for k, v in input_shape]
        tr_shape = sorted(tr_shape.items(), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, v in tr_shape]
        tr_channels = [v.channels for k, v in tr_shape]
        self.transformer_feature_strides = [v.stride for k, v in tr_shape]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        chans = tr_channels[::-1] if self.transformer_num_feature_levels > 1 else [tr_channels[-1]]
        self.input_proj = nn.ModuleList([
            nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), GroupNorm(32, conv_dim)) for c in chans])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = _ConvNorm(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        _c2_xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(np.log2(stride) - np.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[:self.num_fpn_levels]):
            lateral_conv = _ConvNorm(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=_get_norm(norm, conv_dim))
            output_conv = _ConvNorm(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                    norm=_get_norm(norm, conv_dim), activation=F.relu)
            _c2_xavier_fill(lateral_conv)
            _c2_xavier_fill(output_conv)
            self.add_module("adapter_{}".format(idx + 1), lateral_conv)
            self.add_module("layer_{}".format(idx + 1), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]

    @classmethod
    def from_config(cls, cfg, input_shape):
        """Same mapping as msdeformattn.py:294-312 (detectron2 CfgNode)."""
        return dict(
            input_shape={k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
            conv_dim=cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
            norm=cfg.MODEL.SEM_SEG_HEAD.NORM, transformer_dropout=cfg.MODEL.MASK_FORMER.DROPOUT,
            transformer_nheads=cfg.MODEL.MASK_FORMER.NHEADS, transformer_dim_feedforward=1024,
            transformer_enc_layers=cfg.MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS,
            transformer_in_features=cfg.MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES,
            common_stride=cfg.MODEL.SEM_SEG_HEAD.COMMON_STRIDE)

    def forward_features(self, features):
        # the reference pins the whole pixel decoder to fp32 even under AMP (msdeformattn.py:314,320)
        amp_bf16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
        with torch.autocast(device_type="cuda", enabled=False):
            return self._forward_features_fp32(features, amp_bf16 and os.environ.get("MPF_MASK_FEATURES_BF16", "1") == "1")

    def _forward_features_fp32(self, features, mask_features_bf16=False):
        srcs, pos, convs = [], [], []
        for idx, f in enumerate(self.transformer_in_features[::-1]):
            x = features[f]                       # (bf16 under autocast: the 1x1 GEMM form takes it as it is; else .float())
            proj = self.input_proj[idx]
            convs.append(_conv(proj[0], x) if len(proj) == 2 else None)
            pos.append(self.pe_layer(x))
        src_flatten = None
        if all(c is not None for c in convs) and os.environ.get("MPF_GN_FLATTEN", "1") == "1":
            # conv1x1 -> GroupNorm -> flatten -> concat (msdeformattn.py:319-322, :60-66): the norm's apply pass writes
            # each level straight into its rows of the encoder's [N, S, C] input
            src_flatten = group_norm_flatten([p[1] for p in self.input_proj], convs)
        if src_flatten is not None:
            srcs = convs                          # (shapes only)
        else:
            for idx, f in enumerate(self.transformer_in_features[::-1]):
                proj = self.input_proj[idx]
                srcs.append(proj[1](convs[idx]) if convs[idx] is not None else proj(features[f].float()))
        y, spatial_shapes, level_start_index = self.transformer(srcs, pos, src_flatten)
        bs = y.shape[0]
        sizes = [int(s.shape[2]) * int(s.shape[3]) for s in srcs]
        out = [z.transpose(1, 2).reshape(bs, -1, srcs[i].shape[2], srcs[i].shape[3])
               for i, z in enumerate(torch.split(y, sizes, dim=1))]
        for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
            x = features[f]
            lat = self.lateral_convs[idx]
            if isinstance(lat.norm, GroupNorm) and lat.activation is None:
                z = _conv(lat, x)
                if lat.norm.cl_ok(z, out[-1]):
                    # norm(lateral) + upsample2x(top) in the norm's apply pass (msdeformattn.py:349-351)
                    out.append(self.output_convs[idx](lat.norm.forward_cl(z, top=out[-1])))
                    continue
            cur_fpn = self.lateral_convs[idx](x.float())
            top = to_nchw(out[-1])
            y = cur_fpn + F.interpolate(top, size=cur_fpn.shape[-2:], mode="bilinear", align_corners=False)
            out.append(self.output_convs[idx](y))
        multi_scale_features = out[:self.maskformer_num_feature_levels]
        mfc = self.mask_features
        if (mask_features_bf16 and mfc.norm is None and mfc.activation is None and mfc.kernel_size == (1, 1)
                and conv3x3.supported_1x1(out[-1], mfc.weight)):
            # under bf16 autocast the only consumer (the decoder) casts mask_features to bf16 first thing: emit them in bf16
            # from the GEMM epilogue (the same single rounding) and take their gradient in bf16 — two 134 MB cast passes less
            mf = _conv(mfc, out[-1], torch.bfloat16)
        else:
            mf = mfc(out[-1])
        return mf, out[0], multi_scale_features
