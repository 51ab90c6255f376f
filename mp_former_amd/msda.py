"""Host-side mirror of the reference's MSDeformAttn operator interface, backed by the HIP kernels.

Mirrors (same names, argument meaning and error behaviour):
  * the pybind op module ``MultiScaleDeformableAttention`` — ``ms_deform_attn_forward`` /
    ``ms_deform_attn_backward`` (reference: mask2former/modeling/pixel_decoder/ops/src/vision.cpp:18-21,
    ops/src/ms_deform_attn.h:25-66, ops/src/cuda/ms_deform_attn_cuda.cu:25-158);
  * ``MSDeformAttnFunction`` (ops/functions/ms_deform_attn_func.py:32-49);
  * ``MSDeformAttn`` (ops/modules/ms_deform_attn.py:34-125) — but WITHOUT the reference's bare
    ``except:`` fallback to the grid_sample path (:119-121): a failing native op raises.
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.init import constant_, xavier_uniform_

from . import _lib
from .linear import linear_tall

_DTYPES = {torch.float32: _lib.MPF_F32, torch.float64: _lib.MPF_F64}


def _check_inputs(named):
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")  # ms_deform_attn_cuda.cu:33-37
        if not t.is_cuda:
            if name == "value":
                raise RuntimeError("Not implemented on the CPU")  # ms_deform_attn.h:43,65
            raise RuntimeError(f"{name} must be a CUDA tensor")  # ms_deform_attn_cuda.cu:39-43


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if value.dtype not in _DTYPES:
        # AT_DISPATCH_FLOATING_TYPES: float / double only (ms_deform_attn_cuda.cu:69)
        raise RuntimeError(f'"ms_deform_attn" not implemented for \'{value.dtype}\'')
    for name, t in (("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if t.dtype != value.dtype:
            raise RuntimeError(f"{name} dtype {t.dtype} does not match value dtype {value.dtype}")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 tensors")
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (batch, num_query, num_heads, num_levels, num_point, 2):
        raise RuntimeError(f"sampling_loc has shape {tuple(sampling_loc.shape)}")
    if tuple(attn_weight.shape) != (batch, num_query, num_heads, num_levels, num_point):
        raise RuntimeError(f"attn_weight has shape {tuple(attn_weight.shape)}")
    if tuple(level_start_index.shape) != (num_levels,) or tuple(spatial_shapes.shape) != (num_levels, 2):
        raise RuntimeError("spatial_shapes must be [L,2] and level_start_index [L]")
    im2col_step_ = min(batch, int(im2col_step))
    if im2col_step_ <= 0 or batch % im2col_step_ != 0:  # ms_deform_attn_cuda.cu:55-57
        raise RuntimeError(f"batch({batch}) must divide im2col_step({im2col_step_})")
    return batch, spatial_size, num_heads, channels, num_levels, num_query, num_point


def _stream(t):
    return _lib.stream_ptr(t.device)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step,
                           host_shapes=None):
    """-> output [N, Lq, M*D]; freshly allocated, computed on the current stream.  The spatially blocked production kernel runs
    either way: with the level geometry from the host (`host_shapes`, or the copy attached by `attach_host_shapes`: one launch),
    or — the reference's call, device tensors only — with the geometry derived by a prologue kernel on the device
    (mpf_msda_forward_dev).  No device->host copy is ever made here: the op never blocks."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)])
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    hs = host_shapes if host_shapes is not None else _attached_host_shapes(spatial_shapes, level_start_index)
    lib = _lib.lib()
    with _lib.device_guard(value.device):
        if hs is None and _dev_applicable(value, D, L, P):
            # the reference's call (func.py:36): device tensors only -> geometry derived on the device, same blocked kernel
            ws = _workspace(value.device, lib.mpf_msda_dev_workspace_bytes(N, S, M, L, Lq, P, 0))
            code = lib.mpf_msda_forward_dev(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), out.data_ptr(), N, S, M, D, L, Lq, P, _DTYPES[value.dtype], ws.data_ptr(), ws.numel(),
                _stream(value))
        else:
            code = lib.mpf_msda_forward_hs(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                hs.data_ptr() if hs is not None else None,
                sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(),
                N, S, M, D, L, Lq, P, _DTYPES[value.dtype], _stream(value))
    _lib.check(code, "mpf_msda_forward")
    return out


# backward formulation: "auto" = atomics-free binned kernels when applicable, else the atomic kernels;
# "atomic" forces the reference-style scatter with hardware fp32 atomics; "binned" requires the binned path.
BWD_MODE = "auto"

_workspaces = {}


def _contiguous_starts(hs):
    return torch.cat((hs.new_zeros(1), (hs[:, 0] * hs[:, 1]).cumsum(0)[:-1]))


def attach_host_shapes(spatial_shapes, shapes_list, level_start_index=None):
    """Remember the host-side (H, W) list on a device `spatial_shapes` tensor so that the blocked kernels
    need no device->host copy (the pixel decoder builds the tensor from python ints).  The blocked
    kernels assume levels stored back to back; the device `level_start_index` the copy belongs to is
    remembered too (by identity), so a caller passing a DIFFERENT level_start_index later is detected
    host-side and takes the kernels that honour it (the attached copy used to bypass that check)."""
    hs = torch.as_tensor(shapes_list, dtype=torch.int64).contiguous()
    spatial_shapes._mpf_host = hs
    spatial_shapes._mpf_lsi = None
    if level_start_index is not None:
        if not torch.equal(level_start_index.cpu(), _contiguous_starts(hs)):
            raise ValueError("attach_host_shapes: level_start_index is not the running sum of H*W")
        spatial_shapes._mpf_lsi = level_start_index
    return spatial_shapes


def _attached_host_shapes(spatial_shapes, level_start_index):
    """The attached host copy, if it is known to describe `level_start_index` too (no device sync)."""
    hs = getattr(spatial_shapes, "_mpf_host", None)
    if hs is None:
        return None
    known = getattr(spatial_shapes, "_mpf_lsi", None)
    if known is not None and known is not level_start_index and known.data_ptr() != level_start_index.data_ptr():
        return None          # a different level_start_index: unknown layout -> kernels that read it from the device
    return hs


def _host_shapes(spatial_shapes, level_start_index):
    """Host copy of spatial_shapes if the caller attached one that vouches for `level_start_index`, else None — never a
    device->host copy (the backward used to call `.cpu()` here, a blocking copy per call; callers without host
    shapes now get the geometry built on the device, mpf_msda_*_dev)."""
    hs = _attached_host_shapes(spatial_shapes, level_start_index)
    if hs is not None and getattr(spatial_shapes, "_mpf_lsi", None) is not None:
        return hs
    return None


def _dev_applicable(value, D, L, P):
    """shapes of the blocked kernels (csrc/msda_block.hip): fp32, 32 channels per head, 4 points, at most 4 levels"""
    return value.dtype == torch.float32 and D == 32 and P == 4 and 1 <= L <= 4


def _workspace(device, nbytes):
    key = (device, _lib.ws_scope())          # (a graph capture has its own buffers: _lib.workspace_scope)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _binned_applicable(value, D, L, P):
    return value.dtype == torch.float32 and D == 32 and L <= 8 and L * P <= 32


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step, host_shapes=None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]"""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight), ("grad_output", grad_output)])
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    if grad_output.dtype != value.dtype or grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must be [N, Lq, M*D] in the dtype of value")
    gv = torch.empty_like(value)           # fully written by the native call
    gl = torch.empty_like(sampling_loc)
    ga = torch.empty_like(attn_weight)
    hs = None
    if BWD_MODE != "atomic" and _binned_applicable(value, D, L, P):
        hs = host_shapes if host_shapes is not None else _host_shapes(spatial_shapes, level_start_index)
    if hs is None and BWD_MODE != "atomic" and _dev_applicable(value, D, L, P):
        # the reference's call (func.py:46): device tensors only -> bin + tile kernels on a geometry built on the device
        lib = _lib.lib()
        need = lib.mpf_msda_dev_workspace_bytes(N, S, M, L, Lq, P, 1)
        if need:
            ws = _workspace(value.device, need)
            with _lib.device_guard(value.device):
                code = lib.mpf_msda_backward_dev(
                    value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                    attn_weight.data_ptr(), grad_output.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
                    N, S, M, D, L, Lq, P, _DTYPES[value.dtype], ws.data_ptr(), ws.numel(), _stream(value))
            _lib.check(code, "mpf_msda_backward_dev")
            return [gv, gl, ga]
    if BWD_MODE == "binned" and hs is None:
        raise RuntimeError("MSDA binned backward requested but not applicable (needs fp32, D=32, L<=8, L*P<=32, host shapes)")
    if hs is not None:
        lib = _lib.lib()
        need = lib.mpf_msda_backward_workspace_bytes(N, M, L, Lq, P, hs.data_ptr())
        if need == 0:
            raise RuntimeError("mpf_msda_backward_workspace_bytes rejected the level geometry")
        ws = _workspace(value.device, need)
        with _lib.device_guard(value.device):
            code = lib.mpf_msda_backward_ws(
                value.data_ptr(), hs.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
                grad_output.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
                N, S, M, D, L, Lq, P, _DTYPES[value.dtype], ws.data_ptr(), ws.numel(), _stream(value))
        _lib.check(code, "mpf_msda_backward_ws")
        return [gv, gl, ga]
    with _lib.device_guard(value.device):
        code = _lib.lib().mpf_msda_backward(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
            gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
            N, S, M, D, L, Lq, P, _DTYPES[value.dtype], _stream(value))
    _lib.check(code, "mpf_msda_backward")
    return [gv, gl, ga]


def ms_deform_attn_forward_raw(value, spatial_shapes, level_start_index, raw, ref_points, host_shapes=None):
    """Module-level forward (mpf_msda_forward_raw): value [N,S,M,32] fp32, raw [N*Lq, M*L*P*3] (sampling
    offsets | attention logits of ops/modules/ms_deform_attn.py:103-106), ref_points [Lq,2] (the same
    point for every level) -> (out [N,Lq,M*32], loc [N,Lq,M,L,P,2], attn [N,Lq,M,L,P])."""
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq = ref_points.shape[0]
    P = raw.shape[1] // (M * L * 3)
    assert raw.shape == (N * Lq, M * L * P * 3) and raw.is_contiguous() and ref_points.is_contiguous() and value.is_contiguous()
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    loc = torch.empty((N, Lq, M, L, P, 2), dtype=value.dtype, device=value.device)
    attn = torch.empty((N, Lq, M, L, P), dtype=value.dtype, device=value.device)
    with _lib.device_guard(value.device):
        code = _lib.lib().mpf_msda_forward_raw_hs(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            host_shapes.data_ptr() if host_shapes is not None else None, raw.data_ptr(), ref_points.data_ptr(),
            loc.data_ptr(), attn.data_ptr(), out.data_ptr(),
            N, S, M, D, L, Lq, P, _DTYPES[value.dtype], _stream(value))
    _lib.check(code, "mpf_msda_forward_raw_hs")
    return out, loc, attn


def ms_deform_attn_backward_raw(value, host_shapes, sampling_loc, attn_weight, grad_output, output=None, graw_amax=None, gv_amax=None):
    """-> (grad_value, grad_raw [N*Lq, M*L*P*3]) with the atomics-free kernels (mpf_msda_backward_ws_raw; with the forward
    result ``output`` [N, Lq, M*32]: mpf_msda_backward_ws_raw_o, the destination-side bin + tile kernels, which can also record
    the largest magnitudes of both results in amax slots ``graw_amax`` / ``gv_amax`` — zeroed by the caller)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    gv = torch.empty_like(value)
    graw = torch.empty((N * Lq, M * L * P * 3), dtype=value.dtype, device=value.device)
    lib = _lib.lib()
    need = lib.mpf_msda_backward_workspace_bytes(N, M, L, Lq, P, host_shapes.data_ptr())
    if need == 0:
        raise RuntimeError("mpf_msda_backward_workspace_bytes rejected the level geometry")
    ws = _workspace(value.device, need)
    with _lib.device_guard(value.device):
        if output is not None:
            assert output.is_contiguous() and output.numel() == N * Lq * M * D and output.dtype == value.dtype
            code = lib.mpf_msda_backward_ws_raw_o(
                value.data_ptr(), host_shapes.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
                grad_output.data_ptr(), output.data_ptr(), gv.data_ptr(), graw.data_ptr(),
                N, S, M, D, L, Lq, P, _DTYPES[value.dtype], ws.data_ptr(), ws.numel(),
                graw_amax.data_ptr() if graw_amax is not None else None, gv_amax.data_ptr() if gv_amax is not None else None,
                _stream(value))
        else:
            code = lib.mpf_msda_backward_ws_raw(
                value.data_ptr(), host_shapes.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
                grad_output.data_ptr(), gv.data_ptr(), graw.data_ptr(),
                N, S, M, D, L, Lq, P, _DTYPES[value.dtype], ws.data_ptr(), ws.numel(), _stream(value))
    _lib.check(code, "mpf_msda_backward_ws_raw")
    return gv, graw


class MSDeformAttnFunction(Function):
    """ops/functions/ms_deform_attn_func.py:32-49."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.host_shapes = _attached_host_shapes(value_spatial_shapes, value_level_start_index)
        if ctx.host_shapes is not None and getattr(value_spatial_shapes, "_mpf_lsi", None) is None:
            ctx.host_shapes = None      # attached without its level_start_index: layout not vouched for
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                        sampling_locations, attention_weights, ctx.im2col_step, ctx.host_shapes)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, attn,
                                             grad_output.contiguous(), ctx.im2col_step, ctx.host_shapes)
        return gv, None, None, gl, ga, None


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention module; parameter names / init / forward signature of
    ops/modules/ms_deform_attn.py:34-125 so reference checkpoints load unchanged."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a per-head dimension that is not 32 takes the generic (slow) kernels")
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2)
        grid_init = grid_init.repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == Len_in
        value = linear_tall(input_flatten, self.value_proj.weight, self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        sampling_offsets = linear_tall(query, self.sampling_offsets.weight, self.sampling_offsets.bias).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        attention_weights = linear_tall(query, self.attention_weights.weight, self.attention_weights.bias).view(
            N, Len_q, self.n_heads, self.n_levels * self.n_points)
        attention_weights = F.softmax(attention_weights, -1).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] != 2:
            # the reference also accepts 4 numbers per point (box-refinement detectors, ms_deform_attn.py:111-113); nothing on the
            # segmentation path produces them (msdeformattn.py:72-86 builds 2-d points), so that form is not carried here
            raise ValueError("reference_points must be [N, Len_q, n_levels, 2] on this path, got last dim {}".format(
                reference_points.shape[-1]))
        wh = input_spatial_shapes.flip(-1).to(sampling_offsets.dtype)                  # (W_l, H_l): x is normalised by the width
        sampling_locations = reference_points[:, :, None, :, None, :] + sampling_offsets / wh[None, None, None, :, None, :]
        # strict: no silent fallback
        output = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                            sampling_locations.contiguous(), attention_weights,
                                            self.im2col_step)
        return linear_tall(output, self.output_proj.weight, self.output_proj.bias)
