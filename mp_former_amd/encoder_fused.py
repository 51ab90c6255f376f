"""The six MSDeformAttn encoder layers of the pixel decoder as ONE autograd node.

Same math as ``MSDeformAttnTransformerEncoderLayer`` x num_layers (reference
mask2former/modeling/pixel_decoder/msdeformattn.py:92-161 with ops/modules/ms_deform_attn.py:82-125
inside, dropout inactive, no padding mask, valid_ratios == 1), scheduled by hand:

* every Linear (forward and input gradient) runs on the split-bf16 fp32 GEMM (csrc/gemm3.hip) with
  its neighbours folded into the epilogue: bias / ReLU / the residual add / the ReLU-backward gate /
  gradient accumulation; ``sampling_offsets`` and ``attention_weights`` are one 288-wide GEMM;
* weight gradients use the split-K "NT" form of the same GEMM, which also returns the column sums of
  the gradient operand (the bias gradients) from its staging loop;
* the positional term is ``pos + level_embed[level]``: its gradient only reaches ``level_embed``, so
  it is obtained from per-level column sums of the 288-wide gradient instead of accumulating a
  [N, S, 256] tensor over the layers.

Used by ``MSDeformAttnTransformerEncoderOnly`` when the inputs are fp32 CUDA tensors and dropout is
inactive; ``MPF_FUSED_ENCODER=0`` selects the layer-by-layer modules (same results to fp32 round-off;
tests/test_encoder_fused_gpu.py).
"""
import ctypes
import math
import os

import numpy as np
import torch
from torch.autograd import Function

from . import _lib
from .gemm3 import amax, amax_slots, gemm3_h2, gemm3_h2_bits, gemm3_nt, gemm3_nt_grouped, nt_reduce_levels, split_weights_grouped_h2
from .msda import ms_deform_attn_backward_raw, ms_deform_attn_forward_raw
from .resln import LnGradGroup, ln256_forward

PARAMS_PER_LAYER = 16
_EPS = 1e-5

# Range audit of the fp16 x 2 operands (tools/soak.py --range-audit): None, or {operand name: int64 [41] device
# histogram}: bin b counts the rows whose largest magnitude r (relative to what the operand's amax SLOT holds: the scale's
# reference — the true maximum or the upper bound of amax.h) lies in (2^-(b+1), 2^-b], bin 40 everything at or below 2^-40
# incl. all-zero rows.  (floor(-log2 r), not -floor(log2 r), which put every row one bin too high.)
RANGE_AUDIT = None

# Run-time range guard: the fp16 x 2 guarantee is relative to each operand's largest magnitude, so what it
# promises for a ROW depends on how far that row lies below it.  Every RANGE_GUARD_EVERY-th call of the encoder (the third,
# then every 64th: MPF_H2_RANGE_GUARD_EVERY, 0 = off) runs the python-sequenced route and passes every GEMM operand, forward and
# backward, through mpf_h2_range_stats (one read of the operand, two device counters per operand, no synchronisation): rows
# with a non-zero element, and those whose largest magnitude lies below 2^-18 of the slot (where the second piece stops being a
# normal fp16 number).  The counters are copied to pinned memory at the next guarded call and read at the one after (no wait);
# a share above RANGE_GUARD_SHARE warns once per process.  range_guard_report(sync=True) reads them now (tools/soak.py, tests).
RANGE_GUARD_EVERY = int(os.environ.get("MPF_H2_RANGE_GUARD_EVERY", "64"))
RANGE_GUARD_LOG2 = 18
RANGE_GUARD_SHARE = 1e-3
_GUARD_SLOTS = 64
_guard = {"calls": 0, "active": False, "names": {}, "counters": None, "pending": None, "host": None, "warned": False}


def _guard_begin():
    """-> True when this encoder call is a guarded one (counts the call)."""
    n = _guard["calls"]
    _guard["calls"] = n + 1
    if RANGE_GUARD_EVERY <= 0 or RANGE_AUDIT is not None:
        return False
    if not (n == 2 or (n > 2 and (n - 2) % RANGE_GUARD_EVERY == 0) or RANGE_GUARD_EVERY == 1):
        return False
    if _guard["pending"] is not None and _guard["pending"][1].query():          # the copy requested at the last guarded call has landed
        _guard["host"] = _guard["pending"][0].clone()
        _guard["pending"] = None
        _guard_check(_guard["host"])
    if _guard["counters"] is not None and _guard["pending"] is None:
        pinned = torch.empty((_GUARD_SLOTS, 2), dtype=torch.int64).pin_memory()
        pinned.copy_(_guard["counters"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _guard["pending"] = (pinned, ev)
    return True


def _guard_shares(host):
    out = {}
    for name, i in _guard["names"].items():
        nz, below = int(host[i, 0]), int(host[i, 1])
        out[name] = {"rows": nz, "below": below, "share": below / nz if nz else 0.0}
    return out


def _guard_check(host):
    bad = {k: v for k, v in _guard_shares(host).items() if v["share"] > RANGE_GUARD_SHARE}
    if bad and not _guard["warned"]:
        import warnings
        _guard["warned"] = True
        worst = max(bad.items(), key=lambda kv: kv[1]["share"])
        warnings.warn("mp_former_amd: fp16 x 2 GEMM operand '%s' has %.3f %% of its non-zero rows below 2^-%d of the operand's largest "
                      "magnitude (%d of %d): those rows keep fewer than 22 bits in the pixel decoder's fp32 GEMMs (DESIGN.md section 2); "
                      "%d operand(s) over the %.1f %% bar" % (worst[0], 100 * worst[1]["share"], RANGE_GUARD_LOG2, worst[1]["below"],
                                                               worst[1]["rows"], len(bad), 100 * RANGE_GUARD_SHARE))


def range_guard_report(sync=False, reset=False):
    """{operand: {"rows", "below", "share"}} of the guarded calls so far: from the last landed copy, or (sync=True) from the
    device counters now — which also applies the warning check."""
    if _guard["counters"] is None:
        return {}
    if sync:
        host = _guard["counters"].cpu()
        _guard_check(host)
    else:
        host = _guard["host"]
        if host is None:
            return {}
    rep = _guard_shares(host)
    if reset:
        _guard["counters"].zero_()
        _guard["host"] = _guard["pending"] = None
        _guard["warned"] = False
    return rep


def _audit(name, t, slot):
    if _guard["active"]:
        if _guard["counters"] is None or _guard["counters"].device != t.device:
            _guard["counters"] = torch.zeros((_GUARD_SLOTS, 2), dtype=torch.int64, device=t.device)
        i = _guard["names"].setdefault(name, len(_guard["names"]))
        if i < _GUARD_SLOTS and t.dim() == 2 and t.stride(1) == 1 and t.shape[1] % 4 == 0 and t.stride(0) % 4 == 0:
            with _lib.device_guard(t.device):
                code = _lib.lib().mpf_h2_range_stats(t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), slot.data_ptr(), RANGE_GUARD_LOG2,
                                                     _guard["counters"][i].data_ptr(), _lib.stream_ptr(t.device))
            _lib.check(code, "mpf_h2_range_stats")
    if RANGE_AUDIT is None:
        return
    from .gemm3 import amax_value
    ref = amax_value(slot).clamp_min(1e-38)
    r = (t.detach().abs().amax(1) / ref).clamp(2.0 ** -41, 1.0)
    b = torch.floor(-torch.log2(r)).clamp(0, 40).long()
    h = torch.bincount(b, minlength=41)
    RANGE_AUDIT[name] = h if name not in RANGE_AUDIT else RANGE_AUDIT[name] + h


_PARAM_SLOTS = (("self_attn", "sampling_offsets"), ("self_attn", "attention_weights"), ("self_attn", "value_proj"),
                ("self_attn", "output_proj"), ("norm1",), ("linear1",), ("linear2",), ("norm2",))


def layer_params(layer):
    """The 16 parameters of an encoder layer in EncoderFn's order.  The owning modules are looked up once per layer object and
    re-validated per call against the module tree (identity of every child along the path: a swapped submodule rebuilds the
    list); then 16 dict reads — a re-assigned parameter is picked up, and no ``Module.__getattr__`` chain runs (it was ~200 of
    them per step)."""
    owners = layer.__dict__.get("_mpf_param_owners")
    if owners is not None:
        for path, m in owners:
            cur = layer
            for name in path:
                cur = cur._modules.get(name)
                if cur is None:
                    break
            if cur is not m:
                owners = None
                break
    if owners is None:
        owners = []
        for path in _PARAM_SLOTS:
            m = layer
            for name in path:
                m = getattr(m, name)
            owners.append((path, m))
        layer.__dict__["_mpf_param_owners"] = owners
    out = []
    for _, m in owners:
        d = m._parameters
        out.append(d["weight"])
        out.append(d["bias"])
    return out


def rows_per_split(sizes):
    """Rows per split of the weight-gradient GEMMs: 512 measured best at config B; a divisor of every
    level size when one exists (then no split straddles a level and the per-level column sums of the
    288-wide gradient fall out of the per-split sums), else 512 and ``aligned`` is False."""
    g = 0
    for s_ in sizes:
        g = math.gcd(g, s_)
    for r in (512, 256, 128, 1024, 64, 32):
        if g % r == 0:
            return r, True
    return 512, False


_slots = {}


def _balanced_rps(R, M, N, device, base=1 << 30):
    """Rows per split of a weight-gradient GEMM such that its (tile, split) workgroups fill ONE whole round of the chip's
    2 x CU workgroup slots (more rounds only if the output alone has more tiles than slots): with round 1's 512-row
    splits a 256 x 256 gradient at R = 43 008 was 336 workgroups (0.66 of a round) and a 256 x 1024 one 1 344 (2.6
    rounds, run as 3) with 84 partial results to sum; now 492 / 512 workgroups and 123 / 32 partials.  Same tiles, same
    products — only the split boundaries move."""
    slots = _slots.get(device)
    if slots is None:
        slots = 2 * torch.cuda.get_device_properties(device).multi_processor_count
        _slots[device] = slots
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    rounds = max(1, -(-tiles * max(1, R // base) // slots))          # rounds that `base`-row splits would take
    ns = max(1, rounds * slots // tiles)
    rps = max(128, ((-(-R // ns)) + 31) // 32 * 32)
    return rps


def _wgrad_group(pairs, amax_pairs=None):
    """[(dW_i, db_i)] of several Linear layers over the same rows: ONE split-K launch whose splits are as long as the tiles
    of all problems together allow at one round of workgroups (R = 43 008: 40 tiles x 12 splits of 3 584 rows — 19 of 2 272 in
    the fp16 x 2 form — instead of 4 x (4 | 16 tiles x 123 | 31 splits of 352 | 1 376 rows)) and ONE reduction."""
    R = pairs[0][0].shape[0]
    dev = pairs[0][0].device
    slots = _slots.get(dev)
    if slots is None:
        slots = _slots[dev] = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    tiles = sum(((g.shape[1] + 127) // 128) * ((x.shape[1] + 127) // 128) for g, x in pairs)
    if amax_pairs is not None and all(g.shape[1] % 256 == 0 and x.shape[1] % 256 == 0 for g, x in pairs):
        # 256 x 256 tiles (csrc/gemm3_nt2.h): one 8-wave workgroup per CU, half the operand traffic of the 128 x 128 tiles
        tiles = sum((g.shape[1] // 256) * (x.shape[1] // 256) for g, x in pairs)
        slots = slots // 2
    elif amax_pairs is not None:
        slots = slots // 2 * 3        # the fp16 x 2 kernel (159 registers, 32 KB of LDS) is resident three times per CU: measured
    ns = max(1, slots // tiles)       # 1.81 -> 1.45 ms/step against splits sized for two (and 1.86 for four)
    rps = max(128, ((-(-R // ns)) + 31) // 32 * 32)
    return gemm3_nt_grouped(pairs, rps, amax_pairs)


# ---- native forward: one call for all layers (csrc/encoder_layer.hip) -------------------------------------------------------
_ENC_FIELDS = ("pv", "pv_am", "po", "po_am", "p1", "p1_am", "p2", "p2_am", "p288", "p288_am", "bv", "bo", "bb1", "bb2", "b288", "g1", "b1",
               "g2", "b2", "value", "raw", "loc", "attn", "ao", "s1", "mean1", "rstd1", "x1", "h", "hbits", "s2", "mean2", "rstd2", "x2", "qn",
               "ao_am", "x1_am", "h_am", "xn_am", "qn_am")
_ENC_IDX = {k: i for i, k in enumerate(_ENC_FIELDS)}
_NATIVE_FWD = os.environ.get("MPF_ENCODER_NATIVE", "1") != "0"        # (0: the python-sequenced forward / backward the tests compare with)
_NATIVE_BWD = _NATIVE_FWD


class MpfEncoderCall(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("S", ctypes.c_int32), ("M", ctypes.c_int32), ("L", ctypes.c_int32), ("P", ctypes.c_int32),
                ("nl", ctypes.c_int32), ("F", ctypes.c_int32), ("reserved", ctypes.c_int32), ("eps", ctypes.c_float), ("pad_", ctypes.c_float),
                ("host_shapes", ctypes.c_void_p), ("shapes_dev", ctypes.c_void_p), ("lsi_dev", ctypes.c_void_p), ("ref", ctypes.c_void_p),
                ("pos_full", ctypes.c_void_p), ("pos_am", ctypes.c_void_p), ("x0", ctypes.c_void_p), ("x0_am", ctypes.c_void_p),
                ("q0", ctypes.c_void_p), ("q0_am", ctypes.c_void_p), ("layers", ctypes.c_void_p)]


_arena_layouts = {}


def _arena_layout(R, C, F, no, M, L, P):
    """float offsets of a layer's results inside its slice of the arena (every tensor 16-byte aligned); -> ({name: (offset, numel)}, floats per layer)"""
    key = (R, C, F, no, M, L, P)
    hit = _arena_layouts.get(key)
    if hit is None:
        sizes = (("value", R * C), ("raw", R * no), ("loc", R * M * L * P * 2), ("attn", R * M * L * P), ("ao", R * C), ("s1", R * C),
                 ("mean1", R), ("rstd1", R), ("x1", R * C), ("h", R * F), ("hbits", R * F // 32), ("s2", R * C), ("mean2", R), ("rstd2", R),
                 ("x2", R * C), ("qn", R * C))
        offs, tot = {}, 0
        for name, n in sizes:
            offs[name] = (tot, n)
            tot += (n + 3) // 4 * 4
        hit = _arena_layouts[key] = (offs, tot)
    return hit


def _native_forward(x, q, x_am, q_am, pos_full, pos_am, params, planes2, b288_all, am, meta, nl, dims, F):
    """The forward of all layers as ONE native call; returns (arena [nl, floats per layer], layout)."""
    N, S, C = dims
    R = N * S
    M, L, P = meta["n_heads"], meta["n_levels"], meta["n_points"]
    no = M * L * P * 3
    lib = _lib.lib()
    if lib.mpf_encoder_fields() != len(_ENC_FIELDS):
        raise RuntimeError("MpfEncoderCall field table differs between encoder_fused.py and libmpformer_hip.so")
    offs, per = _arena_layout(R, C, F, no, M, L, P)
    arena = torch.empty((nl, per), dtype=torch.float32, device=x.device)
    base = arena.data_ptr()
    tab = np.zeros((nl, len(_ENC_FIELDS)), dtype=np.uint64)
    lay = np.asarray([offs[k][0] for k in ("value", "raw", "loc", "attn", "ao", "s1", "mean1", "rstd1", "x1", "h", "hbits", "s2", "mean2",
                                           "rstd2", "x2", "qn")], dtype=np.uint64) * 4
    c0 = _ENC_IDX["value"]
    am0 = am.data_ptr() if isinstance(am, torch.Tensor) else None
    for i in range(nl):
        (wso, bso, waw, baw, wv, bv, wo, bo, g1, b1, w1, bb1, w2, bb2, g2_, b2) = params[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER]
        (pv, pv_am), (po, po_am), (p1, p1_am), (p2, p2_am), (p288, p288_am) = planes2[10 * i:10 * i + 5]
        row = tab[i]
        row[0:10] = (pv.data_ptr(), pv_am.data_ptr(), po.data_ptr(), po_am.data_ptr(), p1.data_ptr(), p1_am.data_ptr(), p2.data_ptr(),
                     p2_am.data_ptr(), p288.data_ptr(), p288_am.data_ptr())
        row[10:19] = (bv.data_ptr(), bo.data_ptr(), bb1.data_ptr(), bb2.data_ptr(), b288_all[i].data_ptr(), g1.data_ptr(), b1.data_ptr(),
                      g2_.data_ptr(), b2.data_ptr())
        row[c0:c0 + 16] = np.uint64(base + i * per * 4) + lay
        row[_ENC_IDX["ao_am"]:_ENC_IDX["ao_am"] + 5] = [am[5 * i + k].data_ptr() for k in range(5)]
    host_shapes = meta["shapes"]._mpf_host
    call = MpfEncoderCall(N, S, M, L, P, nl, F, 0, _EPS, 0.0, host_shapes.data_ptr(), meta["shapes"].data_ptr(), meta["lsi"].data_ptr(),
                          meta["ref"].data_ptr(), pos_full.data_ptr(), pos_am.data_ptr(), x.data_ptr(), x_am.data_ptr(), q.data_ptr(),
                          q_am.data_ptr(), tab.ctypes.data)
    with _lib.device_guard(x.device):
        code = lib.mpf_encoder_forward(ctypes.byref(call), _lib.stream_ptr(x.device))
    _lib.check(code, "mpf_encoder_forward")
    del am0
    return arena, offs


def _arena_views(arena, offs, i, R, C, F, no, dims, M, L, P):
    """the saved tensors of layer i as views of its arena slice"""
    N, S, _ = dims
    a = arena[i]

    def v(name, *shape):
        o, n = offs[name]
        return a[o:o + n].view(*shape)
    hb = a[offs["hbits"][0]:offs["hbits"][0] + offs["hbits"][1]].view(torch.uint8).view(R, F // 8)
    return dict(value=v("value", R, C), loc=v("loc", N, S, M, L, P, 2), attn=v("attn", N, S, M, L, P), ao=v("ao", R, C), s1=v("s1", R, C),
                mean1=v("mean1", R), rstd1=v("rstd1", R), x1=v("x1", R, C), h=v("h", R, F), hbits=hb, s2=v("s2", R, C), mean2=v("mean2", R),
                rstd2=v("rstd2", R), x2=v("x2", R, C), qn=v("qn", R, C))


_ENCB_FIELDS = ("x", "q", "value", "loc", "attn", "ao", "s1", "mean1", "rstd1", "x1", "h", "hbits", "s2", "mean2", "rstd2",
                "x_am", "q_am", "ao_am", "x1_am", "h_am", "g1", "g2", "tv", "tv_am", "to", "to_am", "t1", "t1_am", "t2", "t2_am", "t288", "t288_am",
                "ds2_am", "dh_am", "ds1_am", "draw_am", "gv_am", "wgrad", "dw288", "lvl", "db288")


class MpfEncoderBwdCall(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("S", ctypes.c_int32), ("M", ctypes.c_int32), ("L", ctypes.c_int32), ("P", ctypes.c_int32),
                ("nl", ctypes.c_int32), ("F", ctypes.c_int32), ("rps288", ctypes.c_int32), ("rps_group", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("group_stride", ctypes.c_int64), ("host_shapes", ctypes.c_void_p), ("gout", ctypes.c_void_p),
                ("split_level", ctypes.c_void_p), ("ds2", ctypes.c_void_p), ("dh", ctypes.c_void_p), ("dx1", ctypes.c_void_p),
                ("ds1", ctypes.c_void_p), ("dao", ctypes.c_void_p), ("gv", ctypes.c_void_p), ("draw", ctypes.c_void_p),
                ("dq", ctypes.c_void_p * 2), ("g", ctypes.c_void_p * 2), ("cpart288", ctypes.c_void_p), ("cs288", ctypes.c_void_p),
                ("part_group", ctypes.c_void_p), ("ln_parts", ctypes.c_void_p), ("ln_stride", ctypes.c_uint64), ("msda_ws", ctypes.c_void_p),
                ("msda_ws_bytes", ctypes.c_uint64), ("dgb_out", ctypes.c_void_p), ("layers", ctypes.c_void_p)]


def _group_rps(R, C, F, dev):
    """rows per split of the four plain weight gradients of a layer: what _wgrad_group picks for them (256 x 256 tiles)"""
    slots = _slots.get(dev)
    if slots is None:
        slots = _slots[dev] = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    tiles = 2 * (F // 256) + 2 * (C // 256) * (C // 256)
    ns = max(1, (slots // 2) // tiles)
    return max(128, ((-(-R // ns)) + 31) // 32 * 32)


def _native_backward(gout, params, saved, planes_t, meta, nl, dims, F, rps, split_level):
    """The backward of all layers as ONE native call (csrc/encoder_layer.hip, mpf_encoder_backward): same launches, same
    arguments as the python-sequenced loop below.  -> (d input [R, C], out arena [nl, per], (o_dw288, o_lvl, o_db288, group stride),
    dgb [2 nl, 2, 256])"""
    from .msda import _workspace
    N, S, C = dims
    R = N * S
    M, L, P = meta["n_heads"], meta["n_levels"], meta["n_points"]
    no3 = M * L * P * 3
    dev = gout.device
    lib = _lib.lib()
    if lib.mpf_encoder_bwd_fields() != len(_ENCB_FIELDS):
        raise RuntimeError("MpfEncoderBwdCall field table differs between encoder_fused.py and libmpformer_hip.so")
    rpg = _group_rps(R, C, F, dev)
    ns288, nsg = (R + rps - 1) // rps, (R + rpg - 1) // rpg
    gstride = 2 * C * F + C + F + 2 * (C * C + C)
    # results of every layer: [four weight gradients + biases | dW288^T | per-level column sums | d bias288]
    o_dw288 = gstride
    o_lvl = o_dw288 + no3 * C
    o_db = o_lvl + L * no3
    per = o_db + no3
    out = torch.empty((nl, per), dtype=torch.float32, device=dev)
    # temporaries, shared by all layers
    sizes = [R * C] * 5 + [R * F, R * no3] + [R * C] * 4 + [ns288 * no3 * C, ns288 * no3, nsg * gstride]
    tmp = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    ptrs, o = [], tmp.data_ptr()
    for n_ in sizes:
        ptrs.append(o)
        o += 4 * n_
    ds2, dx1, ds1, dao, gv, dh, draw, dq0, dq1, g0, g1_, cpart, cs, pgroup = ptrs
    ln_stride = (int(lib.mpf_res_ln256_backward_workspace_bytes(R)) + 255) & ~255
    ln_parts = torch.empty(2 * nl * ln_stride, dtype=torch.uint8, device=dev)
    host_shapes = meta["shapes"]._mpf_host
    need = lib.mpf_msda_backward_workspace_bytes(N, M, L, S, P, host_shapes.data_ptr())
    if need == 0:
        raise RuntimeError("mpf_msda_backward_workspace_bytes rejected the level geometry")
    ws = _workspace(dev, need)
    dgb = torch.empty((2 * nl, 2, 256), dtype=torch.float32, device=dev)
    am = amax_slots(5 * nl, dev)                     # per layer: ds2, dh, ds1, draw, gv
    tab = np.zeros((nl, len(_ENCB_FIELDS)), dtype=np.uint64)
    for i in range(nl):
        g1 = params[i * PARAMS_PER_LAYER + 8]
        g2 = params[i * PARAMS_PER_LAYER + 14]
        (x, q, value, loc, attn, ao, s1, mean1, rstd1, x1, h, s2, mean2, rstd2, x_am, ao_am, x1_am, h_am, q_am, hbits) = saved[i * 20:(i + 1) * 20]
        (tv, tv_am), (to, to_am), (t1, t1_am), (t2, t2_am), (t288, t288_am) = planes_t[i]
        ob = out.data_ptr() + 4 * i * per
        tab[i] = [t_.data_ptr() for t_ in (x, q, value, loc, attn, ao, s1, mean1, rstd1, x1, h, hbits, s2, mean2, rstd2, x_am, q_am, ao_am,
                                           x1_am, h_am, g1, g2, tv, tv_am, to, to_am, t1, t1_am, t2, t2_am, t288, t288_am)] + \
                 [am[5 * i + k].data_ptr() for k in range(5)] + [ob, ob + 4 * o_dw288, ob + 4 * o_lvl, ob + 4 * o_db]
    call = MpfEncoderBwdCall(N, S, M, L, P, nl, F, rps, rpg, 0, gstride, host_shapes.data_ptr(), gout.data_ptr(), split_level.data_ptr(),
                             ds2, dh, dx1, ds1, dao, gv, draw, (ctypes.c_void_p * 2)(dq0, dq1), (ctypes.c_void_p * 2)(g0, g1_), cpart, cs, pgroup,
                             ln_parts.data_ptr(), ln_stride, ws.data_ptr(), ws.numel(), dgb.data_ptr(), tab.ctypes.data)
    with _lib.device_guard(dev):
        code = lib.mpf_encoder_backward(ctypes.byref(call), _lib.stream_ptr(dev))
    _lib.check(code, "mpf_encoder_backward")
    o0 = (g0 - tmp.data_ptr()) // 4
    g = tmp[o0:o0 + R * C].view(R, C)              # g[0]: what layer 0 hands down
    return g, out, (o_dw288, o_lvl, o_db, gstride), dgb


class EncoderFn(Function):
    @staticmethod
    def forward(ctx, src, pos_const, level_embed, meta, *params):
        # src [N, S, C] fp32; pos_const [S, C] (sine embedding, constant); level_embed [L, C]
        N, S, C = src.shape
        R = N * S
        nl = len(params) // PARAMS_PER_LAYER
        M, L, P = meta["n_heads"], meta["n_levels"], meta["n_points"]
        shapes, lsi, ref, normalizer, level_idx = meta["shapes"], meta["lsi"], meta["ref"], meta["normalizer"], meta["level_idx"]
        pos_full = pos_const + level_embed.index_select(0, level_idx)           # [S, C]
        host_shapes = getattr(shapes, "_mpf_host", None)
        x = src.reshape(R, C)
        saved = []
        no = M * L * P * 2
        q = None
        # fp16 x 2 planes (+ largest magnitude) of every weight, both orientations (W for the forward, W^T for the input
        # gradients; sampling_offsets | attention_weights stacked into one 288-row operand on the way): two launches for all layers
        g2 = []
        for i in range(nl):
            (wso, _, waw, _, wv, _, wo, _, _, _, w1, _, w2, _, _, _) = params[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER]
            ws = ([wv], [wo], [w1], [w2], [wso, waw])
            g2 += [(w_, False) for w_ in ws] + [(w_, True) for w_ in ws]
        planes2 = split_weights_grouped_h2(g2)
        # the 288-wide bias of every layer in one concatenation
        b288_all = torch.cat([params[i * PARAMS_PER_LAYER + k] for i in range(nl) for k in (1, 3)]).view(nl, -1)
        # amax slots (largest magnitude, or an upper bound of it) of every GEMM operand: x / q of layer 0 and the positional
        # term by a pass over the data; the rest from their producers — LayerNorm outputs bounded from (gamma, beta), the
        # attention output bounded by max |value| (a convex combination of value rows), the FFN hidden layer from its GEMM
        am = amax_slots(5 * nl + 3, src.device)          # per layer: ao, x1, h, the next layer's x, the next layer's q
        x_am, q_am, pos_am = amax(x, am[5 * nl]), am[5 * nl + 1], amax(pos_full, am[5 * nl + 2])
        F_ = params[10].shape[0]                          # ffn width (linear1.weight [F, C])
        ctx.guarded = guarded = _guard_begin()
        use_native = (_NATIVE_FWD and RANGE_AUDIT is None and not guarded and host_shapes is not None and C == 256 and M == 8 and F_ % 128 == 0
                      and all(params[i * PARAMS_PER_LAYER + 10].shape[0] == F_ for i in range(nl)))
        if use_native:
            q = (x.view(N, S, C) + pos_full).view(R, C)
            amax(q, q_am)
            arena, offs = _native_forward(x, q, x_am, q_am, pos_full, pos_am, params, planes2, b288_all, am, meta, nl, (N, S, C), F_)
            for i in range(nl):
                t_ = _arena_views(arena, offs, i, R, C, F_, no * 3 // 2, (N, S, C), M, L, P)
                ao_am, x1_am, h_am, xn_am, qn_am = am[5 * i:5 * i + 5]
                saved += [x, q, t_["value"], t_["loc"], t_["attn"], t_["ao"], t_["s1"], t_["mean1"], t_["rstd1"], t_["x1"], t_["h"],
                          t_["s2"], t_["mean2"], t_["rstd2"], x_am, ao_am, x1_am, h_am, q_am, t_["hbits"]]
                x, q, x_am, q_am = t_["x2"], t_["qn"], xn_am, qn_am
            ctx.save_for_backward(pos_full, level_embed, *params, *saved)
            ctx.meta, ctx.nl, ctx.dims = meta, nl, (N, S, C)
            ctx.planes_t = [planes2[10 * i + 5:10 * i + 10] for i in range(nl)]
            # the last layer's x2 is a view of the arena every layer's saved tensors live in.  When nothing will back-propagate
            # (eval / no_grad: autograd drops the saved tensors) a view would keep those ~0.6 GB per layer alive for as long as
            # the caller holds `memory` — hand out a copy then
            return x.view(N, S, C) if any(ctx.needs_input_grad) else x.view(N, S, C).clone()
        _guard["active"] = guarded                   # (the operands of this call go through mpf_h2_range_stats in _audit)
        for i in range(nl):
            (wso, bso, waw, baw, wv, bv, wo, bo, g1, b1, w1, bb1, w2, bb2, g2_, b2) = params[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER]
            (pv, pv_am), (po, po_am), (p1, p1_am), (p2, p2_am), (p288, p288_am) = planes2[10 * i:10 * i + 5]
            b288 = b288_all[i]
            ao_am, x1_am, h_am, xn_am, qn_am = am[5 * i:5 * i + 5]
            _audit("fwd x (value_proj)", x, x_am)
            value = gemm3_h2(x, x_am, pv, pv_am, bv, out_amax=ao_am)
            if q is None:       # layer 0; later layers get src + pos from the previous layer's norm2 pass
                q = (x.view(N, S, C) + pos_full).view(R, C)
                amax(q, q_am)
            _audit("fwd q (offsets | weights)", q, q_am)
            raw = gemm3_h2(q, q_am, p288, p288_am, b288)
            # softmax over the 12 logits, loc = ref + offset / (W_l, H_l) happen inside the MSDA kernel
            ao, loc, attn = ms_deform_attn_forward_raw(value.view(N, S, M, C // M), shapes, lsi, raw, ref, host_shapes)
            ao = ao.view(R, C)
            _audit("fwd attention output (output_proj)", ao, ao_am)
            s1 = gemm3_h2(ao, ao_am, po, po_am, bo, cin=x)
            x1, mean1, rstd1, _ = ln256_forward(s1, g1, b1, _EPS, y_bound=x1_am)
            # (the ReLU's gate leaves the product as a bit mask: the backward then reads 1 bit instead of 4 bytes per element of h)
            if p1.shape[1] % 128 == 0:
                h, hbits = gemm3_h2_bits(x1, x1_am, p1, p1_am, bb1, relu=True, out_amax=h_am, want_bits=True)
            else:
                h, hbits = gemm3_h2(x1, x1_am, p1, p1_am, bb1, relu=True, out_amax=h_am), None
            _audit("fwd x1 (linear1)", x1, x1_am)
            _audit("fwd hidden (linear2)", h, h_am)
            s2 = gemm3_h2(h, h_am, p2, p2_am, bb2, cin=x1)
            last = i + 1 == nl
            x2, mean2, rstd2, qn = ln256_forward(s2, g2_, b2, _EPS, padd=None if last else pos_full, y_bound=xn_am,
                                                 padd_amax=None if last else pos_am, yplus_bound=None if last else qn_am)
            saved += [x, q, value, loc, attn, ao, s1, mean1, rstd1, x1, h, s2, mean2, rstd2, x_am, ao_am, x1_am, h_am, q_am, hbits]
            x, q, x_am, q_am = x2, qn, xn_am, qn_am
        _guard["active"] = False
        ctx.save_for_backward(pos_full, level_embed, *params, *saved)
        ctx.meta, ctx.nl, ctx.dims = meta, nl, (N, S, C)
        ctx.planes_t = [planes2[10 * i + 5:10 * i + 10] for i in range(nl)]       # (W^T planes, amax) for the backward
        return x.view(N, S, C)

    @staticmethod
    def backward(ctx, gout):
        N, S, C = ctx.dims
        R = N * S
        meta, nl = ctx.meta, ctx.nl
        M, L, P = meta["n_heads"], meta["n_levels"], meta["n_points"]
        shapes, lsi, normalizer, sizes = meta["shapes"], meta["lsi"], meta["normalizer"], meta["sizes"]
        host_shapes = getattr(shapes, "_mpf_host", None)
        t = ctx.saved_tensors
        pos_full, level_embed = t[0], t[1]
        params = t[2:2 + nl * PARAMS_PER_LAYER]
        saved = t[2 + nl * PARAMS_PER_LAYER:]
        g = gout.reshape(R, C).contiguous()
        no = M * L * P * 2
        rps, aligned = rows_per_split(sizes)
        split_level = None
        if aligned:
            key = ("split_level", rps, N)
            split_level = meta.get(key)
            if split_level is None:
                split_level = meta["level_idx"][::rps].repeat(N)          # level of every split's rows
                meta[key] = split_level
        dparams = [None] * (nl * PARAMS_PER_LAYER)
        F_ = params[10].shape[0]
        guarded = bool(getattr(ctx, "guarded", False))
        if (_NATIVE_BWD and RANGE_AUDIT is None and not guarded and host_shapes is not None and aligned and L <= 4 and C == 256 and M == 8 and F_ % 256 == 0
                and all(saved[i * 20 + 19] is not None and params[i * PARAMS_PER_LAYER + 10].shape[0] == F_ for i in range(nl))):
            g, out, (o_dw, o_lvl, o_db, gs), dgb = _native_backward(g, params, saved, ctx.planes_t, meta, nl, (N, S, C), F_, rps, split_level)
            no3 = M * L * P * 3
            for i in range(nl):
                o_, dp = out[i], [None] * PARAMS_PER_LAYER
                dw288, db288 = o_[o_dw:o_dw + no3 * C].view(no3, C), o_[o_db:o_db + no3]
                dp[0], dp[1], dp[2], dp[3] = dw288[:no], db288[:no], dw288[no:], db288[no:]
                k = 0
                for (wi, bi, m_, n_) in ((12, 13, C, F_), (10, 11, F_, C), (6, 7, C, C), (4, 5, C, C)):
                    dp[wi], dp[bi] = o_[k:k + m_ * n_].view(m_, n_), o_[k + m_ * n_:k + m_ * n_ + m_]
                    k += m_ * n_ + m_
                dparams[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER] = dp
            lvls_all = out[:, o_lvl:o_lvl + L * no3].view(nl, L, no3)
            return EncoderFn._finish(g, dgb, lvls_all, dparams, params, nl, (N, S, C))
        am = amax_slots(5 * nl, g.device)               # per layer: ds2, dh, ds1, draw, gv
        lvls = [None] * nl
        gq = None
        lng = LnGradGroup(2 * nl, R, g.device)           # the 2 nl LayerNorm parameter gradients: one reduce launch at the end
        _guard["active"] = guarded
        for i in reversed(range(nl)):
            (wso, bso, waw, baw, wv, bv, wo, bo, g1, b1, w1, bb1, w2, bb2, g2, b2) = params[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER]
            (x, q, value, loc, attn, ao, s1, mean1, rstd1, x1, h, s2, mean2, rstd2, x_am, ao_am, x1_am, h_am, q_am, hbits) = saved[i * 20:(i + 1) * 20]
            dp = [None] * PARAMS_PER_LAYER
            (tv, tv_am), (to, to_am), (t1, t1_am), (t2, t2_am), (t288, t288_am) = ctx.planes_t[i]
            ds2_am, dh_am, ds1_am, draw_am, gv_am = am[5 * i:5 * i + 5]
            # norm2 <- ffn
            ds2 = lng.backward(s2, mean2, rstd2, g2, g, gq, ds_amax=ds2_am)
            if hbits is not None:
                dh = gemm3_h2_bits(ds2, ds2_am, t2, t2_am, gate_bits=hbits, out_amax=dh_am)
            else:
                dh = gemm3_h2(ds2, ds2_am, t2, t2_am, gate=h, out_amax=dh_am)
            _audit("bwd ds2 (d linear2 output)", ds2, ds2_am)
            _audit("bwd dh (d hidden)", dh, dh_am)
            dx1 = gemm3_h2(dh, dh_am, t1, t1_am, cin=ds2)
            # norm1 <- attention
            ds1 = lng.backward(s1, mean1, rstd1, g1, dx1, ds_amax=ds1_am)
            _audit("bwd ds1 (d output_proj output)", ds1, ds1_am)
            dao = gemm3_h2(ds1, ds1_am, to, to_am)
            # d(raw): the softmax / offset-normaliser backward is the epilogue of the push kernel
            # (the bin + tile kernels record the largest magnitudes of gv / draw themselves: no amax pass over the two tensors)
            gv, draw = ms_deform_attn_backward_raw(value.view(N, S, M, C // M), host_shapes, loc, attn, dao.view(N, S, C), ao.view(N, S, C),
                                                   draw_am, gv_am)
            _audit("bwd draw (d offsets | weights)", draw, draw_am)
            _audit("bwd grad_value (d value_proj output)", gv.view(R, C), gv_am)
            dq = gemm3_h2(draw, draw_am, t288, t288_am)
            # dW288^T = q^T . draw (288 on the 96-wide tile side) + per-split column sums of draw: the
            # bias gradient and, summed per level, the level_embed gradient
            cpart, _, cs = gemm3_nt(q, draw, rps, want_csum_b=True, transpose_out=True, amax_ab=(q_am, draw_am))
            if aligned and L <= 4 and cpart[0].numel() % 4 == 0:
                dw288, lvl, db288 = nt_reduce_levels(cpart, cs, split_level, L)     # one launch, fixed order
            else:
                dw288 = cpart.sum(0)
                if aligned:
                    lvl = torch.zeros((L, draw.shape[1]), dtype=torch.float32, device=g.device).index_add_(0, split_level, cs)
                else:
                    lvl = torch.stack([sl.sum((0, 1)) for sl in draw.view(N, S, -1).split(sizes, 1)])   # [L, 288]
                db288 = lvl.sum(0)
            lvls[i] = lvl
            dp[0], dp[1], dp[2], dp[3] = dw288[:no], db288[:no], dw288[no:], db288[no:]
            gv2 = gv.view(R, C)
            # grad wrt this layer's input: through value_proj + the residual; the (src + pos) path (dq)
            # joins inside the previous layer's norm2 backward (layer 0: added here)
            g = gemm3_h2(gv2, gv_am, tv, tv_am, cin=ds1, cin2=dq if i == 0 else None)
            gq = dq
            # the four plain weight gradients of the layer (their operands are all alive here) as one launch
            (dp[12], dp[13]), (dp[10], dp[11]), (dp[6], dp[7]), (dp[4], dp[5]) = _wgrad_group(
                [(ds2, h), (dh, x1), (ds1, ao), (gv2, x)], [(ds2_am, h_am), (dh_am, x1_am), (ds1_am, ao_am), (gv_am, x_am)])
            dparams[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER] = dp
        _guard["active"] = False
        return EncoderFn._finish(g, lng.finish(), torch.stack(lvls), dparams, params, nl, (N, S, C))

    @staticmethod
    def _finish(g, dgb, lvls_all, dparams, params, nl, dims):
        """LayerNorm parameter gradients into their places + d level_embed; dgb [2 nl, 2, 256] in call order: (norm2, norm1) of
        layers nl - 1 .. 0; lvls_all [nl, L, 288]: per-level column sums of every layer's d raw"""
        N, S, C = dims
        for z, i in enumerate(reversed(range(nl))):
            base = i * PARAMS_PER_LAYER
            dparams[base + 14], dparams[base + 15] = dgb[2 * z, 0], dgb[2 * z, 1]
            dparams[base + 8], dparams[base + 9] = dgb[2 * z + 1, 0], dgb[2 * z + 1, 1]
        # d level_embed = sum_i lvl_i . [W_offsets_i ; W_weights_i]: one stacked product for all layers
        w288_all = torch.cat([params[i * PARAMS_PER_LAYER + k] for i in range(nl) for k in (0, 2)]).view(nl, -1, C)
        # [nl, L, 288] x [nl, 288, C] summed over the layers: 1.3 M multiply-adds — as a broadcast product + one reduction
        # (the library's batched fp32 GEMM for it was the head's last hipBLASLt launch)
        d_level = (lvls_all[:, :, :, None] * w288_all[:, None, :, :]).sum((0, 2))
        return (g.view(N, S, C), None, d_level, None, *dparams)
