"""Host binding of the split-bf16 fp32 GEMM (csrc/gemm3.hip; include/mpformer_hip.h mpf_gemm3_*).

``split_weight`` turns an fp32 weight [out, in] into the three bf16 planes the kernel consumes
(``transpose=True``: planes of W^T, the operand of dX = dY . W); ``gemm3`` runs
C = (A + A2) . B^T + bias + Cin + Cin2, optional ReLU / ReLU-backward gate.  GPU only, fp32 only.
"""
import torch

from . import _lib


def _stream(t):
    return _lib.stream_ptr(t.device)


def split_weight(w, transpose=False):
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
    r, c = w.shape
    out = torch.empty((3, c, r) if transpose else (3, r, c), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().mpf_gemm3_split(w.data_ptr(), r, c, 1 if transpose else 0, out.data_ptr(), _stream(w)),
               "mpf_gemm3_split")
    return out


def split_weights_grouped(groups):
    """Every operand of ``groups`` split in ONE launch.  groups: list of (sources, transpose) where sources
    is a list of fp32 [rows_i, K] matrices stacked along dim 0 (one logical weight [sum rows_i, K]);
    returns one planes tensor per group: [3, R, K] (transpose False) or [3, K, R] (the planes of W^T)."""
    import numpy as np
    from ._h2d import upload
    dev = groups[0][0][0].device
    sizes, tot = [], 0
    for srcs, _ in groups:
        n = sum(t.numel() for t in srcs)
        sizes.append((tot, n))
        tot += 3 * n
    buf = torch.empty(tot, dtype=torch.bfloat16, device=dev)
    base = buf.data_ptr()
    rows_tab, blk, outs = [], 0, []
    for (srcs, tr), (off, n) in zip(groups, sizes):
        K = srcs[0].shape[1]
        R = sum(t.shape[0] for t in srcs)
        r0 = 0
        for t in srcs:
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == K
            # rows r0.. of W (transpose False: rows of the output; True: columns of the output)
            dst = base + 2 * (off + (r0 if tr else r0 * K))
            rows_tab.append((t.data_ptr(), dst, t.shape[0], K, 1 if tr else 0, R if tr else K, n, blk))
            blk += (t.numel() + 1023) // 1024
            r0 += t.shape[0]
        outs.append(buf[off:off + 3 * n].view((3, K, R) if tr else (3, R, K)))
    items = upload(np.asarray(rows_tab, dtype=np.int64).reshape(-1), dev)
    with _lib.device_guard(dev):
        code = _lib.lib().mpf_gemm3_split_grouped(items.data_ptr(), len(rows_tab), blk, _stream(buf))
    _lib.check(code, "mpf_gemm3_split_grouped")
    return outs


AMAX_SLOT = 512       # floats per amax slot (include/mpformer_hip.h MPF_AMAX_SLOT_FLOATS): 16 sub-slots, one per cache line


def amax_slots(n, device):
    """n zeroed amax slots [n, AMAX_SLOT]; row i is the slot handed to a producer / consumer."""
    return torch.zeros((n, AMAX_SLOT), dtype=torch.float32, device=device)


def amax_value(slot):
    """the largest magnitude recorded in a slot (0-dim tensor; tests / diagnostics)"""
    return slot.view(-1)[::32][:16].max()


def amax(t, out=None):
    """max |t| of an fp32 tensor into an amax slot (``out``: an already zeroed or partly filled slot); returns the slot."""
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    if out is None:
        out = torch.zeros(AMAX_SLOT, dtype=torch.float32, device=t.device)
    with _lib.device_guard(t.device):
        code = _lib.lib().mpf_amax_f32(t.data_ptr(), t.numel(), out.data_ptr(), _stream(t))
    _lib.check(code, "mpf_amax_f32")
    return out


def split_weights_grouped_h2(groups):
    """``split_weights_grouped`` for the fp16 x 2 form: per group (planes [2, R, K] or [2, K, R] fp16, the operand's amax slot) —
    two launches for all groups (largest magnitudes, then the scaled split); both orientations of a weight share one slot."""
    import numpy as np
    from ._h2d import upload
    dev = groups[0][0][0].device
    sizes, tot = [], 0
    for srcs, _ in groups:
        n = sum(t.numel() for t in srcs)
        sizes.append((tot, n))
        tot += 2 * n
    buf = torch.empty(tot, dtype=torch.float16, device=dev)
    slots = amax_slots(len(groups), dev)
    base, sbase = buf.data_ptr(), slots.data_ptr()
    am_tab, sp_tab, ablk, sblk, outs = [], [], 0, 0, []
    seen = {}
    for gi, ((srcs, tr), (off, n)) in enumerate(zip(groups, sizes)):
        K = srcs[0].shape[1]
        R = sum(t.shape[0] for t in srcs)
        key = tuple(t.data_ptr() for t in srcs)
        slot = sbase + 4 * AMAX_SLOT * gi
        if key in seen:             # the same weight in the other orientation: same amax, no second pass
            slot_src = seen[key]
        else:
            seen[key] = gi
            slot_src = gi
            for t in srcs:
                am_tab.append((t.data_ptr(), slot, t.numel(), ablk))
                ablk += (t.numel() + 4095) // 4096
        slot = sbase + 4 * AMAX_SLOT * slot_src
        r0 = 0
        for t in srcs:
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == K
            dst = base + 2 * (off + (r0 if tr else r0 * K))
            sp_tab.append((t.data_ptr(), dst, slot, t.shape[0], K, 1 if tr else 0, R if tr else K, n, sblk))
            sblk += (t.numel() + 1023) // 1024
            r0 += t.shape[0]
        outs.append((buf[off:off + 2 * n].view((2, K, R) if tr else (2, R, K)), slots[slot_src]))
    both = upload(np.asarray([x for row in am_tab for x in row] + [x for row in sp_tab for x in row], dtype=np.int64), dev)
    with _lib.device_guard(dev):
        code = _lib.lib().mpf_amax_f32_grouped(both.data_ptr(), len(am_tab), ablk, _stream(buf))
        _lib.check(code, "mpf_amax_f32_grouped")
        code = _lib.lib().mpf_gemm3_split_grouped_h2(both.data_ptr() + 8 * 4 * len(am_tab), len(sp_tab), sblk, _stream(buf))
    _lib.check(code, "mpf_gemm3_split_grouped_h2")
    return outs


def gemm3_h2(a, a_amax, planes, w_amax, bias=None, cin=None, cin2=None, gate=None, relu=False, out_amax=None):
    """``gemm3`` in the fp16 x 2 form: planes / w_amax from ``split_weights_grouped_h2``, a_amax = the amax slot of ``a`` (the
    largest |a|, or an upper bound within a few binades of it); out_amax (a zeroed slot or None) receives max |C|."""
    assert a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
    assert planes.dtype == torch.float16 and planes.is_contiguous() and planes.shape[0] == 2
    M, K = a.shape
    N = planes.shape[1]
    assert planes.shape[2] == K
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with _lib.device_guard(a.device):
        code = _lib.lib().mpf_gemm3_tn_h2(
            a.data_ptr(), a.stride(0), a_amax.data_ptr(), planes.data_ptr(), w_amax.data_ptr(), _p(bias),
            _p(cin), _rows(cin, N) if cin is not None else 0, _p(cin2), _rows(cin2, N) if cin2 is not None else 0,
            _p(gate), _rows(gate, N) if gate is not None else 0, c.data_ptr(), c.stride(0), _p(out_amax), M, N, K, 1 if relu else 0,
            _stream(a))
    _lib.check(code, "mpf_gemm3_tn_h2")
    return c


def gemm3_h2_bits(a, a_amax, planes, w_amax, bias=None, cin=None, cin2=None, gate_bits=None, relu=False, out_amax=None,
                  want_bits=False):
    """``gemm3_h2`` with the ReLU-backward gate as a bit mask (``mpf_gemm3_tn_h2_bits``; N % 128 == 0): gate_bits [M, N / 8]
    uint8 — bit n & 7 of byte n // 8 — instead of the saved activation; want_bits: also return the mask of (C > 0), i.e. with
    ``relu`` the gate of this product's own backward.  Returns C, or (C, bits)."""
    assert a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
    assert planes.dtype == torch.float16 and planes.is_contiguous() and planes.shape[0] == 2
    M, K = a.shape
    N = planes.shape[1]
    assert planes.shape[2] == K and N % 128 == 0
    if gate_bits is not None:
        assert gate_bits.dtype == torch.uint8 and gate_bits.shape == (M, N // 8) and gate_bits.is_contiguous()
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    bits = torch.empty((M, N // 8), dtype=torch.uint8, device=a.device) if want_bits else None
    with _lib.device_guard(a.device):
        code = _lib.lib().mpf_gemm3_tn_h2_bits(
            a.data_ptr(), a.stride(0), a_amax.data_ptr(), planes.data_ptr(), w_amax.data_ptr(), _p(bias),
            _p(cin), _rows(cin, N) if cin is not None else 0, _p(cin2), _rows(cin2, N) if cin2 is not None else 0,
            _p(gate_bits), N // 8, c.data_ptr(), c.stride(0), _p(out_amax), _p(bits), N // 8, M, N, K, 1 if relu else 0, _stream(a))
    _lib.check(code, "mpf_gemm3_tn_h2_bits")
    return (c, bits) if want_bits else c


def _p(t):
    return t.data_ptr() if t is not None else None


def _rows(t, n):
    assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[1] == n
    return t.stride(0)


def gemm3(a, planes, bias=None, a2=None, cin=None, cin2=None, gate=None, relu=False, out=None):
    """a [M, K] fp32 (row stride free), planes [3, N, K] bf16 -> [M, N] fp32."""
    assert a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
    M, K = a.shape
    N = planes.shape[1]
    assert planes.shape[2] == K and planes.dtype == torch.bfloat16 and planes.is_contiguous()
    c = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert c.shape == (M, N) and c.stride(1) == 1
    if a2 is not None:
        assert a2.dtype == torch.float32 and a2.dim() == 2 and a2.is_contiguous() and a2.shape[1] == K
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
    with _lib.device_guard(a.device):
        code = _lib.lib().mpf_gemm3_tn(
            a.data_ptr(), a.stride(0), _p(a2), a2.shape[0] if a2 is not None else 0, planes.data_ptr(), _p(bias),
            _p(cin), _rows(cin, N) if cin is not None else 0, _p(cin2), _rows(cin2, N) if cin2 is not None else 0,
            _p(gate), _rows(gate, N) if gate is not None else 0, c.data_ptr(), c.stride(0), M, N, K,
            1 if relu else 0, _stream(a))
    _lib.check(code, "mpf_gemm3_tn")
    return c


def pick_rows_per_split(R, out_tiles, align=None):
    """Rows per split of the weight-gradient GEMM: enough (tile, split) blocks to fill the chip
    (>= ~400), a multiple of 32, and — if ``align`` is given — a divisor of it (so that no split
    straddles a segment boundary, e.g. a feature level)."""
    want = max(1, 400 // max(out_tiles, 1))
    rps = max(32, (R // want) // 32 * 32)
    if align:
        while rps > 32 and align % rps != 0:
            rps -= 32
        if align % rps != 0:
            return None
    return rps


def gemm3_nt(a, b, rows_per_split, b2=None, want_csum_a=False, want_csum_b=False, transpose_out=False, amax_ab=None):
    """a [R, M], b [R, N] fp32 (row strides free) -> (c_part [nsplit, M, N] (or [nsplit, N, M]),
    csum_a [nsplit, M] or None, csum_b [nsplit, N] or None).  amax_ab = (amax slot of a, of b): the fp16 x 2 form (no b2)."""
    assert a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and a.shape[0] == b.shape[0]
    R, M = a.shape
    N = b.shape[1]
    ns = (R + rows_per_split - 1) // rows_per_split
    c = torch.empty((ns, N, M) if transpose_out else (ns, M, N), dtype=torch.float32, device=a.device)
    ca = torch.empty((ns, M), dtype=torch.float32, device=a.device) if want_csum_a else None
    cb = torch.empty((ns, N), dtype=torch.float32, device=a.device) if want_csum_b else None
    if b2 is not None:
        assert b2.dtype == torch.float32 and b2.dim() == 2 and b2.stride(1) == 1 and b2.shape[1] == N
    with _lib.device_guard(a.device):
        if amax_ab is not None:
            assert b2 is None
            code = _lib.lib().mpf_gemm3_nt_h2(
                a.data_ptr(), a.stride(0), amax_ab[0].data_ptr(), b.data_ptr(), b.stride(0), amax_ab[1].data_ptr(), c.data_ptr(),
                _p(ca), _p(cb), R, M, N, rows_per_split, 1 if transpose_out else 0, _stream(a))
        else:
            code = _lib.lib().mpf_gemm3_nt(
                a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), _p(b2), b2.stride(0) if b2 is not None else 0,
                b2.shape[0] if b2 is not None else 0, c.data_ptr(), _p(ca), _p(cb), R, M, N, rows_per_split,
                1 if transpose_out else 0, _stream(a))
    _lib.check(code, "mpf_gemm3_nt")
    return c, ca, cb


def gemm3_nt_grouped(pairs, rows_per_split, amax_pairs=None):
    """The weight gradients of several Linear layers over the same rows in ONE split-K launch + ONE reduction.
    pairs: [(g_i [R, M_i], x_i [R, N_i])] (fp32, row strides free, at most 8); returns [(dW_i [M_i, N_i], db_i [M_i])] with
    dW_i = g_i^T . x_i and db_i = colsum(g_i) — views of one buffer.  Same partial sums as ``gemm3_nt`` with the same
    ``rows_per_split`` (bit-identical).  amax_pairs: [(amax slot of g_i, of x_i)] selects the fp16 x 2 form."""
    import numpy as np
    R = pairs[0][0].shape[0]
    dev = pairs[0][0].device
    ns = (R + rows_per_split - 1) // rows_per_split
    offs, tot = [], 0
    for g, x in pairs:
        assert g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and g.dim() == 2 and x.dim() == 2
        assert g.stride(1) == 1 and x.stride(1) == 1 and g.shape[0] == R and x.shape[0] == R and x.shape[1] % 4 == 0
        offs.append(tot)
        # every item starts on a 16-byte boundary (float4 epilogue stores / float4 reduction); M_i % 4 != 0 leaves pad floats
        tot += (g.shape[1] * x.shape[1] + g.shape[1] + 3) // 4 * 4
    padded = any((g.shape[1] * x.shape[1] + g.shape[1]) % 4 for g, x in pairs)
    # pad floats are summed by nt_reduce and never read back: zero them so that the reduction reads no uninitialised memory
    part = (torch.zeros if padded else torch.empty)((ns, tot), dtype=torch.float32, device=dev)
    base = part.data_ptr()
    items = np.empty((len(pairs), 8 if amax_pairs is None else 10), dtype=np.int64)
    for i, ((g, x), o) in enumerate(zip(pairs, offs)):
        M, N = g.shape[1], x.shape[1]
        if amax_pairs is None:
            items[i] = (g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), base + 4 * o, base + 4 * (o + M * N), M, N)
        else:
            items[i] = (g.data_ptr(), g.stride(0), amax_pairs[i][0].data_ptr(), x.data_ptr(), x.stride(0), amax_pairs[i][1].data_ptr(),
                        base + 4 * o, base + 4 * (o + M * N), M, N)
    with _lib.device_guard(dev):
        fn = _lib.lib().mpf_gemm3_nt_grouped if amax_pairs is None else _lib.lib().mpf_gemm3_nt_grouped_h2
        code = fn(items.ctypes.data, len(pairs), R, rows_per_split, tot, _stream(part))
    _lib.check(code, "mpf_gemm3_nt_grouped")
    out, _ = nt_reduce(part)
    res = []
    for (g, x), o in zip(pairs, offs):
        M, N = g.shape[1], x.shape[1]
        res.append((out[o:o + M * N].view(M, N), out[o + M * N:o + M * N + M]))
    return res


def nt_reduce(c_part, s_part=None):
    """(sum over splits of c_part [ns, ...], of s_part [ns, n] or None) in ONE launch, fixed order."""
    ns = c_part.shape[0]
    c = torch.empty(c_part.shape[1:], dtype=torch.float32, device=c_part.device)
    s = torch.empty(s_part.shape[1:], dtype=torch.float32, device=c_part.device) if s_part is not None else None
    if c.numel() % 4 or (s is not None and s.numel() % 4):
        return c_part.sum(0), (s_part.sum(0) if s_part is not None else None)
    with _lib.device_guard(c_part.device):
        code = _lib.lib().mpf_gemm3_nt_reduce(c_part.data_ptr(), c.numel(), _p(s_part), s.numel() if s is not None else 0, ns,
                                              c.data_ptr(), _p(s), _stream(c_part))
    _lib.check(code, "mpf_gemm3_nt_reduce")
    return c, s


def nt_reduce_levels(c_part, s_part, level_of_split, n_levels):
    """(sum over splits of c_part, per-level sums of s_part [n_levels, n], their total [n]) in ONE launch, fixed order;
    level_of_split: int64 [ns] on the device."""
    ns = c_part.shape[0]
    c = torch.empty(c_part.shape[1:], dtype=torch.float32, device=c_part.device)
    n = s_part.shape[1]
    lvl = torch.empty((n_levels, n), dtype=torch.float32, device=c_part.device)
    s = torch.empty((n,), dtype=torch.float32, device=c_part.device)
    assert level_of_split.dtype == torch.int64 and level_of_split.numel() == ns and c.numel() % 4 == 0 and n % 4 == 0 and n_levels <= 4
    with _lib.device_guard(c_part.device):
        code = _lib.lib().mpf_gemm3_nt_reduce_levels(c_part.data_ptr(), c.numel(), s_part.data_ptr(), n, ns, level_of_split.data_ptr(),
                                                     n_levels, c.data_ptr(), lvl.data_ptr(), s.data_ptr(), _stream(c_part))
    _lib.check(code, "mpf_gemm3_nt_reduce_levels")
    return c, lvl, s


def gemm3_ex(a, planes, bias=None, cin=None, relu=False, out_dtype=torch.float32):
    """gemm3 with a bf16 or fp32 A [M, K] (row stride free, rows 16-byte aligned) and a bf16 or fp32 result: a bf16
    activation enters the fp32 GEMM as its own first plane (no cast pass, three products instead of six)."""
    assert a.is_cuda and a.dim() == 2 and a.stride(1) == 1 and a.dtype in (torch.float32, torch.bfloat16)
    M, K = a.shape
    N = planes.shape[1]
    assert planes.shape[2] == K and planes.dtype == torch.bfloat16 and planes.is_contiguous()
    c = torch.empty((M, N), dtype=out_dtype, device=a.device)
    dt = {torch.float32: _lib.MPF_F32, torch.bfloat16: _lib.MPF_BF16}
    with _lib.device_guard(a.device):
        code = _lib.lib().mpf_gemm3_tn_ex(a.data_ptr(), dt[a.dtype], a.stride(0), planes.data_ptr(), _p(bias), _p(cin),
                                          _rows(cin, N) if cin is not None else 0, c.data_ptr(), dt[out_dtype], N, M, N, K,
                                          1 if relu else 0, _stream(a))
    _lib.check(code, "mpf_gemm3_tn_ex")
    return c


def gemm3_nt_ex(a, b, rows_per_split, want_csum_a=False):
    """gemm3_nt with ONE bf16 operand: a [R, M], b [R, N] (fp32 or bf16, not both bf16; N % 128 == 0) -> (c_part [ns, M, N],
    csum_a [ns, M] or None).  The bf16 operand is its own first plane: three products per step."""
    assert a.is_cuda and a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and a.shape[0] == b.shape[0]
    R, M = a.shape
    N = b.shape[1]
    ns = (R + rows_per_split - 1) // rows_per_split
    c = torch.empty((ns, M, N), dtype=torch.float32, device=a.device)
    ca = torch.empty((ns, M), dtype=torch.float32, device=a.device) if want_csum_a else None
    dt = {torch.float32: _lib.MPF_F32, torch.bfloat16: _lib.MPF_BF16}
    with _lib.device_guard(a.device):
        code = _lib.lib().mpf_gemm3_nt_ex(a.data_ptr(), dt[a.dtype], a.stride(0), b.data_ptr(), dt[b.dtype], b.stride(0), c.data_ptr(),
                                          _p(ca), R, M, N, rows_per_split, _stream(a))
    _lib.check(code, "mpf_gemm3_nt_ex")
    return c, ca
