"""Device-side linear sum assignment (csrc/lsa.hip; include/mpformer_hip.h mpf_lsa_assign): SciPy's
``linear_sum_assignment`` (matcher.py:149-151 of the reference) without leaving the GPU."""
import numpy as np
import torch

from . import _lib
from ._h2d import upload

FIELDS = ("cost_off", "n_rows", "n_cols", "row_stride", "out_pos", "col_base", "a_base", "a_stride", "b_base", "b_stride",
          "scatter_base")
MAX_DIM = 512

# Failure reporting without a host synchronisation: the kernel ORs 1 into a device word when a problem has no
# finite assignment (NaN / all-infinite costs — scipy.optimize.linear_sum_assignment raises ValueError there, matcher.py:151);
# the word is copied to pinned host memory behind the launch and LOOKED AT on the next call (or by check_status()), i.e.
# one step late at worst, instead of training on with unassigned pairs.
_status = {}     # device -> {"dev": int32[1], "host": pinned int32[1], "event": Event or None}


def _status_of(dev):
    st = _status.get(dev)
    if st is None:
        st = {"dev": torch.zeros(1, dtype=torch.int32, device=dev), "host": torch.zeros(1, dtype=torch.int32).pin_memory(),
              "event": None}
        _status[dev] = st
    return st


def check_status(dev=None, block=False):
    """Raise ValueError if an earlier device assignment was infeasible.  Non-blocking unless ``block``."""
    for d, st in list(_status.items()):
        if dev is not None and d != dev:
            continue
        ev = st["event"]
        if ev is None:
            continue
        if block:
            ev.synchronize()
        if ev.query():
            st["event"] = None
            if int(st["host"][0]) != 0:
                st["dev"].zero_()
                st["host"].zero_()
                raise ValueError("device linear_sum_assignment: cost matrix is infeasible or contains invalid numeric entries "
                                 "(NaN / no finite assignment) — the reference's SciPy call raises here (matcher.py:151)")


def lsa_assign(cost, problems, n_slots, want_rows=True, want_cols=True, want_a=False, want_b=False, scatter_dst=None,
               scatter_src=None, into=None):
    """cost: fp32 CUDA tensor; problems: int64 numpy [n, 11] (FIELDS) -> dict of device index tensors of
    length n_slots: "rows" / "cols" int32, "a" / "b" int64.  ``into`` may supply existing tensors for
    some of them (slots no problem writes keep their contents; fresh tensors are torch.empty)."""
    assert cost.is_cuda and cost.dtype == torch.float32 and cost.is_contiguous()
    problems = np.ascontiguousarray(problems, dtype=np.int64).reshape(-1, len(FIELDS))
    dev = cost.device
    out = dict(into or {})
    for k, dt in (("rows", torch.int32), ("cols", torch.int32), ("a", torch.int64), ("b", torch.int64)):
        if k in out:
            assert out[k].dtype == dt and out[k].is_contiguous() and out[k].numel() >= n_slots
    if want_rows and "rows" not in out:
        out["rows"] = torch.empty(n_slots, dtype=torch.int32, device=dev)
    if want_cols and "cols" not in out:
        out["cols"] = torch.empty(n_slots, dtype=torch.int32, device=dev)
    if want_a and "a" not in out:
        out["a"] = torch.empty(n_slots, dtype=torch.int64, device=dev)
    if want_b and "b" not in out:
        out["b"] = torch.empty(n_slots, dtype=torch.int64, device=dev)
    n = problems.shape[0]
    if n == 0:
        return out
    max_dim = int(max(problems[:, 1].max(), problems[:, 2].max()))
    max_entries = int((problems[:, 1] * problems[:, 2]).max())
    if max_dim > MAX_DIM:
        raise RuntimeError(f"device assignment handles up to {MAX_DIM} rows / columns per problem, got {max_dim}")
    check_status(dev)                       # a failure of an earlier step surfaces here
    st = _status_of(dev)
    pd = upload(problems.reshape(-1), dev)
    p = lambda k: out[k].data_ptr() if k in out else None  # noqa: E731
    with _lib.device_guard(dev):
        code = _lib.lib().mpf_lsa_assign_status(cost.data_ptr(), pd.data_ptr(), n, max_dim, max_entries, p("rows"), p("cols"), p("a"),
                                                p("b"), scatter_dst.data_ptr() if scatter_dst is not None else None,
                                                scatter_src.data_ptr() if scatter_src is not None else None,
                                                st["dev"].data_ptr(), _lib.stream_ptr(dev))
    _lib.check(code, "mpf_lsa_assign_status")
    if st["event"] is None:                 # one read-back in flight at a time (16 bytes, pinned, behind the kernel)
        st["host"].copy_(st["dev"], non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record(torch.cuda.current_stream(dev))
    return out
