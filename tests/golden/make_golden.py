#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (this container
only; /root/reference is absent on the GPU box).  Usage:

    python tests/golden/make_golden.py [msda] [head] ...

Every fixture is data only: inputs (or the closed-form recipe that regenerates them) and the
outputs / gradients the reference's own python path produced for them.

MSDA fixtures come from the reference's `ms_deform_attn_core_pytorch`
(mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72) and autograd
through it; the known-answer problem is the one of the reference's only test,
mask2former/modeling/pixel_decoder/ops/test.py:24-40,66-89 (same seed, same draw order; the
test draws on the CPU generator and then moves to the device, so the inputs are reproducible
here bit for bit).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def _level_start(shapes):
    hw = shapes[:, 0] * shapes[:, 1]
    return torch.cat((hw.new_zeros((1,)), hw.cumsum(0)[:-1]))


def _ref_fwd_bwd(core, value, shapes, loc, attn, grad_out):
    """Run the reference python path in the dtype of `value`; returns out and the three grads."""
    v = value.clone().requires_grad_(True)
    lo = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    out = core(v, shapes, lo, a)
    out.backward(grad_out)
    return out.detach(), v.grad, lo.grad, a.grad


def closed_form(n, salt):
    """Deterministic pseudo-random numbers in [0,1) that need no stored inputs (numpy only)."""
    i = np.arange(n, dtype=np.float64)
    x = np.sin(i * 12.9898 + salt * 78.233) * 43758.5453
    return x - np.floor(x)


def gen_msda():
    core = R.msda_func().ms_deform_attn_core_pytorch

    # ---- (1) the reference's own known-answer problem: ops/test.py ----------------------------
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = _level_start(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    torch.manual_seed(3)

    def draw(channels):
        value = torch.rand(N, S, M, channels) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        return value, loc, attn

    # check_forward_equal_with_pytorch_double (test.py:35-47)
    value, loc, attn = draw(D)
    g = torch.from_numpy(closed_form(N * Lq * M * D, 1).reshape(N, Lq, M * D))
    out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g)
    _save("msda_testpy_double", value=value.numpy(), shapes=shapes.numpy(), level_start=lsi.numpy(),
          loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(),
          out=out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
    # check_forward_equal_with_pytorch_float (test.py:51-63): next draws of the same generator
    value, loc, attn = draw(D)
    out32 = core(value, shapes, loc, attn)
    out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g)
    _save("msda_testpy_float", value=value.numpy(), shapes=shapes.numpy(), level_start=lsi.numpy(),
          loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(), out_f32=out32.numpy(),
          out=out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
    # check_gradient_numerical (test.py:66-89): channel sizes chosen to hit every backward branch
    for channels in [30, 32, 64, 71, 1025, 2048, 3096]:
        value, loc, attn = draw(channels)
        g = torch.from_numpy(closed_form(N * Lq * M * channels, channels).reshape(N, Lq, M * channels))
        out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g)
        if channels <= 71:
            _save(f"msda_testpy_grad_D{channels}", value=value.numpy(), shapes=shapes.numpy(),
                  level_start=lsi.numpy(), loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(),
                  out=out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
        else:
            # big-D: `value` is replaced by a closed-form recipe so it need not be stored; the
            # scatter result is stored as a strided subsample + its total.
            value = torch.from_numpy(closed_form(N * S * M * channels, 7).reshape(N, S, M, channels) * 0.01)
            out, gv, gl, ga = _ref_fwd_bwd(core, value, shapes, loc.double(), attn.double(), g)
            _save(f"msda_testpy_grad_D{channels}", value_recipe=np.array([7, 0.01]), channels=np.array(channels),
                  shapes=shapes.numpy(), level_start=lsi.numpy(), loc=loc.numpy(), attn=attn.numpy(),
                  grad_out_recipe=np.array([channels]),
                  out=out.numpy(), grad_value_stride7=gv.numpy().reshape(-1)[::7].copy(),
                  grad_value_sum=np.array(gv.sum().item()),
                  grad_loc=gl.numpy(), grad_attn=ga.numpy())

    # ---- (2) scaled-down versions of the BASELINE configs (M=8, D=32, L=3, P=4) ---------------
    torch.manual_seed(20260001)
    cases = {
        # name: (N, level shapes)            -- same aspect / level ratios as configs A-E
        "A_square": (2, [(2, 2), (4, 4), (6, 6)]),
        "E_wide": (1, [(1, 2), (2, 4), (4, 8)]),
        "D_odd": (1, [(1, 3), (3, 5), (5, 9)]),          # odd sizes, S*M not a multiple of 32
    }
    for name, (n, lv) in cases.items():
        M, D, L, P = 8, 32, 3, 4
        shapes = torch.as_tensor(lv, dtype=torch.long)
        lsi = _level_start(shapes)
        S = int((shapes[:, 0] * shapes[:, 1]).sum())
        Lq = S
        value = torch.randn(n, S, M, D)
        # locations: mostly in range, ~15 % outside [0,1] (incl. beyond the -1 / H cut-offs),
        # a few exactly on pixel centres and on the borders
        loc = torch.rand(n, Lq, M, L, P, 2) * 1.5 - 0.25
        loc[0, 0, 0, :, :, :] = 0.0
        loc[0, 0, 1, :, :, :] = 1.0
        loc[0, 1, 0, 0, :, 0] = (torch.arange(P, dtype=torch.float32) + 0.5) / lv[0][1]
        loc[0, 1, 0, 0, :, 1] = 0.5 / lv[0][0]
        loc[0, 1, 1, :, :, :] = -2.0
        loc[0, 1, 2, :, :, :] = 3.0
        attn = torch.softmax(torch.randn(n, Lq, M, L * P), -1).view(n, Lq, M, L, P)
        g = torch.randn(n, Lq, M * D)
        out, gv, gl, ga = _ref_fwd_bwd(core, value.double(), shapes, loc.double(), attn.double(), g.double())
        _save(f"msda_cfg_{name}", value=value.numpy(), shapes=shapes.numpy(), level_start=lsi.numpy(),
              loc=loc.numpy(), attn=attn.numpy(), grad_out=g.numpy(),
              out=out.numpy(), grad_value=gv.numpy().astype(np.float32), grad_loc=gl.numpy(),
              grad_attn=ga.numpy())


def main():
    what = sys.argv[1:] or ["msda", "head"]
    torch.set_num_threads(4)
    for w in what:
        print(f"[{w}]")
        globals()["gen_" + w]()


if __name__ == "__main__":
    main()
