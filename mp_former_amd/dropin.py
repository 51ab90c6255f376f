"""Install the native ops behind the reference's import names."""
import sys


def install():
    """Register ``MultiScaleDeformableAttention`` in sys.modules so the reference's
    ``import MultiScaleDeformableAttention as MSDA`` (ops/functions/ms_deform_attn_func.py:22)
    binds to the HIP kernels."""
    from . import MultiScaleDeformableAttention as mod
    sys.modules["MultiScaleDeformableAttention"] = mod
    return mod


def configure_training_process(single_thread_autograd=True):
    """Process-level settings of a one-process-per-GPU trainer (what ``train_net.py:main`` would call once before building the
    model).  single_thread_autograd: run ``backward()`` on the calling thread (``torch.autograd.set_multithreading_enabled(False)``).
    With one device per process the engine's device thread adds nothing but hand-offs — every one of the ~140 python autograd
    nodes of the step bounces the GIL between two threads — and the launch thread is what bounds the step once the GPU side is
    fast: measured 22.7 -> 18.8 ms of enqueue time per step (tools/host_step_time.py, same box).  Returns the previous setting."""
    import torch
    prev = torch.autograd.is_multithreading_enabled()
    if single_thread_autograd:
        torch.autograd.set_multithreading_enabled(False)
    return prev
