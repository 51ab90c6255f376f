"""Child of tests/test_bench_world2_gpu.py: runs bench.py UNCHANGED as rank RANK of a two-rank job inside a one-GPU lease.
The two things a one-GPU box cannot give the driver's launch line are patched here, in the test, not in the product:
every rank uses cuda:0 (LOCAL_RANK is rewritten before bench.py reads it) and the process group is gloo instead of RCCL
(which refuses two ranks on one device)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LOCAL_RANK"] = "0"

from mp_former_amd import dist as mdist  # noqa: E402

_init = mdist.init_from_env
mdist.init_from_env = lambda backend=None, device=None: _init("gloo", None)

# fault injection (tests only) around the unchanged FlatGradSync:
#   MPF_TEST_LATE_GRAD=1     a gradient shows up for the head bucket AFTER its all-reduce was launched from the res5 hook:
#                            finish() has to raise instead of silently dropping it
#   MPF_TEST_NO_RES3_HOOK=1  the second hook never fires (as if autograd had ordered res3 differently): finish() must launch
#                            that bucket itself and the ranks must still end up with identical parameters
_launch, _finish = mdist.FlatGradSync.launch, mdist.FlatGradSync.finish
_state = {"in_finish": False}


def _launch_patched(self, i):
    if os.environ.get("MPF_TEST_NO_RES3_HOOK") == "1" and i == 1 and not _state["in_finish"]:
        return
    _launch(self, i)
    if os.environ.get("MPF_TEST_LATE_GRAD") == "1" and i == 0 and not _state["in_finish"]:
        import torch
        p = self.groups[0]["params"][0]
        p.grad = torch.zeros_like(p)


def _finish_patched(self):
    _state["in_finish"] = True
    try:
        _finish(self)
    finally:
        _state["in_finish"] = False


mdist.FlatGradSync.launch = _launch_patched
mdist.FlatGradSync.finish = _finish_patched
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
