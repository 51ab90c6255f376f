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
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
