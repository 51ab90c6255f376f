"""Solver selection for the library convolutions (backbone, FPN): a MIOpen find-db for gfx950 / 256 CUs.

The ResNet and FPN convolutions run on MIOpen.  Its default (immediate-mode) solver choice for the
shapes of the 1024x1024 training step is ~5 ms/step slower than the solvers its own timing search
("find", what ``torch.backends.cudnn.benchmark = True`` triggers) selects, but that search costs ~3 min
of warm-up per process.  ``miopen_db/`` holds the user find-db / perf-db that search wrote on an MI355X
(tools/miopen_find.sh regenerates it); pointing MIOPEN_USER_DB_PATH at it gives the tuned choice with
no search.  Shapes that are not in the db fall back to MIOpen's default heuristics; a user-set
MIOPEN_USER_DB_PATH wins.  This only selects among MIOpen's own kernels — no results change beyond the
library's solver-to-solver rounding.
"""
import os
import shutil
import tempfile

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_shipped_find_db():
    """Call before the first convolution of the process (MIOpen reads the variable when it initialises)."""
    if "MIOPEN_USER_DB_PATH" in os.environ or os.environ.get("MPF_MIOPEN_DB", "1") != "1" or not os.path.isdir(_DB):
        return os.environ.get("MIOPEN_USER_DB_PATH")
    path = _DB
    if not os.access(_DB, os.W_OK):       # MIOpen appends what it learns: give it a private copy
        path = os.path.join(tempfile.gettempdir(), f"mpf_miopen_db_{os.getuid()}")
        if not os.path.isdir(path):
            shutil.copytree(_DB, path)
    os.environ["MIOPEN_USER_DB_PATH"] = path
    return path
