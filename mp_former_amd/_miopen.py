"""Solver selection for the library convolutions (backbone, FPN): a MIOpen find-db for gfx950 / 256 CUs.

The ResNet and FPN convolutions run on MIOpen.  Its default (immediate-mode) solver choice for the
shapes of the 1024x1024 training step is ~5 ms/step slower than the solvers its own timing search
("find", what ``torch.backends.cudnn.benchmark = True`` triggers) selects, but that search costs ~3 min
of warm-up per process.  ``miopen_db/`` holds the user find-db / perf-db that search wrote on an MI355X
(tools/miopen_find.sh regenerates it); pointing MIOPEN_USER_DB_PATH at a COPY of it gives the tuned choice
with no search.  Shapes that are not in the db fall back to MIOpen's default heuristics; a user-set
MIOPEN_USER_DB_PATH wins.  This only selects among MIOpen's own kernels — no results change beyond the
library's solver-to-solver rounding.

MIOpen appends what it learns to the user db, so it never gets the tracked directory: each process works
on a private copy (per user, per content hash; created under a temporary name and renamed into place, so
ranks that start together cannot collide).  The db files are named after one MIOpen build
(``gfx950100.HIP.<major>_<minor>_<patch>_<hash>``); a different build ignores them silently, which is
logged here once (``logging`` channel "mp_former_amd.miopen") — the step is then ~5 ms slower, not wrong.
"""
import glob
import hashlib
import logging
import os
import shutil
import tempfile

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
_log = logging.getLogger("mp_former_amd.miopen")


def _content_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_DB, "*"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _private_copy():
    root = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    try:
        os.makedirs(root, exist_ok=True)
        if not os.access(root, os.W_OK):
            raise OSError
    except OSError:
        root = tempfile.gettempdir()
    dst = os.path.join(root, "mp_former_amd", f"miopen_db_{_content_hash()}_{os.getuid()}")
    if os.path.isdir(dst):
        return dst
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="miopen_db_tmp_", dir=os.path.dirname(dst))      # unique name, then atomic rename
    for f in glob.glob(os.path.join(_DB, "*")):
        shutil.copy2(f, tmp)
    try:
        os.rename(tmp, dst)
    except OSError:                      # another rank won the race: use its copy
        shutil.rmtree(tmp, ignore_errors=True)
    return dst


def shipped_version():
    """(major, minor, patch) the shipped db files are named after, or None."""
    for f in glob.glob(os.path.join(_DB, "gfx950*.ufdb.txt")):
        parts = os.path.basename(f).split(".HIP.")[-1].split("_")
        try:
            return int(parts[0]), int(parts[1]), int(parts[2])
        except (ValueError, IndexError):
            return None
    return None


def running_version():
    """MIOpen version triple of this process (torch reports it as cudnn.version() = major*1e6 + minor*1e3 + patch)."""
    try:
        import torch
        v = torch.backends.cudnn.version()
        return (v // 1000000, (v // 1000) % 1000, v % 1000) if v else None
    except Exception:      # noqa: BLE001 - version probing must never break the import
        return None


def use_shipped_find_db(check_version=False):
    """Call before the first convolution of the process (MIOpen reads the variable when it initialises)."""
    if "MIOPEN_USER_DB_PATH" in os.environ or os.environ.get("MPF_MIOPEN_DB", "1") != "1" or not os.path.isdir(_DB):
        return os.environ.get("MIOPEN_USER_DB_PATH")
    path = _private_copy()
    os.environ["MIOPEN_USER_DB_PATH"] = path
    if check_version:
        have, want = running_version(), shipped_version()
        if have and want and have != want:
            _log.warning("the shipped MIOpen find-db was written by MIOpen %d.%d.%d, this process runs %d.%d.%d: MIOpen will "
                         "ignore it (default solver heuristics; regenerate with tools/miopen_find.sh)", *want, *have)
    return path


def db_mismatch(path=None):
    """After some convolutions ran: True if MIOpen created db files of its OWN name in the directory, i.e. the shipped
    files (named after another build) are not the ones it reads."""
    path = path or os.environ.get("MIOPEN_USER_DB_PATH")
    if not path or not os.path.isdir(path):
        return False
    shipped = {os.path.basename(f) for f in glob.glob(os.path.join(_DB, "*"))}
    mine = {os.path.basename(f) for f in glob.glob(os.path.join(path, "gfx*.u*db.txt"))}
    extra = mine - shipped
    if extra:
        _log.warning("MIOpen wrote %s next to the shipped find-db: the shipped files do not match this MIOpen build", sorted(extra))
    return bool(extra)
