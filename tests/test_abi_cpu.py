"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares; argument
errors are reported without touching a GPU."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from mp_former_amd import _lib
    return _lib


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(mpf_\w+)\s*\(", src))
    return sorted(names)


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    declared = _declared_symbols()
    assert "mpf_msda_forward" in declared and "mpf_msda_backward" in declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    # and the python binding covers exactly the declared set
    assert sorted(built.SIGNATURES) == declared


def test_abi_version_and_argument_errors(built):
    lib = built.lib()
    assert lib.mpf_abi_version() == built.ABI_VERSION
    one = ctypes.c_void_p(16)  # never dereferenced: argument validation happens first
    # bad dtype
    assert lib.mpf_msda_forward(one, one, one, one, one, one, 1, 1, 1, 1, 1, 1, 1, 99, None) == -1
    assert b"dtype" in lib.mpf_last_error()
    # non-positive size
    assert lib.mpf_msda_forward(one, one, one, one, one, one, 1, 0, 1, 1, 1, 1, 1, 0, None) == -2
    # NULL buffer
    assert lib.mpf_msda_forward(None, one, one, one, one, one, 1, 1, 1, 1, 1, 1, 1, 0, None) == -3
    assert lib.mpf_msda_backward(one, one, one, one, one, one, one, one, None, 1, 1, 1, 1, 1, 1, 1, 0, None) == -3
    assert lib.mpf_set_option(b"no_such_key", 1) == -2
    # round-4 entry points: the bit-mask gate needs N % 128 == 0 and 8-byte aligned mask rows; the next-mask call its buffers
    args = [one, 256, one, one, one, None, None, 0, None, 0, one, 16, one, 128, None, None, 16]
    assert lib.mpf_gemm3_tn_h2_bits(*args, 128, 96, 256, 0, None) == -2
    assert lib.mpf_gemm3_tn_h2_bits(*(args[:11] + [12] + args[12:]), 128, 128, 256, 0, None) == -2           # ldgbits % 8
    assert lib.mpf_gemm3_tn_h2_bits(*(args[:2] + [None] + args[3:]), 128, 128, 256, 0, None) == -3             # no amax slot
    from mp_former_amd.transformer_decoder import MpfNextMask
    m = MpfNextMask()
    assert lib.mpf_next_attn_mask(ctypes.byref(m), None) == -3
    for f in ("x", "ln_gamma", "ln_beta", "w0", "w1", "w2", "pooled", "out", "flags", "scratch"):
        setattr(m, f, 16)
    m.N, m.Q, m.HW, m.pad, m.scratch_bytes = 2, 100, 1024, 0, 16
    assert lib.mpf_next_attn_mask(ctypes.byref(m), None) == -2 and b"scratch" in lib.mpf_last_error()
    assert lib.mpf_next_attn_mask_scratch_bytes(2, 100) >= 3 * 200 * 256 * 2 + 2 * 200 * 4
    assert lib.mpf_next_attn_mask_scratch_bytes(0, 100) == 0


def test_python_mirror_rejects_cpu_tensors(built):
    """Reference behaviour: CPU tensors -> 'Not implemented on the CPU' (ops/src/ms_deform_attn.h:43)."""
    import torch
    from mp_former_amd import MSDeformAttnFunction
    shapes = torch.tensor([[2, 2]], dtype=torch.long)
    lsi = torch.zeros(1, dtype=torch.long)
    v = torch.rand(1, 4, 2, 4)
    loc = torch.rand(1, 3, 2, 1, 2, 2)
    a = torch.rand(1, 3, 2, 1, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(v, shapes, lsi, loc, a, 128)


def test_missing_library_fails_loudly(built, monkeypatch):
    from mp_former_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmpformer_hip.so")
    with pytest.raises(_lib.NativeLibraryError):
        _lib.lib()


def test_shipped_miopen_find_db_is_wired(monkeypatch, tmp_path):
    """the tuned MIOpen solver choice travels with the package: use_shipped_find_db() (an explicit call — bench.py makes
    it; importing the package leaves the environment alone) points MIOPEN_USER_DB_PATH at a private copy of the shipped
    find-db unless the user already chose one (or switched it off)"""
    import glob
    import os
    from mp_former_amd import _miopen
    files = glob.glob(os.path.join(os.path.dirname(_miopen.__file__), "miopen_db", "gfx950*.ufdb.txt"))
    assert files and os.path.getsize(files[0]) > 1000
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("MPF_MIOPEN_DB", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    import importlib
    import mp_former_amd
    importlib.reload(mp_former_amd)
    assert "MIOPEN_USER_DB_PATH" not in os.environ, "importing the package must not touch the environment"
    path = _miopen.use_shipped_find_db()
    assert path and os.path.isdir(path) and os.environ["MIOPEN_USER_DB_PATH"] == path
    # MIOpen appends to its user db: it must get a private copy, never the tracked directory
    assert os.path.realpath(path) != os.path.realpath(os.path.join(os.path.dirname(_miopen.__file__), "miopen_db"))
    assert sorted(os.listdir(path)) == sorted(os.listdir(os.path.join(os.path.dirname(_miopen.__file__), "miopen_db")))
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    assert _miopen.use_shipped_find_db() == path                    # idempotent: a second rank finds the same copy
    assert _miopen.shipped_version() == (3, 5, 0)
    open(os.path.join(path, "gfx950100.HIP.9_9_9_deadbeef.ufdb.txt"), "w").write("x")
    assert _miopen.db_mismatch(path)                                # a db file of another build name is reported
    os.remove(os.path.join(path, "gfx950100.HIP.9_9_9_deadbeef.ufdb.txt"))
    assert not _miopen.db_mismatch(path)
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", str(tmp_path))
    assert _miopen.use_shipped_find_db() == str(tmp_path)           # a user setting wins
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("MPF_MIOPEN_DB", "0")
    assert _miopen.use_shipped_find_db() is None and "MIOPEN_USER_DB_PATH" not in os.environ


def test_encoder_call_structs_match_the_header(tmp_path):
    """The ctypes mirrors of MpfEncoderCall / MpfEncoderBwdCall (mp_former_amd/encoder_fused.py) against the C declarations of
    include/mpformer_hip.h compiled by gcc: same size, same offset of every member, same number of table fields — the two structs
    carry ~30 pointers each, a drift between the two sides would be a silent corruption on the GPU."""
    import ctypes
    import subprocess
    from mp_former_amd import encoder_fused as EF
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    members = {"MpfEncoderCall": [n for n, _ in EF.MpfEncoderCall._fields_], "MpfEncoderBwdCall": [n for n, _ in EF.MpfEncoderBwdCall._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mpformer_hip.h"', 'int main(void) {']
    for st, names in members.items():
        src.append(f'  printf("{st} size %zu\\n", sizeof({st}));')
        for n in names:
            src.append(f'  printf("{st} {n} %zu\\n", offsetof({st}, {n}));')
    src += ['  printf("fields %d %d\\n", (int)MPF_ENC_FIELDS, (int)MPF_ENCB_FIELDS);', '  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    seen = 0
    for line in out:
        t = line.split()
        if not t:
            continue
        if t[0] == "fields":
            assert (int(t[1]), int(t[2])) == (len(EF._ENC_FIELDS), len(EF._ENCB_FIELDS))
            continue
        cls = getattr(EF, t[0])
        if t[1] == "size":
            assert ctypes.sizeof(cls) == int(t[2]), t
        else:
            assert getattr(cls, t[1]).offset == int(t[2]), t
        seen += 1
    assert seen == sum(len(v) for v in members.values()) + 2


def test_item_table_rows_match_the_header(tmp_path):
    """The item tables the mirrors build as int64 rows in numpy (grouped weight gradients, FrozenBN fold, assignment problems)
    against sizeof of the structs they are read as (gcc on include/mpformer_hip.h)."""
    import subprocess
    from mp_former_amd import lsa
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = {"MpfNtItem": 8, "MpfNtItemH2": 10, "MpfScaleCastItem": 6, "MpfLsaProblem": len(lsa.FIELDS)}
    src = ['#include <stdio.h>', '#include "mpformer_hip.h"', 'int main(void) {']
    src += [f'  printf("{k} %zu\\n", sizeof({k}));' for k in want] + ['  return 0;', '}']
    c = tmp_path / "items.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "items"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(c), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).strip().split("\n"))
    assert {k: int(v) for k, v in got.items()} == {k: 8 * n for k, n in want.items()}
