"""CPU: the C-ABI shared library loads and exports every symbol include/ptmi355.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ptmi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptmi_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from probabilisticteacher_amd import build_ext, _lib
    path = build_ext.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ptmi355.h but not exported"
    # the python binding table covers the same set
    assert sorted(_lib.SIGNATURES) == names
    lib.ptmi_abi_version.restype = ctypes.c_int
    assert lib.ptmi_abi_version() == 1
    # pure host-side helpers are callable without a GPU
    lib.ptmi_conv3x3_packed_floats.restype = ctypes.c_int64
    ck = lib.ptmi_conv3x3_ck(64)
    assert lib.ptmi_conv3x3_packed_floats(64, 64) == (64 // ck) * 9 * ck * 64
    assert lib.ptmi_conv3x3_bm(512) == 128 and lib.ptmi_conv3x3_ck(3) == 4 and ck == 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from probabilisticteacher_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    try:
        _lib.load()
    except _lib.PtmiError as e:
        assert "no CPU/eager fallback" in str(e)
    else:
        raise AssertionError("loading a missing libptmi355.so must raise")
