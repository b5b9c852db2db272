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


def test_winograd_size_queries_match_the_documented_layouts():
    """Host-side size queries of the Winograd entry points (no GPU): packed weights [Cout/64][Cin/8][8][4][64][4]; weight-
    gradient workspace = splits x (16 Cout Cin + Cout) with one round of workgroups (256 / channel-tile pairs) split over
    chunks of 7 or 8 k-steps -- the same rule ops.wino_wgrad_issued_flops uses for the bench's issued-FLOP count."""
    from probabilisticteacher_amd import build_ext, ops
    lib = ctypes.CDLL(build_ext.build())
    lib.ptmi_conv3x3_wino_packed_floats.restype = ctypes.c_int64
    lib.ptmi_conv3x3_wino_wgrad_ws_floats.restype = ctypes.c_int64
    lib.ptmi_colsum_ws_floats.restype = ctypes.c_int64
    cd = lambda a, b: -(-a // b)
    for cin, cout in ((64, 64), (256, 512), (20, 70), (3, 64)):
        assert lib.ptmi_conv3x3_wino_packed_floats(cin, cout) == cd(cout, 64) * cd(cin, 8) * 8 * 4 * 64 * 4
    for n, cin, cout, h, w in ((48, 256, 256, 200, 333), (48, 512, 512, 100, 166), (16, 512, 512, 50, 83), (2, 64, 64, 6, 64),
                               (1, 64, 192, 2, 36), (48, 64, 64, 800, 1333)):
        pairs = cd(cout, 64) * cd(cin, 64)
        tile_pairs = cd(cd(w, 2), 2)
        ksn = 7 if cd(tile_pairs, 7) * 7 < cd(tile_pairs, 8) * 8 else 8
        chunks = n * cd(h, 2) * cd(w, 4 * ksn)
        splits = max(1, min(cd(256, pairs), chunks))
        assert lib.ptmi_conv3x3_wino_wgrad_ws_floats(n, cin, cout, h, w) == splits * (16 * cout * cin + cout)
        assert ops.wino_wgrad_issued_flops(n, cin, cout, h, w) == float(pairs) * chunks * ksn * 16 * 4 * 4096
    assert lib.ptmi_colsum_ws_floats(199200, 15) == cd(199200, 1024) * 15 and lib.ptmi_colsum_ws_floats(100, 15) == 0
    # forward: a flat line of (image, band) strips, 32 columns per workgroup, all four waves counted only where a tile is real
    # period 36 -> two workgroups; the second one holds only gap columns: 2 row halves x 2 channel halves x 1 chunk x 4 k-steps
    assert ops.wino_issued_flops(1, 8, 64, 8, 32) == 4 * 1 * 4 * 16 * 4096
    # two strips = 72 flat columns = 3 workgroups, each with real tiles in both row halves: 6 x 2 channel halves x 2 channel tiles x 2 chunks
    assert ops.wino_issued_flops(2, 16, 128, 8, 32) == 6 * 2 * 2 * 2 * 4 * 16 * 4096
    # H = 12: the second band's lower row half (rows 12 .. 15) is empty
    assert ops.wino_issued_flops(1, 8, 64, 12, 32) == (2 + 1 + 1) * 2 * 1 * 1 * 4 * 16 * 4096


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
