"""C-ABI: the library builds for gfx950, loads without a GPU and exports every symbol include/yume_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "yume_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yume_[a-z0-9_]+)\s*\(", src)))


def test_build_and_exports():
    import __graft_entry__ as g
    g.build()
    from yume_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in yume_hip.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms


def test_info_calls_without_gpu():
    from yume_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "yume_hip.h")).read()
    assert lib.yume_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define YUME_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.yume_target_arch() == b"gfx950"


def test_argument_validation_no_compute():
    """bad arguments are rejected on the host (no kernel launch, so this runs without a GPU)."""
    from yume_amd import _lib
    lib = _lib.load()
    rc = lib.yume_gemm_bf16(None, 0, None, 0, None, 1, 1, 64, 0, None, 0, None, 0, None, None, 0, 0, 0, None)
    assert rc == -1 and b"NULL" in lib.yume_last_error()
    rc = lib.yume_gemm_bf16(16, 64, 16, 64, None, 4, 4, 63, 0, 16, 4, None, 0, None, None, 0, 0, 0, None)
    assert rc == -1 and b"multiple of 64" in lib.yume_last_error()
    rc = lib.yume_attn_fwd(16, 128, 16, 128, 16, 7, 16, 128, 4, 4, 1, 1.0, 0, 0, None)
    assert rc == -1


def test_product_path_has_no_oracle_import():
    """nothing under yume_amd/ may import the oracle (it is test infrastructure)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "yume_amd")):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_stale_library_is_refused(monkeypatch):
    """a library reporting another ABI version must not be bound (ctypes signatures would not match its argument lists)."""
    import pytest
    from yume_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.load()
