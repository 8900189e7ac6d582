"""The C-ABI library must build, load, and export every symbol include/ctclip_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ctclip_hip.h")).read()
    return sorted(set(re.findall(r"\b(ctclip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ct_clip_amd import build, _lib
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"missing symbol {s}"
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)


def test_loader_binds_and_reports_version():
    from ct_clip_amd import _lib
    lib = _lib.load()
    assert lib.ctclip_abi_version() == 1
    assert lib.ctclip_target_arch() == b"gfx950"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ct_clip_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(ImportError, match="no PyTorch/CPU fallback"):
        _lib.load()


def test_docs_state_the_header_entry_point_count():
    """DESIGN.md / INTEGRATION.md quote the number of extern "C" entry points: it must be the header's (the docs drifted once: 71 vs 78)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ctclip_hip.h")).read()
    n = len(re.findall(r"^[A-Za-z_][\w \*]*\bctclip_\w+\(", hdr, flags=re.M))
    assert n >= 85
    for doc in ("DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(root, doc)).read()
        counts = {int(m) for m in re.findall(r"(\d+) `extern \"C\"`", text)} | {int(m) for m in re.findall(r"\((\d+) entry points\)", text)}
        assert counts == {n}, (doc, counts, n)
