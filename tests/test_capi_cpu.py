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


def test_argument_checks_answer_before_any_launch():
    """Every entry point validates its arguments before it touches the device: null pointers, misaligned strides, missing workspace and
    geometries no kernel serves come back as CTCLIP_EBADARG (-1) / CTCLIP_EUNSUPPORTED (-2) / CTCLIP_EWORKSPACE (-3) with a message in
    ctclip_last_error() -- no GPU needed (this container has none), no launch attempted."""
    from ct_clip_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)            # a 16-byte aligned host address: never dereferenced by the checks
    p = (p + 255) // 256 * 256
    BF16 = 1
    cases = [
        ("ctclip_gemm_headnorm null output", lambda: lib.ctclip_gemm_headnorm(p, p, 256, 1, 512, 512, 512, None, None, None, 1.0, None, None, None, 1.0, None, None, None, 1.0, BF16, None), -1),
        ("ctclip_gemm_headnorm f32", lambda: lib.ctclip_gemm_headnorm(p, p, 256, 1, 512, 512, 512, p, None, None, 1.0, None, None, None, 1.0, None, None, None, 1.0, 0, None), -2),
        ("ctclip_gemm_residual_comp null residue", lambda: lib.ctclip_gemm_residual_comp(p, p, p, None, p, p, 256, 512, 512, 512, 512, 512, 512, BF16, None), -1),
        ("ctclip_peg_fwd_comp null e_out", lambda: lib.ctclip_peg_fwd_comp(p, p, None, None, p, None, 1, 4, 4, 8, 32, BF16, None), -1),
        ("ctclip_peg_fwd_comp D3 = 5", lambda: lib.ctclip_peg_fwd_comp(p, p, None, None, p, p, 1, 4, 4, 5, 32, BF16, None), -2),
        ("ctclip_attn2_bwd_tok odd stride", lambda: lib.ctclip_attn2_bwd_tok(p, p, p, None, 0, 0, p, p, 8.0, p, 257, p, 256, p, p, p, p, 512, p, 512, p, None, 2, 8, 576, p, 1 << 30, None), -1),
        ("ctclip_attn2_bwd_tok no workspace", lambda: lib.ctclip_attn2_bwd_tok(p, p, p, None, 0, 0, p, p, 8.0, p, 256, p, 256, p, p, p, p, 512, p, 512, p, None, 2, 8, 576, None, 0, None), -3),
        ("ctclip_attn2_unprep_q null", lambda: lib.ctclip_attn2_unprep_q(None, p, p, p, 8.0, p, 256, p, 1152, 8, p, 1 << 20, None), -1),
        ("ctclip_patch_embed_param_bwd null", lambda: lib.ctclip_patch_embed_param_bwd(None, p, p, p, p, p, p, p, 512, 4000, 0, None), -1),
        ("ctclip_adam_step step 0", lambda: lib.ctclip_adam_step(p, p, p, p, 1024, 1e-4, 0.9, 0.99, 1e-8, 0, 0.0, None, None, None), -1),
    ]
    for name, call, want in cases:
        rc = call()
        assert rc == want, (name, rc, lib.ctclip_last_error())
        if want != -2:
            assert lib.ctclip_last_error(), name


def test_ctypes_signatures_agree_with_the_header():
    """ct_clip_amd/_lib.py binds every entry point by hand (restype + argtypes); a wrong width there corrupts the call silently.  Every
    prototype of include/ctclip_hip.h is parsed and compared with its binding: same number of parameters, and each parameter the same ABI
    class (pointer / int / int64 / uint32 / uint64 / float / double)."""
    import ctypes as C
    from ct_clip_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "ctclip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)

    def cls_of_c(decl):
        d = decl.strip()
        if "*" in d or d.startswith("hipStream_t"):
            return "ptr"
        base = " ".join(d.split()[:-1]) if len(d.split()) > 1 else d
        base = base.replace("const ", "").strip()
        return {"int": "i32", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "unsigned": "u32", "unsigned int": "u32", "float": "f32",
                "double": "f64", "void": "void"}[base]

    def cls_of_ctypes(t):
        if t in (C.c_void_p, C.c_char_p):
            return "ptr"
        return {C.c_int: "i32", C.c_int64: "i64", C.c_uint64: "u64", C.c_uint32: "u32", C.c_float: "f32", C.c_double: "f64", None: "void"}[t]

    protos = re.findall(r"^\s*([A-Za-z_][\w\s\*]*?)\b(ctclip_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.M)
    assert len(protos) >= 85
    for ret, name, params in protos:
        assert name in _lib.SIGNATURES, name
        restype, argtypes = _lib.SIGNATURES[name]
        cparams = [] if params.strip() in ("", "void") else [p for p in params.split(",")]
        assert len(cparams) == len(argtypes), (name, len(cparams), len(argtypes))
        for i, (cp, at) in enumerate(zip(cparams, argtypes)):
            assert cls_of_c(cp) == cls_of_ctypes(at), (name, i, cp.strip(), at)
        r = ret.replace("const ", "").strip()
        want = "ptr" if "*" in r else {"int": "i32", "int64_t": "i64", "void": "void"}[r]
        assert want == cls_of_ctypes(restype), (name, ret, restype)


def test_every_entry_point_cites_what_it_replaces():
    """include/ctclip_hip.h documents each entry point with the reference call it replaces: a file:line of the reference, or the third-party /
    framework call behind such a line (HF BERT, vector-quantize-pytorch, torch / accelerate), or says that it is C-ABI plumbing."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "ctclip_hip.h")).read()
    items = re.findall(r"/\* ([^\n]*?) \*/\n[^\n]*?(ctclip_\w+)\(", text)
    assert len(items) >= 85
    bad = [n for c, n in items if not re.search(r"\w+\.py:\d+|HF |vector.quantize|accelerate|no reference counterpart|no FFI of its own", c)]
    assert not bad, bad
