"""Stand-alone timing of the non-GEMM hot kernels at the bench shapes (bf16, B=8): PEG, spatial / temporal attention, the streaming
kernels (LayerNorm, GEGLU, qk-norm, head transpose).
usage: python tools/bench_ops.py [peg|attn|tattn|stream|prep|all] [iters]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
be = backend.get()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, dt=torch.bfloat16: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(dt)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out = {}
if what in ("peg", "all"):
    B, D1, D2, D3, C = 8, 24, 24, 24, 512
    x, dy = rnd(B, D1, D2, D3, C), rnd(B, D1, D2, D3, C)
    w, b = rnd(C, 27, dt=torch.float32), rnd(C, dt=torch.float32)
    dw, db = torch.zeros(C, 27, device=dev), torch.zeros(C, device=dev)
    nbytes = x.numel() * 2
    us = timeit(lambda: be.peg_fwd(x, w, b))
    out["peg_fwd (8,24,24,24,512)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(2 * nbytes / us / 1e3, 1))
    us = timeit(lambda: be.peg_bwd(dy, x, w, dw, db))
    out["peg_bwd dx+dw (8,24,24,24,512)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(4 * nbytes / us / 1e3, 1))
    us = timeit(lambda: be.peg_bwd(dy, x, w, None, None))
    out["peg_bwd dx only"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(2 * nbytes / us / 1e3, 1))
if what in ("attn", "all"):
    nseq, H, gh, gw, D = 192, 8, 24, 24, 32
    L, HD = gh * gw, H * D
    M = nseq * L
    nrm = lambda t: torch.nn.functional.normalize(t.float().reshape(M, H, D), dim=-1).view(M, HD).to(torch.bfloat16)
    q, k, v, do = nrm(rnd(M, HD)), nrm(rnd(M, HD)), rnd(M, HD), rnd(M, HD)
    tab = rnd((2 * gh - 1) * (2 * gw - 1), H, dt=torch.float32)
    full = torch.empty(H, L, L, device=dev)
    be.lib.ctclip_cpb_expand(tab.data_ptr(), full.data_ptr(), H, gh, gw, torch.cuda.current_stream().cuda_stream)
    vt = be.head_transpose(v, nseq, H, L, D)
    o, lse = be.attn_fwd(q, k, vt, tab, None, nseq, H, L, D, 8.0, bias_grid=(gh, gw))
    qt, kt, dot = (be.head_transpose(t, nseq, H, L, D) for t in (q, k, do))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    dtab, dfull = torch.zeros_like(tab), torch.zeros_like(full)
    flops_f = 4.0 * nseq * H * L * L * D
    cases = {
        "attn_fwd spatial, table bias": (lambda: be.attn_fwd(q, k, vt, tab, None, nseq, H, L, D, 8.0, bias_grid=(gh, gw)), flops_f),
        "attn_fwd spatial, expanded bias": (lambda: be.attn_fwd(q, k, vt, full, None, nseq, H, L, D, 8.0), flops_f),
        "attn_fwd spatial, no bias": (lambda: be.attn_fwd(q, k, vt, None, None, nseq, H, L, D, 8.0), flops_f),
        "attn_bwd spatial (delta+dq+dkv), table bias": (lambda: be.attn_bwd(q, k, v, qt, kt, o, do, dot, lse, tab, None, dq, dk, dv, None, nseq, H, L, D, 8.0, bias_grid=(gh, gw)), 3.5 * flops_f),
        "attn_bwd spatial + dbias, table bias": (lambda: be.attn_bwd(q, k, v, qt, kt, o, do, dot, lse, tab, None, dq, dk, dv, dtab, nseq, H, L, D, 8.0, bias_grid=(gh, gw)), 4.5 * flops_f),
        "attn_bwd spatial + dbias, expanded bias": (lambda: be.attn_bwd(q, k, v, qt, kt, o, do, dot, lse, full, None, dq, dk, dv, dfull, nseq, H, L, D, 8.0), 4.5 * flops_f),
    }
    for name, (fn, fl) in cases.items():
        us = timeit(fn)
        out[name] = dict(avg_us=round(us, 1), tflops=round(fl / us / 1e6, 1))
if what in ("attn2", "all"):
    # second-generation spatial attention (csrc/attn2.hip) at the bench shape: prep, forward, backward (+ dBias), un-prep
    nseq, H, gh, gw, D = 192, 8, 24, 24, 32
    L, HD = gh * gw, H * D
    M = nseq * L
    q, kv, do = rnd(M, HD), rnd(M, 2 * HD), rnd(M, HD)
    qs, ks = 1 + 0.1 * rnd(D, dt=torch.float32), 1 + 0.1 * rnd(D, dt=torch.float32)
    tab = rnd((2 * gh - 1) * (2 * gw - 1), H, dt=torch.float32)
    flops_f = 4.0 * nseq * H * L * L * D
    qh, kh, vh, qinv, kinv = be.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
    o, lse2 = be.attn2_fwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, nseq, L)
    dqh, dkh, dvh, dtab = be.attn2_bwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, o, do, lse2, nseq, L, True)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    dqs, dks = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    cases = {
        "attn2_prep (q, kv -> head-planar)": (lambda: be.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H), 0.0),
        "attn2_fwd spatial, table bias": (lambda: be.attn2_fwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, nseq, L), flops_f),
        "attn2_fwd spatial, no bias": (lambda: be.attn2_fwd(qh, kh, vh, None, None, qs, ks, 8.0, nseq, L), flops_f),
        "attn2_bwd spatial (dq + dkv)": (lambda: be.attn2_bwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, o, do, lse2, nseq, L, False), 3.5 * flops_f),
        "attn2_bwd spatial + dbias": (lambda: be.attn2_bwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, o, do, lse2, nseq, L, True), 4.5 * flops_f),
        "attn2_unprep": (lambda: be.attn2_unprep(dqh, dkh, dvh, qh, kh, qinv, kinv, qs, ks, 8.0, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks), 0.0),
    }
    for name, (fn, fl) in cases.items():
        us = timeit(fn)
        out[name] = dict(avg_us=round(us, 1), tflops=round(fl / us / 1e6, 1))
if what in ("tattn", "all"):
    # temporal attention of the CTViT: 4608 sequences of 24 tokens, 8 heads x 32, no bias
    nseq, H, L, D = 8 * 576, 8, 24, 32
    HD, M = H * D, nseq * L
    nrm = lambda t: torch.nn.functional.normalize(t.float().reshape(M, H, D), dim=-1).view(M, HD).to(torch.bfloat16)
    q, k, v, do = nrm(rnd(M, HD)), nrm(rnd(M, HD)), rnd(M, HD), rnd(M, HD)
    vt = be.head_transpose(v, nseq, H, L, D)
    o, lse = be.attn_fwd(q, k, vt, None, None, nseq, H, L, D, 8.0)
    qt, kt, dot = (be.head_transpose(t, nseq, H, L, D) for t in (q, k, do))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    out["attn_fwd temporal (4608 x 24)"] = dict(avg_us=round(timeit(lambda: be.attn_fwd(q, k, vt, None, None, nseq, H, L, D, 8.0)), 1))
    out["attn_bwd temporal (4608 x 24)"] = dict(avg_us=round(timeit(
        lambda: be.attn_bwd(q, k, v, qt, kt, o, do, dot, lse, None, None, dq, dk, dv, None, nseq, H, L, D, 8.0)), 1))
    # the same on the one-wave-per-problem kernels (csrc/attn_short.hip): q / kv in, o / dq / dkv out, nothing else
    qq, kkv = rnd(M, HD), rnd(M, 2 * HD)
    qs, ks = torch.ones(32, device=dev), torch.ones(32, device=dev)
    dqs, dks = torch.zeros(32, device=dev), torch.zeros(32, device=dev)
    us = timeit(lambda: be.attn_short_fwd(qq, kkv, qs, ks, nseq, L, H, 8.0))
    out["attn_short_fwd temporal (4608 x 24)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(4 * M * HD * 2 / us / 1e3, 1))
    us = timeit(lambda: be.attn_short_bwd(qq, kkv, qs, ks, do, nseq, L, H, 8.0, dqs, dks))
    out["attn_short_bwd temporal (4608 x 24)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(7 * M * HD * 2 / us / 1e3, 1))
if what in ("stream", "all"):
    M, d, Hp = 110592, 512, 1408
    x, dy = rnd(M, d), rnd(M, d)
    gamma, beta = rnd(d, dt=torch.float32), rnd(d, dt=torch.float32)
    nb = x.numel() * 2
    y, mean, rstd = be.layernorm_fwd(x, gamma, beta, 1e-5)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    us = timeit(lambda: be.layernorm_fwd(x, gamma, beta, 1e-5))
    out["layernorm_fwd (110592 x 512)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(2 * nb / us / 1e3, 1))
    us = timeit(lambda: be.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db))
    out["layernorm_bwd + dgamma/dbeta"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(3 * nb / us / 1e3, 1))
    us = timeit(lambda: be.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, x, dy))
    out["layernorm_bwd + two addends"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(5 * nb / us / 1e3, 1))
    u, dgl = rnd(M, 2 * Hp), rnd(M, Hp)
    us = timeit(lambda: be.geglu_fwd(u))
    out["geglu_fwd (110592 x 2816)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(3 * M * Hp * 2 / us / 1e3, 1))
    us = timeit(lambda: be.geglu_bwd(dgl, u))
    out["geglu_bwd"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(5 * M * Hp * 2 / us / 1e3, 1))
    q = rnd(M, 256)
    sv = rnd(32, dt=torch.float32)
    us = timeit(lambda: be.qk_norm_fwd(q, sv, 8, 32))
    out["qk_norm_fwd (110592 x 256)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(2 * q.numel() * 2 / us / 1e3, 1))
    us = timeit(lambda: be.head_transpose(q, 192, 8, 576, 32))
    out["head_transpose (192 x 8 x 576 x 32)"] = dict(avg_us=round(us, 1), algorithmic_GBps=round(2 * q.numel() * 2 / us / 1e3, 1))
if what in ("prep", "all"):
    # the input pipeline on a typical chest CT: 512 x 512 x 300 int16 at 0.8 x 0.8 x 1.0 mm -> (1, 240, 480, 480) f32
    from ct_clip_amd import preprocess as PP
    vox = torch.randint(-1200, 2000, (512, 512, 300), dtype=torch.int16, device=dev)
    us = timeit(lambda: PP.volume_to_tensor(vox, 1.0, -1024.0, 0.8, 1.0, device=dev))
    out["preprocess_volume 512x512x300 int16 -> 240x480x480 f32"] = dict(avg_us=round(us, 1), out_GBps=round(240 * 480 * 480 * 4 / us / 1e3, 1))
print(json.dumps(out, indent=1))
