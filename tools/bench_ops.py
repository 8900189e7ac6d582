"""Stand-alone timing of the non-GEMM hot kernels at the bench shapes (bf16, B=8): PEG fwd/bwd, attention fwd/bwd.
usage: python tools/bench_ops.py [peg|attn|all] [iters]"""
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
print(json.dumps(out, indent=1))
