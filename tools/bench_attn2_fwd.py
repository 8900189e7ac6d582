"""Times ctclip_attn2_fwd alone at the bench shape (192 sequences x 576 tokens, 8 heads x 32, table bias) -- for ablation libraries
(CTCLIP_LIB=...) and rocprofv3 --pmc passes.  usage: python tools/bench_attn2_fwd.py [iters] [what: fwd|bwd|all]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
be = backend.get()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, dt=torch.bfloat16: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(dt)
nseq, H, gh, gw, D = 192, 8, 24, 24, 32
L, HD = gh * gw, H * D
M = nseq * L
q, kv, do = rnd(M, HD), rnd(M, 2 * HD), rnd(M, HD)
qs, ks = 1 + 0.1 * rnd(D, dt=torch.float32), 1 + 0.1 * rnd(D, dt=torch.float32)
tab = rnd((2 * gh - 1) * (2 * gw - 1), H, dt=torch.float32)
qh, kh, vh, qinv, kinv = be.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
o, lse2 = be.attn2_fwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, nseq, L)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


out = {"lib": os.environ.get("CTCLIP_LIB", "product")}
if what in ("fwd", "all"):
    out["attn2_fwd_us"] = timeit(lambda: be.attn2_fwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, nseq, L))
if what in ("bwd", "all"):
    out["attn2_bwd_nodbias_us"] = timeit(lambda: be.attn2_bwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, o, do, lse2, nseq, L, False))
    out["attn2_bwd_dbias_us"] = timeit(lambda: be.attn2_bwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, o, do, lse2, nseq, L, True))
print(json.dumps(out))
