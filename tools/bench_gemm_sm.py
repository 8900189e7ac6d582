"""Text-tower GEMM shapes (M = B * T rows) through ctclip_gemm: us per launch and TFLOP/s.  Run twice (CTCLIP_GEMM_SM=0 / 1) to compare the
generic 128 x 128 kernel of gemm.hip with gemm_sm.hip.   python tools/bench_gemm_sm.py [T]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ct_clip_amd import backend

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
be = backend.get()
dev = torch.device("cuda", 0)
bf = torch.bfloat16
M = 8 * T
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g)).to(bf)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


rows = []
for name, N, K in (("qkv", 2304, 768), ("attn out", 768, 768), ("ffn in", 3072, 768), ("ffn out", 768, 3072)):
    a, w, bias, res = rnd(M, K), rnd(N, K), torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    wt = rnd(K, N)
    dy = rnd(M, N)
    dw = torch.zeros(N, K, device=dev)
    us = timed(lambda: be.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32))
    rows.append(dict(gemm=f"{name} fwd NT {M}x{N}x{K}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1)))
    us = timed(lambda: be.gemm(dy, wt, out_dtype=torch.float32))
    rows.append(dict(gemm=f"{name} dX NT (transposed shadow) {M}x{K}x{N}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1)))
    us = timed(lambda: be.gemm(dy, w, a_kc=True, b_kc=False, out_dtype=torch.float32))
    rows.append(dict(gemm=f"{name} dX NN (rounds 1-4) {M}x{K}x{N}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1)))
    us = timed(lambda: be.gemm(dy, a, a_kc=False, b_kc=False, out=dw, accumulate=True, split_k=0, M=N, N=K, K=M))
    rows.append(dict(gemm=f"{name} dW TN {N}x{K}x{M}", us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1)))
for r in rows:
    print(json.dumps(r), flush=True)
print(json.dumps(dict(total_us_per_layer=round(sum(r["us"] for r in rows if "rounds 1-4" not in r["gemm"]), 1), sm=os.environ.get("CTCLIP_GEMM_SM", "1"), T=T)))
