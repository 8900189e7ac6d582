"""Stand-alone timing of the dominant GEMM shapes of the CT-CLIP step (bf16, B=8 full config) through the C-ABI,
for rocprofv3 --pmc passes (HBM traffic, MFMA busy).  usage: python tools/bench_gemm.py [iters]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
be = backend.get()
dev = "cuda"
M, d, H2, Hp = 110592, 512, 2816, 1408
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
x, w1, u = rnd(M, d), rnd(H2, d), rnd(M, H2)
w2, gh = rnd(d, Hp), rnd(M, Hp)
dw = torch.zeros(1365, d, device=dev)
u3, dw3 = rnd(M, 1536), torch.zeros(1536, d, device=dev)
cases = {
    "NT ff_in  M=110592 N=2816 K=512": (lambda: be.gemm(x, w1), 2.0 * M * H2 * d, (M * d + H2 * d + M * H2) * 2),
    "NT ff_out M=110592 N=512 K=1408": (lambda: be.gemm(gh, w2, residual=x), 2.0 * M * d * Hp, (M * Hp + d * Hp + 2 * M * d) * 2),
    "NN dX     M=110592 N=512 K=2816": (lambda: be.gemm(u, w1, a_kc=True, b_kc=False), 2.0 * M * H2 * d, (M * H2 + H2 * d + M * d) * 2),
    "TN dW     M=1365 N=512 K=110592": (lambda: be.gemm(u[:, :1365], x, a_kc=False, b_kc=False, out=dw, accumulate=True, split_k=0,
                                                         M=1365, N=d, K=M), 2.0 * M * 1365 * d, (M * 1365 + M * d) * 2 + 1365 * d * 8),
    "TN dWqkv  M=1536 N=512 K=110592 (dense lda)": (lambda: be.gemm(u3, x, a_kc=False, b_kc=False, out=dw3, accumulate=True, split_k=0),
                                                    2.0 * M * 1536 * d, (M * 1536 + M * d) * 2 + 1536 * d * 8),
}
out = {}
only = os.environ.get("GEMM_CASES")   # e.g. GEMM_CASES=NT
for name, (fn, flops, bytes_) in cases.items():
    if only and not name.startswith(only):
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    out[name] = dict(avg_us=round(us, 1), tflops=round(flops / us / 1e6, 1), algorithmic_GB=round(bytes_ / 1e9, 3),
                     algorithmic_GBps=round(bytes_ / us / 1e3, 1))
print(json.dumps(out, indent=1))
