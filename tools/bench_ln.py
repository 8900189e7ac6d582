"""Times LayerNorm forward / backward at the CTViT token-grid shape (110592 x 512, bf16) through the C ABI.  usage: python tools/bench_ln.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
be = backend.get()
M, D = 110592, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
dy = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
gamma, beta = torch.rand(D, device="cuda") + 0.5, torch.rand(D, device="cuda")


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
    return e0.elapsed_time(e1) * 1e3 / iters


y, mean, rstd = be.layernorm_fwd(x, gamma, beta, 1e-5)
us = timeit(lambda: be.layernorm_fwd(x, gamma, beta, 1e-5))
print(f"layernorm_fwd {us:7.1f} us  {2 * M * D * 2 / us / 1e3:7.0f} GB/s")
yr = torch.nn.functional.layer_norm(x.float(), (D,), gamma, beta, 1e-5)
print("max abs err vs torch", float((y.float() - yr).abs().max()))

dgam, dbet = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
us = timeit(lambda: be.layernorm_bwd(dy, x, gamma, mean, rstd, dgam, dbet))
print(f"layernorm_bwd {us:7.1f} us  {3 * M * D * 2 / us / 1e3:7.0f} GB/s (dy, x -> dx)")
a1 = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
us = timeit(lambda: be.layernorm_bwd(dy, x, gamma, mean, rstd, dgam, dbet, a1, None))
print(f"layernorm_bwd + 1 add {us:7.1f} us  {4 * M * D * 2 / us / 1e3:7.0f} GB/s")
us = timeit(lambda: be.layernorm_bwd(dy, x, gamma, mean, rstd, dgam, dbet, a1, a1))
print(f"layernorm_bwd + 2 adds {us:7.1f} us  {5 * M * D * 2 / us / 1e3:7.0f} GB/s")
