"""Times the fused Adam (+ zero_grad) kernel on a flat buffer of the CT-CLIP size (284 M f32 parameters: 32 bytes per parameter and step).
usage: python tools/bench_adam.py [iters]      (A/B builds: tools/build_variant.py x optim.hip:ADAM_UNROLL=8, CTCLIP_LIB=...)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
be = backend.get()
n = 284_000_000
dev = torch.device("cuda", 0)
p, g, m, v = (torch.randn(n, device=dev) * s for s in (1.0, 1e-3, 1e-3, 1e-6))
v.abs_()
clip = torch.tensor([1.0, 1.0], device=dev)


def run(zero):
    for _ in range(2):
        be.adam_step(p, g, m, v, 1e-4, 0.9, 0.99, 1e-8, 3, 0.0, clip, None, zero_grad=zero)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        be.adam_step(p, g, m, v, 1e-4, 0.9, 0.99, 1e-8, 3, 0.0, clip, None, zero_grad=zero)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for zero, nbytes in ((False, 28), (True, 32)):
    us = run(zero)
    print(f"adam_step zero_grad={zero}: {us:8.1f} us  {n * nbytes / us / 1e3:7.0f} GB/s")
