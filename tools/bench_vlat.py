"""Times to_visual_latent forward / backward at the bench shape (8 x 294 912 -> 512, f32 weight = 604 MB).  usage: python tools/bench_vlat.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
be = backend.get()
dev = torch.device("cuda", 0)
K, N = 294912, 512
w = torch.randn(N, K, device=dev) * 0.01
dw = torch.zeros(N, K, device=dev)


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
    return e0.elapsed_time(e1) * 1e3 / iters


for Bm in (8, 18):
    x = torch.randn(Bm, K, device=dev)
    dy = torch.randn(Bm, N, device=dev)
    print(f"rows {Bm:2d}: fwd {timeit(lambda: be.visual_latent_fwd(x, w)):7.1f} us | bwd overwrite {timeit(lambda: be.visual_latent_bwd(dy, x, w, dw, accumulate=False)):7.1f} us"
          f" | bwd accumulate {timeit(lambda: be.visual_latent_bwd(dy, x, w, dw, accumulate=True)):7.1f} us")
