"""How much does the last partial round of the persistent NT GEMM cost?  N = 512 outputs at M = 110 592 are 864 tiles of 256 x 256 on 256 CUs
(3.375 rounds); M = 98 304 is 768 tiles (3.0 rounds) and M = 131 072 is 1 024 (4.0).  usage: python tools/probe_gemm_tail.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
be = backend.get()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *sh: (torch.rand(*sh, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)


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


for name, K, N in (("to_out", 256, 512), ("to_kv", 512, 512), ("ff_out", 1408, 512), ("ff_in dgrad", 2816, 512)):
    w = rnd(N, K)
    row = []
    for M in (98304, 110592, 131072):
        x = rnd(M, K)
        us = timeit(lambda: be.gemm(x, w))
        row.append((M, us))
    t3, t3375, t4 = (u for _, u in row)
    print(f"{name:12s} K={K:5d} N={N}: 768 tiles {t3:7.1f} us | 864 tiles {t3375:7.1f} us | 1024 tiles {t4:7.1f} us | "
          f"864 / 768 = {t3375 / t3:.3f} (work 1.125, rounds 1.333) | per tile-round {t4 / 4:6.1f} us")
