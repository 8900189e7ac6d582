"""Timing experiment: what do head-planar 64-byte stores cost in the persistent NT GEMM's epilogue?  Times the to_q / to_kv shapes (M = 110592,
K = 512, N = 256 / 512 / 768) with the library given by CTCLIP_LIB (product: token-major rows; tools/build_variant.py planar ...: the
[N / 32][M][32] layout of the attention operands).   usage: [CTCLIP_LIB=ct_clip_amd/libctclip_planar.so] python tools/bench_planar_store.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

be = backend.get()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *sh: (torch.rand(*sh, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
M, K = 110592, 512
x = rnd(M, K)
for N in (256, 512, 768):
    w = rnd(N, K)
    for _ in range(5):
        be.gemm(x, w)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); be.gemm(x, w); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    print(f"lib={os.environ.get('CTCLIP_LIB', 'product'):40s} N={N}: median {ts[10]:.1f} us, min {ts[0]:.1f} us")
