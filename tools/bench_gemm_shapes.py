"""Times every GEMM of one CTViT transformer layer at the bench shape (110592 tokens, bf16) through the C ABI and prices each against
both rooflines: MFMA (2.5 PFLOP/s dense bf16) and HBM (6.3 TB/s achievable for read + write of the operands and the result).
usage: python tools/bench_gemm_shapes.py [iters]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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


M = 110592
layers = [("to_q", 512, 256), ("to_kv", 512, 512), ("to_out", 256, 512), ("ff_in", 512, 2816), ("ff_out", 1408, 512)]
out = {}
if os.environ.get("SHAPES_ONLY") == "fused":      # only the fused GEGLU launches (ablation runs)
    layers = []
elif os.environ.get("SHAPES_ONLY") == "ff":       # feed-forward in-projection, plain and fused, + to_kv
    layers = [l for l in layers if l[0] in ("ff_in", "to_kv")]
for name, K, N in layers:
    x, w, dy = rnd(M, K), rnd(N, K), rnd(M, N)
    wt = w.t().contiguous()
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    dw = torch.zeros(N, K, device=dev)
    cases = {
        "fwd  NT y = x W^T": (lambda: be.gemm(x, w), 2 * M * N * K, (M * K + N * K + M * N) * 2),
        "dgrad NT dx = dy W": (lambda: be.gemm(dy, wt), 2 * M * N * K, (M * N + N * K + M * K) * 2),
        "wgrad TN dW += dy^T x": (lambda: be.gemm(dy, x, a_kc=False, b_kc=False, out=dw, accumulate=True, split_k=0, M=N, N=K, K=M),
                                  2 * M * N * K, (M * N + M * K) * 2 + N * K * 4),
    }
    if name in ("to_out", "ff_out"):      # the two forward GEMMs that add the residual stream in their epilogue (attention.py:324-331)
        res = rnd(M, N)
        cases["fwd  NT y = x W^T + residual"] = (lambda: be.gemm(x, w, residual=res), 2 * M * N * K, (M * K + N * K + 2 * M * N) * 2)
    for cname, (fn, fl, by) in cases.items():
        us = timeit(fn)
        t_mfma, t_hbm = fl / 2.5e15 * 1e6, by / 6.3e12 * 1e6
        out[f"{name} (K={K}, N={N}) {cname}"] = dict(us=round(us, 1), tflops=round(fl / us / 1e6), GBps=round(by / us / 1e3),
                                                     floor_us=round(max(t_mfma, t_hbm), 1), bound="mfma" if t_mfma > t_hbm else "hbm",
                                                     frac_of_floor=round(max(t_mfma, t_hbm) / us, 2))
# the fused GEGLU launches of the feed-forward in-projection (forward with / without the stored u, backward by recomputation) and
# the streaming backward they replace
K, Hp, inner = 512, 1408, 1365
x, w32 = rnd(M, K), (torch.rand(2 * inner, K, device=dev, generator=g) * 2 - 1) * K ** -0.5
w_il = be.geglu_weight_interleave(w32, Hp, torch.bfloat16)
dg = rnd(M, Hp)
u, _ = be.gemm_geglu(x, w_il, Hp)
fl = 2 * M * 2 * Hp * K
for cname, fn, by in [("ff_in fused fwd (u + g)", lambda: be.gemm_geglu(x, w_il, Hp), (M * K + 2 * Hp * K + 3 * M * Hp) * 2),
                      ("ff_in fused fwd (g only)", lambda: be.gemm_geglu(x, w_il, Hp, save_u=False), (M * K + 2 * Hp * K + M * Hp) * 2),
                      ("ff_in geglu bwd by recomputation", lambda: be.gemm_geglu_bwd(x, w_il, dg, Hp), (M * K + 2 * Hp * K + 3 * M * Hp) * 2),
                      ("geglu_bwd streaming (replaced)", lambda: be.geglu_bwd(dg, u), 5 * M * Hp * 2)]:
    us = timeit(fn)
    out[cname] = dict(us=round(us, 1), tflops=round(fl / us / 1e6) if "streaming" not in cname else 0, GBps=round(by / us / 1e3))
# the grad-input GEMM of the out-projection with the GEGLU backward in its epilogue, and the two launches it replaces
dy, wt = rnd(M, K), rnd(Hp, K)
for cname, fn in [("ff_out dgrad + geglu bwd (one launch)", lambda: be.gemm_dgeglu(dy, wt, u)),
                  ("ff_out dgrad, then geglu_bwd (two launches)", lambda: be.geglu_bwd(be.gemm(dy, wt), u))]:
    us = timeit(fn)
    out[cname] = dict(us=round(us, 1), tflops=round(2 * M * Hp * K / us / 1e6), GBps=round((M * K + Hp * K + 4 * M * Hp) * 2 / us / 1e3))
print(json.dumps(out, indent=1))
