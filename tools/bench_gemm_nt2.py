"""Same-process A/B of the two forms of the bf16 NT GEMM (csrc/gemm_nt.hip: one 8-wave workgroup per CU, 256 x 256 tiles; csrc/gemm_nt2.hip:
two 4-wave workgroups per CU, 192 x 128 tiles) on the shapes of the training step at B = 8 (M = 110 592 tokens).
    python tools/bench_gemm_nt2.py [reps]
Operands rotate through four buffer sets (cold operands, as inside the step); the two forms alternate launch by launch."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ct_clip_amd import backend

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
be = backend.get()
dev = torch.device("cuda", 0)
bf = torch.bfloat16
M, Hp, inner = 110592, 1408, 1365
g = torch.Generator(device=dev).manual_seed(0)
NSET = 4


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(bf)


def timed(fn):
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2 * reps)]
    out = {0: [], 7: []}
    for i in range(2 * reps):
        mask = 7 if i & 1 else 0
        be.gemm_nt2_select(mask)
        ev[i][0].record()
        fn(i % NSET)
        ev[i][1].record()
    torch.cuda.synchronize()
    for i in range(2, 2 * reps):
        out[7 if i & 1 else 0].append(ev[i][0].elapsed_time(ev[i][1]) * 1e3)
    return {k: (sum(v) / len(v), min(v)) for k, v in out.items()}


rows = []
cases = []
# plain / residual GEMMs of the step: (name, N, K, residual)
for name, N, K, res in (("to_out fwd + residual", 512, 256, True), ("ff out fwd + residual", 512, 1408, True), ("dX to_q / to_out^T", 512, 256, False),
                        ("dX to_kv", 512, 512, False), ("dX to_out", 256, 512, False), ("dX ff in (K = 2816)", 512, 2816, False)):
    xs = [rnd(M, K) for _ in range(NSET)]
    w = rnd(N, K, scale=K ** -0.5)
    rs = [rnd(M, N) for _ in range(NSET)] if res else None
    outs = [torch.empty(M, N, dtype=bf, device=dev) for _ in range(NSET)]
    cases.append((name, 2.0 * M * N * K, lambda i, xs=xs, w=w, rs=rs, outs=outs: be.gemm(xs[i], w, residual=rs[i] if rs else None, out=outs[i])))
xs = [rnd(M, 512) for _ in range(NSET)]
w_il = be.geglu_weight_interleave(torch.randn(2 * inner, 512, device=dev, generator=g) * 512 ** -0.5, Hp, bf)
cases.append(("ff in fwd + GEGLU (u and g)", 2.0 * M * 2 * Hp * 512, lambda i: be.gemm_geglu(xs[i], w_il, Hp)))
wt = rnd(Hp, 512, scale=inner ** -0.5)
us = [rnd(M, 2 * Hp) for _ in range(2)]
cases.append(("ff out dX + GEGLU bwd", 2.0 * M * Hp * 512, lambda i: be.gemm_dgeglu(xs[i], wt, us[i & 1])))
prev = be.gemm_nt2_select(0)
try:
    for name, flops, fn in cases:
        t = timed(fn)
        rows.append(dict(case=name, first_form_us=round(t[0][0], 1), first_form_min_us=round(t[0][1], 1), second_form_us=round(t[7][0], 1),
                         second_form_min_us=round(t[7][1], 1), ratio=round(t[7][0] / t[0][0], 3), second_form_tflops=round(flops / t[7][0] / 1e6, 1)))
        print(json.dumps(rows[-1]), flush=True)
finally:
    be.gemm_nt2_select(prev)
