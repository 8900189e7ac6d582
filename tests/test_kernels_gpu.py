"""Per-kernel parity on a real MI355X: every primitive of the HIP backend (C-ABI) against the pure-torch checker on the same
seeded inputs.  f32 mode is the parity mode (tight tolerances; MFMA f32 is an exact fmaf chain); bf16 is the performance mode
(tolerances scaled to bf16 rounding of the inputs/outputs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.ref_backend import RefBackend  # noqa: E402

DEV = "cuda"
DT = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def hip():
    from ct_clip_amd import backend
    return backend.HipBackend()


@pytest.fixture(scope="module")
def ref():
    return RefBackend()


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def tol(dtype, f32=(1e-4, 1e-5), bf16=(3e-2, 3e-2)):
    return dict(rtol=f32[0], atol=f32[1]) if dtype == torch.float32 else dict(rtol=bf16[0], atol=bf16[1])


def close(a, b, **kw):
    torch.testing.assert_close(a.float(), b.float(), **kw)


# ---------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (130, 70, 72), (64, 2816, 512), (300, 512, 1408), (8, 8, 8),
                                   (4096, 2816, 256), (4000, 2900, 192), (4096, 2560, 64), (8200, 1408, 2816)])   # large-tile paths
def test_gemm_nt_bias_residual(hip, ref, dtype, M, N, K):
    a, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2)   # asymmetric operands (transpose-detecting)
    bias, res = rnd(N, seed=3), rnd(M, N, dtype=dtype, seed=4)
    y = hip.gemm(a, b, bias=bias, residual=res)
    yr = ref.gemm(a, b, bias=bias, residual=res)
    s = K ** 0.5
    close(y, yr, **tol(dtype, (1e-4, 1e-4 * s), (2e-2, 2e-2 * s)))
    y32 = hip.gemm(a, b, out_dtype=torch.float32, alpha=0.5)
    close(y32, ref.gemm(a, b, out_dtype=torch.float32, alpha=0.5), **tol(dtype, (1e-4, 1e-4 * s), (1e-2, 1e-2 * s)))


@pytest.mark.parametrize("M,N,K", [(8192, 2816, 512), (20000, 1408, 192), (12300, 2100, 128), (16384, 4096, 64)])
def test_gemm_nt_persistent_multi_tile(hip, ref, M, N, K):
    """More tiles than CUs: the persistent NT kernel's panel stream runs across tile boundaries and the counted vmcnt waits have
    to account for the epilogue stores of the previous tile (fast path) or drain them (ragged tiles, residual, accumulate)."""
    dtype = torch.bfloat16
    a, b = rnd(M, K, dtype=dtype, seed=5), rnd(N, K, dtype=dtype, seed=6)
    bias, res = rnd(N, seed=7), rnd(M, N, dtype=dtype, seed=8)
    s = K ** 0.5
    for kw in (dict(), dict(bias=bias), dict(out_dtype=torch.float32, alpha=0.25), dict(bias=bias, out_dtype=torch.float32),
               dict(residual=res), dict(bias=bias, residual=res)):
        close(hip.gemm(a, b, **kw), ref.gemm(a, b, **kw), rtol=2e-2, atol=2e-2 * s)
    acc = rnd(M, N, seed=9)
    out = acc.clone()
    hip.gemm(a, b, out=out, accumulate=True)
    close(out, acc + ref.gemm(a, b, out_dtype=torch.float32), rtol=2e-2, atol=2e-2 * s)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(256, 128, 256), (130, 72, 70), (512, 1408, 512), (49, 64, 2), (4096, 2560, 512), (4100, 2568, 448)])
def test_gemm_nn_grad_input(hip, ref, dtype, M, N, K):
    # dx (M,N) = dy (M,K) @ W (K,N):  b is stored (K, N) and used with b_kc=False; K tails inside a 16-byte chunk are legal
    dy, w = rnd(M, (K + 7) // 8 * 8, dtype=dtype, seed=5)[:, :K], rnd(K, N, dtype=dtype, seed=6)
    y = hip.gemm(dy, w, a_kc=True, b_kc=False)
    close(y, ref.gemm(dy, w, a_kc=True, b_kc=False), **tol(dtype, (1e-4, 1e-4 * K ** 0.5), (2e-2, 2e-2 * K ** 0.5)))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("T,N,K,split", [(512, 128, 256, 1), (1000, 70, 130, 4), (4096, 1365, 512, 8), (640, 512, 4000, 3),
                                         (8192, 1365, 512, 0), (16384, 512, 1365, 0), (8192, 256, 512, 0)])
def test_gemm_tn_grad_weight_splitk_accumulate(hip, ref, dtype, T, N, K, split):
    # dW (N,K) += dy (T,N)^T @ x (T,K): both operands stored token-major, contraction over rows; odd N exercises pair loads
    Np = N + (N % 2) + 2
    dyp, x = rnd(T, Np, dtype=dtype, seed=7), rnd(T, (K + 7) // 8 * 8 + 8, dtype=dtype, seed=8)
    dy = dyp[:, :N]
    out = rnd(N, K, seed=9)
    out_ref = out.clone()
    hip.gemm(dy, x[:, :K], a_kc=False, b_kc=False, out=out, accumulate=True, split_k=split, M=N, N=K, K=T)
    ref.gemm(dy, x[:, :K], a_kc=False, b_kc=False, out=out_ref, accumulate=True, M=N, N=K, K=T)
    close(out, out_ref, **tol(dtype, (1e-4, 2e-4 * T ** 0.5), (2e-2, 2e-2 * T ** 0.5)))


@pytest.mark.parametrize("dtype", DT)
def test_gemm_argmax(hip, ref, dtype):
    xn = torch.nn.functional.normalize(rnd(700, 128, seed=10), dim=-1).to(dtype)
    en = torch.nn.functional.normalize(rnd(1000, 128, seed=11), dim=-1).to(dtype)
    idx, val = hip.gemm_argmax(xn, en)
    ridx, rval = ref.gemm_argmax(xn, en)
    close(val, rval, **tol(dtype, (1e-5, 1e-5), (1e-2, 1e-2)))
    agree = (idx == ridx).float().mean().item()
    assert agree >= (0.999 if dtype == torch.float32 else 0.97), agree
    # ties -> lowest index
    e2 = en.clone(); e2[5] = e2[900]
    idx2, _ = hip.gemm_argmax(xn, e2)
    assert not (idx2 == 900).any()


@pytest.mark.parametrize("M,N,K", [(8192, 8192, 256), (5000, 8200, 128)])
def test_gemm_argmax_large_runs_on_the_nt_kernel(hip, ref, M, N, K):
    """The vector-quantiser code search at training size (110592 x 8192) goes through the persistent NT kernel's arg-max epilogue."""
    xn = torch.nn.functional.normalize(rnd(M, K, seed=12), dim=-1).to(torch.bfloat16)
    en = torch.nn.functional.normalize(rnd(N, K, seed=13), dim=-1).to(torch.bfloat16)
    idx, val = hip.gemm_argmax(xn, en)
    sim = xn.float() @ en.float().t()
    rval, ridx = sim.max(dim=-1)
    close(val, rval, rtol=1e-3, atol=1e-3)
    agree = (idx == ridx).float().mean().item()
    assert agree >= 0.99, agree
    # where the indices differ the values are ties up to accumulation order
    picked = sim.gather(1, idx[:, None])[:, 0]
    assert (rval - picked).max().item() < 2e-3
    e2 = en.clone(); e2[5] = e2[N - 7]                     # ties -> lowest index
    idx2, _ = hip.gemm_argmax(xn, e2)
    assert not (idx2 == N - 7).any()


# ---------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols", [(37, 512), (1000, 768), (5, 64), (130, 2048)])
def test_layernorm_fwd_bwd(hip, ref, dtype, rows, cols):
    x, dy = rnd(rows, cols, dtype=dtype, seed=1), rnd(rows, cols, dtype=dtype, seed=2)
    gamma, beta = 1 + 0.1 * rnd(cols, seed=3), 0.1 * rnd(cols, seed=4)
    y, mean, rstd = hip.layernorm_fwd(x, gamma, beta, 1e-5)
    yr, mr, rr = ref.layernorm_fwd(x, gamma, beta, 1e-5)
    close(y, yr, **tol(dtype)); close(mean, mr, rtol=1e-4, atol=1e-5); close(rstd, rr, rtol=1e-4, atol=1e-5)
    dg, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    dgr, dbr = dg.clone(), db.clone()
    dx = hip.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db)
    dxr = ref.layernorm_bwd(dy, x, gamma, mr, rr, dgr, dbr)
    close(dx, dxr, **tol(dtype, (1e-4, 1e-5), (3e-2, 3e-2)))
    close(dg, dgr, rtol=1e-3, atol=1e-3 * rows ** 0.5); close(db, dbr, rtol=1e-3, atol=1e-3 * rows ** 0.5)
    y2, _, _ = hip.layernorm_fwd(x, gamma, None, 1e-5)   # gamma-only LayerNorm (attention.py:28-35)
    close(y2, ref.layernorm_fwd(x, gamma, None, 1e-5)[0], **tol(dtype))
    # gradients of the input's other consumers folded into dx (residual, k/v projection)
    a1, a2 = rnd(rows, cols, dtype=dtype, seed=5), rnd(rows, cols, dtype=dtype, seed=6)
    close(hip.layernorm_bwd(dy, x, gamma, mean, rstd, None, None, a1, a2), ref.layernorm_bwd(dy, x, gamma, mr, rr, None, None, a1, a2),
          **tol(dtype, (1e-4, 1e-5), (3e-2, 3e-2)))
    close(hip.layernorm_bwd(dy, x, gamma, mean, rstd, None, None, a1), ref.layernorm_bwd(dy, x, gamma, mr, rr, None, None, a1),
          **tol(dtype, (1e-4, 1e-5), (3e-2, 3e-2)))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(2, 32, 64, 64, 16, 16, 16), (1, 20, 40, 60, 10, 20, 20), (1, 6, 9, 6, 3, 3, 3), (1, 40, 80, 100, 10, 20, 20)])   # (the last: 16 rows of 5 tokens -> the XCD-grouped token order)
def test_patch_ln(hip, ref, dtype, geom):
    B, Fr, H, W, pt, p1, p2 = geom
    K = pt * p1 * p2
    kpad = (K + 63) // 64 * 64
    video = rnd(B, 1, Fr, H, W, seed=5)
    out = hip.patch_ln(video, pt, p1, p2, kpad, 1e-5, dtype)
    close(out, ref.patch_ln(video, pt, p1, p2, kpad, 1e-5, dtype), **tol(dtype, (1e-4, 1e-5), (1e-2, 1e-2)))


@pytest.mark.parametrize("dtype", DT)
def test_l2norm_rows(hip, ref, dtype):
    x = rnd(300, 512, dtype=dtype, seed=6)
    y, inv = hip.l2norm_rows(x, dtype)
    yr, ir = ref.l2norm_rows(x, dtype)
    close(y, yr, **tol(dtype, (1e-5, 1e-6), (1e-2, 1e-2))); close(inv, ir, rtol=1e-4, atol=1e-6)
    y32, _ = hip.l2norm_rows(rnd(16, 64, seed=7), torch.float32)
    close(y32, ref.l2norm_rows(rnd(16, 64, seed=7), torch.float32)[0], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- GEGLU fused into the in-projection GEMM
@pytest.mark.parametrize("M,inner,K", [(256 * 40, 1365, 512), (256 * 90, 341, 128), (256 * 14, 1365, 512)])
def test_gemm_geglu_fused(hip, ref, M, inner, K):
    """attention.py:39-48: u = [x | gate] and g = x * gelu(gate) from one launch against the interleaved weight; the last shape does not
    fill the chip and must be declined (None)."""
    bf = torch.bfloat16
    Hp = (inner + 127) // 128 * 128
    x, w = rnd(M, K, dtype=bf, seed=1), rnd(2 * inner, K, seed=2, scale=K ** -0.5)
    w_il, w_ilr = hip.geglu_weight_interleave(w, Hp, bf), ref.geglu_weight_interleave(w, Hp, bf)
    assert torch.equal(w_il, w_ilr)
    got = hip.gemm_geglu(x, w_il, Hp)
    if (M // 256) * (2 * Hp // 256) < 160:
        assert got is None
        return
    u, g = got
    ur, gr = ref.gemm_geglu(x, w_il, Hp)
    close(u, ur, rtol=2e-2, atol=2e-2)
    close(g, gr, rtol=3e-2, atol=2e-2)
    # the split layout: u[:, :inner] = x W_x^T, u[:, Hp:Hp + inner] = x W_gate^T, padding columns zero
    close(u[:, :inner], (x.float() @ w[:inner].to(bf).float().t()), rtol=2e-2, atol=2e-2)
    close(u[:, Hp:Hp + inner], (x.float() @ w[inner:].to(bf).float().t()), rtol=2e-2, atol=2e-2)
    assert float(u[:, inner:Hp].abs().max()) == 0.0 and float(g[:, inner:].abs().max()) == 0.0
    # training variant: g only (bit-identical to the launch that also stores u) ...
    u0, g0 = hip.gemm_geglu(x, w_il, Hp, save_u=False)
    assert u0 is None and torch.equal(g0, g)
    # ... and the backward by recomputation against (a) the checker on the unrounded (x, gate), (b) the streaming geglu_bwd on the stored u
    dg = rnd(M, Hp, dtype=bf, seed=3, scale=0.5)
    dg[:, inner:] = 0
    du = hip.gemm_geglu_bwd(x, w_il, dg, Hp)
    close(du, ref.gemm_geglu_bwd(x, w_il, dg, Hp), rtol=3e-2, atol=2e-2)
    close(du, hip.geglu_bwd(dg, u), rtol=5e-2, atol=3e-2)
    assert float(du[:, inner:Hp].abs().max()) == 0.0 and float(du[:, Hp + inner:].abs().max()) == 0.0
    assert torch.equal(du, hip.gemm_geglu_bwd(x, w_il, dg, Hp))


@pytest.mark.parametrize("M,inner,K", [(256 * 40, 1365, 512), (256 * 90, 341, 128), (256 * 14, 1365, 512)])
def test_gemm_dgeglu_fused(hip, ref, M, inner, K):
    """Backward of the feed-forward block between the out-projection and the GEGLU in one launch: du = [dg gelu(gate) | dg x gelu'(gate)]
    with dg = dy W_out in the accumulators only -- against the checker and against the two launches it replaces (GEMM + geglu_bwd);
    Hp = 1408 ends in a half-filled column tile; the last shape does not fill the chip and must be declined."""
    bf = torch.bfloat16
    Hp = (inner + 127) // 128 * 128
    dy = rnd(M, K, dtype=bf, seed=1, scale=0.5)
    wt = torch.zeros(Hp, K, dtype=bf, device=DEV)
    wt[:inner] = rnd(inner, K, seed=2, scale=inner ** -0.5).to(bf)                   # W_out^T, zero rows for the padded features
    u = rnd(M, 2 * Hp, dtype=bf, seed=3)
    u[:, inner:Hp] = 0; u[:, Hp + inner:] = 0
    du = hip.gemm_dgeglu(dy, wt, u)
    if (M // 256) * ((Hp + 255) // 256) < 160:
        assert du is None
        return
    close(du, ref.gemm_dgeglu(dy, wt, u), rtol=3e-2, atol=2e-2)
    close(du, hip.geglu_bwd(hip.gemm(dy, wt), u), rtol=5e-2, atol=3e-2)
    assert float(du[:, inner:Hp].abs().max()) == 0.0 and float(du[:, Hp + inner:].abs().max()) == 0.0
    assert torch.equal(du, hip.gemm_dgeglu(dy, wt, u))


# ---------------------------------------------------------------- the text tower's GEMM sizes (gemm_sm.hip)
@pytest.mark.parametrize("M,N,K", [(1024, 768, 768), (1024, 2304, 768), (1024, 3072, 768), (1024, 768, 3072), (4096, 768, 768), (1000, 776, 192),
                                   (13824, 512, 256), (72, 64, 128)])
def test_gemm_sm_text_tower_shapes_nt(hip, ref, M, N, K):
    """BERT's forward / grad-input GEMMs at M = B * T rows (and the image tower at one volume): bf16 operands with every epilogue the
    mixed-precision text tower uses -- f32 bias, f32 or bf16 residual, f32 or bf16 output, alpha, f32 accumulate -- ragged tiles included."""
    bf = torch.bfloat16
    a, b = rnd(M, K, dtype=bf, seed=1), rnd(N, K, dtype=bf, seed=2, scale=K ** -0.5)
    bias, res32, res16 = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, dtype=bf, seed=5)
    for kw in (dict(), dict(bias=bias), dict(bias=bias, residual=res32, out_dtype=torch.float32), dict(residual=res16),
               dict(out_dtype=torch.float32, alpha=0.5), dict(bias=bias, out_dtype=torch.float32)):
        y, yr = hip.gemm(a, b, **kw), ref.gemm(a, b, **kw)
        assert y.dtype == yr.dtype
        close(y, yr, rtol=2e-2, atol=2e-2)
        assert torch.equal(y, hip.gemm(a, b, **kw))
    acc = rnd(M, N, seed=9)
    out = acc.clone()
    hip.gemm(a, b, out=out, accumulate=True)
    close(out, acc + ref.gemm(a, b, out_dtype=torch.float32), rtol=2e-2, atol=2e-2)
    # strided operands / outputs (column views of wider buffers, as QkvSdpaFn uses them)
    wide = rnd(M, N + 64, dtype=bf, seed=6)
    hip.gemm(a, b, out=wide[:, 32:32 + N])
    close(wide[:, 32:32 + N], ref.gemm(a, b), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("T,M,N", [(1024, 2304, 768), (1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072), (960, 200, 136), (512, 64, 64)])
def test_gemm_sm_text_tower_shapes_tn(hip, ref, T, M, N):
    """dW = dy^T x with the reduction over the T = B * T token rows (k-major operands, transposing LDS reads), accumulated into an f32 slice
    of a wider buffer as functional.weight_grad does; column views of a stacked dq | dk | dv buffer as the A operand."""
    bf = torch.bfloat16
    dywide = rnd(T, M + 128, dtype=bf, seed=1)
    dy = dywide[:, 64:64 + M] if M % 8 == 0 else rnd(T, M, dtype=bf, seed=1)
    x = rnd(T, N, dtype=bf, seed=2)
    want = dy.float().t() @ x.float()
    out = torch.zeros(M, N, device=DEV)
    hip.gemm(dy, x, a_kc=False, b_kc=False, out=out, accumulate=False, split_k=0, M=M, N=N, K=T)
    close(out, want, rtol=2e-2, atol=2e-2 * T ** 0.5)
    base = rnd(M, N + 8, seed=3)
    got = base.clone()
    hip.gemm(dy, x, a_kc=False, b_kc=False, out=got[:, :N], accumulate=True, split_k=0, M=M, N=N, K=T)
    close(got[:, :N], base[:, :N] + want, rtol=2e-2, atol=2e-2 * T ** 0.5)
    assert torch.equal(got[:, N:], base[:, N:])
    again = base.clone()
    hip.gemm(dy, x, a_kc=False, b_kc=False, out=again[:, :N], accumulate=True, split_k=0, M=M, N=N, K=T)
    assert torch.equal(got, again)


@pytest.mark.parametrize("T,n_out,k_in", [(1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072), (4096, 768, 768), (960, 200, 136), (128, 64, 64)])
def test_gemm_dw_db_weight_and_bias_gradient_in_one_launch(hip, ref, T, n_out, k_in):
    """ctclip_gemm_dw_db: dW (+)= dy^T x and db (+)= colsum(dy) from ONE launch (the column sums ride the A fragments of the first column tile's
    workgroups) -- against the composed gemm + colsum, with dy a column view of a stacked dq | dk | dv buffer and dW / db slices of a flat buffer."""
    bf = torch.bfloat16
    wide = rnd(T, n_out + 128, dtype=bf, seed=1)
    dy = wide[:, 64:64 + n_out] if n_out % 8 == 0 else rnd(T, n_out, dtype=bf, seed=1)
    x = rnd(T, k_in, dtype=bf, seed=2)
    flat = rnd(n_out * k_in + n_out + 32, seed=3)
    base = flat.clone()
    dw, db = flat[16:16 + n_out * k_in].view(n_out, k_in), flat[16 + n_out * k_in:16 + n_out * k_in + n_out]
    ok = hip.gemm_dw_db(dy, x, dw, db, accumulate=True)
    if ((n_out + 63) // 64) * ((k_in + 63) // 64) < 8:
        assert not ok          # (fewer than 8 workgroups: declined, the caller composes)
        return
    assert ok
    want_w, want_b = dy.float().t() @ x.float(), dy.float().sum(0)
    bw, bb = base[16:16 + n_out * k_in].view(n_out, k_in), base[16 + n_out * k_in:16 + n_out * k_in + n_out]
    close(dw, bw + want_w, rtol=2e-2, atol=2e-2 * T ** 0.5)
    close(db, bb + want_b, rtol=2e-2, atol=2e-2 * T ** 0.5)
    assert torch.equal(flat[:16], base[:16]) and torch.equal(flat[16 + n_out * k_in + n_out:], base[16 + n_out * k_in + n_out:])
    again = base.clone()
    hip.gemm_dw_db(dy, x, again[16:16 + n_out * k_in].view(n_out, k_in), again[16 + n_out * k_in:16 + n_out * k_in + n_out], accumulate=True)
    assert torch.equal(again, flat)
    # overwrite form
    dw2, db2 = torch.full((n_out, k_in), 7.0, device=DEV), torch.full((n_out,), 7.0, device=DEV)
    assert hip.gemm_dw_db(dy, x, dw2, db2, accumulate=False)
    close(dw2, want_w, rtol=2e-2, atol=2e-2 * T ** 0.5)
    close(db2, want_b, rtol=2e-2, atol=2e-2 * T ** 0.5)


# ---------------------------------------------------------------- second form of the NT GEMM: two 4-wave workgroups per CU (gemm_nt2.hip)
@pytest.mark.parametrize("M,K", [(768 * 40, 512), (768 * 16, 128), (110592, 256)])
def test_gemm_nt2_bit_identical_to_first_form(hip, ref, M, K):
    """csrc/gemm_nt2.hip (192 x 128 tiles, double-buffered stages, a tile's epilogue under the other workgroup's main loop) accumulates every
    output element over the same k-steps in the same order as gemm_nt.hip: plain, residual, GEGLU forward (with and without u) and the
    out-projection grad-input + GEGLU backward are BIT-IDENTICAL between the two forms, and close to the checker."""
    bf = torch.bfloat16
    inner, Hp = 1365, 1408
    x = rnd(M, K, dtype=bf, seed=1)
    w = rnd(512, K, dtype=bf, seed=2, scale=K ** -0.5)
    res = rnd(M, 512, dtype=bf, seed=3)
    w_in = rnd(2 * inner, K, seed=4, scale=K ** -0.5)
    w_il = hip.geglu_weight_interleave(w_in, Hp, bf)
    wt = torch.zeros(Hp, K, dtype=bf, device=DEV)
    wt[:inner] = rnd(inner, K, seed=5, scale=inner ** -0.5).to(bf)
    u_in = rnd(M, 2 * Hp, dtype=bf, seed=6)
    u_in[:, inner:Hp] = 0; u_in[:, Hp + inner:] = 0

    def run():
        out = [hip.gemm(x, w), hip.gemm(x, w, residual=res)]
        u, g = hip.gemm_geglu(x, w_il, Hp)
        out += [u, g, hip.gemm_geglu(x, w_il, Hp, save_u=False)[1], hip.gemm_dgeglu(x, wt, u_in)]
        torch.cuda.synchronize()
        return out
    prev = hip.gemm_nt2_select(0)
    try:
        first = run()
        hip.gemm_nt2_select(7)
        second = run()
        again = run()
    finally:
        hip.gemm_nt2_select(prev)
    names = ["plain", "residual", "geglu u", "geglu g", "geglu g (no u)", "dgeglu"]
    for n, a, b, c in zip(names, first, second, again):
        assert torch.equal(a, b), f"{n}: the two forms differ ({float((a.float() - b.float()).abs().max()):.3e})"
        assert torch.equal(b, c), f"{n}: not reproducible"
    close(second[0], ref.gemm(x, w), rtol=2e-2, atol=2e-2)
    close(second[1], ref.gemm(x, w, residual=res), rtol=2e-2, atol=2e-2)
    close(second[5], ref.gemm_dgeglu(x, wt, u_in), rtol=3e-2, atol=2e-2)


# ---------------------------------------------------------------- compensated residual stream (gemm_nt epilogue family 3, peg_march<.., 2>)
@pytest.mark.parametrize("M,N,K", [(20480, 512, 256), (20480, 512, 1408), (110592, 512, 256), (512, 512, 256)])
def test_gemm_residual_comp(hip, ref, M, N, K):
    """s = a b^T + residual + e in f32 -> (y, e') = (bf16(s), bf16(s - y)): y must be THE bf16 rounding of s (up to f32 summation order
    at rounding ties), y + e' must carry s to ~2^-16, and the launch must be bit-reproducible; the last shape cannot fill the chip and is
    declined."""
    bf = torch.bfloat16
    a, b = rnd(M, K, dtype=bf, seed=1, scale=0.5), rnd(N, K, dtype=bf, seed=2, scale=K ** -0.5)
    res = rnd(M, N, dtype=bf, seed=3, scale=4.0)
    e1 = rnd(M, N, dtype=bf, seed=4, scale=4.0 * 2 ** -9)
    out = hip.gemm_residual_comp(a, b, res, e1)
    if (M // 256) * ((N + 255) // 256) < 160:
        assert out is None
        return
    y, e = out
    s = a.float() @ b.float().t() + res.float() + e1.float()
    yr, er = ref.gemm_residual_comp(a, b, res, e1)
    ulp = s.abs() * 2 ** -8 + 1e-30
    assert float((((y.float() - s).abs() / ulp) > 1.001).float().mean()) < 1e-3      # y is the nearest bf16 of s (|s| 2^-8 >= half a bf16 spacing; ties aside)
    close((y.float() + e.float()), s, rtol=0, atol=float(s.abs().max()) * 2 ** -15)
    assert float((y.float() + e.float() - s).norm() / s.norm()) < 3e-5                # against 2e-3 for the rounded value alone
    assert float((y.float() != yr.float()).float().mean()) < 2e-3                     # ties / summation order only
    y2, e2_ = hip.gemm_residual_comp(a, b, res, e1)
    assert torch.equal(y, y2) and torch.equal(e, e2_)


@pytest.mark.parametrize("shape", [(8, 6, 24, 24, 512), (2, 4, 30, 16, 512), (1, 3, 13, 24, 96), (2, 5, 7, 8, 64)])
@pytest.mark.parametrize("with_e", [False, True])
def test_peg_fwd_comp(hip, ref, shape, with_e):
    """The marching PEG forward on the compensated residual stream: s = x + e_in + conv(x): y + e_out = s to ~2^-16; without e_in, y is
    bit-identical to ctclip_peg_fwd."""
    C, bf = shape[-1], torch.bfloat16
    x = rnd(*shape, dtype=bf, seed=1)
    w, b = rnd(C, 27, seed=3, scale=0.2), rnd(C, seed=4, scale=0.2)
    e_in = rnd(*shape, dtype=bf, seed=5, scale=2 ** -9) if with_e else None
    y, r = hip.peg_fwd_comp(x, w, b, e_in)
    if not with_e:
        assert torch.equal(y, hip.peg_fwd(x, w, b))
    xc = x.float().permute(0, 4, 1, 2, 3)
    s = (torch.nn.functional.conv3d(torch.nn.functional.pad(xc, (1, 1, 1, 1, 2, 0)), w.view(C, 1, 3, 3, 3), b, groups=C).permute(0, 2, 3, 4, 1) + x.float())
    if with_e:
        s = s + e_in.float()
    assert float((y.float() + r.float() - s).norm() / s.norm()) < 5e-5               # (f32 summation order of 27 taps)
    assert float((y.float() - s).norm() / s.norm()) > 5e-4                            # ... against the rounded value alone
    y2, r2 = hip.peg_fwd_comp(x, w, b, e_in)
    assert torch.equal(y, y2) and torch.equal(r, r2)


# ---------------------------------------------------------------- attention operands from the projection GEMM (gemm_nt epilogue family 4)
@pytest.mark.parametrize("M,K", [(110592, 512), (40960, 128), (512, 512)])
def test_gemm_headnorm_equals_gemm_plus_prep(hip, ref, M, K):
    """ctclip_gemm_headnorm (to_q / to_kv with the attention-operand layout written by the epilogue) against the two launches it replaces,
    ctclip_gemm + ctclip_attn2_prep: same arithmetic on the bf16-rounded projection -> equal up to one bf16 ulp where the compilers contract
    the sum of squares differently; v bit for bit.  The last shape cannot fill the chip and is declined."""
    bf = torch.bfloat16
    x, xk = rnd(M, K, dtype=bf, seed=1), rnd(M, K, dtype=bf, seed=2)
    wq, wkv = rnd(256, K, dtype=bf, seed=3, scale=K ** -0.5), rnd(512, K, dtype=bf, seed=4, scale=K ** -0.5)
    qs, ks = torch.rand(32, device=DEV) + 0.5, torch.rand(32, device=DEV) + 0.5
    c = float(np.float32(8.0) * np.float32(1.4426950408889634))
    got_q = hip.gemm_headnorm(x, wq, [(qs, c)])
    got_kv = hip.gemm_headnorm(xk, wkv, [(ks, 1.0), (None, 1.0)])
    if M // 256 < 160:
        assert got_q is None
        return
    (qh, qinv), = got_q
    (kh, kinv), (vh, none) = got_kv
    assert none is None
    q, kv = hip.gemm(x, wq), hip.gemm(xk, wkv)
    qh0, kh0, vh0, qinv0, kinv0 = hip.attn2_prep(q, kv[:, :256], kv[:, 256:], qs, ks, 8.0, 8)
    assert torch.equal(vh, vh0)
    for a, b in ((qinv, qinv0), (kinv, kinv0)):
        close(a, b, rtol=1e-6, atol=0)
    for a, b in ((qh, qh0), (kh, kh0)):
        close(a, b, rtol=2 ** -7, atol=1e-6)                                      # one bf16 ulp
        assert float((a != b).float().mean()) < 1e-3
    r_q, = ref.gemm_headnorm(x, wq, [(qs, c)])
    close(qh, r_q[0], rtol=2e-2, atol=2e-2)
    (qh2, qinv2), = hip.gemm_headnorm(x, wq, [(qs, c)])
    assert torch.equal(qh, qh2) and torch.equal(qinv, qinv2)


# ---------------------------------------------------------------- short-sequence cosine attention (csrc/attn_short.hip)
@pytest.mark.parametrize("nseq,H,L,strided", [(5, 8, 24, False), (3, 2, 32, False), (7, 3, 1, False), (4, 8, 2, True), (300, 8, 24, True),
                                              (2, 1, 9, False), (1100, 8, 24, False)])
def test_attn_short_fwd_bwd(hip, ref, nseq, H, L, strided):
    """One wave per (sequence, head): forward, dq / dk / dv and the learned-scale gradients against torch f32 on the same bf16
    inputs; `strided`: q / kv / do are column slices of wider buffers (leading dimension > H * 32)."""
    M, HD, bf = nseq * L, H * 32, torch.bfloat16
    pad = 64 if strided else 0
    q = rnd(M, HD + pad, dtype=bf, seed=1)[:, :HD]
    kv = rnd(M, 2 * HD + pad, dtype=bf, seed=2)[:, :2 * HD]
    do = rnd(M, HD + pad, dtype=bf, seed=3)[:, :HD]
    qs, ks = 1.0 + 0.3 * rnd(32, seed=4), 1.0 + 0.3 * rnd(32, seed=5)
    assert hip.attn_short_supported(bf, L, 32)
    o, orf = hip.attn_short_fwd(q, kv, qs, ks, nseq, L, H, 8.0), ref.attn_short_fwd(q, kv, qs, ks, nseq, L, H, 8.0)
    close(o, orf, rtol=3e-2, atol=3e-2)
    dqs, dks = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    dqsr, dksr = dqs.clone(), dks.clone()
    dq, dkv = hip.attn_short_bwd(q, kv, qs, ks, do, nseq, L, H, 8.0, dqs, dks)
    dqr, dkvr = ref.attn_short_bwd(q, kv, qs, ks, do, nseq, L, H, 8.0, dqsr, dksr)
    for a, b, what in ((dq, dqr, "dq"), (dkv[:, :HD], dkvr[:, :HD], "dk"), (dkv[:, HD:], dkvr[:, HD:], "dv"), (dqs, dqsr, "dq_scale"), (dks, dksr, "dk_scale")):
        err = float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
        assert err < 2e-2, (what, err)
    # deterministic: the second run reproduces the first bit for bit (fixed summation order of the scale gradients)
    dqs2, dks2 = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    dq2, dkv2 = hip.attn_short_bwd(q, kv, qs, ks, do, nseq, L, H, 8.0, dqs2, dks2)
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2) and torch.equal(dqs, dqs2) and torch.equal(dks, dks2)


# ---------------------------------------------------------------- PEG
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 2, 4, 4, 128), (1, 5, 3, 7, 64), (2, 24, 6, 6, 512),
                                   # LDS-resident marching kernels (bf16, D3 in {8, 16, 24, 32}, C % 32 == 0): 4-row tiles with a ragged
                                   # last tile / 3 channel chunks; 12-row tiles full and ragged; D3 = 32 (weight gradient falls back)
                                   (2, 5, 7, 8, 64), (1, 3, 13, 24, 96), (8, 6, 24, 24, 512), (2, 4, 30, 16, 512), (7, 3, 17, 8, 512),
                                   (13, 2, 12, 32, 512), (1, 1, 1, 8, 32)])
def test_peg_fwd_bwd(hip, ref, dtype, shape):
    C = shape[-1]
    x, dy = rnd(*shape, dtype=dtype, seed=1), rnd(*shape, dtype=dtype, seed=2)
    w, b = rnd(C, 27, seed=3, scale=0.2), rnd(C, seed=4, scale=0.2)
    close(hip.peg_fwd(x, w, b), ref.peg_fwd(x, w, b), **tol(dtype, (1e-4, 1e-4), (3e-2, 3e-2)))
    dw, db = torch.zeros(C, 27, device=DEV), torch.zeros(C, device=DEV)
    dwr, dbr = dw.clone(), db.clone()
    dx = hip.peg_bwd(dy, x, w, dw, db)
    dxr = ref.peg_bwd(dy, x, w, dwr, dbr)
    n = (x.numel() / C) ** 0.5
    close(dx, dxr, **tol(dtype, (1e-4, 1e-4), (3e-2, 3e-2)))
    close(dw, dwr, rtol=1e-3, atol=2e-3 * n); close(db, dbr, rtol=1e-3, atol=2e-3 * n)


# ---------------------------------------------------------------- attention
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nseq,H,L,D,use_bias,use_mask", [(2, 4, 16, 32, True, False), (3, 2, 9, 32, True, False),
                                                          (4, 8, 24, 32, False, False), (2, 8, 576, 32, True, False),
                                                          (2, 12, 128, 64, False, True), (3, 4, 50, 64, False, True),
                                                          (64, 2, 2, 32, False, False),
                                                          # workgroup-shared (LDS) kernels: fast path, ragged length with idle waves, bias
                                                          (2, 4, 256, 32, False, False), (3, 2, 200, 32, False, False),
                                                          (2, 4, 160, 32, True, False), (2, 3, 136, 32, False, True),
                                                          # BERT shape (d_head 64) on the workgroup-shared kernels: four workgroups per head, idle waves, no mask
                                                          (2, 3, 512, 64, False, True), (3, 2, 160, 64, False, True), (2, 2, 64, 64, False, False)])
def test_attention_fwd_bwd(hip, ref, dtype, nseq, H, L, D, use_bias, use_mask):
    M, HD = nseq * L, H * D
    kv = rnd(M, 2 * HD, dtype=dtype, seed=1, scale=0.5)
    q = rnd(M, HD, dtype=dtype, seed=2, scale=0.5)
    k, v = kv[:, :HD], kv[:, HD:]
    bias = rnd(H, L, L, seed=3) if use_bias else None
    mask = None
    if use_mask:
        lens = torch.randint(L // 2, L + 1, (nseq,))
        mask = ((torch.arange(L)[None] >= lens[:, None]).float() * torch.finfo(torch.float32).min).to(DEV)
    scale = 8.0 if D == 32 else 0.125
    if D == 32:   # cosine attention operates on unit vectors
        q = torch.nn.functional.normalize(q.float().view(M, H, D), dim=-1).view(M, HD).to(dtype)
        kv = torch.cat([torch.nn.functional.normalize(k.float().reshape(M, H, D), dim=-1).view(M, HD).to(dtype), v], 1).contiguous()
        k, v = kv[:, :HD], kv[:, HD:]
    vt = hip.head_transpose(v, nseq, H, L, D)
    close(vt, ref.head_transpose(v, nseq, H, L, D), rtol=0, atol=0)
    o, lse = hip.attn_fwd(q, k, vt, bias, mask, nseq, H, L, D, scale)
    orf, lser = ref.attn_fwd(q, k, vt, bias, mask, nseq, H, L, D, scale)
    close(o, orf, **tol(dtype, (1e-4, 1e-5), (2e-2, 2e-2))); close(lse, lser, **tol(dtype, (1e-4, 1e-4), (1e-2, 3e-2)))
    do = rnd(M, HD, dtype=dtype, seed=4)
    qt, kt, dot = (hip.head_transpose(t, nseq, H, L, D) for t in (q, k, do))
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    dk = torch.empty_like(q)
    dbias = torch.zeros_like(bias) if use_bias else None
    hip.attn_bwd(q, k, v, qt, kt, orf.to(dtype), do, dot, lser, bias, mask, dq, dk, dkv[:, HD:], dbias, nseq, H, L, D, scale)
    dqr, dkr, dvr = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    dbr = torch.zeros_like(bias) if use_bias else None
    ref.attn_bwd(q, k, v, qt, kt, orf.to(dtype), do, dot, lser, bias, mask, dqr, dkr, dvr, dbr, nseq, H, L, D, scale)
    t = tol(dtype, (1e-3, 1e-4), (3e-2, 3e-2))
    close(dq, dqr, **t); close(dk, dkr, **t); close(dkv[:, HD:], dvr, **t)
    if use_bias:
        close(dbias, dbr, rtol=t["rtol"], atol=t["atol"] * nseq ** 0.5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nseq,H,gh,gw,D", [(2, 4, 4, 6, 32), (3, 8, 24, 24, 32), (2, 2, 5, 7, 64), (5, 3, 1, 9, 32)])
def test_attention_relative_bias_table(hip, ref, dtype, nseq, H, gh, gw, D):
    """The attention kernels gather the continuous position bias from its (nclass, H) table (LDS) instead of reading the expanded
    (H, L, L) matrix; dbias comes back folded into the table."""
    L, HD = gh * gw, H * D
    M = nseq * L
    q = torch.nn.functional.normalize(rnd(M, HD, dtype=torch.float32, seed=1).reshape(M, H, D), dim=-1).view(M, HD).to(dtype)
    k = torch.nn.functional.normalize(rnd(M, HD, dtype=torch.float32, seed=2).reshape(M, H, D), dim=-1).view(M, HD).to(dtype)
    v, do = rnd(M, HD, dtype=dtype, seed=3, scale=0.5), rnd(M, HD, dtype=dtype, seed=4)
    tab = rnd((2 * gh - 1) * (2 * gw - 1), H, seed=5)
    grid, scale = (gh, gw), 8.0
    vt = hip.head_transpose(v, nseq, H, L, D)
    o, lse = hip.attn_fwd(q, k, vt, tab, None, nseq, H, L, D, scale, bias_grid=grid)
    orf, lser = ref.attn_fwd(q, k, vt, tab, None, nseq, H, L, D, scale, bias_grid=grid)
    close(o, orf, **tol(dtype, (1e-4, 1e-5), (2e-2, 2e-2))); close(lse, lser, **tol(dtype, (1e-4, 1e-4), (1e-2, 3e-2)))
    # and it is the same function as the expanded-matrix path (the fast path works in the log2 domain: last-bit differences)
    o2, _ = hip.attn_fwd(q, k, vt, ref.cpb_expand(tab, gh, gw), None, nseq, H, L, D, scale)
    close(o, o2, **tol(dtype, (1e-5, 1e-6), (1e-2, 4e-3)))
    qt, kt, dot = (hip.head_transpose(t, nseq, H, L, D) for t in (q, k, do))
    dq, dk, dv, dtab = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q), torch.full_like(tab, 7.0)
    hip.attn_bwd(q, k, v, qt, kt, orf.to(dtype), do, dot, lser, tab, None, dq, dk, dv, dtab, nseq, H, L, D, scale, bias_grid=grid)
    dqr, dkr, dvr, dtr = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q), torch.zeros_like(tab)
    ref.attn_bwd(q, k, v, qt, kt, orf.to(dtype), do, dot, lser, tab, None, dqr, dkr, dvr, dtr, nseq, H, L, D, scale, bias_grid=grid)
    t = tol(dtype, (1e-3, 1e-4), (3e-2, 3e-2))
    close(dq, dqr, **t); close(dk, dkr, **t); close(dv, dvr, **t)
    close(dtab, dtr, rtol=t["rtol"], atol=t["atol"] * (nseq * L) ** 0.5)


@pytest.mark.parametrize("dtype", DT)
def test_dropout_is_philox(hip, ref, dtype):
    """nn.Dropout of the BERT tower: the mask is Philox4x32-10(seed, element, stream) -- reproduced bit for bit in torch."""
    x, res = rnd(64, 768, dtype=dtype, seed=1), rnd(64, 768, dtype=dtype, seed=2)
    seed = 0x1234_5678_9ABC_DEF
    for p, sid, r in ((0.1, 0, None), (0.1, 7, res), (0.5, 3, res)):
        y, yr = hip.dropout(x, r, p, seed, sid), ref.dropout(x, r, p, seed, sid)
        if r is None:
            close(y, yr, rtol=0, atol=0)                     # same mask, same single multiply
        else:
            close(y, yr, **tol(dtype, (1e-6, 1e-6), (8e-3, 8e-3)))   # x * m + r is one fused multiply-add on the device
    y = hip.dropout(x, None, 0.1, seed, 0)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01
    assert not torch.equal(y, hip.dropout(x, None, 0.1, seed, 1)) and not torch.equal(y, hip.dropout(x, None, 0.1, seed + 1, 0))
    close(hip.attn_dropout_mask(2, 3, 40, 0.1, seed, DEV), ref.attn_dropout_mask(2, 3, 40, 0.1, seed, DEV), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nseq,H,L,D,use_mask", [(2, 12, 128, 64, True), (3, 4, 50, 64, False), (2, 4, 24, 32, False),
                                                 (2, 3, 512, 64, True), (3, 2, 160, 64, True), (2, 2, 96, 64, False)])
def test_attention_probability_dropout(hip, ref, dtype, nseq, H, L, D, use_mask):
    """HF BertSelfAttention in train mode: dropout on the softmax output, regenerated identically in forward, dQ and dK/dV."""
    M, HD = nseq * L, H * D
    q, k = rnd(M, HD, dtype=dtype, seed=1, scale=0.5), rnd(M, HD, dtype=dtype, seed=2, scale=0.5)
    v, do = rnd(M, HD, dtype=dtype, seed=3, scale=0.5), rnd(M, HD, dtype=dtype, seed=4)
    mask = None
    if use_mask:
        mask = torch.zeros(nseq, L, device=DEV)
        mask[0, L - 5:] = torch.finfo(torch.float32).min
        if L >= 160:
            mask[1, :40] = torch.finfo(torch.float32).min        # left padding: the first key tile of sequence 1 is masked entirely
    drop, scale = (0.1, 987654321012345), D ** -0.5
    vt = hip.head_transpose(v, nseq, H, L, D)
    o, lse = hip.attn_fwd(q, k, vt, None, mask, nseq, H, L, D, scale, dropout=drop)
    orf, lser = ref.attn_fwd(q, k, vt, None, mask, nseq, H, L, D, scale, dropout=drop)
    close(o, orf, **tol(dtype, (1e-4, 1e-5), (2e-2, 2e-2))); close(lse, lser, **tol(dtype, (1e-4, 1e-4), (1e-2, 3e-2)))
    o0, _ = hip.attn_fwd(q, k, vt, None, mask, nseq, H, L, D, scale)
    assert (o.float() - o0.float()).abs().max() > 1e-3            # and it does something
    qt, kt, dot = (hip.head_transpose(t, nseq, H, L, D) for t in (q, k, do))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    hip.attn_bwd(q, k, v, qt, kt, orf.to(dtype), do, dot, lser, None, mask, dq, dk, dv, None, nseq, H, L, D, scale, dropout=drop)
    dqr, dkr, dvr = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    ref.attn_bwd(q, k, v, qt, kt, orf.to(dtype), do, dot, lser, None, mask, dqr, dkr, dvr, None, nseq, H, L, D, scale, dropout=drop)
    t = tol(dtype, (1e-3, 1e-4), (3e-2, 3e-2))
    close(dq, dqr, **t); close(dk, dkr, **t); close(dv, dvr, **t)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("H,D", [(4, 32), (12, 64)])
def test_qk_norm(hip, ref, dtype, H, D):
    M, HD = 200, H * D
    kv = rnd(M, 2 * HD, dtype=dtype, seed=1)
    sv = 1 + 0.1 * rnd(D, seed=2)
    x = kv[:, :HD]
    y, inv = hip.qk_norm_fwd(x, sv, H, D)
    yr, ir = ref.qk_norm_fwd(x, sv, H, D)
    close(y, yr, **tol(dtype, (1e-5, 1e-6), (1e-2, 1e-2))); close(inv, ir, rtol=1e-4, atol=1e-6)
    dy = rnd(M, HD, dtype=dtype, seed=3)
    dkv, dkvr = torch.zeros_like(kv), torch.zeros_like(kv)
    ds, dsr = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    hip.qk_norm_bwd(dy, x, inv, sv, dkv[:, :HD], ds, H, D)
    ref.qk_norm_bwd(dy, x, ir, sv, dkvr[:, :HD], dsr, H, D)
    close(dkv, dkvr, **tol(dtype, (1e-4, 1e-5), (3e-2, 3e-2))); close(ds, dsr, rtol=1e-3, atol=1e-2)


# ---------------------------------------------------------------- streaming kernels
@pytest.mark.parametrize("dtype", DT)
def test_geglu_gelu_leaky(hip, ref, dtype):
    u, dg = rnd(100, 2 * 384, dtype=dtype, seed=1), rnd(100, 384, dtype=dtype, seed=2)
    close(hip.geglu_fwd(u), ref.geglu_fwd(u), **tol(dtype, (1e-5, 1e-6), (2e-2, 2e-2)))
    close(hip.geglu_bwd(dg, u), ref.geglu_bwd(dg, u), **tol(dtype, (1e-4, 1e-5), (2e-2, 2e-2)))
    close(hip.gelu_fwd(u), ref.gelu_fwd(u), **tol(dtype, (1e-5, 1e-6), (2e-2, 2e-2)))
    du = rnd(100, 768, dtype=dtype, seed=3)
    close(hip.gelu_bwd(du, u), ref.gelu_bwd(du, u), **tol(dtype, (1e-4, 1e-5), (2e-2, 2e-2)))
    x, dy = rnd(50, 77, seed=4), rnd(50, 77, seed=5)
    close(hip.leaky_relu_fwd(x, 0.1), ref.leaky_relu_fwd(x, 0.1), rtol=0, atol=0)
    close(hip.leaky_relu_bwd(dy, x, 0.1), ref.leaky_relu_bwd(dy, x, 0.1), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", DT)
def test_colsum_permute_pool_convert(hip, ref, dtype):
    x = rnd(1300, 776, dtype=dtype, seed=1)
    out, outr = torch.ones(768, device=DEV), torch.ones(768, device=DEV)
    hip.colsum(x, out, N=768); ref.colsum(x, outr, N=768)
    close(out, outr, rtol=1e-3, atol=5e-2)
    out, outr = torch.ones(5, device=DEV), torch.ones(5, device=DEV)     # generic path (N not a multiple of 8)
    hip.colsum(x[:, 3:], out, N=5); ref.colsum(x[:, 3:], outr, N=5)
    close(out, outr, rtol=1e-3, atol=5e-2)
    t4 = rnd(3, 5, 7, 64, dtype=dtype, seed=2)
    close(hip.permute0213(t4), ref.permute0213(t4), rtol=0, atol=0)
    t3 = rnd(3, 6, 256, dtype=dtype, seed=3)
    close(hip.pool_fwd(t3), ref.pool_fwd(t3), **tol(dtype, (1e-5, 1e-6), (1e-2, 1e-2)))
    close(hip.pool_bwd(t3[:, 0].contiguous(), 6), ref.pool_bwd(t3[:, 0].contiguous(), 6), **tol(dtype, (1e-6, 1e-7), (1e-2, 1e-2)))
    w, cs = rnd(37, 50, seed=4), rnd(50, seed=5)
    close(hip.convert_pad(w, 40, 64, dtype, colscale=cs), ref.convert_pad(w, 40, 64, dtype, colscale=cs), **tol(dtype, (1e-6, 1e-7), (1e-2, 1e-2)))


@pytest.mark.parametrize("dtype", DT)
def test_transpose2d(hip, ref, dtype):
    x = rnd(300, 1416, dtype=dtype, seed=9)[:, :1408]
    close(hip.transpose2d(x), ref.transpose2d(x), rtol=0, atol=0)
    x = rnd(37, 50, dtype=dtype, seed=10)
    close(hip.transpose2d(x), ref.transpose2d(x), rtol=0, atol=0)


def test_cpb_expand_reduce(hip, ref):
    for gh, gw in [(4, 4), (3, 5), (24, 24)]:
        ncls = (2 * gh - 1) * (2 * gw - 1)
        tab = rnd(ncls, 8, seed=gh)
        close(hip.cpb_expand(tab, gh, gw), ref.cpb_expand(tab, gh, gw), rtol=0, atol=0)
        db = rnd(8, gh * gw, gh * gw, seed=gw)
        close(hip.cpb_reduce(db, gh, gw), ref.cpb_reduce(db, gh, gw), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DT)
def test_bert_embed(hip, ref, dtype):
    V, T, Hd, Bz = 200, 16, 128, 3
    word, pos, typ = rnd(V, Hd, seed=1), rnd(32, Hd, seed=2), rnd(2, Hd, seed=3)
    ids = torch.randint(0, V, (Bz, T)).to(DEV)
    close(hip.bert_embed_fwd(ids, word, pos, typ[0].contiguous(), dtype), ref.bert_embed_fwd(ids, word, pos, typ[0], dtype),
          **tol(dtype, (1e-6, 1e-6), (1e-2, 1e-2)))
    dx = rnd(Bz * T, Hd, dtype=dtype, seed=4)
    outs = [torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)]
    outr = [o.clone() for o in outs]
    hip.bert_embed_bwd(ids, dx, *outs); ref.bert_embed_bwd(ids, dx, *outr)
    for a, b in zip(outs, outr):
        close(a, b, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", DT)
def test_vq_kernels(hip, ref, dtype):
    C, d, M = 512, 128, 1000
    embed = torch.nn.functional.normalize(rnd(C, d, seed=1), dim=-1)
    idx = torch.randint(0, C // 2, (M,)).to(DEV)      # upper half of the codebook unused -> bins == 0 branch
    x = rnd(M, d, seed=2).to(dtype)
    inv = 1.0 / x.float().norm(dim=-1)
    close(hip.vq_gather(embed, idx, dtype), ref.vq_gather(embed, idx, dtype), **tol(dtype, (0, 0), (1e-2, 1e-2)))
    bins, esum = hip.vq_ema(idx, x, inv, None, embed, 0.8)
    br, er = ref.vq_ema(idx, x, inv, None, embed, 0.8)
    close(bins, br, rtol=0, atol=0); close(esum, er, rtol=1e-5, atol=1e-5)
    b2, e2 = hip.vq_ema(idx, x, inv, None, embed, 0.8)
    assert torch.equal(esum, e2) and torch.equal(bins, b2)         # row-order summation: bit-identical from run to run
    cl, em = torch.rand(C, device=DEV), embed.clone()
    clr, emr = cl.clone(), em.clone()
    hip.vq_ema_update(cl, em, bins, esum, 0.8); ref.vq_ema_update(clr, emr, br, er, 0.8)
    close(cl, clr, rtol=1e-5, atol=1e-6); close(em, emr, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- attention, second generation (csrc/attn2.hip)
def _attn2_case(nseq, H, gh, gw, qk_gain, with_tab, seed=0):
    L, D = gh * gw, 32
    M = nseq * L
    q, kv = rnd(M, H * D, dtype=torch.bfloat16, seed=seed + 1), rnd(M, 2 * H * D, dtype=torch.bfloat16, seed=seed + 2)
    qs = (1.0 + 0.2 * rnd(D, seed=seed + 3)) * qk_gain
    ks = (1.0 + 0.2 * rnd(D, seed=seed + 4)) * qk_gain
    tab = rnd((2 * gh - 1) * (2 * gw - 1), H, seed=seed + 5, scale=0.5) if with_tab else None
    return L, D, M, q, kv, qs, ks, tab


@pytest.mark.parametrize("nseq,H,gh,gw,gain,with_tab", [(3, 8, 24, 24, 1.0, True), (2, 2, 8, 8, 1.0, True), (2, 4, 8, 16, 1.0, False),
                                                         (2, 8, 24, 24, 3.5, True), (3, 2, 2, 32, 4.0, False)])
def test_attn2_fwd_bwd(hip, ref, nseq, H, gh, gw, gain, with_tab):
    """gain 1: the bounded-logit path (|q_scale||k_scale| small); gain >= 3.5: 2 c qs ks > 100 -> the online-softmax path."""
    L, D, M, q, kv, qs, ks, tab = _attn2_case(nseq, H, gh, gw, gain, with_tab)
    HD = H * D
    grid = (gh, gw) if with_tab else None
    assert hip.attn2_supported(torch.bfloat16, H, L, D, grid, with_tab)
    outs = hip.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
    outr = ref.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
    for a, b in zip(outs[:3], outr[:3]):
        close(a, b, rtol=1e-2, atol=2e-2 * gain)
    close(outs[3], outr[3], rtol=1e-5, atol=0); close(outs[4], outr[4], rtol=1e-5, atol=0)
    qh, kh, vh, qinv, kinv = outs
    o, lse2 = hip.attn2_fwd(qh, kh, vh, tab, grid, qs, ks, 8.0, nseq, L)
    orf, lser = ref.attn2_fwd(qh, kh, vh, tab, grid, qs, ks, 8.0, nseq, L)
    close(lse2, lser, rtol=1e-4, atol=2e-3)
    close(o, orf, rtol=2e-2, atol=2e-2)
    do = rnd(M, HD, dtype=torch.bfloat16, seed=9)
    dqh, dkh, dvh, dtab = hip.attn2_bwd(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, nseq, L, with_tab)
    rq, rk, rv, rtab = ref.attn2_bwd(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, nseq, L, with_tab)
    for name, a, b in (("dq", dqh, rq), ("dk", dkh, rk), ("dv", dvh, rv)):
        err = (a.float() - b.float()).norm() / b.float().norm()
        assert err < 2e-2, (name, float(err))
        close(a, b, rtol=5e-2, atol=5e-2 * float(b.float().abs().max()))
    if with_tab:
        err = (dtab - rtab).norm() / rtab.norm()
        assert err < 1e-2, float(err)
        again = hip.attn2_bwd(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, nseq, L, True)[3]
        assert torch.equal(dtab, again)                                  # no atomics: bit-identical
    dq, dkv = torch.empty(M, HD, dtype=torch.bfloat16, device=DEV), torch.empty(M, 2 * HD, dtype=torch.bfloat16, device=DEV)
    dqr, dkvr = torch.empty_like(dq), torch.empty_like(dkv)
    dqs, dks, dqsr, dksr = (torch.ones(D, device=DEV) for _ in range(4))
    hip.attn2_unprep(rq, rk, rv, qh, kh, qinv, kinv, qs, ks, 8.0, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks)
    ref.attn2_unprep(rq, rk, rv, qh, kh, qinv, kinv, qs, ks, 8.0, dqr, dkvr[:, :HD], dkvr[:, HD:], dqsr, dksr)
    close(dq, dqr, rtol=3e-2, atol=3e-2 * float(dqr.float().abs().max()))
    close(dkv, dkvr, rtol=3e-2, atol=3e-2 * float(dkvr.float().abs().max()))
    close(dqs, dqsr, rtol=1e-3, atol=1e-3 * float(dqsr.abs().max())); close(dks, dksr, rtol=1e-3, atol=1e-3 * float(dksr.abs().max()))
    dqs2, dks2 = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    hip.attn2_unprep(rq, rk, rv, qh, kh, qinv, kinv, qs, ks, 8.0, dq, dkv[:, :HD], dkv[:, HD:], dqs2, dks2)
    assert torch.equal(dqs, dqs2) and torch.equal(dks, dks2)


@pytest.mark.parametrize("nseq,H,gh,gw,gain,with_tab", [(70, 8, 24, 24, 1.0, True), (9, 8, 24, 24, 4.0, True), (33, 8, 16, 16, 1.0, False), (5, 4, 8, 8, 1.0, True)])
def test_attn2_bwd_tok_equals_bwd_plus_unprep(hip, ref, nseq, H, gh, gw, gain, with_tab):
    """ctclip_attn2_bwd_tok + ctclip_attn2_unprep_q (the slab key pass writes row-major dk / dv and the k_scale partials itself) against the
    pair it replaces, ctclip_attn2_bwd + ctclip_attn2_unprep: dq and dv bit for bit (same accumulators, same rounding), dk to one bf16 ulp, the scale gradients to f32
    summation order, the table gradient bit for bit; persistent runs across items and heads (70 x 8 items on 256 CUs), the unbounded-logit
    path (gain 4) and a grid the slab kernels serve without a table."""
    L, D, M, q, kv, qs, ks, tab = _attn2_case(nseq, H, gh, gw, gain, with_tab, seed=50)
    HD, bf = H * D, torch.bfloat16
    grid = (gh, gw) if with_tab else None
    qh, kh, vh, qinv, kinv = hip.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
    o, lse2 = hip.attn2_fwd(qh, kh, vh, tab, grid, qs, ks, 8.0, nseq, L)
    do = rnd(M, HD, dtype=bf, seed=9)
    dqh, dkh, dvh, dtab0 = hip.attn2_bwd(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, nseq, L, with_tab)
    dq0, dkv0 = torch.empty(M, HD, dtype=bf, device=DEV), torch.empty(M, 2 * HD, dtype=bf, device=DEV)
    dqs0, dks0 = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    hip.attn2_unprep(dqh, dkh, dvh, qh, kh, qinv, kinv, qs, ks, 8.0, dq0, dkv0[:, :HD], dkv0[:, HD:], dqs0, dks0)
    dq, dkv = torch.full_like(dq0, float("nan")), torch.full_like(dkv0, float("nan"))
    dqs, dks = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    res = hip.attn2_bwd_tok(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, qinv, kinv, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks, nseq, L, with_tab)
    if L > 576:
        assert res is None
        return
    dtab, ws = res
    assert ws is None
    assert torch.equal(dq, dq0)
    close(dqs, dqs0, rtol=1e-5, atol=1e-5 * float(dqs0.abs().max()))                  # the q-only un-prep sums its partials in another order
    assert torch.equal(dkv[:, HD:], dkv0[:, HD:])                                     # dv: the same bf16 rounding of the same accumulators
    close(dkv[:, :HD], dkv0[:, :HD], rtol=2 ** -7, atol=1e-6 * float(dkv0.float().abs().max()) + 1e-30)
    assert float((dkv[:, :HD] != dkv0[:, :HD]).float().mean()) < 5e-3
    close(dks, dks0, rtol=1e-4, atol=1e-4 * float(dks0.abs().max()))
    if with_tab:
        assert torch.equal(dtab, dtab0)
    dq2, dkv2, dqs2, dks2 = torch.empty_like(dq), torch.empty_like(dkv), torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    hip.attn2_bwd_tok(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, qinv, kinv, dq2, dkv2[:, :HD], dkv2[:, HD:], dqs2, dks2, nseq, L, with_tab)
    assert torch.equal(dkv, dkv2) and torch.equal(dks, dks2) and torch.equal(dqs, dqs2)      # no atomics: bit-reproducible


@pytest.mark.parametrize("nseq,H,gh,gw,gain,with_tab,want_dtab", [(70, 8, 24, 24, 1.0, True, True), (6, 8, 24, 24, 4.0, True, True), (33, 4, 16, 16, 1.0, False, False),
                                                                   (5, 2, 16, 24, 1.0, True, True), (24, 8, 24, 24, 1.0, True, False), (3, 8, 16, 32, 1.0, True, True)])
def test_attn2_bwd_fused_one_pass(hip, ref, nseq, H, gh, gw, gain, with_tab, want_dtab):
    """ctclip_attn2_bwd_fused (csrc/attn2_bwd1.hip: dq, dk, dv, both scale gradients and the bias-table gradient from ONE sweep over the score
    tiles) against (a) the f32 checker and (b) the three-pass path it replaces (ctclip_attn2_bwd + ctclip_attn2_unprep).  dv is the same sum in
    the same order -> bit for bit when the key block is not split between two waves; dq / dk are f32 sums in another order -> one bf16 ulp on few
    elements; the table gradient is a fixed-point sum (2^-21 of the item's bound per addend).  Twice: bit-reproducible.  Cases: 70 x 8 items on
    256 CUs (persistent runs of 3 items), a 4 x logit gain (the three-pass kernels leave their bounded-logit form there; the one pass has a single form:
    the row's lse2 is subtracted inside the exponent), no table (16 x 16), L = 384 (nkb = 12) and L = 512 (nkb = 16: the
    position stride P is not the plain quotient), no table gradient wanted."""
    L, D, M, q, kv, qs, ks, tab = _attn2_case(nseq, H, gh, gw, gain, with_tab, seed=70)
    HD, bf = H * D, torch.bfloat16
    grid = (gh, gw) if with_tab else None
    qh, kh, vh, qinv, kinv = hip.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
    o, lse2 = hip.attn2_fwd(qh, kh, vh, tab, grid, qs, ks, 8.0, nseq, L)
    do = rnd(M, HD, dtype=bf, seed=9) * 1e-3                                           # gradients are small numbers: the fixed-point scale must adapt
    outs = []
    for be in (hip, ref):
        dq, dkv = torch.full((M, HD), float("nan"), dtype=bf, device=DEV), torch.full((M, 2 * HD), float("nan"), dtype=bf, device=DEV)
        dqs, dks = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
        res = be.attn2_bwd_fused(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, qinv, kinv, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks, nseq, L, want_dtab)
        assert res is not None
        outs.append((dq, dkv, dqs, dks, res[0]))
    (dq, dkv, dqs, dks, dtab), (rq, rkv, rqs, rks, rtab) = outs
    for name, a, b in (("dq", dq, rq), ("dk", dkv[:, :HD], rkv[:, :HD]), ("dv", dkv[:, HD:], rkv[:, HD:])):
        assert torch.isfinite(a.float()).all(), name
        err = (a.float() - b.float()).norm() / b.float().norm()
        assert err < 2e-2, (name, float(err))
    close(dqs, rqs, rtol=2e-2, atol=2e-2 * float((rqs - 1).abs().max())); close(dks, rks, rtol=2e-2, atol=2e-2 * float((rks - 1).abs().max()))
    if want_dtab:
        err = (dtab - rtab).norm() / rtab.norm()
        assert err < 1e-2, float(err)
    else:
        assert dtab is None
    # (b) the three-pass path on the same operands
    dqh, dkh, dvh, dtab0 = hip.attn2_bwd(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, nseq, L, want_dtab)
    dq0, dkv0 = torch.empty(M, HD, dtype=bf, device=DEV), torch.empty(M, 2 * HD, dtype=bf, device=DEV)
    dqs0, dks0 = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    hip.attn2_unprep(dqh, dkh, dvh, qh, kh, qinv, kinv, qs, ks, 8.0, dq0, dkv0[:, :HD], dkv0[:, HD:], dqs0, dks0)
    # The one-pass kernel rounds to bf16 at other points than the three passes (K x probability and the RAW dO slab, against bounded exponentials and
    # the row factor folded into a re-rounded dO): element-wise the two agree to a few bf16 ulps, and against the f32 checker the one pass must not
    # be the worse of the two.
    for name, a, b, r in (("dq", dq, dq0, rq), ("dk", dkv[:, :HD], dkv0[:, :HD], rkv[:, :HD]), ("dv", dkv[:, HD:], dkv0[:, HD:], rkv[:, HD:])):
        close(a, b, rtol=2 ** -5, atol=4e-3 * float(b.float().abs().max()) + 1e-30)
        e1 = float((a.float() - r.float()).norm() / r.float().norm()); e3 = float((b.float() - r.float()).norm() / r.float().norm())
        print(f"[attn2_bwd_fused {nseq}x{H} {gh}x{gw}] {name}: one pass {e1:.3e}, three passes {e3:.3e} against the f32 checker; {float((a != b).float().mean()):.3f} of the elements differ")
        assert e1 <= 1.1 * e3 + 1e-6, (name, e1, e3)
    close(dqs, dqs0, rtol=1e-3, atol=1e-3 * float((dqs0 - 1).abs().max())); close(dks, dks0, rtol=1e-3, atol=1e-3 * float((dks0 - 1).abs().max()))
    if want_dtab:
        err = (dtab - dtab0).norm() / dtab0.norm()
        e1, e3 = float((dtab - rtab).norm() / rtab.norm()), float((dtab0 - rtab).norm() / rtab.norm())
        print(f"[attn2_bwd_fused {nseq}x{H} {gh}x{gw}] table gradient: one pass {e1:.3e}, three passes {e3:.3e} against the f32 checker, {float(err):.3e} between them")
        assert err < 4e-3 and e1 <= max(1.1 * e3, 2e-4), (float(err), e1, e3)      # (2e-4: the fixed-point grid, 2^-21 of the item's bound per addend)
    # bit-reproducible (integer scatter, fixed summation orders)
    dq2, dkv2, dqs2, dks2 = torch.empty_like(dq), torch.empty_like(dkv), torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    res2 = hip.attn2_bwd_fused(qh, kh, vh, tab, grid, qs, ks, 8.0, o, do, lse2, qinv, kinv, dq2, dkv2[:, :HD], dkv2[:, HD:], dqs2, dks2, nseq, L, want_dtab)
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2) and torch.equal(dks, dks2) and torch.equal(dqs, dqs2)
    if want_dtab:
        assert torch.equal(dtab, res2[0])


def test_attn2_bwd_fused_declines_small_and_large_shapes(hip):
    """L < 256 (fewer than eight key blocks: the waves could not be on eight different query tiles) and L > 576 (LDS) go to the three-pass path."""
    for gh, gw in ((8, 8), (8, 16), (32, 32)):
        assert not hip.lib.ctclip_attn2_bwd_fused_supported(4, 8, gh * gw, 32, gh, gw, 1)
    assert hip.lib.ctclip_attn2_bwd_fused_supported(4, 8, 576, 32, 24, 24, 1)
    assert not hip.lib.ctclip_attn2_bwd_fused_supported(4, 8, 576, 64, 24, 24, 1)


def test_attn2_fwd_persistent_runs_across_items_and_heads(hip, ref):
    """The slab-resident forward keeps one workgroup per CU on a run of (head, sequence) items: more items than CUs, and a run
    that crosses a head boundary (the bias table is re-staged)."""
    nseq, H, gh, gw = 70, 8, 24, 24
    L, D, M, q, kv, qs, ks, tab = _attn2_case(nseq, H, gh, gw, 1.0, True, seed=40)
    HD = H * D
    qh, kh, vh, _, _ = hip.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, 8.0, H)
    o, lse2 = hip.attn2_fwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, nseq, L)
    orf, lser = ref.attn2_fwd(qh, kh, vh, tab, (gh, gw), qs, ks, 8.0, nseq, L)
    close(lse2, lser, rtol=1e-4, atol=2e-3)
    close(o, orf, rtol=2e-2, atol=2e-2)


def test_attn2_matches_first_generation_operator(hip):
    """End to end through the autograd layer: the head-planar path and the round-1 path (separate qk-norm / transposes / attention)
    are the same operator (attention.py:145-178) -- outputs and all five gradients."""
    from ct_clip_amd import backend, functional as Fn
    prev = backend.use(hip)
    try:
        nseq, H, gh, gw = 4, 8, 24, 24
        L, D, M, q, kv, qs, ks, tab = _attn2_case(nseq, H, gh, gw, 1.0, True, seed=20)
        res = []
        for fn in (Fn.CosineAttn2Fn, Fn.CosineAttnFn):
            ins = [t.clone().requires_grad_(True) for t in (q, kv, qs, ks, tab)]
            o = fn.apply(ins[0], ins[1], ins[2], ins[3], ins[4], nseq, L, H, D, 8.0, (gh, gw))
            do = rnd(M, H * D, dtype=torch.bfloat16, seed=30)
            o.backward(do)
            res.append([o] + [t.grad for t in ins])
        for a, b in zip(*res):
            err = (a.float() - b.float()).norm() / b.float().norm()
            assert err < 2e-2, float(err)
    finally:
        backend.use(prev)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,d,nseg", [(5000, 512, 8192), (1024, 768, 30522), (3001, 64, 7), (40, 8, 1)])
def test_segment_sum(hip, ref, dtype, M, d, nseg):
    x = rnd(M, d, dtype=dtype, seed=5)
    keys = torch.randint(0, nseg, (M,)).to(DEV)
    sc = rnd(M, seed=6)
    out, outr = torch.ones(nseg, d, device=DEV), torch.ones(nseg, d, device=DEV)
    cnt, cntr = torch.empty(nseg, device=DEV), torch.empty(nseg, device=DEV)
    hip.segment_sum(keys, x, out, nseg, rowscale=sc, counts=cnt, accumulate=True)
    ref.segment_sum(keys, x, outr, nseg, rowscale=sc, counts=cntr, accumulate=True)
    close(cnt, cntr, rtol=0, atol=0); close(out, outr, rtol=1e-4, atol=1e-4)
    out2 = torch.full((nseg, d), 3.0, device=DEV)
    hip.segment_sum(None, x, out2, nseg, key_mod=min(nseg, 5))
    outr2 = torch.empty(nseg, d, device=DEV)
    ref.segment_sum(None, x, outr2, nseg, key_mod=min(nseg, 5))
    close(out2, outr2, rtol=1e-4, atol=1e-4)
    again = torch.empty(nseg, d, device=DEV)
    hip.segment_sum(None, x, again, nseg, key_mod=min(nseg, 5))
    assert torch.equal(out2, again)


def test_vq_split3_search_is_f32_grade(hip, ref):
    """The three-term bf16 expansion reproduces the f32 cosine arg-max (vector_quantize_pytorch searches in f32)."""
    C, d, M = 8192, 512, 20000
    embed = torch.nn.functional.normalize(rnd(C, d, seed=1), dim=-1)
    x = rnd(M, d, seed=2).to(torch.bfloat16)
    xs, inv = hip.l2norm_split3(x, 0)
    es, _ = hip.l2norm_split3(embed, 1)
    xr, invr = ref.l2norm_split3(x, 0)
    assert torch.equal(xs, xr) or (xs.float() - xr.float()).abs().max() < 1e-2
    close(inv, invr, rtol=1e-5, atol=0)
    idx, val = hip.gemm_argmax(xs, es)
    xn = torch.nn.functional.normalize(x.float(), dim=-1)
    full = xn @ embed.t()
    refval, refidx = full.max(dim=-1)
    agree = (idx == refidx).float().mean().item()
    close(val, refval, rtol=0, atol=2e-5)
    assert agree >= 0.9995, agree


@pytest.mark.parametrize("M,C,d", [(20000, 8192, 512), (5000, 2100, 128), (4096, 2560, 64 * 3)])
def test_vq_hilo_search_on_raw_tokens(hip, ref, M, C, d):
    """Round 6: the code search on the RAW bf16 tokens against the unit codebook's (hi, lo) pair (K = 2 d) picks the f32 cosine arg-max like
    the three-term form (K = 3 d) does -- the norm of a row does not move its arg-max and a bf16 token has no low part."""
    embed = torch.nn.functional.normalize(rnd(C, d, seed=1), dim=-1) * (1 + 0.3 * rnd(C, 1, seed=4))      # not unit: the kernel normalises
    x = (rnd(M, d, seed=2) * (0.2 + 3 * rnd(M, 1, seed=3).abs())).to(torch.bfloat16)                          # rows of very different norms
    es, einv = hip.l2norm_split3(embed, 2)
    esr, einvr = ref.l2norm_split3(embed, 2)
    assert es.shape == (C, 2 * d)
    # (the normalisation's last bit may differ from torch's: hi + lo is what the search multiplies)
    close(es[:, :d].float() + es[:, d:].float(), esr[:, :d].float() + esr[:, d:].float(), rtol=0, atol=4e-6)      # (lo is a bf16 too: 2^-17 of the value)
    close(es[:, :d].float() + es[:, d:].float(), torch.nn.functional.normalize(embed, dim=-1), rtol=0, atol=4e-6)
    close(einv, einvr, rtol=1e-5, atol=0)
    close(hip.row_inv_norms(x), ref.row_inv_norms(x), rtol=1e-5, atol=0)
    assert hip.gemm_argmax_hilo_ok(x, C)
    idx, val = hip.gemm_argmax_hilo(x, es)
    full = torch.nn.functional.normalize(x.float(), dim=-1) @ torch.nn.functional.normalize(embed, dim=-1).t()
    refval, refidx = full.max(dim=-1)
    assert (idx == refidx).float().mean().item() >= 0.9995
    close(val * hip.row_inv_norms(x), refval, rtol=0, atol=2e-5)
    # ... and agrees with the three-term search wherever that one agrees with f32
    xs, _ = hip.l2norm_split3(x, 0)
    e3, _ = hip.l2norm_split3(embed, 1)
    idx3, _ = hip.gemm_argmax(xs, e3)
    assert (idx == idx3).float().mean().item() >= 0.9995
    ri, rv = ref.gemm_argmax_hilo(x, es)
    assert (idx == ri).float().mean().item() >= 0.9995
    # ties go to the lowest code: duplicate codes
    e2 = embed.clone(); e2[C // 2] = e2[3]
    es2, _ = hip.l2norm_split3(e2, 2)
    idx2, _ = hip.gemm_argmax_hilo(x, es2)
    assert not (idx2 == C // 2).any()


def test_vq_hilo_declines_small_shapes(hip):
    x = rnd(100, 128, dtype=torch.bfloat16, seed=1)
    assert not hip.gemm_argmax_hilo_ok(x, 64)
    es, _ = hip.l2norm_split3(rnd(64, 128, seed=2), 2)
    with pytest.raises(RuntimeError):
        hip.gemm_argmax_hilo(x, es)


# ---------------------------------------------------------------- CLIP head
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Bm,N,K", [(2, 64, 2048), (8, 512, 4096 + 1024), (11, 32, 1024),
                                    # more than 8 rows in f32: the 24-row kernels (VocabFine: 18 pooled vectors), two launches at 27
                                    (18, 512, 4096 + 1024 + 8), (27, 64, 2048 + 24)])
def test_visual_latent(hip, ref, dtype, Bm, N, K):
    x, w = rnd(Bm, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=0.05)
    close(hip.visual_latent_fwd(x, w), ref.visual_latent_fwd(x, w), **tol(dtype, (1e-4, 1e-3), (1e-2, 5e-2)))
    dy = rnd(Bm, N, seed=3)
    dw, dwr = torch.ones(N, K, device=DEV), torch.ones(N, K, device=DEV)
    dx = hip.visual_latent_bwd(dy, x, w, dw, accumulate=True)
    dxr = ref.visual_latent_bwd(dy, x, w, dwr, accumulate=True)
    close(dx, dxr, **tol(dtype, (1e-4, 1e-4), (3e-2, 3e-2))); close(dw, dwr, rtol=1e-3, atol=1e-3)
    dw2 = torch.full((N, K), 7.0, device=DEV)
    hip.visual_latent_bwd(dy, x, w, dw2, accumulate=False, want_dx=False)
    close(dw2, dwr - 1.0, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("G,Dl", [(2, 64), (8, 512), (64, 512), (3, 32)])
def test_clip_loss(hip, ref, G, Dl):
    tl, il = rnd(G, Dl, seed=1), rnd(G, Dl, seed=2)
    temp = torch.tensor([1.0], device=DEV)
    out, logits, dtl, dil, dtemp = hip.clip_loss(tl, il, temp, want_logits=True)
    outr, lr, dtlr, dilr, dtr = ref.clip_loss(tl, il, temp, want_logits=True)
    close(out, outr, rtol=1e-5, atol=1e-5); close(logits, lr, rtol=1e-4, atol=1e-5)
    close(dtl, dtlr, rtol=1e-3, atol=1e-6); close(dil, dilr, rtol=1e-3, atol=1e-6); close(dtemp, dtr, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_finetune_head_kernels(hip, ref, p):
    x = rnd(16, 512, seed=1)
    y, yr = hip.relu_dropout(x, None, p, 1234567, 3), ref.relu_dropout(x, None, p, 1234567, 3)
    close(y, yr, rtol=0, atol=0)
    dy = rnd(16, 512, seed=2)
    close(hip.relu_dropout(x, dy, p, 1234567, 3), ref.relu_dropout(x, dy, p, 1234567, 3), rtol=0, atol=0)
    logits, tgt, pw = rnd(8, 18, seed=3, scale=3.0), (rnd(8, 18, seed=4) > 0.5).float(), rnd(18, seed=5).abs() * 5 + 1
    (l, dl), (lr, dlr) = hip.bce_logits(logits, tgt, pw), ref.bce_logits(logits, tgt, pw)
    close(l, lr, rtol=1e-5, atol=1e-6); close(dl, dlr, rtol=1e-4, atol=1e-7)
    (l, dl), (lr, dlr) = hip.bce_logits(logits, tgt, None), ref.bce_logits(logits, tgt, None)
    close(l, lr, rtol=1e-5, atol=1e-6); close(dl, dlr, rtol=1e-4, atol=1e-7)
    sims = rnd(6, 2, seed=6, scale=2.0)
    (l, ds), (lr, dsr) = hip.pair_softmax_mse(sims), ref.pair_softmax_mse(sims)
    close(l, lr, rtol=1e-5, atol=1e-7); close(ds, dsr, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("nt,ni", [(2, 1), (3, 3), (1, 4)])
def test_latent_similarity(hip, ref, nt, ni):
    """ct_clip.py:805-807 with broadcasting: forward and the three gradients against torch."""
    t, v = rnd(nt, 512, seed=1), rnd(ni, 512, seed=2)
    temp = torch.tensor([0.7], device=DEV)
    ds = rnd(max(nt, ni), seed=3)
    close(hip.latent_similarity(t, v, temp), ref.latent_similarity(t, v, temp), rtol=1e-5, atol=1e-6)
    for a, b in zip(hip.latent_similarity(t, v, temp, ds), ref.latent_similarity(t, v, temp, ds)):
        close(a, b, rtol=1e-4, atol=1e-7)


def test_adam_weight_decay_mask(hip, ref):
    n = 4096 + 36
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.01)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    mask = (torch.arange((n + 3) // 4) % 3 == 0).to(torch.uint8).to(DEV)
    hip.adam_step(p, g, m, v, 1e-3, 0.9, 0.99, 1e-8, 1, 0.1, None, mask)
    ref.adam_step(pr, g, mr, vr, 1e-3, 0.9, 0.99, 1e-8, 1, 0.1, None, mask)
    close(p, pr, rtol=1e-6, atol=1e-7)


def test_grad_norm_and_adam(hip, ref):
    n = 1_000_003
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.01)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    for step in (1, 2, 3):
        clip = hip.grad_norm_clip(g, 0.5)
        clipr = ref.grad_norm_clip(g, 0.5)
        close(clip, clipr, rtol=1e-5, atol=1e-7)
        hip.adam_step(p, g, m, v, 1.25e-6, 0.9, 0.99, 1e-8, step, 0.0, clip)
        ref.adam_step(pr, g, mr, vr, 1.25e-6, 0.9, 0.99, 1e-8, step, 0.0, clipr)
    close(p, pr, rtol=1e-6, atol=1e-7); close(m, mr, rtol=1e-4, atol=1e-9); close(v, vr, rtol=1e-4, atol=1e-12)
    # step + zero_grad in one pass (ctclip_adam_step_zero_grad): the same update bit for bit, the gradient buffer cleared -- tail elements included
    p2, m2, v2, g2 = p.clone(), m.clone(), v.clone(), g.clone()
    clip = hip.grad_norm_clip(g, 0.5)
    hip.adam_step(p, g, m, v, 1.25e-6, 0.9, 0.99, 1e-8, 4, 0.01, clip)
    hip.adam_step(p2, g2, m2, v2, 1.25e-6, 0.9, 0.99, 1e-8, 4, 0.01, clip, zero_grad=True)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)
    assert float(g2.abs().max()) == 0.0 and float(g.abs().max()) > 0.0


# ---------------------------------------------------------------- batched weight-shadow refresh (csrc/shadow.hip)
def test_batched_shadow_refresh(hip):
    """One launch rebuilds every kind of bf16 GEMM operand from the f32 weights, bit for bit what convert_pad / transpose2d /
    geglu_weight_interleave make one by one (the CPU twin of this test runs the same check against the torch restatement)."""
    from tests.test_host_logic_cpu import check_batched_shadow_refresh
    check_batched_shadow_refresh(torch.device(DEV))


def test_peg_bwd_split_halves(hip):
    """ctclip_peg_bwd with dx = NULL (weight / bias gradient only) and with dw = NULL (grad-input only) give, together, exactly what
    the single call gives (the trainer launches the two halves on different streams)."""
    bf = torch.bfloat16
    x, dy = rnd(2, 24, 24, 24, 512, dtype=bf, seed=1), rnd(2, 24, 24, 24, 512, dtype=bf, seed=2)
    w = rnd(512, 27, seed=3, scale=0.2)
    dw0, db0 = torch.zeros(512, 27, device=DEV), torch.zeros(512, device=DEV)
    dx0 = hip.peg_bwd(dy, x, w, dw0, db0)
    dw1, db1 = torch.zeros(512, 27, device=DEV), torch.zeros(512, device=DEV)
    assert hip.peg_bwd(dy, x, w, dw1, db1, want_dx=False) is None
    dx1 = hip.peg_bwd(dy, x, w, None, None)
    assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1) and torch.equal(db0, db1)


@pytest.mark.parametrize("G,Dl", [(200, 512), (513, 64), (64, 512)])
def test_clip_loss_any_batch(hip, ref, monkeypatch, G, Dl):
    """ct_clip.py:771-901 beyond the single-block kernel's 128 pairs (16 ranks x batch 8): logits and their gradient in global memory,
    f32 GEMMs around ctclip_clip_loss_logits, l2norm backward -- against the checker; at G = 64 the two paths against each other."""
    tl, il = rnd(G, Dl, seed=1), rnd(G, Dl, seed=2)
    temp = torch.tensor([1.3], device=DEV)
    want = ref.clip_loss(tl, il, temp, want_logits=True)
    if G <= 128:
        small = hip.clip_loss(tl, il, temp, want_logits=True)
        monkeypatch.setattr(type(hip), "CLIP_LOSS_ONE_BLOCK", 0)
    got = hip.clip_loss(tl, il, temp, want_logits=True)
    for a, b, name in zip(got, want, ("out", "logits", "dtl", "dil", "dtemp")):
        close(a.reshape(-1), b.reshape(-1), rtol=2e-4, atol=2e-6 if name in ("dtl", "dil") else 2e-5)
    if G <= 128:
        for a, b in zip(got, small):
            close(a.reshape(-1), b.reshape(-1), rtol=2e-4, atol=2e-6)
    out2 = hip.clip_loss(tl, il, temp, want_grads=False)
    assert out2[2] is None and torch.equal(out2[0], got[0])
