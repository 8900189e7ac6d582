"""Size-independent properties at the sizes `bench.py` times (BASELINE.json configs[1]: batch 8 of 480x480x240 volumes = 110 592 tokens of
dim 512, 12+12 layers, 8 heads x 32, 8 192 codes, 284 M parameters).  The golden fixtures stop at batch 2 (the real reference needs minutes
per step on the CPU); these tests run every hot kernel and the whole forward at the FULL batch and check what must hold at any size:

  * a GEMM by the identity returns its input bit for bit (every one of the 864 / 432 tiles of the persistent kernel, with and without the
    residual epilogue); the head-planar attention-operand epilogue writes unit rows;
  * attention over constant values returns them (the softmax rows sum to one), spatial (192 x 576 with the position-bias table) and temporal
    (4 608 x 24);
  * PEG of a constant field and of an impulse on a tile boundary are their closed forms (causal padding, halo rows, plane order of the march);
  * LayerNorm rows come out with zero mean and unit variance;
  * the vector quantiser maps its own codebook to itself (idempotence);
  * Adam with zero gradient and zero moments leaves 284 M parameters untouched, the two-stage gradient norm of a known vector is exact;
  * the image tower is equivariant under permutations of the batch, BIT FOR BIT (no kernel's arithmetic depends on where a volume sits), and the
    symmetric InfoNCE loss is invariant when the (report, volume) pairs are permuted together;
  * every backward kernel is the ADJOINT of its forward (<F x, dy> = <x, F^T dy>: the dot-product test): PEG, the three products of a Linear
    layer, LayerNorm (whose Jacobian is symmetric), attention in its values, spatial and temporal;
  * one training step (forward + backward into the flat gradient buffer) repeated on the same batch gives the same 284 M gradients bit for bit,
    and the same loss / gradient norm when the pairs are permuted.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda", 0)
BF = torch.bfloat16
M, D = 110592, 512          # tokens of a batch of 8 volumes, model width


@pytest.fixture(scope="module")
def be():
    from ct_clip_amd import backend
    return backend.HipBackend()


def rnd(*shape, seed=0, scale=1.0, dtype=BF):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


# ------------------------------------------------------------------------------------------------------------------ GEMM
def test_gemm_by_the_identity_returns_the_input_in_every_tile(be):
    x, r = rnd(M, D, seed=1), rnd(M, D, seed=2)
    eye = torch.eye(D, device=DEV, dtype=BF)
    assert torch.equal(be.gemm(x, eye), x)                                         # N = 512: 864 tiles of 256 x 256 on 256 CUs
    assert torch.equal(be.gemm(x, eye[:256].contiguous()), x[:, :256])             # N = 256: 432 tiles
    y = be.gemm(x, torch.zeros(D, D, device=DEV, dtype=BF), residual=r)            # the residual epilogue alone
    assert torch.equal(y, r)
    wide = torch.zeros(2816, D, device=DEV, dtype=BF)                              # N = 2816 (the feed-forward width): column j <- input column j % 512
    wide[torch.arange(2816, device=DEV), torch.arange(2816, device=DEV) % D] = 1
    yw = be.gemm(x, wide)
    assert torch.equal(yw[:, 512:1024], x) and torch.equal(yw[:, 2560:], x[:, :256])


def test_attention_operand_epilogue_writes_unit_rows(be):
    """ctclip_gemm_headnorm at the bench size: every (token, head) row of q~ has norm |q_scale-weighted unit vector| -- with q_scale = 1 and
    multiplier 1: norm 1 -- and the plain section is the projection itself, head-planar."""
    x = rnd(M, D, seed=3)
    w = rnd(512, D, seed=4, scale=0.05)
    ones = torch.ones(32, device=DEV)
    out = be.gemm_headnorm(x, w, [(ones, 1.0), (None, 1.0)])
    assert out is not None
    (kh, kinv), (vh, none) = out
    assert none is None and kh.shape == (8, M, 32)
    n = kh.float().norm(dim=-1)
    assert float((n - 1).abs().max()) < 1e-2
    y = be.gemm(x, w)                                                               # (M, 512) token-major
    assert torch.equal(vh.permute(1, 0, 2).reshape(M, 256), y[:, 256:])
    ref_inv = 1.0 / y[:, :256].float().view(M, 8, 32).norm(dim=-1).clamp(min=1e-12)
    torch.testing.assert_close(kinv, ref_inv, rtol=1e-4, atol=0)


# ------------------------------------------------------------------------------------------------------------------ attention
def test_attention_over_constant_values_returns_them(be):
    nseq, L, H = 192, 576, 8
    q, k = rnd(nseq * L, 256, seed=5), rnd(nseq * L, 256, seed=6)
    c = rnd(1, 256, seed=7)
    v = c.expand(nseq * L, 256).contiguous()
    qs, ks = torch.rand(32, device=DEV) + 0.5, torch.rand(32, device=DEV) + 0.5
    tab = rnd(47 * 47, H, seed=8, scale=0.5, dtype=torch.float32)
    qh, kh, vh, _, _ = be.attn2_prep(q, k, v, qs, ks, 8.0, H)
    o, lse2 = be.attn2_fwd(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, nseq, L)
    assert torch.isfinite(lse2).all()
    torch.testing.assert_close(o.float(), v.float(), rtol=1.6e-2, atol=1e-3)
    # temporal phase: 4 608 sequences of 24 frames
    nseq_t, L_t = 4608, 24
    kv = torch.cat([k, v], dim=1).contiguous()
    ot = be.attn_short_fwd(q, kv, qs, ks, nseq_t, L_t, H, 8.0)
    torch.testing.assert_close(ot.float(), v.float(), rtol=1.6e-2, atol=1e-3)


# ------------------------------------------------------------------------------------------------------------------ PEG
def test_peg_constant_field_and_impulse_closed_forms(be):
    B, G, C = 8, 24, 512
    w = rnd(C, 27, seed=9, scale=0.1, dtype=torch.float32)
    b = rnd(C, seed=10, scale=0.1, dtype=torch.float32)
    c = rnd(C, seed=11)
    x = c.view(1, 1, 1, 1, C).expand(B, G, G, G, C).contiguous()
    y = be.peg_fwd(x, w, b).float()
    cf = c.float()
    interior = cf * (1 + w.sum(1)) + b                                              # all 27 taps see the constant
    torch.testing.assert_close(y[:, 2:, 1:-1, 1:-1], interior.view(1, 1, 1, 1, C).expand(B, G - 2, G - 2, G - 2, C), rtol=1e-2, atol=1e-2)
    w3 = w.view(C, 3, 3, 3)
    corner = cf * (1 + w3[:, 2, 1:, 1:].sum((1, 2))) + b                            # first plane, first row, first column: causal / zero padding
    torch.testing.assert_close(y[:, 0, 0, 0], corner.view(1, C).expand(B, C), rtol=1e-2, atol=1e-2)
    last = cf * (1 + w3[:, :, :2, :2].sum((1, 2, 3))) + b                           # last row, last column of an inner plane
    torch.testing.assert_close(y[:, 5, -1, -1], last.view(1, C).expand(B, C), rtol=1e-2, atol=1e-2)
    # an impulse on the boundary between the two 12-row tiles of a workgroup pair, in the last column, batch item 7
    a0, b0, g0 = 10, 11, 23
    x = torch.zeros(B, G, G, G, C, device=DEV, dtype=BF)
    x[7, a0, b0, g0] = 1
    exp = b.view(1, 1, 1, 1, C).expand(B, G, G, G, C).clone()
    exp[7, a0, b0, g0] += 1
    for d1 in range(3):
        for d2 in range(3):
            for d3 in range(3):
                a, bb, gg = a0 + 2 - d1, b0 + 1 - d2, g0 + 1 - d3                   # y[a, b, g] += w[d1, d2, d3] x[a + d1 - 2, b + d2 - 1, g + d3 - 1]
                if 0 <= a < G and 0 <= bb < G and 0 <= gg < G:
                    exp[7, a, bb, gg] += w3[:, d1, d2, d3]
    torch.testing.assert_close(be.peg_fwd(x, w, b).float(), exp, rtol=1e-2, atol=2e-3)


# ------------------------------------------------------------------------------------------------------------------ LayerNorm, VQ, optimiser
def test_layernorm_rows_are_standardised(be):
    x = rnd(M, D, seed=12, scale=3.0) + 5
    y, mean, rstd = be.layernorm_fwd(x.contiguous(), None, None, 1e-5)
    yf = y.float()
    assert float(yf.mean(1).abs().max()) < 2e-2 and float((yf.var(1, unbiased=False) - 1).abs().max()) < 3e-2
    torch.testing.assert_close(mean, x.float().mean(1), rtol=1e-4, atol=1e-4)


def test_vector_quantiser_maps_its_codebook_to_itself(be):
    code = rnd(8192, D, seed=13, dtype=torch.float32)
    xs, es = be.l2norm_split3(code, 0)[0], be.l2norm_split3(code, 1)[0]
    idx, val = be.gemm_argmax(xs, es)
    assert torch.equal(idx, torch.arange(8192, device=DEV)) and float((val - 1).abs().max()) < 1e-4
    # ... and 110 592 tokens that ARE codes get exactly those codes
    pick = torch.randint(0, 8192, (M,), device=DEV)
    idx2, _ = be.gemm_argmax(be.l2norm_split3(code[pick].contiguous(), 0)[0], es)
    assert torch.equal(idx2, pick)


def test_adam_with_zero_gradient_is_the_identity_and_the_norm_of_ones_is_exact(be):
    n = 284_000_000
    p = torch.randn(n, device=DEV)
    p0 = p.clone()
    g, m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    be.adam_step(p, g, m, v, 1.25e-6, 0.9, 0.99, 1e-8, 1, weight_decay=0.0)
    assert torch.equal(p, p0) and not m.any() and not v.any()
    g.fill_(1.0)
    out = be.grad_norm_clip(g, 0.5)
    assert abs(float(out[0]) - n ** 0.5) / n ** 0.5 < 1e-6 and abs(float(out[1]) - 0.5 / (n ** 0.5 + 1e-6)) < 1e-9


# ------------------------------------------------------------------------------------------------------------------ backward = adjoint
def dot(a, b):
    return float((a.double() * b.double()).sum())


def same(a, b, rel):
    assert abs(a - b) <= rel * max(abs(a), abs(b)), (a, b)


def test_backward_kernels_are_the_adjoints_of_their_forwards(be):
    # PEG: F(x) = x + K x (no bias), backward B(dy) = dy + K^T dy
    B, G, C = 8, 24, 512
    w = rnd(C, 27, seed=20, scale=0.1, dtype=torch.float32)
    x5, n5 = rnd(B, G, G, G, C, seed=21), rnd(B, G, G, G, C, seed=22)
    y5 = be.peg_fwd(x5, w, None)
    dy5 = (y5.float() + 0.5 * n5.float()).to(BF)                    # correlated with F(x): the inner products are of the order of |F x|^2
    same(dot(y5, dy5), dot(x5, be.peg_bwd(dy5, x5, w, None, None)), 2e-3)
    # Linear: y = x W^T, dx = dy W, dW = dy^T x
    x, n = rnd(M, D, seed=23), rnd(M, D, seed=24)
    W = rnd(D, D, seed=25, scale=0.05)
    y = be.gemm(x, W)
    dy = (y.float() + 0.5 * n.float()).to(BF)
    ref = dot(y, dy)
    same(ref, dot(x, be.gemm(dy, W.t().contiguous())), 2e-3)
    dW = torch.zeros(D, D, device=DEV)
    be.gemm(dy, x, a_kc=False, b_kc=False, out=dW, accumulate=True, split_k=0, M=D, N=D, K=M)
    same(ref, dot(W, dW), 2e-3)
    # LayerNorm (no affine): dx = J dy with J symmetric
    xl = (rnd(M, D, seed=26, scale=2.0) + 1).contiguous()
    _, mean, rstd = be.layernorm_fwd(xl, None, None, 1e-5)
    u = rnd(M, D, seed=27)
    v = (u.float() + 0.5 * rnd(M, D, seed=28).float()).to(BF)
    same(dot(be.layernorm_bwd(u, xl, None, mean, rstd), v), dot(u, be.layernorm_bwd(v, xl, None, mean, rstd)), 2e-3)
    # attention is linear in its values: <P V, dO> = <V, P^T dO>
    nseq, L, H = 192, 576, 8
    q, k, vv = rnd(nseq * L, 256, seed=29), rnd(nseq * L, 256, seed=30), rnd(nseq * L, 256, seed=31)
    qs, ks = torch.rand(32, device=DEV) + 0.5, torch.rand(32, device=DEV) + 0.5
    tab = rnd(47 * 47, H, seed=32, scale=0.5, dtype=torch.float32)
    qh, kh, vh, _, _ = be.attn2_prep(q, k, vv, qs, ks, 8.0, H)
    o, lse2 = be.attn2_fwd(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, nseq, L)
    do = (o.float() + 0.5 * rnd(nseq * L, 256, seed=33).float() * float(o.float().std())).to(BF)
    dvh = be.attn2_bwd(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, o, do, lse2, nseq, L, True)[2]
    same(dot(o, do), dot(vh, dvh), 5e-3)
    kv = torch.cat([k, vv], dim=1).contiguous()
    ot = be.attn_short_fwd(q, kv, qs, ks, 4608, 24, H, 8.0)
    dot_ = (ot.float() + 0.5 * rnd(nseq * L, 256, seed=34).float() * float(ot.float().std())).to(BF)
    dkv = be.attn_short_bwd(q, kv, qs, ks, dot_, 4608, 24, H, 8.0)[1]
    same(dot(ot, dot_), dot(vv, dkv[:, 256:]), 5e-3)


# ------------------------------------------------------------------------------------------------------------------ the whole forward / step
def test_bench_model_batch_equivariance_and_training_step_repeatability():
    sys.path.insert(0, ROOT)
    import bench
    args = type("A", (), dict(image=480, frames=240, spatial_depth=12, temporal_depth=12, bert_dropout=0.1, batch=8, text_len=128))()
    clip, trainer = bench.build(args, DEV, BF)
    try:
        clip.eval()
        g = torch.Generator(device=DEV).manual_seed(99)
        vol = torch.rand(4, 1, 240, 480, 480, generator=g, device=DEV) * 2 - 1
        video = torch.cat([vol, vol])                                                  # items 4 .. 7 repeat items 0 .. 3
        perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4], device=DEV)
        ids, mask = bench.synth_text(8, 128, torch.Generator().manual_seed(99), DEV)
        with torch.no_grad():
            vit = clip.visual_transformer
            tok, (b, t, h, w) = vit.tokens_before_vq(video)
            tok = tok.view(8, -1, D)
            assert torch.isfinite(tok.float()).all()
            assert torch.equal(tok[:4], tok[4:])                                        # where a volume sits in the batch changes nothing, bit for bit
            tok_p = vit.tokens_before_vq(video[perm].contiguous())[0].view(8, -1, D)
            assert torch.equal(tok_p, tok[perm])
            codes = vit(video, return_only_codebook_ids=True)
            assert torch.equal(codes[:4], codes[4:])
            # batch-size independence: one volume alone (13 824 tokens: the small-problem dispatch of several kernels -- unfused attention
            # operands, other GEMM tile counts) against the same volume inside the batch of 8: the same function up to bf16 evaluation order
            tok1 = vit.tokens_before_vq(video[:1].contiguous())[0].view(-1, D).float()
            rel1 = float((tok1 - tok[0].float()).norm() / tok[0].float().norm())
            agree1 = float((vit(video[:1].contiguous(), return_only_codebook_ids=True)[0] == codes[0]).float().mean())
            print(f"[bench size] one volume alone against the batch of 8: token rel. difference {rel1:.2e}, code agreement {agree1:.4f}")
            assert rel1 < 3e-2 and agree1 > 0.93
            # symmetric InfoNCE over (report, volume) pairs: distinct volumes, pairs permuted together
            video2 = torch.rand(8, 1, 240, 480, 480, generator=g, device=DEV) * 2 - 1
            l0 = float(clip(bench.Text(ids, mask), video2, return_loss=True, device=DEV))
            l1 = float(clip(bench.Text(ids[perm].contiguous(), mask[perm].contiguous()), video2[perm].contiguous(), return_loss=True, device=DEV))
            assert l0 == l0 and abs(l0 - l1) <= 1e-5 * abs(l0)
        # one training step's forward + backward at the bench batch (eval mode: no dropout, no EMA update -- the only run-to-run inputs): the
        # 284 M gradients of the flat buffer repeat bit for bit, loss and gradient norm survive a permutation of the pairs
        from ct_clip_amd import functional as Fn
        from ct_clip_amd import backend

        def step(text, vid):
            trainer.optim.zero_grad()
            loss = trainer.forward_backward(vid, text)
            Fn.join_side_streams()
            torch.cuda.synchronize()
            flat = trainer.optim.flat_grad
            return float(loss.detach()), float(backend.get().grad_norm_clip(flat, 0.5)[0]), flat.clone()

        la, na, ga = step(bench.Text(ids, mask), video2)
        lb, nb, gb = step(bench.Text(ids, mask), video2)
        assert la == lb and torch.equal(ga, gb) and na > 0 and torch.isfinite(ga).all()
        del gb
        lc, nc, gc = step(bench.Text(ids[perm].contiguous(), mask[perm].contiguous()), video2[perm].contiguous())
        assert abs(la - lc) <= 1e-5 * abs(la) and abs(na - nc) <= 2e-3 * na
        assert float((ga - gc).norm() / ga.norm()) < 2e-2               # (bf16 activations: the reductions over the batch run in another order)
    finally:
        trainer.close()
        del clip, trainer
        torch.cuda.empty_cache()
