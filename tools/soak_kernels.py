"""Bit-reproducibility soak of the hot kernels: every kernel is launched REPS times on the same inputs and a checksum of all its outputs is
compared with the first launch's.  A kernel with a timing-dependent race (a counted vmcnt that is one too generous, a missing barrier, an
LDS-DMA prefetch overtaking a store) shows up as a handful of deviating launches out of thousands; a deterministic kernel never deviates.
Written for DESIGN.md section 8 item 7 (a rare run-to-run difference of the bf16 training step).

usage: python tools/soak_kernels.py [reps] [kernel-name-substring]      shapes = the bench shapes (B = 8 volumes, 12+12 layers irrelevant)
       CTCLIP_LIB=ct_clip_amd/libctclip_<abl>.so python tools/soak_kernels.py ...      (ablation builds of tools/build_ablation.py)
A second stream keeps the chip busy with an unrelated GEMM while the kernel under test runs (SOAK_NOISE=0 switches that off): races hide on
an idle chip."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
only = sys.argv[2] if len(sys.argv) > 2 else ""
be = backend.get()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16


def rnd(*sh, scale=1.0, dtype=bf):
    return ((torch.rand(*sh, device=dev, generator=gen) * 2 - 1) * scale).to(dtype)


def checksum(outs):
    """order-sensitive 64-bit checksum of the raw bits of every output tensor"""
    acc = 0
    for t in outs:
        if t is None:
            continue
        raw = t.detach().contiguous().reshape(-1).view(torch.uint8)
        n = raw.numel() // 8 * 8
        w = raw[:n].view(torch.int64)
        idx = torch.arange(1, w.numel() + 1, device=dev, dtype=torch.int64)
        acc = (acc * 1000003 + int((w * (idx | 1)).sum().item()) + int(raw[n:].to(torch.int64).sum().item())) & 0xFFFFFFFFFFFFFFFF
    return acc


noise_stream = torch.cuda.Stream(device=dev)
na, nb = rnd(8192, 2048), rnd(4096, 2048)


def soak(name, fn):
    if only and only not in name:
        return
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ref = checksum(fn())
    bad = 0
    for i in range(reps):
        if os.environ.get("SOAK_NOISE", "1") != "0":
            with torch.cuda.stream(noise_stream):
                be.gemm(na, nb)
        outs = fn()
        torch.cuda.synchronize()
        if checksum(outs) != ref:
            bad += 1
    print(f"{name:60s} {reps} launches, {bad} deviating", flush=True)


def as_list(x):
    return list(x) if isinstance(x, (tuple, list)) else [x]


M, D = 110592, 512
# ---- GEMMs (csrc/gemm_nt.hip, gemm_tn.hip)
x512, w_q, w_kv, w_ffout = rnd(M, 512), rnd(256, 512, scale=0.05), rnd(512, 512, scale=0.05), rnd(512, 1408, scale=0.03)
res = rnd(M, 512)
g1408 = rnd(M, 1408)
soak("gemm NT to_q (N=256, K=512)", lambda: [be.gemm(x512, w_q)])
soak("gemm NT to_kv (N=512, K=512)", lambda: [be.gemm(x512, w_kv)])
soak("gemm NT ff_out + residual (N=512, K=1408)", lambda: [be.gemm(g1408, w_ffout, residual=res)])
w32 = (torch.rand(2730, 512, device=dev, generator=gen) * 2 - 1) * 0.05
w_il = be.geglu_weight_interleave(w32, 1408, bf)
soak("gemm NT in-projection + GEGLU (u, g)", lambda: as_list(be.gemm_geglu(x512, w_il, 1408)))
u, _ = be.gemm_geglu(x512, w_il, 1408)
wt_out = rnd(1408, 512, scale=0.03)
soak("gemm NT out-projection grad-input + GEGLU backward", lambda: [be.gemm_dgeglu(x512, wt_out, u)])
du = be.gemm_dgeglu(x512, wt_out, u)
w_in_t = rnd(512, 2816, scale=0.02)
soak("gemm NT in-projection grad-input (K=2816)", lambda: [be.gemm(du, w_in_t)])
dw = torch.zeros(2816, 512, device=dev)
soak("gemm TN weight gradient (M=2816, N=512, K=110592)",
     lambda: [be.gemm(du, x512, a_kc=False, b_kc=False, out=dw.zero_(), accumulate=True, split_k=0, M=2816, N=512, K=M)])
# ---- vector-quantiser code search (arg-max epilogue)
xs, es = be.l2norm_split3(rnd(M, 512, dtype=torch.float32), 0)[0], be.l2norm_split3(rnd(8192, 512, dtype=torch.float32), 1)[0]
soak("gemm_argmax (110592 x 8192 x 1536)", lambda: as_list(be.gemm_argmax(xs, es)))
eh = be.l2norm_split3(rnd(8192, 512, dtype=torch.float32), 2)[0]
soak("gemm_argmax_hilo (round 6: raw tokens x (hi, lo) codebook, 110592 x 8192 x 1024)", lambda: as_list(be.gemm_argmax_hilo(x512, eh)))
# ---- LayerNorm, PEG
gamma, beta = torch.rand(D, device=dev) + 0.5, torch.rand(D, device=dev)
soak("layernorm_fwd", lambda: as_list(be.layernorm_fwd(x512, gamma, beta, 1e-5)))
x5, w27, b27 = rnd(8, 24, 24, 24, 512), rnd(512, 27, scale=0.2, dtype=torch.float32), rnd(512, dtype=torch.float32)
soak("peg_fwd", lambda: [be.peg_fwd(x5, w27, b27)])
dw27, db27 = torch.zeros(512, 27, device=dev), torch.zeros(512, device=dev)
soak("peg_bwd (grad-input + weight gradient)", lambda: [be.peg_bwd(x5, x5, w27, dw27.zero_(), db27.zero_()), dw27, db27])
# ---- spatial attention (csrc/attn2*.hip): 192 sequences x 576 tokens, 8 heads x 32, position-bias table 47 x 47
nseq, L, H, Dh = 192, 576, 8, 32
q, kv = rnd(nseq * L, 256), rnd(nseq * L, 512)
qs, ks = torch.rand(Dh, device=dev) + 0.5, torch.rand(Dh, device=dev) + 0.5
tab = rnd(47 * 47, H, scale=0.5, dtype=torch.float32)
qh, kh, vh, qinv, kinv = be.attn2_prep(q, kv[:, :256], kv[:, 256:], qs, ks, 8.0, H)
soak("attn2_prep", lambda: as_list(be.attn2_prep(q, kv[:, :256], kv[:, 256:], qs, ks, 8.0, H)))
soak("attn2_fwd (slab)", lambda: as_list(be.attn2_fwd(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, nseq, L)))
o, lse2 = be.attn2_fwd(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, nseq, L)
do = rnd(nseq * L, 256, scale=0.1)
soak("attn2_bwd (dq, dkv, dbias)", lambda: as_list(be.attn2_bwd(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, o, do, lse2, nseq, L, True)))
dq_t, dkv_t = torch.empty(nseq * L, 256, dtype=bf, device=dev), torch.empty(nseq * L, 512, dtype=bf, device=dev)
dqs2, dks2 = torch.zeros(Dh, device=dev), torch.zeros(Dh, device=dev)
soak("attn2_bwd_tok + unprep_q (round 3: key pass writes row-major dk / dv)",
     lambda: as_list(be.attn2_bwd_tok(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, o, do, lse2, qinv, kinv, dq_t, dkv_t[:, :256], dkv_t[:, 256:], dqs2.zero_(),
                                      dks2.zero_(), nseq, L, True))[:1] + [dq_t, dkv_t, dqs2, dks2])
dq_f, dkv_f = torch.empty(nseq * L, 256, dtype=bf, device=dev), torch.empty(nseq * L, 512, dtype=bf, device=dev)
dqs3, dks3 = torch.zeros(Dh, device=dev), torch.zeros(Dh, device=dev)
soak("attn2_bwd_fused (round 6: four-wave form attn2_bwd2 at L = 576; fixed-point table scatter)",
     lambda: as_list(be.attn2_bwd_fused(qh, kh, vh, tab, (24, 24), qs, ks, 8.0, o, do, lse2, qinv, kinv, dq_f, dkv_f[:, :256], dkv_f[:, 256:], dqs3.zero_(),
                                        dks3.zero_(), nseq, L, True)) + [dq_f, dkv_f, dqs3, dks3])
w_qn, w_kvn = rnd(256, 512, scale=0.05), rnd(512, 512, scale=0.05)
soak("gemm_headnorm to_q (round 3: attention operands from the epilogue)", lambda: [t for pair in be.gemm_headnorm(x512, w_qn, [(qs, 8.0 * 1.4426950408889634)]) for t in pair])
soak("gemm_headnorm to_kv", lambda: [t for pair in be.gemm_headnorm(x512, w_kvn, [(ks, 1.0), (None, 1.0)]) for t in pair])
e5 = rnd(8, 24, 24, 24, 512, scale=0.004)
soak("peg_fwd_comp (round 3: compensated residual stream)", lambda: as_list(be.peg_fwd_comp(x5, w27, b27, e5)))
e2 = rnd(M, 512, scale=0.004)
soak("gemm_residual_comp ff_out (N=512, K=1408)", lambda: as_list(be.gemm_residual_comp(g1408, w_ffout, res, e2)))
# ---- temporal attention (csrc/attn_short.hip): 4608 sequences x 24 tokens
nseq_t, L_t = 4608, 24
soak("attn_short_fwd", lambda: [be.attn_short_fwd(q, kv, qs, ks, nseq_t, L_t, H, 8.0)])
dqs, dks = torch.zeros(Dh, device=dev), torch.zeros(Dh, device=dev)
soak("attn_short_bwd", lambda: as_list(be.attn_short_bwd(q, kv, qs, ks, do, nseq_t, L_t, H, 8.0, dqs.zero_(), dks.zero_())) + [dqs, dks])
# ---- round 5: the text tower's GEMM sizes (csrc/gemm_sm.hip: LDS-DMA ring with counted vmcnt waits, one barrier per k-step) and the second form of the NT GEMM
Mt = 1024
xt, wt768, wt3072, bt = rnd(Mt, 768), rnd(768, 768, scale=0.05), rnd(3072, 768, scale=0.05), rnd(3072, dtype=torch.float32)
rt = rnd(Mt, 3072, dtype=torch.float32)
soak("gemm_sm NT 64x64 (1024 x 768 x 768, bf16 out)", lambda: [be.gemm(xt, wt768)])
soak("gemm_sm NT 128x128 (1024 x 3072 x 768, bias + f32 residual, f32 out)", lambda: [be.gemm(xt, wt3072, bias=bt, residual=rt, out_dtype=torch.float32)])
dyt = rnd(Mt, 3072, scale=0.1)
dwt, dbt = torch.zeros(3072, 768, device=dev), torch.zeros(3072, device=dev)
soak("gemm_sm TN 128x128 (3072 x 768 x 1024, accumulate)",
     lambda: [be.gemm(dyt, xt, a_kc=False, b_kc=False, out=dwt.zero_(), accumulate=True, split_k=0, M=3072, N=768, K=Mt)])
soak("gemm_dw_db (dW + db in one launch, 3072 x 768 x 1024)", lambda: [be.gemm_dw_db(dyt, xt, dwt.zero_(), dbt.zero_(), accumulate=True), dwt, dbt][1:])
dyt7 = rnd(Mt, 768, scale=0.1)
dw7, db7 = torch.zeros(768, 768, device=dev), torch.zeros(768, device=dev)
soak("gemm_dw_db 64x64 (768 x 768 x 1024)", lambda: [be.gemm_dw_db(dyt7, xt, dw7.zero_(), db7.zero_(), accumulate=True), dw7, db7][1:])
prev_mask = be.gemm_nt2_select(7)
soak("gemm_nt2 plain + residual (N=512, K=1408)", lambda: [be.gemm(g1408, w_ffout, residual=res)])
soak("gemm_nt2 in-projection + GEGLU (u, g)", lambda: as_list(be.gemm_geglu(x512, w_il, 1408)))
soak("gemm_nt2 out-projection grad-input + GEGLU backward", lambda: [be.gemm_dgeglu(x512, wt_out, u)])
be.gemm_nt2_select(prev_mask)
# ---- round 6, late: the BERT attention on workgroup-shared LDS tiles (csrc/attn.hip attn64_*: double-buffered ring, one barrier per step), the
# counting-sort placement, the latent projection's new forms, LayerNorm with 8 / 4 rows in flight
nsq, Hb, Lb, Db = 8, 12, 512, 64
qb, kb, vb, dob = (rnd(nsq * Lb, Hb * Db, scale=0.5) for _ in range(4))
maskb = torch.zeros(nsq, Lb, device=dev)
maskb[:, Lb - 37:] = torch.finfo(torch.float32).min
vtb = be.head_transpose(vb, nsq, Hb, Lb, Db)
dropb = (0.1, 1234567)
soak("attn64 fwd (8 x 12 x 512 x 64, key mask, dropout)", lambda: as_list(be.attn_fwd(qb, kb, vtb, None, maskb, nsq, Hb, Lb, Db, 0.125, dropout=dropb)))
ob, lseb = be.attn_fwd(qb, kb, vtb, None, maskb, nsq, Hb, Lb, Db, 0.125, dropout=dropb)
qtb, ktb, dotb = (be.head_transpose(t, nsq, Hb, Lb, Db) for t in (qb, kb, dob))
dqb, dkb, dvb = torch.empty_like(qb), torch.empty_like(qb), torch.empty_like(qb)


def attn64_bwd():
    be.attn_bwd(qb, kb, vb, qtb, ktb, ob, dob, dotb, lseb, None, maskb, dqb, dkb, dvb, None, nsq, Hb, Lb, Db, 0.125, dropout=dropb)
    return [dqb, dkb, dvb]


soak("attn64 bwd (dq, dk / dv)", attn64_bwd)
keys = torch.randint(0, 8192, (M,), device=dev)
sc = rnd(M, dtype=torch.float32)
seg_out, seg_cnt = torch.zeros(8192, 512, device=dev), torch.zeros(8192, device=dev)
soak("segment_sum (110592 rows -> 8192 codes: hist / scan / place / sum)", lambda: [be.segment_sum(keys, x512, seg_out.zero_(), 8192, rowscale=sc, counts=seg_cnt), seg_out, seg_cnt][1:])
xl, wl, dyl = rnd(8, 294912, dtype=torch.float32), rnd(512, 294912, scale=0.01, dtype=torch.float32), rnd(8, 512, dtype=torch.float32)
dwl = torch.zeros(512, 294912, device=dev)
soak("visual_latent_fwd (8 x 294912 -> 512)", lambda: [be.visual_latent_fwd(xl, wl)])
soak("visual_latent_bwd (overwrite)", lambda: [be.visual_latent_bwd(dyl, xl, wl, dwl, accumulate=False), dwl])
xl18, dyl18 = rnd(18, 294912, dtype=torch.float32), rnd(18, 512, dtype=torch.float32)
soak("visual_latent_bwd (18 rows, accumulate)", lambda: [be.visual_latent_bwd(dyl18, xl18, wl, dwl.zero_(), accumulate=True), dwl])
yl, meanl, rstdl = be.layernorm_fwd(x512, gamma, beta, 1e-5)
dgl, dbl = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
soak("layernorm_bwd + two addends", lambda: [be.layernorm_bwd(x512, x512, gamma, meanl, rstdl, dgl.zero_(), dbl.zero_(), x512, x512), dgl, dbl])
