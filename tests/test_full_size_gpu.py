"""BASELINE.json configs[1] geometry (480x480x240, patch 20x20x10, dim 512, 24^3 tokens, 8192 codes, BERT-base, T=128) against outputs
of the REAL reference at that size (oracle/gen_golden.py):
  * tests/golden/full1.pt -- the reference scripts' own depth, 4+4 layers (run_train.py:17-27), B = 2, forward + backward;
  * tests/golden/full2.pt -- the BENCHMARKED stack, 12+12 layers ("24 layers" of BASELINE.json configs[1], bench.py's default), B = 2,
    with the residual stream sampled at every layer boundary.
Weights and inputs are rebuilt from the seeds (the fixtures prove with fingerprints that they are the tensors the reference ran on).
Protocol of SURVEY.md Appendix D:
  * f32 parity mode: tokens / latents / loss <= 1e-4, VQ code agreement >= 99.9 %, gradients by relative Frobenius error;
  * bf16 performance mode with TEACHER-FORCED VQ indices (kernel error without code flips): loss within the north_star bar 1e-3 rel;
  * bf16 free-running: code agreement, loss and BOTH latent cosines bounded at <= 3x the measured values (profiles/r03_full_size_parity.log).
"""
import functools
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import TextBatch, build_model, fingerprint_ok, perturb_1d, synth_inputs  # noqa: E402

DEV = torch.device("cuda", 0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@functools.lru_cache(maxsize=None)
def _load(name):
    g = torch.load(os.path.join(ROOT, "tests", "golden", f"{name}.pt"), weights_only=False)
    c = g["config"]
    clip = build_model(c, None, torch.device("cpu"), torch.float32)
    perturb_1d(clip, c["seed"])
    sd = clip.state_dict()
    bad = [k for k, fp in g["weight_fingerprints"].items()
           if not (k.endswith("position_ids") or k.endswith("token_type_ids")) and not fingerprint_ok(sd[k], fp)]
    assert not bad, f"seeded rebuild differs from the reference weights: {bad[:5]}"
    video, ids, mask = synth_inputs(c)
    assert fingerprint_ok(video, g["video_fingerprint"])
    assert torch.equal(ids, g["input_ids"]) and torch.equal(mask, g["attention_mask"])
    state = {k: v.clone() for k, v in sd.items()}
    g["name"] = name
    return g, clip, state, video, ids, mask


@pytest.fixture(scope="module", params=["full1", "full2"])
def full(request):
    yield _load(request.param)
    _load(request.param)[1].to("cpu")          # one model on the device at a time


@pytest.fixture(scope="module")
def full1():
    return _load("full1")


def sub(rec, mine):
    m = mine.detach().float().cpu()
    return m if rec["full"] else m.reshape(-1)[::rec["stride"]]


def rel_err(rec, mine):
    a, b = sub(rec, mine), rec["value"].float().reshape(-1)
    return float((a.reshape(-1) - b).norm() / (b.norm() + 1e-30))


def prepare(full, dtype, train):
    g, clip, state, video, ids, mask = full
    clip.load_state_dict(state)
    clip.compute_dtype = dtype
    clip.visual_transformer.compute_dtype = dtype
    clip.to(DEV)
    clip.train(train)
    for p in clip.parameters():
        p.grad = None
    clip.visual_transformer.vq.__dict__.pop("teacher_indices", None)
    return g, clip, TextBatch(ids.to(DEV), mask.to(DEV)), video.to(DEV)


def test_f32_full_size_matches_reference(full):
    g, clip, text, video = prepare(full, torch.float32, True)
    loss = clip(text, video, return_loss=True, device=DEV)
    rel = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
    print(f"[{g['name']} f32] loss {float(loss):.6f} reference {float(g['loss']):.6f} rel {rel:.2e}")
    assert rel < 1e-4
    loss.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in clip.parameters() if p.grad is not None))
    assert abs(float(gn) - float(g["grad_norm"])) / float(g["grad_norm"]) < 1e-3
    grads = dict((n, p.grad) for n, p in clip.named_parameters() if p.grad is not None)
    worst, n = ("", 0.0), 0
    for k, rec in g["grads"].items():
        if rec["value"].numel() == 0 or float(rec["norm"]) < 1e-7 * float(g["grad_norm"]):
            continue        # mathematically-zero gradients (a bias in front of a LayerNorm) are rounding noise on both sides
        e = rel_err(rec, grads[k])
        if e > worst[1]:
            worst = (k, e)
        n += 1
    print(f"[{g['name']} f32] {n} gradients, worst relative error {worst[1]:.2e} ({worst[0]})")
    assert n > 150 and worst[1] < 1e-3
    sd = clip.state_dict()
    cs, cs_ref = sd["visual_transformer.vq._codebook.cluster_size"].cpu(), g["vq_after"]["visual_transformer.vq._codebook.cluster_size"]
    ntok = g["vq_indices"].numel()
    assert float((cs - cs_ref).abs().sum()) <= 0.4 * 1e-3 * ntok + 1e-3     # a flipped code moves two bins by (1 - decay) = 0.2 each
    assert rel_err(g["vq_after"]["visual_transformer.vq._codebook.embed"], sd["visual_transformer.vq._codebook.embed"]) < 1e-3
    # eval-mode products, from the ORIGINAL buffers (the train forward above applied the EMA update in place)
    clip.load_state_dict(full[2])
    clip.eval()
    with torch.no_grad():
        tl, il, toks = clip(text, video, return_latents=True, device=DEV)
        ids = clip.visual_transformer(video, return_only_codebook_ids=True)
    agree = (ids.reshape(-1).cpu() == g["eval_vq_indices"].reshape(-1).long()).float().mean().item()
    print(f"[{g['name']} f32] VQ code agreement {agree:.5f}")
    assert agree >= 0.999
    torch.testing.assert_close(tl.cpu(), g["eval_text_latents"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(il.cpu(), g["eval_image_latents"], rtol=1e-3, atol=2e-4)
    assert rel_err(g["eval_tokens"], toks) < 2e-2      # a flipped code replaces a whole 512-vector: bounded by the agreement above


def _bf16_run(full, teacher):
    g, clip, text, video = prepare(full, torch.bfloat16, True)
    if teacher:
        clip.visual_transformer.vq.teacher_indices = g["vq_indices"].long().to(DEV)
    seen = {}
    hook = clip.visual_transformer.vq.register_forward_hook(lambda _m, _i, o: seen.__setitem__("idx", o[1].detach().reshape(-1).cpu()))
    loss = clip(text, video, return_loss=True, device=DEV)
    hook.remove()
    # code agreement of the TRAINING forward (the configuration bench.py times: plain bf16 residual stream) -- the eval forward below runs the
    # compensated stream and agrees better
    train_agree = (seen["idx"] == g["vq_indices"].reshape(-1).long()).float().mean().item()
    rel = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
    loss.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in clip.parameters() if p.grad is not None))
    gn_rel = abs(float(gn) - float(g["grad_norm"])) / float(g["grad_norm"])
    grads = dict((n, p.grad) for n, p in clip.named_parameters() if p.grad is not None)
    cos_min, cos_name = 1.0, ""
    for k, rec in g["grads"].items():
        if rec["value"].numel() < 64 or float(rec["norm"]) < 1e-6 * float(g["grad_norm"]):
            continue
        a, b_ = sub(rec, grads[k]).reshape(-1).double(), rec["value"].reshape(-1).double()
        c = float((a * b_).sum() / (a.norm() * b_.norm() + 1e-300))
        if c < cos_min:
            cos_min, cos_name = c, k
    clip.load_state_dict(full[2])
    clip.eval()
    with torch.no_grad():
        tl, il, _ = clip(text, video, return_latents=True, device=DEV)
        clip.visual_transformer.vq.__dict__.pop("teacher_indices", None)
        ids = clip.visual_transformer(video, return_only_codebook_ids=True)
    agree = (ids.reshape(-1).cpu() == g["eval_vq_indices"].reshape(-1).long()).float().mean().item()
    cos_i = torch.nn.functional.cosine_similarity(il.cpu(), g["eval_image_latents"]).min().item()
    cos_t = torch.nn.functional.cosine_similarity(tl.cpu(), g["eval_text_latents"]).min().item()
    return dict(name=g["name"], rel=rel, gn_rel=gn_rel, agree=agree, train_agree=train_agree, cos_i=cos_i, cos_t=cos_t, grad_cos_min=cos_min,
                grad_cos_name=cos_name)


# Bounds = at most 3x the deviation measured on MI355X (profiles/r03_full_size_parity.log), never above the north_star bar where one exists.
# Measured, teacher-forced: full1 loss 4.78e-4, grad norm 1.73e-2, cosines image 0.999996 / text 0.999949, worst gradient cosine 0.974 (BERT position
# embeddings); full2 (12+12) loss 1.00e-3, grad norm 9.3e-3, same cosines, worst gradient cosine 0.967.  The text tower (12 bf16 BERT layers: cosine
# 0.99995 = 1 % of latent error) carries the teacher-forced loss error; with CTCLIP_TEXT_DTYPE=f32 it is below 1e-3 at both depths (test below).
# Round 5 (default precision policy: text tower f32 stream + bf16 operands, f32 image head), measured (profiles/r05_full_size_parity.log): full1
# loss 1.05e-5, grad norm 2.5e-3, cosines image 0.999999 / text 0.999991, worst gradient cosine 0.9982; full2 (12+12) loss 7.9e-5, grad norm 2.6e-3,
# same cosines, worst gradient cosine 0.9950.  Bounds <= 4x measured: the loss bound of both depths is 3e-4, a third of the north_star bar.
# (full1's loss error is the text tower's bf16 rounding noise: 2.76e-4 with the first-generation BERT attention kernels, 3.14e-4 with the
# LDS-shared ones of round 6 -- whose full2 / full8 errors went DOWN, 4.7e-5 -> 2.0e-5 and 7.2e-5 -> 4.4e-5; the bar is 1e-3)
TEACHER_BOUNDS = {"full1": dict(rel=6e-4, gn_rel=8e-3, cos_i=0.99999, cos_t=0.99997, grad_cos=0.99),
                  "full2": dict(rel=3e-4, gn_rel=8e-3, cos_i=0.99999, cos_t=0.99997, grad_cos=0.985)}
# Free-running: the loss / gradient deviations are dominated by WHICH codes flip (2.1 % at 4+4 layers, 3.6 % at 12+12: a discrete, chaotic
# event -- two builds of round 2 measured 1.15e-4 and 2.03e-3 for the same loss); bounds from the largest values seen.  The agreement itself is
# what the bf16 residual stream allows (profiles/r03_bf16_error_budget.md: 0.968 emulated on the CPU oracle, 0.991 with an f32 residual stream).
# (agreement / latent cosines come from the EVAL forward: compensated residual stream by default -- measured 0.9872 / 0.9878 at 4+4 and
# 0.9841 / 0.9882 at 12+12; with the plain stream 0.9793 / 0.9804 and 0.9640 / 0.9737.)
# train_agree: the TRAINING forward's own code agreement (plain bf16 stream, what bench.py times): measured 0.979 at 4+4 and 0.966 at 12+12
FREE_BOUNDS = {"full1": dict(rel=3e-3, gn_rel=0.3, agree=0.98, train_agree=0.97, cos_i=0.965, cos_t=0.99997),
               "full2": dict(rel=3e-3, gn_rel=0.3, agree=0.975, train_agree=0.955, cos_i=0.965, cos_t=0.99997)}


def test_bf16_full_size_teacher_forced(full):
    """Kernel error alone (the reference's code ids forced): the bf16 loss meets the north_star bar (1e-3 rel) at full size."""
    r = _bf16_run(full, True)
    print(f"[{r['name']} bf16 teacher-forced] loss rel {r['rel']:.2e}, grad-norm rel {r['gn_rel']:.2e}, free-running code agreement {r['agree']:.4f}, "
          f"latent cosine image {r['cos_i']:.6f} text {r['cos_t']:.6f}, worst gradient cosine {r['grad_cos_min']:.5f} ({r['grad_cos_name']})")
    b = TEACHER_BOUNDS[r["name"]]
    assert r["rel"] < b["rel"] and r["gn_rel"] < b["gn_rel"] and r["cos_i"] > b["cos_i"] and r["cos_t"] > b["cos_t"] and r["grad_cos_min"] > b["grad_cos"]


def test_bf16_full_size_free_running(full):
    r = _bf16_run(full, False)
    print(f"[{r['name']} bf16 free-running] loss rel {r['rel']:.2e}, grad-norm rel {r['gn_rel']:.2e}, code agreement eval {r['agree']:.4f} "
          f"training forward {r['train_agree']:.4f}, latent cosine image {r['cos_i']:.6f} text {r['cos_t']:.6f}")
    b = FREE_BOUNDS[r["name"]]
    assert r["train_agree"] >= b["train_agree"]
    assert r["agree"] >= b["agree"] and r["rel"] < b["rel"] and r["gn_rel"] < b["gn_rel"] and r["cos_i"] > b["cos_i"] and r["cos_t"] > b["cos_t"]


def test_bf16_image_tower_with_f32_text_tower_teacher_forced(full):
    """The cheap lever: BERT in f32 (M = B*T rows, ~1 % of the step's FLOPs), image tower in bf16.  The teacher-forced loss then meets the
    north_star bar (1e-3 rel) at both depths with margin."""
    g, clip, text, video = prepare(full, torch.bfloat16, True)
    clip.text_compute_dtype = torch.float32
    try:
        clip.visual_transformer.vq.teacher_indices = g["vq_indices"].long().to(DEV)
        loss = clip(text, video, return_loss=True, device=DEV)
        rel = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
        loss.backward()
        clip.load_state_dict(full[2])
        clip.eval()
        with torch.no_grad():
            tl, il, _ = clip(text, video, return_latents=True, device=DEV)
    finally:
        clip.text_compute_dtype = None
        clip.visual_transformer.vq.__dict__.pop("teacher_indices", None)
    cos_t = torch.nn.functional.cosine_similarity(tl.cpu(), g["eval_text_latents"]).min().item()
    print(f"[{g['name']} bf16 image tower + f32 text tower, teacher-forced] loss rel {rel:.2e}, text latent cosine {cos_t:.7f}")
    assert rel < 1e-3 and cos_t > 0.999999


def test_layer_error_trace(full, monkeypatch):
    """Where does a low-precision run leave the reference?  Relative error of the residual stream at every layer boundary against the
    fixture's samples (full2 only: full1 holds two boundaries): f32, bf16 with the plain bf16 residual stream (training forward) and bf16
    with the compensated stream (inference default), printed as a table (committed under profiles/)."""
    g = full[0]
    if "s1_in" not in g["intermediates"]:
        pytest.skip("fixture without per-layer samples")
    rows = {}
    for dtype, tag, comp in ((torch.float32, "f32", "0"), (torch.bfloat16, "bf16", "0"), (torch.bfloat16, "bf16 comp", "1")):
        monkeypatch.setenv("CTCLIP_RESIDUAL_COMP", comp)
        g, clip, text, video = prepare(full, dtype, False)
        vt = clip.visual_transformer
        errs = {}
        for tr, pre in ((vt.enc_spatial_transformer, "s"), (vt.enc_temporal_transformer, "t")):
            tr.__dict__["layer_tap"] = (lambda i, x, pre=pre: errs.__setitem__(f"{pre}{i}_in", rel_err(g["intermediates"][f"{pre}{i}_in"], x)))
        try:
            with torch.no_grad():
                ids = vt(video, return_only_codebook_ids=True)
        finally:
            for tr in (vt.enc_spatial_transformer, vt.enc_temporal_transformer):
                tr.__dict__.pop("layer_tap", None)
        errs["vq_agreement"] = (ids.reshape(-1).cpu() == g["eval_vq_indices"].reshape(-1).long()).float().mean().item()
        rows[tag] = errs
    keys = [k for k in rows["f32"] if k.endswith("_in")]
    print(f"[{g['name']}] residual-stream relative error per layer boundary (vs the real reference, f32 CPU)")
    print("   boundary        f32        bf16   bf16 + compensated stream")
    for k in keys:
        print(f"   {k:10s} {rows['f32'][k]:10.2e} {rows['bf16'][k]:10.2e} {rows['bf16 comp'][k]:10.2e}")
    print(f"   VQ code agreement: f32 {rows['f32']['vq_agreement']:.5f}, bf16 {rows['bf16']['vq_agreement']:.5f}, "
          f"bf16 + compensated stream {rows['bf16 comp']['vq_agreement']:.5f}")
    assert max(rows["f32"][k] for k in keys) < 1e-4
    assert max(rows["bf16"][k] for k in keys) < 3e-2                       # measured 1.64e-2 at the last boundary
    assert max(rows["bf16 comp"][k] for k in keys) < 1.5e-2                # measured 7.3e-3
    assert rows["bf16 comp"]["vq_agreement"] >= 0.975 > 0.0                # measured 0.9841 (plain stream: 0.9640)


def test_zz_side_stream_backward_is_bit_identical(full1, tmp_path, monkeypatch):
    """Inside the trainer's backward the weight-gradient GEMMs, the PEG weight gradient and the position-bias table gradient run on a
    side stream under the grad-input chain (functional.wgrad_stream_begin).  Same kernels, same order of every sum: the flat gradient
    buffer must be bit-identical to the single-stream backward (CTCLIP_WGRAD_STREAM=0), and so must the loss."""
    import ct_clip_amd
    g, clip, text, video = prepare(full1, torch.bfloat16, True)
    vq = clip.visual_transformer.vq._codebook
    vq0 = (vq.embed.clone(), vq.cluster_size.clone())
    data0 = {id(p): p.data for p in clip.parameters()}
    trainer = ct_clip_amd.CTClipTrainer(clip, num_train_steps=1, batch_size=2, tokenizer=object(), lr=1e-6, train_dataset=[0], evaluate=False,
                                        checkpoint=False, results_folder=str(tmp_path), num_workers=0)
    try:
        out = {"0": [], "1": []}
        for mode in ("0", "1", "0", "1", "1"):
            monkeypatch.setenv("CTCLIP_WGRAD_STREAM", mode)
            trainer.optim.zero_grad()
            vq.embed.copy_(vq0[0]); vq.cluster_size.copy_(vq0[1])      # the forward's EMA update must not carry over
            loss = trainer.forward_backward(video, text)
            torch.cuda.synchronize()
            fg = trainer.optim.flat_grad
            assert float(fg.abs().max()) > 0
            out[mode].append((float(loss.detach()), float(fg.double().sum()), float(fg.double().abs().sum()), int(fg.view(torch.int32).sum())))
        assert len(set(out["0"] + out["1"])) == 1, out      # 5 of 5 runs bit-identical (strict since the attn_short fix)
    finally:      # the module-scoped model goes back to ordinary parameters for whoever uses the fixture next
        for p in clip.parameters():
            p.__dict__.pop("_ctclip_grad_sink", None)
            p.grad = None
            if id(p) in data0:
                p.data = data0[id(p)]
        vq.embed.copy_(vq0[0]); vq.cluster_size.copy_(vq0[1])


def test_zz_batched_shadow_refresh_in_training(full1, tmp_path, monkeypatch):
    """Two optimisation steps at the full geometry with the one-launch weight-shadow refresh after the optimiser step against the lazy
    per-shadow makers: every bf16 GEMM operand of step 2 (plain, padded, GEGLU split / interleaved, stacked q|k|v, transposed) must be
    bit-identical, hence the loss of step 2 and the parameters after it."""
    import ct_clip_amd
    from ct_clip_amd import functional as Fn
    g, clip, text, video = prepare(full1, torch.bfloat16, True)
    vq = clip.visual_transformer.vq._codebook
    vq0 = (vq.embed.clone(), vq.cluster_size.clone())
    data0 = {id(p): p.data for p in clip.parameters()}
    state0 = {k: v.clone() for k, v in clip.state_dict().items()}
    outs = {True: [], False: []}
    try:
        for batched in (True, False, True, False):
            clip.load_state_dict(state0)
            vq.embed.copy_(vq0[0]); vq.cluster_size.copy_(vq0[1])
            monkeypatch.setattr(Fn, "_SHADOW_BATCH", batched)
            Fn.bump_weight_epoch()
            trainer = ct_clip_amd.CTClipTrainer(clip, num_train_steps=2, batch_size=2, tokenizer=object(), lr=1e-4, train_dataset=[0], evaluate=False,
                                                checkpoint=False, results_folder=str(tmp_path), num_workers=0)
            losses = []
            for _ in range(2):
                loss = trainer.forward_backward(video, text)
                trainer.optim.step(trainer.max_grad_norm)
                trainer.optim.zero_grad()
                losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            fp = trainer.optim.flat_param
            assert losses[0] != losses[1]                          # the step changed the weights (lr 1e-4)
            outs[batched].append((losses[0], losses[1], float(fp.double().sum()), int(fp.view(torch.int32).sum())))
            for p in clip.parameters():      # back to ordinary parameters before the next trainer wraps them again
                p.__dict__.pop("_ctclip_grad_sink", None)
                p.grad = None
                p.data = p.data.clone()
        assert len(set(outs[True] + outs[False])) == 1, outs      # 4 of 4 bit-identical
    finally:
        for p in clip.parameters():
            p.__dict__.pop("_ctclip_grad_sink", None)
            p.grad = None
            if id(p) in data0:
                p.data = data0[id(p)]
        clip.load_state_dict(state0)
        vq.embed.copy_(vq0[0]); vq.cluster_size.copy_(vq0[1])


# ---------------------------------------------------------------------------------------------------------------- the shape bench.py times
# tests/golden/full8_fwd.pt: the REAL reference at B = 8, 12+12 layers (110 592 image tokens: 3.4 rounds of GEMM tiles, non-temporal store paths,
# the large-problem dispatch of several kernels), train-mode forward under no_grad -- loss, both latents, logits, all 110 592 code ids and the
# residual stream at four layer boundaries.  B = 2 parity (full2) does not transfer by itself: kernel dispatch depends on the problem size.
@pytest.fixture(scope="module")
def full8():
    yield _load("full8_fwd")
    _load("full8_fwd")[1].to("cpu")


def _bench_shape_forward(full8, dtype, monkeypatch, comp):
    monkeypatch.setenv("CTCLIP_RESIDUAL_COMP", comp)
    g, clip, text, video = prepare(full8, dtype, True)
    vt = clip.visual_transformer
    errs, got = {}, {}
    for tr, pre in ((vt.enc_spatial_transformer, "s"), (vt.enc_temporal_transformer, "t")):
        tr.__dict__["layer_tap"] = (lambda i, x, pre=pre: errs.__setitem__(f"{pre}{i}_in", rel_err(g["intermediates"][f"{pre}{i}_in"], x))
                                    if f"{pre}{i}_in" in g["intermediates"] else None)
    hook = vt.vq.register_forward_hook(lambda _m, _i, o: got.__setitem__("idx", o[1].detach().reshape(-1).cpu()))
    try:
        with torch.no_grad():
            loss = clip(text, video, return_loss=True, device=DEV)
            clip.load_state_dict(full8[2])                       # (the train-mode forward moved the VQ buffers: latents from the ORIGINAL ones)
            clip.to(DEV)
            tl, il, _ = clip(text, video, return_latents=True, device=DEV)
    finally:
        hook.remove()
        for tr in (vt.enc_spatial_transformer, vt.enc_temporal_transformer):
            tr.__dict__.pop("layer_tap", None)
    rel = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
    agree = (got["idx"] == g["vq_indices"].reshape(-1).long()).float().mean().item()
    tr_, ir_ = torch.nn.functional.normalize(g["text_latents_raw"], dim=-1), torch.nn.functional.normalize(g["image_latents_raw"], dim=-1)
    cos_t = torch.nn.functional.cosine_similarity(tl.float().cpu(), tr_).min().item()
    cos_i = torch.nn.functional.cosine_similarity(il.float().cpu(), ir_).min().item()
    logits = (tl.float() @ il.float().t()) * clip.temperature.exp()
    dlog = float((logits.cpu() - g["logits"]).abs().max())
    return dict(rel=rel, agree=agree, cos_t=cos_t, cos_i=cos_i, dlog=dlog, errs=errs, n=int(got["idx"].numel()))


def test_bench_shape_f32_forward_matches_reference(full8, monkeypatch):
    r = _bench_shape_forward(full8, torch.float32, monkeypatch, "0")
    print(f"[full8_fwd f32] loss rel {r['rel']:.2e}, {r['n']} codes agreement {r['agree']:.5f}, latent cosines text {r['cos_t']:.7f} image {r['cos_i']:.7f}, "
          f"max logit difference {r['dlog']:.2e}, residual stream {', '.join(f'{k} {v:.1e}' for k, v in r['errs'].items())}")
    assert r["n"] == 8 * 24 * 24 * 24
    assert r["rel"] < 1e-4 and r["agree"] >= 0.999 and r["cos_t"] > 0.999999 and r["cos_i"] > 0.9995 and r["dlog"] < 2e-3
    assert len(r["errs"]) == 4 and max(r["errs"].values()) < 1e-4


def test_bench_shape_bf16_forward_as_timed(full8, monkeypatch):
    """The EXACT forward configuration bench.py times: bf16, the plain (uncompensated) bf16 residual stream of the training forward, bf16 text
    tower, B = 8.  Free-running (a flipped code is booked as error).  Bounds <= 3x the values measured on MI355X (profiles/r04_full_size_parity.log)."""
    r = _bench_shape_forward(full8, torch.bfloat16, monkeypatch, "0")
    print(f"[full8_fwd bf16 as timed] loss rel {r['rel']:.2e}, code agreement {r['agree']:.4f}, latent cosines text {r['cos_t']:.6f} image {r['cos_i']:.6f}, "
          f"max logit difference {r['dlog']:.2e}, residual stream {', '.join(f'{k} {v:.1e}' for k, v in r['errs'].items())}")
    assert r["rel"] < 3e-3 and r["agree"] >= 0.955 and r["cos_t"] > 0.99997 and r["cos_i"] > 0.96 and r["dlog"] < 0.16
    assert max(r["errs"].values()) < 3e-2


# tests/golden/full8_bwd.pt: EVERY gradient of that step (B = 8, 12+12 layers) from the real reference (oracle/gen_golden.py full8_bwd: the
# reference's own modules, backward accumulated volume by volume because its B = 8 autograd graph does not fit the build container).  This pins
# what only runs at the benchmarked size: the TN split-K weight-gradient GEMMs at K = 110 592 tokens, the one-pass attention backward over 192
# sequences, the large-problem dispatch of the LayerNorm / PEG / column-sum backward kernels.
@functools.lru_cache(maxsize=None)
def _load_bwd():
    gb = torch.load(os.path.join(ROOT, "tests", "golden", "full8_bwd.pt"), weights_only=False)
    assert gb["config"] == _load("full8_fwd")[0]["config"]
    assert float(gb["loss"]) == float(_load("full8_fwd")[0]["loss"])          # same weights, same inputs: the generator reproduced full8_fwd's loss
    return gb


def _grad_table(gb, grads):
    gnorm = float(gb["grad_norm"])
    rows = []
    for k, rec in gb["grads"].items():
        if rec["value"].numel() == 0 or float(rec["norm"]) < 1e-7 * gnorm:
            continue        # mathematically-zero gradients (a bias in front of a LayerNorm) are rounding noise on both sides
        a, b_ = sub(rec, grads[k]).reshape(-1).double(), rec["value"].reshape(-1).double()
        rows.append((k, float((a - b_).norm() / (b_.norm() + 1e-300)), float((a * b_).sum() / (a.norm() * b_.norm() + 1e-300)),
                     float(rec["norm"]) / gnorm))
    return rows


def test_bench_shape_f32_backward_matches_reference(full8):
    gb = _load_bwd()
    g, clip, text, video = prepare(full8, torch.float32, True)
    loss = clip(text, video, return_loss=True, device=DEV)
    rel = abs(float(loss) - float(gb["loss"])) / abs(float(gb["loss"]))
    loss.backward()
    torch.cuda.synchronize()
    grads = dict((n, p.grad) for n, p in clip.named_parameters() if p.grad is not None)
    gn = torch.sqrt(sum((p.double() ** 2).sum() for p in grads.values()))
    gn_rel = abs(float(gn) - float(gb["grad_norm"])) / float(gb["grad_norm"])
    rows = _grad_table(gb, grads)
    worst = max(rows, key=lambda r: r[1])
    print(f"[full8_bwd f32] loss rel {rel:.2e}, grad norm {float(gn):.5f} reference {float(gb['grad_norm']):.5f} rel {gn_rel:.2e}, {len(rows)} gradients, "
          f"worst relative Frobenius error {worst[1]:.2e} ({worst[0]})")
    for p in clip.parameters():
        p.grad = None
    assert rel < 1e-4 and gn_rel < 1e-3
    assert len(rows) > 400 and worst[1] < 1e-3


# bf16, teacher-forced codes (the kernels' error without code flips), default settings = what bench.py times apart from the forced codes.
# Bounds <= 3x the values measured on MI355X (profiles/r05_full_size_parity.log).
def test_bench_shape_bf16_backward_teacher_forced(full8):
    gb = _load_bwd()
    g, clip, text, video = prepare(full8, torch.bfloat16, True)
    clip.visual_transformer.vq.teacher_indices = g["vq_indices"].long().to(DEV)
    try:
        loss = clip(text, video, return_loss=True, device=DEV)
        rel = abs(float(loss) - float(gb["loss"])) / abs(float(gb["loss"]))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        clip.visual_transformer.vq.__dict__.pop("teacher_indices", None)
    grads = dict((n, p.grad) for n, p in clip.named_parameters() if p.grad is not None)
    gn = torch.sqrt(sum((p.double() ** 2).sum() for p in grads.values()))
    gn_rel = abs(float(gn) - float(gb["grad_norm"])) / float(gb["grad_norm"])
    rows = _grad_table(gb, grads)
    big = [r for r in rows if r[3] > 1e-3]          # tensors that carry more than 0.1 % of the gradient norm
    worst_cos = min(big, key=lambda r: r[2])
    worst_fro = max(big, key=lambda r: r[1])
    # norm-weighted mean relative error: sqrt(sum_k |g_k - r_k|^2 / sum_k |r_k|^2) over the sampled entries
    num = sum((r[1] * r[3]) ** 2 for r in rows) ** 0.5
    den = sum(r[3] ** 2 for r in rows) ** 0.5
    print(f"[full8_bwd bf16 teacher-forced] loss rel {rel:.2e}, grad norm rel {gn_rel:.2e}, {len(rows)} gradients ({len(big)} above 0.1 % of the norm): "
          f"norm-weighted relative error {num / den:.2e}, worst relative Frobenius {worst_fro[1]:.2e} ({worst_fro[0]}), worst cosine {worst_cos[2]:.5f} ({worst_cos[0]})")
    for p in clip.parameters():
        p.grad = None
    # measured: loss 7.0e-5, gradient norm 4.6e-4, norm-weighted relative error 4.2e-2, worst cosine 0.9979 (a temporal k_scale)
    assert rel < 3e-4
    assert gn_rel < 2e-3 and num / den < 0.12 and worst_cos[2] > 0.99
