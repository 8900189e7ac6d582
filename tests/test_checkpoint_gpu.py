"""SURVEY.md section 8(f) rank 4 on the device: latent export (scripts/forward_data.py:131-148), trainer checkpoint round trip
(scripts/CTCLIPTrainer.py:205-223) and the `module.`-prefixed model checkpoints the reference trainer writes from a DDP-wrapped model
(CTCLIPTrainer.py:331-337, `get_state_dict(..., unwrap=False)`) into CTCLIP.load (ct_clip.py:593-597)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import TextBatch, build_model  # noqa: E402
from tests.test_zero_shot_cpu import StubTokenizer  # noqa: E402

DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_latent_export_equals_return_latents_on_device(golden, tmp_path, dtype):
    """text/<acc>.npz and image/<acc>.npz hold exactly CTCLIP.forward(return_latents=True)[0:2] of the HIP towers; f32 latents also
    match the real reference's eval latents of the golden fixture."""
    from ct_clip_amd.forward_data import CTClipInference as Export
    g = golden("tiny")
    c = g["config"]
    clip = build_model(c, g["state_dict"], DEV, dtype).eval()
    tok = StubTokenizer(c["vocab"], c["T"])
    ds = [(g["video"][i], f"report number {i} no acute findings", torch.zeros(1, 2), f"case_{i}") for i in range(g["video"].shape[0])]
    Export(clip, results_folder=str(tmp_path / "lat"), dataset=ds, tokenizer=tok, max_text_len=c["T"]).infer()
    for i, (vol, text, _, name) in enumerate(ds):
        with torch.no_grad():
            tl, il, _ = clip(tok([text], max_length=c["T"]).to(DEV), vol[None].to(DEV), device=DEV, return_latents=True)
        np.testing.assert_array_equal(np.load(tmp_path / "lat" / "text" / f"{name}.npz")["arr"], tl.float().cpu().numpy())
        np.testing.assert_array_equal(np.load(tmp_path / "lat" / "image" / f"{name}.npz")["arr"], il.float().cpu().numpy())
        if dtype == torch.float32:       # the image latent does not depend on the text: pinned by the real reference's eval latents
            np.testing.assert_allclose(il.cpu().numpy(), g["eval_image_latents"][i:i + 1].numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trainer_checkpoint_round_trip_is_bit_exact(golden, tmp_path, dtype):
    """Step 1, save, step 2 -- against: fresh model + fresh trainer, load, step 2.  Loss of step 2, every parameter, both Adam moments,
    the VQ buffers and the `steps` buffer must be bit-identical (the flat f32 state and the bf16 weight shadows are restored exactly)."""
    import ct_clip_amd
    g = golden("tiny")
    c = g["config"]
    text = TextBatch(g["input_ids"].to(DEV), g["attention_mask"].to(DEV))
    video = g["video"].to(DEV)

    def make(state):
        clip = build_model(c, state, DEV, dtype)
        clip.train()
        return clip, ct_clip_amd.CTClipTrainer(clip, num_train_steps=3, batch_size=2, tokenizer=object(), lr=1e-3, train_dataset=[0], evaluate=False,
                                               checkpoint=False, results_folder=str(tmp_path / "res"), num_workers=0)

    def one_step(tr):
        loss = tr.forward_backward(video, text)
        tr.optim.step(tr.max_grad_norm)
        tr.optim.zero_grad()
        tr.steps += 1
        return float(loss.detach())

    clip_a, tr_a = make(g["state_dict"])
    l1 = one_step(tr_a)
    ck = str(tmp_path / "ck.pt")
    tr_a.save(ck)
    l2 = one_step(tr_a)
    torch.cuda.synchronize()
    ref = {k: v.detach().clone() for k, v in clip_a.state_dict().items()}
    ref_m, ref_v = tr_a.optim.exp_avg.clone(), tr_a.optim.exp_avg_sq.clone()

    perturbed = {k: (v + 0.01 if v.is_floating_point() else v) for k, v in g["state_dict"].items()}      # NOT the checkpoint's weights
    clip_b, tr_b = make(perturbed)
    tr_b.load(ck)
    assert int(tr_b.steps.item()) == 1 and tr_b.optim.step_count == 1
    l2b = one_step(tr_b)
    torch.cuda.synchronize()
    assert l2b == l2 and l2 != l1
    sd_b = clip_b.state_dict()
    for k, v in ref.items():
        assert torch.equal(sd_b[k], v), k
    assert torch.equal(tr_b.optim.exp_avg, ref_m) and torch.equal(tr_b.optim.exp_avg_sq, ref_v)
    assert int(tr_b.steps.item()) == 2


def test_ctclip_load_accepts_the_ddp_prefixed_checkpoint(golden, tmp_path):
    g = golden("tiny")
    c = g["config"]
    clip = build_model(c, g["state_dict"], DEV, torch.float32).eval()
    plain, wrapped = str(tmp_path / "plain.pt"), str(tmp_path / "module.pt")
    sd = {k: v.cpu() for k, v in clip.state_dict().items()}
    torch.save(sd, plain)
    torch.save({"module." + k: v for k, v in sd.items()}, wrapped)            # CTCLIPTrainer.py:331-337 under DDP
    text = TextBatch(g["input_ids"].to(DEV), g["attention_mask"].to(DEV))
    with torch.no_grad():
        want = clip(text, g["video"].to(DEV), device=DEV, return_latents=True)[1]
    for path in (plain, wrapped):
        other = build_model(c, None, DEV, torch.float32).eval()
        other.load(path)
        with torch.no_grad():
            got = other(text, g["video"].to(DEV), device=DEV, return_latents=True)[1]
        assert torch.equal(got, want), path


def test_graphed_step_replays_the_eager_step_bit_for_bit(golden, tmp_path):
    """ct_clip_amd.trainer.GraphedStep: the optimisation step captured once into a hipGraph.  With dropout off the step is a pure function of
    (weights, moments, step count, batch): three replays after three eager warm-up steps leave the flat f32 parameters, both Adam moments and the
    VQ buffers BIT-IDENTICAL to six eager steps (the step count that feeds Adam's bias correction lives on the device and advances inside the
    graph).  With dropout on, consecutive replays draw different masks (the seed offset advances too): the losses differ."""
    import ct_clip_amd
    from ct_clip_amd.trainer import GraphedStep
    g = golden("tiny")
    c = g["config"]
    text = TextBatch(g["input_ids"].to(DEV), g["attention_mask"].to(DEV))
    video = g["video"].to(DEV)

    def run(n_eager, n_graph, dropout, lr=1e-3):
        clip = build_model(c, g["state_dict"], DEV, torch.bfloat16)
        clip.text_transformer.config.hidden_dropout_prob = dropout
        clip.text_transformer.config.attention_probs_dropout_prob = dropout
        clip.train()
        tr = ct_clip_amd.CTClipTrainer(clip, num_train_steps=10, batch_size=2, tokenizer=object(), lr=lr, train_dataset=[0], evaluate=False,
                                       checkpoint=False, results_folder=str(tmp_path / f"res{n_graph}{dropout}"), num_workers=0)
        torch.manual_seed(11)
        losses = []
        for _ in range(n_eager):
            loss = tr.forward_backward(video, text)
            tr.optim.step(tr.max_grad_norm)
            tr.optim.zero_grad()
            losses.append(float(loss))
        if n_graph:
            gs = GraphedStep(tr).capture(video, text)
            try:
                for _ in range(n_graph):
                    losses.append(float(gs.run()))
            finally:
                gs.close()
        torch.cuda.synchronize()
        sd = clip.state_dict()
        return (losses, tr.optim.flat_param.clone(), tr.optim.exp_avg.clone(), tr.optim.exp_avg_sq.clone(),
                {k: v.clone() for k, v in sd.items() if "_codebook" in k}, tr.optim.step_count)
    la, pa, ma, va, qa, na = run(6, 0, 0.0)
    lb, pb, mb, vb, qb, nb = run(3, 3, 0.0)
    assert na == nb == 6
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert all(torch.equal(qa[k], qb[k]) for k in qa)
    lc = run(3, 3, 0.1)[0]
    assert len(set(lc[3:])) == 3, lc          # fresh dropout masks per replay (and a moving model)


def test_graphed_step_mixed_with_eager_forwards_and_failed_capture(golden, tmp_path, monkeypatch):
    """ADVICE r04.  (1) An eager forward BETWEEN replays (the trainer's own validation / zero-shot) must see the weights the replay just wrote:
    shadows without a batched-refresh recipe (the stacked q|k|v bias of the text tower, ...) are rebuilt from the host-side weight epoch, which
    GraphedStep.run() now advances -- replay, eval, replay, eval gives the eval results of the eager sequence bit for bit.  (2) A capture that
    raises leaves no trace: the library's step-state pointer is cleared and the optimiser's step count restored, so the eager steps that follow
    are the steps of a process that never tried (bit-identical parameters)."""
    import ct_clip_amd
    from ct_clip_amd.trainer import GraphedStep
    g = golden("tiny")
    c = g["config"]
    text = TextBatch(g["input_ids"].to(DEV), g["attention_mask"].to(DEV))
    video = g["video"].to(DEV)

    def trainer(tag):
        clip = build_model(c, g["state_dict"], DEV, torch.bfloat16)
        clip.train()
        return clip, ct_clip_amd.CTClipTrainer(clip, num_train_steps=10, batch_size=2, tokenizer=object(), lr=1e-3, train_dataset=[0], evaluate=False,
                                               checkpoint=False, results_folder=str(tmp_path / tag), num_workers=0)

    def eager(tr):
        loss = tr.forward_backward(video, text)
        tr.optim.step(tr.max_grad_norm)
        tr.optim.zero_grad()
        return loss

    def evaluate(clip):
        clip.eval()
        with torch.no_grad():
            tl, il, _ = clip(text, video, return_latents=True, device=DEV)
        clip.train()
        return tl.clone(), il.clone()

    # (1)
    outs = {}
    for mode in ("eager", "graph"):
        clip, tr = trainer(mode)
        for _ in range(3):
            eager(tr)
        evals = []
        gs = GraphedStep(tr).capture(video, text) if mode == "graph" else None
        try:
            for _ in range(3):
                gs.run() if gs is not None else eager(tr)
                evals.append(evaluate(clip))
        finally:
            if gs is not None:
                gs.close()
        eager(tr)                                    # the first eager step after close()
        evals.append(evaluate(clip))
        outs[mode] = evals
        tr.close()
    for (ta, ia), (tb, ib) in zip(outs["eager"], outs["graph"]):
        assert torch.equal(ta, tb) and torch.equal(ia, ib)
    assert not torch.equal(outs["eager"][0][0], outs["eager"][1][0])      # (the weights did move between the evaluations)

    # (2)
    ref_clip, ref_tr = trainer("ref")
    for _ in range(5):
        eager(ref_tr)
    want = ref_tr.optim.flat_param.clone()
    ref_tr.close()
    clip, tr = trainer("failed")
    for _ in range(3):
        eager(tr)
    boom = RuntimeError("injected")
    from ct_clip_amd import functional as Fn
    real = Fn.refresh_shadows      # (the last call of the recorded step: every side stream has been joined, optim.step() has bumped its counter)
    monkeypatch.setattr(Fn, "refresh_shadows", lambda *a, **k: (_ for _ in ()).throw(boom))
    gs = GraphedStep(tr)
    with pytest.raises(RuntimeError):
        gs.capture(video, text)
    monkeypatch.setattr(Fn, "refresh_shadows", real)
    assert gs.state is None and tr.optim.step_count == 3
    tr.optim.zero_grad()
    for _ in range(2):
        eager(tr)
    torch.cuda.synchronize()
    assert tr.optim.step_count == 5 and torch.equal(tr.optim.flat_param, want)
    tr.close()


def test_graphed_step_captured_behind_an_eval_forward_and_recaptured(golden, tmp_path):
    """ADVICE r05.  (1) capture() right behind an EAGER forward with no optimiser step in between (validation, then capture): the lazily rebuilt
    weight shadows (the stacked q | k | v bias of the text tower, non-2D weights) were fresh then, their makers were not recorded and every replay
    read the capture-time values -- capture() now makes them stale first; eval + capture + three replays must equal eval + three eager steps bit for
    bit.  (2) Re-capture: an old GraphedStep finalised AFTER a new capture must not clear the step-state pointer the new one installed -- the new
    graph's replays and an eager step behind them still follow the eager sequence."""
    import gc
    import ct_clip_amd
    from ct_clip_amd.trainer import GraphedStep
    g = golden("tiny")
    c = g["config"]
    text = TextBatch(g["input_ids"].to(DEV), g["attention_mask"].to(DEV))
    video = g["video"].to(DEV)

    def run(mode):
        clip = build_model(c, g["state_dict"], DEV, torch.bfloat16)
        clip.train()
        tr = ct_clip_amd.CTClipTrainer(clip, num_train_steps=10, batch_size=2, tokenizer=object(), lr=1e-3, train_dataset=[0], evaluate=False,
                                       checkpoint=False, results_folder=str(tmp_path / mode), num_workers=0)

        def eager():
            loss = tr.forward_backward(video, text)
            tr.optim.step(tr.max_grad_norm)
            tr.optim.zero_grad()
            return float(loss)
        losses = [eager() for _ in range(3)]
        clip.eval()
        with torch.no_grad():
            clip(text, video, return_latents=True, device=DEV)      # every lazy shadow is FRESH behind this forward
        clip.train()
        if mode == "eager":
            losses += [eager() for _ in range(6)]
        else:
            gs = GraphedStep(tr).capture(video, text)
            losses += [float(gs.run()) for _ in range(3)]
            gs = GraphedStep(tr).capture(video, text)          # re-capture over the old instance: the old one is finalised after the new capture
            gc.collect()
            assert GraphedStep._state_owner is gs
            losses += [float(gs.run()) for _ in range(2)]
            gs.close()
            assert GraphedStep._state_owner is None
            losses.append(eager())
        torch.cuda.synchronize()
        return losses, tr.optim.flat_param.clone()
    la, pa = run("eager")
    lb, pb = run("graph")
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb)
