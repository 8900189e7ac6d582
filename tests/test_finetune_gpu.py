"""Fine-tuning loops on the HIP path against tests/golden/finetune_tiny.pt (outputs of the loops' arithmetic on the REAL reference
towers: oracle/gen_golden.py run_finetune_case) -- ClassFine / CT-LiPro head + BCE(pos_weight) and the VocabFine objective."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_finetune_cpu import check_lipro, check_vocabfine, load  # noqa: E402

DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("skip_text", [True, False])
def test_lipro_f32_matches_reference(skip_text):
    g, f = load()
    check_lipro(g, f, DEV, torch.float32, skip_text, 5e-4)


@pytest.mark.parametrize("fused", [True, False])
def test_vocabfine_f32_matches_reference(fused):
    """fused = one pass of each tower per volume (default); not fused = the reference's 18-forward loop literally."""
    g, f = load()
    check_vocabfine(g, f, DEV, torch.float32, 5e-4, 1e-2, fused=fused)


def test_lipro_bf16_stays_close():
    """bf16 towers: the frozen image latents move by bf16 rounding (and possible VQ code flips), the head itself is f32."""
    import ct_clip_amd.finetune as FT
    from tests.test_finetune_cpu import lipro_setup
    from tests.helpers import TextBatch
    g, f = load()
    clip, head, tr = lipro_setup(g, f, DEV, torch.bfloat16, True)
    blank = TextBatch(g["input_ids"][:1].to(DEV), g["attention_mask"][:1].to(DEV))
    loss, logits = tr.forward_backward(blank, g["video"].to(DEV), f["lipro"]["labels"])
    rel = abs(float(loss) - float(f["lipro"]["loss"])) / float(f["lipro"]["loss"])
    print(f"[lipro bf16] loss rel {rel:.2e}")
    assert rel < 5e-2


# ---------------------------------------------------------------------------------------------------------------- full geometry (round 6)
# tests/golden/finetune_full.pt (oracle/gen_golden.py finetune_full): the loops of scripts/ct_lipro_train.py:17-38,79-107 and
# scripts/ct_vocabfine_train.py:88-121 on the REAL reference towers at BASELINE configs[3] / configs[4] geometry (480 x 480 x 240, dim 512, the
# scripts' 4+4 layers, BERT-base, T = 128) -- same seed / weights / inputs as full1.pt, rebuilt here from the seeds and checked against full1's fingerprints.
import functools  # noqa: E402
import os  # noqa: E402


@functools.lru_cache(maxsize=None)
def _full():
    from tests.test_full_size_gpu import _load
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = torch.load(os.path.join(root, "tests", "golden", "finetune_full.pt"), weights_only=False)
    g, clip, state, video, ids, mask = _load("full1")
    assert f["config"] == g["config"]
    return f, clip, state, video, ids, mask


def _prepare(dtype):
    from tests.helpers import TextBatch
    f, clip, state, video, ids, mask = _full()
    clip.load_state_dict(state)
    clip.compute_dtype = dtype
    clip.visual_transformer.compute_dtype = dtype
    clip.to(DEV)
    for p in clip.parameters():
        p.grad = None
        p.requires_grad_(True)
    clip.visual_transformer.vq.__dict__.pop("teacher_indices", None)
    return f, clip, video, TextBatch(ids[:1].to(DEV), mask[:1].to(DEV))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_lipro_full_geometry_matches_reference(dtype, tol):
    """One CT-LiPro step at the full geometry, B = 2: latents, logits, BCE(pos_weight) loss, the head's gradients and the post-step VQ buffers
    against the real reference (f32: to rounding; bf16: the frozen tower's latents move by bf16 rounding and VQ code flips)."""
    import ct_clip_amd.finetune as FT
    f, clip, video, blank = _prepare(dtype)
    L = f["lipro"]
    head = FT.ImageLatentsClassifier(clip, f["config"]["dim_latent"], 18, dropout_prob=0.0).to(DEV)
    with torch.no_grad():
        head.classifier.weight.copy_(L["W"]); head.classifier.bias.copy_(L["b"])
    tr = FT.LiProTrainer(head, lr=1e-5, wd=0.1, warmup_length=2, total_steps=10, pos_weight=L["pos_weight"].tolist())
    loss, logits = tr.forward_backward(blank, video.to(DEV), L["labels"])
    rel = abs(float(loss) - float(L["loss"])) / float(L["loss"])
    lerr = float((logits.float().cpu() - L["logits"]).norm() / L["logits"].norm())
    werr = float((head.classifier.weight.grad.cpu() - L["dW"]).norm() / L["dW"].norm())
    berr = float((head.classifier.bias.grad.cpu() - L["db"]).norm() / L["db"].norm())
    cs = clip.state_dict()["visual_transformer.vq._codebook.cluster_size"].float().cpu()
    cerr = float((cs - L["vq_after"]["visual_transformer.vq._codebook.cluster_size"]).norm() / L["vq_after"]["visual_transformer.vq._codebook.cluster_size"].norm())
    print(f"[lipro full geometry {dtype}] loss rel {rel:.2e}, logits {lerr:.2e}, dW {werr:.2e}, db {berr:.2e}, VQ cluster sizes {cerr:.2e}")
    assert rel < tol and lerr < 5 * tol and werr < 10 * tol and berr < 10 * tol and cerr < (1e-3 if dtype == torch.float32 else 0.1)
    for p in clip.parameters():
        assert p.grad is None                      # frozen towers (ct_lipro_train.py:20-21)
    clip.to("cpu")


def test_vocabfine_full_geometry_matches_reference():
    """One VocabFine step at the full geometry (one volume, four pathologies in two groups, every parameter trains): similarities, group losses
    and ALL gradients of the fused one-pass form against the reference's loop literally run on its own towers (f32)."""
    import ct_clip_amd.finetune as FT
    from tests.helpers import TextBatch
    from tests.test_full_size_gpu import rel_err
    f, clip, video, _ = _prepare(torch.float32)
    V = f["vocabfine"]
    tr = FT.VocabFineTrainer(clip, tokenize=None, lr=1e-5, wd=0.1, warmup_length=2, total_steps=10, pathologies=["a", "b", "c", "d"], group_size=V["group"])
    pairs = [TextBatch(V["prompt_ids"][i].to(DEV), V["prompt_mask"][i].to(DEV)) for i in range(V["prompt_ids"].shape[0])]
    losses, sims = tr.forward_backward(video[:1].to(DEV), pairs)
    for a, b in zip(sims, V["sims"]):
        torch.testing.assert_close(a.float().cpu(), b, rtol=5e-4, atol=5e-4)
    for a, b in zip(losses, V["losses"]):
        torch.testing.assert_close(a.float().cpu(), b, rtol=1e-3, atol=1e-6)
    grads = dict((n, p.grad) for n, p in clip.named_parameters() if p.grad is not None)
    worst, n = (0.0, None), 0
    gn = float(V["grad_norm"])
    for k, rec in V["grads"].items():
        if rec["value"].numel() == 0 or k not in grads or float(rec["norm"]) < 1e-4 * gn:
            continue
        e = rel_err(rec, grads[k])
        n += 1
        if e > worst[0]:
            worst = (e, k)
    mine = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    print(f"[vocabfine full geometry f32] {n} gradients above 1e-4 of the norm, worst relative error {worst[0]:.2e} ({worst[1]}); gradient norm {mine:.6f} reference {gn:.6f}")
    assert n > 100 and worst[0] < 5e-3 and abs(mine - gn) <= 1e-3 * gn
    cs = clip.state_dict()["visual_transformer.vq._codebook.cluster_size"].float().cpu()
    torch.testing.assert_close(cs, V["vq_after"]["visual_transformer.vq._codebook.cluster_size"], rtol=1e-3, atol=1e-3)
    clip.to("cpu")
