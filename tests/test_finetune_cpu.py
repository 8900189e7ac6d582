"""Fine-tuning loops (ClassFine / CT-LiPro and VocabFine) -- the product's composition on the torch checker
backend against tests/golden/finetune_tiny.pt, which oracle/gen_golden.py produced by running the loops' arithmetic on the REAL
reference towers (scripts/ct_lipro_train.py:17-38,79-107; scripts/ct_vocabfine_train.py:88-121)."""
import os

import pytest
import torch

from ct_clip_amd import backend, finetune as FT
from tests.helpers import TextBatch, build_model, check_grad
from tests.ref_backend import RefBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def ref_backend():
    prev = backend.use(RefBackend())
    yield
    backend.use(prev)


def load():
    g = torch.load(os.path.join(ROOT, "tests", "golden", "tiny.pt"), weights_only=False)
    f = torch.load(os.path.join(ROOT, "tests", "golden", "finetune_tiny.pt"), weights_only=False)
    return g, f


def lipro_setup(g, f, device, dtype, skip_text):
    clip = build_model(g["config"], g["state_dict"], device, dtype)
    head = FT.ImageLatentsClassifier(clip, g["config"]["dim_latent"], 18, dropout_prob=0.0, skip_text=skip_text).to(device)
    with torch.no_grad():
        head.classifier.weight.copy_(f["lipro"]["W"]); head.classifier.bias.copy_(f["lipro"]["b"])
    tr = FT.LiProTrainer(head, lr=1e-3, wd=0.1, warmup_length=2, total_steps=10, pos_weight=f["lipro"]["pos_weight"].tolist())
    return clip, head, tr


def check_lipro(g, f, device, dtype, skip_text, tol):
    clip, head, tr = lipro_setup(g, f, device, dtype, skip_text)
    blank = TextBatch(g["input_ids"][:1].to(device), g["attention_mask"][:1].to(device))
    loss, logits = tr.forward_backward(blank, g["video"].to(device), f["lipro"]["labels"])
    L = f["lipro"]
    torch.testing.assert_close(logits.detach().cpu(), L["logits"], rtol=tol, atol=tol)
    torch.testing.assert_close(loss.detach().cpu(), L["loss"], rtol=tol, atol=tol)
    torch.testing.assert_close(head.classifier.weight.grad.cpu(), L["dW"], rtol=10 * tol, atol=tol)
    torch.testing.assert_close(head.classifier.bias.grad.cpu(), L["db"], rtol=10 * tol, atol=tol)
    assert all(p.grad is None for p in clip.parameters())                      # frozen towers (ct_lipro_train.py:20-21)
    sd = clip.state_dict()
    for k, v in L["vq_after"].items():                                        # train mode still moves the VQ buffers, as in the reference
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-3, atol=1e-4)
    clip2, head2, tr2 = lipro_setup(g, f, device, dtype, skip_text)
    probs = tr2.predict(blank, g["video"].to(device))
    torch.testing.assert_close(probs.cpu(), L["eval_probs"], rtol=tol, atol=tol)


def check_vocabfine(g, f, device, dtype, tol, gtol, fused=True):
    V = f["vocabfine"]
    clip = build_model(g["config"], g["state_dict"], device, dtype)
    tr = FT.VocabFineTrainer(clip, tokenize=None, lr=1e-5, wd=0.1, warmup_length=2, total_steps=10, pathologies=["a", "b", "c", "d"],
                             group_size=V["group"])
    pairs = [TextBatch(V["prompt_ids"][i].to(device), V["prompt_mask"][i].to(device)) for i in range(V["prompt_ids"].shape[0])]
    losses, sims = tr.forward_backward(g["video"][:1].to(device), pairs, fused=fused)
    for a, b in zip(sims, V["sims"]):
        torch.testing.assert_close(a.cpu(), b, rtol=tol, atol=tol)
    for a, b in zip(losses, V["losses"]):
        torch.testing.assert_close(a.cpu(), b, rtol=tol, atol=tol * 0.1)
    grads = dict((n, p.grad) for n, p in clip.named_parameters() if p.grad is not None)
    n, bad = 0, []
    for k, rec in V["grads"].items():
        if rec["value"].numel() == 0 or k not in grads or float(rec["norm"]) < 1e-6 * float(V["grad_norm"]):
            continue
        try:
            check_grad(rec, grads[k], rtol=gtol, atol_rel=gtol * 0.2, floor=1e-7 * float(V["grad_norm"]))
        except AssertionError:
            m = grads[k] if rec["full"] else grads[k].reshape(-1)[::rec["stride"]]
            bad.append((k, float((m.float().cpu().reshape(-1) - rec["value"].reshape(-1)).norm() / rec["value"].norm())))
        n += 1
    assert not bad, f"{len(bad)} of {n} gradients differ: {bad[:12]}"
    assert n > 30
    sd = clip.state_dict()
    for k, v in V["vq_after"].items():                                        # four train-mode forwards = four EMA updates
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("skip_text", [True, False])
def test_lipro_matches_reference(ref_backend, skip_text):
    g, f = load()
    check_lipro(g, f, torch.device("cpu"), torch.float32, skip_text, 2e-4)


@pytest.mark.parametrize("fused", [True, False])
def test_vocabfine_matches_reference(ref_backend, fused):
    """fused: one pass of each tower per volume; not fused: the reference's loop literally.  Both against the REAL reference's loop."""
    g, f = load()
    check_vocabfine(g, f, torch.device("cpu"), torch.float32, 2e-4, 5e-3, fused=fused)


def test_cosine_lr_schedule():
    class Opt:
        param_groups = [{"lr": 0.0}]
    o = Opt()
    sched = FT.cosine_lr(o, 1e-3, 4, 20)
    got = []
    for s in range(20):
        sched(s)
        got.append(o.param_groups[0]["lr"])
    assert got[0] == pytest.approx(1e-3 / 4) and got[3] == pytest.approx(1e-3) and got[4] == pytest.approx(1e-3)
    assert got[12] == pytest.approx(0.5e-3) and got[19] < 0.02e-3 and all(a >= b for a, b in zip(got[4:], got[5:]))


def test_vocabfine_prompts_follow_the_label():
    assert FT.vocabfine_prompts("Emphysema", 1) == ["Emphysema is present. ", "Emphysema is not present. "]
    assert FT.vocabfine_prompts("Emphysema", 0) == ["Emphysema is not present. ", "Emphysema is present. "]
