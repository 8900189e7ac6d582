"""CPU tests (pure-torch checker backend) of the zero-shot path of SURVEY.md section 8(f) rank 1: one image-tower pass per volume and
cached prompt latents must give exactly the scores of the reference's pair-by-pair loop (scripts/zero_shot.py:133-143), which calls
the no-loss similarity mode of CTCLIP.forward -- a mode the golden fixtures pin to the real reference (test_host_logic_cpu.py)."""
import numpy as np
import pytest
import torch

from ct_clip_amd import backend
from ct_clip_amd.zero_shot import CTClipInference, ZeroShotClassifier, prompts_for
from tests.helpers import TextBatch, build_model
from tests.ref_backend import RefBackend


@pytest.fixture()
def ref_backend():
    prev = backend.use(RefBackend())
    yield
    backend.use(prev)


class StubTokenizer:
    """Deterministic stand-in for BertTokenizer: ids from a hash of the words; same call signature as the reference uses."""

    def __init__(self, vocab, T):
        self.vocab, self.T = vocab, T

    def __call__(self, texts, return_tensors="pt", padding="max_length", truncation=True, max_length=512):
        T = min(self.T, max_length)
        ids = torch.zeros(len(texts), T, dtype=torch.int64)
        mask = torch.zeros(len(texts), T, dtype=torch.int64)
        for i, t in enumerate(texts):
            toks = [1] + [2 + (sum(map(ord, w)) * 31 + len(w)) % (self.vocab - 2) for w in t.replace(".", " .").split()][:T - 1]
            ids[i, :len(toks)] = torch.tensor(toks)
            mask[i, :len(toks)] = 1
        return TextBatch(ids, mask)


def test_cached_zero_shot_equals_pair_by_pair(golden, ref_backend):
    g = golden("tiny")
    c = g["config"]
    clip = build_model(c, g["state_dict"], torch.device("cpu"), torch.float32).eval()
    tok = StubTokenizer(c["vocab"], c["text_len"] if "text_len" in c else 32)
    pathologies = ["Cardiomegaly", "Pleural effusion", "Lung nodule"]
    torch.manual_seed(3)
    vol = torch.rand(1, 1, c["frames"], c["image"], c["image"]) * 2 - 1
    zs = ZeroShotClassifier(clip, tok, pathologies, max_length=32)
    fast = zs.predict(vol)
    slow = []
    with torch.no_grad():
        for name in pathologies:                      # the reference's loop: whole model per pathology
            pair = tok([f"{name} is present.", f"{name} is not present."], max_length=32)
            slow.append(torch.softmax(clip(pair, vol, device=torch.device("cpu")), dim=0)[0])
    torch.testing.assert_close(fast, torch.stack(slow), rtol=1e-5, atol=1e-6)
    assert zs.text_latents() is zs.text_latents()     # cached
    assert prompts_for(["A"]) == ["A is present.", "A is not present."]


def test_inference_driver_writes_the_reference_files(golden, ref_backend, tmp_path):
    g = golden("tiny")
    c = g["config"]
    clip = build_model(c, g["state_dict"], torch.device("cpu"), torch.float32)
    tok = StubTokenizer(c["vocab"], 32)
    torch.manual_seed(4)
    ds = [(torch.rand(1, c["frames"], c["image"], c["image"]) * 2 - 1, "report", torch.tensor([[1.0, 0.0]]), f"acc_{i}") for i in range(2)]
    inf = CTClipInference(clip, results_folder=str(tmp_path / "zs"), dataset=ds, tokenizer=tok, pathologies=["Emphysema", "Atelectasis"],
                          max_text_len=32)
    pred = inf.infer()
    assert pred.shape == (2, 2) and np.all((pred > 0) & (pred < 1))
    assert np.load(tmp_path / "zs" / "predicted_weights.npz")["data"].shape == (2, 2)
    assert np.load(tmp_path / "zs" / "labels_weights.npz")["data"].shape == (2, 2)
    assert (tmp_path / "zs" / "accessions.txt").read_text().split() == ["acc_0", "acc_1"]
    assert int(inf.steps.item()) == 1


def test_latent_export_matches_return_latents(golden, ref_backend, tmp_path):
    """scripts/forward_data.py: text/<acc>.npz and image/<acc>.npz hold exactly CTCLIP.forward(return_latents=True)[0:2]."""
    from ct_clip_amd.forward_data import CTClipInference as Export
    g = golden("tiny")
    c = g["config"]
    clip = build_model(c, g["state_dict"], torch.device("cpu"), torch.float32).eval()
    tok = StubTokenizer(c["vocab"], 32)
    torch.manual_seed(5)
    vol = torch.rand(1, c["frames"], c["image"], c["image"]) * 2 - 1
    Export(clip, results_folder=str(tmp_path / "lat"), dataset=[(vol, "no acute findings", torch.zeros(1, 2), "case_7")], tokenizer=tok,
           max_text_len=32).infer()
    with torch.no_grad():
        tl, il, _ = clip(tok(["no acute findings"], max_length=32), vol[None], device=torch.device("cpu"), return_latents=True)
    np.testing.assert_allclose(np.load(tmp_path / "lat" / "text" / "case_7.npz")["arr"], tl.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.load(tmp_path / "lat" / "image" / "case_7.npz")["arr"], il.numpy(), rtol=1e-6, atol=1e-7)


def test_ctclip_load_strips_the_ddp_prefix(golden, ref_backend, tmp_path):
    """CTCLIPTrainer.py:331-337 saves `get_state_dict(model, unwrap=False)`: under DDP every key starts with `module.`."""
    g = golden("tiny")
    clip = build_model(g["config"], g["state_dict"], torch.device("cpu"), torch.float32)
    sd = clip.state_dict()
    torch.save({"module." + k: v for k, v in sd.items()}, tmp_path / "m.pt")
    other = build_model(g["config"], None, torch.device("cpu"), torch.float32)
    other.load(tmp_path / "m.pt")
    for k, v in sd.items():
        assert torch.equal(other.state_dict()[k], v), k
