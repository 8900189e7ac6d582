"""Zero-shot scoring on the HIP path: cached towers == the reference's pair-by-pair loop."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ct_clip_amd.zero_shot import ZeroShotClassifier  # noqa: E402
from tests.helpers import build_model  # noqa: E402
from tests.test_zero_shot_cpu import StubTokenizer  # noqa: E402

DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_cached_zero_shot_equals_pair_by_pair_on_the_gpu(golden, dtype, tol):
    g = golden("tiny")
    c = g["config"]
    clip = build_model(c, g["state_dict"], DEV, dtype).eval()
    tok = StubTokenizer(c["vocab"], 32)
    pathologies = ["Cardiomegaly", "Pleural effusion", "Lung nodule", "Emphysema"]
    vol = (torch.rand(1, 1, c["frames"], c["image"], c["image"], generator=torch.Generator().manual_seed(3)) * 2 - 1).to(DEV)
    zs = ZeroShotClassifier(clip, tok, pathologies, max_length=32)
    fast = zs.predict(vol)
    slow = []
    with torch.no_grad():
        for name in pathologies:
            pair = tok([f"{name} is present.", f"{name} is not present."], max_length=32).to(DEV)
            slow.append(torch.softmax(clip(pair, vol, device=DEV), dim=0)[0])
    torch.testing.assert_close(fast, torch.stack(slow), rtol=tol, atol=tol)


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_zero_shot_scores_match_the_real_reference(golden, name):
    """The fixture holds the REAL reference's no-loss similarity of prompts 0-1 with volume 0 (ct_clip.py:805-807, the quantity
    scripts/zero_shot.py:133-143 feeds to its softmax): the cached-latent scorer must reproduce it on the HIP path."""
    from ct_clip_amd.zero_shot import ZeroShotClassifier
    g = golden(name)
    c = g["config"]
    clip = build_model(c, g["state_dict"], DEV, torch.float32).eval()

    class FixedTok:                       # returns the fixture's own token ids for the one prompt pair
        def __call__(self, texts, **kw):
            from tests.helpers import TextBatch
            return TextBatch(g["input_ids"][:2], g["attention_mask"][:2])
    zs = ZeroShotClassifier(clip, FixedTok(), ["p"], max_length=c["T"])
    got = zs.predict(g["video"][:1].to(DEV))
    want = torch.softmax(g["eval_similarity_2v1"], dim=0)[0]
    torch.testing.assert_close(got.reshape(-1).cpu(), want.reshape(-1), rtol=1e-3, atol=1e-4)
