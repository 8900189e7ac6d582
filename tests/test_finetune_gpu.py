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
