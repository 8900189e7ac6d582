"""debug: ratio of the HIP VocabFine gradients to the fixture's (GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_finetune_cpu import load
from tests.helpers import TextBatch, build_model
from ct_clip_amd import finetune as FT
DEV = torch.device("cuda", 0)
g, f = load()
V = f["vocabfine"]
for groups in (1, 2):
    clip = build_model(g["config"], g["state_dict"], DEV, torch.float32)
    tr = FT.VocabFineTrainer(clip, tokenize=None, lr=1e-5, wd=0.1, warmup_length=2, total_steps=10, pathologies=["a", "b", "c", "d"], group_size=V["group"])
    pairs = [TextBatch(V["prompt_ids"][i].to(DEV), V["prompt_mask"][i].to(DEV)) for i in range(V["prompt_ids"].shape[0])]
    losses, sims = tr.forward_backward(g["video"][:1].to(DEV), pairs[:2 * groups])
    torch.cuda.synchronize()
    print("groups", groups, "losses", [float(x) for x in losses], "ref", [float(x) for x in V["losses"]])
    grads = dict((n, p.grad) for n, p in clip.named_parameters() if p.grad is not None)
    for k in ["temperature", "to_text_latent.weight", "to_visual_latent.weight", "text_transformer.encoder.layer.0.attention.self.query.weight",
              "visual_transformer.enc_spatial_transformer.layers.0.3.1.weight", "visual_transformer.to_patch_emb.2.weight"]:
        rec = V["grads"][k]
        m = grads[k] if rec["full"] else grads[k].reshape(-1)[::rec["stride"]]
        m = m.float().cpu().reshape(-1); r = rec["value"].reshape(-1)
        print(f"  {k}: |mine| {float(m.norm()):.4e} |ref(2 groups)| {float(r.norm()):.4e} cos {float((m @ r) / (m.norm() * r.norm() + 1e-30)):.4f}")
