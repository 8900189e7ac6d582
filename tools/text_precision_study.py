"""CPU study (test infrastructure; uses the torch checker backend tests/ref_backend.py, which rounds where the HIP kernels round):
how far the text tower's CLS rows are from the f32 HuggingFace BertModel in the three precision modes of CTCLIP.text_compute_dtype.
    python tools/text_precision_study.py [layers] [B] [T]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import BertConfig, BertModel
from ct_clip_amd import backend, bert as Bm
from tests.ref_backend import RefBackend

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 12
Bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
backend.use(RefBackend())
torch.manual_seed(3)
cfg = BertConfig(num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
model = BertModel(cfg).eval()
g = torch.Generator().manual_seed(99)
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.ndim <= 1:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
ids = torch.randint(3, cfg.vocab_size, (Bsz, T), generator=g)
lens = torch.randint(T // 2, T + 1, (Bsz,), generator=g)
mask = (torch.arange(T)[None] < lens[:, None]).long()
ids = ids * mask
with torch.no_grad():
    ref = model(ids, attention_mask=mask)[0][:, 0]
    for name, (dt, od) in dict(f32=(torch.float32, None), bf16=(torch.bfloat16, None), mixed=(torch.float32, torch.bfloat16)).items():
        out = Bm.bert_last_hidden_state(model, ids, mask, dt, od).view(Bsz, T, -1)[:, 0].float()
        rel = float((out - ref).norm() / ref.norm())
        cos = float(torch.nn.functional.cosine_similarity(out, ref, dim=-1).min())
        print(f"{name:6s} layers {layers}: CLS rel err {rel:.3e}  min cosine {cos:.7f}")
