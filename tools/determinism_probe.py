"""Repeats forward + backward of the full-geometry fixture model under the trainer and prints loss / gradient checksums per run:
which stream configuration (if any) makes two runs differ?  usage: python tools/determinism_probe.py [runs]   (PROBE_POISON=nan|big poisons every torch.empty)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

if os.environ.get("PROBE_POISON"):      # every torch.empty / empty_like of a floating tensor comes back filled with NaN (or a huge value):
    _empty, _empty_like = torch.empty, torch.empty_like      # a kernel whose result depends on memory it never wrote shows up at once
    _val = float("nan") if os.environ["PROBE_POISON"] == "nan" else 3.0e4

    def _poison(t):
        if t.is_floating_point() and t.is_cuda and t.numel():
            t.fill_(_val)
        elif t.is_cuda and t.numel() and t.dtype == torch.uint8:
            t.fill_(0x7f)
        return t
    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
import ct_clip_amd  # noqa: E402
from tests.helpers import TextBatch, build_model, perturb_1d, synth_inputs  # noqa: E402

g = torch.load(os.path.join(ROOT, "tests", "golden", "full1.pt"), weights_only=False)
c = g["config"]
clip = build_model(c, None, torch.device("cpu"), torch.float32)
perturb_1d(clip, c["seed"])
video, ids, mask = synth_inputs(c)
dev = torch.device("cuda", 0)
clip.compute_dtype = torch.bfloat16
clip.visual_transformer.compute_dtype = torch.bfloat16
clip.to(dev).train()
text, video = TextBatch(ids.to(dev), mask.to(dev)), video.to(dev)
vq = clip.visual_transformer.vq._codebook
vq0 = (vq.embed.clone(), vq.cluster_size.clone())
trainer = ct_clip_amd.CTClipTrainer(clip, num_train_steps=1, batch_size=2, tokenizer=object(), lr=1e-6, train_dataset=[0], evaluate=False,
                                    checkpoint=False, results_folder="/tmp/probe", num_workers=0)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seen = {}
for i in range(runs):
    trainer.optim.zero_grad()
    vq.embed.copy_(vq0[0]); vq.cluster_size.copy_(vq0[1])
    torch.cuda.synchronize()
    loss = trainer.forward_backward(video, text)
    torch.cuda.synchronize()
    fg = trainer.optim.flat_grad
    key = (f"{float(loss.detach()):.10f}", f"{float(fg.double().sum()):.10e}")
    seen[key] = seen.get(key, 0) + 1
print(f"lib={os.environ.get('CTCLIP_LIB', 'product')} runs={runs} distinct (loss, gradient sum) results: {len(seen)}  {sorted(seen.items(), key=lambda kv: -kv[1])}", flush=True)
