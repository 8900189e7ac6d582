"""Which host call sites issue device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) in one bf16 training step at the bench geometry
(DEPTH + DEPTH layers, BERT-base)?  torch.profiler with stacks."""
import collections
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import ct_clip_amd  # noqa: E402

import argparse
import bench as Bm
args = argparse.Namespace(image=480, frames=240, spatial_depth=int(os.environ.get("DEPTH", "2")), temporal_depth=int(os.environ.get("DEPTH", "2")), batch=8,
                          bert_dropout=0.1, text_len=128)
clip, tr = Bm.build(args, torch.device("cuda", 0), torch.bfloat16)
clip.train()
gcpu = torch.Generator().manual_seed(1)
ids, mask = Bm.synth_text(8, 128, gcpu, torch.device("cuda", 0))
text = Bm.Text(ids, mask)
video = torch.rand(8, 1, 240, 480, 480, device="cuda") * 2 - 1
for _ in range(2):
    tr.forward_backward(video, text); tr.optim.step(0.5, zero_grad=True)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.forward_backward(video, text); tr.optim.step(0.5, zero_grad=True)
    torch.cuda.synchronize()
cnt = collections.Counter()
names = collections.Counter()
for ev in prof.events():
    n = ev.name
    if "emcpy" in n or "copyBuffer" in n or n in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::add_", "aten::add"):
        names[n] += 1
        if n in ("aten::copy_", "aten::cat", "aten::add_", "aten::add", "aten::clone", "aten::contiguous"):
            st = [s for s in (ev.stack or []) if "ct_clip_amd" in s or "transformers" in s]
            cnt[(n, tuple(str(s) for s in ev.input_shapes)[:2].__repr__()[:60], (st[0][-90:] if st else "(no python frame: autograd engine)"))] += 1
print(dict(names))
for k, v in cnt.most_common(40):
    print(v, k)
