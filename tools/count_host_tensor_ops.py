import os, sys, collections, traceback, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from ct_clip_amd import backend
from tests.ref_backend import RefBackend
from tests.helpers import TextBatch, build_model
import ct_clip_amd

g = torch.load(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests/golden/tiny.pt'), weights_only=False)
clip = build_model(g["config"], g["state_dict"], torch.device("cuda"), torch.bfloat16)
clip.train()
tr = ct_clip_amd.CTClipTrainer(clip, num_train_steps=2, batch_size=2, tokenizer=object(), lr=1e-3, train_dataset=[0], evaluate=False, checkpoint=False, results_folder='/tmp/res', num_workers=0, device='cuda:0')
text = TextBatch(g["input_ids"].cuda(), g["attention_mask"].cuda()); g["video"] = g["video"].cuda()
tr.forward_backward(g["video"], text); tr.optim.step(0.5, zero_grad=True)
cnt = collections.Counter()
ON = [False]
def wrap(owner, name):
    orig = getattr(owner, name)
    def f(*a, **k):
        if ON[0]:
            ON[0] = False
            try:
                st = traceback.extract_stack()[:-1]
                fr = [s for s in st if 'ct_clip_amd/' in s.filename]
                if fr and 'ref_backend' not in fr[-1].filename:
                    is_noop = False
                    if name == 'contiguous' and a[0].is_contiguous(): is_noop = True
                    if name == 'to' and isinstance(a[0], torch.Tensor):
                        pass
                    if not is_noop:
                        cnt[(name, f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line[:80]}")] += 1
            finally:
                ON[0] = True
        return orig(*a, **k)
    setattr(owner, name, f)
for n in ("copy_", "clone", "contiguous", "to", "float", "add", "add_", "__add__", "__mul__", "mul", "zero_", "fill_", "bfloat16", "__getitem__"):
    if n != "__getitem__": wrap(torch.Tensor, n)
for n in ("cat", "stack", "zeros", "empty_like", "zeros_like"):
    wrap(torch, n)
ON[0] = True
tr.forward_backward(g["video"], text); tr.optim.step(0.5, zero_grad=True)
ON[0] = False
tot = collections.Counter()
for (n, w), v in cnt.items(): tot[n] += v
print(dict(tot))
for k, v in cnt.most_common(50): print(v, k)
