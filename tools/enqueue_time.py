"""How long does the host take to ENQUEUE one optimisation step (all launches issued, nothing waited for) against the step's device time?
usage: python tools/enqueue_time.py [steps]   (bench.py's default configuration)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sys.argv = [sys.argv[0]]
ap_args = type("A", (), dict(image=480, frames=240, spatial_depth=12, temporal_depth=12, bert_dropout=0.1, batch=8, text_len=128))()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
clip, trainer = bench.build(ap_args, dev, torch.bfloat16)
clip.train()
gd = torch.Generator(device=dev).manual_seed(1234)
video = torch.rand(8, 1, 240, 480, 480, generator=gd, device=dev) * 2 - 1
ids, mask = bench.synth_text(8, 128, torch.Generator().manual_seed(1234), dev)
text = bench.Text(ids, mask)


def step():
    loss = trainer.forward_backward(video, text)
    trainer.optim.step(trainer.max_grad_norm)
    trainer.optim.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(steps):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"eager: enqueue {sum(enq) / len(enq):.1f} ms (min {min(enq):.1f}) of {sum(tot) / len(tot):.1f} ms per step, synchronised after every step")
# the same step captured into a hipGraph (ct_clip_amd.trainer.GraphedStep): what the host pays per step then
from ct_clip_amd.trainer import GraphedStep  # noqa: E402
gs = GraphedStep(trainer).capture(video, text)
gs.run()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(steps):
    t0 = time.perf_counter()
    gs.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
gs.close()
print(f"hipGraph replay: enqueue {sum(enq) / len(enq):.2f} ms (min {min(enq):.2f}) of {sum(tot) / len(tot):.1f} ms per step, synchronised after every step")
