"""Per-step wall times of bench.py's three configurations run back to back in ONE process (12+12, T=128 -> 4+4 -> 12+12, T=512): does a
configuration depend on what ran before it?  usage: python tools/probe_config_sequence.py [steps]"""
import gc
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
for sd, td, T in ((12, 12, 128), (4, 4, 128), (12, 12, 512), (4, 4, 128)):
    args = types.SimpleNamespace(image=bench.FULL["image"], frames=bench.FULL["frames"], spatial_depth=sd, temporal_depth=td, batch=8, bert_dropout=0.1)
    clip, trainer = bench.build(args, dev, torch.bfloat16)
    clip.train()
    gd = torch.Generator(device=dev).manual_seed(1)
    video = torch.rand(8, 1, args.frames, args.image, args.image, generator=gd, device=dev) * 2 - 1
    text = bench.Text(*bench.synth_text(8, T, torch.Generator().manual_seed(1), dev))
    ts = []
    free_run = os.environ.get("FREE_RUN") == "1"          # bench.py's loop: no synchronisation between the timed steps
    for i in range(3 if free_run else 0):
        loss = trainer.forward_backward(video, text); trainer.optim.step(trainer.max_grad_norm); trainer.optim.zero_grad()
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for i in range(steps):
        if not free_run:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = trainer.forward_backward(video, text)
        trainer.optim.step(trainer.max_grad_norm)
        trainer.optim.zero_grad()
        if not free_run:
            torch.cuda.synchronize()
        ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all) * 1e3 / steps
    print(f"{sd}+{td} T={T}: {tot:.1f} ms per step; per step ({'host enqueue' if free_run else 'synchronised'}) {ts}  reserved {torch.cuda.memory_reserved() / 2 ** 30:.1f} GiB", flush=True)
    del clip, trainer, video, text, loss
    if os.environ.get("NO_GC") != "1":
        gc.collect()
    torch.cuda.empty_cache()
