"""What does the text tower cost the step while it runs on its side stream under the image tower?  Runs bench.py's timed loop twice in
separate processes -- as shipped, and with the BERT forward replaced by a constant activation (no text kernels at all, forward or backward;
PROBE_SKIP_TEXT=1) -- so that the difference is the interference of the side-stream kernels with the main stream (they share CUs and HBM).
usage: python tools/probe_text_cost.py [bench.py flags]     (diagnostic only: the stubbed run is not a training step)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from ct_clip_amd import ctclip  # noqa: E402

if os.environ.get("PROBE_SKIP_TEXT") == "1":
    cache = {}

    def constant_text(bert, ids, mask, dt, od=None, cls_only=False):
        key = (tuple(ids.shape), dt, cls_only)
        if key not in cache:
            rows = ids.shape[0] if cls_only else ids.numel()
            cache[key] = torch.randn(rows, bert.config.hidden_size, device=ids.device).to(dt or torch.float32)
        return cache[key]

    ctclip._bert.bert_last_hidden_state = constant_text
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
