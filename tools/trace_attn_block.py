"""Runs bench.py's attention block (forward + backward) a few times: under `rocprofv3 --kernel-trace --output-format csv` the kernel trace shows what
the block is made of.  usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/trace_attn_block.py; python tools/trace_attn_block.py --summarise OUT"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    # the last repetition: kernels after the last spin marker pair
    marks = [i for i, r in enumerate(rows) if "spin_kernel" in r[2]]
    seg = rows[marks[-2] + 1:marks[-1]] if len(marks) >= 2 else rows
    t0 = seg[0][0]
    agg = collections.OrderedDict()
    for s, e, n in seg:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n[:100]}")
    print("sum of kernel durations", sum(e - s for s, e, _ in seg) / 1e3, "us; span", (seg[-1][1] - seg[0][0]) / 1e3, "us")
    sys.exit(0)

import argparse  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

args = argparse.Namespace(image=480, frames=240, batch=8)
be = backend.get()
dev = torch.device("cuda", 0)
spin = lambda: be.lib.ctclip_spin(20, torch.cuda.current_stream().cuda_stream)      # markers in the trace
spin(); out = bench.attention_block_util(args, dev, torch.bfloat16, iters=1); spin()
torch.cuda.synchronize()
print(out["fwd_us"], out["fwd_bwd_us"])
