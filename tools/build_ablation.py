"""Builds ablation variants of one kernel source (ct_clip_amd/libctclip_<tag>.so) with a compile-time mask, so that
`CTCLIP_LIB=ct_clip_amd/libctclip_<tag>.so python tools/bench_*.py` times the kernel with one phase removed.  (Run-time switches are
useless for this: the compiler unswitches the loops and the extra branches perturb the production code.)
usage: python tools/build_ablation.py [file.hip:MACRO] mask[,EXTRA=1] ...      default file: gemm_nt.hip:NT_ABL
e.g.   python tools/build_ablation.py attn2.hip:ATTN2_ABL 1 2 4 8 32"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ct_clip_amd import build as B  # noqa: E402

args = sys.argv[1:]
src, macro = "gemm_nt.hip", "NT_ABL"
if args and ":" in args[0]:
    src, macro = args.pop(0).split(":")
stem = src.replace(".hip", "")
B.build()
objs = [os.path.join(B.HERE, "build", os.path.basename(s) + ".o") for s in B.sources() if not s.endswith(src)]
procs = []
for m in args:
    tag = m.replace(",", "_").replace("=", "")
    obj = os.path.join(B.HERE, "build", f"{stem}_abl{tag}.o")
    mask, *defs = m.split(",")            # "12" = mask 12; "0,NT_STAGGER=1" = mask 0 plus an extra define
    cmd = (["hipcc"] + [f for f in B.FLAGS if f != "-shared"] + B.FILE_FLAGS.get(src, []) + [f"-D{macro}={mask or 0}"] + [f"-D{d}" for d in defs]
           + ["-c", os.path.join(B.HERE, "csrc", src), "-o", obj])
    procs.append((m, obj, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)))
for m, obj, p in procs:
    assert p.wait() == 0, m
    pre = "abl" if src == "gemm_nt.hip" else stem + "_abl"
    lib = os.path.join(B.HERE, f"libctclip_{pre}{m.replace(',', '_').replace('=', '')}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [obj, "-o", lib])
    print("built", lib)
