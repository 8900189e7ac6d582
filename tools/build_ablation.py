"""Builds ablation variants of the NT GEMM (ct_clip_amd/libctclip_abl<mask>.so, NT_ABL compile-time mask, see gemm_nt.hip) so that
`CTCLIP_LIB=ct_clip_amd/libctclip_abl<mask>.so python tools/bench_gemm.py` times the kernel with one phase removed.
usage: python tools/build_ablation.py 1 4 5 64 0,NT_STAGGER=1 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ct_clip_amd import build as B  # noqa: E402

B.build()
objs = [os.path.join(B.HERE, "build", os.path.basename(s) + ".o") for s in B.sources() if not s.endswith("gemm_nt.hip")]
procs = []
for m in sys.argv[1:]:
    tag = m.replace(",", "_").replace("=", "")
    obj = os.path.join(B.HERE, "build", f"gemm_nt_abl{tag}.o")
    mask, *defs = m.split(",")            # "12" = mask 12; "0,NT_STAGGER=1" = mask 0 plus an extra define
    cmd = ["hipcc"] + [f for f in B.FLAGS if f != "-shared"] + [f"-DNT_ABL={mask or 0}"] + [f"-D{d}" for d in defs] + ["-c", os.path.join(B.HERE, "csrc", "gemm_nt.hip"), "-o", obj]
    procs.append((m, obj, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)))
for m, obj, p in procs:
    assert p.wait() == 0, m
    lib = os.path.join(B.HERE, f"libctclip_abl{m.replace(',', '_').replace('=', '')}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [obj, "-o", lib])
    print("built", lib)
