"""Builds ct_clip_amd/libctclip_<tag>.so with extra -D defines on chosen sources (determinism bisection / A-B builds).
usage: python tools/build_variant.py <tag> file.hip:MACRO=1[,MACRO2=0] [file2.hip:...]      then CTCLIP_LIB=ct_clip_amd/libctclip_<tag>.so ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ct_clip_amd import build as B  # noqa: E402

tag, specs = sys.argv[1], dict(a.split(":") for a in sys.argv[2:])
B.build()
objs, procs = [], []
for s in B.sources():
    base = os.path.basename(s)
    if base in specs:
        obj = os.path.join(B.HERE, "build", f"{base}.{tag}.o")
        cmd = (["hipcc"] + [f for f in B.FLAGS if f != "-shared"] + B.FILE_FLAGS.get(base, []) + [f"-D{d}" for d in specs[base].split(",")] + ["-c", s, "-o", obj])
        procs.append((base, subprocess.Popen(cmd)))
    else:
        obj = os.path.join(B.HERE, "build", base + ".o")
    objs.append(obj)
for base, p in procs:
    assert p.wait() == 0, base
lib = os.path.join(B.HERE, f"libctclip_{tag}.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
print("built", lib)
