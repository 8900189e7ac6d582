"""Builds libctclip_hip.so from csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libctclip_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]
# MFMA results that are post-processed by VALU (softmax) stay in VGPRs: the AGPR form costs a copy per register and tile
FILE_FLAGS = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attn2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
              "attn2_slab.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attn2_bwd1.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attn2_bwd2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attn_short.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(HERE, "csrc", "*.h"))
    return any(os.path.getmtime(f) > t for f in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    subprocess.check_call(cmd)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
