"""Where does the bf16 error of the image tower come from, and what would an f32 residual stream buy?  (cost/benefit.)

The CPU oracle (f32 restatement of the reference, oracle/ctclip_oracle.py) runs the CTViT forward at the bench geometry (480x480x240, 12+12
layers, B = 1) with bf16 ROUNDING inserted at chosen storage points (values are rounded to bf16 and computed on in f32: exactly what bf16
storage + f32 accumulation does):
    R  the residual stream (after every residual add, the patch-embedding output, norm_out)
    A  every other stored activation (LayerNorm outputs, GEMM outputs, q / k / v, softmax probabilities, GEGLU output)
    W  the GEMM weights
R+A+W is the product's bf16 mode; A+W is the "f32 residual stream, bf16 GEMM operands" lever.  Printed: relative error of the pre-VQ tokens
and the VQ code agreement against the unrounded f32 run.     usage: python tools/bf16_rounding_study.py [sdepth tdepth]  (CPU, minutes)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as TF  # noqa: E402
from oracle import ctclip_oracle as O  # noqa: E402  (checker used as a numerical laboratory: a tool, not the product)

FLAGS = dict(R=False, A=False, W=False)


def r(t, which):
    return t.bfloat16().float() if FLAGS[which] else t


class FProxy:
    """torch.nn.functional with rounding at the storage points of the product's bf16 mode"""

    def __getattr__(self, name):
        return getattr(TF, name)

    @staticmethod
    def linear(x, w, b=None):
        return r(TF.linear(x, r(w, "W"), b), "A")

    @staticmethod
    def layer_norm(x, shape, g, b, eps):
        return r(TF.layer_norm(x, shape, g, b, eps), "A")

    @staticmethod
    def gelu(x):
        return TF.gelu(x)

    @staticmethod
    def normalize(t, dim=-1):
        return TF.normalize(t, dim=dim)


def transformer(sd, pre, cfg, depth, x, video_shape, attn_bias=None, trace=None):
    x = r(x, "R")
    for l in range(depth):
        p = f"{pre}layers.{l}."
        x = r(O.peg(sd, p + "0.", x, video_shape) + x, "R")
        x = r(O.attention(sd, p + "1.", cfg, x, attn_bias) + x, "R")
        x = r(O.feedforward(sd, p + "3.", x) + x, "R")
    g = sd[pre + "norm_out.gamma"]
    return r(TF.layer_norm(x, x.shape[-1:], g, torch.zeros_like(g), 1e-5), "R")


def feedforward(sd, pre, x):
    y = O.F.layer_norm(x, x.shape[-1:], sd[pre + "0.weight"], sd[pre + "0.bias"], 1e-5)
    y = TF.linear(y, r(sd[pre + "1.weight"], "W"))          # the GEGLU runs on the f32 accumulators in the GEMM epilogue: u is not rounded first
    a, gate = y.chunk(2, dim=-1)
    return O.F.linear(r(TF.gelu(gate) * a, "A"), sd[pre + "4.weight"])


O.F = FProxy()
O.transformer = transformer
O.feedforward = feedforward

sdepth, tdepth = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 12)
import ct_clip_amd  # noqa: E402
torch.manual_seed(0)
enc = ct_clip_amd.CTViT(dim=512, codebook_size=8192, image_size=480, patch_size=20, temporal_patch_size=10, spatial_depth=sdepth, temporal_depth=tdepth,
                        dim_head=32, heads=8, compute_dtype=torch.float32)
sd = {"visual_transformer." + k: v for k, v in enc.state_dict().items()}
cfg = O.OracleConfig(dim=512, codebook_size=8192, image_size=480, patch_size=20, temporal_patch_size=10, spatial_depth=sdepth, temporal_depth=tdepth,
                     dim_head=32, heads=8, bert_layers=12, bert_heads=12, dim_latent=512)
video = torch.rand(1, 1, 240, 480, 480, generator=torch.Generator().manual_seed(1234)) * 2 - 1
torch.set_num_threads(len(os.sched_getaffinity(0)))


def run(flags):
    FLAGS.update(dict(R="R" in flags, A="A" in flags, W="W" in flags))
    trace = {}
    t0 = time.time()
    with torch.no_grad():
        O.ctvit_forward(sd, cfg, video, training=False, trace=trace)
    return trace["pre_vq"].reshape(-1, 512), trace["vq_indices"].reshape(-1), time.time() - t0


ref_tok, ref_idx, dt = run("")
print(f"CTViT {sdepth}+{tdepth} layers, B = 1, random init (seed 0); f32 forward {dt:.0f} s on {torch.get_num_threads()} threads")
print("| rounding points | pre-VQ token rel. error | VQ code agreement |\n|---|---:|---:|")
for flags, label in (("R", "R: residual stream only"), ("A", "A: other activations only"), ("W", "W: weights only"), ("AW", "A+W: f32 residual stream, bf16 GEMM operands (the lever)"),
                     ("RAW", "R+A+W: the product's bf16 mode")):
    tok, idx, _ = run(flags)
    err = float((tok - ref_tok).norm() / ref_tok.norm())
    print(f"| {label} | {err:.2e} | {float((idx == ref_idx).float().mean()):.4f} |", flush=True)
