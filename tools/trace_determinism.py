"""In-situ bit-reproducibility trace of the training step: which LAUNCH deviates first?

The model (the full-geometry fixture `full1`: 480x480x240, B = 2, 4+4 layers -- or `--config bench`: B = 8, 12+12 layers, random init) runs
forward + backward under the trainer RUNS times from identical state.  The backend is wrapped: after every primitive call a 64-bit
integer checksum of every tensor the call touched (arguments AND results: in-place outputs are arguments) is computed ON THE DEVICE, on
the launch stream of that call (no host synchronisation: stream concurrency -- text tower, weight-gradient stream -- stays as in
production).  After the run the checksum vector is compared with run 0's: the first deviating entry names the call (primitive, call
index, tensor position, shape) whose output -- or whose input, i.e. the torch glue in front of it -- differed.

usage: python tools/trace_determinism.py [--runs N] [--config full1|bench] [--noise 0|1] [--forward-only]
env:   CTCLIP_LIB=... (ablation / strict-wait builds), CTCLIP_TEXT_STREAM=0, CTCLIP_WGRAD_STREAM=0
prints one line per deviating run and a JSON summary line (TRACE_SUMMARY {...})."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=60)
ap.add_argument("--config", default="full1")
ap.add_argument("--noise", type=int, default=0, help="1 = an unrelated GEMM stream keeps the chip busy underneath")
ap.add_argument("--forward-only", action="store_true")
ap.add_argument("--max-report", type=int, default=6)
args = ap.parse_args()

import ct_clip_amd  # noqa: E402
from ct_clip_amd import backend  # noqa: E402

dev = torch.device("cuda", 0)


def cks(t):
    """device-side 64-bit checksum of the raw bits (no sync)"""
    raw = t.detach()
    if not raw.is_contiguous():
        raw = raw.contiguous()
    raw = raw.reshape(-1)
    nb = raw.numel() * raw.element_size()
    if nb % 4 == 0 and raw.data_ptr() % 4 == 0:
        return raw.view(torch.int32).sum(dtype=torch.int64)
    return raw.view(torch.uint8).sum(dtype=torch.int64)


def tensors_of(obj, out):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda and obj.numel() > 0 and obj.dtype not in (torch.bool, torch.uint8):      # (uint8 = workspaces: partly unwritten)
            out.append(obj)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            tensors_of(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            tensors_of(o, out)


class Tracer:
    SKIP = {"workspace", "start_gemm_timing", "stop_gemm_timing"}

    def __init__(self, inner):
        self.__dict__["inner"] = inner
        self.__dict__["sums"] = []
        self.__dict__["meta"] = []

    def reset(self):
        self.sums.clear()
        self.meta.clear()

    def __getattr__(self, name):
        attr = getattr(self.inner, name)
        if not callable(attr) or name.startswith("_") or name in self.SKIP:
            return attr

        def wrapped(*a, **k):
            res = attr(*a, **k)
            ts = []
            tensors_of(res, ts)
            nres = len(ts)
            tensors_of(a, ts)
            tensors_of(k, ts)
            call = self.meta[-1][1] + 1 if self.meta else 0
            for i, t in enumerate(ts):
                self.sums.append(cks(t))
                self.meta.append((name, call, ("out" if i < nres else "arg") + str(i if i < nres else i - nres), tuple(t.shape), str(t.dtype)[6:],
                                  int(torch.cuda.current_stream() != torch.cuda.default_stream())))
            return res
        return wrapped

    def __setattr__(self, k, v):
        setattr(self.inner, k, v)


tr = Tracer(backend.get())
backend.use(tr)

from tests.helpers import TextBatch, build_model, perturb_1d, synth_inputs  # noqa: E402

if args.config == "full1":
    g = torch.load(os.path.join(ROOT, "tests", "golden", "full1.pt"), weights_only=False)
    c = g["config"]
    clip = build_model(c, None, torch.device("cpu"), torch.float32)
    perturb_1d(clip, c["seed"])
    video, ids, mask = synth_inputs(c)
    B = c["batch"]
else:
    import bench as BN
    ns = argparse.Namespace(image=480, frames=240, spatial_depth=12, temporal_depth=12, bert_dropout=0.0, batch=8)
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    enc = ct_clip_amd.CTViT(dim=512, codebook_size=8192, image_size=480, patch_size=20, temporal_patch_size=10, spatial_depth=12,
                            temporal_depth=12, dim_head=32, heads=8, compute_dtype=torch.float32)
    bert = BertModel(BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    clip = ct_clip_amd.CTCLIP(image_encoder=enc, text_encoder=bert, dim_text=768, dim_image=24 * 24 * 512, dim_latent=512,
                              compute_dtype=torch.float32)
    B = 8
    gen = torch.Generator().manual_seed(1234)
    ids, mask = BN.synth_text(B, 128, gen, "cpu")
    video = None
clip.compute_dtype = torch.bfloat16
clip.visual_transformer.compute_dtype = torch.bfloat16
clip.to(dev).train()
if video is None:
    video = torch.rand(B, 1, 240, 480, 480, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)) * 2 - 1
text, video = TextBatch(ids.to(dev), mask.to(dev)), video.to(dev)
vq = clip.visual_transformer.vq._codebook
vq0 = (vq.embed.clone(), vq.cluster_size.clone())
trainer = ct_clip_amd.CTClipTrainer(clip, num_train_steps=1, batch_size=B, tokenizer=object(), lr=1e-6, train_dataset=[0], evaluate=False,
                                    checkpoint=False, results_folder="/tmp/probe", num_workers=0)

noise_stream = torch.cuda.Stream(device=dev)
na = (torch.rand(8192, 2048, device=dev) - 0.5).bfloat16()
nb_ = (torch.rand(4096, 2048, device=dev) - 0.5).bfloat16()
inner = tr.inner


def one_run():
    tr.reset()
    trainer.optim.zero_grad()
    vq.embed.copy_(vq0[0]); vq.cluster_size.copy_(vq0[1])
    torch.cuda.synchronize()
    if args.noise:
        with torch.cuda.stream(noise_stream):
            for _ in range(40):
                inner.gemm(na, nb_)
    if args.forward_only:
        with torch.no_grad():
            loss = clip(text, video, return_loss=True, device=dev)
    else:
        loss = trainer.forward_backward(video, text)
    torch.cuda.synchronize()
    sums = torch.stack(tr.sums).cpu()
    return float(loss.detach()), sums, list(tr.meta)


for _ in range(2):      # warm-up: lazy shadows, workspaces, allocator pools
    one_run()
loss0, ref, meta0 = one_run()
print(f"lib={os.environ.get('CTCLIP_LIB', 'product')} config={args.config} text_stream={os.environ.get('CTCLIP_TEXT_STREAM', '1')} "
      f"wgrad_stream={os.environ.get('CTCLIP_WGRAD_STREAM', '1')} noise={args.noise}: {ref.numel()} checksums over {meta0[-1][1] + 1} calls per run, "
      f"loss {loss0:.8f}", flush=True)
bad_runs = 0
first_names = {}
for r in range(args.runs):
    loss, sums, meta = one_run()
    if sums.numel() != ref.numel():
        print(f"run {r}: call sequence differs ({sums.numel()} vs {ref.numel()} checksums)", flush=True)
        bad_runs += 1
        continue
    diff = (sums != ref).nonzero().flatten().tolist()
    if diff:
        bad_runs += 1
        m = meta[diff[0]]
        # the first deviating entry that is an OUTPUT whose call's args all agree = the culprit launch; if an arg deviates first, the glue
        key = f"{m[0]}#{m[1]}:{m[2]}{m[3]}"
        first_names[key] = first_names.get(key, 0) + 1
        print(f"run {r}: loss {loss:.8f} (ref {loss0:.8f}), {len(diff)} of {ref.numel()} checksums deviate; first:", flush=True)
        for d in diff[:args.max_report]:
            print(f"     [{d}] {meta[d]}", flush=True)
print("TRACE_SUMMARY " + json.dumps(dict(lib=os.environ.get("CTCLIP_LIB", "product"), config=args.config, runs=args.runs, deviating_runs=bad_runs,
                                         text_stream=os.environ.get("CTCLIP_TEXT_STREAM", "1"), wgrad_stream=os.environ.get("CTCLIP_WGRAD_STREAM", "1"),
                                         noise=args.noise, forward_only=args.forward_only, first_deviation=first_names)), flush=True)
