"""bench.py -- CT-CLIP training-step throughput on MI355X (BASELINE.json metric: CT volumes/s/node at 480x480x240, bs=8/GPU).

    python bench.py --gpus N --steps K --warmup W [--workload train|lipro|vocabfine]
    (N > 1: either launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or run as plain
     `python bench.py --gpus N`: bench.py then re-executes itself under that launcher, one rank per GPU, rank 0 prints the line)

A "step" is one full optimisation step of the hot path (scripts/CTCLIPTrainer.py:233-264): text tower + image tower forward,
gathered-negatives CLIP loss, backward, gradient all-reduce (N > 1), global grad-norm clip, Adam -- on synthetic inputs that are
already resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- the GEMM launch group with the largest total time in the timed steps (main stream), timed with an event pair on the
                  launch stream around every launch (in `--profile-steps` extra steps after the timed region, with the weight-gradient
                  side stream switched off so that one event pair spans one kernel).  Each group is priced against both floors -- algorithmic FLOPs / 2.5 PFLOP/s dense
                  bf16 and algorithmic bytes / 8 TB/s -- and the larger one names its bound: the feed-forward GEMMs with a GEGLU
                  epilogue move 1.0-1.4 GB per launch and are HBM-bound, the plain ones MFMA-bound.  `achieved` / `peak` / `unit` are
                  in the bound's unit, `tflops` and `algorithmic_GBps` give both; top5 = the other groups.  `traffic` = HBM bytes per
                  launch of that launch group MEASURED in this run: after the timed region rank 0 re-runs the one launch under
                  `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, FETCH_SIZE x2 for gfx950 as MI355X_MICROARCH.md
                  prescribes); null with a reason when rocprofv3 is unavailable (--no-pmc skips it).
  attn_block   -- the second half of BASELINE.json's metric ("CTViT MFMA util %"): one spatial attention block (LayerNorm, to_q / to_kv
                  projections, cosine attention with position bias, to_out + residual; attention.py:127-181) at the bench batch, timed
                  with event pairs around the block, forward and forward+backward; FLOPs per SURVEY.md 8(d) (22.65 GF / volume / layer
                  forward, backward = 2x), utilisation against the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline -- the CPU oracle (a restatement of the reference, kind "port") timed on the host cores on a bounded sample: one
                  training step on ONE volume at the bench's own depth (same configuration as `value`), and B=2 at the reference scripts'
                  4+4 layers when the time bound allows.
  reference_depth_4+4 -- the same train step with the reference scripts' own 4+4 transformer layers (run_train.py:17-27), timed after the
                  main configuration (--no-reference-depth skips it).
  text_len_512 -- the same train step at the bench depth with reports padded to 512 tokens, as the reference trainer pads them
                  (CTCLIPTrainer.py:251; BASELINE.json quotes T = 128), timed in a fresh process at N = 1 (as this process's third
                  configuration it varied between 91 and 110 ms from run to run); --no-text512 skips it.
--workload lipro / vocabfine: BASELINE.json configs[4] / configs[3] (one CT-LiPro step at batch 16 with the frozen tower; one VocabFine step =
one volume x 18 prompt pairs), same contract line with `roofline` and a bounded `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FULL = dict(image=480, frames=240, patch=20, tpatch=10, dim=512, heads=8, dim_head=32, codebook=8192, dim_latent=512)


def synth_text(B, T, gen, device, vocab=30522):
    ids = torch.randint(1000, vocab, (B, T), generator=gen)
    lens = torch.randint(T // 2, T + 1, (B,), generator=gen)
    ids[:, 0] = 101
    mask = torch.arange(T)[None, :] < lens[:, None]
    for b in range(B):
        ids[b, lens[b] - 1] = 102
    ids = ids * mask
    return ids.to(device), mask.long().to(device)


class Text:
    def __init__(self, ids, mask):
        self.input_ids, self.attention_mask = ids, mask


def build(args, device, dtype):
    from transformers import BertConfig, BertModel
    import ct_clip_amd
    torch.manual_seed(0)
    enc = ct_clip_amd.CTViT(dim=FULL["dim"], codebook_size=FULL["codebook"], image_size=args.image, patch_size=FULL["patch"],
                            temporal_patch_size=FULL["tpatch"], spatial_depth=args.spatial_depth,
                            temporal_depth=args.temporal_depth, dim_head=FULL["dim_head"], heads=FULL["heads"], compute_dtype=dtype)
    # BERT-base, random init, HF default dropout 0.1 active in train mode as in the reference's training (run_train.py:9)
    bert = BertModel(BertConfig(hidden_dropout_prob=args.bert_dropout, attention_probs_dropout_prob=args.bert_dropout))
    hw = args.image // FULL["patch"]
    clip = ct_clip_amd.CTCLIP(image_encoder=enc, text_encoder=bert, dim_text=768, dim_image=hw * hw * FULL["dim"],
                              dim_latent=FULL["dim_latent"], compute_dtype=dtype)
    # the two never-used *_extra projections (151 M parameters) stay on the host: they receive no gradient
    clip.to(device)
    trainer = ct_clip_amd.CTClipTrainer(clip, num_train_steps=10 ** 9, batch_size=args.batch, tokenizer=object(), lr=1.25e-6,
                                        train_dataset=[0], evaluate=False, checkpoint=False, num_workers=0,
                                        results_folder=os.path.join(ROOT, "gpurun_out", "bench_results"), sync_loss_every=0, device=device)
    return clip, trainer


def algorithmic_flops_per_volume(args, T):
    """SURVEY.md section 8(d) accounting (2*M*N*K per GEMM, backward = 2x forward for differentiable GEMMs)."""
    d, H = FULL["dim"], int(4 * (2 / 3) * FULL["dim"])
    hw = args.image // FULL["patch"]
    t = args.frames // FULL["tpatch"]
    n = t * hw * hw
    K = FULL["patch"] ** 2 * FULL["tpatch"]
    inner = FULL["heads"] * FULL["dim_head"]
    patch = 2 * n * K * d
    proj = 2 * n * d * (inner + 2 * inner) + 2 * n * inner * d
    ff = 2 * n * d * 2 * H + 2 * n * H * d
    peg = 2 * 27 * n * d
    attn_s = 4 * n * (hw * hw) * inner
    attn_t = 4 * n * t * inner
    layers = args.spatial_depth * (proj + ff + peg + attn_s) + args.temporal_depth * (proj + ff + peg + attn_t)
    vq = 2 * n * FULL["codebook"] * d
    vis = 2 * hw * hw * d * FULL["dim_latent"]
    img_fwd = patch + layers + vq + vis
    img_train = img_fwd + 2 * (patch + layers + vis)
    hb, fb, Lb = 768, 3072, 12
    bert_fwd = Lb * (2 * T * hb * 3 * hb + 4 * T * T * hb + 2 * T * hb * hb + 4 * T * hb * fb)
    return img_train + 3 * bert_fwd, img_fwd + bert_fwd


def cpu_baseline(args, T):
    """Time the CPU oracle (oracle/ctclip_oracle.py, a restatement of the reference path): one training step on `cpu_batch` volumes
    with `cpu_spatial_depth`+`cpu_temporal_depth` layers.  Prints one CPU_BASELINE line per finished sample (the parent keeps the
    last complete ones if the wall-clock bound cuts a later sample short)."""
    from oracle import ctclip_oracle as O
    from transformers import BertConfig, BertModel
    import ct_clip_amd
    cores = min(args.cpu_threads, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())
    torch.set_num_threads(cores)     # more threads than ~32 only adds synchronisation overhead to torch's CPU kernels
    hw = args.image // FULL["patch"]

    def sample(nb, sdepth, tdepth):
        torch.manual_seed(0)
        enc = ct_clip_amd.CTViT(dim=FULL["dim"], codebook_size=FULL["codebook"], image_size=args.image, patch_size=FULL["patch"],
                                temporal_patch_size=FULL["tpatch"], spatial_depth=sdepth, temporal_depth=tdepth,
                                dim_head=FULL["dim_head"], heads=FULL["heads"], compute_dtype=torch.float32)
        bert = BertModel(BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
        sd = {"temperature": torch.tensor(1.0)}
        sd.update({"visual_transformer." + k: v for k, v in enc.state_dict().items()})
        sd.update({"text_transformer." + k: v for k, v in bert.state_dict().items()})
        sd["to_text_latent.weight"] = torch.randn(FULL["dim_latent"], 768) * 0.02
        sd["to_visual_latent.weight"] = torch.randn(FULL["dim_latent"], hw * hw * FULL["dim"]) * 0.002
        cfg = O.OracleConfig(dim=FULL["dim"], codebook_size=FULL["codebook"], image_size=args.image, patch_size=FULL["patch"],
                             temporal_patch_size=FULL["tpatch"], spatial_depth=sdepth, temporal_depth=tdepth,
                             dim_head=FULL["dim_head"], heads=FULL["heads"], bert_layers=12, bert_heads=12, dim_latent=FULL["dim_latent"])
        g = torch.Generator().manual_seed(1234)
        video = torch.rand(nb, 1, args.frames, args.image, args.image, generator=g) * 2 - 1
        ids, mask = synth_text(nb, T, g, "cpu")
        t0 = time.time()
        O.train_step_reference(sd, cfg, ids, mask, video)
        dt = time.time() - t0
        return dict(value=round(nb / dt, 5), unit="volumes/s", cores=cores, kind="port", batch=nb, layers=f"{sdepth}+{tdepth}", seconds=round(dt, 1),
                    sample=f"oracle/ctclip_oracle.train_step_reference (fwd+bwd+grad-clip+Adam, f32, torch CPU kernels, {cores} threads) on {nb} "
                           f"volume(s) {args.image}x{args.image}x{args.frames}, {sdepth}+{tdepth} transformer layers, T={T}: {dt:.1f} s wall")
    print("CPU_BASELINE " + json.dumps(dict(tag="bench_depth", **sample(args.cpu_batch, args.spatial_depth, args.temporal_depth))), flush=True)
    if (args.cpu_spatial_depth, args.cpu_temporal_depth) != (args.spatial_depth, args.temporal_depth) or args.cpu_batch != 2:
        print("CPU_BASELINE " + json.dumps(dict(tag="reference_depth_b2", **sample(2, args.cpu_spatial_depth, args.cpu_temporal_depth))), flush=True)


def run_cpu_baseline_bounded(args, sdepth, tdepth):
    """The oracle runs in a child process with a wall-clock bound so that the default bench always finishes in minutes."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--spatial-depth", str(sdepth), "--temporal-depth", str(tdepth),
           "--cpu-spatial-depth", str(args.cpu_spatial_depth), "--cpu-temporal-depth", str(args.cpu_temporal_depth),
           "--cpu-batch", str(args.cpu_batch), "--cpu-threads", str(args.cpu_threads), "--text-len", str(args.text_len),
           "--image", str(args.image), "--frames", str(args.frames)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    out, err, timed_out = "", "", False
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
        out, err = res.stdout, res.stderr
    except subprocess.TimeoutExpired as e:
        timed_out = True
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    samples = {}
    for line in out.splitlines():
        if line.startswith("CPU_BASELINE "):
            rec = json.loads(line[len("CPU_BASELINE "):])
            samples[rec.pop("tag")] = rec
    if "bench_depth" in samples:
        res = samples["bench_depth"]
        res["note"] = ("kind 'port': oracle/ctclip_oracle.py, a CPU restatement of the reference modules (pinned against the real reference by "
                       "tests/golden/*, oracle/gen_golden.py); the reference itself is Python under /root/reference and cannot travel to the GPU box")
        if "reference_depth_b2" in samples:
            res["reference_depth_b2"] = samples["reference_depth_b2"]
        elif timed_out:
            res["reference_depth_b2"] = f"not finished within the {args.cpu_timeout:.0f} s bound"
        return res
    dt = time.time() - t0
    if timed_out:
        return dict(value=None, unit="volumes/s", kind="port", cores=args.cpu_threads, upper_bound=round(args.cpu_batch / dt, 5),
                    sample=f"oracle train step on {args.cpu_batch} volume(s), {sdepth}+{tdepth} layers did not finish within the {args.cpu_timeout:.0f} s "
                           "bound: rate < upper_bound")
    return dict(value=None, unit="volumes/s", kind="port", sample="CPU baseline failed: " + err[-300:])


# ----------------------------------------------------------------------------------------------------------------- PMC companion
def gemm_probe(spec, iters=6):
    """Child mode: run ONE GEMM launch group through the C ABI a few times (under rocprofv3 --pmc).
    spec = 'NT|NN|TN M N K variant' (variant '-' = plain, or the fused GEGLU launches of backend.py's timing keys)."""
    from ct_clip_amd import backend
    be = backend.get()
    layout, M, N, K, variant = (spec.split() + ["-"])[:5]
    M, N, K = int(M), int(N), int(K)
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *sh: (torch.rand(*sh, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    if variant.startswith("+geglu") and variant != "+geglu-bwd" or variant == "geglu-bwd_recompute":
        hp = N // 2
        x, w = rnd(M, K), (torch.rand(2 * hp, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5
        w_il = be.geglu_weight_interleave(w, hp, torch.bfloat16)
        dg = rnd(M, hp)
        fn = {"+geglu": lambda: be.gemm_geglu(x, w_il, hp), "+geglu(g_only)": lambda: be.gemm_geglu(x, w_il, hp, save_u=False),
              "geglu-bwd_recompute": lambda: be.gemm_geglu_bwd(x, w_il, dg, hp)}[variant]
    elif variant == "+geglu-bwd":
        dy, wt, u = rnd(M, K), rnd(N, K), rnd(M, 2 * N)
        fn = lambda: be.gemm_dgeglu(dy, wt, u)
    elif layout == "NT":
        a, b = rnd(M, K), rnd(N, K)
        fn = lambda: be.gemm(a, b)
    elif layout == "NN":
        a, b = rnd(M, K), rnd(K, N)
        fn = lambda: be.gemm(a, b, a_kc=True, b_kc=False)
    else:
        a, b = rnd(K, (M + 7) // 8 * 8), rnd(K, N)
        out = torch.zeros(M, N, device="cuda")
        fn = lambda: be.gemm(a[:, :M], b, a_kc=False, b_kc=False, out=out, accumulate=True, split_k=0, M=M, N=N, K=K)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()


def measure_traffic(spec, timeout=150.0):
    """HBM bytes per launch of the dominant GEMM launch group, from two rocprofv3 --pmc passes over `bench.py --gemm-probe` (rank 0, N = 1)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not spec:
        return None, "dominant kernel is not a GEMM"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    per = {}
    root = tempfile.mkdtemp(prefix="ctclip_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(root, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--gemm-probe", spec]
            try:
                subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd="/tmp")
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} pass exceeded {timeout:.0f} s"
            vals = {}
            for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(path)):
                    # "void (anonymous namespace)::gemm_tn_kernel(...)": drop the namespace before cutting at the argument list
                    kname = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                    if r.get("Counter_Name") == counter and "gemm" in kname and "reduce" not in kname:
                        vals.setdefault(kname, []).append(float(r["Counter_Value"]))
            if not vals:
                return None, f"no {counter} rows for a gemm kernel in the rocprofv3 output"
            name = max(vals, key=lambda k: sum(vals[k]))          # the GEMM kernel proper (not its split-K reduce)
            per[counter] = (name, sum(vals[name]) / len(vals[name]))
        fetch = per["FETCH_SIZE"][1] * 1024 * 2      # KiB units; gfx950 reports half of the bytes of wide coalesced reads
        write = per["WRITE_SIZE"][1] * 1024
        return dict(bytes=fetch + write, fetch_bytes=fetch, write_bytes=write, device_kernel=per["FETCH_SIZE"][0].replace("void ", ""),
                    source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --gemm-probe` in this run; "
                           "FETCH_SIZE x2 (gfx950), units of 1024 B"), None
    finally:
        shutil.rmtree(root, ignore_errors=True)


# ----------------------------------------------------------------------------------------------------------------- attention block
def attention_block_util(args, device, dtype, iters=10):
    """One spatial attention block of the CTViT at the bench batch: x -> x + to_out(attn(to_q(LN(x)), to_kv(x))) with the position
    bias (attention.py:127-181, 322-325), timed with event pairs around the block on the launch stream."""
    import ct_clip_amd
    from ct_clip_amd import functional as Fn
    from ct_clip_amd.ctvit import Attention, ContinuousPositionBias
    torch.manual_seed(0)
    hw = args.image // FULL["patch"]
    t = args.frames // FULL["tpatch"]
    nseq, L = args.batch * t, hw * hw
    attn = Attention(dim=FULL["dim"], dim_head=FULL["dim_head"], heads=FULL["heads"]).to(device)
    cpb = ContinuousPositionBias(dim=FULL["dim"], heads=FULL["heads"]).to(device)
    gd = torch.Generator(device=device).manual_seed(7)
    x = (torch.randn(nseq * L, FULL["dim"], device=device, generator=gd)).to(dtype).requires_grad_(True)
    dy = (torch.randn(nseq * L, FULL["dim"], device=device, generator=gd) * 0.1).to(dtype)

    with torch.no_grad():
        tab0 = cpb(hw, hw)                       # the position-bias MLP runs once per forward for ALL spatial layers (ctvit.py:293)
    tab = tab0.detach().requires_grad_(True)     # ... but its table gradient (dBias) is part of every layer's backward

    # the block's parameters get gradient sinks (views of one flat f32 buffer) as every parameter of the trainer has: their gradients are written by
    # the kernels themselves, not by autograd's AccumulateGrad (~10 fill / add launches of 5 us per backward that the training step does not have)
    from ct_clip_amd.trainer import FusedAdam
    sinks = FusedAdam(list(attn.named_parameters()), lr=0.0)      # noqa: F841  (keeps the flat buffers alive)

    def block():
        xn, x_kv, xr = Fn.layer_norm_branch(x, attn.norm.gamma, None, 2)
        # (the product's own composition, ctvit.Transformer.forward: the projections write the attention operands from their epilogues)
        o = Fn.qkv_attention(xn, x_kv, attn.to_q.weight, attn.to_kv.weight, attn.q_scale, attn.k_scale, tab, nseq, L, attn.heads, attn.dim_head,
                             float(attn.scale), (hw, hw))
        return Fn.linear(o, attn.to_out.weight, residual=xr)
    for _ in range(2):
        block().backward(dy)
        x.grad = None; tab.grad = None
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    for e0, e1, e2 in ev:
        e0.record()
        y = block()
        e1.record()
        y.backward(dy)
        e2.record()
        x.grad = None; tab.grad = None
    torch.cuda.synchronize()
    fwd = sum(a.elapsed_time(b) for a, b, _ in ev) / iters * 1e3
    tot = sum(a.elapsed_time(c) for a, _, c in ev) / iters * 1e3
    inner = FULL["heads"] * FULL["dim_head"]
    n = nseq * L
    flops_fwd = 2.0 * n * FULL["dim"] * 3 * inner + 2.0 * n * inner * FULL["dim"] + 4.0 * n * L * inner     # 22.65 GF / volume
    peak = 2500.0 if dtype == torch.bfloat16 else 157.3
    return dict(what="CTViT spatial attention block (LN + to_q/to_kv + cosine attention with position bias + to_out + residual), "
                     f"{nseq} sequences x {L} tokens, 8 heads x 32", fwd_us=round(fwd, 1), fwd_bwd_us=round(tot, 1),
                gflop_fwd=round(flops_fwd / 1e9, 1), gflop_fwd_bwd=round(3 * flops_fwd / 1e9, 1),
                tflops_fwd=round(flops_fwd / fwd / 1e6, 1), tflops_fwd_bwd=round(3 * flops_fwd / tot / 1e6, 1),
                mfma_util_fwd=round(flops_fwd / fwd / 1e6 / peak, 4), mfma_util_fwd_bwd=round(3 * flops_fwd / tot / 1e6 / peak, 4),
                peak_tflops=peak, note="event pairs on the launch stream; the block's parameters have gradient sinks as in the trainer (round 6); every layout / normalisation kernel between the projections and the "
                                       "attention core and the position-bias table gradient (dBias) are inside; the position-bias MLP itself "
                                       "(once per forward for all layers) is outside")


# ----------------------------------------------------------------------------------------------------------------- fine-tuning workloads
def build_towers(args, device, dtype, dropout):
    from transformers import BertConfig, BertModel
    import ct_clip_amd
    torch.manual_seed(0)
    enc = ct_clip_amd.CTViT(dim=FULL["dim"], codebook_size=FULL["codebook"], image_size=args.image, patch_size=FULL["patch"],
                            temporal_patch_size=FULL["tpatch"], spatial_depth=args.spatial_depth, temporal_depth=args.temporal_depth,
                            dim_head=FULL["dim_head"], heads=FULL["heads"], compute_dtype=dtype)
    bert = BertModel(BertConfig(hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout))
    hw = args.image // FULL["patch"]
    clip = ct_clip_amd.CTCLIP(image_encoder=enc, text_encoder=bert, dim_text=768, dim_image=hw * hw * FULL["dim"], dim_latent=FULL["dim_latent"],
                              compute_dtype=dtype)
    return clip.to(device)


def finetune_setup(args, device, dtype, rank):
    """-> (step function, units per step, description).  lipro: scripts/ct_lipro_train.py:92-107 (frozen CT-CLIP in train mode -> image
    latents -> ReLU -> Dropout(0.3) -> Linear(512, 18), BCEWithLogits(pos_weight), clip 1.0, AdamW + cosine schedule; the reference also
    runs the unused text tower on " " -- skipped here, the logits do not depend on it).  vocabfine: scripts/ct_vocabfine_train.py:77-123
    (18 pathologies x (true, false) prompt, similarity of both with the volume = one full CTCLIP forward each, softmax-MSE per group of six,
    three backwards, one AdamW step per volume, every parameter trains)."""
    from ct_clip_amd import finetune as FT
    gd = torch.Generator(device=device).manual_seed(1234 + rank)
    g = torch.Generator().manual_seed(1234 + rank)
    video = torch.rand(args.batch, 1, args.frames, args.image, args.image, generator=gd, device=device) * 2 - 1
    if args.workload == "lipro":
        clip = build_towers(args, device, dtype, 0.0)
        head = FT.ImageLatentsClassifier(clip, FULL["dim_latent"], 18).to(device)
        trainer = FT.LiProTrainer(head, lr=1e-5, wd=0.1, warmup_length=500, total_steps=10 ** 6)
        labels = (torch.rand(args.batch, 18, generator=g) < 0.3).float().to(device)
        blank = Text(*synth_text(1, args.text_len, g, device))

        def step():
            return trainer.train_step(blank, video, labels)[0]
        what = (f"CT-LiPro / ClassFine step (ct_lipro_train.py:92-107): frozen CTViT {args.image}x{args.image}x{args.frames} "
                f"{args.spatial_depth}+{args.temporal_depth} layers in train mode (VQ EMA on) -> image latents -> ReLU/Dropout(0.3)/Linear(512,18), "
                f"BCEWithLogits(pos_weight), clip 1.0, AdamW on the 9 234 head parameters; batch {args.batch}/GPU; text tower skipped")
        return step, args.batch, what, trainer
    clip = build_towers(args, device, dtype, args.bert_dropout)
    clip.train()
    npath = len(FT.PATHOLOGIES)
    pairs = [Text(*synth_text(2, args.text_len, g, device)) for _ in range(npath)]      # synthetic token ids stand in for the tokenizer
    trainer = FT.VocabFineTrainer(clip, tokenize=None, lr=1e-5, wd=0.1, warmup_length=500, total_steps=10 ** 6)
    vol = video[:1]

    def step():
        trainer.scheduler(trainer.step)
        losses, _ = trainer.forward_backward(vol, pairs, fused=not args.vocabfine_literal)
        trainer.optim.step(None)
        trainer.step += 1
        return losses[-1]
    what = (f"VocabFine step (ct_vocabfine_train.py:77-123): one volume {args.image}x{args.image}x{args.frames}, {npath} pathologies x (true, false) prompt "
            f"(T={args.text_len}), softmax-MSE per group of 6, AdamW on every parameter (end-to-end), {args.spatial_depth}+{args.temporal_depth} layers + BERT-base; "
            + ("the reference's loop literally: 18 full CTCLIP forwards, 3 backwards" if args.vocabfine_literal else
               "same numbers from ONE pass of each tower per volume (the 18 forwards share weights and volume: image transformer once, 36 prompts as one "
               "BERT batch, the quantiser's EMA sequence + pooling + latents re-applied per pathology, one backward of the summed group losses)"))
    return step, 1, what, trainer


def finetune_bench(args, device, dtype, world, rank):
    from ct_clip_amd import backend
    be = backend.get()
    step, units, what, trainer = finetune_setup(args, device, dtype, rank)
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if trainer.reducer is not None:
        trainer.reducer.start_timing()            # the trainers of ct_clip_amd/finetune.py all-reduce (mean) their gradients when a process group is up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    comm = trainer.reducer.stop_timing() if trainer.reducer is not None else None
    timing = None
    if args.profile_steps > 0:
        be.start_gemm_timing()
        for _ in range(min(args.profile_steps, 3)):
            step()
        timing = be.stop_gemm_timing()
        timing["timed_in"] = "extra steps after the timed region, event pair around every GEMM launch on its stream"
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    cfgi = 4 if args.workload == "lipro" else 3
    out = {"metric": f"CT volumes/sec/node, BASELINE.json configs[{cfgi}] ({'ClassFine / CT-LiPro' if cfgi == 4 else 'VocabFine'} step)",
           "value": round(world * units * args.steps / dt, 3), "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
           "data": "synthetic (uniform [-1,1] volumes generated on device, random token ids, random labels; random-init weights)",
           "config": {"workload": what, "global_batch": world * units, "text_len": args.text_len, "parallelism": f"dp{world}",
                      "layers": f"{args.spatial_depth}+{args.temporal_depth}"},
           "loss": round(float(loss), 5), "peak_mem_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1)}
    if trainer.reducer is not None:
        out["config"]["note"] = ("one process per GPU, every rank on its own volumes; the trainable parameters' gradients are all-reduced (mean) before the clip "
                                 "and the AdamW step and the quantiser's EMA statistics are all-reduced (sum) -- the reference's nn.DataParallel "
                                 "(ct_lipro_train.py:75, ct_vocabfine_train.py:62) as data-parallel ranks")
        if comm:
            out["comm"] = dict(comm, backend=dist.get_backend(), op="mean",
                               reduced=("the 9 234 head parameters (one collective)" if args.workload == "lipro" else "every trainable parameter, bucketed under backward"))
    if timing:
        out["roofline"] = timing
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_finetune_cpu_bounded(args)
        emit_line(out)


def finetune_cpu_worker(args):
    """Child process: the CPU oracle (a restatement of the reference modules, kind "port") on a bounded sample of the fine-tuning step."""
    from oracle import ctclip_oracle as O
    from transformers import BertConfig, BertModel
    import torch.nn.functional as F
    import ct_clip_amd
    cores = min(args.cpu_threads, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())
    torch.set_num_threads(cores)
    hw = args.image // FULL["patch"]
    torch.manual_seed(0)
    enc = ct_clip_amd.CTViT(dim=FULL["dim"], codebook_size=FULL["codebook"], image_size=args.image, patch_size=FULL["patch"],
                            temporal_patch_size=FULL["tpatch"], spatial_depth=args.spatial_depth, temporal_depth=args.temporal_depth,
                            dim_head=FULL["dim_head"], heads=FULL["heads"], compute_dtype=torch.float32)
    bert = BertModel(BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    sd = {"temperature": torch.tensor(1.0)}
    sd.update({"visual_transformer." + k: v for k, v in enc.state_dict().items()})
    sd.update({"text_transformer." + k: v for k, v in bert.state_dict().items()})
    sd["to_text_latent.weight"] = torch.randn(FULL["dim_latent"], 768) * 0.02
    sd["to_visual_latent.weight"] = torch.randn(FULL["dim_latent"], hw * hw * FULL["dim"]) * 0.002
    cfg = O.OracleConfig(dim=FULL["dim"], codebook_size=FULL["codebook"], image_size=args.image, patch_size=FULL["patch"],
                         temporal_patch_size=FULL["tpatch"], spatial_depth=args.spatial_depth, temporal_depth=args.temporal_depth,
                         dim_head=FULL["dim_head"], heads=FULL["heads"], bert_layers=12, bert_heads=12, dim_latent=FULL["dim_latent"])
    g = torch.Generator().manual_seed(1234)
    video = torch.rand(1, 1, args.frames, args.image, args.image, generator=g) * 2 - 1
    t0 = time.time()
    if args.finetune_cpu_worker == "lipro":
        with torch.no_grad():
            toks, _ = O.ctvit_forward(sd, cfg, video, training=True)
            lat = O.l2norm(F.linear(toks.mean(dim=1).reshape(1, -1), sd["to_visual_latent.weight"]))
        W = (torch.randn(18, FULL["dim_latent"]) * 0.05).requires_grad_(True)
        b = torch.zeros(18, requires_grad=True)
        loss = F.binary_cross_entropy_with_logits(F.linear(F.relu(lat), W, b), torch.zeros(1, 18), pos_weight=torch.ones(18))
        loss.backward()
        dt = time.time() - t0
        rec = dict(value=round(1 / dt, 5), unit="volumes/s", cores=cores, kind="port", seconds=round(dt, 1),
                   sample=f"oracle image tower forward (train mode) + head forward/backward on ONE volume, {args.spatial_depth}+{args.temporal_depth} layers, "
                          f"f32, torch CPU kernels, {cores} threads: {dt:.1f} s (the step is linear in the batch)")
    else:
        ids, mask = synth_text(2, args.text_len, g, "cpu")
        leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "vq._codebook" not in k and not k.endswith(".beta")
                                                       and "position_ids" not in k) for k, v in sd.items()}
        enc_text = O.bert_forward(leaves, cfg, ids, mask)
        toks, _ = O.ctvit_forward(leaves, cfg, video, training=True)
        tl = O.l2norm(F.linear(enc_text[:, 0, :], leaves["to_text_latent.weight"]))
        il = O.l2norm(F.linear(toks.mean(dim=1).reshape(1, -1), leaves["to_visual_latent.weight"]))
        sim = O.similarity_no_loss(tl, il, leaves["temperature"])
        loss = F.mse_loss(F.softmax(sim, dim=0), torch.tensor([1.0, 0.0]))
        loss.backward()
        dt = time.time() - t0
        rec = dict(value=round(1 / (18 * dt), 5), unit="volumes/s", cores=cores, kind="port", seconds=round(dt, 1),
                   sample=f"oracle forward + backward of ONE of the 18 prompt pairs of one volume ({args.spatial_depth}+{args.temporal_depth} layers + BERT-base, "
                          f"T={args.text_len}, f32, {cores} threads): {dt:.1f} s; value = 1 / (18 x that) -- a step is 18 such forwards and their backwards")
    print("CPU_BASELINE " + json.dumps(rec), flush=True)


def run_finetune_cpu_bounded(args):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--finetune-cpu-worker", args.workload,
           "--spatial-depth", str(args.spatial_depth), "--temporal-depth", str(args.temporal_depth), "--cpu-threads", str(args.cpu_threads),
           "--text-len", str(args.text_len), "--image", str(args.image), "--frames", str(args.frames)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="volumes/s", kind="port", cores=args.cpu_threads, sample=f"CPU sample did not finish within {args.cpu_timeout:.0f} s")
    for line in res.stdout.splitlines():
        if line.startswith("CPU_BASELINE "):
            return json.loads(line[len("CPU_BASELINE "):])
    return dict(value=None, unit="volumes/s", kind="port", sample="CPU baseline failed: " + res.stderr[-300:])


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: run this same command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` and pass the ranks'
    output through (rank 0 prints the one JSON line).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


_LINE_FD = None


def emit_line(obj):
    """The contract line: to the process's ORIGINAL stdout (see main())."""
    data = (json.dumps(obj) + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_LINE_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="volumes per GPU")
    ap.add_argument("--text-len", type=int, default=128)
    ap.add_argument("--spatial-depth", type=int, default=12, help="12+12 = the '24 layers' BASELINE.json names; reference scripts use 4+4")
    ap.add_argument("--temporal-depth", type=int, default=12)
    ap.add_argument("--image", type=int, default=FULL["image"])
    ap.add_argument("--frames", type=int, default=FULL["frames"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--bert-dropout", type=float, default=0.1, help="HF BertConfig hidden / attention dropout (reference default 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=1)
    ap.add_argument("--cpu-spatial-depth", type=int, default=4, help="the CPU sample uses the reference's own 4+4 layers to stay bounded")
    ap.add_argument("--cpu-temporal-depth", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--cpu-timeout", type=float, default=150.0, help="wall-clock bound (s) for the CPU baseline subprocess")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc companion passes that measure roofline.traffic")
    ap.add_argument("--no-attn-block", action="store_true", help="skip the attention-block MFMA utilisation measurement")
    ap.add_argument("--profile-steps", type=int, default=10, help="extra steps after the timed region in which every GEMM launch is event-timed (roofline)")
    ap.add_argument("--comm-sweep-steps", type=int, default=3, help="N > 1: steps per row of the wire-format x bucket-size sweep printed as `comm_sweep` (0 = off)")
    ap.add_argument("--gemm-probe", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--also-reference-depth", action="store_true", help=argparse.SUPPRESS)      # (default on since round 3; kept for old command lines)
    ap.add_argument("--no-reference-depth", action="store_true", help="skip the second timed configuration (reference scripts' 4+4 layers)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the optimisation step into a hipGraph after the warm-up steps and time replays (N = 1 only; falls back to eager "
                         "launches if the capture fails); the line says which in config.launch")
    ap.add_argument("--with-input-pipeline", action="store_true",
                    help="feed every step through the device-side input pipeline (pinned int16 ring + copy stream + preprocess kernel, SURVEY 8(f)3); "
                         "adds an `input_pipeline` object with that configuration's volumes/s (host decode of the NIfTI file excluded)")
    ap.add_argument("--no-text512", action="store_true", help="skip the extra timed configuration with reports padded to 512 tokens (CTCLIPTrainer.py:251)")
    ap.add_argument("--input-dist", default="uniform", choices=["uniform", "normal"],
                    help="synthetic voxels: uniform [-1, 1] (default) or N(0, 0.3) clamped to [-1, 1] (SURVEY.md 8d: a second distribution, for DVFS honesty)")
    ap.add_argument("--workload", default="train", choices=["train", "lipro", "vocabfine"],
                    help="train = BASELINE.json configs[1]/[2] (default); lipro = configs[4] (ClassFine / CT-LiPro step, frozen tower, batch 16); "
                         "vocabfine = configs[3] (VocabFine step: 18 prompt pairs per volume)")
    ap.add_argument("--vocabfine-literal", action="store_true", help="vocabfine: the reference's loop literally (18 image-tower passes per volume) "
                                                                       "instead of one pass of each tower per volume")
    ap.add_argument("--finetune-cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.with_input_pipeline:      # one configuration only: the pipeline's pinned ring and batch buffers are built per run_config
        args.no_reference_depth = args.no_text512 = True
    if args.workload != "train":      # the fine-tuning scripts build the towers with 4+4 layers (ct_lipro_train.py:47-51, ct_vocabfine_train.py:29-33)
        if "--spatial-depth" not in sys.argv:
            args.spatial_depth = 4
        if "--temporal-depth" not in sys.argv:
            args.temporal_depth = 4
        if "--batch" not in sys.argv:
            args.batch = 16 if args.workload == "lipro" else 1
        if "--steps" not in sys.argv:
            args.steps = 20 if args.workload == "lipro" else 5
    if args.finetune_cpu_worker:
        finetune_cpu_worker(args)
        return

    if args.cpu_baseline_worker:      # child process: time the oracle and print its JSON lines
        cpu_baseline(args, args.text_len)
        return
    if args.gemm_probe:               # child process under rocprofv3 --pmc
        gemm_probe(args.gemm_probe)
        return

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio on stdout when its first communicator comes up
    # (seen on the MI355X box: "RCCL version : 2.26.6 ... Librccl path : ..."), and a library may do the same tomorrow: from here on file descriptor 1
    # is the process's stderr, and the line is written to the saved descriptor by emit_line().  (The worker modes above keep their stdout:
    # their parent reads it through a pipe, and the self-launching parent hands its stdout to the ranks.)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (rank 0 prints the line)
        sys.exit(self_launch(args.gpus))
    global _LINE_FD
    sys.stdout.flush()
    _LINE_FD = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # CTCLIP_BENCH_BACKEND=gloo CTCLIP_BENCH_SINGLE_DEVICE=1: rehearsal of the multi-rank code path on a ONE-GPU box (all ranks on
    # cuda:0, collectives staged through the host) -- used to check the N > 1 path where no multi-GPU node is available
    backend_name = os.environ.get("CTCLIP_BENCH_BACKEND", "nccl")
    if os.environ.get("CTCLIP_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # CTCLIP_DIST_SINGLE_RANK=1 with --gpus 1: a process group of ONE rank, under which the data-parallel branches are taken and every collective
    # is issued (each the identity) -- RCCL's kernels, the communication stream and the `comm` diagnostics on a 1-GPU box; NOT the headline run
    single_rank_dp = world == 1 and os.environ.get("CTCLIP_DIST_SINGLE_RANK", "") == "1"
    if single_rank_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if (world > 1 or single_rank_dp) and not dist.is_initialized():
        if backend_name == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend_name)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    from ct_clip_amd import backend
    be = backend.get()
    if args.workload != "train":
        finetune_bench(args, device, dtype, world, rank)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    def run_config(sdepth, tdepth, steps, warmup, profile_gemm):
        args.spatial_depth, args.temporal_depth = sdepth, tdepth
        clip, trainer = build(args, device, dtype)
        clip.train()
        g = torch.Generator().manual_seed(1234 + rank)
        gd = torch.Generator(device=device).manual_seed(1234 + rank)
        if args.input_dist == "normal":
            video = (torch.randn(args.batch, 1, args.frames, args.image, args.image, generator=gd, device=device) * 0.3).clamp_(-1, 1)
        else:
            video = torch.rand(args.batch, 1, args.frames, args.image, args.image, generator=gd, device=device) * 2 - 1
        ids, mask = synth_text(args.batch, args.text_len, g, device)
        text = Text(ids, mask)

        pipe = None
        if getattr(args, "with_input_pipeline", False):
            # every step's volumes come through the device-side input pipeline -- int16 voxels (512 x 512 x 300 as stored: 157 MB) in
            # PINNED host memory -> two-slot ring -> H2D on a copy stream -> preprocess kernel writing the (1, 240, 480, 480) f32 model input straight
            # into the next step's batch buffer, all under the current step.  The host decode of the gz NIfTI file is NOT included (synthetic voxels
            # are staged once); scripts/data.py:92-162 is the CPU path this replaces.
            from ct_clip_amd import preprocess as PP
            shape = (512, 512, 300)
            up = PP.VolumeUploader(device, max_voxels=shape[0] * shape[1] * shape[2], slots=2, target_shape=(args.image, args.image, args.frames))
            gh = torch.Generator().manual_seed(77 + rank)
            for sl in range(2):
                up.host_buffer(sl).copy_(torch.randint(-1000, 1000, (up.max_voxels,), generator=gh, dtype=torch.int16))
            bufs = [torch.empty_like(video), torch.empty_like(video)]
            used = [None, None]                                   # event: the main stream is done with this batch buffer

            def fill(which):
                if used[which] is not None:
                    up.stream.wait_event(used[which])
                # (source spacing chosen so that the 512 x 512 x 300 voxels resample to exactly image x image x frames at 0.75 / 0.75 / 1.5 mm)
                return [up.submit(None, 1.0, 0.0, 0.75 * args.image / shape[0], 1.5 * args.frames / shape[2], out=bufs[which][b], staged=shape)
                        for b in range(args.batch)]
            pipe = {"k": 0, "tickets": fill(0)}

        def step():
            vid = video
            if pipe is not None:
                cur = pipe["k"] & 1
                for t_ in pipe["tickets"]:
                    up.result(t_)
                vid = bufs[cur]
                pipe["tickets"] = fill(cur ^ 1)                  # the NEXT step's batch goes up under this step
            loss = trainer.forward_backward(vid, text)
            if os.environ.get("CTCLIP_ADAM_ZERO", "1") != "0":      # as CTClipTrainer.train_step: Adam clears the gradients it has just read
                trainer.optim.step(trainer.max_grad_norm, zero_grad=True)
            else:
                trainer.optim.step(trainer.max_grad_norm)
                trainer.optim.zero_grad(overlap=os.environ.get("CTCLIP_ZERO_OVERLAP", "1") != "0")
            if pipe is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
                used[pipe["k"] & 1] = ev
                pipe["k"] += 1
            return loss

        for _ in range(warmup):
            loss = step()
        graphed = None
        if getattr(args, "graph", False) and world == 1 and pipe is None:
            # the whole optimisation step captured once into a hipGraph and replayed (ct_clip_amd.trainer.GraphedStep): the device work is
            # the same kernels in the same order; what changes is the host's share (~1 900 launches through Python + ctypes per step)
            from ct_clip_amd.trainer import GraphedStep
            eager_step = step
            graphed = GraphedStep(trainer)         # (constructed BEFORE the capture so that close() is reachable when it fails)
            try:
                graphed.capture(video, text)
                step = lambda: graphed.run()      # noqa: E731
                step()
            except Exception as e:                # never lose the measurement to the capture: eager steps are always valid
                print(f"bench: hipGraph capture failed, staying eager: {e!r}", file=sys.stderr)
                graphed.close()
                graphed, step = None, eager_step
        torch.cuda.synchronize()
        if world > 1 or single_rank_dp:
            dist.barrier()
            trainer.reducer.start_timing()        # event pairs on the communication stream (no host synchronisation): `comm` of the JSON line
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        run_config.comm = trainer.reducer.stop_timing() if (world > 1 or single_rank_dp) else None
        run_config.comm_sweep = None
        if (world > 1 or single_rank_dp) and profile_gemm and args.comm_sweep_steps > 0 and graphed is None:
            # The first multi-GPU record must be able to DECIDE the reducer's defaults: the same step re-timed, back to back in this process, with
            # f32 and bf16 buckets at two bucket thresholds (the timed region above ran the first row's settings unless overridden by the
            # environment).  Per row: one untimed step, then `--comm-sweep-steps` steps between barriers, MAX over ranks.
            sweep = []
            red = trainer.reducer
            default = (red.comm_dtype, red.min_elems * 4)
            for wire, bucket in ((torch.float32, 16 << 20), (torch.bfloat16, 16 << 20), (torch.float32, 64 << 20), (torch.bfloat16, 64 << 20)):
                red.reconfigure(wire, bucket)
                step()
                torch.cuda.synchronize()
                dist.barrier()
                red.start_timing()
                torch.cuda.synchronize()
                s0 = time.perf_counter()
                for _ in range(args.comm_sweep_steps):
                    step()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                ts = torch.tensor([time.perf_counter() - s0], device=device, dtype=torch.float64)
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
                row = red.stop_timing() or {}
                row["ms_per_step"] = round(float(ts[0]) / args.comm_sweep_steps * 1e3, 3)
                row["steps"] = args.comm_sweep_steps
                sweep.append(row)
            red.reconfigure(*default)
            run_config.comm_sweep = sweep
        run_config.graphed = graphed is not None
        if graphed is not None:
            graphed.close()
            args.profile_steps = 0                # (the per-launch GEMM timing needs eager launches)
        timing = None
        if profile_gemm and args.profile_steps > 0:
            # Per-launch durations for `roofline`: the timed steps run the weight-gradient GEMMs on a side stream UNDER the grad-input
            # GEMMs (functional.wgrad_stream_begin), so an event pair around one launch there spans two kernels sharing the chip.  The
            # launches are therefore timed in extra steps AFTER the timed region with that overlap switched off (same kernels, same
            # shapes, one stream): kernel efficiency, not the schedule, is what the roofline fraction states.
            prev = os.environ.get("CTCLIP_WGRAD_STREAM")
            os.environ["CTCLIP_WGRAD_STREAM"] = "0"
            try:
                step()
                torch.cuda.synchronize()
                be.start_gemm_timing()
                for _ in range(args.profile_steps):
                    step()
                timing = be.stop_gemm_timing()
                timing["timed_in"] = (f"{args.profile_steps} extra steps after the timed region with the weight-gradient stream off "
                                      "(CTCLIP_WGRAD_STREAM=0: launches serialised on one stream)")
            finally:
                if prev is None:
                    del os.environ["CTCLIP_WGRAD_STREAM"]
                else:
                    os.environ["CTCLIP_WGRAD_STREAM"] = prev
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lossv = float(loss)
        peak_mem = torch.cuda.max_memory_allocated(device) / 2 ** 30
        del clip, trainer, video
        torch.cuda.empty_cache()
        return float(t[0]), lossv, timing, peak_mem

    dt, lossv, timing, peak_mem = run_config(args.spatial_depth, args.temporal_depth, args.steps, args.warmup, True)
    main_graphed = getattr(run_config, "graphed", False)
    sdepth, tdepth = args.spatial_depth, args.temporal_depth
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    train_flops, _ = algorithmic_flops_per_volume(args, args.text_len)

    out = {
        "metric": "CT volumes/sec/node (480x480x240, bs=8/GPU), full training step + CTViT MFMA util % (attn_block)",
        "value": round(value, 3), "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": ("synthetic (uniform [-1,1] volumes generated on device, random token ids; random-init weights)" if args.input_dist == "uniform"
                                      else "synthetic (N(0, 0.3) voxels clamped to [-1,1], generated on device, random token ids; random-init weights)"),
        "config": {"workload": f"CT-CLIP train step: CTViT {args.image}x{args.image}x{args.frames} patch 20x20x10 dim 512 "
                               f"{sdepth}+{tdepth} layers + BERT-base T={args.text_len}, batch {args.batch}/GPU, global batch {world * args.batch}, "
                               "gathered-negatives InfoNCE, grad clip 0.5, Adam",
                   "global_batch": world * args.batch, "text_len": args.text_len, "parallelism": f"dp{world}",
                   "layers": f"{sdepth}+{tdepth}",
                   "launch": "hipGraph replay of the captured step (ct_clip_amd.trainer.GraphedStep)" if main_graphed else "eager launches (Python + ctypes per kernel)"},
        "loss": round(lossv, 5),
        "model_tflops_per_step_per_gpu": round(train_flops * args.batch / 1e12, 2),
        "model_tflops_per_s_per_gpu": round(train_flops * args.batch / 1e12 / (ms / 1e3), 1),
        "mfma_fraction_whole_step": round(train_flops * args.batch / (ms / 1e3) / 2.5e15, 4),
        "peak_mem_gib": round(peak_mem, 1),
    }
    from ct_clip_amd import streams as _streams
    out["side_streams"] = _streams.report()      # how each side stream was found (probed for running beside the default stream: ct_clip_amd/streams.py)
    if getattr(run_config, "comm", None):
        # rank 0's view of the gradient all-reduce: how long the buckets occupied the communication stream and how much of that the compute
        # stream had to WAIT for at the end of backward (the rest was hidden under backward); diagnoses the first multi-GPU run
        out["comm"] = dict(run_config.comm, backend=dist.get_backend(), vq_stats="fused buffer on the communication stream, EMA applied at finish()")
        if single_rank_dp:
            out["comm"]["note"] = "ONE rank (CTCLIP_DIST_SINGLE_RANK=1): every collective issued, each the identity -- stream plumbing and RCCL launch cost, no xGMI traffic"
        if getattr(run_config, "comm_sweep", None):
            out["comm_sweep"] = run_config.comm_sweep
    if timing:
        timing["traffic"] = None
        if rank == 0 and world == 1 and not args.no_pmc:
            tr, why = measure_traffic(timing.get("probe_spec"))
            if tr is not None:
                timing["traffic"] = tr.pop("bytes")
                timing["traffic_detail"] = tr
                timing["traffic_over_algorithmic"] = round(timing["traffic"] / timing["algorithmic_bytes_per_launch"], 3)
            else:
                timing["traffic_note"] = why
        elif world > 1:
            timing["traffic_note"] = "measured at N = 1 only"
        out["roofline"] = timing
    if not args.no_attn_block and rank == 0:
        try:
            out["attn_block"] = attention_block_util(args, device, dtype)
        except Exception as e:      # never lose the headline number to the auxiliary measurement
            out["attn_block"] = {"error": repr(e)[:300]}
    if not args.no_reference_depth and (sdepth, tdepth) != (4, 4):
        # the reference's own scripts build 4+4 layers (run_train.py:17-27): the same step at that depth, same batch, fewer timed steps
        dt2, loss2, _, _ = run_config(4, 4, max(5, args.steps // 3), max(1, args.warmup), False)
        args.steps_ref = max(5, args.steps // 3)
        out["reference_depth_4+4"] = {"value": round(world * args.batch * args.steps_ref / dt2, 3), "unit": "volumes/s", "steps": args.steps_ref,
                                      "ms_per_step": round(dt2 / args.steps_ref * 1e3, 3), "loss": round(loss2, 5),
                                      "workload": "the same train step with the reference scripts' 4+4 transformer layers (run_train.py:17-27)",
                                      "launch": "hipGraph replay" if getattr(run_config, "graphed", False) else "eager"}
    if args.with_input_pipeline:
        out["input_pipeline"] = {"value": out["value"], "unit": "volumes/s", "ms_per_step": out["ms_per_step"],
                                 "note": "THIS line's steps took their volumes through the input pipeline: 8 x 157 MB of int16 voxels (512 x 512 x 300) per step "
                                         "from pinned host memory through a two-slot ring, H2D + preprocess kernel (rescale, trilinear resample to 0.75 / 0.75 / 1.5 mm, "
                                         "clip, crop / pad) on a copy stream under the previous step; the gz-NIfTI decode on the host is NOT included "
                                         "(scripts/data.py:92-162 is the CPU path replaced)"}
    if not args.no_text512 and args.text_len != 512:
        # the reference trainer pads every report to 512 tokens (scripts/CTCLIPTrainer.py:251: max_length=512); BASELINE.json quotes T = 128.
        # The same step (bench depth) with T = 512, fewer timed steps.
        keep = args.text_len
        args.text_len = 512
        try:
            n3 = max(5, args.steps // 3)
            what = (f"the same train step ({sdepth}+{tdepth} layers) with reports padded to 512 tokens as the reference trainer pads them "
                    "(CTCLIPTrainer.py:251)")
            if world == 1:
                # In a FRESH process (as the CPU baseline and the counter passes are): as the third configuration of this one the T = 512 step came
                # out anywhere between 91 and 110 ms from run to run (its text tower is 50 ms of side-stream kernels whose overlap with the image
                # tower depended on what the process had done before); as the first configuration of a process it is 91.1-91.5 ms every time.
                import gc
                import subprocess
                gc.collect()
                torch.cuda.empty_cache()
                cmd = [sys.executable, os.path.abspath(__file__), "--text-len", "512", "--steps", str(n3), "--warmup", str(max(1, args.warmup)),
                       "--batch", str(args.batch), "--spatial-depth", str(sdepth), "--temporal-depth", str(tdepth), "--image", str(args.image),
                       "--frames", str(args.frames), "--dtype", args.dtype, "--bert-dropout", str(args.bert_dropout), "--input-dist", args.input_dist,
                       "--no-cpu-baseline", "--no-pmc", "--no-attn-block", "--no-reference-depth", "--no-text512", "--profile-steps", "0"]
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
                sub = json.loads(r.stdout.strip().splitlines()[-1])
                out["text_len_512"] = {"value": sub["value"], "unit": "volumes/s", "steps": n3, "ms_per_step": sub["ms_per_step"], "loss": sub["loss"],
                                       "workload": what, "timed_in": "a fresh process (python bench.py --text-len 512 ...), after this one's configurations"}
            else:
                dt3, loss3, _, _ = run_config(sdepth, tdepth, n3, max(1, args.warmup), False)
                out["text_len_512"] = {"value": round(world * args.batch * n3 / dt3, 3), "unit": "volumes/s", "steps": n3,
                                       "ms_per_step": round(dt3 / n3 * 1e3, 3), "loss": round(loss3, 5), "workload": what}
        except Exception as e:      # auxiliary measurement: never lose the headline line to it
            out["text_len_512"] = {"error": repr(e)[:300]}
        finally:
            args.text_len = keep
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline_bounded(args, sdepth, tdepth)
        emit_line(out)
    if world > 1 or single_rank_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
