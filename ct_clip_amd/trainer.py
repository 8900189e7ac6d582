"""CTClipTrainer (drop-in for scripts/CTCLIPTrainer.py:113-348) for one-process-per-GPU training on MI355X.

Same constructor keywords and methods (``train``, ``train_step``, ``save``, ``load``, ``print``, ``is_main``,
buffer ``steps``).  Differences, all additive and documented in INTEGRATION.md:
  * no Accelerate: the process group is ``torch.distributed`` (RCCL) initialised from RANK/WORLD_SIZE when present;
  * ``train_dataset`` / ``valid_dataset`` / ``evaluate`` / ``checkpoint`` keywords let a caller inject data and switch off
    the every-step evaluation + 1.75 GB checkpoint that dominate the reference loop (CTCLIPTrainer.py:266-337);
  * the optimiser is the fused HIP Adam over one flat f32 buffer (grad-norm clip 0.5 + Adam(0.9, 0.99), optimizer.py:24).
"""
import os
from pathlib import Path
from shutil import rmtree

import torch
from torch import nn
from torch.utils.data import DataLoader

from . import backend as _be
from . import distributed as _dist
from . import functional as Fn

UNUSED_PARAM_MARKERS = ("_extra.", "to_pixels", "to_patch_emb_first_frame", ".pooler.", "context_norm.", "null_kv")


def exists(v):
    return v is not None


def noop(*a, **k):
    pass


def cycle(dl, sampler=None):
    epoch = 0
    while True:
        if sampler is not None:
            sampler.set_epoch(epoch)
        for data in dl:
            yield data
        epoch += 1


def hot_path_parameters(model):
    """Parameters that receive gradients on the CT-CLIP path (the rest never do: SURVEY.md section 2 collective table)."""
    return [(n, p) for n, p in model.named_parameters()
            if p.requires_grad and not any(m in n for m in UNUSED_PARAM_MARKERS)]


class FusedAdam:
    """Flat-buffer Adam/AdamW + global grad-norm clip on HIP kernels; minimal torch.optim-like surface.

    weight_decay > 0 is AdamW with the reference's grouping (transformer_maskgit/optimizer.py:3-8,27-32): parameters with
    ndim < 2 (LayerNorm gammas, biases, q/k scales, temperature) are not decayed (`group_wd_params`)."""

    def __init__(self, named_params, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, group_wd_params=True):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]     # keep every view 16-byte aligned
        total = sum(sizes)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p, sz in zip(self.params, sizes):
                view = self.flat_param[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                g = self.flat_grad[off:off + p.numel()].view(p.shape)
                p._ctclip_grad_sink = g
                p.grad = g
                self.offsets.append(off)
                off += sz
        self.decay_mask4 = None
        if weight_decay and group_wd_params:
            mask = torch.zeros(total // 4, dtype=torch.uint8)
            for p, off in zip(self.params, self.offsets):
                if p.ndim >= 2:
                    mask[off // 4:(off + p.numel() + 3) // 4] = 1
            self.decay_mask4 = mask.to(dev)
        wd_params = [p for p in self.params if p.ndim >= 2] if (weight_decay and group_wd_params) else self.params
        no_wd = [p for p in self.params if p.ndim < 2] if (weight_decay and group_wd_params) else []
        self.param_groups = [dict(params=wd_params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        if no_wd:
            self.param_groups.append(dict(params=no_wd, lr=lr, betas=betas, eps=eps, weight_decay=0.0))
        self.step_count = 0
        self.last_norm = None
        Fn.bump_weight_epoch()

    def range_of(self, params):
        """[start, end) of the flat buffers covered by `params` (must be a contiguous run in registration order)."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx:
            return None
        assert idx == list(range(idx[0], idx[-1] + 1)), "parameters of a gradient bucket must be contiguous in the flat buffer"
        last = idx[-1]
        return self.offsets[idx[0]], self.offsets[last] + (self.params[last].numel() + 3) // 4 * 4

    def zero_grad(self, set_to_none=False, overlap=False):
        """overlap=True (the trainer's step): the 1.1-GB fill runs on a side stream behind the optimiser step, UNDER the next step's forward
        (nothing writes a gradient before the next backward); `wait_zero()` -- called by the trainer right before backward -- orders the
        backward behind it.  Default: on the caller's stream, complete in stream order."""
        self._mark_fresh()
        if not (overlap and self.flat_grad.is_cuda) or torch.cuda.is_current_stream_capturing():
            self.wait_zero()
            self.flat_grad.zero_()
            return
        dev = self.flat_grad.device
        if getattr(self, "_zero_stream", None) is None:
            from . import streams
            self._zero_stream = streams.concurrent_stream(dev, "zero_grad")
        self.wait_zero()
        self._zero_stream.wait_stream(torch.cuda.current_stream(dev))      # behind Adam (which reads the gradients) and the shadow refresh
        with torch.cuda.stream(self._zero_stream):
            self.flat_grad.zero_()
            self._zero_done = self._zero_stream.record_event()

    def _mark_fresh(self):
        """Every gradient of this optimiser is (being) cleared: a backward function that is the ONLY writer of a parameter's gradient may
        overwrite instead of read-modify-write on its first write (functional.take_fresh_grad; today: to_visual_latent's 604-MB gradient)."""
        for p in self.params:
            p._ctclip_grad_fresh = True

    def wait_zero(self):
        """The current stream waits for an overlapped zero_grad (no-op otherwise)."""
        ev = getattr(self, "_zero_done", None)
        if ev is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_event(ev)
            self._zero_done = None

    def step(self, max_grad_norm=None, extra_sq=None, zero_grad=False):
        """zero_grad=True: optimizer.step() AND optimizer.zero_grad() (scripts/CTCLIPTrainer.py:259-264) -- the Adam kernel overwrites every
        gradient with zero right after reading it (ctclip_adam_step_zero_grad): the caller must not clear the buffer again."""
        be = _be.get()
        self.wait_zero()
        self.step_count += 1
        # the text tower's backward writes its gradients into the flat buffer from its own stream; autograd never saw them
        Fn.join_side_streams()
        clip = be.grad_norm_clip(self.flat_grad, max_grad_norm or 0.0, extra_sq)
        self.last_norm = clip
        self.lr = self.param_groups[0]["lr"]          # learning-rate schedules write param_groups (finetune.cosine_lr)
        be.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1],
                     self.eps, self.step_count, self.weight_decay, clip if max_grad_norm else None, self.decay_mask4, zero_grad=zero_grad)
        if zero_grad:
            self._mark_fresh()
        Fn.bump_weight_epoch(self.params)
        Fn.refresh_shadows(self.params)   # every bf16 GEMM operand of THIS optimiser's parameters rebuilt from the new f32 weights in one launch

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, names=self.names,
                    lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class GraphedStep:
    """One optimisation step (forward, backward, gradient clip, Adam, weight-shadow refresh, zero_grad) captured ONCE into a hipGraph and
    replayed: the ~1 900 kernel launches of a step cost the host 42 ms to enqueue through Python + ctypes, a graph launch costs microseconds
    (the reference is eager PyTorch, scripts/CTCLIPTrainer.py:249-264: it has no counterpart).  What varies from step to step and would be
    frozen by the capture lives on the device: the dropout seed offset and the optimiser step (ctclip_set_step_state), advanced by the first
    kernel of the graph; the batch is copied into static input tensors.  Single process only (collectives are not captured).

        gs = GraphedStep(trainer); gs.capture(video, text)        # after >= 3 eager warm-up steps
        loss = gs.run(video, text)                                  # every later step

    While a GraphedStep is live every Adam launch of the PROCESS reads the device step: do not mix eager optimiser steps in; `close()`."""

    def __init__(self, trainer):
        self.t = trainer
        self.graph = None
        self.state = None
        self.lr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library may already be gone
            pass

    def capture(self, video, text):
        be = _be.get()
        t = self.t
        dev = video.device
        self.state = torch.zeros(2, dtype=torch.int64, device=dev)
        self.state[1] = t.optim.step_count
        step0 = t.optim.step_count
        self.video = video.clone()
        self.ids, self.mask = text.input_ids.clone(), text.attention_mask.clone()
        self.text = type(text)(self.ids, self.mask) if not hasattr(text, "_replace") else text._replace(input_ids=self.ids, attention_mask=self.mask)
        torch.cuda.synchronize(dev)
        _be._lib.check(be.lib.ctclip_set_step_state(self.state.data_ptr()), "ctclip_set_step_state")
        GraphedStep._state_owner = self
        # Every lazily rebuilt weight shadow (the stacked q | k | v bias, non-2D weights, everything under CTCLIP_SHADOW_BATCH=0) must be STALE at
        # capture time so that its maker is recorded in the graph: after an eager forward with no optimiser step in between (validation, then
        # capture) they were fresh, their makers were not captured, and every replay read the capture-time values while Adam moved the f32 parameter
        Fn.invalidate_lazy_shadows(t.optim.params)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                _be._lib.check(be.lib.ctclip_advance_step_state(self.state.data_ptr(), _be._stream()), "ctclip_advance_step_state")
                self.loss = t.forward_backward(self.video, self.text)
                t.optim.step(t.max_grad_norm, zero_grad=True)
            torch.cuda.synchronize(dev)
        except BaseException:
            # a failed capture must leave the process as it found it: the library's step-state pointer (every later EAGER Adam launch and
            # dropout seed would read a device counter nobody advances) and the host step counter the recorded optim.step() bumped
            be.lib.ctclip_set_step_state(None)
            GraphedStep._state_owner = None
            t.optim.step_count = step0
            self.state = None
            raise
        self.graph = graph
        self.lr = t.optim.param_groups[0]["lr"]
        # the capture only RECORDED the step (nothing ran): state and step counter are where the first replay expects them
        t.optim.step_count = step0
        return self

    def run(self, video=None, text=None):
        t = self.t
        # the learning rate is a by-value kernel argument frozen by the capture: a schedule needs a re-capture
        assert t.optim.param_groups[0]["lr"] == self.lr, "the learning rate changed since capture(): re-capture the GraphedStep"
        if video is not None and video.data_ptr() != self.video.data_ptr():
            self.video.copy_(video, non_blocking=True)
        if text is not None and text.input_ids.data_ptr() != self.ids.data_ptr():
            self.ids.copy_(text.input_ids, non_blocking=True)
            self.mask.copy_(text.attention_mask, non_blocking=True)
        self.graph.replay()
        # the HOST half of optim.step(): the replay updated the parameters and rewrote the registered weight shadows on the device; shadows
        # WITHOUT a batched-refresh recipe (the stacked q|k|v bias, non-2D weights, everything under CTCLIP_SHADOW_BATCH=0) are rebuilt lazily
        # from the parameter's epoch, which only the host can advance -- an eager forward between replays would otherwise read stale ones
        t.optim.step_count += 1
        Fn.bump_weight_epoch(t.optim.params)
        Fn.restamp_shadows(t.optim.params)
        return self.loss

    _state_owner = None      # the instance whose device counters the library's process-global step-state pointer refers to

    def close(self):
        if self.state is not None:
            # only the OWNER clears the library's pointer: in the re-capture flow (gs = GraphedStep(t).capture(...) assigned over an old instance) the
            # old object is finalised AFTER the new capture installed its own state
            if GraphedStep._state_owner is self:
                _be.get().lib.ctclip_set_step_state(None)
                GraphedStep._state_owner = None
            self.state = None
        self.graph = None


class CTClipTrainer(nn.Module):
    def __init__(self, CTClip, *, num_train_steps, batch_size, data_train="train", data_valid="valid",
                 reports_file_train="data_reports.xslx", reports_file_valid="data_reports.xslx",
                 train_meta_file="meta_data.csv", valid_meta_file="meta_data.csv", labels="labels.csv", tokenizer=None,
                 lr=1.25e-6, wd=0.0, max_grad_norm=0.5, save_results_every=1, save_model_every=1,
                 results_folder="./ctclip/", num_workers=8, accelerate_kwargs: dict = dict(),
                 train_dataset=None, valid_dataset=None, evaluate=True, checkpoint=True, max_text_len=512,
                 sync_loss_every=1, device=None, grad_comm_dtype=None, overlap_grad_reduce=True, data_seed=0,
                 grad_bucket_bytes=32 << 20):
        super().__init__()
        if "RANK" in os.environ and "WORLD_SIZE" in os.environ and not _dist.is_on() and int(os.environ["WORLD_SIZE"]) > 1:
            torch.distributed.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
        if device is not None:      # (additive) explicit placement instead of cuda:LOCAL_RANK
            self.device = torch.device(device)
        else:
            self.device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        self.CTClip = CTClip.to(self.device)
        if tokenizer is None:
            from transformers import BertTokenizer
            tokenizer = BertTokenizer.from_pretrained("microsoft/BiomedVLP-CXR-BERT-specialized", do_lower_case=True)
        self.tokenizer = tokenizer
        self.register_buffer("steps", torch.Tensor([0]))
        self.num_train_steps = num_train_steps
        self.batch_size = batch_size
        self.max_grad_norm = max_grad_norm
        self.lr = lr
        self.max_text_len = max_text_len
        self.sync_loss_every = sync_loss_every

        self.optim = FusedAdam(hot_path_parameters(self.CTClip), lr=lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
        gather = getattr(self.CTClip, "gather_negatives", True)
        if grad_comm_dtype is None:
            # f32 buckets by default: the reference all-reduces f32 gradients (DDP, CTCLIPTrainer.py:138-140) and a bf16 wire format makes
            # RCCL SUM across ranks in bf16 (2-3 digits lost in every gradient at 8 ranks).  1.14 GB per step in f32 against 0.57 GB in
            # bf16, overlapped with ~60 ms of backward either way; CTCLIP_GRAD_COMM_DTYPE=bf16 (or the argument) opts in.
            env = os.environ.get("CTCLIP_GRAD_COMM_DTYPE", "").lower()
            grad_comm_dtype = torch.bfloat16 if env in ("bf16", "bfloat16") else torch.float32
        self.reducer = _dist.GradReducer(self.optim, op="sum" if gather else "mean", comm_dtype=grad_comm_dtype,
                                         min_bucket_bytes=grad_bucket_bytes, overlap=overlap_grad_reduce).install(self.CTClip)
        # VQ EMA statistics: one fused buffer, all-reduced on the communication stream, EMA applied when the step's collectives are joined
        # (reducer.finish, before the optimiser step); CTCLIP_VQ_SYNC=immediate restores the blocking in-forward all-reduce
        Fn.VqFn.stat_sync = staticmethod(_dist.sync_vq_stats if os.environ.get("CTCLIP_VQ_SYNC", "") == "immediate" else self.reducer.vq_sync)

        if train_dataset is None:
            from data import CTReportDataset  # the reference's scripts/data.py, when run from its scripts directory
            train_dataset = CTReportDataset(data_folder=data_train, reports_file=reports_file_train, meta_file=train_meta_file)
        self.ds = train_dataset
        # one process per GPU: every rank must draw DIFFERENT samples (with gathered negatives identical batches would be scored as
        # negatives of themselves) -- a DistributedSampler partitions each epoch's permutation across the ranks
        self.sampler = None
        if _dist.world_size() > 1:
            from torch.utils.data.distributed import DistributedSampler
            self.sampler = DistributedSampler(self.ds, num_replicas=_dist.world_size(), rank=_dist.rank(), shuffle=True, seed=data_seed)
        self.dl = DataLoader(self.ds, num_workers=num_workers, batch_size=self.batch_size, shuffle=self.sampler is None,
                             sampler=self.sampler)
        self.dl_iter = cycle(self.dl, self.sampler)
        self.evaluate = evaluate
        if evaluate:
            if valid_dataset is None:
                from data_inference_nii import CTReportDatasetinfer
                valid_dataset = CTReportDatasetinfer(data_folder=data_valid, reports_file=reports_file_valid,
                                                     meta_file=valid_meta_file, labels=labels)
            self.valid_ds = valid_dataset
            self.valid_dl = DataLoader(self.valid_ds, num_workers=num_workers, batch_size=1, shuffle=False)
            self.valid_dl_iter = cycle(self.valid_dl)
        self.checkpoint = checkpoint
        self.save_model_every = save_model_every
        self.save_results_every = save_results_every
        self.results_folder = Path(results_folder)
        # the reference asks interactively before clearing (CTCLIPTrainer.py:200); only do so on an interactive main rank
        if self.is_main and len([*self.results_folder.glob("**/*")]) > 0 and os.isatty(0):
            answer = input("do you want to clear previous experiment checkpoints and results? (y/n) ")
            if answer.lower() in ("yes", "y"):
                rmtree(str(self.results_folder))
        self.results_folder.mkdir(parents=True, exist_ok=True)

    # -- reference surface
    def save(self, path):
        # a deferred codebook update (distributed.VqStatSync: applied in reducer.finish() or at the next quantiser call) must be in the buffers
        # that are written -- a custom loop that bypasses forward_backward could otherwise checkpoint a codebook one EMA step behind.  flush()
        # issues no collective: calling it on every rank (or on rank 0 only) is safe.
        self.reducer.vq_sync.flush()
        if not self.is_main:
            return
        # CTCLIPTrainer.py:205-213 keeps {model, optim}; `steps` is additive so that a resumed run continues its counters
        torch.save(dict(model=self.CTClip.state_dict(), optim=self.optim.state_dict(), steps=self.steps.detach().cpu()), path)

    def load(self, path):
        path = Path(path)
        assert path.exists()
        self.reducer.vq_sync.pending = []      # statistics of the run that is being replaced must not be applied to the loaded codebook
        pkg = torch.load(path, weights_only=False)
        self.CTClip.load_state_dict(pkg["model"])      # in place: the parameters stay views of the optimiser's flat buffer
        self.optim.load_state_dict(pkg["optim"])
        if "steps" in pkg:
            self.steps.copy_(pkg["steps"])
        Fn.bump_weight_epoch()
        Fn.refresh_shadows()                            # every bf16 GEMM operand rebuilt from the restored f32 weights

    def print(self, msg):
        if self.is_main:
            print(msg)

    def close(self):
        """Detach this trainer from the process-global hooks it installed (the gradient reducer's grad-ready hook): a later model or
        trainer in the same process must not call into this one's reducer."""
        self.reducer.uninstall()
        if getattr(Fn.VqFn, "stat_sync", None) is self.reducer.vq_sync:
            self.reducer.vq_sync.flush()
            Fn.VqFn.stat_sync = None

    @property
    def is_main(self):
        return _dist.rank() == 0

    def tokenize(self, text):
        return self.tokenizer(list(text), return_tensors="pt", padding="max_length", truncation=True,
                              max_length=self.max_text_len).to(self.device)

    def forward_backward(self, video, text_tokens):
        """fwd + bwd + gradient all-reduce; returns the (device) loss."""
        loss = self.CTClip(text_tokens, video, return_loss=True, device=self.device)
        self.optim.wait_zero()   # an overlapped zero_grad of the previous step (side stream, under this forward) must be complete before a gradient is written
        Fn.wgrad_stream_begin()  # the big weight-gradient GEMMs go to a side stream, under the grad-input GEMMs of the main stream
        try:
            loss.backward()      # announces finished layers to the reducer as it goes (functional.grad_ready)
        finally:
            Fn.wgrad_stream_end()
        self.reducer.finish()
        return loss

    def train_step(self):
        steps = int(self.steps.item())
        self.CTClip.train()
        logs = {}
        video, text = next(self.dl_iter)
        video = video.to(self.device, non_blocking=True)
        text_tokens = text if hasattr(text, "input_ids") else self.tokenize(text)
        loss = self.forward_backward(video, text_tokens)
        self.optim.step(self.max_grad_norm, zero_grad=True)      # (step + zero_grad of CTCLIPTrainer.py:259-264 in one pass over the flat buffers)
        if self.sync_loss_every and steps % self.sync_loss_every == 0:
            logs["loss"] = loss.item()
            self.print(f"{steps}: loss: {logs['loss']}")
        if self.evaluate and self.is_main and not (steps % self.save_results_every):
            self.run_validation(steps)
        if self.checkpoint and self.is_main and not (steps % self.save_model_every):
            model_path = str(self.results_folder / f"CTClip.{steps}.pt")
            torch.save(self.CTClip.state_dict(), model_path)
            self.print(f"{steps}: saving model to {str(self.results_folder)}")
        self.steps += 1
        return logs

    def run_validation(self, steps):
        """Reference-compatibility glue, not part of the hot path: see ct_clip_amd/validation.py."""
        from .validation import run_validation
        return run_validation(self, steps)

    def train(self, log_fn=noop):
        while self.steps < self.num_train_steps:
            logs = self.train_step()
            log_fn(logs)
        self.print("training complete")
