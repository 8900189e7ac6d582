"""The two fine-tuning loops built on the CT-CLIP towers (BASELINE.json configs[3], configs[4]).

* ClassFine / CT-LiPro -- ``ImageLatentsClassifier`` (scripts/ct_lipro_train.py:17-38): frozen CT-CLIP -> image latents -> ReLU ->
  Dropout(0.3) -> Linear(512, 18); ``LiProTrainer.train_step`` = BCEWithLogitsLoss(pos_weight) + clip_grad_norm_(1.0) + AdamW with
  the cosine schedule (ct_lipro_train.py:79-107, src/models/utils.py:15-32); ``predict`` = sigmoid(logits)
  (ct_lipro_inference.py:44-93).  The reference runs the (unused) text tower on the prompt " " for every batch
  (ct_lipro_train.py:99-102); here ``skip_text=True`` (default) runs the image tower only -- the logits do not depend on the text.
* VocabFine -- ``VocabFineTrainer.train_step`` (scripts/ct_vocabfine_train.py:77-123): for each of 3 groups of 6 pathologies, one
  (present, absent) prompt pair per pathology ordered by the label, similarity of both prompts with the volume (CTCLIP.forward
  without return_loss, ct_clip.py:805-807), softmax over the pair, MSE against (1, 0), one backward per group, one AdamW step per
  volume.  Every parameter trains ("Fine-tuning end-to-end", ct_vocabfine_train.py:44-51).

The heads run on HIP kernels (csrc/finetune.hip) behind the same C ABI as the rest of the path.

Data parallelism (BASELINE.json configs[3] "8 GPUs", configs[4] "grad all-reduce only").  The reference scripts wrap the model in
``nn.DataParallel`` (ct_lipro_train.py:75, ct_vocabfine_train.py:62): ONE process scatters the batch over the visible GPUs, gathers the
outputs, evaluates the loss on the gathered batch and sums the replicas' gradients into the master parameters -- the gradient of the
GLOBAL-batch loss.  Here it is one process per GPU (``torch.distributed``, RCCL): every rank evaluates the mean loss of ITS shard, and the
trainers all-reduce the flat gradient buffer with op = mean (equal shards: mean of the shard gradients = gradient of the global-batch mean
loss) before the clip and the AdamW step -- 9 234 head parameters for LiPro (one 37-KB collective), every parameter for VocabFine
(bucketed on the communication stream under backward, like the main trainer).  The vector quantiser runs in train mode in both loops
(``model.train()``, ct_lipro_train.py:76): its EMA statistics are all-reduced (SUM) as in the main trainer so that every rank applies the
global-batch codebook update and the replicas' codebooks stay identical (under nn.DataParallel only replica 0's buffer update survives
the next scatter; a one-process run on the global batch -- the parity oracle of SURVEY.md section 8(e) -- sees the whole batch).
``data_parallel_loader`` shards a dataset across the ranks with a DistributedSampler.
"""
import math
import os

import torch
from torch import nn
from torch.autograd import Function

from . import backend as _be
from . import distributed as _dist
from . import functional as Fn
from .trainer import FusedAdam


def data_parallel_loader(dataset, batch_size, shuffle=True, seed=0, num_workers=0, drop_last=True):
    """-> (DataLoader, sampler or None): with a process group of W > 1 ranks every rank draws a disjoint 1/W of each epoch's permutation
    (call ``sampler.set_epoch(epoch)`` per epoch); without one it is the reference's plain shuffled loader (ct_lipro_train.py:67,
    ct_vocabfine_train.py:54).  drop_last: equal shard sizes on every rank, which is what makes mean-of-shard-gradients the global-batch
    gradient."""
    from torch.utils.data import DataLoader
    sampler = None
    if _dist.world_size() > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dataset, num_replicas=_dist.world_size(), rank=_dist.rank(), shuffle=shuffle, seed=seed, drop_last=drop_last)
    dl = DataLoader(dataset, num_workers=num_workers, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                    drop_last=drop_last)
    return dl, sampler


def _comm_dtype(arg):
    if arg is not None:
        return arg
    env = os.environ.get("CTCLIP_GRAD_COMM_DTYPE", "").lower()
    return torch.bfloat16 if env in ("bf16", "bfloat16") else torch.float32


class _DataParallelMixin:
    """Gradient all-reduce (mean) + VQ-statistic all-reduce (sum) of a fine-tuning trainer; inert without a process group."""
    reducer = None

    def _setup_data_parallel(self, optim, model=None, comm_dtype=None, bucket_bytes=16 << 20, overlap=True, sync_vq=True):
        if not _dist.collectives_on():
            return
        self.reducer = _dist.GradReducer(optim, op="mean", comm_dtype=_comm_dtype(comm_dtype), min_bucket_bytes=bucket_bytes, overlap=overlap)
        if model is not None:
            self.reducer.install(model)        # per-layer buckets announced from inside backward (VocabFine: every parameter trains)
        if sync_vq:
            Fn.VqFn.stat_sync = staticmethod(self.reducer.vq_sync)

    def _finish_data_parallel(self):
        if self.reducer is not None:
            self.reducer.finish()              # remaining buckets, join the communication stream, apply the deferred codebook update, / W

    def close(self):
        """Detach from the process-global hooks (grad-ready hook, VqFn.stat_sync) after flushing a pending codebook update."""
        if self.reducer is not None:
            self.reducer.uninstall()
            self.reducer.vq_sync.flush()
            if getattr(Fn.VqFn, "stat_sync", None) is self.reducer.vq_sync:
                Fn.VqFn.stat_sync = None

PATHOLOGIES = ['Medical material', 'Arterial wall calcification', 'Cardiomegaly', 'Pericardial effusion',
               'Coronary artery wall calcification', 'Hiatal hernia', 'Lymphadenopathy', 'Emphysema', 'Atelectasis', 'Lung nodule',
               'Lung opacity', 'Pulmonary fibrotic sequela', 'Pleural effusion', 'Mosaic attenuation pattern',
               'Peribronchial thickening', 'Consolidation', 'Bronchiectasis', 'Interlobular septal thickening']
# ct_lipro_train.py:79-83
LIPRO_POS_WEIGHT = [9.211362733, 2.384068466, 8.295479204, 32.8629776, 2.992233613, 6.064870808, 3.176470588, 4.187083754,
                    3.022222222, 1.216071737, 1.677849552, 3.152851834, 7.123261694, 18.16629381, 13.8480647, 6.335045662,
                    10.81701149, 13.40695067]


def cosine_lr(optimizer, base_lrs, warmup_length, steps):
    """src/models/utils.py:15-32: linear warm-up then half-cosine decay, applied to optimizer.param_groups."""
    if not isinstance(base_lrs, list):
        base_lrs = [base_lrs for _ in optimizer.param_groups]
    assert len(base_lrs) == len(optimizer.param_groups)

    def _lr_adjuster(step):
        for group, base_lr in zip(optimizer.param_groups, base_lrs):
            if step < warmup_length:
                lr = base_lr * (step + 1) / warmup_length
            else:
                e, es = step - warmup_length, steps - warmup_length
                lr = 0.5 * (1 + math.cos(math.pi * e / es)) * base_lr
            group["lr"] = lr
    return _lr_adjuster


class ReluDropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, seed, stream_id):
        ctx.save_for_backward(x)
        ctx.args = (p, seed, stream_id)
        return _be.get().relu_dropout(x.contiguous(), None, p, seed, stream_id)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        p, seed, stream_id = ctx.args
        return _be.get().relu_dropout(x.contiguous(), dy.contiguous(), p, seed, stream_id), None, None, None


class BceLogitsFn(Function):
    """torch.nn.BCEWithLogitsLoss(pos_weight) (mean), forward and gradient in one launch."""

    @staticmethod
    def forward(ctx, logits, targets, pos_weight):
        loss, dlogits = _be.get().bce_logits(logits.contiguous(), targets.contiguous(), pos_weight)
        ctx.save_for_backward(dlogits)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dloss, None, None


class PairSoftmaxMseFn(Function):
    @staticmethod
    def forward(ctx, sims):
        loss, dsims = _be.get().pair_softmax_mse(sims.contiguous())
        ctx.save_for_backward(dsims)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (dsims,) = ctx.saved_tensors
        return dsims * dloss


class ImageLatentsClassifier(nn.Module):
    """scripts/ct_lipro_train.py:17-38, same constructor / forward / save / load and the same state_dict keys
    (``trained_model.*``, ``classifier.weight``, ``classifier.bias``)."""

    def __init__(self, trained_model, latent_dim, num_classes, dropout_prob=0.3, skip_text=True):
        super().__init__()
        self.trained_model = trained_model
        for param in self.trained_model.parameters():
            param.requires_grad = False
        self.dropout = nn.Dropout(dropout_prob)        # container for p (the fused ReLU + dropout kernel does the work)
        self.relu = nn.ReLU()
        self.classifier = nn.Linear(latent_dim, num_classes)
        self.skip_text = skip_text

    def image_latents(self, *args, **kwargs):
        if self.skip_text:
            image = args[1] if len(args) > 1 else kwargs["image"]
            with torch.no_grad():
                return self.trained_model.encode_image(image)
        kwargs["return_latents"] = True
        with torch.no_grad():
            _, lat, _ = self.trained_model(*args, **kwargs)
        return lat

    def forward(self, *args, **kwargs):
        kwargs.pop("return_latents", None)
        lat = self.image_latents(*args, **kwargs)                                   # (B, latent_dim) f32, l2-normalised
        p = float(self.dropout.p) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        h = ReluDropoutFn.apply(lat, p, seed, 0)
        return Fn.linear(h, self.classifier.weight, self.classifier.bias, out_dtype=torch.float32)

    def save(self, file_path):
        torch.save(self.state_dict(), file_path)

    def load(self, file_path):
        self.load_state_dict(torch.load(file_path))


class LiProTrainer(_DataParallelMixin):
    """One optimisation step of scripts/ct_lipro_train.py:92-107 on device-resident tensors.  In a process group every rank passes its shard
    of the global batch (data_parallel_loader); the head's gradients are averaged over the ranks before the clip (module docstring)."""

    def __init__(self, classifier, lr=1e-5, wd=0.1, warmup_length=500, total_steps=10000, pos_weight=None, max_grad_norm=1.0,
                 grad_comm_dtype=None, sync_vq=True):
        self.model = classifier
        dev = classifier.classifier.weight.device
        pw = LIPRO_POS_WEIGHT if pos_weight is None else pos_weight
        self.pos_weight = torch.tensor(pw, dtype=torch.float32, device=dev)
        named = [(n, p) for n, p in classifier.named_parameters() if p.requires_grad]      # the 18 x 512 head and its bias
        self.optim = FusedAdam(named, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, group_wd_params=False)   # torch.optim.AdamW defaults
        self.scheduler = cosine_lr(self.optim, lr, warmup_length, total_steps)
        self.max_grad_norm = max_grad_norm
        self.step = 0
        # "grad all-reduce only" (BASELINE configs[4]): the 9 234 head parameters are ONE collective after backward -- nothing to overlap with
        self._setup_data_parallel(self.optim, None, grad_comm_dtype, overlap=False, sync_vq=sync_vq)

    def forward_backward(self, text_tokens, volumes, labels):
        """logits, BCEWithLogitsLoss(pos_weight), backward into the head's gradients (ct_lipro_train.py:103-105)."""
        self.model.train()
        dev = self.pos_weight.device
        logits = self.model(text_tokens, volumes, device=dev)
        loss = BceLogitsFn.apply(logits, labels.to(device=dev, dtype=torch.float32), self.pos_weight)
        self.optim.zero_grad()
        loss.backward()
        self._finish_data_parallel()
        return loss, logits

    def train_step(self, text_tokens, volumes, labels):
        loss, logits = self.forward_backward(text_tokens, volumes, labels)
        self.optim.step(self.max_grad_norm)
        self.scheduler(self.step)                       # ct_lipro_train.py:107: the schedule is applied AFTER the step
        self.step += 1
        return loss.detach(), logits.detach()

    @torch.no_grad()
    def predict(self, text_tokens, volumes):
        """ct_lipro_inference.py:62-66: sigmoid of the logits in eval mode."""
        self.model.eval()
        return torch.sigmoid(self.model(text_tokens, volumes, device=self.pos_weight.device))


def vocabfine_prompts(pathology, label):
    """ct_vocabfine_train.py:100-109: the first prompt is the TRUE statement."""
    if int(label) == 1:
        return [f"{pathology} is present. ", f"{pathology} is not present. "]
    return [f"{pathology} is not present. ", f"{pathology} is present. "]


class VocabFineTrainer(_DataParallelMixin):
    """scripts/ct_vocabfine_train.py:77-123 for one volume per step and rank.  `tokenize(list_of_two_strings)` must return an object with
    .input_ids / .attention_mask on the model's device (the reference pads to 512 tokens).  In a process group every rank steps on its own
    volume; the gradients of ALL parameters are averaged over the ranks in buckets launched from inside backward (module docstring)."""

    def __init__(self, model, tokenize, lr=1e-5, wd=0.1, warmup_length=500, total_steps=10000, pathologies=None, group_size=6,
                 grad_comm_dtype=None, grad_bucket_bytes=16 << 20, overlap_grad_reduce=True, sync_vq=True):
        from .trainer import hot_path_parameters
        self.model, self.tokenize = model, tokenize
        self.pathologies = list(pathologies or PATHOLOGIES)
        self.group_size = group_size
        self.optim = FusedAdam(hot_path_parameters(model), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, group_wd_params=False)
        self.scheduler = cosine_lr(self.optim, lr, warmup_length, total_steps)
        self.step = 0
        self._setup_data_parallel(self.optim, model, grad_comm_dtype, grad_bucket_bytes, overlap_grad_reduce, sync_vq)

    def forward_backward(self, volume, token_pairs, fused=True):
        # the literal loop runs one backward per group INTO THE SAME gradients: a bucket reduced from inside the first backward would be added
        # to afterwards -- buckets are launched from inside backward only in the fused form (one backward), otherwise all at finish()
        prev = Fn.set_grad_ready_hook(None) if (self.reducer is not None and not fused) else None
        try:
            losses, sims = self._forward_backward(volume, token_pairs, fused)
        finally:
            if prev is not None:
                Fn.set_grad_ready_hook(prev)
        self._finish_data_parallel()
        return losses, sims

    def _forward_backward(self, volume, token_pairs, fused=True):
        """token_pairs: one (true prompt, false prompt) token batch per pathology.  Returns (losses per group, similarities per group);
        the gradients of all groups accumulate in the parameters' .grad as in ct_vocabfine_train.py:88-121.

        fused=False is the reference's loop literally: one full CTCLIP forward per pathology (18 image-tower passes over the SAME volume
        with the SAME weights), one backward per group of `group_size`.
        fused=True (default) computes the same numbers from ONE pass of each tower: within a step the weights do not change, so the image
        transformer's output is identical in all 18 forwards -- only the vector quantiser differs (train mode: its EMA update after every
        forward moves the codebook the next forward reads).  Hence: patch embedding + transformers once; all prompts through BERT as one
        batch; the quantiser + pooling re-applied per pathology in the reference's order (same EMA sequence), the latent projection of all 18
        pooled vectors as one batch; the group losses summed and ONE backward (backward is linear in the incoming gradient, so the sum of the reference's three backwards equals
        the backward of the summed loss).  M = 13 824 token rows per launch either way -- but 1 image-tower pass instead of 18."""
        model = self.model
        model.train()
        dev = model.temperature.device
        self.optim.zero_grad()
        losses, sims_all = [], []
        if not fused:
            for k in range(0, len(token_pairs), self.group_size):
                sims = [model(tokens, volume, device=dev) for tokens in token_pairs[k:k + self.group_size]]     # (2,) each: true prompt first
                st = torch.stack(sims)
                loss = PairSoftmaxMseFn.apply(st)
                loss.backward()
                losses.append(loss.detach())
                sims_all.append(st.detach())
            return losses, sims_all
        vt = model.visual_transformer
        pre, (b, t, h, w) = vt.tokens_before_vq(volume)                              # once per volume
        ids = torch.cat([tp.input_ids for tp in token_pairs]).to(dev)
        mask = torch.cat([tp.attention_mask for tp in token_pairs]).to(dev)
        text_lat = model.text_latents_raw(ids, mask)                                 # (2 P, Dl) f32, every prompt in one BERT batch
        # the quantiser per pathology, in the reference's order (its EMA update after every forward moves the codebook the next one reads);
        # the pooled vectors of all pathologies then go through the 151-M-parameter latent projection as ONE batch (one pass over the
        # 302-MB weight forward, one 604-MB weight-gradient write backward instead of one per pathology)
        assert b == 1, "the fused VocabFine step pairs pathology j with row j of the stacked latents: one volume per step (ct_vocabfine_train.py batch_size=1); use fused=False for more"
        pooled = []
        for j in range(len(token_pairs)):
            q, _ = vt.vq(pre)
            pooled.append(model.pool_tokens(q.view(b, t, h, w, -1)))                  # (1, h*w*d)
        image_lat = Fn.visual_latent(torch.cat(pooled, dim=0), model.to_visual_latent.weight)      # (P, Dl) f32, pre-l2norm
        total = None
        for k in range(0, len(token_pairs), self.group_size):
            sims = [Fn.LatentSimilarityFn.apply(text_lat[2 * j:2 * j + 2], image_lat[j:j + 1], model.temperature)
                    for j in range(k, min(k + self.group_size, len(token_pairs)))]
            st = torch.stack(sims)
            loss = PairSoftmaxMseFn.apply(st)
            total = loss if total is None else total + loss
            losses.append(loss.detach())
            sims_all.append(st.detach())
        total.backward()
        return losses, sims_all

    def train_step(self, volume, labels):
        """volume: (1, 1, F, H, W); labels: (n_pathologies,) 0/1.  Returns the per-group losses."""
        self.scheduler(self.step)                       # ct_vocabfine_train.py:82: the schedule is applied BEFORE the step
        pairs = [self.tokenize(vocabfine_prompts(name, lab)) for name, lab in zip(self.pathologies, labels)]
        losses, _ = self.forward_backward(volume, pairs)
        self.optim.step(None)
        self.step += 1
        return losses
