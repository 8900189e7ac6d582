"""Data-parallel pieces of the CT-CLIP hot path: one process per GPU, ``torch.distributed`` (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

* ``all_gather_latents``: gathered-negatives CLIP.  The reference's helper for this is dead code
  (CT_CLIP/ct_clip/distributed.py:9-51, never imported by ct_clip.py); its intended semantics are adopted:
  forward all-gather of the (B, Dl) latents, backward = the local slice of the gradient.  Every rank evaluates the
  SAME global loss, so parameter gradients must be SUMMED across ranks to equal the single-process gradient of the
  global-batch loss.
* ``GradReducer``: all-reduce of the trainer's flat f32 gradient buffer in large contiguous buckets on a side stream
  (xGMI is 7 point-to-point links per GPU: few, large messages).
* ``sync_vq_stats``: all-reduce(SUM) of the VQ EMA statistics so every rank applies the global-batch codebook update.
"""
import os

import torch
import torch.distributed as dist


def is_on():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_on() else 1


def rank():
    return dist.get_rank() if is_on() else 0


def collectives_on():
    """True when this process takes the data-parallel branches: a process group of more than one rank -- or of ONE rank with
    CTCLIP_DIST_SINGLE_RANK=1, where every collective is the identity but is still issued (how the RCCL branch of GradReducer / VqStatSync /
    the latent all-gather is executed on a 1-GPU box: tests/test_ddp_gpu.py)."""
    return is_on() and (dist.get_world_size() > 1 or os.environ.get("CTCLIP_DIST_SINGLE_RANK", "") == "1")


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        W = dist.get_world_size()
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        r = dist.get_rank()
        return dout[r * ctx.n:(r + 1) * ctx.n].contiguous()


def all_gather_latents(text_lat, image_lat):
    """(B, Dl) x2 -> (W*B, Dl) x2 in rank order, one collective for both towers."""
    both = torch.stack([text_lat, image_lat], dim=1)          # (B, 2, Dl)
    g = _AllGatherRows.apply(both)                            # (W*B, 2, Dl)
    return g[:, 0].contiguous(), g[:, 1].contiguous()


def _fused_stats(bins, esum):
    """The [bins | esum] buffer both are views of (backend.vq_ema allocates them as one), or None."""
    base = bins._base
    if base is not None and base is esum._base and base.is_contiguous() and base.numel() == bins.numel() + esum.numel():
        return base
    return None


def sync_vq_stats(bins, esum, *_):
    """Immediate form (VqFn.stat_sync hook): all-reduce(SUM) in place on the caller's stream, the EMA update follows in the forward."""
    if collectives_on():
        flat = _fused_stats(bins, esum)
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(bins, op=dist.ReduceOp.SUM)
            dist.all_reduce(esum, op=dist.ReduceOp.SUM)
    return False


class VqStatSync:
    """Deferred form (the trainer's default): the EMA statistics of the vector quantiser -- 32 KB of bin counts + 16.8 MB of code sums in
    f32 -- are ONE buffer, all-reduced (SUM) on the COMMUNICATION stream while the forward carries on, and the EMA update of the codebook
    (which nothing reads before the next forward) is applied by `flush()` when the step's collectives are joined (GradReducer.finish).
    The immediate form put two blocking all-reduces on the main stream in the middle of every forward.  A quantiser call that finds an update
    of the same codebook pending (VocabFine's per-pathology sequence, an evaluation forward inside the step) applies it BEFORE reading the
    codebook (`before_forward`, called at the top of VqFn.forward): the EMA sequence is preserved."""

    def __init__(self, comm_stream=None):
        self.comm_stream = comm_stream
        self.pending = []                  # (bins, esum, cluster_size, embed, decay)
        self.calls = 0                     # collectives issued (tests / inspection)

    def before_forward(self, embed):
        """Called by VqFn.forward before it reads the codebook: a pending update of the same codebook is applied first."""
        if any(e[3].data_ptr() == embed.data_ptr() for e in self.pending):
            self.flush()

    def __call__(self, bins, esum, cluster_size, embed, decay):
        if not collectives_on():
            return False
        self.before_forward(embed)
        flat = _fused_stats(bins, esum)
        pieces = [flat] if flat is not None else [bins, esum]
        if self.comm_stream is not None and bins.is_cuda:
            cur = torch.cuda.current_stream(bins.device)
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                for t in pieces:
                    t.record_stream(self.comm_stream)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            for t in pieces:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        self.calls += len(pieces)
        self.pending.append((bins, esum, cluster_size, embed, decay))
        return True

    def flush(self):
        """Join the communication stream and apply the pending EMA updates on the current stream."""
        if not self.pending:
            return
        from . import backend as _be
        be = _be.get()
        if self.comm_stream is not None and self.pending[0][0].is_cuda:
            torch.cuda.current_stream(self.pending[0][0].device).wait_stream(self.comm_stream)
        with torch.no_grad():
            for bins, esum, cluster_size, embed, decay in self.pending:
                be.vq_ema_update(cluster_size, embed, bins, esum, decay)
        self.pending = []


class GradReducer:
    """All-reduce of the trainer's flat f32 gradient buffer, OVERLAPPED with backward.

    The model announces (functional.grad_ready / notify_grad_ready) when the gradients of a block are final; blocks are contiguous
    ranges of the flat buffer (registration order).  Ready ranges are coalesced with their neighbours and, once a run reaches
    `min_bucket_bytes`, handed to the communication stream: [f32 -> comm dtype] -> all_reduce(SUM) over RCCL -> [back to f32 in
    place], while the compute stream carries on with the backward of the earlier layers.  `finish()` reduces whatever is left
    (embeddings, patch embedding, position-bias MLP, temperature) and joins the streams.  Every rank sees the same sequence of
    notifications (same graph), hence the same sequence of collectives.

    comm_dtype bf16 halves the bytes on the xGMI links (0.57 GB per step for CT-CLIP); the sum over ranks is then accumulated in
    bf16 by RCCL -- use f32 (default in f32 parity mode) when bit-comparability with a single-process run matters.
    op: 'sum' (gathered-negatives objective: every rank differentiates the same global loss) or 'mean' (DDP semantics)."""

    def __init__(self, optim_or_flat, op="sum", comm_dtype=torch.float32, min_bucket_bytes=16 << 20, max_bucket_bytes=256 << 20,
                 overlap=True):
        self.optim = optim_or_flat if hasattr(optim_or_flat, "flat_grad") else None
        self.flat = self.optim.flat_grad if self.optim is not None else optim_or_flat
        self.op, self.comm_dtype, self.overlap = op, comm_dtype, overlap
        self.min_elems = max(1, min_bucket_bytes // 4)
        self.max_elems = max(1, max_bucket_bytes // 4)
        self.comm_stream = None
        if self.flat.is_cuda:
            # hardware queues are handed out in the order of asking (streams.py): the streams that carry kernels first, then this one, whose
            # event waits must not sit in the main stream's queue
            from . import functional as Fn, streams
            Fn.reserve_side_streams(self.flat.device)
            self.comm_stream = streams.concurrent_stream(self.flat.device, "comm")
        self.stage = torch.empty(self.flat.numel(), dtype=comm_dtype, device=self.flat.device) if comm_dtype != torch.float32 else None
        self.ranges = {}          # tag key -> (start, end)
        self.pending = []         # ready, not yet launched: sorted list of [start, end)
        self.launched = []        # ranges already handed to the communication stream this step
        self.works = []
        self.log = []             # (start, end) in launch order -- for tests / inspection
        self.timing = None        # start_timing(): [(event before, event after)] per collective on the communication stream + exposed waits
        self.vq_sync = VqStatSync(self.comm_stream)      # the quantiser's EMA statistics ride the same communication stream (install it as VqFn.stat_sync)

    def reconfigure(self, comm_dtype=None, min_bucket_bytes=None):
        """Change the wire format and / or the bucket threshold between steps (bench.py's sweep: the first multi-GPU record decides the
        defaults from ONE invocation).  Must be called on every rank with the same arguments, outside a step."""
        assert not self.pending and not self.launched, "reconfigure between steps, not inside one"
        if comm_dtype is not None and comm_dtype != self.comm_dtype:
            self.comm_dtype = comm_dtype
            self.stage = torch.empty(self.flat.numel(), dtype=comm_dtype, device=self.flat.device) if comm_dtype != torch.float32 else None
        if min_bucket_bytes is not None:
            self.min_elems = max(1, min_bucket_bytes // 4)
        return self

    # ---- diagnostics: where does the step's communication time go?  (bench.py --gpus N prints it with the throughput line)
    def start_timing(self):
        self.timing = dict(coll=[], exposed=[], bytes=0, steps=0)

    def stop_timing(self):
        """-> per-step means: collectives, bytes, time the collectives occupied the communication stream, and the EXPOSED part -- how long the
        compute stream waited in finish() for the last bucket (the rest ran under backward)."""
        t, self.timing = self.timing, None
        if not t or not t["steps"] or self.comm_stream is None:
            return None
        torch.cuda.synchronize(self.flat.device)
        n = t["steps"]
        return dict(collectives_per_step=round(len(t["coll"]) / n, 1), MB_per_step=round(t["bytes"] / n / 1e6, 1),
                    comm_stream_busy_ms_per_step=round(sum(a.elapsed_time(b) for a, b in t["coll"]) / n, 3),
                    exposed_wait_ms_per_step=round(sum(a.elapsed_time(b) for a, b in t["exposed"]) / n, 3),
                    wire_dtype=str(self.comm_dtype).replace("torch.", ""), min_bucket_MB=round(self.min_elems * 4 / 2 ** 20, 1))

    # ---- registration
    @staticmethod
    def _key(tag):
        return tag if isinstance(tag, (str, tuple)) else id(tag)

    def register(self, tag, params):
        assert self.optim is not None, "bucket registration needs the optimiser's flat layout"
        r = self.optim.range_of(list(params))
        if r is not None:
            self.ranges[self._key(tag)] = r

    def install(self, model):
        """Register the blocks the model announces and hook functional.grad_ready.  Returns self."""
        from . import functional as Fn
        vt, tt = model.visual_transformer, model.text_transformer
        for tr in (vt.enc_spatial_transformer, vt.enc_temporal_transformer):
            n = len(tr.layers)
            for i, layer in enumerate(tr.layers):
                extra = list(tr.norm_out.parameters()) if i == n - 1 else []      # norm_out sits right behind the last layer
                self.register(layer, list(layer.parameters()) + extra)
        for layer in tt.encoder.layer:
            self.register(layer, layer.parameters())
        emb = tt.embeddings
        self.register(("bert_embeddings", id(emb.word_embeddings.weight)), emb.parameters())
        self.register(model.to_text_latent, model.to_text_latent.parameters())
        self.register(model.to_visual_latent, model.to_visual_latent.parameters())
        if collectives_on() and self.overlap:
            self._prev_hook = Fn.set_grad_ready_hook(self.ready)
            self._hooked = True
        return self

    def uninstall(self):
        """Restore the grad-ready hook that was active before install() (a later model / trainer in the same process must not call
        into this reducer)."""
        if getattr(self, "_hooked", False):
            from . import functional as Fn
            Fn.set_grad_ready_hook(self._prev_hook)
            self._hooked = False

    # ---- backward-time
    def _events_now(self):
        """Events behind everything that may have written the range announced right now: the announcing stream (the image tower's
        main stream or the text tower's side stream -- whichever backward is running on) and the weight-gradient stream(s)."""
        if self.comm_stream is None:
            return []
        from . import functional as Fn
        return [torch.cuda.current_stream().record_event()] + list(Fn.wgrad_event() or ())

    def ready(self, tag):
        r = self.ranges.get(self._key(tag))
        if r is None or not collectives_on():
            return
        a, b = r
        # the events travel WITH the segment: a segment announced on stream A and merged into a run launched from a notification on
        # stream B is still ordered behind A's kernels
        self.pending.append([a, b, self._events_now()])
        self.pending.sort(key=lambda seg: seg[0])
        merged = []
        for seg in self.pending:
            if merged and merged[-1][1] == seg[0]:
                merged[-1][1] = seg[1]
                merged[-1][2] = merged[-1][2] + seg[2]
            else:
                merged.append(seg)
        self.pending = []
        for seg in merged:
            if seg[1] - seg[0] >= self.min_elems:
                self._launch(seg[0], seg[1], seg[2])
            else:
                self.pending.append(seg)

    def _launch(self, a, b, events=None):
        if events is None:
            events = self._events_now()
        for s in range(a, b, self.max_elems):
            e = min(b, s + self.max_elems)
            self.log.append((s, e))
            self.launched.append((s, e))
            if self.comm_stream is not None:
                for ev in events:
                    self.comm_stream.wait_event(ev)
                with torch.cuda.stream(self.comm_stream):
                    if self.timing is not None:
                        e0 = torch.cuda.Event(enable_timing=True); e0.record(self.comm_stream)
                    self._reduce_slice(s, e)
                    if self.timing is not None:
                        e1 = torch.cuda.Event(enable_timing=True); e1.record(self.comm_stream)
                        self.timing["coll"].append((e0, e1))
                        self.timing["bytes"] += (e - s) * (4 if self.stage is None else self.stage.element_size())
            else:
                self._reduce_slice(s, e)

    def _reduce_slice(self, s, e):
        piece = self.flat[s:e]
        if self.stage is None:
            if dist.get_backend() == "nccl" and os.environ.get("CTCLIP_NCCL_ASYNC", "0") != "1":
                # RCCL, synchronous form: the collective is launched ON the communication stream (the current stream here) and is ordered like
                # any kernel of it -- the host does not block.  The asynchronous form + wait() goes through the process group's internal
                # stream: one more stream competing for four hardware queues (streams.py) and two event hops per bucket.
                dist.all_reduce(piece, op=dist.ReduceOp.SUM)
                return
            w = dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=self.comm_stream is not None)
            if w is not None and self.comm_stream is not None:
                if dist.get_backend() == "nccl":
                    w.wait()              # stream-ordered (the communication stream waits for the internal stream; the host does not block)
                else:
                    self.works.append(w)  # gloo: wait() blocks the host -- deferred to finish() so that backward keeps being enqueued
            return
        from . import backend as _be
        be = _be.get()
        st = self.stage[s:e]
        be.convert_pad(piece.view(1, -1), 1, e - s, self.comm_dtype, out=st.view(1, -1))
        dist.all_reduce(st, op=dist.ReduceOp.SUM)                 # stream-ordered on the communication stream
        be.convert_pad(st.view(1, -1), 1, e - s, torch.float32, out=piece.view(1, -1))

    def finish(self):
        """Reduce every range not yet reduced this step, then make the compute stream wait for the communication stream."""
        W = world_size()
        if not collectives_on():
            return
        self.vq_sync.flush()            # (joins the communication stream for the statistics' all-reduce, then the EMA update on the compute stream)
        if self.comm_stream is not None:
            from . import functional as Fn
            Fn.join_side_streams()      # gradients written from the text tower's stream: the events below are recorded on the current stream
        for seg in self.pending:
            self._launch(seg[0], seg[1], seg[2] + self._events_now())
        self.pending = []
        done = sorted(self.launched)
        pos, n = 0, self.flat.numel()
        rest = []
        for a, b in done:
            if a > pos:
                rest.append((pos, a))
            pos = max(pos, b)
        if pos < n:
            rest.append((pos, n))
        for a, b in rest:
            self._launch(a, b)
        if self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                for w in self.works:
                    w.wait()
            cur = torch.cuda.current_stream()
            if self.timing is not None:
                m0 = torch.cuda.Event(enable_timing=True); m0.record(cur)
            cur.wait_stream(self.comm_stream)
            if self.timing is not None:
                m1 = torch.cuda.Event(enable_timing=True); m1.record(cur)
                self.timing["exposed"].append((m0, m1))
                self.timing["steps"] += 1
        self.works = []
        self.launched = []
        if self.op == "mean":
            self.flat.div_(W)

    # round-1 name
    def reduce(self):
        self.finish()
