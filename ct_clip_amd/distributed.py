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
import torch
import torch.distributed as dist


def is_on():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_on() else 1


def rank():
    return dist.get_rank() if is_on() else 0


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


def sync_vq_stats(bins, esum):
    if world_size() > 1:
        dist.all_reduce(bins, op=dist.ReduceOp.SUM)
        dist.all_reduce(esum, op=dist.ReduceOp.SUM)


class GradReducer:
    """All-reduce a flat gradient buffer in ``bucket_bytes`` pieces; ``op`` 'sum' (gathered-negatives objective) or 'mean'."""

    def __init__(self, flat_grad, op="sum", bucket_bytes=256 << 20):
        self.flat = flat_grad
        self.op = op
        n = max(1, bucket_bytes // flat_grad.element_size())
        self.buckets = [(s, min(s + n, flat_grad.numel())) for s in range(0, flat_grad.numel(), n)]
        self.comm_stream = torch.cuda.Stream() if flat_grad.is_cuda else None

    def reduce(self):
        W = world_size()
        if W == 1:
            return
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                works = [dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b in self.buckets]
                for w in works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for a, b in self.buckets:
                dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM)
        if self.op == "mean":
            self.flat.div_(W)
