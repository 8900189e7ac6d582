"""Autograd layer: torch.autograd.Function wrappers whose forward/backward are sequences of C-ABI kernel
launches (``backend.get()``).  The backward formulas are hand-derived; ``tests/test_host_logic_cpu.py``
checks them on CPU against the oracle by swapping in the pure-torch checker backend.

Weight handling
  * master parameters stay f32 with the reference's shapes/keys (state_dict compatibility);
  * each Linear weight has a compute-dtype *shadow* (bf16 in performance mode), possibly padded / re-laid
    out for 16-byte rows and MFMA tiles, refreshed when the parameter's version counter changes;
  * weight gradients are written straight into ``param._ctclip_grad_sink`` (a view of the trainer's flat f32
    gradient buffer; split-K partial sums go through f32 slabs and a fixed-order reduce -- the library has no float atomics)
    when present -- autograd then sees ``None`` for that input; otherwise a fresh f32 gradient tensor is returned as usual.
"""
import os

import numpy as np
import torch
from torch.autograd import Function

from . import backend as _be
from . import streams as _streams


def B():
    return _be.get()


def round_up(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------ shadows / sinks

_WEIGHT_EPOCH = 0


def bump_weight_epoch(params=None):
    """The fused optimiser updates parameters through raw kernels, invisible to torch's version counters: it calls this so that
    the weight shadows of ITS parameters are rebuilt (params = the tensors it updated).  Without arguments every shadow of the
    process is invalidated (checkpoint load, tests)."""
    global _WEIGHT_EPOCH
    if params is None:
        _WEIGHT_EPOCH += 1
        return
    for p in params:
        p.__dict__["_ctclip_epoch"] = p.__dict__.get("_ctclip_epoch", 0) + 1


def invalidate_lazy_shadows(params):
    """Drop the cached shadows of `params` that have NO batched-refresh recipe (the stacked q | k | v bias, non-2D weights, everything under
    CTCLIP_SHADOW_BATCH=0): their makers run again at the next use.  GraphedStep.capture() calls this so that those makers are RECORDED in the graph
    whatever ran before the capture (shadows with a recipe are rewritten in place by the captured refresh_shadows launch and keep their storage)."""
    for p in params:
        cache = p.__dict__.get("_ctclip_shadow")
        if cache:
            for key in [k for k in cache if (id(p), k) not in _SHADOW_PLAN]:
                del cache[key]


def _stamp(tensors):
    return (_WEIGHT_EPOCH,) + tuple((t._version, t.data_ptr(), t.__dict__.get("_ctclip_epoch", 0)) for t in tensors)


def shadow(param, tag, dtype, maker, recipe=None, deps=()):
    """Cached derived tensor of a parameter (recomputed when the parameter -- or one of `deps`, the other parameters it is built
    from -- is modified).
    recipe: how the batched refresh (refresh_shadows) rebuilds this tensor from f32 master weights -- a list of jobs
    (source parameter, first destination row, destination rows, destination columns, map, aux, transposed[, first destination column]),
    see csrc/shadow.hip."""
    cache = param.__dict__.setdefault("_ctclip_shadow", {})
    key = (tag, dtype)
    ent = cache.get(key)
    stamp = _stamp((param,) + tuple(deps))
    if ent is not None and ent[0] == stamp:
        return ent[1]
    with torch.no_grad():
        val = maker()
    cache[key] = (stamp, val)
    if recipe is not None and dtype == torch.bfloat16 and val.dim() == 2 and val.stride(1) == 1 and _SHADOW_BATCH:
        _register_shadow(param, key, recipe, deps)
    return val


# ---- batched refresh: one launch after the optimiser step instead of one small launch per shadow in front of its first GEMM
_SHADOW_BATCH = os.environ.get("CTCLIP_SHADOW_BATCH", "1") != "0"
_SHADOW_PLAN = {}            # (id(owner parameter), key) -> (weakref(owner), key, [(weakref(source), row0, rows, cols, map, aux, transposed)], [weakref(dep)])
_SHADOW_PLAN_VERSION = 0     # bumped by every registration
_SHADOW_JOBS = {}            # scope key -> cached job list for the backend
_SHADOW_VERSION = 0
MAP_PLAIN, MAP_GEGLU_SPLIT, MAP_GEGLU_INTERLEAVE = 0, 1, 2


def _register_shadow(param, key, recipe, deps=()):
    global _SHADOW_PLAN_VERSION
    import weakref
    _SHADOW_PLAN[(id(param), key)] = (weakref.ref(param), key, [(weakref.ref(src),) + tuple(rest) for src, *rest in recipe],
                                      [weakref.ref(d) for d in deps])
    _SHADOW_PLAN_VERSION += 1


def _build_shadow_jobs(scope):
    """-> (jobs, owners, checks).  Sources are held WEAKLY (a job keeps raw pointers; `checks` = (weakref(tensor), data_ptr) pairs that
    refresh_shadows re-validates before every launch): a freed or re-based parameter rebuilds the list instead of being read stale."""
    import weakref
    jobs, owners, checks = [], [], []
    for pk, (pref, key, recipe, deprefs) in list(_SHADOW_PLAN.items()):
        param = pref()
        ent = param.__dict__.get("_ctclip_shadow", {}).get(key) if param is not None else None
        srcs = [r[0]() for r in recipe]
        if ent is None or any(s_ is None for s_ in srcs):
            del _SHADOW_PLAN[pk]           # the module is gone
            continue
        val = ent[1]
        if scope is not None and not any(id(s_) in scope for s_ in srcs):
            continue                       # none of this shadow's sources belongs to the calling optimiser: untouched by its step
        if any(s_.device != val.device or s_.dtype != torch.float32 for s_ in srcs):
            continue                       # (a model moved to another device keeps its lazy makers)
        for src, (_, row0, rows, cols, mp, aux, tr, *rest) in zip(srcs, recipe):
            col0 = rest[0] if rest else 0          # (optional 8th field: first destination column -- stacked transposed shadows)
            dst = val[row0:row0 + rows, col0:col0 + cols]
            jobs.append(dict(src_ref=weakref.ref(src), src_ptr=src.data_ptr(), src_stride=src.stride(0), src_shape=tuple(src.shape), dst=dst,
                             map=mp, aux=aux, transposed=bool(tr)))
            checks.append((weakref.ref(src), src.data_ptr()))
        checks.append((weakref.ref(val), val.data_ptr()))
        owners.append((pref, key, deprefs))
    return jobs, owners, checks


def refresh_shadows(params=None):
    """Rebuild the registered bf16 weight shadows in place from the current f32 parameters (ONE launch) and stamp them valid.
    Called by the fused optimiser right after its parameter update with params = the parameters it owns: only shadows built from
    them are touched (a frozen tower keeps its operands; another optimiser's shadows are its own business).  params=None: every
    registered shadow.  Shadows without a recipe stay lazy."""
    global _SHADOW_VERSION
    if not _SHADOW_BATCH or not _SHADOW_PLAN:
        return
    scope = None if params is None else frozenset(id(p) for p in params)
    ent = _SHADOW_JOBS.get(scope)
    if ent is not None:
        pv, jobs, owners, checks, version = ent
        if pv != _SHADOW_PLAN_VERSION or any(r() is None or r().data_ptr() != ptr for r, ptr in checks):
            ent = None
    if ent is None:
        jobs, owners, checks = _build_shadow_jobs(scope)
        _SHADOW_VERSION += 1
        version = _SHADOW_VERSION
        if len(_SHADOW_JOBS) > 8:
            _SHADOW_JOBS.clear()
        _SHADOW_JOBS[scope] = (_SHADOW_PLAN_VERSION, jobs, owners, checks, version)
    if not jobs:
        return
    with torch.no_grad():
        B().shadow_refresh(jobs, version)
    for pref, key, deprefs in owners:
        param = pref()
        deps = [d() for d in deprefs]
        if param is None or any(d is None for d in deps):
            continue
        cache = param.__dict__.get("_ctclip_shadow", {})
        ent = cache.get(key)
        if ent is not None:
            cache[key] = (_stamp((param,) + tuple(deps)), ent[1])


def restamp_shadows(params=None):
    """Stamp the registered (batched-refresh) shadows of `params` valid WITHOUT launching anything: for a hipGraph replay whose captured
    shadow_refresh launch already rewrote them on the device (GraphedStep.run).  Shadows without a recipe keep their old stamp and are
    rebuilt by their lazy makers at the next use."""
    if not _SHADOW_BATCH or not _SHADOW_PLAN:
        return
    scope = None if params is None else frozenset(id(p) for p in params)
    ent = _SHADOW_JOBS.get(scope)
    if ent is None or ent[0] != _SHADOW_PLAN_VERSION:
        return            # no job list was ever built for this scope (nothing was refreshed on the device either): everything stays lazy
    for pref, key, deprefs in ent[2]:
        param = pref()
        deps = [d() for d in deprefs]
        if param is None or any(d is None for d in deps):
            continue
        cache = param.__dict__.get("_ctclip_shadow", {})
        e = cache.get(key)
        if e is not None:
            cache[key] = (_stamp((param,) + tuple(deps)), e[1])


def plain_shadow(weight, dtype, kpad=None, npad=None):
    """(N, K) f32 -> (Np, Kp) compute dtype, zero padded."""
    N, K = weight.shape
    Np, Kp = npad or N, kpad or K

    def make():
        w = weight.detach()
        if dtype == torch.float32 and Np == N and Kp == K and w.is_contiguous():
            return w
        return B().convert_pad(w, Np, Kp, dtype)
    return shadow(weight, ("plain", Np, Kp), dtype, make, recipe=[(weight, 0, Np, Kp, MAP_PLAIN, 0, False)] if weight.dim() == 2 else None)


def transposed_shadow(weight, wsh, segments):
    """(Kp, Np) transposed form of the (Np, Kp) shadow `wsh` of `weight` (both operands of the grad-input GEMM k-contiguous).
    segments: the LinearFn row segments of wsh -- two segments = the padded [x | gate] layout of the GEGLU in-projection."""
    split = len(segments) == 2
    recipe = [(weight, 0, wsh.shape[1], wsh.shape[0], MAP_GEGLU_SPLIT if split else MAP_PLAIN, segments[0][1] if split else 0, True)]
    return shadow(weight, ("T",) + tuple(wsh.shape), wsh.dtype, lambda: B().transpose2d(wsh), recipe=recipe if weight.dim() == 2 else None)


def sink_of(param):
    return getattr(param, "_ctclip_grad_sink", None)


def take_fresh_grad(param):
    """True once after the owning FusedAdam cleared its gradients (FusedAdam._mark_fresh): the caller -- which must be the only writer of this
    parameter's gradient, on the stream the clearing is ordered before -- may then OVERWRITE the (zero) gradient instead of accumulating into it.
    Anything else that clears gradients leaves the flag alone: the caller accumulates into zeros, which is merely slower."""
    fresh = getattr(param, "_ctclip_grad_fresh", False)
    if fresh:
        param._ctclip_grad_fresh = False
    return fresh


def _split_k_for(n_rows, n_cols, K, dtype):
    tiles = ((n_rows + 127) // 128) * ((n_cols + 127) // 128)
    bk = 32 if dtype == torch.float32 else 64
    ktiles = (K + bk - 1) // bk
    return max(1, min(ktiles, 1024 // max(tiles, 1)))


# ---- side streams whose kernels write parameter gradients straight into the flat gradient buffer (the text tower's stream): autograd's
# end-of-backward synchronisation only covers gradients it accumulates itself, so the optimiser joins these explicitly.
_SIDE_STREAMS = []


def register_side_stream(st):
    if all(st is not x for x in _SIDE_STREAMS):
        _SIDE_STREAMS.append(st)
    return st


_SHARED_STREAMS = {}


def shared_side_stream(device, name):
    """ONE side stream per (device, purpose) for the whole process, registered for the optimiser's join.  Every model instance used to create its
    own text-tower stream; the third model of a process (bench.py's T = 512 configuration after the 12+12 and 4+4 ones) then ran 4 ... 19 ms per
    step slower than the same configuration in a fresh process (box-dependent; consistent with HIP multiplexing streams onto a handful of hardware
    queues in creation order: a text stream that shares its queue with the main stream runs serialised behind the image tower).  With one stream per
    process a configuration no longer depends on what ran before it (tools/probe_config_sequence.py)."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), name)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        # (round 5) ... and the stream is PROBED for running beside the default stream: see streams.py
        st = _SHARED_STREAMS[key] = register_side_stream(_streams.concurrent_stream(device, name))
    return st


def reserve_side_streams(device):
    """Create the streams that carry kernels (text tower, weight gradients) NOW: hardware queues are handed out in the order of asking, and
    whoever sets up a communication stream should ask after these."""
    if torch.device(device).type == "cuda":
        shared_side_stream(device, "text")
        _wgrad_stream(torch.device(device))


def join_side_streams():
    """The current stream of each device waits for every registered side stream of that device."""
    for st in _SIDE_STREAMS:
        torch.cuda.current_stream(st.device).wait_stream(st)


# ---- weight-gradient stream.  dW = dy^T x is a leaf of the backward graph: nothing downstream of it runs before the optimiser (or the
# gradient all-reduce of its bucket).  Between wgrad_stream_begin() and wgrad_stream_end() (the trainer brackets loss.backward() with
# them) the big split-K GEMMs that write into the flat gradient buffer are launched on a side stream, UNDER the grad-input GEMMs of the
# main stream: the persistent kernels leave CUs idle in their last partial round (864 tiles on 256 CUs = 3.375 rounds) and the other
# stream's workgroups start there.  Off outside the bracket: a caller that reads gradients right after backward() sees them complete.
_WG = {"on": False, "streams": {}, "used": set()}


def wgrad_stream_begin():
    _WG["on"] = os.environ.get("CTCLIP_WGRAD_STREAM", "1") != "0"
    _WG["bracket"] = True      # inside the trainer's backward: the weight-gradient GEMMs take _side_wgs() workgroups whether or not the stream is used
                               # (CTCLIP_WGRAD_STREAM=0 then sums in the same order: test_zz_side_stream_backward_is_bit_identical)


def _wgrad_side(t):
    if not (_WG["on"] and t.is_cuda):
        return None
    dev = t.device
    if torch.cuda.current_stream(dev) != torch.cuda.default_stream(dev):
        return None            # the text tower's backward already runs on its own side stream
    return _wgrad_stream(dev)


def _wgrad_stream(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _WG["streams"].get(idx)
    if st is None:
        st = _WG["streams"][idx] = _streams.concurrent_stream(dev, "wgrad")
    return st


def wgrad_event():
    """An event behind everything launched on the weight-gradient stream(s) so far (for the gradient all-reduce), or None."""
    evs = [_WG["streams"][i].record_event() for i in _WG["used"]]
    return evs or None


def wgrad_stream_end():
    """Join: the current stream waits for the weight-gradient stream."""
    for i in _WG["used"]:
        torch.cuda.current_stream(torch.device("cuda", i)).wait_stream(_WG["streams"][i])
    _WG["used"].clear()
    _WG["on"] = False
    _WG["bracket"] = False


def weight_grad(dy, x, weight, segments, K):
    """dW[r0:r0+n, :K] (+)= dy[:, c0:c0+n]^T @ x[:, :K]   for (r0, n, c0) in segments.  Returns grad or None (sink)."""
    sink = sink_of(weight)
    side = _wgrad_side(dy) if (sink is not None and dy.shape[0] >= 4096) else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream(dy.device))      # dy and x were produced on the main stream
        dy.record_stream(side); x.record_stream(side)               # ... and may be freed there while the side stream still reads them
        _WG["used"].add(dy.device.index)
        with torch.cuda.stream(side):
            _weight_grad(dy, x, weight, segments, K, sink, wgs=_side_wgs())
        return None
    return _weight_grad(dy, x, weight, segments, K, sink, wgs=_side_wgs() if (_WG.get("bracket") and sink is not None and dy.shape[0] >= 4096) else 0)


def _side_wgs():
    """Workgroups (tiles x k-splits) of a weight-gradient GEMM launched on the SIDE stream: 192 of the 256 CUs.  The split-K kernel and the
    grad-input GEMM of the main stream are both persistent one-workgroup-per-CU kernels: with 256 workgroups each, the second to arrive waits
    for whole CUs; 192 leave a quarter of the chip to the main stream and write a quarter less split-K slab traffic.  Measured (same box, twice
    each, ms per 12+12 step): 256: 79.4 / 79.5; 224: 78.6; 208: 78.9; 192: 78.0 / 78.1; 176: 78.4; 160: 78.1 / 78.5; 128: 81.2.
    Outside the trainer's backward bracket (fine-tuning heads, the attention-block benchmark) a launch keeps the whole chip."""
    return int(os.environ.get("CTCLIP_WGRAD_WGS", "192"))


def _split_for(wgs, rows, cols):
    """k-split request of ctclip_gemm for a (rows x cols) weight gradient limited to `wgs` workgroups of 256 x 256 tiles (0 = the library's choice)"""
    if not wgs:
        return 0
    tiles = ((rows + 255) // 256) * ((cols + 255) // 256)
    return max(1, wgs // tiles)


def _weight_grad(dy, x, weight, segments, K, sink, wgs=0):
    if sink is None:
        dst = torch.zeros(weight.shape, dtype=torch.float32, device=dy.device)
    else:
        dst = sink
    dst2 = dst.view(weight.shape[0], -1)
    if (len(segments) > 1 and dy.dtype == torch.bfloat16 and dy.shape[0] >= 4096 and K == dst2.shape[1] and K % 4 == 0
            and all(n * K % 4 == 0 for _, n, _ in segments)):
        # stacked outputs (GEGLU in-projection [x | pad | gate | pad]): ONE pass over dy and x into a scratch, then the row blocks are
        # added to their places -- two launches per segment read x (113 MB) once per segment
        c_lo, c_hi = min(c0 for _, _, c0 in segments), max(c0 + n for _, n, c0 in segments)
        tmp = torch.empty((c_hi - c_lo, K), dtype=torch.float32, device=dy.device)
        B().gemm(dy[:, c_lo:c_hi], x[:, :K], a_kc=False, b_kc=False, out=tmp, accumulate=False, split_k=_split_for(wgs, c_hi - c_lo, K),
                 M=c_hi - c_lo, N=K, K=dy.shape[0])
        for (r0, n, c0) in segments:
            B().accumulate(dst2[r0:r0 + n], tmp[c0 - c_lo:c0 - c_lo + n])
        return None if sink is not None else dst
    for (r0, n, c0) in segments:
        B().gemm(dy[:, c0:c0 + n], x[:, :K], a_kc=False, b_kc=False, out=dst2[r0:r0 + n, :K], accumulate=True,
                 split_k=_split_for(wgs, n, K), M=n, N=K, K=dy.shape[0])
    return None if sink is not None else dst


def weight_bias_grad(dy, x, weight, bias, K):
    """Weight AND bias gradient of a Linear in one launch (ctclip_gemm_dw_db: the column sums of dy ride the dW GEMM's A fragments) for the text
    tower's sizes: -> (dw, db) as weight_grad / vec_grad would return them (None = written to the flat gradient buffer), or None when the shape
    is not served."""
    if not (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and weight.dim() == 2 and 128 <= dy.shape[0] <= 16384 and dy.shape[0] % 64 == 0
            and K == weight.shape[1] and os.environ.get("CTCLIP_DW_DB", "1") != "0"):
        return None
    wsink, bsink = sink_of(weight), sink_of(bias)
    dstw = wsink if wsink is not None else torch.zeros(weight.shape, dtype=torch.float32, device=dy.device)
    dstb = bsink if bsink is not None else torch.zeros(bias.shape, dtype=torch.float32, device=dy.device)
    if not B().gemm_dw_db(dy, x[:, :K], dstw.view(weight.shape[0], -1), dstb, accumulate=True):
        return None
    return (None if wsink is not None else dstw), (None if bsink is not None else dstb)


def vec_grad(param, compute):
    """compute(dst) accumulates a vector gradient into dst (f32, param-shaped).  Returns grad or None (sink)."""
    sink = sink_of(param)
    dst = sink if sink is not None else torch.zeros(param.shape, dtype=torch.float32, device=param.device)
    compute(dst)
    return None if sink is not None else dst


# ------------------------------------------------------------------------------------------ gradient-ready notifications
# Data-parallel training overlaps the gradient all-reduce with backward (distributed.GradReducer).  Weight gradients are written
# straight into the flat gradient buffer by the kernels, so autograd's own parameter hooks never fire; instead the model marks
# the INPUT activation of each block: when backward reaches the marker, every parameter used after it has its final gradient.

_GRAD_READY_HOOK = None


def set_grad_ready_hook(fn):
    """fn(tag) is called from backward when the gradients of the parameters registered under `tag` are final; None = off."""
    global _GRAD_READY_HOOK
    prev, _GRAD_READY_HOOK = _GRAD_READY_HOOK, fn
    return prev


def notify_grad_ready(tag):
    if _GRAD_READY_HOOK is not None:
        _GRAD_READY_HOOK(tag)


class _GradReadyFn(Function):
    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        notify_grad_ready(ctx.tag)
        return dy, None


def grad_ready(x, tag):
    """Identity; in backward, announces `tag` (no-op and no graph node unless a hook is installed and x carries gradient)."""
    if _GRAD_READY_HOOK is None or not (torch.is_grad_enabled() and x.requires_grad):
        return x
    return _GradReadyFn.apply(x, tag)


# ------------------------------------------------------------------------------------------ Linear

class LinearFn(Function):
    """y = x @ Wshadow^T (+ bias) (+ residual).  ``segments``/``K`` describe how dW maps back onto the real weight."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, wsh, segments, K, out_dtype, comp=None, passthrough=False):
        """passthrough: also return a VIEW of x for x's other consumer (BERT's residual connections read the tensor the dense layer reads):
        backward then receives that consumer's gradient and adds it in the grad-input GEMM's epilogue (no elementwise accumulation kernel).
        comp = the residue e of `residual`: the residual add runs on the compensated stream -> (y, e_out) (see residual_comp_enabled).
        Mixed precision (x f32, wsh bf16 -- the text tower's default in bf16 mode): x is rounded ONCE to the operand dtype for the matrix
        cores, the product is accumulated, biased, added to the residual and stored in f32; backward rounds dy the same way and returns dx
        in x's dtype."""
        e_out = None
        ctx.x_dtype = x.dtype
        ctx.passthrough = bool(passthrough)
        x_in = x
        if x.dtype != wsh.dtype:
            x = B().convert_pad(x, x.shape[0], x.shape[1], wsh.dtype)
        if comp is not None and bias is None and residual is not None:
            pair = B().gemm_residual_comp(x, wsh, residual, comp)
            if pair is not None:
                y, e_out = pair
        if e_out is None:
            y = B().gemm(x, wsh, bias=bias.detach() if bias is not None else None, residual=residual,
                         out_dtype=out_dtype or ctx.x_dtype)
        ctx.save_for_backward(x, wsh)
        ctx.weight, ctx.bias, ctx.segments, ctx.K = weight, bias, segments, K
        ctx.has_res = residual is not None
        ctx.res_dtype = residual.dtype if residual is not None else None
        if passthrough:
            assert comp is None
            return y, x_in.view_as(x_in)
        if comp is None:
            return y
        if e_out is None:      # shape not served: plain bf16 rounding at this add, the incoming residue travels on
            e_out = comp.clone()
        ctx.mark_non_differentiable(e_out)
        return y, e_out

    @staticmethod
    def backward(ctx, dy, _de=None):
        x, wsh = ctx.saved_tensors
        dy = dy.contiguous()
        d_pass = _de.contiguous() if (ctx.passthrough and _de is not None) else None      # gradient of x's other consumer: added in the dX epilogue
        dres = None
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy if dy.dtype == ctx.res_dtype else dy.to(ctx.res_dtype)
        ce = 4 if x.dtype == torch.float32 else 8
        nout = dy.shape[1]
        if dy.dtype != x.dtype or nout % ce:
            # (tiny layers only, e.g. the position-bias MLP's heads-wide output) 16-byte aligned rows for the kernels
            dyc = B().convert_pad(dy, dy.shape[0], round_up(nout, 8), x.dtype)[:, :nout]
        else:
            dyc = dy
        dw, db, fused_wb = None, None, None
        if (ctx.weight.requires_grad and ctx.bias is not None and ctx.bias.requires_grad and len(ctx.segments) == 1
                and ctx.segments[0] == (0, ctx.weight.shape[0], 0) and dyc.shape[1] == ctx.weight.shape[0]):
            # dW and db in one launch (text tower sizes: ctclip_gemm_dw_db serves T % 64 == 0, 128 <= T <= 16384).  NOTE the numerics: this path sums
            # the bf16-ROUNDED dy (dyc) for the bias gradient, the unfused path below the unrounded f32 dy -- a relative difference of 2^-9 / sqrt(T)
            # per element, two orders below the text gradients' agreement with the reference (4.4e-5 on tests/golden/full2.pt), but it means the
            # bias-gradient rounding depends on whether the shape is served by the fused kernel; the weight-gradient side stream is not used here.
            fused_wb = weight_bias_grad(dyc, x, ctx.weight, ctx.bias, ctx.K)
        if fused_wb is not None:
            dw, db = fused_wb
        elif ctx.weight.requires_grad:      # first: on the weight-gradient stream it then runs under the grad-input GEMM below
            dw = weight_grad(dyc, x, ctx.weight, ctx.segments, ctx.K)
        dx = None
        if ctx.needs_input_grad[0]:
            if dyc.dtype == torch.bfloat16 and ((dyc.shape[0] >= 4096 and dyc.shape[1] % 32 == 0) or
                                                (dyc.shape[0] >= 256 and dyc.shape[1] % 64 == 0 and dyc.shape[1] >= 128 and wsh.shape[1] >= 64)):
                # grad-input GEMM against the TRANSPOSED weight shadow: both operands k-contiguous (the LDS-DMA kernels: gemm_nt.hip for the
                # image tower's sizes, gemm_sm.hip for the text tower's)
                wt = transposed_shadow(ctx.weight, wsh, ctx.segments)
                dx = B().gemm(dyc, wt, residual=d_pass, out_dtype=ctx.x_dtype)
            else:
                dx = B().gemm(dyc, wsh, a_kc=True, b_kc=False, residual=d_pass, out_dtype=ctx.x_dtype)
            if x.stride(0) != x.shape[1]:  # strided-view input (e.g. CLS rows): match its logical shape
                dx = dx[:, :x.shape[1]]
        elif d_pass is not None:
            dx = d_pass
        if fused_wb is None and ctx.bias is not None and ctx.bias.requires_grad:
            dyb = dy if (dy.dtype == torch.float32 and dy.stride(1) == 1) else dyc      # (mixed precision: the bias gradient sums the unrounded dy)
            db = vec_grad(ctx.bias, lambda dst: B().colsum(dyb, dst, N=ctx.bias.numel()))
        return dx, dw, db, dres, None, None, None, None, None, None


def linear(x, weight, bias=None, residual=None, out_dtype=None, kpad=None, comp=None, operand_dtype=None, passthrough=False):
    """nn.Linear on a (M, K[p]) activation.  kpad: activation/weight K padding (zeros).  comp = the residue e of `residual`: the residual add on the
    compensated residual stream -> (y, e_out).  operand_dtype: dtype of the matrix-core operands when it differs from the activation's
    (mixed precision: f32 activations, bf16 operands, f32 accumulate / bias / residual / output)."""
    N, K = weight.shape
    wsh = plain_shadow(weight, operand_dtype or x.dtype, kpad=kpad)
    return LinearFn.apply(x, weight, bias, residual, wsh, [(0, N, 0)], K, out_dtype, comp, passthrough)


def geglu_hidden_pad(inner):
    return round_up(inner, 128)


def linear_geglu_in(x, weight):
    """FeedForward[1]: Linear(d, 2*inner, no bias) (attention.py:48) producing the padded [x | gate] layout (M, 2*Hp)."""
    two_inner, K = weight.shape
    inner = two_inner // 2
    Hp = geglu_hidden_pad(inner)

    def make():
        w = weight.detach()
        out = torch.empty((2 * Hp, K), dtype=x.dtype, device=w.device)
        B().convert_pad(w[:inner], Hp, K, x.dtype, out=out[:Hp])
        B().convert_pad(w[inner:], Hp, K, x.dtype, out=out[Hp:])
        return out
    wsh = shadow(weight, ("geglu_in", Hp), x.dtype, make, recipe=[(weight, 0, 2 * Hp, K, MAP_GEGLU_SPLIT, inner, False)])
    return LinearFn.apply(x, weight, None, None, wsh, [(0, inner, 0), (inner, inner, Hp)], K, None)


class FfInGegluFn(Function):
    """FeedForward[1] + GEGLU in one GEMM launch (bf16, whole 256-row tiles): the in-projection weight's rows are interleaved in
    groups of four so that the epilogue lane that owns an x column owns its gate.  Default backward: the launch also stores
    u = [x | gate] in the split layout and geglu_bwd streams it.  CTCLIP_GEGLU_RECOMPUTE=1 keeps NOTHING but the layer input: the
    backward launch recomputes (x, gate) with the same GEMM and writes du = [dg gelu(gate) | dg x gelu'(gate)] from its epilogue
    (ctclip_gemm_geglu_bwd) -- 2 Hp bf16 values per token and layer less memory (15 GB at 12+12 layers, batch 8) for ~45 us more
    per layer (measured in the step: 297 + 377 us against 330 + 300 us, profiles/r02b_*)."""

    @staticmethod
    def forward(ctx, x, weight, wsh, w_il, Hp, inner, K):
        recompute = os.environ.get("CTCLIP_GEGLU_RECOMPUTE", "0") == "1"
        u, g = B().gemm_geglu(x, w_il, Hp, save_u=not recompute)
        if recompute:
            ctx.save_for_backward(x, wsh, w_il)
        else:
            ctx.save_for_backward(x, wsh, u)
        ctx.recompute = recompute
        ctx.weight, ctx.dims = weight, (Hp, inner, K)
        return g

    @staticmethod
    def backward(ctx, dg):
        x, wsh, third = ctx.saved_tensors
        Hp, inner, K = ctx.dims
        be = B()
        dg = dg.contiguous()
        du = be.gemm_geglu_bwd(x, third, dg, Hp) if ctx.recompute else be.geglu_bwd(dg, third)
        dw = weight_grad(du, x, ctx.weight, [(0, inner, 0), (inner, inner, Hp)], K) if ctx.weight.requires_grad else None
        dx = None
        if ctx.needs_input_grad[0]:
            if du.shape[0] >= 4096:
                wt = transposed_shadow(ctx.weight, wsh, [(0, inner, 0), (inner, inner, Hp)])
                dx = be.gemm(du, wt)
            else:
                dx = be.gemm(du, wsh, a_kc=True, b_kc=False)
        return dx, dw, None, None, None, None, None


def feed_forward_in(x, weight):
    """LayerNormed tokens -> GEGLU hidden (M, Hp): fused launch when the large-tile kernel serves the shape, else GEMM + GEGLU kernel."""
    two_inner, K = weight.shape
    inner = two_inner // 2
    Hp = geglu_hidden_pad(inner)
    M = x.shape[0]
    if x.dtype == torch.bfloat16 and M % 256 == 0 and (2 * Hp) % 256 == 0 and (M // 256) * (2 * Hp // 256) >= 160 and K % 64 == 0 and K >= 128:
        def make():
            w = weight.detach()
            out = torch.empty((2 * Hp, K), dtype=x.dtype, device=w.device)
            B().convert_pad(w[:inner], Hp, K, x.dtype, out=out[:Hp])
            B().convert_pad(w[inner:], Hp, K, x.dtype, out=out[Hp:])
            return out
        wsh = shadow(weight, ("geglu_in", Hp), x.dtype, make, recipe=[(weight, 0, 2 * Hp, K, MAP_GEGLU_SPLIT, inner, False)])
        w_il = shadow(weight, ("geglu_il", Hp), x.dtype, lambda: B().geglu_weight_interleave(weight.detach(), Hp, x.dtype),
                      recipe=[(weight, 0, 2 * Hp, K, MAP_GEGLU_INTERLEAVE, inner, False)])
        return FfInGegluFn.apply(x, weight, wsh, w_il, Hp, inner, K)
    return GegluFn.apply(linear_geglu_in(x, weight))


class FeedForwardFn(Function):
    """FeedForward[1..4] (attention.py:44-52 without its LayerNorm) + the residual add as ONE autograd node on the fused bf16 path:
      forward   u = [x | gate], g = x gelu(gate)   (one launch, GEGLU in the in-projection epilogue);   out = g W_out^T + residual
      backward  dW_out = dout^T g;   du = [dg gelu(gate) | dg x gelu'(gate)] with dg = dout W_out formed ONLY in the accumulators of the
                grad-input GEMM (ctclip_gemm_dgeglu: no dg tensor, no streaming geglu_bwd pass);   dy = du W_in;   dW_in = du^T y.
    CTCLIP_GEGLU_RECOMPUTE=1: u is not stored; the backward recomputes it (ctclip_gemm_geglu_bwd) from y after an ordinary dg GEMM."""

    @staticmethod
    def forward(ctx, y, w_in, w_out, residual, wsh_in, w_il, wsh_out, Hp, inner, K, need_bwd=True, comp=None):
        be = B()
        recompute = os.environ.get("CTCLIP_GEGLU_RECOMPUTE", "0") == "1"
        u, g = be.gemm_geglu(y, w_il, Hp, save_u=need_bwd and not recompute)      # inference / frozen towers: u is never read, not written
        e_out = None
        if comp is not None and residual is not None:      # the residual add on the compensated stream (see residual_comp_enabled)
            pair = be.gemm_residual_comp(g, wsh_out, residual, comp)
            if pair is not None:
                out, e_out = pair
        if e_out is None:
            out = be.gemm(g, wsh_out, residual=residual)
        if need_bwd:
            ctx.save_for_backward(y, g, wsh_in, wsh_out, w_il if recompute else u)
            ctx.recompute, ctx.has_res = recompute, residual is not None
            ctx.w_in, ctx.w_out, ctx.dims = w_in, w_out, (Hp, inner, K)
        if comp is None:
            return out
        if e_out is None:
            e_out = comp.clone()
        ctx.mark_non_differentiable(e_out)
        return out, e_out

    @staticmethod
    def backward(ctx, dout, _de=None):
        y, g, wsh_in, wsh_out, last = ctx.saved_tensors
        Hp, inner, K = ctx.dims
        be = B()
        dout = dout.contiguous()
        N = ctx.w_out.shape[0]
        dw_out = weight_grad(dout, g, ctx.w_out, [(0, N, 0)], inner) if ctx.w_out.requires_grad else None
        wt_out = transposed_shadow(ctx.w_out, wsh_out, [(0, N, 0)])                   # (Hp, N): hidden feature j in row j
        du = None
        if not ctx.recompute:
            du = be.gemm_dgeglu(dout, wt_out, last)
        if du is None:
            dg = be.gemm(dout, wt_out)
            du = be.gemm_geglu_bwd(y, last, dg, Hp) if ctx.recompute else be.geglu_bwd(dg, last)
        segs = [(0, inner, 0), (inner, inner, Hp)]
        dw_in = weight_grad(du, y, ctx.w_in, segs, K) if ctx.w_in.requires_grad else None      # (weight-gradient stream: under the next GEMM)
        dy = be.gemm(du, transposed_shadow(ctx.w_in, wsh_in, segs)) if ctx.needs_input_grad[0] else None
        dres = dout if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dy, dw_in, dw_out, dres, None, None, None, None, None, None, None, None


def feed_forward(y, w_in, w_out, residual=None, comp=None):
    """LayerNormed tokens -> FeedForward output (+ residual).  One autograd node with the fused launches when the large-tile kernel
    serves the shape (bf16, whole 256-row tiles), otherwise the composition of the separate pieces.  comp = the residue e of `residual`: the residual add
    on the compensated residual stream -> (out, e_out)."""
    two_inner, K = w_in.shape
    inner = two_inner // 2
    Hp = geglu_hidden_pad(inner)
    M = y.shape[0]
    fused = (y.dtype == torch.bfloat16 and M % 256 == 0 and (2 * Hp) % 256 == 0 and (M // 256) * (2 * Hp // 256) >= 160 and K % 64 == 0 and K >= 128
             and (residual is None or residual.dtype == y.dtype) and os.environ.get("CTCLIP_FF_NODE", "1") != "0")
    if not fused:
        return linear_geglu_out(feed_forward_in(y, w_in), w_out, residual, comp)

    def make():
        w = w_in.detach()
        out = torch.empty((2 * Hp, K), dtype=y.dtype, device=w.device)
        B().convert_pad(w[:inner], Hp, K, y.dtype, out=out[:Hp])
        B().convert_pad(w[inner:], Hp, K, y.dtype, out=out[Hp:])
        return out
    wsh_in = shadow(w_in, ("geglu_in", Hp), y.dtype, make, recipe=[(w_in, 0, 2 * Hp, K, MAP_GEGLU_SPLIT, inner, False)])
    w_il = shadow(w_in, ("geglu_il", Hp), y.dtype, lambda: B().geglu_weight_interleave(w_in.detach(), Hp, y.dtype),
                  recipe=[(w_in, 0, 2 * Hp, K, MAP_GEGLU_INTERLEAVE, inner, False)])
    wsh_out = plain_shadow(w_out, y.dtype, kpad=Hp)
    need_bwd = torch.is_grad_enabled() and (y.requires_grad or w_in.requires_grad or w_out.requires_grad or
                                            (residual is not None and residual.requires_grad))
    return FeedForwardFn.apply(y, w_in, w_out, residual, wsh_in, w_il, wsh_out, Hp, inner, K, need_bwd, comp)


def linear_geglu_out(g, weight, residual, comp=None):
    """FeedForward[4]: Linear(inner, d, no bias) (attention.py:51) consuming the padded hidden (M, Hp), + residual (comp: on the
    compensated residual stream -> (out, e_out))."""
    N, inner = weight.shape
    Hp = g.shape[1]
    wsh = plain_shadow(weight, g.dtype, kpad=Hp)
    return LinearFn.apply(g, weight, None, residual, wsh, [(0, N, 0)], inner, None, comp)


# ------------------------------------------------------------------------------------------ LayerNorm

def _layernorm_bwd(dy, x, g, mean, rstd, dgam, dbet, sunk, add1=None, add2=None):
    """LayerNorm backward through the backend.  CTCLIP_LN_REDUCE_SIDE=1: inside the trainer's backward (wgrad_stream_begin) on the image tower's
    big token grids, with the parameter gradients going to the flat gradient buffer (`sunk`), the dgamma / dbeta fold -- a leaf of the backward
    graph, 13 us of latency by 16 workgroups, 76 times per step -- is launched on the weight-gradient side stream instead of in front of the next
    grad-input kernel.  Measured in round 5 (same box, 2 x 20 steps each): 84.28 / 84.36 ms against 84.04 / 84.01 with the fold in line -- the
    two extra stream hand-offs per LayerNorm cost what the fold's latency saved.  Off by default; the split entry points stay."""
    be = B()
    side = _wgrad_side(dy) if (sunk and (dgam is not None or dbet is not None) and dy.shape[0] >= 4096
                               and os.environ.get("CTCLIP_LN_REDUCE_SIDE", "0") == "1") else None      # (measured: no gain, +0.3 ms -- off by default)
    if side is None:
        return be.layernorm_bwd(dy, x, g, mean, rstd, dgam, dbet, add1, add2)
    dx, part = be.layernorm_bwd_partials(dy, x, g, mean, rstd, add1, add2)
    side.wait_stream(torch.cuda.current_stream(dy.device))
    part.record_stream(side)
    _WG["used"].add(dy.device.index)
    with torch.cuda.stream(side):
        be.layernorm_bwd_reduce(part, dgam, dbet, x.shape[0], x.shape[1])
    return dx


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = B().layernorm_fwd(x, gamma.detach() if gamma is not None else None,
                                          beta.detach() if beta is not None else None, eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        g, b = ctx.gamma, ctx.beta
        want_g = g is not None and g.requires_grad
        want_b = b is not None and b.requires_grad
        gs = sink_of(g) if want_g else None
        bs = sink_of(b) if want_b else None
        dgam = (gs if gs is not None else torch.zeros_like(g, dtype=torch.float32)) if want_g else None
        dbet = (bs if bs is not None else torch.zeros_like(b, dtype=torch.float32)) if want_b else None
        sunk = (not want_g or gs is not None) and (not want_b or bs is not None)
        dx = _layernorm_bwd(dy.contiguous(), x, g.detach() if g is not None else None, mean, rstd, dgam, dbet, sunk)
        return dx, (None if (not want_g or gs is not None) else dgam), (None if (not want_b or bs is not None) else dbet), None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


class LayerNormBranchFn(Function):
    """y = LayerNorm(x) together with `nviews` pass-through views of x for x's OTHER consumers (the residual connection,
    the k/v projection of the raw tokens: attention.py:139-143, 324-326).  Routing those uses through this node lets the
    backward add their gradients inside the LayerNorm backward kernel (ctclip_layernorm_bwd add1/add2) instead of through
    autograd's elementwise accumulation kernels -- three passes over a 113-MB tensor each, 72 times per step."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, nviews):
        y, mean, rstd = B().layernorm_fwd(x, gamma.detach() if gamma is not None else None,
                                          beta.detach() if beta is not None else None, eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta = gamma, beta
        return (y,) + tuple(x.view_as(x) for _ in range(nviews))

    @staticmethod
    def backward(ctx, dy, *dviews):
        x, mean, rstd = ctx.saved_tensors
        g, b = ctx.gamma, ctx.beta
        want_g = g is not None and g.requires_grad
        want_b = b is not None and b.requires_grad
        gs = sink_of(g) if want_g else None
        bs = sink_of(b) if want_b else None
        dgam = (gs if gs is not None else torch.zeros_like(g, dtype=torch.float32)) if want_g else None
        dbet = (bs if bs is not None else torch.zeros_like(b, dtype=torch.float32)) if want_b else None
        adds = [d.contiguous() for d in dviews if d is not None]
        extra = None
        while len(adds) > 2:                      # the kernel takes two addends
            extra = adds.pop() if extra is None else extra + adds.pop()
        if extra is not None:
            adds[0] = adds[0] + extra
        if dy is None:
            dy = torch.zeros_like(x)
        sunk = (not want_g or gs is not None) and (not want_b or bs is not None)
        dx = _layernorm_bwd(dy.contiguous(), x, g.detach() if g is not None else None, mean, rstd, dgam, dbet, sunk,
                            adds[0] if len(adds) > 0 else None, adds[1] if len(adds) > 1 else None)
        return dx, (None if (not want_g or gs is not None) else dgam), (None if (not want_b or bs is not None) else dbet), None, None


def layer_norm_branch(x, gamma, beta, nviews, eps=1e-5):
    """-> (LayerNorm(x), x, ..., x): use the returned views of x wherever the unnormalised tensor is consumed as well."""
    return LayerNormBranchFn.apply(x, gamma, beta, eps, nviews)


# ------------------------------------------------------------------------------------------ patch embedding

class PatchEmbedFn(Function):
    """CTViT.to_patch_emb (ctvit.py:170-175): Rearrange + LayerNorm(K) + Linear(K, d) + LayerNorm(d).

    The first LayerNorm's affine is folded into the GEMM (W' = W * gamma1 per column, b' = W beta1 + b) so the
    gather kernel writes xhat once and no dX GEMM is needed for gamma1/beta1:
        dW = G * gamma1 + db' (x) beta1,  dgamma1 = sum_n W * G,  dbeta1 = W^T db',  G = dZ^T xhat.
    """

    @staticmethod
    def forward(ctx, video, g1, b1, W, bl, g2, b2, pt, p1, p2, dtype):
        be = B()
        N, K = W.shape
        kpad = round_up(K, 64)
        xhat = be.patch_ln(video, pt, p1, p2, kpad, 1e-5, dtype)
        g1d, b1d = g1.detach(), b1.detach()
        Wf = be.convert_pad(W.detach(), N, kpad, dtype, colscale=g1d)
        # b' = W beta1 + b  (f32 GEMV through the same GEMM kernel; K padded to a 16-byte multiple by construction)
        bf = be.gemm(W.detach(), b1d.view(1, K), residual=bl.detach().view(N, 1), out_dtype=torch.float32).view(N)
        z = be.gemm(xhat, Wf, bias=bf)
        y, mean, rstd = be.layernorm_fwd(z, g2.detach(), b2.detach(), 1e-5)
        ctx.save_for_backward(xhat, z, mean, rstd)
        ctx.params = (g1, b1, W, bl, g2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = B()
        xhat, z, mean, rstd = ctx.saved_tensors
        g1, b1, W, bl, g2, b2 = ctx.params
        N, K = W.shape
        g2s, b2s = sink_of(g2), sink_of(b2)
        dg2 = g2s if g2s is not None else torch.zeros_like(g2)
        db2 = b2s if b2s is not None else torch.zeros_like(b2)
        dz = be.layernorm_bwd(dy.contiguous(), z, g2.detach(), mean, rstd, dg2, db2)
        dbp = torch.zeros(N, dtype=torch.float32, device=dz.device)
        be.colsum(dz, dbp)
        G = torch.empty((N, K), dtype=torch.float32, device=dz.device)       # (overwritten: no 8-MB fill in front of the GEMM)
        be.gemm(dz, xhat[:, :K], a_kc=False, b_kc=False, out=G, accumulate=False,
                split_k=0, M=N, N=K, K=dz.shape[0])
        # parameter-space epilogue (N x K elementwise + two column reductions, one launch): straight into the flat gradient buffer
        def sink_or_zeros(param):
            s_ = sink_of(param)
            return (s_, True) if s_ is not None else (torch.zeros_like(param, dtype=torch.float32), False)
        (dW, w_sunk), (dg1, g_sunk), (db1, b_sunk) = sink_or_zeros(W), sink_or_zeros(g1), sink_or_zeros(b1)
        if w_sunk == g_sunk == b_sunk:
            be.patch_embed_param_bwd(G, W.detach().contiguous(), g1.detach(), b1.detach(), dbp, dW, dg1, db1, accumulate=w_sunk)
        else:
            # a partially registered / partially frozen patch embedding: the kernel accumulates or overwrites all three together, so it runs
            # into temporaries and the results are added to whichever flat-gradient views exist
            tW, tg, tb = torch.zeros_like(W, dtype=torch.float32), torch.zeros_like(g1, dtype=torch.float32), torch.zeros_like(b1, dtype=torch.float32)
            be.patch_embed_param_bwd(G, W.detach().contiguous(), g1.detach(), b1.detach(), dbp, tW, tg, tb, accumulate=False)
            for dst, sunk, t in ((dW, w_sunk, tW), (dg1, g_sunk, tg), (db1, b_sunk, tb)):
                if sunk:
                    be.accumulate(dst, t)
                else:
                    dst.copy_(t)
        bls = sink_of(bl)
        if bls is not None:
            be.accumulate(bls, dbp)
        return (None, None if g_sunk else dg1, None if b_sunk else db1, None if w_sunk else dW, None if bls is not None else dbp,
                None if g2s is not None else dg2, None if b2s is not None else db2, None, None, None, None)


# ------------------------------------------------------------------------------------------ PEG

class PegFn(Function):
    @staticmethod
    def forward(ctx, x5, weight, bias, comp=None):
        """comp: None = plain add; otherwise the compensated residual stream (see residual_comp_enabled): comp = the incoming residue e (a
        tensor shaped like x5, or False for "none yet") and the result is (y, e_out)."""
        w27 = weight.detach().reshape(weight.shape[0], 27)
        e_out = None
        if comp is not None:
            pair = B().peg_fwd_comp(x5, w27, bias.detach(), comp if torch.is_tensor(comp) else None)
            if pair is not None:
                y, e_out = pair
        if e_out is None:
            y = B().peg_fwd(x5, w27, bias.detach())
        ctx.save_for_backward(x5)
        ctx.weight, ctx.bias = weight, bias
        if comp is None:
            return y
        if e_out is None:      # grids the marching kernels do not serve: plain rounding at this add, the incoming residue travels on
            e_out = comp.reshape(y.shape).clone() if torch.is_tensor(comp) else torch.zeros_like(y)      # (a tensor of its own, as LinearFn / FeedForwardFn)
        ctx.mark_non_differentiable(e_out)
        return y, e_out

    @staticmethod
    def backward(ctx, dy, _dr=None):
        (x5,) = ctx.saved_tensors
        w, b = ctx.weight, ctx.bias
        ws, bs = sink_of(w), sink_of(b)
        dw = ws.view(-1, 27) if ws is not None else torch.zeros((w.shape[0], 27), dtype=torch.float32, device=dy.device)
        db = bs if bs is not None else torch.zeros_like(b)
        dy = dy.contiguous()
        w27 = w.detach().reshape(w.shape[0], 27)
        side = _wgrad_side(dy) if (ws is not None and bs is not None) else None
        if side is not None:       # the weight / bias gradient is a leaf: on the weight-gradient stream, under the grad-input kernel
            side.wait_stream(torch.cuda.current_stream(dy.device))
            dy.record_stream(side); x5.record_stream(side)
            _WG["used"].add(dy.device.index)
            with torch.cuda.stream(side):
                B().peg_bwd(dy, x5, w27, dw, db, want_dx=False)
            dx = B().peg_bwd(dy, x5, w27, None, None)
        else:
            dx = B().peg_bwd(dy, x5, w27, dw, db)
        return dx, (None if ws is not None else dw.view_as(w)), (None if bs is not None else db), None


def peg_residual(x5, weight, bias, comp=None):
    """x + PEG(x) on a contiguous (b, D1, D2, D3, C) view (attention.py:63-84,324).  comp (the incoming residue tensor, or False for none
    yet): the add on the compensated residual stream -> (y, e_out)."""
    return PegFn.apply(x5, weight, bias, comp)


# ---- compensated residual stream.  bf16 storage rounds the residual stream at each of its three adds per layer; over 24 layers those 72
# roundings -- not the bf16 GEMM operands -- are the error of the bf16 mode (profiles/r03_bf16_error_budget.md: pre-VQ token error 1.6e-2 and
# 96.4 % code agreement with them, 7.3e-3 and 98.4 % without, measured at 12+12 layers).  The stream can therefore be carried as a bf16 PAIR
# (x, e): x is what every consumer reads (LayerNorm, the k/v projection, PEG's taps: one fresh rounding, not an accumulated one), e = the
# residue the last rounding cut off, added back in f32 inside the next residual add (PEG kernel / GEMM epilogue), which emits the next pair.
# e never enters autograd (backward is unchanged) and is dead after the next add.  It costs 680 MB of extra traffic per layer (+4.5 % of the
# training step, measured), so the default policy is CTCLIP_RESIDUAL_COMP=auto: ON whenever no gradient is being recorded (zero-shot scoring,
# latent export, frozen towers: fidelity of the latents to the f32 reference is what matters there), OFF inside a training forward.
# =1 forces it everywhere (the mixed-precision semantics of torch.autocast, whose residual stream stays f32), =0 disables it.
def residual_comp_enabled(x, kind=None, layer=0, nlayers=1):
    """kind / layer / nlayers: which transformer ("spatial" | "temporal") and which of its layers asks.  Partial forms (round 6, the cost of the
    full form in training is +4.2 ms per step at 12+12 layers): CTCLIP_RESIDUAL_COMP=temporal compensates the temporal transformer only (where
    the stream's error grows fastest), =lastN the last N layers of the temporal transformer; both apply with and without autograd."""
    if x.dtype != torch.bfloat16:
        return False
    mode = os.environ.get("CTCLIP_RESIDUAL_COMP", "auto").lower()
    if mode in ("0", "off"):
        return False
    if mode in ("1", "on"):
        return True
    if mode == "temporal" or mode.startswith("last"):
        if not torch.is_grad_enabled():
            return True                     # (inference keeps the full form)
        if kind != "temporal":
            return False
        return True if mode == "temporal" else layer >= nlayers - int(mode[4:])
    return not torch.is_grad_enabled()


# ------------------------------------------------------------------------------------------ attention

class CosineAttnFn(Function):
    """attention.py:145-178 (self-attention, no null kv): l2norm(q)*q_scale, l2norm(k)*k_scale, sim*8 (+bias), softmax, @v."""

    @staticmethod
    def forward(ctx, q, kv, q_scale, k_scale, bias, nseq, L, H, D, scale, bias_grid=None):
        """bias: (H, L, L) f32, or -- with bias_grid = (gh, gw) -- the (nclass, H) relative-position table itself."""
        be = B()
        HD = H * D
        k, v = kv[:, :HD], kv[:, HD:]
        qh, qinv = be.qk_norm_fwd(q, q_scale.detach(), H, D)
        kh, kinv = be.qk_norm_fwd(k, k_scale.detach(), H, D)
        vt = be.head_transpose(v, nseq, H, L, D)
        if bias is not None:
            bias = bias.contiguous()
        o, lse = be.attn_fwd(qh, kh, vt, bias, None, nseq, H, L, D, scale, bias_grid=bias_grid)
        ctx.save_for_backward(q, kv, qh, kh, qinv, kinv, o, lse, bias if bias is not None else q.new_empty(0))
        ctx.scales = (q_scale, k_scale)
        ctx.dims = (nseq, L, H, D, scale, bias is not None, bias_grid)
        return o

    @staticmethod
    def backward(ctx, do):
        be = B()
        q, kv, qh, kh, qinv, kinv, o, lse, bias = ctx.saved_tensors
        nseq, L, H, D, scale, has_bias, bias_grid = ctx.dims
        q_scale, k_scale = ctx.scales
        HD = H * D
        do = do.contiguous()
        k, v = kv[:, :HD], kv[:, HD:]
        qt = be.head_transpose(qh, nseq, H, L, D)
        kt = be.head_transpose(kh, nseq, H, L, D)
        dot = be.head_transpose(do, nseq, H, L, D)
        dqh = torch.empty_like(qh)
        dkv = torch.empty_like(kv)
        dkh = torch.empty_like(kh)
        dbias = torch.zeros_like(bias) if (has_bias and ctx.needs_input_grad[4]) else None
        be.attn_bwd(qh, kh, v, qt, kt, o, do, dot, lse, bias if has_bias else None, None, dqh, dkh, dkv[:, HD:], dbias,
                    nseq, H, L, D, scale, bias_grid=bias_grid)
        qs_sink, ks_sink = sink_of(q_scale), sink_of(k_scale)
        dqs = qs_sink if qs_sink is not None else torch.zeros_like(q_scale)
        dks = ks_sink if ks_sink is not None else torch.zeros_like(k_scale)
        dq = torch.empty_like(q)
        be.qk_norm_bwd(dqh, q, qinv, q_scale.detach(), dq, dqs, H, D)
        be.qk_norm_bwd(dkh, k, kinv, k_scale.detach(), dkv[:, :HD], dks, H, D)
        return (dq, dkv, None if qs_sink is not None else dqs, None if ks_sink is not None else dks, dbias,
                None, None, None, None, None, None)


class CosineAttn2Fn(Function):
    """The same operator on the second-generation kernels (csrc/attn2.hip): bf16, d_head 32, L % 32 == 0.  One prep pass writes the
    head-planar q~ / k^ / v (l2norm, learned scale and the logit scale folded in), the attention kernels read the position-bias
    table directly, and one un-prep pass applies the l2norm backward: no transposed copies, no separate qk-norm / delta kernels."""

    @staticmethod
    def forward(ctx, q, kv, q_scale, k_scale, tab, nseq, L, H, D, scale, bias_grid):
        be = B()
        HD = H * D
        qs, ks = q_scale.detach(), k_scale.detach()
        qh, kh, vh, qinv, kinv = be.attn2_prep(q, kv[:, :HD], kv[:, HD:], qs, ks, scale, H)
        tabc = tab.detach().contiguous() if tab is not None else None
        o, lse2 = be.attn2_fwd(qh, kh, vh, tabc, bias_grid, qs, ks, scale, nseq, L)
        ctx.save_for_backward(qh, kh, vh, qinv, kinv, o, lse2, tabc if tabc is not None else q.new_empty(0))
        ctx.scales = (q_scale, k_scale)
        ctx.dims = (nseq, L, H, D, scale, tab is not None, bias_grid, q.dtype)
        _attn2_register_table(ctx, tab)
        return o

    @staticmethod
    def backward(ctx, do):
        qh, kh, vh, qinv, kinv, o, lse2, tab = ctx.saved_tensors
        dq, dkv, dqs, dks, dtab = _attn2_backward(ctx, do, qh, kh, vh, qinv, kinv, o, lse2, tab, ctx.needs_input_grad[4])
        return (dq, dkv, dqs, dks, dtab, None, None, None, None, None, None)


def _attn2_register_table(ctx, tab):
    """The layers that share one position-bias table (ctvit.py:293): the table gradient of all of them is handed to autograd by the FIRST
    layer (the last one in backward) when the weight-gradient stream is on -- see _attn2_backward."""
    ctx.tab_users = None
    if tab is not None and tab.requires_grad:
        st = tab.__dict__.setdefault("_ctclip_tab_users", {"n": 0, "acc": None})
        ctx.tab_users, ctx.tab_index = st, st["n"]
        st["n"] += 1


def _attn2_backward(ctx, do, qh, kh, vh, qinv, kinv, o, lse2, tab, need_dtab):
    """Backward of the slab attention on prepared operands + the l2norm / scale backward (un-prep): -> token-major dq (M, HD), dkv (M, 2 HD),
    the scale gradients (None when they went to the flat gradient buffer) and the table gradient.  ctx carries dims / scales / tab_users.
    Fast path (ctclip_attn2_bwd_tok): the slab key pass writes row-major dk / dv itself, only dq^ takes the planar round trip."""
    be = B()
    nseq, L, H, D, scale, has_tab, bias_grid, dtype = ctx.dims
    q_scale, k_scale = ctx.scales
    qs, ks = q_scale.detach(), k_scale.detach()
    HD = H * D
    do = do.contiguous()
    M = o.shape[0]
    dq = torch.empty((M, HD), dtype=dtype, device=o.device)
    dkv = torch.empty((M, 2 * HD), dtype=dtype, device=o.device)
    qs_sink, ks_sink = sink_of(q_scale), sink_of(k_scale)
    dqs = qs_sink if qs_sink is not None else torch.zeros_like(q_scale)
    dks = ks_sink if ks_sink is not None else torch.zeros_like(k_scale)
    want_dtab = has_tab and need_dtab
    tabk = tab if has_tab else None
    # One pass over the score tiles (csrc/attn2_bwd1.hip): dq / dk / dv, both scale gradients AND the table gradient.  The layers that share
    # the table add theirs up in backward order; the first layer hands the sum to autograd.
    one = be.attn2_bwd_fused(qh, kh, vh, tabk, bias_grid, qs, ks, scale, o, do, lse2, qinv, kinv, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks, nseq, L,
                             want_dtab)
    if one is not None:
        dtab = one[0]
        st = ctx.tab_users
        if st is not None:
            if dtab is not None:
                if st["acc"] is None:
                    st["acc"] = dtab
                else:
                    be.accumulate(st["acc"], dtab)
                dtab = None
            if ctx.tab_index == 0:
                dtab, st["acc"], st["n"] = st["acc"], None, 0
        return dq, dkv, (None if qs_sink is not None else dqs), (None if ks_sink is not None else dks), dtab
    side = _wgrad_side(do) if (want_dtab and ctx.tab_users is not None) else None
    fused = None
    if os.environ.get("CTCLIP_ATTN_FUSED_UNPREP", "1") != "0":
        fused = be.attn2_bwd_tok(qh, kh, vh, tabk, bias_grid, qs, ks, scale, o, do, lse2, qinv, kinv, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks, nseq, L,
                                 want_dtab, defer_dtab=side is not None)
    if fused is not None:
        dtab, ws = fused
    elif side is not None:
        dqh, dkh, dvh, ws = be.attn2_bwd(qh, kh, vh, tab, bias_grid, qs, ks, scale, o, do, lse2, nseq, L, True, defer_dtab=True)
    else:
        dqh, dkh, dvh, dtab = be.attn2_bwd(qh, kh, vh, tabk, bias_grid, qs, ks, scale, o, do, lse2, nseq, L, want_dtab)
    if side is not None:
        # The table gradient is a leaf until the position-bias MLP's backward, which runs after the FIRST layer's attention backward.
        # Its pass (a third recomputation of S and dP) goes to the weight-gradient stream, under the rest of this layer's backward;
        # the layers' tables are summed there, in backward order, and the first layer joins the stream and returns the sum.
        st = ctx.tab_users
        side.wait_stream(torch.cuda.current_stream(do.device))
        for t in (qh, kh, vh, lse2, tab, ws):
            t.record_stream(side)
        _WG["used"].add(do.device.index)
        with torch.cuda.stream(side):
            d = be.attn2_bwd_dbias(qh, kh, vh, tab, bias_grid, qs, ks, scale, lse2, nseq, L, ws)
            if st["acc"] is None:
                st["acc"] = d
            else:
                be.accumulate(st["acc"], d)
        dtab = None
        if ctx.tab_index == 0:
            torch.cuda.current_stream(do.device).wait_stream(side)
            dtab, st["acc"], st["n"] = st["acc"], None, 0
            dtab.record_stream(torch.cuda.current_stream(do.device))
    elif ctx.tab_users is not None and ctx.tab_index == 0:
        ctx.tab_users["n"], ctx.tab_users["acc"] = 0, None
    if fused is None:
        be.attn2_unprep(dqh, dkh, dvh, qh, kh, qinv, kinv, qs, ks, scale, dq, dkv[:, :HD], dkv[:, HD:], dqs, dks)
    return dq, dkv, (None if qs_sink is not None else dqs), (None if ks_sink is not None else dks), dtab


class QkvAttn2Fn(Function):
    """to_q / to_kv + the slab attention as ONE node (attention.py:139-178 for the spatial transformer, bf16): the two projection GEMMs write
    the attention kernels' operands themselves -- head-planar q~ = l2norm(q) q_scale (scale log2 e), k^ = l2norm(k) k_scale, v, and the
    inverse norms -- from their epilogues (ctclip_gemm_headnorm: ctclip_attn2_prep's arithmetic on the bf16-rounded projection), so the
    token-major q / kv tensors and the prep pass (340 MB of traffic per layer) do not exist.  Backward = CosineAttn2Fn's (slab kernels +
    un-prep, which needs only the planar operands and the inverse norms) followed by the two Linear backwards."""

    @staticmethod
    def forward(ctx, xn, x_kv, wq, wkv, wsh_q, wsh_kv, q_scale, k_scale, tab, nseq, L, H, D, scale, bias_grid):
        be = B()
        qs, ks = q_scale.detach(), k_scale.detach()
        c = float(np.float32(scale) * np.float32(LOG2E))      # (f32 product, as ctclip_attn2_prep forms it)
        (qh, qinv), = be.gemm_headnorm(xn, wsh_q, [(qs, c)])
        (kh, kinv), (vh, _) = be.gemm_headnorm(x_kv, wsh_kv, [(ks, 1.0), (None, 1.0)])
        tabc = tab.detach().contiguous() if tab is not None else None
        o, lse2 = be.attn2_fwd(qh, kh, vh, tabc, bias_grid, qs, ks, scale, nseq, L)
        ctx.save_for_backward(xn, x_kv, wsh_q, wsh_kv, qh, kh, vh, qinv, kinv, o, lse2, tabc if tabc is not None else xn.new_empty(0))
        ctx.scales, ctx.weights = (q_scale, k_scale), (wq, wkv)
        ctx.dims = (nseq, L, H, D, scale, tab is not None, bias_grid, xn.dtype)
        _attn2_register_table(ctx, tab)
        return o

    @staticmethod
    def backward(ctx, do):
        be = B()
        xn, x_kv, wsh_q, wsh_kv, qh, kh, vh, qinv, kinv, o, lse2, tab = ctx.saved_tensors
        wq, wkv = ctx.weights
        dq, dkv, dqs, dks, dtab = _attn2_backward(ctx, do, qh, kh, vh, qinv, kinv, o, lse2, tab, ctx.needs_input_grad[8])
        outs = []
        for dy, x, w, wsh, need in ((dq, xn, wq, wsh_q, ctx.needs_input_grad[0]), (dkv, x_kv, wkv, wsh_kv, ctx.needs_input_grad[1])):
            N, K = w.shape
            segs = [(0, N, 0)]
            dw = weight_grad(dy, x, w, segs, K) if w.requires_grad else None      # (weight-gradient stream: under the grad-input GEMM below)
            dx = be.gemm(dy, transposed_shadow(w, wsh, segs)) if need else None
            outs.append((dx, dw))
        (dxn, dwq), (dxkv, dwkv) = outs
        return (dxn, dxkv, dwq, dwkv, None, None, dqs, dks, dtab, None, None, None, None, None, None)


LOG2E = 1.4426950408889634


def qkv_attention(xn, x_kv, wq, wkv, q_scale, k_scale, bias, nseq, L, H, D, scale, bias_grid=None):
    """q = to_q(xn), k | v = to_kv(x_kv), cosine attention (attention.py:139-178).  Spatial transformer in bf16 (table bias, slab kernels,
    whole 256-row tiles, inner width 256): one node with the operand layout written by the projection GEMMs (QkvAttn2Fn); anything else:
    two Linear nodes + cosine_attention.  CTCLIP_ATTN_FUSED_PREP=0 forces the composition."""
    M, HD = xn.shape[0], H * D
    table = bias is not None and bias_grid is not None
    fused = (xn.dtype == torch.bfloat16 and HD == 256 and D == 32 and M % 256 == 0 and (M // 256) >= 160 and xn.shape[1] % 64 == 0
             and xn.shape[1] >= 128 and xn.data_ptr() % 16 == 0 and x_kv.data_ptr() % 16 == 0      # (ctclip_gemm_nt_headnorm_try: two k-steps at least, 16-byte rows)
             and (bias is None or table) and not (bias is None and B().attn_short_supported(xn.dtype, L, D))
             and B().attn2_supported(xn.dtype, H, L, D, bias_grid if table else None, table)
             and os.environ.get("CTCLIP_ATTN_FUSED_PREP", "1") != "0")
    if fused:
        return QkvAttn2Fn.apply(xn, x_kv, wq, wkv, plain_shadow(wq, xn.dtype), plain_shadow(wkv, xn.dtype), q_scale, k_scale,
                                bias if table else None, nseq, L, H, D, scale, bias_grid if table else None)
    return cosine_attention(linear(xn, wq), linear(x_kv, wkv), q_scale, k_scale, bias, nseq, L, H, D, scale, bias_grid)


class CosineAttnShortFn(Function):
    """The same operator for sequences of at most 32 tokens (csrc/attn_short.hip: CTViT's temporal transformer): one wave per
    (sequence, head), q / kv read and o / dq / dkv written in place -- nothing saved but the inputs, no layout passes."""

    @staticmethod
    def forward(ctx, q, kv, q_scale, k_scale, nseq, L, H, scale):
        o = B().attn_short_fwd(q, kv, q_scale.detach(), k_scale.detach(), nseq, L, H, scale)
        ctx.save_for_backward(q, kv)
        ctx.scales = (q_scale, k_scale)
        ctx.dims = (nseq, L, H, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv = ctx.saved_tensors
        nseq, L, H, scale = ctx.dims
        q_scale, k_scale = ctx.scales
        qs_sink, ks_sink = sink_of(q_scale), sink_of(k_scale)
        dqs = qs_sink if qs_sink is not None else torch.zeros_like(q_scale)
        dks = ks_sink if ks_sink is not None else torch.zeros_like(k_scale)
        dq, dkv = B().attn_short_bwd(q, kv, q_scale.detach(), k_scale.detach(), do.contiguous(), nseq, L, H, scale, dqs, dks)
        return dq, dkv, None if qs_sink is not None else dqs, None if ks_sink is not None else dks, None, None, None, None


def cosine_attention(q, kv, q_scale, k_scale, bias, nseq, L, H, D, scale, bias_grid=None):
    """attention.py:145-178 on whichever kernel generation serves the shape (bias: the (ncls, H) table when bias_grid is given)."""
    if bias is None and B().attn_short_supported(q.dtype, L, D):
        return CosineAttnShortFn.apply(q, kv, q_scale, k_scale, nseq, L, H, scale)
    table = bias is not None and bias_grid is not None
    if (bias is None or table) and B().attn2_supported(q.dtype, H, L, D, bias_grid if table else None, table):
        return CosineAttn2Fn.apply(q, kv, q_scale, k_scale, bias if table else None, nseq, L, H, D, scale, bias_grid if table else None)
    return CosineAttnFn.apply(q, kv, q_scale, k_scale, bias, nseq, L, H, D, scale, bias_grid)


class SdpaFn(Function):
    """HF BertSelfAttention core: softmax(q k^T / sqrt(d) + mask) v; q, k, v are (M, H*D) activations."""

    @staticmethod
    def forward(ctx, q, k, v, keymask, nseq, L, H, D, scale, dropout=None):
        """dropout = (p, seed) or None: HF attention_probs_dropout_prob in train mode."""
        be = B()
        vt = be.head_transpose(v, nseq, H, L, D)
        o, lse = be.attn_fwd(q, k, vt, None, keymask, nseq, H, L, D, scale, dropout=dropout)
        ctx.save_for_backward(q, k, v, o, lse, keymask if keymask is not None else q.new_empty(0))
        ctx.dims = (nseq, L, H, D, scale, keymask is not None, dropout)
        return o

    @staticmethod
    def backward(ctx, do):
        be = B()
        q, k, v, o, lse, keymask = ctx.saved_tensors
        nseq, L, H, D, scale, has_mask, dropout = ctx.dims
        do = do.contiguous()
        qt = be.head_transpose(q, nseq, H, L, D)
        kt = be.head_transpose(k, nseq, H, L, D)
        dot = be.head_transpose(do, nseq, H, L, D)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        be.attn_bwd(q, k, v, qt, kt, o, do, dot, lse, None, keymask if has_mask else None, dq, dk, dv, None,
                    nseq, H, L, D, scale, dropout=dropout)
        return dq, dk, dv, None, None, None, None, None, None, None


class QkvSdpaFn(Function):
    """HF BertSelfAttention as one node: q, k, v = x Wq^T + bq, x Wk^T + bk, x Wv^T + bv as ONE GEMM against the stacked weight
    shadow (3 * hidden outputs: three times the column tiles for the M = B * T = a-few-hundred-rows text tower, one launch instead
    of three), softmax(q k^T / sqrt(d) + mask) v on column views of that buffer; backward writes dq | dk | dv side by side and
    runs one grad-input GEMM and three weight-gradient GEMMs on column views."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, bq, bk, bv, keymask, nseq, L, H, D, scale, dropout, operand_dtype=None, passthrough=False):
        """passthrough: also return a view of x for the residual connection (its gradient is added in the grad-input GEMM's epilogue).
        operand_dtype (mixed precision): x arrives in f32, is rounded once to the operand dtype; q | k | v, the attention core and its
        output are in the operand dtype (they are matrix-core operands of QK^T, PV and the output projection), dx returns in f32."""
        be = B()
        N, K = wq.shape
        ctx.x_dtype = x.dtype
        ctx.passthrough = bool(passthrough)
        x_in = x
        if operand_dtype is not None and x.dtype != operand_dtype:
            x = be.convert_pad(x, x.shape[0], x.shape[1], operand_dtype)

        def make_w():
            out = torch.empty((3 * N, K), dtype=x.dtype, device=x.device)
            for i, w in enumerate((wq, wk, wv)):
                be.convert_pad(w.detach(), N, K, x.dtype, out=out[i * N:(i + 1) * N])
            return out
        # (the stamp of the cache entry is wq's; the fused optimiser bumps the weight epoch, load_state_dict touches all three)
        wsh = shadow(wq, ("qkv", id(wk), id(wv)), x.dtype, make_w, recipe=[(w, i * N, N, K, MAP_PLAIN, 0, False) for i, w in enumerate((wq, wk, wv))],
                     deps=(wk, wv))
        bias = shadow(bq, ("qkv_bias", id(bk), id(bv)), torch.float32, lambda: torch.cat([bq.detach(), bk.detach(), bv.detach()]).float(),
                      deps=(bk, bv))
        qkv = be.gemm(x, wsh, bias=bias)
        q, k, v = qkv[:, :N], qkv[:, N:2 * N], qkv[:, 2 * N:]
        vt = be.head_transpose(v, nseq, H, L, D)
        o, lse = be.attn_fwd(q, k, vt, None, keymask, nseq, H, L, D, scale, dropout=dropout)
        wt = None
        if x.dtype == torch.bfloat16 and x.shape[0] >= 256 and N % 64 == 0 and K % 8 == 0 and ctx.needs_input_grad[0]:
            # (K, 3 N) = the stacked weight transposed: the grad-input GEMM dx = [dq | dk | dv] W then has both operands k-contiguous (gemm_sm.hip)
            def make_wt():
                out = torch.empty((K, 3 * N), dtype=x.dtype, device=x.device)
                for i, w in enumerate((wq, wk, wv)):
                    out[:, i * N:(i + 1) * N].copy_(be.transpose2d(be.convert_pad(w.detach(), N, K, x.dtype)))
                return out
            wt = shadow(wq, ("qkvT", id(wk), id(wv)), x.dtype, make_wt,
                        recipe=[(w, 0, K, N, MAP_PLAIN, 0, True, i * N) for i, w in enumerate((wq, wk, wv))], deps=(wk, wv))
        ctx.wt = wt
        ctx.save_for_backward(x, wsh, qkv, o, lse, keymask if keymask is not None else x.new_empty(0))
        ctx.params = (wq, wk, wv, bq, bk, bv)
        ctx.dims = (nseq, L, H, D, scale, keymask is not None, dropout, N, K)
        if passthrough:
            return o, x_in.view_as(x_in)
        return o

    @staticmethod
    def backward(ctx, do, d_pass=None):
        be = B()
        x, wsh, qkv, o, lse, keymask = ctx.saved_tensors
        d_pass = d_pass.contiguous() if (ctx.passthrough and d_pass is not None) else None
        nseq, L, H, D, scale, has_mask, dropout, N, K = ctx.dims
        wq, wk, wv, bq, bk, bv = ctx.params
        do = do.contiguous()
        q, k, v = qkv[:, :N], qkv[:, N:2 * N], qkv[:, 2 * N:]
        qt, kt, dot = (be.head_transpose(t, nseq, H, L, D) for t in (q, k, do))
        dqkv = torch.empty_like(qkv)
        be.attn_bwd(q, k, v, qt, kt, o, do, dot, lse, None, keymask if has_mask else None, dqkv[:, :N], dqkv[:, N:2 * N], dqkv[:, 2 * N:], None,
                    nseq, H, L, D, scale, dropout=dropout)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (be.gemm(dqkv, ctx.wt, residual=d_pass, out_dtype=ctx.x_dtype) if ctx.wt is not None
                  else be.gemm(dqkv, wsh, a_kc=True, b_kc=False, residual=d_pass, out_dtype=ctx.x_dtype))
        gw, gb = [], []
        for i, (w, b) in enumerate(zip((wq, wk, wv), (bq, bk, bv))):
            dyi = dqkv[:, i * N:(i + 1) * N]
            fused = weight_bias_grad(dyi, x, w, b, K) if (w.requires_grad and b.requires_grad) else None      # dW and db in one launch
            if fused is not None:
                gw.append(fused[0]); gb.append(fused[1])
                continue
            gw.append(weight_grad(dqkv, x, w, [(0, N, i * N)], K) if w.requires_grad else None)
            gb.append(vec_grad(b, lambda dst, i=i: be.colsum(dqkv[:, i * N:(i + 1) * N], dst, N=N)) if b.requires_grad else None)
        return (dx, *gw, *gb, None, None, None, None, None, None, None, None, None)


class DropoutAddFn(Function):
    """y = dropout(x) (+ residual) -- nn.Dropout(hidden_dropout_prob) of HF BertEmbeddings / BertSelfOutput / BertOutput and the
    residual add that follows it.  The mask is regenerated from (seed, stream_id) in backward."""

    @staticmethod
    def forward(ctx, x, residual, p, seed, stream_id):
        ctx.args = (p, seed, stream_id, residual is not None)
        return B().dropout(x.contiguous(), residual.contiguous() if residual is not None else None, p, seed, stream_id)

    @staticmethod
    def backward(ctx, dy):
        p, seed, stream_id, has_res = ctx.args
        dy = dy.contiguous()
        return B().dropout(dy, None, p, seed, stream_id), (dy if has_res else None), None, None, None


# ------------------------------------------------------------------------------------------ small ops

class GegluFn(Function):
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return B().geglu_fwd(u)

    @staticmethod
    def backward(ctx, dg):
        (u,) = ctx.saved_tensors
        return B().geglu_bwd(dg.contiguous(), u)


class GeluFn(Function):
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return B().gelu_fwd(u)

    @staticmethod
    def backward(ctx, dh):
        (u,) = ctx.saved_tensors
        return B().gelu_bwd(dh.contiguous(), u)


class LeakyFn(Function):
    @staticmethod
    def forward(ctx, x, slope):
        ctx.save_for_backward(x)
        ctx.slope = slope
        return B().leaky_relu_fwd(x, slope)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return B().leaky_relu_bwd(dy.contiguous(), x, ctx.slope), None


class Permute0213Fn(Function):
    @staticmethod
    def forward(ctx, x):
        return B().permute0213(x)

    @staticmethod
    def backward(ctx, dy):
        return B().permute0213(dy.contiguous())


class PoolFn(Function):
    """ct_clip.py:724: mean over the depth axis of the (B, t, h*w*d) token grid.  out_dtype (mixed-precision head: f32 from bf16 tokens):
    the pooled vector and everything behind it -- to_visual_latent, l2norm, logits -- are f32; backward hands the tokens' dtype back."""

    @staticmethod
    def forward(ctx, x, out_dtype=None):
        ctx.t, ctx.in_dtype = x.shape[1], x.dtype
        return B().pool_fwd(x, out_dtype)

    @staticmethod
    def backward(ctx, dy):
        return B().pool_bwd(dy.contiguous(), ctx.t, ctx.in_dtype), None


class CpbExpandFn(Function):
    @staticmethod
    def forward(ctx, tab, gh, gw):
        ctx.g = (gh, gw)
        return B().cpb_expand(tab.contiguous(), gh, gw)

    @staticmethod
    def backward(ctx, dbias):
        return B().cpb_reduce(dbias.contiguous(), *ctx.g), None, None


class VqFn(Function):
    """vector_quantize_pytorch 1.1.2 cosine codebook (ctvit.py:403): argmax of cosine similarity, gather, straight-through,
    EMA buffer update in training mode.

    The package forces `x.float()` and searches in f32 whatever the autocast state.  In bf16 mode the tokens are l2-normalised in
    f32 and the search GEMM runs on the bf16 matrix cores over the three-term expansion hi.hi' + hi.lo' + lo.hi' (K = 3 d,
    f32 accumulate) -- or, on the big token grids (round 6), the raw bf16 tokens against the codebook's (hi, lo) pair (K = 2 d: the arg-max does
    not see a row's norm) --: f32-grade distances (error ~1e-7 against a median top-1 / top-2 margin of 8e-3), so
    code choices differ from the f32 reference only where the bf16 TOKENS themselves differ.  The EMA statistics are summed in
    f32 from x * inv in row order (deterministic).  `forced_idx` (test hook: teacher forcing) bypasses the search."""

    @staticmethod
    def forward(ctx, x, embed, cluster_size, training, decay, forced_idx=None):
        be = B()
        inv = None
        pre = getattr(getattr(VqFn, "stat_sync", None), "before_forward", None)
        if pre is not None:
            pre(embed)                      # a deferred EMA update of THIS codebook (distributed.VqStatSync) is applied before the codebook is read
        if forced_idx is not None:
            idx = forced_idx.reshape(-1).to(device=x.device, dtype=torch.int64).contiguous()
        elif x.dtype == torch.float32:
            xn, inv = be.l2norm_rows(x, torch.float32)
            en, _ = be.l2norm_rows(embed, torch.float32)
            idx, _ = be.gemm_argmax(xn, en)
        elif be.gemm_argmax_hilo_ok(x, embed.shape[0]) and os.environ.get("CTCLIP_VQ_HILO", "1") != "0":
            # arg-max_c x^ . e_c = arg-max_c x . e_c (|x| is a positive factor) and a bf16 token has no low part: the RAW tokens against the
            # (hi, lo) expansion of the unit codebook are two products per (token, code, dim) -- K = 2 d instead of the 3 d of the form below,
            # no expanded copy of the tokens, and nothing dropped but e_lo's own rounding (~2^-17)
            es, _ = be.l2norm_split3(embed, 2)
            idx, _ = be.gemm_argmax_hilo(x, es)
            if training:
                inv = be.row_inv_norms(x)
        else:
            xs, inv = be.l2norm_split3(x, 0)
            es, _ = be.l2norm_split3(embed, 1)
            idx, _ = be.gemm_argmax(xs, es)
        q = be.vq_gather(embed, idx, x.dtype)   # raw (pre-update) codebook rows
        if training:
            if inv is None:
                _, inv = be.l2norm_rows(x, torch.float32)
            bins, esum = be.vq_ema(idx, x, inv, cluster_size, embed, decay)
            hook = getattr(VqFn, "stat_sync", None)
            # data-parallel: all-reduce(SUM) the statistics.  The hook either reduces them in place (returns falsy: the update follows here) or
            # takes them over (returns True: distributed.VqStatSync reduces them on the communication stream and applies the EMA update when
            # the step's collectives are joined -- nothing in this forward reads the updated codebook)
            if hook is None or not hook(bins, esum, cluster_size, embed, decay):
                be.vq_ema_update(cluster_size, embed, bins, esum, decay)
        ctx.training = training
        ctx.mark_non_differentiable(idx)
        return q, idx

    @staticmethod
    def backward(ctx, dq, _didx):
        return (dq if ctx.training else None), None, None, None, None, None


class VisualLatentFn(Function):
    @staticmethod
    def forward(ctx, x, weight, wsh):
        y = B().visual_latent_fwd(x, wsh)
        ctx.save_for_backward(x, wsh)
        ctx.weight = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wsh = ctx.saved_tensors
        w = ctx.weight
        sink = sink_of(w)
        dw = None
        if w.requires_grad:
            dw = sink if sink is not None else torch.empty(w.shape, dtype=torch.float32, device=dy.device)
        # the 604-MB gradient of the 151-M-parameter weight: the first write after the optimiser cleared it overwrites (no read of the zeros)
        accumulate = sink is not None and not take_fresh_grad(w)
        dx = B().visual_latent_bwd(dy.contiguous(), x, wsh, dw, accumulate=accumulate, want_dx=ctx.needs_input_grad[0])
        return dx, (None if sink is not None else dw), None


def visual_latent(x, weight):
    return VisualLatentFn.apply(x, weight, plain_shadow(weight, x.dtype))


class ClipLossFn(Function):
    """ct_clip.py:771,796,845-901: l2norm, logits * exp(temperature), symmetric InfoNCE.  Forward and backward in one launch."""

    @staticmethod
    def forward(ctx, tl, il, temperature, replicas=1):
        """replicas: number of data-parallel ranks that evaluate this SAME (gathered) loss.  The latent gradients reach the
        parameters through each rank's local slice only, but the temperature gradient is complete on every rank: it is
        divided by `replicas` so that the all-reduce(SUM) of parameter gradients yields it once."""
        out, _, dtl, dil, dtemp = B().clip_loss(tl.contiguous(), il.contiguous(), temperature.detach().reshape(1))
        ctx.save_for_backward(dtl, dil, dtemp)
        ctx.temperature = temperature
        ctx.replicas = replicas
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        dtl, dil, dtemp = (t.clone() for t in ctx.saved_tensors)      # (not in place: backward may run again with retain_graph)
        be = B()
        s = dloss.reshape(1).to(torch.float32).contiguous()
        be.scale_by_scalar(dtl, s)
        be.scale_by_scalar(dil, s)
        be.scale_by_scalar(dtemp, s / ctx.replicas if ctx.replicas != 1 else s)
        t = ctx.temperature
        sink = sink_of(t)
        if sink is not None:
            sink.add_(dtemp.view_as(sink))
            return dtl, dil, None, None
        return dtl, dil, dtemp.view_as(t), None


class BertEmbedFn(Function):
    @staticmethod
    def forward(ctx, ids, word, pos, typ, dtype):
        x = B().bert_embed_fwd(ids, word.detach(), pos.detach(), typ.detach()[0].contiguous(), dtype)
        ctx.save_for_backward(ids)
        ctx.params = (word, pos, typ)
        return x

    @staticmethod
    def backward(ctx, dx):
        (ids,) = ctx.saved_tensors
        word, pos, typ = ctx.params
        ws, ps, ts = sink_of(word), sink_of(pos), sink_of(typ)
        dw = ws if ws is not None else torch.zeros_like(word)
        dp = ps if ps is not None else torch.zeros_like(pos)
        dt = ts if ts is not None else torch.zeros_like(typ)
        B().bert_embed_bwd(ids, dx.contiguous(), dw, dp, dt)   # dt row 0 only (token_type_ids are all zero)
        notify_grad_ready(("bert_embeddings", id(word)))        # the last gradients of the text tower
        return None, (None if ws is not None else dw), (None if ps is not None else dp), (None if ts is not None else dt), None


class _L2NormF32Fn(Function):
    @staticmethod
    def forward(ctx, x):
        y, _ = B().l2norm_rows(x.contiguous(), torch.float32)
        return y

    @staticmethod
    def backward(ctx, dy):
        raise NotImplementedError("return_latents=True is an inference output here (ct_lipro_train.py runs it under no_grad); "
                                  "gradients flow through the similarity / contrastive-loss kernels instead")


def l2norm_f32(x):
    return _L2NormF32Fn.apply(x)


class LatentSimilarityFn(Function):
    """ct_clip.py:771,796,805-807: l2norm both latents, pairwise dot with broadcasting, * exp(temperature)."""

    @staticmethod
    def forward(ctx, text_lat, image_lat, temperature):
        tl, il = text_lat.contiguous(), image_lat.contiguous()
        ctx.save_for_backward(tl, il)
        ctx.temperature = temperature
        return B().latent_similarity(tl, il, temperature.detach().reshape(1))

    @staticmethod
    def backward(ctx, dsims):
        tl, il = ctx.saved_tensors
        t = ctx.temperature
        dt, di, dtemp = B().latent_similarity(tl, il, t.detach().reshape(1), dsims.contiguous())
        sink = sink_of(t)
        if sink is not None:
            sink.add_(dtemp.view_as(sink))
            return dt, di, None
        return dt, di, dtemp.view_as(t)
