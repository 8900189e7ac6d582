"""CTViT image tower (drop-in for ``transformer_maskgit.CTViT``, reference
transformer_maskgit/transformer_maskgit/ctvit.py:118-412 and attention.py) on hand-written gfx950 kernels.

Same constructor signature, same ``forward(video, return_encoded_tokens=True)`` contract, same ``state_dict``
keys and -- because sub-modules are created in the reference's order with the same torch
initialisers -- the same random initialisation for a given seed.  The modules below are parameter containers;
all arithmetic goes through ``ct_clip_amd.functional`` (HIP kernels via the C-ABI library).
"""
import os
from pathlib import Path

import torch
from torch import nn

from . import functional as Fn


def default_compute_dtype():
    v = os.environ.get("CTCLIP_COMPUTE_DTYPE", "bf16").lower()
    return torch.float32 if v in ("fp32", "f32", "float32") else torch.bfloat16


def pair(val):
    ret = (val, val) if not isinstance(val, tuple) else val
    assert len(ret) == 2
    return ret


class LayerNorm(nn.Module):
    """attention.py:28-35: learnable gamma, beta is a zero buffer."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))


class GEGLU(nn.Module):
    pass


def FeedForward(dim, mult=4, dropout=0.0):
    """attention.py:44-52 (container only)."""
    inner_dim = int(mult * (2 / 3) * dim)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner_dim * 2, bias=False), GEGLU(), nn.Dropout(dropout),
                         nn.Linear(inner_dim, dim, bias=False))


class PEG(nn.Module):
    """attention.py:56-84 (container only)."""

    def __init__(self, dim, causal=False):
        super().__init__()
        assert causal, "CTViT builds PEG with peg_causal=True (ctvit.py:184); the kernel implements that padding"
        self.causal = causal
        self.dsconv = nn.Conv3d(dim, dim, 3, groups=dim)


class Attention(nn.Module):
    """attention.py:88-125 (container only; self-attention, no null kv, non-causal)."""

    def __init__(self, dim, dim_context=None, dim_head=64, heads=8, causal=False, num_null_kv=0, norm_context=True,
                 dropout=0.0, scale=8):
        super().__init__()
        assert not causal and num_null_kv == 0 and dropout == 0.0
        self.heads, self.dim_head, self.scale = heads, dim_head, scale
        inner_dim = dim_head * heads
        dim_context = dim_context or dim
        self.norm = LayerNorm(dim)
        self.context_norm = LayerNorm(dim_context) if norm_context else nn.Identity()
        self.num_null_kv = num_null_kv
        self.null_kv = nn.Parameter(torch.randn(heads, 2 * num_null_kv, dim_head))
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim_context, inner_dim * 2, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.to_out = nn.Linear(inner_dim, dim, bias=False)


class ContinuousPositionBias(nn.Module):
    """attention.py:229-276.  The MLP is evaluated on the (2h-1)(2w-1) DISTINCT offsets and gathered to (heads, hw, hw)."""

    def __init__(self, *, dim, heads, num_dims=2, layers=2, log_dist=True, cache_rel_pos=False):
        super().__init__()
        assert num_dims == 2 and log_dist
        self.net = nn.ModuleList([])
        self.net.append(nn.Sequential(nn.Linear(num_dims, dim), nn.LeakyReLU(0.1)))
        for _ in range(layers - 1):
            self.net.append(nn.Sequential(nn.Linear(dim, dim), nn.LeakyReLU(0.1)))
        self.net.append(nn.Linear(dim, heads))
        self.cache_rel_pos = cache_rel_pos
        self.register_buffer("rel_pos", None, persistent=False)
        self._tables = {}

    def offsets(self, gh, gw, device):
        key = (gh, gw, str(device))
        t = self._tables.get(key)
        if t is None:
            dy = torch.arange(-(gh - 1), gh, dtype=torch.float32)
            dx = torch.arange(-(gw - 1), gw, dtype=torch.float32)
            rel = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), dim=-1).reshape(-1, 2)
            rel = torch.sign(rel) * torch.log(rel.abs() + 1)
            t = torch.zeros(rel.shape[0], 8)
            t[:, :2] = rel
            t = t.to(device)
            self._tables[key] = t
        return t

    def forward(self, gh, gw, device=None):
        device = self.net[0][0].weight.device
        x = self.offsets(gh, gw, device)
        n = len(self.net)
        for i, layer in enumerate(self.net):
            lin = layer[0] if i < n - 1 else layer
            x = Fn.linear(x, lin.weight, lin.bias, kpad=8 if i == 0 else None)
            if i < n - 1:
                x = Fn.LeakyFn.apply(x, 0.1)
        # the attention kernels gather from this (nclass, heads) table; Fn.CpbExpandFn materialises the reference's (heads, hw, hw)
        return x


class Transformer(nn.Module):
    """attention.py:280-333."""

    def __init__(self, dim, *, depth, dim_context=None, causal=False, dim_head=64, heads=8, ff_mult=4, peg=False,
                 peg_causal=False, attn_num_null_kv=2, has_cross_attn=False, attn_dropout=0.0, ff_dropout=0.0):
        super().__init__()
        assert not has_cross_attn and not causal and peg
        if attn_dropout or ff_dropout:
            raise NotImplementedError("attention / feed-forward dropout is 0 everywhere in CT-CLIP (run_train.py:17-27)")
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PEG(dim=dim, causal=peg_causal),
                Attention(dim=dim, dim_head=dim_head, heads=heads, causal=causal, dropout=attn_dropout),
                None,
                FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout),
            ]))
        self.norm_out = LayerNorm(dim)

    def forward(self, x, video_shape, nseq, L, attn_bias=None, bias_grid=None):
        """x: (nseq*L, dim) activation in compute dtype (flat view of the reference's (nseq, L, dim))."""
        b, t, h, w = video_shape
        d = x.shape[1]
        tap = self.__dict__.get("layer_tap")      # test hook (never set by the product): tap(i, x) sees the residual stream at every layer boundary
        # bf16 inference (or CTCLIP_RESIDUAL_COMP=1): the residual stream is the bf16 pair (x, e) -- x what the consumers read, e the rounding
        # residue of the last add, added back in f32 by the next one (functional.residual_comp_enabled: the 72 roundings of a 24-layer stream
        # stop accumulating)
        e = None
        kind, nl = self.__dict__.get("kind"), len(self.layers)
        for i, layer in enumerate(self.layers):
            peg, attn, _, ff = layer
            comp = Fn.residual_comp_enabled(x, kind, i, nl)      # (may switch ON at a later layer -- CTCLIP_RESIDUAL_COMP=lastN: the pair then starts with e = None)
            if tap is not None:
                tap(i, x if e is None else x.float() + e.float())
            x = Fn.grad_ready(x, layer)     # backward passing this point = the layer's parameter gradients are final
            # x = peg(x) + x  -- PEG sees the buffer flat-reinterpreted as (b, t, h, w, d) (attention.py:69-70)
            if comp:
                x, e = Fn.peg_residual(x.view(b, t, h, w, d), peg.dsconv.weight, peg.dsconv.bias, comp=False if e is None else e.view(b, t, h, w, d))
                x, e = x.view(-1, d), e.view(-1, d)
            else:
                x = Fn.peg_residual(x.view(b, t, h, w, d), peg.dsconv.weight, peg.dsconv.bias).view(-1, d)
            # x = attn(x) + x  -- q from LayerNorm(x), k/v from the RAW x (attention.py:139-143)
            xn, x_kv, x = Fn.layer_norm_branch(x, attn.norm.gamma, None, 2)   # the three consumers of x: LayerNorm, to_kv, residual
            o = Fn.qkv_attention(xn, x_kv, attn.to_q.weight, attn.to_kv.weight, attn.q_scale, attn.k_scale, attn_bias, nseq, L, attn.heads,
                                 attn.dim_head, float(attn.scale), bias_grid)
            if comp:
                x, e = Fn.linear(o, attn.to_out.weight, residual=x, comp=e)
            else:
                x = Fn.linear(o, attn.to_out.weight, residual=x)
            # x = ff(x) + x
            y, x = Fn.layer_norm_branch(x, ff[0].weight, ff[0].bias, 1)
            if comp:
                x, e = Fn.feed_forward(y, ff[1].weight, ff[4].weight, residual=x, comp=e)
            else:
                x = Fn.feed_forward(y, ff[1].weight, ff[4].weight, residual=x)
        # norm_out reads x alone: the last residue e (|e| <= half a bf16 ulp of x) is dropped, i.e. the stream takes ONE plain bf16 rounding here
        # instead of 72 along the way.  The compensated stream is an inference-time refinement (auto = only without autograd): a training forward
        # and a no_grad forward of the same bf16 weights differ by that rounding noise (INTEGRATION.md CTCLIP_RESIDUAL_COMP).
        return Fn.layer_norm(x, self.norm_out.gamma, None)


class CosineSimCodebook(nn.Module):
    def __init__(self, dim, codebook_size, decay=0.8, eps=1e-5):
        super().__init__()
        self.decay, self.eps, self.codebook_size = decay, eps, codebook_size
        embed = torch.empty(1, codebook_size, dim)
        nn.init.kaiming_uniform_(embed)
        embed = torch.nn.functional.normalize(embed, dim=-1)
        self.register_buffer("initted", torch.Tensor([1.0]))
        self.register_buffer("cluster_size", torch.zeros(1, codebook_size))
        self.register_buffer("embed", embed)


class VectorQuantize(nn.Module):
    """State/keys of vector_quantize_pytorch.VectorQuantize(dim, codebook_size, use_cosine_sim=True) (ctvit.py:188)."""

    def __init__(self, dim, codebook_size, use_cosine_sim=True, decay=0.8):
        super().__init__()
        assert use_cosine_sim
        self.codebook_size = codebook_size
        self._codebook = CosineSimCodebook(dim, codebook_size, decay=decay)

    @property
    def codebook(self):
        return self._codebook.embed[0]

    def forward(self, x2d):
        cb = self._codebook
        # `teacher_indices` (test hook, never set by the product): quantise with the given code ids instead of searching --
        # SURVEY.md Appendix D protocol 2a, separates kernel error from code flips in bf16 parity tests
        forced = getattr(self, "teacher_indices", None)
        q, idx = Fn.VqFn.apply(x2d, cb.embed[0], cb.cluster_size[0], self.training, cb.decay, forced)
        return q, idx


class CTViT(nn.Module):
    def __init__(self, *, dim, codebook_size, image_size, patch_size, temporal_patch_size, spatial_depth, temporal_depth,
                 discr_base_dim=16, dim_head=64, heads=8, channels=1, use_vgg_and_gan=True, vgg=None,
                 discr_attn_res_layers=(16,), use_hinge_loss=True, attn_dropout=0.0, ff_dropout=0.0, compute_dtype=None):
        super().__init__()
        if channels != 1:
            raise NotImplementedError("CT volumes are single-channel (scripts/data.py:104-162)")
        self.image_size = pair(image_size)
        self.patch_size = pair(patch_size)
        patch_height, patch_width = self.patch_size
        self.temporal_patch_size = temporal_patch_size
        self.compute_dtype = compute_dtype or default_compute_dtype()

        self.spatial_rel_pos_bias = ContinuousPositionBias(dim=dim, heads=heads)
        image_height, image_width = self.image_size
        assert (image_height % patch_height) == 0 and (image_width % patch_width) == 0

        self.to_patch_emb_first_frame = nn.Sequential(
            nn.Identity(), nn.LayerNorm(channels * patch_width * patch_height),
            nn.Linear(channels * patch_width * patch_height, dim), nn.LayerNorm(dim))
        self.to_patch_emb = nn.Sequential(
            nn.Identity(), nn.LayerNorm(channels * patch_width * patch_height * temporal_patch_size),
            nn.Linear(channels * patch_width * patch_height * temporal_patch_size, dim), nn.LayerNorm(dim))

        kw = dict(dim=dim, dim_head=dim_head, heads=heads, attn_dropout=attn_dropout, ff_dropout=ff_dropout, peg=True,
                  peg_causal=True)
        self.enc_spatial_transformer = Transformer(depth=spatial_depth, **kw)
        self.enc_temporal_transformer = Transformer(depth=temporal_depth, **kw)
        self.enc_spatial_transformer.__dict__["kind"], self.enc_temporal_transformer.__dict__["kind"] = "spatial", "temporal"
        self.vq = VectorQuantize(dim=dim, codebook_size=codebook_size, use_cosine_sim=True)

        self.to_pixels_first_frame = nn.Sequential(nn.Linear(dim, channels * patch_width * patch_height), nn.Identity())
        self.to_pixels = nn.Sequential(nn.Linear(dim, channels * patch_width * patch_height * temporal_patch_size),
                                       nn.Identity())
        self.dim = dim

    @property
    def patch_height_width(self):
        return self.image_size[0] // self.patch_size[0], self.image_size[1] // self.patch_size[1]

    @property
    def image_num_tokens(self):
        h, w = self.patch_height_width
        return h * w

    def load(self, path):
        path = Path(path)
        assert path.exists()
        self.load_state_dict(torch.load(str(path)))

    # -- ctvit.py:282-307
    def encode(self, tokens, b, t, h, w):
        d = tokens.shape[1]
        video_shape = (b, t, h, w)
        attn_bias = self.spatial_rel_pos_bias(h, w)
        x = self.enc_spatial_transformer(tokens, video_shape, nseq=b * t, L=h * w, attn_bias=attn_bias, bias_grid=(h, w))
        x = Fn.Permute0213Fn.apply(x.view(b, t, h * w, d)).view(-1, d)              # (b t)(h w) -> (b h w) t
        x = self.enc_temporal_transformer(x, video_shape, nseq=b * h * w, L=t, attn_bias=None)
        x = Fn.Permute0213Fn.apply(x.view(b, h * w, t, d)).view(-1, d)              # back to b t h w
        return x

    def tokens_before_vq(self, video):
        """Patch embedding + spatial / temporal transformers (ctvit.py:385-395): -> ((b*t*h*w, dim) tokens in the compute dtype, (b, t, h, w))."""
        assert video.ndim == 5, "expected (b, c, frames, H, W)"
        b, c, f, *image_dims = video.shape
        assert tuple(image_dims) == self.image_size
        assert f % self.temporal_patch_size == 0
        pt, (p1, p2) = self.temporal_patch_size, self.patch_size
        t, h, w = f // pt, image_dims[0] // p1, image_dims[1] // p2
        e = self.to_patch_emb
        video = video.to(device=e[2].weight.device, dtype=torch.float32).contiguous()
        tokens = Fn.PatchEmbedFn.apply(video, e[1].weight, e[1].bias, e[2].weight, e[2].bias, e[3].weight, e[3].bias,
                                       pt, p1, p2, self.compute_dtype)
        return self.encode(tokens, b, t, h, w), (b, t, h, w)

    def forward(self, video, mask=None, return_recons=False, return_recons_only=False, return_discr_loss=False,
                apply_grad_penalty=True, return_only_codebook_ids=False, return_encoded_tokens=False):
        if mask is not None:
            raise NotImplementedError("frame masks are never passed on the CT-CLIP path (ct_clip.py:715)")
        tokens, (b, t, h, w) = self.tokens_before_vq(video)
        q, idx = self.vq(tokens)
        if return_only_codebook_ids:
            return idx.view(b, t, h, w)
        tokens = q.view(b, t, h, w, -1)
        if return_encoded_tokens:
            return tokens
        raise NotImplementedError("the reconstruction / GAN branch of CTViT is dead code in CT-CLIP (ctvit.py:414-525 "
                                  "references modules that are never created); only return_encoded_tokens=True is reachable")
