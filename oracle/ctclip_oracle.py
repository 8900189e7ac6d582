"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32) restatement of the CT-CLIP training hot path.

This file is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product (``ct_clip_amd``) never does; it fails loudly
when its HIP extension is missing.

Each function restates one row of SURVEY.md section 8(a) and cites the reference lines it follows
(paths relative to /root/reference).  It is pinned against outputs of the reference itself:
``oracle/gen_golden.py`` shim-imports the real reference modules in the build container, runs them
on seeded inputs and commits the results under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this restatement against those fixtures (loss, latents, every intermediate, gradients, VQ
buffers).  The reference itself ships no tests/golden vectors, and the
vector-quantiser is an un-vendored third-party package restated from its published algorithm
(``oracle/vq_restatement.py``) -- that part is PARITY UNPINNED and is declared so in DESIGN.md.

All tensors are taken from a state dict that uses the reference's ``CTCLIP.state_dict()`` key
names, so the same dict drives the reference, this oracle and the product.
"""
from dataclasses import dataclass
import math

import torch
import torch.nn.functional as F


@dataclass
class OracleConfig:
    # CTViT (transformer_maskgit/transformer_maskgit/ctvit.py:119-139)
    dim: int = 512
    codebook_size: int = 8192
    image_size: int = 480
    patch_size: int = 20
    temporal_patch_size: int = 10
    spatial_depth: int = 4
    temporal_depth: int = 4
    dim_head: int = 32
    heads: int = 8
    # BERT (HF BertConfig)
    bert_layers: int = 12
    bert_heads: int = 12
    bert_eps: float = 1e-12
    # CTCLIP (CT_CLIP/ct_clip/ct_clip.py:408-450)
    dim_latent: int = 512
    vq_decay: float = 0.8

    @property
    def ff_inner(self):  # attention.py:45
        return int(4 * (2 / 3) * self.dim)


def l2norm(t):  # attention.py:22-23, ct_clip.py:49-50
    return F.normalize(t, dim=-1)


# --------------------------------------------------------------------------- image tower

def patch_embed(sd, cfg, video, pre="visual_transformer.to_patch_emb."):
    """ctvit.py:170-175,385 -- Rearrange 'b c (t pt)(h p1)(w p2) -> b t h w (c pt p1 p2)',
    LayerNorm(4000), Linear(4000,512), LayerNorm(512)."""
    b, c, f, H, W = video.shape
    pt, p = cfg.temporal_patch_size, cfg.patch_size
    t, h, w = f // pt, H // p, W // p
    x = video.reshape(b, c, t, pt, h, p, w, p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, t, h, w, c * pt * p * p)
    x = F.layer_norm(x, x.shape[-1:], sd[pre + "1.weight"], sd[pre + "1.bias"], 1e-5)
    x = F.linear(x, sd[pre + "2.weight"], sd[pre + "2.bias"])
    x = F.layer_norm(x, x.shape[-1:], sd[pre + "3.weight"], sd[pre + "3.bias"], 1e-5)
    return x


def continuous_position_bias(sd, h, w, pre="visual_transformer.spatial_rel_pos_bias."):
    """attention.py:257-276 -- signed-log relative offsets -> MLP 2->dim->dim->heads, LeakyReLU(0.1)."""
    pos = [torch.arange(h), torch.arange(w)]
    grid = torch.stack(torch.meshgrid(*pos, indexing="ij")).reshape(2, -1).t()
    rel = (grid[:, None, :] - grid[None, :, :]).float()
    rel = torch.sign(rel) * torch.log(rel.abs() + 1)
    x = F.leaky_relu(F.linear(rel, sd[pre + "net.0.0.weight"], sd[pre + "net.0.0.bias"]), 0.1)
    x = F.leaky_relu(F.linear(x, sd[pre + "net.1.0.weight"], sd[pre + "net.1.0.bias"]), 0.1)
    x = F.linear(x, sd[pre + "net.2.weight"], sd[pre + "net.2.bias"])
    return x.permute(2, 0, 1)  # (heads, hw, hw)


def peg(sd, pre, x, shape):
    """attention.py:63-84 -- flat reshape to (b,D1,D2,D3,d), causal pad (1,1,1,1,2,0), depthwise conv3d."""
    orig = x.shape
    x = x.reshape(*shape, -1).permute(0, 4, 1, 2, 3)
    x = F.pad(x, (1, 1, 1, 1, 2, 0), value=0.0)
    x = F.conv3d(x, sd[pre + "dsconv.weight"], sd[pre + "dsconv.bias"], groups=x.shape[1])
    x = x.permute(0, 2, 3, 4, 1)
    return x.reshape(orig)


def attention(sd, pre, cfg, x, attn_bias=None):
    """attention.py:127-181 -- q from LayerNorm(x), k/v from RAW x, cosine-sim attention scale 8."""
    hds = cfg.heads
    xn = F.layer_norm(x, x.shape[-1:], sd[pre + "norm.gamma"], torch.zeros_like(sd[pre + "norm.gamma"]), 1e-5)
    q = F.linear(xn, sd[pre + "to_q.weight"])
    k, v = F.linear(x, sd[pre + "to_kv.weight"]).chunk(2, dim=-1)
    B, n, _ = q.shape
    q, k, v = (t.reshape(B, n, hds, -1).permute(0, 2, 1, 3) for t in (q, k, v))
    q, k = l2norm(q) * sd[pre + "q_scale"], l2norm(k) * sd[pre + "k_scale"]
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * 8.0
    if attn_bias is not None:
        sim = sim + attn_bias
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(B, n, -1)
    return F.linear(out, sd[pre + "to_out.weight"])


def feedforward(sd, pre, x):
    """attention.py:39-52 -- LayerNorm, Linear(d, 2*inner, no bias), GEGLU = gelu(gate)*x, Linear(inner, d)."""
    y = F.layer_norm(x, x.shape[-1:], sd[pre + "0.weight"], sd[pre + "0.bias"], 1e-5)
    y = F.linear(y, sd[pre + "1.weight"])
    a, gate = y.chunk(2, dim=-1)
    y = F.gelu(gate) * a
    return F.linear(y, sd[pre + "4.weight"])


def transformer(sd, pre, cfg, depth, x, video_shape, attn_bias=None, trace=None):
    """attention.py:312-333."""
    for l in range(depth):
        p = f"{pre}layers.{l}."
        x = peg(sd, p + "0.", x, video_shape) + x
        if trace is not None:
            trace[p + "peg"] = x
        x = attention(sd, p + "1.", cfg, x, attn_bias) + x
        if trace is not None:
            trace[p + "attn"] = x
        x = feedforward(sd, p + "3.", x) + x
        if trace is not None:
            trace[p + "ff"] = x
    g = sd[pre + "norm_out.gamma"]
    return F.layer_norm(x, x.shape[-1:], g, torch.zeros_like(g), 1e-5)


def ctvit_encode(sd, cfg, tokens, trace=None):
    """ctvit.py:282-307."""
    b, t, h, w, d = tokens.shape
    video_shape = (b, t, h, w)
    pre = "visual_transformer."
    x = tokens.reshape(b * t, h * w, d)
    bias = continuous_position_bias(sd, h, w)
    if trace is not None:
        trace["attn_bias"] = bias
    x = transformer(sd, pre + "enc_spatial_transformer.", cfg, cfg.spatial_depth, x, video_shape, bias, trace)
    x = x.reshape(b, t, h, w, d).permute(0, 2, 3, 1, 4).reshape(b * h * w, t, d)
    x = transformer(sd, pre + "enc_temporal_transformer.", cfg, cfg.temporal_depth, x, video_shape, None, trace)
    return x.reshape(b, h, w, t, d).permute(0, 3, 1, 2, 4)


def vq_forward(sd, cfg, x, training, pre="visual_transformer.vq._codebook."):
    """vector-quantize-pytorch 1.1.2 cosine codebook (see oracle/vq_restatement.py; PARITY UNPINNED).
    Returns (out, indices, new_buffers or None)."""
    shape = x.shape
    embed = sd[pre + "embed"][0]
    flatten = l2norm(x.float().reshape(-1, shape[-1]))
    embed_n = l2norm(embed)
    dist = flatten.detach() @ embed_n.t()
    ind = dist.argmax(dim=-1)
    quant = embed[ind].reshape(shape)
    new = None
    if training:
        with torch.no_grad():
            C = embed.shape[0]
            bins = torch.bincount(ind, minlength=C).float()
            cluster = sd[pre + "cluster_size"][0] * cfg.vq_decay + bins * (1 - cfg.vq_decay)
            esum = torch.zeros_like(embed).index_add_(0, ind, flatten.detach())
            enorm = l2norm(esum / bins.clamp(min=1.0).unsqueeze(-1))
            enorm = torch.where((bins == 0).unsqueeze(-1), embed_n, enorm)
            new = {"cluster_size": cluster[None], "embed": (embed * cfg.vq_decay + enorm * (1 - cfg.vq_decay))[None]}
        out = x + (quant - x).detach()
    else:
        out = quant
    return out, ind.reshape(shape[:-1]), new


def ctvit_forward(sd, cfg, video, training=True, trace=None):
    """ctvit.py:353-412 with return_encoded_tokens=True."""
    tokens = patch_embed(sd, cfg, video)
    if trace is not None:
        trace["patch_emb"] = tokens
    tokens = ctvit_encode(sd, cfg, tokens, trace)
    if trace is not None:
        trace["pre_vq"] = tokens
    b, t, h, w, d = tokens.shape
    q, ind, new = vq_forward(sd, cfg, tokens.reshape(b, t * h * w, d), training)
    if trace is not None:
        trace["vq_indices"] = ind
    return q.reshape(b, t, h, w, d), new


# --------------------------------------------------------------------------- text tower

def bert_forward(sd, cfg, input_ids, attention_mask, pre="text_transformer."):
    """HF transformers BertModel (post-LN BERT; modeling_bert.py embeddings/self-attn/output/intermediate),
    dropout 0.  Called at ct_clip.py:685-686; only last_hidden_state is used."""
    B, T = input_ids.shape
    e = pre + "embeddings."
    x = sd[e + "word_embeddings.weight"][input_ids] + sd[e + "position_embeddings.weight"][:T][None] \
        + sd[e + "token_type_embeddings.weight"][0][None, None]
    x = F.layer_norm(x, x.shape[-1:], sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg.bert_eps)
    nh = cfg.bert_heads
    dh = x.shape[-1] // nh
    ext = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for l in range(cfg.bert_layers):
        p = f"{pre}encoder.layer.{l}."
        q = F.linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
        k = F.linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
        v = F.linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
        q, k, v = (t_.reshape(B, T, nh, dh).permute(0, 2, 1, 3) for t_ in (q, k, v))
        s = torch.einsum("bhid,bhjd->bhij", q, k) / math.sqrt(dh) + ext
        a = s.softmax(dim=-1)
        c = torch.einsum("bhij,bhjd->bhid", a, v).permute(0, 2, 1, 3).reshape(B, T, nh * dh)
        c = F.linear(c, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        x = F.layer_norm(x + c, x.shape[-1:], sd[p + "attention.output.LayerNorm.weight"],
                         sd[p + "attention.output.LayerNorm.bias"], cfg.bert_eps)
        m = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        m = F.linear(m, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = F.layer_norm(x + m, x.shape[-1:], sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"],
                         cfg.bert_eps)
    return x


# --------------------------------------------------------------------------- CLIP head

def clip_loss_from_latents(text_latents, image_latents, temperature):
    """ct_clip.py:796,845-846,858-878,890-901 (exp / diag / sum / log(t+1e-20) form, symmetric mean)."""
    temp = temperature.exp()
    t2i = text_latents @ image_latents.t() * temp
    i2t = t2i.t()
    e1, e2 = t2i.exp(), i2t.exp()
    l1 = (-torch.log(e1.diagonal() + 1e-20) + torch.log(e1.sum(-1) + 1e-20)).mean()
    l2 = (-torch.log(e2.diagonal() + 1e-20) + torch.log(e2.sum(-1) + 1e-20)).mean()
    return (l1 + l2) / 2, t2i


def ctclip_forward(sd, cfg, input_ids, attention_mask, video, training=True, trace=None):
    """ct_clip.py:614-901 with return_loss=True.  Returns dict(loss, text_latents, image_latents,
    logits, enc_image_tokens, vq_new)."""
    enc_text = bert_forward(sd, cfg, input_ids, attention_mask)                   # ct_clip.py:685-686
    enc_tokens, vq_new = ctvit_forward(sd, cfg, video, training, trace)           # ct_clip.py:715
    enc_image = enc_tokens.mean(dim=1).reshape(enc_tokens.shape[0], -1)           # ct_clip.py:724,740
    text_lat = l2norm(F.linear(enc_text[:, 0, :], sd["to_text_latent.weight"]))   # ct_clip.py:762,765,771
    image_lat = l2norm(F.linear(enc_image, sd["to_visual_latent.weight"]))        # ct_clip.py:767,771
    loss, logits = clip_loss_from_latents(text_lat, image_lat, sd["temperature"])
    return dict(loss=loss, text_latents=text_lat, image_latents=image_lat, logits=logits,
                enc_text=enc_text, enc_image_tokens=enc_tokens, vq_new=vq_new)


def similarity_no_loss(text_lat, image_lat, temperature):
    """ct_clip.py:805-807 -- einsum('b d, b d -> b') * temp with broadcasting (2 prompts vs 1 volume)."""
    return (text_lat * image_lat).sum(-1) * temperature.exp()


def train_step_reference(sd, cfg, input_ids, attention_mask, video, lr=1.25e-6, max_grad_norm=0.5,
                         betas=(0.9, 0.99), eps=1e-8, grad_keys=None):
    """One optimisation step as scripts/CTCLIPTrainer.py:233-264 does it (fwd, bwd, clip_grad_norm_(0.5),
    Adam lr 1.25e-6 betas (0.9,0.99) -- optimizer.py:24), starting from zero Adam state.
    Returns (loss, grads dict, new params dict, total grad norm, vq_new)."""
    leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "vq._codebook" not in k
                                                   and not k.endswith(".beta") and "position_ids" not in k)
              for k, v in sd.items()}
    out = ctclip_forward(leaves, cfg, input_ids, attention_mask, video, training=True)
    out["loss"].backward()
    grads = {k: v.grad for k, v in leaves.items() if v.requires_grad and v.grad is not None}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    clip = torch.clamp(max_grad_norm / (total + 1e-6), max=1.0)
    new = {}
    for k, g in grads.items():
        g = g * clip
        m = (1 - betas[0]) * g
        v = (1 - betas[1]) * g * g
        mhat = m / (1 - betas[0])
        vhat = v / (1 - betas[1])
        new[k] = leaves[k].detach() - lr * mhat / (vhat.sqrt() + eps)
    return out["loss"].detach(), grads, new, total, out["vq_new"]
