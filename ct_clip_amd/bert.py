"""Text tower: forward/backward of a HuggingFace ``BertModel`` on the gfx950 kernels.

The reference calls ``self.text_transformer(input_ids, attention_mask=...)[0]`` (CT_CLIP/ct_clip/ct_clip.py:685-686) on a
``transformers.BertModel`` (scripts/run_train.py:9).  Here the HF module is kept as the *parameter container* (so
``state_dict`` keys ``text_transformer.*`` are unchanged) and its arithmetic -- embeddings gather, post-LN encoder layers
with biased dense projections, softmax attention with key-padding mask, erf-GELU FFN (HF modeling_bert.py: BertEmbeddings,
BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput) -- runs through ``ct_clip_amd.functional``.
The pooler is skipped: CT-CLIP discards it (ct_clip.py:686,762).
"""
import math

import torch

from . import functional as Fn

def is_hf_bert(module):
    return hasattr(module, "embeddings") and hasattr(module, "encoder") and hasattr(module.encoder, "layer") \
        and hasattr(module.embeddings, "word_embeddings")


def bert_last_hidden_state(bert, input_ids, attention_mask, dtype, operand_dtype=None, cls_only=False):
    """Returns last_hidden_state as a (B*T, hidden) activation in ``dtype`` -- or, with cls_only, its rows of the [CLS] positions, (B, hidden).

    cls_only: CT-CLIP reads enc_text[:, 0, :] and nothing else (ct_clip.py:762), so in the LAST layer only the [CLS] rows of the attention output
    have a consumer: the attention output projection, both LayerNorms, the feed-forward block, their dropouts and all their backward run on B rows
    instead of B*T (9 of the 12 d^2 multiply-adds per token of that layer; the q | k | v projection and the attention core still see every
    token: the keys and values of all positions feed the [CLS] query).  Same numbers for every consumer of the [CLS] rows, same parameter gradients.

    operand_dtype (mixed precision, the default of the bf16 mode: dtype = f32, operand_dtype = bf16): the residual stream, both
    LayerNorms of a layer, bias / dropout / residual adds, GELU and every GEMM's accumulate-and-store are f32 -- what torch.autocast
    keeps in f32 around an HF BertModel -- and bf16 appears only where the matrix cores read it: the GEMM operands (each f32 activation
    rounded once on its way into a GEMM), q | k | v and the attention core.  M = B*T rows x 768: the f32 tensors are a few MB."""
    od = operand_dtype if (operand_dtype is not None and operand_dtype != dtype) else None
    cfg = bert.config
    if getattr(cfg, "hidden_act", "gelu") != "gelu":
        raise NotImplementedError(f"hidden_act={cfg.hidden_act!r}: only erf-GELU BERT is implemented")
    if getattr(cfg, "position_embedding_type", "absolute") != "absolute":
        raise NotImplementedError("only absolute position embeddings are implemented")
    # train-mode dropout (HF modeling_bert.py: BertEmbeddings, BertSelfAttention, BertSelfOutput, BertOutput).  Masks are
    # counter-based (Philox, ct_clip_amd/csrc/common.h): one 62-bit seed per forward drawn from torch's CPU generator (so that
    # torch.manual_seed reproduces a run), call sites separated by stream ids / seed offsets, regenerated in backward.
    p_hid = float(cfg.hidden_dropout_prob) if bert.training else 0.0
    p_att = float(cfg.attention_probs_dropout_prob) if bert.training else 0.0
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p_hid > 0 or p_att > 0) else 0

    def drop(h, residual, site):   # only reached with p_hid > 0, except for the embeddings (identity then)
        return Fn.DropoutAddFn.apply(h, residual, p_hid, seed, site) if p_hid > 0 else h
    emb = bert.embeddings
    dev = emb.word_embeddings.weight.device
    ids = input_ids.to(dev).contiguous()
    Bsz, T = ids.shape
    nh = cfg.num_attention_heads
    hidden = cfg.hidden_size
    dh = hidden // nh
    eps = cfg.layer_norm_eps

    x = Fn.BertEmbedFn.apply(ids, emb.word_embeddings.weight, emb.position_embeddings.weight,
                             emb.token_type_embeddings.weight, dtype)
    x = Fn.layer_norm(x, emb.LayerNorm.weight, emb.LayerNorm.bias, eps)
    x = drop(x, None, 0)
    keymask = None
    if attention_mask is not None:
        # additive key mask, as HF builds it: (1 - mask) * finfo.min
        keymask = ((1.0 - attention_mask.to(device=dev, dtype=torch.float32)) * torch.finfo(torch.float32).min).contiguous()
    scale = 1.0 / math.sqrt(dh)
    nlayers = len(bert.encoder.layer)
    for li, layer in enumerate(bert.encoder.layer):
        x = Fn.grad_ready(x, layer)
        sa, so = layer.attention.self, layer.attention.output
        # (x feeds the attention AND the residual connection: the node hands back a view of x for the latter, so that in backward the residual
        # path's gradient is added inside the grad-input GEMM's epilogue instead of by an elementwise accumulation kernel)
        c, x_res = Fn.QkvSdpaFn.apply(x, sa.query.weight, sa.key.weight, sa.value.weight, sa.query.bias, sa.key.bias, sa.value.bias, keymask,
                                      Bsz, T, nh, dh, scale, (p_att, seed + 1 + li) if p_att > 0 else None, od, True)
        if cls_only and li == nlayers - 1 and T > 1:
            # rows b * T of the attention output and of the layer input: everything behind them in this layer is row-wise
            c = c.view(Bsz, T * c.shape[1])[:, :c.shape[1]].contiguous()
            x_res = x_res.view(Bsz, T * hidden)[:, :hidden].contiguous()
        if p_hid > 0:    # dense -> dropout -> + input -> LayerNorm
            h1 = drop(Fn.linear(c, so.dense.weight, so.dense.bias, out_dtype=dtype, operand_dtype=od), x_res, 1 + 2 * li)
        else:
            h1 = Fn.linear(c, so.dense.weight, so.dense.bias, residual=x_res, out_dtype=dtype, operand_dtype=od)
        x = Fn.layer_norm(h1, so.LayerNorm.weight, so.LayerNorm.bias, eps)
        u, x_res = Fn.linear(x, layer.intermediate.dense.weight, layer.intermediate.dense.bias, operand_dtype=od, passthrough=True)
        m = Fn.GeluFn.apply(u)
        if p_hid > 0:
            h2 = drop(Fn.linear(m, layer.output.dense.weight, layer.output.dense.bias, operand_dtype=od), x_res, 2 + 2 * li)
        else:
            h2 = Fn.linear(m, layer.output.dense.weight, layer.output.dense.bias, residual=x_res, operand_dtype=od)
        x = Fn.layer_norm(h2, layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, eps)
    return x
