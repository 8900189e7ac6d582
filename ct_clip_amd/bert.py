"""Text tower: forward/backward of a HuggingFace ``BertModel`` on the gfx950 kernels.

The reference calls ``self.text_transformer(input_ids, attention_mask=...)[0]`` (CT_CLIP/ct_clip/ct_clip.py:685-686) on a
``transformers.BertModel`` (scripts/run_train.py:9).  Here the HF module is kept as the *parameter container* (so
``state_dict`` keys ``text_transformer.*`` are unchanged) and its arithmetic -- embeddings gather, post-LN encoder layers
with biased dense projections, softmax attention with key-padding mask, erf-GELU FFN (HF modeling_bert.py: BertEmbeddings,
BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput) -- runs through ``ct_clip_amd.functional``.
The pooler is skipped: CT-CLIP discards it (ct_clip.py:686,762).
"""
import math
import warnings

import torch

from . import functional as Fn

_warned_dropout = False


def is_hf_bert(module):
    return hasattr(module, "embeddings") and hasattr(module, "encoder") and hasattr(module.encoder, "layer") \
        and hasattr(module.embeddings, "word_embeddings")


def bert_last_hidden_state(bert, input_ids, attention_mask, dtype):
    """Returns last_hidden_state as a (B*T, hidden) activation in ``dtype``."""
    global _warned_dropout
    cfg = bert.config
    if getattr(cfg, "hidden_act", "gelu") != "gelu":
        raise NotImplementedError(f"hidden_act={cfg.hidden_act!r}: only erf-GELU BERT is implemented")
    if getattr(cfg, "position_embedding_type", "absolute") != "absolute":
        raise NotImplementedError("only absolute position embeddings are implemented")
    if bert.training and (cfg.hidden_dropout_prob > 0 or cfg.attention_probs_dropout_prob > 0) and not _warned_dropout:
        warnings.warn("ct_clip_amd: BERT dropout (p=%.2f/%.2f) is not applied by the HIP text tower; the forward is "
                      "deterministic (the reference's train-mode loss is stochastic)" %
                      (cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob))
        _warned_dropout = True
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
    keymask = None
    if attention_mask is not None:
        # additive key mask, as HF builds it: (1 - mask) * finfo.min
        keymask = ((1.0 - attention_mask.to(device=dev, dtype=torch.float32)) * torch.finfo(torch.float32).min).contiguous()
    scale = 1.0 / math.sqrt(dh)
    for layer in bert.encoder.layer:
        sa, so = layer.attention.self, layer.attention.output
        q = Fn.linear(x, sa.query.weight, sa.query.bias)
        k = Fn.linear(x, sa.key.weight, sa.key.bias)
        v = Fn.linear(x, sa.value.weight, sa.value.bias)
        c = Fn.SdpaFn.apply(q, k, v, keymask, Bsz, T, nh, dh, scale)
        x = Fn.layer_norm(Fn.linear(c, so.dense.weight, so.dense.bias, residual=x), so.LayerNorm.weight, so.LayerNorm.bias, eps)
        u = Fn.linear(x, layer.intermediate.dense.weight, layer.intermediate.dense.bias)
        m = Fn.GeluFn.apply(u)
        x = Fn.layer_norm(Fn.linear(m, layer.output.dense.weight, layer.output.dense.bias, residual=x),
                          layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, eps)
    return x
