"""Shared test helpers: build the product model for a golden case."""
import torch


class TextBatch:
    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device):
        return TextBatch(self.input_ids.to(device), self.attention_mask.to(device))


def build_model(c, state_dict, device, compute_dtype):
    from transformers import BertConfig, BertModel
    import ct_clip_amd
    torch.manual_seed(c["seed"])
    enc = ct_clip_amd.CTViT(dim=c["dim"], codebook_size=c["codebook"], image_size=c["image"], patch_size=c["patch"],
                            temporal_patch_size=c["tpatch"], spatial_depth=c["sdepth"], temporal_depth=c["tdepth"],
                            dim_head=c["dim_head"], heads=c["heads"], compute_dtype=compute_dtype)
    bcfg = BertConfig(vocab_size=c["vocab"], hidden_size=c["bert_hidden"], num_hidden_layers=c["bert_layers"],
                      num_attention_heads=c["bert_heads"], intermediate_size=c["bert_inter"],
                      max_position_embeddings=c["max_pos"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bert = BertModel(bcfg)
    hw = c["image"] // c["patch"]
    clip = ct_clip_amd.CTCLIP(image_encoder=enc, text_encoder=bert, dim_text=c["bert_hidden"], dim_image=hw * hw * c["dim"],
                              dim_latent=c["dim_latent"], extra_latent_projection=False, use_mlm=False,
                              downsample_image_embeds=False, use_all_token_embeds=False, compute_dtype=compute_dtype)
    if state_dict is not None:
        missing, unexpected = clip.load_state_dict(state_dict, strict=False)
        assert not unexpected, unexpected
        bad = [k for k in missing if not any(u in k for u in ("_extra.", "to_pixels", "to_patch_emb_first_frame", "pooler.",
                                                                "position_ids", "token_type_ids"))]
        assert not bad, bad
    return clip.to(device)


def check_grad(rec, mine, rtol, atol_rel, floor=0.0):
    """floor: absolute noise floor (gradients that are mathematically zero, e.g. a bias in front of a LayerNorm, come
    out as rounding noise ~1e-11 in both implementations)."""
    ref = rec["value"]
    m = mine if rec["full"] else mine.reshape(-1)[::rec["stride"]]
    scale = float(ref.abs().max()) + 1e-12
    torch.testing.assert_close(m.float().cpu(), ref, rtol=rtol, atol=atol_rel * scale + floor)
