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


def perturb_1d(clip, seed):
    """The perturbation oracle/gen_golden.py applies to the reference after init (every 1-D parameter += 0.1 * randn)."""
    g = torch.Generator().manual_seed(99 + seed)
    with torch.no_grad():
        for name, p in clip.named_parameters():
            if p.ndim <= 1 and p.numel() > 0 and name != "temperature":
                p.add_(torch.randn(p.shape, generator=g) * 0.1)


def synth_inputs(c):
    """Same seeded inputs as oracle/gen_golden.py:synth_inputs (the full-size fixture stores only their fingerprints)."""
    g = torch.Generator().manual_seed(1234 + c["seed"])
    video = torch.rand(c["batch"], 1, c["frames"], c["image"], c["image"], generator=g) * 2 - 1
    T = c["T"]
    ids = torch.randint(3, c["vocab"], (c["batch"], T), generator=g)
    lens = torch.randint(T // 2, T + 1, (c["batch"],), generator=g)
    ids[:, 0] = 1
    mask = torch.arange(T)[None, :] < lens[:, None]
    for b in range(c["batch"]):
        ids[b, lens[b] - 1] = 2
    ids = ids * mask
    return video, ids, mask.long()


def fingerprint_ok(t, fp, exact=True):
    f = t.detach().reshape(-1).double()
    return (tuple(t.shape) == tuple(fp["shape"]) and torch.equal(t.detach().reshape(-1)[:4].cpu(), fp["head"])
            and torch.equal(t.detach().reshape(-1)[-4:].cpu(), fp["tail"])
            and abs(float(f.sum()) - fp["sum"]) <= 1e-9 * max(1.0, fp["abssum"]))


def check_grad(rec, mine, rtol, atol_rel, floor=0.0):
    """floor: absolute noise floor (gradients that are mathematically zero, e.g. a bias in front of a LayerNorm, come
    out as rounding noise ~1e-11 in both implementations)."""
    ref = rec["value"]
    m = mine if rec["full"] else mine.reshape(-1)[::rec["stride"]]
    scale = float(ref.abs().max()) + 1e-12
    torch.testing.assert_close(m.float().cpu(), ref, rtol=rtol, atol=atol_rel * scale + floor)
