"""CTCLIP (drop-in for ``ct_clip.CTCLIP``, reference CT_CLIP/ct_clip/ct_clip.py:407-901) on gfx950 kernels.

Keeps the reference constructor / ``forward`` signature, return modes and ``state_dict`` keys.  Both towers, the pooling,
the two latent projections, l2norm, the logit matmul and the symmetric InfoNCE run as HIP kernels; under
``torch.distributed`` the loss uses gathered negatives (all-gather of the raw latents over RCCL).
"""
import copy
from pathlib import Path

import torch
from torch import nn

from . import bert as _bert
from . import distributed as _dist
from . import functional as Fn
from .ctvit import default_compute_dtype


class CTCLIP(nn.Module):
    def __init__(self, *, image_encoder=None, text_encoder=None, dim_text=512, dim_image=512, dim_latent=512,
                 num_text_tokens=28897, text_enc_depth=6, text_seq_len=256, text_heads=8, text_dim_head=64,
                 text_has_cls_token=False, text_pad_id=0, text_rotary_pos_emb=False, text_causal_mask=False,
                 text_eos_id=None, text_encode_without_mask=False, visual_enc_depth=6, visual_heads=8, visual_dim_head=64,
                 visual_image_size=256, visual_patch_size=32, visual_patch_dropout=0.5, visual_has_cls_token=False,
                 channels=3, use_all_token_embeds=False, downsample_image_embeds=False,
                 decoupled_contrastive_learning=False, extra_latent_projection=False, use_mlm=False,
                 text_ssl_loss_weight=0.05, use_visual_ssl=False, visual_ssl=None, visual_ssl_type="simsiam",
                 visual_ssl_hidden_layer=-1, simclr_temperature=0.1, image_ssl_loss_weight=0.05,
                 multiview_loss_weight=0.1, checkpoint_during_training=False, tokenizer=None, compute_dtype=None,
                 gather_negatives=True, **kwargs):
        super().__init__()
        self.dtype = torch.float32
        self.dim_text, self.dim_image, self.dim_latent = dim_text, dim_image, dim_latent
        if image_encoder is None or text_encoder is None:
            raise NotImplementedError("CT-CLIP always supplies external encoders (run_train.py:17-42); the built-in "
                                      "TextTransformer/VisionTransformer of ct_clip.py:103-385 are out of scope")
        for flag, name in ((use_all_token_embeds, "use_all_token_embeds"), (downsample_image_embeds, "downsample_image_embeds"),
                           (decoupled_contrastive_learning, "decoupled_contrastive_learning"),
                           (extra_latent_projection, "extra_latent_projection"), (use_mlm, "use_mlm"),
                           (use_visual_ssl or visual_ssl is not None, "use_visual_ssl"), (text_causal_mask, "text_causal_mask")):
            if flag:
                raise NotImplementedError(f"{name}=True is disabled in every CT-CLIP entry script (run_train.py:31-42)")
        if not _bert.is_hf_bert(text_encoder):
            raise NotImplementedError("text_encoder must be a HuggingFace BertModel (run_train.py:9)")
        self.text_transformer = text_encoder
        self.visual_transformer = image_encoder
        self.text_pad_id = text_pad_id
        self.use_mlm = False
        self.use_visual_ssl = False
        self.use_all_token_embeds = False
        self.extra_latent_projection = False
        self.decoupled_contrastive_learning = False
        self.text_ssl_loss_weight = 0
        self.image_ssl_loss_weight = 0
        self.multiview_loss_weight = multiview_loss_weight

        self.to_text_latent = nn.Linear(dim_text, dim_latent, bias=False)
        self.to_visual_latent = nn.Linear(dim_image, dim_latent, bias=False)
        self.temperature = nn.Parameter(torch.tensor(1.0))
        # kept for checkpoint-key compatibility (ct_clip.py:579-581); never used, never receive gradients
        self.to_text_latent_extra = copy.deepcopy(self.to_text_latent)
        self.to_visual_latent_extra = copy.deepcopy(self.to_visual_latent)

        # ct_clip.py:585 downloads a tokenizer; offline-safe here: only fetched on demand by tokenize()
        self.tokenizer = tokenizer
        self.compute_dtype = compute_dtype or getattr(image_encoder, "compute_dtype", None) or default_compute_dtype()
        if hasattr(image_encoder, "compute_dtype"):
            image_encoder.compute_dtype = self.compute_dtype
        self.gather_negatives = gather_negatives
        # Precision of the text tower when the image tower computes in bf16 (attribute, or CTCLIP_TEXT_DTYPE):
        #   None / "mixed" (default)  f32 residual stream, LayerNorms, bias / dropout / residual adds and GELU; bf16 only where the matrix cores
        #                             read it (GEMM operands, q | k | v, the attention core) -- torch.autocast's split for an HF BertModel.  The
        #                             all-bf16 tower's 48 roundings of the post-LN stream were most of the teacher-forced loss error (1.0e-3 at
        #                             12+12 layers against 1.9e-5 with an f32 tower); M = B*T rows x 768 in f32 is a few MB per tensor;
        #   torch.float32 / "f32"     everything in f32, GEMMs included (16x the matrix-core time);
        #   torch.bfloat16 / "bf16"   the all-bf16 tower of rounds 1-4.
        import os
        env = os.environ.get("CTCLIP_TEXT_DTYPE", "").lower()
        self.head_dtype = torch.bfloat16 if os.environ.get("CTCLIP_HEAD_DTYPE", "").lower() in ("bf16", "bfloat16") else torch.float32
        self.text_compute_dtype = (torch.float32 if env in ("f32", "fp32", "float32") else torch.bfloat16 if env in ("bf16", "bfloat16")
                                   else "mixed" if env == "mixed" else None)

    def pool_tokens(self, enc_tokens):
        """(Bi, t, h, w, d) quantised tokens -> (Bi, h*w*d) depth mean (ct_clip.py:724,740).  bf16 mode: the pooled vector is f32 and so is
        everything behind it (to_visual_latent against the f32 master weight, l2norm, logits): the head of the image tower is a 151-M-weight
        GEMV of B rows -- HBM-bound either way (+0.3 GB of weight reads per pass) -- and its bf16 operands were most of what the image side
        contributed to the loss error with teacher-forced codes (tiny configuration: 6.0e-4 of 1.33e-3).  CTCLIP_HEAD_DTYPE=bf16 restores it."""
        Bi, t = enc_tokens.shape[0], enc_tokens.shape[1]
        out = torch.float32 if (enc_tokens.dtype == torch.bfloat16 and self.head_dtype != torch.bfloat16) else None
        return Fn.PoolFn.apply(enc_tokens.reshape(Bi, t, -1), out)

    def _text_dtypes(self):
        """-> (activation dtype, matrix-core operand dtype or None = the same)."""
        t = self.text_compute_dtype
        if t in (None, "mixed"):
            return (torch.float32, torch.bfloat16) if self.compute_dtype == torch.bfloat16 else (self.compute_dtype, None)
        return t, None

    def load(self, path):
        """ct_clip.py:593-597.  Additive: the trainer's periodic checkpoints are written from the DDP-wrapped model with
        `accelerator.get_state_dict(..., unwrap=False)` (CTCLIPTrainer.py:331-337), i.e. every key carries a `module.` prefix when the run
        was multi-GPU; such a dict is accepted as well (the prefix is stripped when ALL keys have it)."""
        path = Path(path)
        assert path.exists()
        pt = torch.load(str(path), map_location="cpu")
        if len(pt) and all(k.startswith("module.") for k in pt):
            pt = {k[len("module."):]: v for k, v in pt.items()}
        self.load_state_dict(pt)
        Fn.bump_weight_epoch()

    def _text_stream(self, device):
        import os
        if device.type != "cuda" or os.environ.get("CTCLIP_TEXT_STREAM", "1") == "0":
            return None
        return Fn.shared_side_stream(device, "text")      # one per device and process (the fused optimiser joins it before reading gradients)

    def tokenize(self, prompt):
        if self.tokenizer is None:
            from transformers import BertTokenizer
            self.tokenizer = BertTokenizer.from_pretrained("microsoft/BiomedVLP-CXR-BERT-specialized", do_lower_case=True)
        dev = self.temperature.device
        return self.tokenizer(prompt, return_tensors="pt", padding="max_length", truncation=True, max_length=512).to(dev)

    # ---- the two towers on their own (additive API: zero-shot scoring, latent export -- the reference re-runs the whole forward)
    def encode_text(self, text):
        """HF BatchEncoding-like (.input_ids, .attention_mask) -> (Bt, dim_latent) l2-normalised f32 text latents (ct_clip.py:685-686,762,771)."""
        ids, mask = text.input_ids, text.attention_mask
        cls = _bert.bert_last_hidden_state(self.text_transformer, ids, mask, *self._text_dtypes(), cls_only=True)      # (Bt, dim_text)
        return Fn.l2norm_f32(Fn.linear(cls, self.to_text_latent.weight, out_dtype=torch.float32))

    def text_latents_raw(self, ids, mask):
        """(Bt, T) ids / mask -> (Bt, dim_latent) f32 text latents BEFORE l2norm (ct_clip.py:685-686,762,765), differentiable."""
        cls = _bert.bert_last_hidden_state(self.text_transformer, ids, mask, *self._text_dtypes(), cls_only=True)      # (Bt, dim_text)
        cls = Fn.grad_ready(cls, self.to_text_latent)
        return Fn.linear(cls, self.to_text_latent.weight, out_dtype=torch.float32)

    def encode_image(self, image, return_tokens=False):
        """(Bi, 1, F, H, W) volume -> (Bi, dim_latent) l2-normalised f32 image latents (ct_clip.py:715-767,771) [, the token grid]."""
        enc_tokens = self.visual_transformer(image, return_encoded_tokens=True)
        enc_image = self.pool_tokens(enc_tokens)
        lat = Fn.l2norm_f32(Fn.visual_latent(enc_image, self.to_visual_latent.weight))
        return (lat, enc_tokens) if return_tokens else lat

    def forward(self, text, image, device=None, return_loss=False, return_encodings=False, return_latents=False,
                freeze_image_encoder=False, freeze_text_encoder=False, text_to_image=True, aug_text=None, aug_image=None):
        # freeze_* are accepted and ignored exactly as in the reference (ct_clip.py:709-715)
        if aug_text is not None or aug_image is not None:
            raise NotImplementedError("multiview augmentation is never used by CT-CLIP's entry scripts")
        dt, od = self._text_dtypes()
        ids, mask = text.input_ids, text.attention_mask
        Bt, T = ids.shape
        # The text tower (M = B*T rows: far too small to fill 256 CUs) runs on a side stream underneath the image tower;
        # autograd replays each tower's backward on the stream its forward used.
        # only enc_text[:, 0, :] is read below (ct_clip.py:762) unless the caller asks for the encodings: (Bt, dim_text) rows then, else (Bt*T, dim_text)
        cls_only = not return_encodings
        side = self._text_stream(ids.device if ids.is_cuda else self.temperature.device)
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                enc_text = _bert.bert_last_hidden_state(self.text_transformer, ids, mask, dt, od, cls_only=cls_only)
        else:
            enc_text = _bert.bert_last_hidden_state(self.text_transformer, ids, mask, dt, od, cls_only=cls_only)
        enc_tokens = self.visual_transformer(image, return_encoded_tokens=True)                 # (Bi, t, h, w, d)
        if side is not None:
            main.wait_stream(side)
            enc_text.record_stream(main)
        Bi = enc_tokens.shape[0]
        enc_image = self.pool_tokens(enc_tokens)                                                 # ct_clip.py:724,740
        if return_encodings:
            return enc_text.view(Bt, T, -1), enc_image
        cls = enc_text                                                                           # enc_text[:, 0, :] (ct_clip.py:762)
        cls = Fn.grad_ready(cls, self.to_text_latent)
        enc_image = Fn.grad_ready(enc_image, self.to_visual_latent)
        text_lat = Fn.linear(cls, self.to_text_latent.weight, out_dtype=torch.float32)          # (Bt, Dl) f32, pre-l2norm
        image_lat = Fn.visual_latent(enc_image, self.to_visual_latent.weight)                   # (Bi, Dl) f32, pre-l2norm
        if return_latents:
            return Fn.l2norm_f32(text_lat), Fn.l2norm_f32(image_lat), enc_tokens
        if not return_loss:
            # ct_clip.py:805-807: einsum('b d, b d -> b') * temp with broadcasting (e.g. 2 prompts vs 1 volume)
            return Fn.LatentSimilarityFn.apply(text_lat, image_lat, self.temperature)
        assert Bt == Bi, "contrastive loss needs as many texts as volumes"
        replicas = 1
        if self.gather_negatives and _dist.collectives_on():
            text_lat, image_lat = _dist.all_gather_latents(text_lat, image_lat)
            replicas = _dist.world_size()
        return Fn.ClipLossFn.apply(text_lat, image_lat, self.temperature, replicas)
