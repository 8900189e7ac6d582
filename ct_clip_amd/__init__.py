"""ct_clip_amd -- MI355X (gfx950) native CT-CLIP training hot path.

Drop-in surface (same names/signatures/state_dict keys as the reference):
    CTViT          <- transformer_maskgit.CTViT          (transformer_maskgit/transformer_maskgit/ctvit.py)
    CTCLIP         <- ct_clip.CTCLIP                      (CT_CLIP/ct_clip/ct_clip.py)
    CTClipTrainer  <- CTCLIPTrainer.CTClipTrainer         (scripts/CTCLIPTrainer.py)
    get_optimizer  <- transformer_maskgit.optimizer.get_optimizer
All arithmetic runs in hand-written HIP kernels behind the C-ABI library libctclip_hip.so (include/ctclip_hip.h).
"""
from .ctvit import CTViT  # noqa: F401
from .ctclip import CTCLIP  # noqa: F401
from .trainer import CTClipTrainer, FusedAdam, hot_path_parameters  # noqa: F401


def get_optimizer(params, lr=1e-4, wd=1e-4, betas=(0.9, 0.99), eps=1e-8, filter_by_requires_grad=False, group_wd_params=True,
                  **kwargs):
    """transformer_maskgit/optimizer.py:10-34 on the fused HIP Adam (params: iterable of (name, param) or params): wd == 0 is Adam,
    otherwise AdamW; with group_wd_params the ndim < 2 parameters form a no-decay group; filter_by_requires_grad drops frozen ones."""
    params = list(params)
    named = params if params and isinstance(params[0], tuple) else [(f"p{i}", p) for i, p in enumerate(params)]
    if filter_by_requires_grad:
        named = [(n, p) for n, p in named if p.requires_grad]
    return FusedAdam(named, lr=lr, betas=betas, eps=eps, weight_decay=wd, group_wd_params=group_wd_params)
