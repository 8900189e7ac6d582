"""Drop-in for the reference package ``transformer_maskgit`` (only the hot-path symbol ``CTViT`` + ``get_optimizer``).
Put ``<repo>/dropin`` and ``<repo>`` on PYTHONPATH ahead of the reference packages; scripts/run_train.py then runs unchanged."""
from ct_clip_amd import CTViT  # noqa: F401
