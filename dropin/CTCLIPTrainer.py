"""Drop-in for scripts/CTCLIPTrainer.py (``from CTCLIPTrainer import CTClipTrainer``)."""
from ct_clip_amd import CTClipTrainer  # noqa: F401
