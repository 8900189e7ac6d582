"""Import shim: ``from zero_shot import CTClipInference`` (scripts/run_zero_shot.py:4) -> the MI355X implementation."""
from ct_clip_amd.zero_shot import CTClipInference, ZeroShotClassifier, PATHOLOGIES  # noqa: F401
