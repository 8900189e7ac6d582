"""Import shim: ``from forward_data import CTClipInference`` (scripts/run_forward_data.py:4) -> the MI355X implementation."""
from ct_clip_amd.forward_data import CTClipInference  # noqa: F401
