"""Drop-in for the reference package ``ct_clip`` (``from ct_clip import CTCLIP``)."""
from ct_clip_amd import CTCLIP  # noqa: F401
