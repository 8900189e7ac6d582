from ct_clip_amd import get_optimizer  # noqa: F401
