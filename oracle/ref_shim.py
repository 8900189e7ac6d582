"""TEST INFRASTRUCTURE ONLY -- shim loader for the *real* reference under /root/reference.

Used by ``oracle/gen_golden.py`` (and nothing in the product) to import the reference's own
``transformer_maskgit.attention`` / ``transformer_maskgit.ctvit`` / ``ct_clip.ct_clip`` modules
in this GPU-less, network-less container and run them on CPU, so that golden vectors can be
generated from the reference itself.  It cannot travel to the GPU box (``/root/reference`` does
not exist there); only the fixtures it produces under ``tests/golden/`` do.

What blocks a plain import, and the work-around used here:

* ``transformer_maskgit/__init__.py:1-3`` imports the whole MaskGIT stack (ema_pytorch, nibabel,
  cv2, T5 ...).  We register an *empty* parent package whose ``__path__`` points at the reference
  directory, so sub-modules import without running that ``__init__``.
* ``attention.py:6`` needs ``beartype``; ``ctvit.py:10,13`` need ``torchvision``;
  ``ctvit.py:18`` needs ``vector_quantize_pytorch`` (pinned ==1.1.2 in
  ``transformer_maskgit/setup.py:19``, not installed, no wheel): stub modules are injected.
  The VQ stub is ``oracle.vq_restatement.VectorQuantize`` -- a restatement of the published
  1.1.2 cosine-sim codebook algorithm ("parity unpinned": no sdist here to diff against).
* ``ct_clip.py:585`` downloads a tokenizer: ``BertTokenizer.from_pretrained`` is patched to a stub.
* ``attention.py:260`` hard-codes ``torch.device('cuda')`` in ContinuousPositionBias.forward: we
  pre-register the ``rel_pos`` buffer and set ``cache_rel_pos=True`` so that branch is skipped
  (no reference source is edited or copied).
"""
import importlib
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("CTCLIP_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "transformer_maskgit", "transformer_maskgit"))


def _stub_module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    """Idempotently install stub modules + empty parent packages for the reference."""
    if getattr(install_shims, "_done", False):
        return
    sys.dont_write_bytecode = True
    # transformers must be imported BEFORE a fake torchvision is visible (transformers 5.x probes it)
    import transformers  # noqa: F401
    from transformers import BertModel, BertTokenizer  # noqa: F401

    # beartype: identity decorator
    def beartype(fn=None, **_kw):
        if fn is None:
            return lambda f: f
        return fn

    _stub_module("beartype", beartype=beartype)
    _stub_module("beartype.door", is_bearable=lambda *a, **k: True)
    _stub_module("beartype.typing", **{k: getattr(__import__("typing"), k) for k in
                                      ("Tuple", "List", "Optional", "Union", "Callable")})

    # torchvision: inert
    tv = _stub_module("torchvision")
    tv.transforms = _stub_module("torchvision.transforms")
    tv.utils = _stub_module("torchvision.utils")
    tv.datasets = _stub_module("torchvision.datasets")
    tv.models = _stub_module("torchvision.models")
    tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)

    # vector_quantize_pytorch: our restatement of 1.1.2's cosine-sim path
    here = os.path.dirname(os.path.abspath(__file__))
    if os.path.dirname(here) not in sys.path:
        sys.path.insert(0, os.path.dirname(here))
    from oracle.vq_restatement import VectorQuantize
    _stub_module("vector_quantize_pytorch", VectorQuantize=VectorQuantize)

    # empty parent packages so sub-modules import without the heavy __init__
    tm = types.ModuleType("transformer_maskgit")
    tm.__path__ = [os.path.join(REF_ROOT, "transformer_maskgit", "transformer_maskgit")]
    sys.modules["transformer_maskgit"] = tm
    cc = types.ModuleType("ct_clip")
    cc.__path__ = [os.path.join(REF_ROOT, "CT_CLIP", "ct_clip")]
    sys.modules["ct_clip"] = cc

    # tokenizer download stub (ct_clip.py:585)
    class _NullTokenizer:
        pad_token_id = 0

        def __call__(self, *a, **k):
            raise RuntimeError("tokenizer is stubbed offline; feed token ids directly")

    BertTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: _NullTokenizer())
    install_shims._done = True


def load_reference():
    """Returns (attention_module, ctvit_module, ct_clip_module) of the real reference."""
    assert reference_available(), f"reference not found under {REF_ROOT}"
    install_shims()
    att = importlib.import_module("transformer_maskgit.attention")
    ctvit = importlib.import_module("transformer_maskgit.ctvit")
    ctclip = importlib.import_module("ct_clip.ct_clip")
    return att, ctvit, ctclip


def seed_rel_pos(ctvit_model, h, w):
    """Pre-register ContinuousPositionBias.rel_pos (attention.py:259-269 skipped thereafter)."""
    cpb = ctvit_model.spatial_rel_pos_bias
    pos = [torch.arange(h), torch.arange(w)]
    grid = torch.stack(torch.meshgrid(*pos, indexing="ij")).reshape(2, -1).t()  # (hw, 2)
    rel = grid[:, None, :] - grid[None, :, :]
    rel = torch.sign(rel) * torch.log(rel.abs() + 1)
    cpb.cache_rel_pos = True
    cpb.register_buffer("rel_pos", rel, persistent=False)


class TextBatch:
    """Minimal stand-in for a HF BatchEncoding: attribute access + .to()."""

    def __init__(self, input_ids, attention_mask):
        self.input_ids = input_ids
        self.attention_mask = attention_mask

    def to(self, device):
        return TextBatch(self.input_ids.to(device), self.attention_mask.to(device))
