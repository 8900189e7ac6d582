"""The oracle's restatement of the reference preprocessing (oracle/preprocess_oracle.py <- scripts/data.py:12-34,92-162) against
tests/golden/preprocess.pt, which holds outputs of the REAL reference method (oracle/gen_golden.py preprocess)."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "preprocess.pt")
STRIDE = (7, 11, 13)


def load():
    return torch.load(GOLD, weights_only=False)


def make_volume(rec):
    vox = PO.synthetic_volume(rec["seed"], rec["shape"])
    return vox.astype(np.float32) * 0.37 if rec["dtype"] == "float32" else vox


def check_against_golden(y, rec, exact):
    assert y.shape == (1, 240, 480, 480) and y.dtype == torch.float32
    y = y.cpu()
    sd, sh, sw = STRIDE
    sample, slab = y[0, ::sd, ::sh, ::sw], y[0, 117:123, 236:244, 232:248]
    if exact:
        assert torch.equal(sample, rec["sample"]) and torch.equal(slab, rec["slab"])
        assert int((y == -1).sum()) == int(rec["n_pad"])
    else:
        # f64 blend with a different association order, then one rounding to f32: equal up to an ulp at rounding ties
        torch.testing.assert_close(sample, rec["sample"], rtol=0, atol=2e-7)
        torch.testing.assert_close(slab, rec["slab"], rtol=0, atol=2e-7)
        assert float((sample == rec["sample"]).float().mean()) > 0.999
        assert abs(int((y == -1).sum()) - int(rec["n_pad"])) <= 8
    assert abs(float(y.double().sum()) - float(rec["sum"])) < 1e-3 * (1 if exact else 10)
    assert abs(float(y.double().abs().sum()) - float(rec["abs_sum"])) < 1e-3 * (1 if exact else 10)


@pytest.mark.parametrize("name", ["pad", "float"])
def test_restatement_matches_reference_fixture(name):
    rec = load()[name]
    y = PO.volume_to_tensor(make_volume(rec), rec["slope"], rec["intercept"], rec["xy"], rec["z"])
    check_against_golden(y, rec, exact=True)


def test_crop_pad_geometry_small():
    """centre crop on one axis, centre pad on the others, pad value -1, clip to [-1, 1] (data.py:122-160) at a small target shape"""
    vox = PO.synthetic_volume(3, (40, 30, 50))
    y = PO.volume_to_tensor(vox, 1.0, -1024.0, 0.75, 1.5, target_shape=(32, 48, 40))      # identity resample: (40, 30, 50) -> crop h, pad w, crop d
    assert y.shape == (1, 40, 32, 48)
    ref = np.clip(vox.astype(np.float64) - 1024.0, -1000, 1000) / 1000
    inner = y[0, :, :, 9:39].numpy()                                                      # w padded by (48 - 30) // 2 = 9 before
    np.testing.assert_allclose(inner, ref[4:36, :, 5:45].transpose(2, 0, 1).astype(np.float32), rtol=0, atol=1e-7)
    assert float(y[0, :, :, :9].max()) == -1.0 and float(y[0, :, :, 39:].min()) == -1.0
    assert float(y.max()) <= 1.0 and float(y.min()) >= -1.0
