"""The HIP input pipeline (csrc/preprocess.hip, ct_clip_amd/preprocess.py) against the real-reference fixture and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_preprocess_cpu import check_against_golden, load, make_volume  # noqa: E402

DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("name", ["pad", "crop", "float"])
def test_kernel_matches_reference_fixture(name):
    from ct_clip_amd import preprocess as PP
    rec = load()[name]
    y = PP.volume_to_tensor(make_volume(rec), rec["slope"], rec["intercept"], rec["xy"], rec["z"], device=DEV)
    check_against_golden(y, rec, exact=False)


@pytest.mark.parametrize("shape,xy,z,target", [((40, 30, 50), 0.75, 1.5, (32, 48, 40)), ((37, 41, 23), 1.3, 2.2, (48, 40, 24)),
                                               ((64, 64, 30), 0.5, 1.0, (40, 40, 24)), ((9, 7, 5), 2.0, 5.0, (16, 16, 8))])
@pytest.mark.parametrize("dtype", ["int16", "float32", "float64"])
def test_kernel_matches_oracle_small(shape, xy, z, target, dtype):
    from oracle import preprocess_oracle as PO
    from ct_clip_amd import preprocess as PP
    vox = PO.synthetic_volume(5, shape)
    if dtype != "int16":
        vox = (vox.astype(np.float64) * 0.61).astype(dtype)
    want = PO.volume_to_tensor(vox, 0.8, -500.0, xy, z, target_shape=target)
    got = PP.volume_to_tensor(vox, 0.8, -500.0, xy, z, device=DEV, target_shape=target).cpu()
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=0, atol=2e-7)
    assert float((got == want).float().mean()) > 0.995
