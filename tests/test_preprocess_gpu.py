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


def test_nii_img_to_tensor_drop_in(monkeypatch):
    """ct_clip_amd.preprocess.nii_img_to_tensor(path, df) = the reference method's signature (data.py:92): file name -> metadata row ->
    tensor, with nibabel's loader replaced by an in-memory image (integer voxels with an identity header scaling, and a float image)."""
    import sys
    import types
    import pandas as pd
    from oracle import preprocess_oracle as PO
    from ct_clip_amd import preprocess as PP
    vox = PO.synthetic_volume(21, (48, 40, 30))

    class _Obj:
        def __init__(self, arr, slope, inter):
            self._a, self.slope, self.inter = arr, slope, inter

        def get_unscaled(self):
            return self._a

    class _Img:
        def __init__(self, arr, slope=1.0, inter=0.0):
            self.dataobj = _Obj(arr, slope, inter)
            self._f = arr.astype(np.float64) * slope + inter

        def get_fdata(self):
            return self._f
    store = {"/d/a.nii.gz": _Img(vox), "/d/b.nii.gz": _Img(vox, 0.5, 3.0)}
    fake = types.ModuleType("nibabel")
    fake.load = lambda path: store[str(path)]
    monkeypatch.setitem(sys.modules, "nibabel", fake)
    df = pd.DataFrame([dict(VolumeName="a.nii.gz", RescaleSlope=1.0, RescaleIntercept=-1024.0, XYSpacing="[0.9, 0.9]", ZSpacing=2.0),
                       dict(VolumeName="b.nii.gz", RescaleSlope=2.0, RescaleIntercept=-100.0, XYSpacing="[0.75, 0.75]", ZSpacing=1.5)])
    assert PP.parse_xy_spacing("[0.9, 0.9]") == 0.9
    for name, arr, slope, inter, xy, z in (("a", vox.astype(np.float64), 1.0, -1024.0, 0.9, 2.0),
                                          ("b", vox.astype(np.float64) * 0.5 + 3.0, 2.0, -100.0, 0.75, 1.5)):
        got = PP.nii_img_to_tensor(f"/d/{name}.nii.gz", df, device=DEV).cpu()
        want = PO.volume_to_tensor(arr, slope, inter, xy, z)
        assert got.shape == (1, 240, 480, 480)
        torch.testing.assert_close(got, want, rtol=0, atol=2e-7)


def test_pinned_ring_pipeline_equals_synchronous_path():
    """The host half of the input pipeline (ct_clip_amd.preprocess.VolumeUploader: pinned two-slot int16 ring + copy stream): volume k + 1 is
    uploaded and preprocessed under the main stream's work on volume k; every output equals the synchronous `volume_to_tensor` bit for bit,
    also when the ring wraps (5 volumes through 2 slots), with a pre-staged pinned buffer, into a caller-provided batch slot, and for a
    non-int16 source (synchronous fallback)."""
    from oracle import preprocess_oracle as PO
    from ct_clip_amd import preprocess as PP
    target = (48, 40, 24)
    up = PP.VolumeUploader(DEV, max_voxels=64 * 64 * 40, slots=2, target_shape=target)
    cases = [(PO.synthetic_volume(30 + i, sh), 0.8 + 0.1 * i, -500.0 + 7 * i, xy, z)
             for i, (sh, xy, z) in enumerate([((40, 30, 50), 0.75, 1.5), ((37, 41, 23), 1.3, 2.2), ((64, 64, 30), 0.5, 1.0), ((9, 7, 5), 2.0, 5.0),
                                              ((50, 33, 40), 0.9, 1.1)])]
    want = [PP.volume_to_tensor(v, s, i, xy, z, device=DEV, target_shape=target) for v, s, i, xy, z in cases]
    a, b = torch.randn(2048, 2048, device=DEV), torch.randn(2048, 2048, device=DEV)
    batch = torch.empty((len(cases), 1, target[2], target[0], target[1]), dtype=torch.float32, device=DEV)
    tick = up.submit(*cases[0], out=batch[0])
    for k in range(len(cases)):
        nxt = up.submit(*cases[k + 1], out=batch[k + 1]) if k + 1 < len(cases) else None      # volume k + 1 goes up ...
        for _ in range(3):
            a = (a @ b) * 1e-3                                                                  # ... under the "step" of volume k
        got = up.result(tick)
        assert got.data_ptr() == batch[k].data_ptr()
        assert torch.equal(got.view_as(want[k]), want[k]), k
        tick = nxt
    # a decoder that writes straight into the pinned slot
    v, s, i, xy, z = cases[2]
    slot = up._k
    up.host_buffer(slot)[:v.size].copy_(torch.as_tensor(v).view(-1))
    got = up.result(up.submit(None, s, i, xy, z, staged=v.shape))
    assert torch.equal(got, want[2])
    # float sources take the synchronous path with the same result
    vf = (cases[1][0].astype(np.float64) * 0.61)
    got = up.result(up.submit(vf, *cases[1][1:]))
    assert torch.equal(got, PP.volume_to_tensor(vf, *cases[1][1:], device=DEV, target_shape=target))
