"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's volume preprocessing (scripts/data.py:12-34 resize_array,
:92-162 CTReportDataset.nii_img_to_tensor), parameterised on the target shape so that small cases run in milliseconds.

Pinned: tests/golden/preprocess.pt holds outputs of the REAL reference method (run through oracle/gen_golden.py with nibabel's
loader stubbed by an in-memory array) and tests/test_preprocess_cpu.py checks this restatement against them.
"""
import numpy as np
import torch
import torch.nn.functional as F


def resize_array(array, current_spacing, target_spacing):
    """data.py:12-34"""
    original_shape = array.shape[2:]
    scaling = [current_spacing[i] / target_spacing[i] for i in range(len(original_shape))]
    new_shape = [int(original_shape[i] * scaling[i]) for i in range(len(original_shape))]
    return F.interpolate(array, size=new_shape, mode="trilinear", align_corners=False).cpu().numpy()


def volume_to_tensor(voxels, slope, intercept, xy_spacing, z_spacing, target_shape=(480, 480, 240)):
    """data.py:92-162 from the decoded (H, W, D) array on: float64 like nibabel's get_fdata()."""
    img = slope * np.asarray(voxels, dtype=np.float64) + intercept                     # :113
    img = img.transpose(2, 0, 1)                                                       # :115
    t = torch.tensor(img).unsqueeze(0).unsqueeze(0)
    img = resize_array(t, (z_spacing, xy_spacing, xy_spacing), (1.5, 0.75, 0.75))[0][0]   # :108-119
    img = np.transpose(img, (1, 2, 0))
    img = np.clip(img, -1000, 1000)                                                    # :122-123
    img = (img / 1000).astype(np.float32)                                              # :125
    t = torch.tensor(img)
    dh, dw, dd = target_shape
    h, w, d = t.shape
    hs, ws, ds = max((h - dh) // 2, 0), max((w - dw) // 2, 0), max((d - dd) // 2, 0)   # :135-140
    t = t[hs:min(hs + dh, h), ws:min(ws + dw, w), ds:min(ds + dd, d)]
    ph, pw, pd = (dh - t.size(0)) // 2, (dw - t.size(1)) // 2, (dd - t.size(2)) // 2   # :145-152
    t = F.pad(t, (pd, dd - t.size(2) - pd, pw, dw - t.size(1) - pw, ph, dh - t.size(0) - ph), value=-1)
    return t.permute(2, 0, 1).unsqueeze(0)                                             # :156-160


def synthetic_volume(seed, shape):
    """A smooth int16 'CT': low-frequency field + noise, values across and beyond the clip range (shared by the generator and the tests)."""
    rng = np.random.default_rng(seed)
    h, w, d = shape
    gh, gw, gd = np.meshgrid(np.linspace(0, 3, h), np.linspace(0, 2, w), np.linspace(0, 4, d), indexing="ij")
    field = 900 * np.sin(gh * 2.1 + 0.3) * np.cos(gw * 1.7) + 600 * np.sin(gd * 1.3 + gw) + rng.normal(0, 40, shape)
    return np.clip(field + 200, -2000, 3000).astype(np.int16)
