"""Device-side input pipeline: the arithmetic of ``CTReportDataset.nii_img_to_tensor``
(scripts/data.py:92-162; the same code in scripts/data_inference_nii.py:96-166) as one HIP kernel (csrc/preprocess.hip).

The reference decodes the NIfTI file to float64 on the host, rescales, resamples with ``F.interpolate`` on the CPU, clips, crops /
pads and ships a 221-MB float32 volume to the GPU.  Here the host uploads the voxels as stored (int16: a quarter of the bytes) and the
kernel writes the (1, 240, 480, 480) float32 model input straight into HBM.  File decoding stays with the caller (nibabel is the
reference's own dependency for that and is optional here).
"""
import torch

from . import backend as _be

TARGET_SPACING = (0.75, 0.75, 1.5)          # x, y, z in mm (data.py:104-106)
TARGET_SHAPE = (480, 480, 240)              # h, w, d (data.py:129)
HU_RANGE = (-1000.0, 1000.0)                # data.py:122


def parse_xy_spacing(field):
    """The metadata column XYSpacing is a string like "[0.78, 0.78]" (data.py:100)."""
    return float(field[1:][:-2].split(",")[0])


def volume_to_tensor(voxels, slope, intercept, xy_spacing, z_spacing, device=None, target_shape=TARGET_SHAPE):
    """voxels: the (H, W, D) array of the NIfTI file (numpy or torch; int16, float32 or float64) -> (1, D', H', W') float32 on the device,
    equal to the reference's ``nii_img_to_tensor`` output for the same volume and metadata."""
    t = torch.as_tensor(voxels)
    if t.dtype not in (torch.int16, torch.float32, torch.float64):
        t = t.to(torch.float64)             # what nibabel's get_fdata() would have produced
    dev = torch.device(device) if device is not None else (t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    t = t.to(dev, non_blocking=True).contiguous()
    return _be.get().preprocess_volume(t, slope, intercept, xy_spacing, z_spacing, TARGET_SPACING[0], TARGET_SPACING[2], target_shape, HU_RANGE)


def nii_img_to_tensor(path, df, device=None):
    """Drop-in for CTReportDataset.nii_img_to_tensor(path, df) (data.py:92-162): needs nibabel for the decode."""
    import nibabel as nib
    img = nib.load(str(path))
    dobj = img.dataobj
    unscaled = getattr(dobj, "get_unscaled", None)
    identity = getattr(dobj, "slope", 1.0) in (None, 1.0) and getattr(dobj, "inter", 0.0) in (None, 0.0)
    # stored integers go up as they are (the header's own scaling is the identity); anything else as get_fdata()'s float64
    arr = unscaled() if (unscaled is not None and identity) else img.get_fdata()
    file_name = str(path).split("/")[-1]
    row = df[df["VolumeName"] == file_name]
    slope, intercept = float(row["RescaleSlope"].iloc[0]), float(row["RescaleIntercept"].iloc[0])
    xy, z = parse_xy_spacing(row["XYSpacing"].iloc[0]), float(row["ZSpacing"].iloc[0])
    return volume_to_tensor(arr, slope, intercept, xy, z, device=device)
