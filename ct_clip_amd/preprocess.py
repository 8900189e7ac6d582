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


class VolumeUploader:
    """The host half of the device-side input pipeline: a ring of PINNED int16 staging buffers and a copy stream, so that the upload (157 MB per
    512 x 512 x 300 volume: a quarter of the float32 bytes the reference ships, scripts/data.py:92-162) and the preprocessing kernel of volume
    k + 1 run under whatever the caller's stream does with volume k.

        up = VolumeUploader(device)
        t = up.submit(voxels, slope, intercept, xy, z)      # returns at once: pinned copy, async H2D, kernel on the copy stream
        ...                                                 # (the training step of the previous batch)
        x = up.result(t)                                    # (1, 240, 480, 480) f32; the caller's stream waits for the kernel's event

    Results are bit-identical to `volume_to_tensor` (same kernel, same arguments).  Only int16 voxel arrays take the ring (the stored format of
    CT-RATE); anything else goes through `volume_to_tensor` synchronously."""

    class Ticket:
        __slots__ = ("out", "done")

        def __init__(self, out, done):
            self.out, self.done = out, done

    def __init__(self, device=None, max_voxels=512 * 512 * 640, slots=2, target_shape=TARGET_SHAPE):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        from . import streams
        self.stream = streams.concurrent_stream(self.device, "h2d")      # (one copy stream per device and process, probed: streams.py)
        self.target_shape = target_shape
        self.max_voxels = int(max_voxels)
        self._host = [torch.empty(self.max_voxels, dtype=torch.int16).pin_memory() for _ in range(slots)]
        self._dev = [torch.empty(self.max_voxels, dtype=torch.int16, device=self.device) for _ in range(slots)]
        self._copied = [None] * slots        # event: the slot's H2D copy has left the pinned buffer (the host may overwrite it)
        self._k = 0

    def host_buffer(self, slot):
        """The pinned staging buffer of a slot (a decoder can write the voxels here directly and pass `staged=(H, W, D)` to submit).  Waits until
        the slot's previous upload has LEFT the pinned memory (the asynchronous H2D copy of submit k - slots may still be reading it)."""
        s = slot % len(self._host)
        if self._copied[s] is not None:
            self._copied[s].synchronize()
        return self._host[s]

    def submit(self, voxels, slope, intercept, xy_spacing, z_spacing, out=None, staged=None):
        """voxels: (H, W, D) int16 array (numpy / torch, host).  staged = (H, W, D): the voxels are ALREADY in this call's pinned slot
        (`host_buffer(k)` of the k-th submit): no host copy.  out: optional (1, D', H', W') f32 destination on the device."""
        if staged is None:
            t = torch.as_tensor(voxels)
            if t.dtype != torch.int16 or t.is_cuda or t.numel() > self.max_voxels:
                y = volume_to_tensor(voxels, slope, intercept, xy_spacing, z_spacing, device=self.device, target_shape=self.target_shape)
                if out is not None:
                    out.copy_(y.view_as(out))
                    y = out
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                return self.Ticket(y, ev)
            shape = tuple(t.shape)
        else:
            shape = tuple(staged)
        n = shape[0] * shape[1] * shape[2]
        s = self._k % len(self._host)
        self._k += 1
        if staged is None:
            if self._copied[s] is not None:
                self._copied[s].synchronize()            # the previous upload from this pinned buffer has been read by the copy engine
            self._host[s][:n].copy_(t.contiguous().view(-1))
        be = _be.get()
        with torch.cuda.stream(self.stream):
            # (same stream as the previous kernel that read this slot's device buffer: ordered)
            self._dev[s][:n].copy_(self._host[s][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._copied[s] = ev
            y = be.preprocess_volume(self._dev[s][:n].view(shape), slope, intercept, xy_spacing, z_spacing, TARGET_SPACING[0], TARGET_SPACING[2],
                                     self.target_shape, HU_RANGE, out=out)
            done = torch.cuda.Event()
            done.record(self.stream)
        return self.Ticket(y, done)

    def result(self, ticket):
        """The tensor of a submitted volume, ordered behind its kernel on the CALLER's current stream (no host synchronisation)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ticket.done)
        ticket.out.record_stream(cur)
        return ticket.out
