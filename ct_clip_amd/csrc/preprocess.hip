// Device-side input pipeline: scripts/data.py:92-162 (CTReportDataset.nii_img_to_tensor; the same code
// in data_inference_nii.py:96-166) from the decoded NIfTI voxel array to the model's input volume, in ONE pass:
//   HU = slope * raw + intercept                                            (data.py:113)
//   trilinear resample to 0.75 x 0.75 x 1.5 mm, F.interpolate(..., mode='trilinear', align_corners=False) with
//   new size = int(size * spacing / target)                                 (data.py:12-34,118)
//   clip to [-1000, 1000], / 1000, float32                                  (data.py:122-125)
//   centre crop / pad with -1 to (480, 480, 240), permute to (240, 480, 480)  (data.py:129-160)
// The host side uploads the raw int16 (or f32 / f64) voxels once -- a quarter of the bytes of the f32 volume the reference pipeline
// ships -- and gets the (1, D, H, W) f32 tensor in HBM.  Arithmetic in f64 like the reference (nibabel's get_fdata() is float64 and
// torch interpolates in the tensor's dtype), so the result equals the reference's up to the association order of the eight-corner
// blend before the final cast.
#include "common.h"

namespace {

struct PreParams {
  const void* src; int src_dtype;        // (H, W, D) voxels as stored by nibabel: 0 = int16, 1 = f32, 2 = f64
  int H, W, D;                           // source extents
  int rh, rw, rd;                        // resampled extents
  double sh, sw, sd;                     // source / resampled extent ratios (area_pixel_compute_scale, align_corners = False)
  double slope, intercept, lo, hi, inv_scale;
  float pad;
  int oh, ow, od;                        // output extents (480, 480, 240)
  int h0, w0, d0;                        // crop start in the resampled volume
  int ph, pw, pd;                        // padding before
  int ch, cw, cd;                        // cropped extents
  float* out;                            // (od, oh, ow)
};

__device__ __forceinline__ double fetch(const PreParams& p, int h, int w, int d) {
  const int64_t i = ((int64_t)h * p.W + w) * p.D + d;
  double v;
  if (p.src_dtype == 0) v = (double)reinterpret_cast<const int16_t*>(p.src)[i];
  else if (p.src_dtype == 1) v = (double)reinterpret_cast<const float*>(p.src)[i];
  else v = reinterpret_cast<const double*>(p.src)[i];
  return p.slope * v + p.intercept;
}

// torch's upsample index rule (UpSample.h area_pixel_compute_source_index, align_corners = False): src = scale * (dst + 0.5) - 0.5,
// clamped at 0; i0 = floor, i1 = min(i0 + 1, n - 1), lambda1 = src - i0
__device__ __forceinline__ void source_index(double scale, int dst, int n, int& i0, int& i1, double& l0, double& l1) {
  double s = scale * ((double)dst + 0.5) - 0.5;
  if (s < 0.0) s = 0.0;
  i0 = (int)s;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = s - (double)i0;
  l0 = 1.0 - l1;
}

// one thread per output voxel, ow fastest
__global__ __launch_bounds__(256) void preprocess_kernel(PreParams p) {
  const int64_t n = (int64_t)p.od * p.oh * p.ow;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int w = (int)(idx % p.ow), h = (int)((idx / p.ow) % p.oh), d = (int)(idx / ((int64_t)p.ow * p.oh));
    const int ch = h - p.ph, cw = w - p.pw, cd = d - p.pd;           // position in the cropped volume
    float r = p.pad;
    if (ch >= 0 && ch < p.ch && cw >= 0 && cw < p.cw && cd >= 0 && cd < p.cd) {
      int h_0, h_1, w_0, w_1, d_0, d_1;
      double hl0, hl1, wl0, wl1, dl0, dl1;
      // the reference interpolates the (D, H, W)-transposed array: output index order (d, h, w)
      source_index(p.sd, cd + p.d0, p.D, d_0, d_1, dl0, dl1);
      source_index(p.sh, ch + p.h0, p.H, h_0, h_1, hl0, hl1);
      source_index(p.sw, cw + p.w0, p.W, w_0, w_1, wl0, wl1);
      // UpSampleKernel.cpp cpu_upsample_linear / upsample_trilinear3d: t0 h0 w0, t0 h0 w1, t0 h1 w0, ... (d outermost, w innermost)
      const double v = dl0 * (hl0 * (wl0 * fetch(p, h_0, w_0, d_0) + wl1 * fetch(p, h_0, w_1, d_0)) +
                              hl1 * (wl0 * fetch(p, h_1, w_0, d_0) + wl1 * fetch(p, h_1, w_1, d_0))) +
                       dl1 * (hl0 * (wl0 * fetch(p, h_0, w_0, d_1) + wl1 * fetch(p, h_0, w_1, d_1)) +
                              hl1 * (wl0 * fetch(p, h_1, w_0, d_1) + wl1 * fetch(p, h_1, w_1, d_1)));
      const double c = v < p.lo ? p.lo : (v > p.hi ? p.hi : v);
      r = (float)(c * p.inv_scale);
    }
    p.out[idx] = r;
  }
}

}  // namespace

// data.py:92-162 without the file I/O.  src: the voxel array as nibabel hands it over, (H, W, D) row-major, src_dtype 0 = int16,
// 1 = f32, 2 = f64 (device memory).  xy_spacing / z_spacing: the volume's voxel size in mm (metadata columns XYSpacing, ZSpacing);
// target_*: 0.75, 0.75, 1.5.  out: (out_d, out_h, out_w) f32 = (240, 480, 480) in the reference.  hu_lo / hu_hi / hu_div: -1000,
// 1000, 1000; pad_value: -1.
extern "C" int ctclip_preprocess_volume(const void* src, int src_dtype, int H, int W, int D, double slope, double intercept, double xy_spacing,
                                        double z_spacing, double target_xy, double target_z, float* out, int out_h, int out_w, int out_d,
                                        double hu_lo, double hu_hi, double hu_div, float pad_value, hipStream_t stream) {
  if (!src || !out || H < 1 || W < 1 || D < 1 || src_dtype < 0 || src_dtype > 2 || out_h < 1 || out_w < 1 || out_d < 1 || !(xy_spacing > 0) ||
      !(z_spacing > 0) || !(target_xy > 0) || !(target_z > 0) || hu_div == 0) { ctclip_set_error("preprocess_volume: bad args"); return CTCLIP_EBADARG; }
  PreParams p{};
  p.src = src; p.src_dtype = src_dtype; p.H = H; p.W = W; p.D = D;
  // data.py:24-31: new size = int(size * (current / target)) per axis
  p.rd = (int)((double)D * (z_spacing / target_z)); p.rh = (int)((double)H * (xy_spacing / target_xy)); p.rw = (int)((double)W * (xy_spacing / target_xy));
  if (p.rd < 1 || p.rh < 1 || p.rw < 1) { ctclip_set_error("preprocess_volume: resampled volume is empty"); return CTCLIP_EBADARG; }
  p.sd = (double)D / p.rd; p.sh = (double)H / p.rh; p.sw = (double)W / p.rw;
  p.slope = slope; p.intercept = intercept; p.lo = hu_lo; p.hi = hu_hi; p.inv_scale = 1.0 / hu_div; p.pad = pad_value;
  p.oh = out_h; p.ow = out_w; p.od = out_d; p.out = out;
  // data.py:135-152: centre crop then centre pad
  auto crop = [](int n, int t, int& start, int& len, int& before) {
    start = (n - t) / 2 > 0 ? (n - t) / 2 : 0;
    const int end = start + t < n ? start + t : n;
    len = end - start;
    before = (t - len) / 2;
  };
  crop(p.rh, out_h, p.h0, p.ch, p.ph); crop(p.rw, out_w, p.w0, p.cw, p.pw); crop(p.rd, out_d, p.d0, p.cd, p.pd);
  const int64_t n = (int64_t)out_d * out_h * out_w;
  int64_t nb = cdiv(n, 256); if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p);
  return ctclip_check_launch("preprocess_volume");
}
