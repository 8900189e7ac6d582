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

// Workgroup = one output row (d, h fixed; blockIdx.y, blockIdx.z), thread = w: the d / h source indices and weights are uniform per
// workgroup (scalar registers), no 64-bit divisions; the two d-neighbours of a corner are adjacent in the source (d is its
// contiguous axis) and come in one access when the voxels are int16.  (First version: one flat index per thread, three 64-bit
// divisions and eight scalar gathers per voxel: 5.5 ms per volume.)
__global__ __launch_bounds__(256) void preprocess_kernel(PreParams p) {
  const int d = blockIdx.z, h = blockIdx.y;
  const int ch = h - p.ph, cd = d - p.pd;
  const bool row_in = ch >= 0 && ch < p.ch && cd >= 0 && cd < p.cd;
  int h_0 = 0, h_1 = 0, d_0 = 0, d_1 = 0;
  double hl0 = 0.0, hl1 = 0.0, dl0 = 0.0, dl1 = 0.0;
  if (row_in) {
    // the reference interpolates the (D, H, W)-transposed array: output index order (d, h, w)
    source_index(p.sd, cd + p.d0, p.D, d_0, d_1, dl0, dl1);
    source_index(p.sh, ch + p.h0, p.H, h_0, h_1, hl0, hl1);
  }
  float* orow = p.out + ((int64_t)d * p.oh + h) * p.ow;
  for (int w = blockIdx.x * 256 + threadIdx.x; w < p.ow; w += gridDim.x * 256) {
    const int cw = w - p.pw;
    float r = p.pad;
    if (row_in && cw >= 0 && cw < p.cw) {
      int w_0, w_1;
      double wl0, wl1;
      source_index(p.sw, cw + p.w0, p.W, w_0, w_1, wl0, wl1);
      // UpSampleKernel.cpp cpu_upsample_linear / upsample_trilinear3d: t0 h0 w0, t0 h0 w1, t0 h1 w0, ... (d outermost, w innermost)
      const double v = dl0 * (hl0 * (wl0 * fetch(p, h_0, w_0, d_0) + wl1 * fetch(p, h_0, w_1, d_0)) +
                              hl1 * (wl0 * fetch(p, h_1, w_0, d_0) + wl1 * fetch(p, h_1, w_1, d_0))) +
                       dl1 * (hl0 * (wl0 * fetch(p, h_0, w_0, d_1) + wl1 * fetch(p, h_0, w_1, d_1)) +
                              hl1 * (wl0 * fetch(p, h_1, w_0, d_1) + wl1 * fetch(p, h_1, w_1, d_1)));
      const double c = v < p.lo ? p.lo : (v > p.hi ? p.hi : v);
      r = (float)(c * p.inv_scale);
    }
    orow[w] = r;
  }
}

// int16 volumes (the stored form): a workgroup owns 16 output columns of one output row-plane (h fixed, all d) and first stages the
// source slab it needs -- rows h_0, h_1, source columns wlo .. whi, ALL of d (d is the source's contiguous axis: 600-byte runs,
// coalesced) -- in LDS; the eight taps of a voxel are then LDS reads, and 16 lanes write 64 contiguous bytes of an output row.
// The row kernel above reads 2 bytes out of every 600-byte source run per output row: 2.2 ms per volume, this one ~5x less.
__global__ __launch_bounds__(256) void preprocess_tile_kernel(PreParams p, int nws_max) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int16_t* slab = reinterpret_cast<int16_t*>(smem);                  // [2][nws_max][D]
  const int h = blockIdx.y, wt0 = blockIdx.x * 16;
  const int ch = h - p.ph;
  const bool row_in = ch >= 0 && ch < p.ch;
  const int cw_first = wt0 - p.pw > 0 ? wt0 - p.pw : 0;
  const int cw_last = wt0 + 15 - p.pw < p.cw - 1 ? wt0 + 15 - p.pw : p.cw - 1;
  const bool any = row_in && cw_first <= cw_last;
  int h_0 = 0, h_1 = 0, wlo = 0, nws = 0;
  double hl0 = 0.0, hl1 = 0.0;
  if (any) {
    int a0, a1, b0, b1; double t0, t1;
    source_index(p.sh, ch + p.h0, p.H, h_0, h_1, hl0, hl1);
    source_index(p.sw, cw_first + p.w0, p.W, a0, a1, t0, t1);
    source_index(p.sw, cw_last + p.w0, p.W, b0, b1, t0, t1);
    wlo = a0; nws = b1 - a0 + 1;
    // stage: 2 * nws runs of D voxels, as 32-bit words (D even, checked by the launcher)
    const int D2 = p.D >> 1, nwords = 2 * nws * D2;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.src);
    uint32_t* dst = reinterpret_cast<uint32_t*>(slab);
    for (int i = threadIdx.x; i < nwords; i += 256) {
      const int run = i / D2, col = i - run * D2;
      const int hh = run / nws, ws = run - hh * nws;
      dst[(hh * nws_max + ws) * D2 + col] = src[((int64_t)(hh ? h_1 : h_0) * p.W + wlo + ws) * D2 + col];
    }
  }
  __syncthreads();
  const int wl = threadIdx.x & 15, w = wt0 + wl, cw = w - p.pw;
  const bool col_in = any && w < p.ow && cw >= 0 && cw < p.cw;
  int w_0 = 0, w_1 = 0; double wl0 = 0.0, wl1 = 0.0;
  if (col_in) source_index(p.sw, cw + p.w0, p.W, w_0, w_1, wl0, wl1);
  const int16_t* r00 = slab + (0 * nws_max + (w_0 - wlo)) * p.D;     // (h_0, w_0), (h_0, w_1), (h_1, w_0), (h_1, w_1)
  const int16_t* r01 = slab + (0 * nws_max + (w_1 - wlo)) * p.D;
  const int16_t* r10 = slab + (1 * nws_max + (w_0 - wlo)) * p.D;
  const int16_t* r11 = slab + (1 * nws_max + (w_1 - wlo)) * p.D;
  if (w >= p.ow) return;
  for (int d = threadIdx.x >> 4; d < p.od; d += 16) {
    const int cd = d - p.pd;
    float r = p.pad;
    if (col_in && cd >= 0 && cd < p.cd) {
      int d_0, d_1; double dl0, dl1;
      source_index(p.sd, cd + p.d0, p.D, d_0, d_1, dl0, dl1);
      auto hu = [&](const int16_t* row, int dd) { return p.slope * (double)row[dd] + p.intercept; };
      const double v = dl0 * (hl0 * (wl0 * hu(r00, d_0) + wl1 * hu(r01, d_0)) + hl1 * (wl0 * hu(r10, d_0) + wl1 * hu(r11, d_0))) +
                       dl1 * (hl0 * (wl0 * hu(r00, d_1) + wl1 * hu(r01, d_1)) + hl1 * (wl0 * hu(r10, d_1) + wl1 * hu(r11, d_1)));
      const double c = v < p.lo ? p.lo : (v > p.hi ? p.hi : v);
      r = (float)(c * p.inv_scale);
    }
    p.out[((int64_t)d * p.oh + h) * p.ow + w] = r;
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
  if (out_h > 65535 || out_d > 65535) { ctclip_set_error("preprocess_volume: output extents above 65535"); return CTCLIP_EBADARG; }
  // tiled kernel: int16 voxels, even D, 4-byte aligned source, and a slab that fits in LDS
  const int nws_max = (int)(15.0 * p.sw) + 4;
  const size_t slab_bytes = (size_t)2 * nws_max * D * 2;
  if (src_dtype == 0 && D % 2 == 0 && (reinterpret_cast<uintptr_t>(src) % 4) == 0 && slab_bytes <= 150 * 1024) {
    static size_t raised = 0;
    if (slab_bytes > 64 * 1024 && slab_bytes > raised) {
      if (hipFuncSetAttribute((const void*)preprocess_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { (void)hipGetLastError(); goto rows; }
      raised = 150 * 1024;
    }
    hipLaunchKernelGGL(preprocess_tile_kernel, dim3((unsigned)cdiv(out_w, 16), (unsigned)out_h), dim3(256), slab_bytes, stream, p, nws_max);
    return ctclip_check_launch("preprocess_volume");
  }
rows:
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)cdiv(out_w, 256), (unsigned)out_h, (unsigned)out_d), dim3(256), 0, stream, p);
  return ctclip_check_launch("preprocess_volume");
}
