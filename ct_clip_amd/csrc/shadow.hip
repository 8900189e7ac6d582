// Batched refresh of the compute-dtype weight shadows: after the optimiser step ONE launch rebuilds every bf16 operand the GEMMs read
// (plain / zero-padded copies, the padded [x | gate] and the four-row interleaved forms of the GEGLU in-projection, the stacked q|k|v
// weight of a BERT layer, and the transposed forms the grad-input GEMMs use) from the f32 master weights, in place.
// Before: ~370 launches of 3-6 us per step (convert_pad, transpose2d, geglu_weight_interleave), each one created lazily in front of the
// GEMM that needed it -- launch-latency bound, on the critical path of the step.
//
// A job describes one destination matrix dst (dst_rows x dst_cols, row stride dst_ld, bf16) as a gather from one f32 source
// (src_rows x src_cols, row stride src_ld):
//   plain:       dst[r][c] = src[map(r)][c]            transposed:  dst[r][c] = src[map(c)][r]        (0 outside the source)
//   map 0: n -> n;   map 1 (GEGLU split, aux = inner, hp = (mapped extent) / 2): n < hp -> n (n < inner), n >= hp -> inner + n - hp
//   (n - hp < inner);   map 2 (GEGLU interleave, aux = inner): n -> part * inner + j with j = 4 (n >> 3) + (n & 3), part = (n >> 2) & 1
//   (j < inner)  [= ctclip_geglu_weight_interleave].
// Work unit = one 64 x 64 tile of dst; a job's tiles are [tile0, tile0 + tiles); the block finds its job by bisection.
#include "common.h"

namespace {

struct ShadowJob {     // 12 x int64 (mirrored by ct_clip_amd/functional.py)
  int64_t src, dst, src_ld, dst_ld, src_rows, src_cols, dst_rows, dst_cols, map, aux, transposed, tile0;
};

__device__ __forceinline__ int64_t shadow_map(int map, int64_t n, int64_t mapped_extent, int64_t src_rows, int64_t aux) {
  if (map == 0) return n < src_rows ? n : -1;
  if (map == 1) {
    const int64_t hp = mapped_extent >> 1;
    if (n < hp) return n < aux ? n : -1;
    return (n - hp < aux && n < mapped_extent) ? aux + n - hp : -1;
  }
  const int64_t j = 4 * (n >> 3) + (n & 3), part = (n >> 2) & 1;
  return j < aux ? part * aux + j : -1;
}

__global__ __launch_bounds__(256) void shadow_refresh_kernel(const ShadowJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[64][65];
  // bisection for the job that owns this tile (jobs sorted by tile0, jobs[0].tile0 == 0)
  int lo = 0, hi = njobs - 1;
  const int64_t id = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= id) lo = mid; else hi = mid - 1;
  }
  const ShadowJob j = jobs[lo];
  const int64_t tcols = (j.dst_cols + 63) >> 6;
  const int64_t t = id - j.tile0;
  const int64_t r0 = (t / tcols) << 6, c0 = (t % tcols) << 6;
  const float* src = reinterpret_cast<const float*>(j.src);
  bf16_t* dst = reinterpret_cast<bf16_t*>(j.dst);
  const int tid = threadIdx.x;
  if (!j.transposed) {
    // 16 lanes x 4 consecutive columns per row, 16 rows per pass
    const int64_t c = c0 + (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + (tid >> 4) + 16 * i;
      if (r >= j.dst_rows || c >= j.dst_cols) continue;
      const int64_t sr = shadow_map((int)j.map, r, j.dst_rows, j.src_rows, j.aux);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (sr >= 0) {
        const float* sp = src + sr * j.src_ld + c;
        if (c + 4 <= j.src_cols && ((reinterpret_cast<uintptr_t>(sp) & 15) == 0)) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(sp);
          v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < j.src_cols) v[e] = sp[e];
        }
      }
      bf16_t* d = dst + r * j.dst_ld + c;
      if (c + 4 <= j.dst_cols && ((reinterpret_cast<uintptr_t>(d) & 7) == 0)) {
        *reinterpret_cast<u32x2*>(d) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < j.dst_cols) d[e] = f2bf(v[e]);
      }
    }
  } else {
    // dst rows = source columns k, dst columns = mapped source rows: stage 64 source rows x 64 k through LDS
    const int lane = tid & 63, w = tid >> 6;
    for (int cc = w; cc < 64; cc += 4) {
      const int64_t n = c0 + cc;
      const int64_t sr = n < j.dst_cols ? shadow_map((int)j.map, n, j.dst_cols, j.src_rows, j.aux) : -1;
      const int64_t k = r0 + lane;
      tile[cc][lane] = (sr >= 0 && k < j.src_cols) ? src[sr * j.src_ld + k] : 0.f;
    }
    __syncthreads();
    for (int kk = w; kk < 64; kk += 4) {
      const int64_t r = r0 + kk, c = c0 + lane;
      if (r < j.dst_rows && c < j.dst_cols) dst[r * j.dst_ld + c] = f2bf(tile[lane][kk]);
    }
  }
}

}  // namespace

// jobs: device array of njobs records of 12 int64 {src, dst, src_ld, dst_ld, src_rows, src_cols, dst_rows, dst_cols, map, aux,
// transposed, tile0}, sorted by tile0 (tile0 of job i = sum of ceil(dst_rows / 64) * ceil(dst_cols / 64) over the jobs before it);
// ntiles = the total.  src f32, dst bf16.  Replaces the per-parameter `.to(bf16)` / `.t().contiguous()` copies a torch module would make.
extern "C" int ctclip_shadow_refresh(const void* jobs, int njobs, int64_t ntiles, hipStream_t stream) {
  if (!jobs || njobs < 1 || ntiles < 1 || ntiles > 0x7fffffff) { ctclip_set_error("shadow_refresh: bad args"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(shadow_refresh_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, (const ShadowJob*)jobs, njobs);
  return ctclip_check_launch("shadow_refresh");
}
