// Row statistics kernels: LayerNorm forward/backward, patch gather + LayerNorm (CTViT patch embedding
// front-end), l2 row normalisation.  All HBM-bound: one wave64 per row, 16-byte vector accesses,
// statistics in f32 with wave shuffles (no atomics: deterministic).
#include "common.h"

namespace {

#ifndef PATCH_LN_XCD
#define PATCH_LN_XCD 1     // 0 = token = workgroup index (A/B builds)
#endif
#ifndef LN_FWD_RU
#define LN_FWD_RU 8     // rows in flight per wave, forward (round 6: 4 -> 8 once the reductions left the LDS crossbar: 44.3 -> 41.4 us isolated)
#endif
#ifndef LN_BWD_RU
#define LN_BWD_RU 4     // rows in flight per wave, backward (round 6: 2 -> 4: 82 -> 74 us plain, 118 -> 110.5 with two addends, isolated)
#endif
constexpr int LN_MAXV = 4;  // up to 64 lanes * 8 * 4 = 2048 columns per row

// One wave64 per row, several rows per wave (grid-stride).  NV = ceil(cols / 512) is a compile-time constant so that the
// row's loads are unconditional and issued together (columns past `cols` are clamped and masked).  RU rows are loaded before any of
// them is reduced: with one 1-KB row in flight per wave the kernel sat at 4.3 TB/s (bytes in flight x 256 CUs / latency).
template <typename T, int NV, int RU>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  int cc[NV]; bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { const int c = (i * 64 + lane) * 8; ok[i] = c < cols; cc[i] = ok[i] ? c : cols - 8; }
  float gm[NV][8], bt[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[i][e] = gamma ? gamma[cc[i] + e] : 1.f; bt[i][e] = beta ? beta[cc[i] + e] : 0.f; }
  for (int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RU; row0 < rows; row0 += nwaves * RU) {
    float v[RU][NV][8];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u < rows ? row0 + u : rows - 1;      // clamped, never branch around a load
#pragma unroll
      for (int i = 0; i < NV; ++i) load8(x + row * cols + cc[i], v[u][i]);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += ok[i] ? v[u][i][e] : 0.f;
      const float mean = wave_sum(s) / cols;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[u][i][e] - mean; q += ok[i] ? d * d : 0.f; }
      const float rstd = rsqrtf(wave_sum(q) / cols + eps);
      if (row < rows) {
        if (lane == 0) {
          if (mean_out) mean_out[row] = mean;
          if (rstd_out) rstd_out[row] = rstd;
        }
        T* yr = y + row * cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[u][i][e] - mean) * rstd * gm[i][e] + bt[i][e];
          if (ok[i]) store8(yr + cc[i], o);
        }
      }
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ add1) (+ add2),  g = dy * gamma.
// add1 / add2: gradients that reach the same activation through its other consumers (the residual connection, the k/v projection
// of the raw tokens): summed here instead of by two elementwise kernels over the 113-MB tensor.
// Per-block partial sums of dgamma = sum dy*xhat and dbeta = sum dy are written to part[block][2][cols].
template <typename T, int NV, int RU>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, T* __restrict__ dx,
                                                            float* __restrict__ part, int64_t rows, int cols,
                                                            const T* __restrict__ add1, const T* __restrict__ add2) {
  // RU rows per wave in flight, and the add operands requested TOGETHER with dy / x (they used to be requested behind the two wave
  // reductions: a second memory round trip per row).  tools/ubench/stream_rates.hip, cold operands: 109 -> 104 us with two adds.
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4 waves][2][cols]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cc[NV]; bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { const int c = (i * 64 + lane) * 8; ok[i] = c < cols; cc[i] = ok[i] ? c : cols - 8; }
  float gm[NV][8], dg[NV][8], db[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[i][e] = gamma ? gamma[cc[i] + e] : 1.f; dg[i][e] = 0.f; db[i][e] = 0.f; }
  for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RU; row0 < rows; row0 += (int64_t)gridDim.x * 4 * RU) {
    float a[RU][NV][8], b[RU][NV][8], r1[RU][NV][8], r2[RU][NV][8], mu[RU], rs[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u < rows ? row0 + u : rows - 1;      // clamped, never branch around a load
      mu[u] = mean[row]; rs[u] = rstd[row];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        load8(dy + row * cols + cc[i], a[u][i]); load8(x + row * cols + cc[i], b[u][i]);
        if (add1) load8(add1 + row * cols + cc[i], r1[u][i]);       // kernel-uniform
        if (add2) load8(add2 + row * cols + cc[i], r2[u][i]);
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const bool live = row0 + u < rows;
      const int64_t row = live ? row0 + u : rows - 1;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (b[u][i][e] - mu[u]) * rs[u];
          const float dyv = (ok[i] && live) ? a[u][i][e] : 0.f;
          const float g = dyv * gm[i][e];
          b[u][i][e] = xh; a[u][i][e] = g;
          s1 += g; s2 += g * xh;
          dg[i][e] += dyv * xh; db[i][e] += dyv;
        }
      s1 = wave_sum(s1) / cols;
      s2 = wave_sum(s2) / cols;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs[u] * (a[u][i][e] - s1 - b[u][i][e] * s2);
        if (add1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += r1[u][i][e];
        }
        if (add2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += r2[u][i][e];
        }
        if (ok[i] && live) store8(dx + row * cols + cc[i], o);
      }
    }
  }
  if (!part) return;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (ok[i]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sm[(wave * 2 + 0) * cols + cc[i] + e] = dg[i][e]; sm[(wave * 2 + 1) * cols + cc[i] + e] = db[i][e]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * cols; c += 256) {
    const int which = c / cols, col = c % cols;
    float t = 0.f;
    for (int w = 0; w < 4; ++w) t += sm[(w * 2 + which) * cols + col];
    part[((int64_t)blockIdx.x * 2 + which) * cols + col] = t;
  }
}

// out[c] += sum_b part[b][which][c].  One workgroup of 1024 threads per 64 columns: sixteen interleaved slices of the partial rows are
// summed in parallel and combined in a fixed order (deterministic: the previous version combined chunks with f32 atomics; the
// first version walked all 1024 partial rows with 16 workgroups of 64 threads: 58 us of pure latency, 76 times per step).
__global__ __launch_bounds__(1024) void ln_partial_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int nblocks, int cols) {
  __shared__ float red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  float t = 0.f;
  if (c < 2 * cols) {
    const int which = c / cols, col = c % cols;
#pragma unroll 16
    for (int b = sl; b < nblocks; b += 16) t += part[((int64_t)b * 2 + which) * cols + col];      // (64 rows per thread: 16 loads in flight, added in row order)
  }
  red[sl][threadIdx.x & 63] = t;
  __syncthreads();
  if (sl == 0 && c < 2 * cols) {
    const int which = c / cols, col = c % cols;
    float* dst = which == 0 ? dgamma : dbeta;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += red[i][threadIdx.x];
    if (dst) dst[col] += a;
  }
}

// CTViT patch embedding front-end (ctvit.py:171-172): gather the (pt, p1, p2) patch of one token from the
// (B,1,F,H,W) f32 volume in 16-byte row segments, LayerNorm over its pt*p1*p2 values (eps 1e-5, affine
// folded into the following GEMM's weights by the host), write xhat as a K-padded row of the embed GEMM's A.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void patch_ln_kernel(const float* __restrict__ video, T* __restrict__ out, int F, int H,
                                                       int W, int pt, int p1, int p2, int kpad, float eps) {
  __shared__ float red[16];
  const int t = F / pt, h = H / p1, w = W / p2;
  const int K = pt * p1 * p2;
  // Workgroups go to the 8 XCDs round-robin.  The w tokens of one grid row read the same 128-byte lines of the volume (a patch row is p2 floats
  // = 80 bytes at CT-CLIP's 20 x 20 x 10 patches: neither aligned to nor a multiple of a line), so a whole row of tokens is dealt to ONE XCD:
  // every line is then fetched into one L2 only.
  int64_t tok = blockIdx.x;
#if PATCH_LN_XCD
  if ((gridDim.x / w) % 8 == 0) {
    const int64_t r = tok >> 3;
    tok = ((r / w) * 8 + (tok & 7)) * w + r % w;
  }
#endif
  const int j = tok % w; const int i = (tok / w) % h; const int tau = (tok / ((int64_t)w * h)) % t; const int64_t b = tok / ((int64_t)w * h * t);
  const float* vb = video + b * (int64_t)F * H * W;
  constexpr int NV = 4;  // up to 256 * 4 * 4 = 4096 patch elements
  float v[NV][4];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int e = (q * 256 + threadIdx.x) * 4;
    if (e < K) {
      if constexpr (VEC) {
        const int a = e / (p1 * p2), u = (e / p2) % p1, vv = e % p2;
        load4(vb + ((int64_t)(tau * pt + a) * H + (i * p1 + u)) * W + j * p2 + vv, v[q]);
      } else {
#pragma unroll
        for (int z = 0; z < 4; ++z) {
          const int ee = e + z;
          const int a = ee / (p1 * p2), u = (ee / p2) % p1, vv = ee % p2;
          v[q][z] = ee < K ? vb[((int64_t)(tau * pt + a) * H + (i * p1 + u)) * W + j * p2 + vv] : 0.f;
        }
      }
#pragma unroll
      for (int z = 0; z < 4; ++z) s += v[q][z];
    }
  }
  const float mean = block_sum(s, red) / K;
  float qv = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int e = (q * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int z = 0; z < 4; ++z)
      if (e + z < K) { const float d = v[q][z] - mean; qv += d * d; }
  }
  const float rstd = rsqrtf(block_sum(qv, red) / K + eps);
  T* o = out + tok * (int64_t)kpad;
#pragma unroll
  for (int q = 0; q < NV + 1; ++q) {
    const int e = (q * 256 + threadIdx.x) * 4;
    if (e < kpad) {
      float r[4];
#pragma unroll
      for (int z = 0; z < 4; ++z) r[z] = (q < NV && e + z < K) ? (v[q < NV ? q : 0][z] - mean) * rstd : 0.f;
      store4(o + e, r);
    }
  }
}

// y = x / max(||x||_2, eps) per row (F.normalize); optional inverse norm output
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const TI* __restrict__ x, TO* __restrict__ y, float* __restrict__ inv_out,
                                                          int64_t rows, int cols, int64_t ldx, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[LN_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
      load8(x + row * ldx + c, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e] * v[i][e];
    }
  }
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), eps);
  if (lane == 0 && inv_out) inv_out[row] = inv;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[i][e] * inv;
      store8(y + row * cols + c, o);
    }
  }
}

// Unit rows as a THREE-TERM bf16 expansion for an f32-grade dot product on the bf16 matrix cores: y = [hi | hi | lo] (order 0) or
// [hi | lo | hi] (order 1) with hi = bf16(x^), lo = bf16(x^ - hi), x^ = x / max(||x||, eps) in f32.  A row of order 0 times a row of
// order 1 is  hi.hi' + hi.lo' + lo.hi'  = x^ . x^'  up to the dropped lo.lo' term (~2^-17 of the summands).
// order 2: y = [hi | lo] (rows of 2 cols: the codebook operand of ctclip_gemm_argmax_hilo); y == NULL: only the
// inverse norms (the quantiser's EMA statistics need them when the search reads the raw tokens).
template <typename TI>
__global__ __launch_bounds__(256) void l2norm_split3_kernel(const TI* __restrict__ x, bf16_t* __restrict__ y, float* __restrict__ inv_out,
                                                            int64_t rows, int cols, int64_t ldx, float eps, int order) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[LN_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
      load8(x + row * ldx + c, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e] * v[i][e];
    }
  }
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), eps);
  if (lane == 0 && inv_out) inv_out[row] = inv;
  if (!y) return;                                            // kernel-uniform
  bf16_t* o = y + row * (order == 2 ? 2 : 3) * (int64_t)cols;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
      float hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = v[i][e] * inv;
        hi[e] = bf2f(f2bf(t));
        lo[e] = t - hi[e];
      }
      store8(o + c, hi);
      store8(o + cols + c, order == 0 ? hi : lo);
      if (order != 2) store8(o + 2 * cols + c, order == 0 ? lo : hi);
    }
  }
}

}  // namespace

// F.layer_norm / nn.LayerNorm forward (attention.py:28-35,47; ctvit.py:174; HF BertLayerNorm).
extern "C" int ctclip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    int64_t rows, int cols, float eps, int dtype, hipStream_t stream) {
  if (!x || !y || rows <= 0 || cols <= 0 || cols % 8 || cols > 64 * 8 * LN_MAXV) { ctclip_set_error("layernorm_fwd: cols must be a multiple of 8 and <= 2048"); return CTCLIP_EBADARG; }
  const int nv = (cols + 511) / 512;
  const int ru = (nv == 1 && rows >= 65536) ? LN_FWD_RU : 1;      // rows in flight per wave (big token grids of the image tower)
  int64_t nb = cdiv(rows, 8 * ru); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb);
#define LNF(T, NVV, RUU) hipLaunchKernelGGL((layernorm_fwd_kernel<T, NVV, RUU>), grid, dim3(256), 0, stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, cols, eps)
#define LNF_NV(T) do { if (nv == 1) { if (ru > 1) LNF(T, 1, LN_FWD_RU); else LNF(T, 1, 1); } else if (nv == 2) LNF(T, 2, 1); else if (nv == 3) LNF(T, 3, 1); else LNF(T, 4, 1); } while (0)
  if (dtype == DT_F32) LNF_NV(float);
  else if (dtype == DT_BF16) LNF_NV(bf16_t);
  else return CTCLIP_EUNSUPPORTED;
#undef LNF
#undef LNF_NV
  return ctclip_check_launch("layernorm_fwd");
}

#ifndef LN_BWD_BLOCKS
#define LN_BWD_BLOCKS 512      // workgroups (= partial rows of the dgamma / dbeta fold) of the backward on big grids (two per CU: 1024 / 512 / 2048 measured 73.4 / 72.9 / 79.8 us plain, 112.7 / 110.4 / 118.5 with two addends)
#endif
static int64_t ln_bwd_blocks(int64_t rows) { int64_t nb = cdiv(rows, 8); if (nb > LN_BWD_BLOCKS) nb = LN_BWD_BLOCKS; return nb < 1 ? 1 : nb; }
extern "C" int64_t ctclip_layernorm_bwd_workspace(int64_t rows, int cols) { return ln_bwd_blocks(rows) * 2 * cols * 4; }

// LayerNorm backward, first half: dx (+ add1 + add2: optional same-shape gradients of the input's other consumers, see the kernel) and, when
// `partials` is given (ctclip_layernorm_bwd_workspace bytes), the per-workgroup partial sums of dgamma / dbeta.  The second half
// (ctclip_layernorm_bwd_reduce) folds them; a caller may launch it on ANOTHER stream: dgamma / dbeta are leaves of the backward graph, the 13-us
// fold of 1 024 partial rows by 16 workgroups is pure latency on the stream that carries the grad-input chain (76 times per step).
extern "C" int ctclip_layernorm_bwd_partials(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                             void* dx, const void* add1, const void* add2, int64_t rows, int cols, int dtype, void* partials,
                                             int64_t partials_bytes, hipStream_t stream) {
  if (!dy || !x || !dx || !mean || !rstd || cols % 8 || cols > 64 * 8 * LN_MAXV) { ctclip_set_error("layernorm_bwd: bad args"); return CTCLIP_EBADARG; }
  const int64_t nb = ln_bwd_blocks(rows);
  if (partials && partials_bytes < ctclip_layernorm_bwd_workspace(rows, cols)) { ctclip_set_error("layernorm_bwd: workspace too small"); return CTCLIP_EWORKSPACE; }
  float* part = (float*)partials;
  const size_t shm = (size_t)4 * 2 * cols * sizeof(float);
  const int nv = (cols + 511) / 512;
  const bool two = nv == 1 && rows >= 65536;        // LN_BWD_RU rows in flight per wave on the big token grids of the image tower
#define LNB(T, NVV, RUU) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NVV, RUU>), dim3((unsigned)nb), dim3(256), shm, stream, (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, part, rows, cols, (const T*)add1, (const T*)add2)
#define LNB_NV(T) do { if (nv == 1) { if (two) LNB(T, 1, LN_BWD_RU); else LNB(T, 1, 1); } else if (nv == 2) LNB(T, 2, 1); else if (nv == 3) LNB(T, 3, 1); else LNB(T, 4, 1); } while (0)
  if (dtype == DT_F32) LNB_NV(float);
  else if (dtype == DT_BF16) LNB_NV(bf16_t);
  else return CTCLIP_EUNSUPPORTED;
#undef LNB
#undef LNB_NV
  return ctclip_check_launch("layernorm_bwd");
}

// LayerNorm backward, second half: dgamma / dbeta (either may be null) += the fixed-order sum of the partial rows written by
// ctclip_layernorm_bwd_partials for the same (rows, cols).
extern "C" int ctclip_layernorm_bwd_reduce(const void* partials, float* dgamma, float* dbeta, int64_t rows, int cols, hipStream_t stream) {
  if (!partials || cols % 8) { ctclip_set_error("layernorm_bwd_reduce: bad args"); return CTCLIP_EBADARG; }
  if (!dgamma && !dbeta) return CTCLIP_OK;
  hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((unsigned)cdiv(2 * cols, 64)), dim3(1024), 0, stream, (const float*)partials, dgamma, dbeta,
                     (int)ln_bwd_blocks(rows), cols);
  return ctclip_check_launch("ln_partial_reduce");
}

// LayerNorm backward (both halves on one stream): dx (+ add1 + add2), and dgamma / dbeta ACCUMULATED (+=) into f32 buffers (either may be null).
extern "C" int ctclip_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                    void* dx, float* dgamma, float* dbeta, const void* add1, const void* add2, int64_t rows, int cols,
                                    int dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  const bool want = dgamma || dbeta;
  if (want && (!workspace || workspace_bytes < ctclip_layernorm_bwd_workspace(rows, cols))) { ctclip_set_error("layernorm_bwd: workspace too small"); return CTCLIP_EWORKSPACE; }
  int rc = ctclip_layernorm_bwd_partials(dy, x, gamma, mean, rstd, dx, add1, add2, rows, cols, dtype, want ? workspace : nullptr, workspace_bytes, stream);
  if (rc || !want) return rc;
  return ctclip_layernorm_bwd_reduce(workspace, dgamma, dbeta, rows, cols, stream);
}

// CTViT.to_patch_emb[0:2] (ctvit.py:171-172): Rearrange + LayerNorm statistics; out is (B*t*h*w, kpad).
extern "C" int ctclip_patch_ln_fwd(const float* video, void* out, int64_t B, int F, int H, int W, int pt, int p1, int p2,
                                   int kpad, float eps, int out_dtype, hipStream_t stream) {
  const int K = pt * p1 * p2;
  if (!video || !out || F % pt || H % p1 || W % p2 || K > 4096 || kpad < K || kpad % 4 || kpad > 5 * 1024) { ctclip_set_error("patch_ln_fwd: unsupported patch geometry (pt*p1*p2 <= 4096)"); return CTCLIP_EUNSUPPORTED; }
  const int64_t ntok = B * (F / pt) * (H / p1) * (W / p2);
  const bool vec = (p2 % 4 == 0) && (W % 4 == 0) && (((uintptr_t)video) % 16 == 0) && (K % 4 == 0);
  dim3 grid((unsigned)ntok);
#define LAUNCH(T, V) hipLaunchKernelGGL((patch_ln_kernel<T, V>), grid, dim3(256), 0, stream, video, (T*)out, F, H, W, pt, p1, p2, kpad, eps)
  if (out_dtype == DT_F32) { if (vec) LAUNCH(float, true); else LAUNCH(float, false); }
  else if (out_dtype == DT_BF16) { if (vec) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false); }
  else return CTCLIP_EUNSUPPORTED;
#undef LAUNCH
  return ctclip_check_launch("patch_ln_fwd");
}

// F.normalize(x, dim=-1) rows (attention.py:22-23, ct_clip.py:49-50, VQ codebook l2norm).  in_dtype -> out_dtype.
extern "C" int ctclip_l2norm_rows(const void* x, void* y, float* inv, int64_t rows, int cols, int64_t ldx, float eps, int in_dtype,
                                  int out_dtype, hipStream_t stream) {
  if (!x || !y || cols % 8 || cols > 64 * 8 * LN_MAXV || ldx % 8) { ctclip_set_error("l2norm_rows: cols must be a multiple of 8 and <= 2048"); return CTCLIP_EBADARG; }
  dim3 grid((unsigned)cdiv(rows, 4));
#define LAUNCH(TI, TO) hipLaunchKernelGGL((l2norm_rows_kernel<TI, TO>), grid, dim3(256), 0, stream, (const TI*)x, (TO*)y, inv, rows, cols, ldx, eps)
  if (in_dtype == DT_F32 && out_dtype == DT_F32) LAUNCH(float, float);
  else if (in_dtype == DT_F32 && out_dtype == DT_BF16) LAUNCH(float, bf16_t);
  else if (in_dtype == DT_BF16 && out_dtype == DT_BF16) LAUNCH(bf16_t, bf16_t);
  else if (in_dtype == DT_BF16 && out_dtype == DT_F32) LAUNCH(bf16_t, float);
  else return CTCLIP_EUNSUPPORTED;
#undef LAUNCH
  return ctclip_check_launch("l2norm_rows");
}

// l2-normalised rows as the bf16 pair expansion the vector-quantiser code search multiplies (vector_quantize_pytorch 1.1.2
// CosineSimCodebook.forward: `flatten = l2norm(x.float())`, `embed = l2norm(self.embed)`, `dist = einsum(flatten, embed)` in f32).
// y: (rows, 3 * cols) bf16; order 0 = [hi | hi | lo] (tokens), order 1 = [hi | lo | hi] (codes); inv (rows) f32 optional.
extern "C" int ctclip_l2norm_split3(const void* x, void* y, float* inv, int64_t rows, int cols, int64_t ldx, float eps, int in_dtype,
                                    int order, hipStream_t stream) {
  if (!x || (!y && !inv) || cols % 8 || cols > 64 * 8 * LN_MAXV || ldx % 8 || order < 0 || order > 2) { ctclip_set_error("l2norm_split3: cols must be a multiple of 8 and <= 2048, order 0, 1 or 2"); return CTCLIP_EBADARG; }
  dim3 grid((unsigned)cdiv(rows, 4));
  if (in_dtype == DT_F32) hipLaunchKernelGGL(l2norm_split3_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (bf16_t*)y, inv, rows, cols, ldx, eps, order);
  else if (in_dtype == DT_BF16) hipLaunchKernelGGL(l2norm_split3_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, inv, rows, cols, ldx, eps, order);
  else return CTCLIP_EUNSUPPORTED;
  return ctclip_check_launch("l2norm_split3");
}
