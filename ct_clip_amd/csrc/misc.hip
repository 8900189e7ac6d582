// Streaming (HBM-bound) kernels of the CT-CLIP hot path: GEGLU / GELU, bias-gradient column sums, token-grid
// permutation, depth mean-pool, dtype conversion with padding (weight shadows), continuous-position-bias
// gather/scatter, BERT embedding gather / scatter-add, vector-quantiser gather and EMA statistics.
// All use 16-byte accesses (8 x bf16 / 2 x 4 x f32 per lane) and grid-stride loops.
#include "common.h"

namespace {

constexpr int64_t MAXBLOCKS = 256 * 16;
inline dim3 grid_for(int64_t nthreads) { int64_t b = cdiv(nthreads, 256); if (b > MAXBLOCKS) b = MAXBLOCKS; if (b < 1) b = 1; return dim3((unsigned)b); }

// ---- GEGLU (attention.py:39-42): u = [x | gate] (M, 2*Hp);  g = x * gelu(gate)
template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ u, T* __restrict__ g, int64_t M, int Hp) {
  const int G = Hp / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * G; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / G; const int c = (int)(i % G) * 8;
    float a[8], b[8], o[8];
    load8(u + row * 2 * Hp + c, a);
    load8(u + row * 2 * Hp + Hp + c, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] * gelu_erf(b[e]);
    store8(g + row * Hp + c, o);
  }
}
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ u, T* __restrict__ du, int64_t M, int Hp) {
  const int G = Hp / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * G; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / G; const int c = (int)(i % G) * 8;
    float a[8], b[8], d[8], o1[8], o2[8];
    load8(u + row * 2 * Hp + c, a);
    load8(u + row * 2 * Hp + Hp + c, b);
    load8(dg + row * Hp + c, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) { o1[e] = d[e] * gelu_erf(b[e]); o2[e] = d[e] * a[e] * gelu_erf_grad(b[e]); }
    store8(du + row * 2 * Hp + c, o1);
    store8(du + row * 2 * Hp + Hp + c, o2);
  }
}
// ---- GELU (HF BertIntermediate, gelu-erf)
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ h, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float a[8], o[8];
    load8(u + i * 8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = gelu_erf(a[e]);
    store8(h + i * 8, o);
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ u, T* __restrict__ du, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float a[8], d[8], o[8];
    load8(u + i * 8, a);
    load8(dh + i * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = d[e] * gelu_erf_grad(a[e]);
    store8(du + i * 8, o);
  }
}
// ---- LeakyReLU (attention.py:19-20), f32 only (position-bias MLP)
__global__ void leaky_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const float v = x[i]; y[i] = v > 0.f ? v : v * slope; }
}
__global__ void leaky_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dx[i] = x[i] > 0.f ? dy[i] : dy[i] * slope;
}

// ---- out[n] += sum_m x[m][n]   (bias gradients).  block: 32 column-groups of 8 x 8 row lanes; 512 rows per block
// ROWS rows per workgroup (eight row lanes x ROWS / 8 steps).  Small M (the text tower: M = B T = 1024, 73 bias gradients per step) takes
// ROWS = 64: sixteen row blocks instead of two, and a thread's eight loads are all in flight (round 3: 512 rows = 64 dependent-latency
// iterations on 6-24 workgroups, 40 us per launch for a 3-MB read).
template <typename T, int ROWS>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ part, int64_t M, int N, int64_t ld) {
  __shared__ float red[8][256];
  const int cgp = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.y * 32 + cgp) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c < N) {
    const int64_t r0 = (int64_t)blockIdx.x * ROWS;
#pragma unroll 8
    for (int i = 0; i < ROWS / 8; ++i) {
      int64_t r = r0 + rl + 8 * i;
      const bool ok = r < M;
      r = ok ? r : M - 1;                                  // (clamped, masked below: never branch around a load)
      float v[8];
      load8(x + r * ld + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += ok ? v[e] : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cgp * 8 + e] = acc[e];
  __syncthreads();
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
    part[(int64_t)blockIdx.x * N + col] = t;          // per-row-block partial: summed in block order by colsum_reduce_kernel
  }
}
// out[col] += sum over row blocks (in order) of part[block][col]
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int nblk, int N, float* __restrict__ out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  float t = 0.f;
  int b = 0;
  for (; b + 8 <= nblk; b += 8) {                            // eight partials in flight, added in block order
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(int64_t)(b + k) * N + col];
#pragma unroll
    for (int k = 0; k < 8; ++k) t += v[k];
  }
  for (; b < nblk; ++b) t += part[(int64_t)b * N + col];
  out[col] += t;
}

// generic fallback (any N / pitch): one thread per column, 64 rows per block row-lane
template <typename T>
__global__ __launch_bounds__(256) void colsum_generic_kernel(const T* __restrict__ x, float* __restrict__ part, int64_t M, int N, int64_t ld) {
  __shared__ float red[4][64];
  const int col = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float acc = 0.f;
  const int64_t r0 = (int64_t)blockIdx.x * 1024;
  if (col < N)
    for (int64_t r = r0 + rl; r < r0 + 1024 && r < M; r += 4) acc += Elem<T>::ld(x + r * ld + col);
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && col < N) part[(int64_t)blockIdx.x * N + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- (A, B, C, D) -> (A, C, B, D)   (ctvit.py:297-305 rearranges between the spatial and temporal phases)
template <typename T>
__global__ void permute0213_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t A, int B, int C, int D) {
  const int G = D / 8;
  const int64_t n = A * B * C * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % G); int64_t r = i / G;
    const int b = (int)(r % B); r /= B;
    const int c = (int)(r % C); const int64_t a = r / C;   // output index (a, c, b)
    float v[8];
    load8(x + ((a * B + b) * C + c) * D + g * 8, v);
    store8(y + ((a * C + c) * B + b) * D + g * 8, v);
  }
}

// ---- y (C, R) = x (R, C)^T  (weight-shadow transposes for the grad-input GEMMs), 32 x 32 LDS tiles
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* __restrict__ x, T* __restrict__ y, int R, int C, int64_t ldx, int64_t ldy) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? Elem<T>::ld(x + (int64_t)r * ldx + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) Elem<T>::st(y + (int64_t)c * ldy + r, tile[tx][i]);
  }
}

// ---- mean over the depth-token axis (ct_clip.py:724): x (B, t, R) -> y (B, R); backward broadcasts dy / t
template <typename T, typename TO>
__global__ void pool_fwd_kernel(const T* __restrict__ x, TO* __restrict__ y, int64_t B, int t, int64_t R) {
  const int64_t G = R / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B * G; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / G, c = (i % G) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int tau = 0; tau < t; ++tau) {
      float v[8];
      load8(x + (b * t + tau) * R + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
    const float inv = 1.f / t;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    store8(y + b * R + c, acc);
  }
}
template <typename T, typename TO>
__global__ void pool_bwd_kernel(const T* __restrict__ dy, TO* __restrict__ dx, int64_t B, int t, int64_t R) {
  const int64_t G = R / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B * t * G; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = (i % G) * 8; const int64_t bt = i / G; const int64_t b = bt / t;
    float v[8];
    load8(dy + b * R + c, v);
    const float inv = 1.f / t;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= inv;
    store8(dx + bt * R + c, v);
  }
}

// ---- dst[r][c] = (r < rows && c < cols) ? src[r*lds + c] * colscale[c] : 0   over (rows_dst, cols_dst); any dtypes
template <typename TS, typename TD>
__global__ void convert_pad_kernel(const TS* __restrict__ src, TD* __restrict__ dst, const float* __restrict__ colscale, int64_t rows,
                                   int64_t cols, int64_t lds_, int64_t rows_dst, int64_t cols_dst, int64_t ldd) {
  const int64_t n = rows_dst * cols_dst;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols_dst, c = i % cols_dst;
    float v = 0.f;
    if (r < rows && c < cols) { v = Elem<TS>::ld(src + r * lds_ + c); if (colscale) v *= colscale[c]; }
    Elem<TD>::st(dst + r * ldd + c, v);
  }
}

// ---- continuous position bias (attention.py:257-276): table (nclass, H) -> bias (H, L, L), L = gh*gw,
//      class(i,j) = (iy-jy+gh-1)*(2gw-1) + (ix-jx+gw-1).  Backward: table-grad gather of dbias (deterministic).
__global__ void cpb_expand_kernel(const float* __restrict__ tab, float* __restrict__ bias, int H, int gh, int gw) {
  const int L = gh * gw;
  const int64_t n = (int64_t)H * L * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % L); const int ii = (int)((i / L) % L); const int h = (int)(i / ((int64_t)L * L));
    const int cls = (ii / gw - j / gw + gh - 1) * (2 * gw - 1) + (ii % gw - j % gw + gw - 1);
    bias[i] = tab[(int64_t)cls * H + h];
  }
}
__global__ void cpb_reduce_kernel(const float* __restrict__ dbias, float* __restrict__ dtab, int H, int gh, int gw) {
  const int L = gh * gw, ncls = (2 * gh - 1) * (2 * gw - 1);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncls * H) return;
  const int h = i % H, cls = i / H;
  const int dy = cls / (2 * gw - 1) - (gh - 1), dx = cls % (2 * gw - 1) - (gw - 1);
  float t = 0.f;
  for (int iy = (dy > 0 ? dy : 0); iy < gh && iy - dy < gh; ++iy)
    for (int ix = (dx > 0 ? dx : 0); ix < gw && ix - dx < gw; ++ix) {
      const int qi = iy * gw + ix, kj = (iy - dy) * gw + (ix - dx);
      t += dbias[((int64_t)h * L + qi) * L + kj];
    }
  dtab[(int64_t)cls * H + h] = t;
}

// ---- dropout (HF BertEmbeddings / BertSelfOutput / BertOutput: nn.Dropout(hidden_dropout_prob) in train mode):
//      y = x * keep / (1 - p) (+ residual); element i takes word (i & 3) of philox(seed, i >> 2, stream).
//      The same call with dy in place of x is the backward.
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, int64_t n4, float p, float inv_keep,
                               uint64_t seed, uint32_t stream, const unsigned long long* __restrict__ st) {
  if (st) seed += st[0];                                     // (device-resident step state: see ctclip_set_step_state)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 w = philox4x32(seed, (uint64_t)i, stream);
    float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
    load4(x + i * 4, v);
    if (res) load4(res + i * 4, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * dropout_mult(w[e], p, inv_keep) + r[e];
    store4(y + i * 4, v);
  }
}
// the multiplier the attention kernels apply to probability (seq, h, i, j) (common.h attn_drop_block / attn_drop_pick) -- for tests
__global__ void attn_dropout_mask_kernel(float* __restrict__ mask, int64_t n, int L, float p, float inv_keep, uint64_t seed) {
  const uint32_t thr = attn_drop_threshold(p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int kj = (int)(i % L), qi = (int)((i / L) % L);
    const int64_t sh = i / ((int64_t)L * L);
    mask[i] = attn_drop_pick(attn_drop_block(seed, sh, L, qi, kj), qi, kj, thr, inv_keep);
  }
}

// ---- BERT embeddings (HF BertEmbeddings): x[r] = word[ids[r]] + pos[r % T] + type[0]
template <typename T>
__global__ void bert_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                      const float* __restrict__ type0, T* __restrict__ x, int64_t rows, int Tlen, int Hd) {
  const int G = Hd / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * G; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / G; const int c = (int)(i % G) * 4;
    float a[4], b[4], d[4], o[4];
    load4(word + ids[r] * Hd + c, a);
    load4(pos + (r % Tlen) * Hd + c, b);
    load4(type0 + c, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e] + b[e] + d[e];
    store4(x + r * Hd + c, o);
  }
}
// (the embedding-table gradients are deterministic segmented sums: segsum.hip)

// ---- vector quantiser (vector_quantize_pytorch 1.1.2 cosine codebook; called at ctvit.py:403)
template <typename T>
__global__ void vq_gather_kernel(const float* __restrict__ embed, const int64_t* __restrict__ idx, T* __restrict__ out, int64_t M, int d) {
  const int G = d / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * G; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / G; const int c = (int)(i % G) * 4;
    float v[4];
    load4(embed + idx[r] * d + c, v);
    store4(out + r * d + c, v);
  }
}
// (bins / esum: deterministic segmented sums, segsum.hip)
// one wave per code: cluster_size <- lerp ; embed <- decay*embed + (1-decay)*(bins ? l2norm(esum/bins) : l2norm(embed))
__global__ __launch_bounds__(256) void vq_ema_update_kernel(float* __restrict__ cluster, float* __restrict__ embed,
                                                            const float* __restrict__ bins, const float* __restrict__ esum, int C, int d,
                                                            float decay) {
  const int lane = threadIdx.x & 63;
  const int code = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (code >= C) return;
  const float b = bins[code];
  if (lane == 0) cluster[code] = cluster[code] * decay + b * (1.f - decay);
  const float* src = (b == 0.f) ? embed + (int64_t)code * d : esum + (int64_t)code * d;
  const float div = (b == 0.f) ? 1.f : b;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) { const float v = src[c] / div; s += v * v; }
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int c = lane; c < d; c += 64) {
    const float v = src[c] / div * inv;
    embed[(int64_t)code * d + c] = embed[(int64_t)code * d + c] * decay + v * (1.f - decay);
  }
}

}  // namespace

#define BY_DTYPE(dtype, CALL)                                                 \
  if (dtype == DT_F32) { using T = float; CALL; }                             \
  else if (dtype == DT_BF16) { using T = bf16_t; CALL; }                      \
  else { ctclip_set_error("unsupported dtype"); return CTCLIP_EUNSUPPORTED; }

extern "C" int ctclip_geglu_fwd(const void* u, void* g, int64_t M, int Hp, int dtype, hipStream_t s) {
  if (!u || !g || Hp % 8) { ctclip_set_error("geglu_fwd: hidden must be a multiple of 8"); return CTCLIP_EBADARG; }
  BY_DTYPE(dtype, hipLaunchKernelGGL(geglu_fwd_kernel<T>, grid_for(M * Hp / 8), dim3(256), 0, s, (const T*)u, (T*)g, M, Hp));
  return ctclip_check_launch("geglu_fwd");
}
extern "C" int ctclip_geglu_bwd(const void* dg, const void* u, void* du, int64_t M, int Hp, int dtype, hipStream_t s) {
  if (!dg || !u || !du || Hp % 8) { ctclip_set_error("geglu_bwd: bad args"); return CTCLIP_EBADARG; }
  BY_DTYPE(dtype, hipLaunchKernelGGL(geglu_bwd_kernel<T>, grid_for(M * Hp / 8), dim3(256), 0, s, (const T*)dg, (const T*)u, (T*)du, M, Hp));
  return ctclip_check_launch("geglu_bwd");
}
extern "C" int ctclip_gelu_fwd(const void* u, void* h, int64_t n, int dtype, hipStream_t s) {
  if (!u || !h || n % 8) { ctclip_set_error("gelu_fwd: n must be a multiple of 8"); return CTCLIP_EBADARG; }
  BY_DTYPE(dtype, hipLaunchKernelGGL(gelu_fwd_kernel<T>, grid_for(n / 8), dim3(256), 0, s, (const T*)u, (T*)h, n / 8));
  return ctclip_check_launch("gelu_fwd");
}
extern "C" int ctclip_gelu_bwd(const void* dh, const void* u, void* du, int64_t n, int dtype, hipStream_t s) {
  if (!dh || !u || !du || n % 8) { ctclip_set_error("gelu_bwd: bad args"); return CTCLIP_EBADARG; }
  BY_DTYPE(dtype, hipLaunchKernelGGL(gelu_bwd_kernel<T>, grid_for(n / 8), dim3(256), 0, s, (const T*)dh, (const T*)u, (T*)du, n / 8));
  return ctclip_check_launch("gelu_bwd");
}
extern "C" int ctclip_leaky_relu_fwd(const float* x, float* y, int64_t n, float slope, hipStream_t s) {
  hipLaunchKernelGGL(leaky_fwd_kernel, grid_for(n), dim3(256), 0, s, x, y, n, slope);
  return ctclip_check_launch("leaky_relu_fwd");
}
extern "C" int ctclip_leaky_relu_bwd(const float* dy, const float* x, float* dx, int64_t n, float slope, hipStream_t s) {
  hipLaunchKernelGGL(leaky_bwd_kernel, grid_for(n), dim3(256), 0, s, dy, x, dx, n, slope);
  return ctclip_check_launch("leaky_relu_bwd");
}
// out (f32, N) += column sums of x (M, N)
static inline int colsum_rows(int64_t M) { return M <= 16384 ? 64 : 512; }
extern "C" int64_t ctclip_colsum_workspace(int64_t M, int N) { return cdiv(M, colsum_rows(M)) * (int64_t)N * 4; }
// out[0:N] += column sums of x (M, N) -- bias gradients of the Linear layers.  Two stages with a fixed summation order (the first
// version added per-block partial sums with f32 atomics); workspace >= ctclip_colsum_workspace(M, N) bytes.
extern "C" int ctclip_colsum(const void* x, float* out, int64_t M, int N, int64_t ld, int dtype, void* workspace, int64_t workspace_bytes,
                             hipStream_t s) {
  if (!x || !out) { ctclip_set_error("colsum: null argument"); return CTCLIP_EBADARG; }
  if (!workspace || workspace_bytes < ctclip_colsum_workspace(M, N)) { ctclip_set_error("colsum: workspace too small"); return CTCLIP_EWORKSPACE; }
  float* part = (float*)workspace;
  int nblk;
  if (N % 8 || ld % 8 || ((uintptr_t)x % 16)) {
    nblk = (int)cdiv(M, 1024);
    dim3 gridg((unsigned)nblk, (unsigned)cdiv(N, 64));
    BY_DTYPE(dtype, hipLaunchKernelGGL(colsum_generic_kernel<T>, gridg, dim3(256), 0, s, (const T*)x, part, M, N, ld));
  } else {
    const int rows = colsum_rows(M);
    nblk = (int)cdiv(M, rows);
    dim3 grid((unsigned)nblk, (unsigned)cdiv(N, 256));
    if (rows == 64) { BY_DTYPE(dtype, hipLaunchKernelGGL((colsum_kernel<T, 64>), grid, dim3(256), 0, s, (const T*)x, part, M, N, ld)); }
    else { BY_DTYPE(dtype, hipLaunchKernelGGL((colsum_kernel<T, 512>), grid, dim3(256), 0, s, (const T*)x, part, M, N, ld)); }
  }
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, (const float*)part, nblk, N, out);
  return ctclip_check_launch("colsum");
}
extern "C" int ctclip_permute0213(const void* x, void* y, int64_t A, int B, int C, int D, int dtype, hipStream_t s) {
  if (!x || !y || D % 8) { ctclip_set_error("permute0213: D must be a multiple of 8"); return CTCLIP_EBADARG; }
  BY_DTYPE(dtype, hipLaunchKernelGGL(permute0213_kernel<T>, grid_for(A * B * C * D / 8), dim3(256), 0, s, (const T*)x, (T*)y, A, B, C, D));
  return ctclip_check_launch("permute0213");
}
extern "C" int ctclip_transpose2d(const void* x, void* y, int R, int C, int64_t ldx, int64_t ldy, int dtype, hipStream_t s) {
  if (!x || !y) return CTCLIP_EBADARG;
  dim3 grid((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32));
  BY_DTYPE(dtype, hipLaunchKernelGGL(transpose2d_kernel<T>, grid, dim3(256), 0, s, (const T*)x, (T*)y, R, C, ldx, ldy));
  return ctclip_check_launch("transpose2d");
}
// (dtype, out_dtype): the mixed-precision head pools bf16 tokens into an f32 vector and hands the f32 gradient back as bf16
#define BY_DTYPE2(din, dout, KERNEL, GRID, ...)                                                                                        \
  if (din == DT_F32 && dout == DT_F32) hipLaunchKernelGGL((KERNEL<float, float>), GRID, dim3(256), 0, s, (const float*)src_, (float*)dst_, __VA_ARGS__);          \
  else if (din == DT_BF16 && dout == DT_BF16) hipLaunchKernelGGL((KERNEL<bf16_t, bf16_t>), GRID, dim3(256), 0, s, (const bf16_t*)src_, (bf16_t*)dst_, __VA_ARGS__); \
  else if (din == DT_BF16 && dout == DT_F32) hipLaunchKernelGGL((KERNEL<bf16_t, float>), GRID, dim3(256), 0, s, (const bf16_t*)src_, (float*)dst_, __VA_ARGS__);   \
  else if (din == DT_F32 && dout == DT_BF16) hipLaunchKernelGGL((KERNEL<float, bf16_t>), GRID, dim3(256), 0, s, (const float*)src_, (bf16_t*)dst_, __VA_ARGS__);   \
  else { ctclip_set_error("unsupported dtype pair"); return CTCLIP_EUNSUPPORTED; }
extern "C" int ctclip_pool_fwd(const void* x, void* y, int64_t B, int t, int64_t R, int dtype, int out_dtype, hipStream_t s) {
  if (!x || !y || R % 8) { ctclip_set_error("pool_fwd: R must be a multiple of 8"); return CTCLIP_EBADARG; }
  const void* src_ = x; void* dst_ = y;
  BY_DTYPE2(dtype, out_dtype, pool_fwd_kernel, grid_for(B * R / 8), B, t, R);
  return ctclip_check_launch("pool_fwd");
}
extern "C" int ctclip_pool_bwd(const void* dy, void* dx, int64_t B, int t, int64_t R, int dtype, int out_dtype, hipStream_t s) {
  if (!dy || !dx || R % 8) { ctclip_set_error("pool_bwd: bad args"); return CTCLIP_EBADARG; }
  const void* src_ = dy; void* dst_ = dx;
  BY_DTYPE2(dtype, out_dtype, pool_bwd_kernel, grid_for(B * t * R / 8), B, t, R);
  return ctclip_check_launch("pool_bwd");
}
#undef BY_DTYPE2
extern "C" int ctclip_convert_pad(const void* src, void* dst, const float* colscale, int64_t rows, int64_t cols, int64_t lds_,
                                  int64_t rows_dst, int64_t cols_dst, int64_t ldd, int src_dtype, int dst_dtype, hipStream_t s) {
  if (!src || !dst) return CTCLIP_EBADARG;
  dim3 g = grid_for(rows_dst * cols_dst);
#define L(TS, TD) hipLaunchKernelGGL((convert_pad_kernel<TS, TD>), g, dim3(256), 0, s, (const TS*)src, (TD*)dst, colscale, rows, cols, lds_, rows_dst, cols_dst, ldd)
  if (src_dtype == DT_F32 && dst_dtype == DT_F32) L(float, float);
  else if (src_dtype == DT_F32 && dst_dtype == DT_BF16) L(float, bf16_t);
  else if (src_dtype == DT_BF16 && dst_dtype == DT_F32) L(bf16_t, float);
  else if (src_dtype == DT_BF16 && dst_dtype == DT_BF16) L(bf16_t, bf16_t);
  else return CTCLIP_EUNSUPPORTED;
#undef L
  return ctclip_check_launch("convert_pad");
}
extern "C" int ctclip_cpb_expand(const float* tab, float* bias, int H, int gh, int gw, hipStream_t s) {
  const int64_t L = (int64_t)gh * gw;
  hipLaunchKernelGGL(cpb_expand_kernel, grid_for(H * L * L), dim3(256), 0, s, tab, bias, H, gh, gw);
  return ctclip_check_launch("cpb_expand");
}
extern "C" int ctclip_cpb_reduce(const float* dbias, float* dtab, int H, int gh, int gw, hipStream_t s) {
  const int n = (2 * gh - 1) * (2 * gw - 1) * H;
  hipLaunchKernelGGL(cpb_reduce_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, dbias, dtab, H, gh, gw);
  return ctclip_check_launch("cpb_reduce");
}
// nn.Dropout in train mode (HF modeling_bert.py BertEmbeddings / BertSelfOutput / BertOutput), optionally fused with the residual
// add that follows it: y = dropout(x) + residual.  n % 4 == 0.  `stream_id` separates the call sites of one step.
extern "C" int ctclip_dropout(const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed, uint32_t stream_id,
                              int dtype, hipStream_t s) {
  if (!x || !y || n % 4 || p < 0.f || p >= 1.f) { ctclip_set_error("dropout: n % 4 == 0, 0 <= p < 1"); return CTCLIP_EBADARG; }
  const float inv_keep = 1.f / (1.f - p);
  BY_DTYPE(dtype, hipLaunchKernelGGL(dropout_kernel<T>, grid_for(n / 4), dim3(256), 0, s, (const T*)x, (const T*)residual, (T*)y, n / 4, p, inv_keep, seed, stream_id, ctclip_step_state()));
  return ctclip_check_launch("dropout");
}
// mask[(seq, h, i, j)] = the multiplier ctclip_attn_fwd/bwd apply to attention probability (seq, h, i, j) for (p, seed): 0 or 1/(1-p)
extern "C" int ctclip_attn_dropout_mask(float* mask, int nseq, int H, int L, float p, uint64_t seed, hipStream_t s) {
  if (!mask || p < 0.f || p >= 1.f) { ctclip_set_error("attn_dropout_mask: bad args"); return CTCLIP_EBADARG; }
  const int64_t n = (int64_t)nseq * H * L * L;
  hipLaunchKernelGGL(attn_dropout_mask_kernel, grid_for(n), dim3(256), 0, s, mask, n, L, p, 1.f / (1.f - p), seed);
  return ctclip_check_launch("attn_dropout_mask");
}
extern "C" int ctclip_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, void* x, int64_t rows,
                                     int Tlen, int Hd, int dtype, hipStream_t s) {
  if (!ids || !word || !pos || !type0 || !x || Hd % 8) { ctclip_set_error("bert_embed_fwd: bad args"); return CTCLIP_EBADARG; }
  BY_DTYPE(dtype, hipLaunchKernelGGL(bert_embed_fwd_kernel<T>, grid_for(rows * Hd / 4), dim3(256), 0, s, ids, word, pos, type0, (T*)x, rows, Tlen, Hd));
  return ctclip_check_launch("bert_embed_fwd");
}
extern "C" int ctclip_vq_gather(const float* embed, const int64_t* idx, void* out, int64_t M, int d, int dtype, hipStream_t s) {
  if (!embed || !idx || !out || d % 8) { ctclip_set_error("vq_gather: bad args"); return CTCLIP_EBADARG; }
  BY_DTYPE(dtype, hipLaunchKernelGGL(vq_gather_kernel<T>, grid_for(M * d / 4), dim3(256), 0, s, embed, idx, (T*)out, M, d));
  return ctclip_check_launch("vq_gather");
}
extern "C" int ctclip_vq_ema_update(float* cluster, float* embed, const float* bins, const float* esum, int C, int d, float decay, hipStream_t s) {
  if (!cluster || !embed || !bins || !esum) return CTCLIP_EBADARG;
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, s, cluster, embed, bins, esum, C, d, decay);
  return ctclip_check_launch("vq_ema_update");
}

// ---- parameter-space epilogue of the patch-embedding backward (ctvit.py:170-175 with LayerNorm(K)'s affine folded into the Linear:
// W' = W * gamma1 per column, b' = W beta1 + b).  G = dZ^T xhat (N x K, f32) and dbp = colsum(dZ) are in hand; this turns them into
//   dW[n][k] (+)= G[n][k] * gamma1[k] + dbp[n] * beta1[k],   dgamma1[k] (+)= sum_n W[n][k] G[n][k],   dbeta1[k] (+)= sum_n W[n][k] dbp[n].
// One workgroup owns 64 columns; eight row groups walk N / 8 rows each and are folded through LDS in a fixed order (deterministic).
namespace {
__global__ __launch_bounds__(512) void patch_param_bwd_kernel(const float* __restrict__ G, const float* __restrict__ W, const float* __restrict__ g1,
                                                              const float* __restrict__ b1, const float* __restrict__ dbp, float* __restrict__ dW,
                                                              float* __restrict__ dg1, float* __restrict__ db1, int N, int K, int accumulate) {
  __shared__ float red[2][8][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + c;
  const bool on = k < K;
  const int kk = on ? k : K - 1;
  const float gam = g1[kk], bet = b1[kk];
  float sg = 0.f, sb = 0.f;
  const int per = (N + 7) / 8;
  const int n1 = (rg + 1) * per < N ? (rg + 1) * per : N;
  for (int n = rg * per; n < n1; ++n) {
    const float g = G[(int64_t)n * K + kk], w = W[(int64_t)n * K + kk], d = dbp[n];
    sg = fmaf(w, g, sg);
    sb = fmaf(w, d, sb);
    if (on) {
      const float v = fmaf(g, gam, d * bet);
      float* o = dW + (int64_t)n * K + k;
      *o = accumulate ? *o + v : v;
    }
  }
  red[0][rg][c] = sg; red[1][rg][c] = sb;
  __syncthreads();
  if (rg == 0 && on) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a += red[0][j][c]; b += red[1][j][c]; }
    dg1[k] = accumulate ? dg1[k] + a : a;
    db1[k] = accumulate ? db1[k] + b : b;
  }
}
}  // namespace

extern "C" int ctclip_patch_embed_param_bwd(const float* G, const float* W, const float* gamma1, const float* beta1, const float* dbp, float* dW,
                                            float* dgamma1, float* dbeta1, int N, int K, int accumulate, hipStream_t s) {
  if (!G || !W || !gamma1 || !beta1 || !dbp || !dW || !dgamma1 || !dbeta1 || N <= 0 || K <= 0) { ctclip_set_error("patch_embed_param_bwd: bad args"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(patch_param_bwd_kernel, dim3((unsigned)cdiv(K, 64)), dim3(512), 0, s, G, W, gamma1, beta1, dbp, dW, dgamma1, dbeta1, N, K, accumulate);
  return ctclip_check_launch("patch_embed_param_bwd");
}

// One thread spinning on the 100-MHz wall clock for `microseconds`: holds a hardware queue busy without touching memory.  ct_clip_amd/streams.py
// probes with it whether a side stream runs BESIDE the default stream (it used torch.cuda._sleep, a private API, up to round 5).
namespace {
__global__ void spin_kernel(unsigned long long ticks, unsigned int* sink) {
  const unsigned long long t0 = wall_clock64();
  unsigned int n = 0;
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++n; }
  if (sink && ticks == ~0ull) *sink = n;      // (never true: keeps the loop observable)
}
}  // namespace
extern "C" int ctclip_spin(int64_t microseconds, hipStream_t s) {
  if (microseconds < 0 || microseconds > 1000000) { ctclip_set_error("spin: 0 .. 1 000 000 microseconds"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, s, (unsigned long long)microseconds * 100ull, (unsigned int*)nullptr);
  return ctclip_check_launch("spin");
}
