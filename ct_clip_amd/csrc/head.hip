// CLIP head of CT-CLIP on gfx950:
//   * to_visual_latent (ct_clip.py:564,767): Linear(h*w*dim -> dim_latent, no bias) at M = per-GPU batch.
//     K = 294 912, N = 512, M = 8: a pure HBM stream of the 151 M-parameter weight (GEMV-like), forward and backward.
//   * l2norm + logits * exp(temperature) + symmetric InfoNCE in the exp/sum/log(t+1e-20) form, forward AND
//     backward in one launch (ct_clip.py:771,796,845-846,858-878,890-901).
#include "common.h"

namespace {

constexpr int VL_MAXB = 8;    // rows of X per call (the host loops over chunks of 8 samples)
constexpr int VL_KCH = 2048;  // k-chunk per block
#ifndef VL_FWD_RG
#define VL_FWD_RG 2             // f32 forward: groups of four weight rows per wave and staged X chunk
#endif
// wide form of the BACKWARD (f32 only): up to 24 rows per launch, 4 k per lane -- the VocabFine step projects its 18 pooled vectors (one per
// pathology) and used to stream the 604-MB weight and its gradient (read + write) once per chunk of 8 rows: 3 x 440 us, now 644-655.
// Measured and dropped: 2 k per lane (1 032 us), the weight rows split over the halves of a workgroup (668 us: 253 registers allow two
// waves per SIMD, so twice the waves run in two rounds), a 24-row FORWARD (665-843 us against 3 x 172: the cross-lane reductions of
// 4 x 24 partial sums per wave and chunk cost as much as the products).
constexpr int VLW_MAXB = 24, VLW_KW = 4, VLW_THREADS = 256;

template <int KW> __device__ __forceinline__ void loadk(const float* p, float (&v)[KW]);
template <> __device__ __forceinline__ void loadk<8>(const float* p, float (&v)[8]) { load8(p, v); }
template <> __device__ __forceinline__ void loadk<4>(const float* p, float (&v)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
template <int KW> __device__ __forceinline__ void loadk(const bf16_t* p, float (&v)[KW]) { static_assert(KW == 8, "bf16: 8 per lane"); load8(p, v); }
template <int KW> __device__ __forceinline__ void storek(float* p, const float (&v)[KW]);
template <> __device__ __forceinline__ void storek<8>(float* p, const float (&v)[8]) { store8(p, v); }
template <> __device__ __forceinline__ void storek<4>(float* p, const float (&v)[4]) {
  f32x4 a; a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
  *reinterpret_cast<f32x4*>(p) = a;
}
template <int KW> __device__ __forceinline__ void storek(bf16_t* p, const float (&v)[KW]) { static_assert(KW == 8, "bf16: 8 per lane"); store8(p, v); }

// part[chunk][b][n] = sum_{k in chunk} X[b][k] * W[n][k].  block = 4 waves x 4 weight rows; X chunk staged in LDS as f32.
// (The k-chunks are combined by vlat_reduce_kernel in chunk order: the first version added them with f32 atomics.)
template <typename T, int RG = 1>      // RG: groups of four weight rows per wave and staged X chunk (2: half the staging traffic)
__global__ __launch_bounds__(256) void vlat_fwd_kernel(const T* __restrict__ X, const T* __restrict__ W, float* __restrict__ part, int Bm,
                                                       int N, int64_t K) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [Bm][VL_KCH]
  const int64_t k0 = (int64_t)blockIdx.y * VL_KCH;
  const int kc = (int)((K - k0) < VL_KCH ? (K - k0) : VL_KCH);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbase = (blockIdx.x * 4 + wave) * 4 * RG;
  // the weight rows of k-slice `it` (64 lanes x 8 k): requested one slice AHEAD of their use, the first one before the X chunk is staged
  // (round 6: the loop used to fetch and consume a slice per iteration -- one memory round trip per slice and wave, behind the staging barrier:
  // 2.8 TB/s)
  constexpr int NIT = VL_KCH / 512;
  float w[2][4][8];
  auto loadw = [&](int j, float (&dst)[4][8]) {            // j = row group * NIT + k-slice
    const int c = lane * 8 + (j % NIT) * 512, n0 = nbase + (j / NIT) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[r][e] = 0.f;
      if (n0 + r < N && c < kc) load8(W + (int64_t)(n0 + r) * K + k0 + c, dst[r]);
    }
  };
  loadw(0, w[0]);
  for (int i = threadIdx.x; i < Bm * (VL_KCH / 8); i += 256) {
    const int b = i / (VL_KCH / 8), c = (i % (VL_KCH / 8)) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (c < kc) load8(X + (int64_t)b * K + k0 + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) xs[b * VL_KCH + c + e] = v[e];
  }
  __syncthreads();
  float acc[4][VL_MAXB];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int b = 0; b < VL_MAXB; ++b) acc[r][b] = 0.f;
#pragma unroll
  for (int j = 0; j < RG * NIT; ++j) {
    if (j + 1 < RG * NIT) loadw(j + 1, w[(j + 1) & 1]);
    const int c = lane * 8 + (j % NIT) * 512;
#pragma unroll
    for (int b = 0; b < VL_MAXB; ++b) {
      if (b < Bm) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = xs[b * VL_KCH + c + e];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r][b] += w[j & 1][r][e] * xv[e];
      }
    }
    if (j % NIT == NIT - 1) {                               // this row group is complete
      const int n0 = nbase + (j / NIT) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < VL_MAXB; ++b) {
          if (b < Bm) {
            const float t = wave_sum(acc[r][b]);
            if (lane == 0 && n0 + r < N) part[((int64_t)blockIdx.y * Bm + b) * N + n0 + r] = t;
          }
          acc[r][b] = 0.f;
        }
    }
  }
}

__global__ __launch_bounds__(256) void vlat_reduce_kernel(const float* __restrict__ part, int nchunk, int n, float* __restrict__ Y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
  for (int c = 0; c < nchunk; ++c) t += part[(int64_t)c * n + i];
  Y[i] = t;
}

// Each thread owns KW consecutive k.  dX[b][k] = sum_n dY[b][n] W[n][k] ;  dW[n][k] (+)= sum_b dY[b][n] X[b][k].
// The walk over n is unrolled by four with every load of the group issued first (the first version fetched W and the old dW of one
// row per iteration, the latter behind a branch: one dependent round trip per row at 2 waves per CU -- 800 us for 1.5 GB).
template <typename T, bool ACC, bool WANT_DW, int MAXB = VL_MAXB, int KW = 8, int THREADS = 128, int UNR = 4>
__global__ __launch_bounds__(THREADS) void vlat_bwd_kernel(const float* __restrict__ dY, const T* __restrict__ X, const T* __restrict__ W,
                                                           T* __restrict__ dX, float* __restrict__ dW, int Bm, int N, int64_t K) {
  extern __shared__ __attribute__((aligned(16))) float dys[];  // [Bm][N]
  for (int i = threadIdx.x; i < Bm * N; i += THREADS) dys[i] = dY[i];
  __syncthreads();
  const int64_t k = ((int64_t)blockIdx.x * THREADS + threadIdx.x) * KW;
  if (k >= K) return;
  float xv[MAXB][KW], dx[MAXB][KW];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
#pragma unroll
    for (int e = 0; e < KW; ++e) { xv[b][e] = 0.f; dx[b][e] = 0.f; }
    if (b < Bm) loadk<KW>(X + (int64_t)b * K + k, xv[b]);
  }
  for (int n0 = 0; n0 < N; n0 += UNR) {
    float w[UNR][KW], old[UNR][KW];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int n = n0 + u < N ? n0 + u : N - 1;           // clamped, always issued
      loadk<KW>(W + (int64_t)n * K + k, w[u]);
      if (ACC && WANT_DW) loadk<KW>(dW + (int64_t)n * K + k, old[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int n = n0 + u;
      if (n >= N) break;
      float g[KW];
#pragma unroll
      for (int e = 0; e < KW; ++e) g[e] = (ACC && WANT_DW) ? old[u][e] : 0.f;
#pragma unroll
      for (int b = 0; b < MAXB; ++b) {
        if (b < Bm) {
          const float d = dys[b * N + n];
#pragma unroll
          for (int e = 0; e < KW; ++e) { dx[b][e] += d * w[u][e]; g[e] += d * xv[b][e]; }
        }
      }
      if (WANT_DW) storek<KW>(dW + (int64_t)n * K + k, g);
    }
  }
  if (dX) {
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
      if (b < Bm) storek<KW>(dX + (int64_t)b * K + k, dx[b]);
  }
}

// ------------------------------------------------------------------------------------------- CLIP loss
// single block.  tl, il: (G, Dl) raw (pre-l2norm) latents f32.  out[0] = loss, out[1] = exp(temperature).
// grads: dtl, dil (G, Dl) f32, dtemp[0] (+=).  logits (G,G) optional.
__global__ __launch_bounds__(256) void clip_loss_kernel(const float* __restrict__ tl, const float* __restrict__ il,
                                                        const float* __restrict__ temperature, float* __restrict__ out,
                                                        float* __restrict__ logits, float* __restrict__ dtl, float* __restrict__ dil,
                                                        float* __restrict__ dtemp, int G, int Dl) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* S = sm;                  // G*G : logits, later dS
  float* tinv = S + G * G;        // G
  float* iinv = tinv + G;         // G
  float* rsum = iinv + G;         // G
  float* csum = rsum + G;         // G
  float* red = csum + G;          // 16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float temp = __expf(temperature[0]);
  const float eps = 1e-20f;
  // 1. inverse norms (F.normalize eps 1e-12)
  for (int r = wave; r < 2 * G; r += 4) {
    const float* src = (r < G) ? tl + (int64_t)r * Dl : il + (int64_t)(r - G) * Dl;
    float s = 0.f;
    for (int c = lane; c < Dl; c += 64) s += src[c] * src[c];
    s = wave_sum(s);
    if (lane == 0) { const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f); if (r < G) tinv[r] = inv; else iinv[r - G] = inv; }
  }
  __syncthreads();
  // 2. logits S[i][j] = temp * <t_i, v_j> * tinv_i * iinv_j   (rows = texts, cols = images)
  for (int ij = wave; ij < G * G; ij += 4) {
    const int i = ij / G, j = ij % G;
    float s = 0.f;
    for (int c = lane; c < Dl; c += 64) s += tl[(int64_t)i * Dl + c] * il[(int64_t)j * Dl + c];
    s = wave_sum(s);
    if (lane == 0) { const float v = s * tinv[i] * iinv[j] * temp; S[ij] = v; if (logits) logits[ij] = v; }
  }
  __syncthreads();
  // 3. row / column sums of exp(S)
  for (int r = tid; r < 2 * G; r += 256) {
    float s = 0.f;
    if (r < G) { for (int j = 0; j < G; ++j) s += __expf(S[r * G + j]); rsum[r] = s; }
    else { const int j = r - G; for (int i = 0; i < G; ++i) s += __expf(S[i * G + j]); csum[j] = s; }
  }
  __syncthreads();
  float part = 0.f;
  for (int i = tid; i < G; i += 256) {
    const float e = __expf(S[i * G + i]);
    part += -2.f * __logf(e + eps) + __logf(rsum[i] + eps) + __logf(csum[i] + eps);
  }
  const float total = block_sum(part, red);
  if (tid == 0) { out[0] = total / (2.f * G); out[1] = temp; }
  if (!dtl) return;
  // 4. dS (in place), d temperature
  float dth = 0.f;
  for (int ij = tid; ij < G * G; ij += 256) {
    const int i = ij / G, j = ij % G;
    const float s = S[ij], e = __expf(s);
    float d = e / (rsum[i] + eps) + e / (csum[j] + eps);
    if (i == j) d -= 2.f * e / (e + eps);
    d /= (2.f * G);
    dth += d * s;
    S[ij] = d;
  }
  __syncthreads();
  const float dtheta = block_sum(dth, red);
  if (tid == 0 && dtemp) dtemp[0] += dtheta;
  // 5. latent grads through logits and l2norm:  u = raw * inv ;  du_t[i] = temp * sum_j dS_ij u_v[j] ; d raw = inv (du - u <u,du>)
  for (int r = wave; r < 2 * G; r += 4) {
    const bool text = r < G;
    const int row = text ? r : r - G;
    const float* raw = text ? tl + (int64_t)row * Dl : il + (int64_t)row * Dl;
    const float inv = text ? tinv[row] : iinv[row];
    float dot = 0.f;
    // Dl <= 64 * 16
    float du[16], u[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = q * 64 + lane;
      du[q] = 0.f; u[q] = 0.f;
      if (c < Dl) {
        float a = 0.f;
        for (int o = 0; o < G; ++o) {
          const float d = text ? S[row * G + o] : S[o * G + row];
          const float ou = text ? il[(int64_t)o * Dl + c] * iinv[o] : tl[(int64_t)o * Dl + c] * tinv[o];
          a += d * ou;
        }
        du[q] = a * temp;
        u[q] = raw[c] * inv;
        dot += u[q] * du[q];
      }
    }
    dot = wave_sum(dot);
    float* dst = text ? dtl + (int64_t)row * Dl : dil + (int64_t)row * Dl;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = q * 64 + lane;
      if (c < Dl) dst[c] = inv * (du[q] - u[q] * dot);
    }
  }
}

// ------------------------------------------------------------------------------------------- CLIP loss, any G (logits in global memory)
// The single-block kernel above keeps the G x G logits in LDS (G <= 128 = 16 ranks x batch 8).  Larger gathered batches run as a few
// launches around the f32 GEMMs the host issues: S = temp * Ut Uv^T (ctclip_gemm), then the kernels below, then dUt = temp dS Uv,
// dUv = temp dS^T Ut (ctclip_gemm) and the l2norm backward.  Every sum has a fixed order.
// rsum[i] = sum_j exp(S_ij) (one wave per row); csum[j] = sum_i exp(S_ij) (one thread per column, coalesced across the block)
__global__ __launch_bounds__(256) void clip_rowcol_kernel(const float* __restrict__ S, int64_t ld, const float* __restrict__ temperature, int G,
                                                          float* __restrict__ rsum, float* __restrict__ csum) {
  const int nrowblocks = (G + 3) / 4;
  const float temp = __expf(temperature[0]);
  if ((int)blockIdx.x < nrowblocks) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= G) return;
    float s = 0.f;
    for (int j = lane; j < G; j += 64) s += __expf(S[(int64_t)i * ld + j] * temp);
    s = wave_sum(s);
    if (lane == 0) rsum[i] = s;
  } else {
    const int j = (blockIdx.x - nrowblocks) * 256 + threadIdx.x;
    if (j >= G) return;
    float s = 0.f;
    for (int i = 0; i < G; ++i) s += __expf(S[(int64_t)i * ld + j] * temp);
    csum[j] = s;
  }
}
// S (cosines) -> temp * d loss / d logits in place (the factor the latent gradients carry); per row: loss term and sum_j dS_ij logit_ij
// (for d temperature).  One wave per row.
__global__ __launch_bounds__(256) void clip_ds_kernel(float* __restrict__ S, int64_t ld, const float* __restrict__ temperature, int G, const float* __restrict__ rsum,
                                                      const float* __restrict__ csum, float* __restrict__ row_loss, float* __restrict__ row_dth) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= G) return;
  const float temp = __expf(temperature[0]);
  const float eps = 1e-20f, ri = rsum[i] + eps, inv2g = 1.f / (2.f * G);
  float dth = 0.f, diag = 0.f;
  for (int j = lane; j < G; j += 64) {
    const float sv = S[(int64_t)i * ld + j] * temp, e = __expf(sv);
    float d = e / ri + e / (csum[j] + eps);
    if (i == j) { d -= 2.f * e / (e + eps); diag = e; }
    d *= inv2g;
    dth += d * sv;
    S[(int64_t)i * ld + j] = d * temp;
  }
  dth = wave_sum(dth); diag = wave_sum(diag);
  if (lane == 0) { row_dth[i] = dth; row_loss[i] = -2.f * __logf(diag + eps) + __logf(ri) + __logf(csum[i] + eps); }
}
__global__ __launch_bounds__(256) void clip_final_kernel(const float* __restrict__ row_loss, const float* __restrict__ row_dth, int G,
                                                         const float* __restrict__ temperature, float* __restrict__ out, float* __restrict__ dtemp) {
  __shared__ float red[16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < G; i += 256) { a += row_loss[i]; b += row_dth[i]; }
  const float ta = block_sum(a, red);
  __syncthreads();
  const float tb = block_sum(b, red);
  if (threadIdx.x == 0) { out[0] = ta / (2.f * G); out[1] = __expf(temperature[0]); if (dtemp) dtemp[0] += tb; }
}
// d raw = inv * (du - u <u, du>), u = raw * inv (backward of F.normalize).  One wave per row.
__global__ __launch_bounds__(256) void l2norm_bwd_rows_kernel(const float* __restrict__ raw, const float* __restrict__ inv, const float* __restrict__ du,
                                                              float* __restrict__ out, int rows, int cols) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float iv = inv[r];
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += raw[(int64_t)r * cols + c] * iv * du[(int64_t)r * cols + c];
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) out[(int64_t)r * cols + c] = iv * (du[(int64_t)r * cols + c] - raw[(int64_t)r * cols + c] * iv * dot);
}

__global__ __launch_bounds__(256) void accumulate_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<f32x4*>(dst)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(src)[i];
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    reinterpret_cast<f32x4*>(dst)[i] = a;
  }
}
__global__ void scale_by_scalar_kernel(float* __restrict__ x, const float* __restrict__ s, int64_t n) {
  const float f = s[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= f;
}

}  // namespace

extern "C" int64_t ctclip_visual_latent_fwd_workspace(int Bm, int N, int64_t K) { return cdiv(K, VL_KCH) * (int64_t)Bm * N * 4; }
// Y (Bm, N) f32 = X (Bm, K) W^T (N, K) (overwritten).  workspace >= ctclip_visual_latent_fwd_workspace(Bm, N, K): the k-chunk partial
// sums, combined in chunk order (deterministic).
extern "C" int ctclip_visual_latent_fwd(const void* X, const void* W, float* Y, int Bm, int N, int64_t K, int dtype, void* workspace,
                                        int64_t workspace_bytes, hipStream_t s) {
  if (!X || !W || !Y || Bm < 1 || Bm > VL_MAXB || K % 8) { ctclip_set_error("visual_latent_fwd: batch must be 1..8 per call and K a multiple of 8"); return CTCLIP_EBADARG; }
  if (!workspace || workspace_bytes < ctclip_visual_latent_fwd_workspace(Bm, N, K)) { ctclip_set_error("visual_latent_fwd: workspace too small"); return CTCLIP_EWORKSPACE; }
  const int nchunk = (int)cdiv(K, VL_KCH);
  dim3 grid((unsigned)cdiv(N, 16), (unsigned)nchunk);
  const size_t shm = (size_t)Bm * VL_KCH * sizeof(float);
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute((const void*)vlat_fwd_kernel<float, VL_FWD_RG>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)vlat_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    raised = true;
  }
  float* part = (float*)workspace;
  if (dtype == DT_F32) hipLaunchKernelGGL((vlat_fwd_kernel<float, VL_FWD_RG>), dim3((unsigned)cdiv(N, 16 * VL_FWD_RG), (unsigned)nchunk), dim3(256), shm, s, (const float*)X, (const float*)W, part, Bm, N, K);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(vlat_fwd_kernel<bf16_t>, grid, dim3(256), shm, s, (const bf16_t*)X, (const bf16_t*)W, part, Bm, N, K);
  else return CTCLIP_EUNSUPPORTED;
  hipLaunchKernelGGL(vlat_reduce_kernel, dim3((unsigned)cdiv(Bm * N, 256)), dim3(256), 0, s, (const float*)part, nchunk, Bm * N, Y);
  return ctclip_check_launch("visual_latent_fwd");
}
// dX (Bm, K) in `dtype` (may be null), dW (N, K) f32 overwritten or accumulated (may be null).  Bm 1..8 per call, 1..24 in f32.
extern "C" int ctclip_visual_latent_bwd(const float* dY, const void* X, const void* W, void* dX, float* dW, int Bm, int N, int64_t K,
                                        int accumulate, int dtype, hipStream_t s) {
  const bool wide = Bm > VL_MAXB;
  if (!dY || !X || !W || Bm < 1 || Bm > VLW_MAXB || K % 8 || (int64_t)Bm * N * 4 > 64 * 1024 || (wide && dtype != DT_F32)) {
    ctclip_set_error("visual_latent_bwd: batch must be 1..8 per call (1..24 in f32), K a multiple of 8, Bm x N x 4 <= 64 KiB");
    return CTCLIP_EBADARG;
  }
  const size_t shm = (size_t)Bm * N * sizeof(float);
  if (wide) {
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute((const void*)vlat_bwd_kernel<float, false, false, VLW_MAXB, VLW_KW, VLW_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipFuncSetAttribute((const void*)vlat_bwd_kernel<float, true, true, VLW_MAXB, VLW_KW, VLW_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipFuncSetAttribute((const void*)vlat_bwd_kernel<float, false, true, VLW_MAXB, VLW_KW, VLW_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      raised = true;
    }
    dim3 grid((unsigned)cdiv(K / VLW_KW, VLW_THREADS));
#define VLW(ACC, WDW) hipLaunchKernelGGL((vlat_bwd_kernel<float, ACC, WDW, VLW_MAXB, VLW_KW, VLW_THREADS>), grid, dim3(VLW_THREADS), shm, s, dY, (const float*)X, (const float*)W, (float*)dX, dW, Bm, N, K)
    if (!dW) VLW(false, false); else if (accumulate) VLW(true, true); else VLW(false, true);
#undef VLW
    return ctclip_check_launch("visual_latent_bwd (wide)");
  }
  dim3 grid((unsigned)cdiv(K / 8, 128));
#ifndef VL_F32_KW
#define VL_F32_KW 4          // f32 operands: k per lane / weight rows in flight per lane of the 8-row backward (8 / 4: 576 waves for the whole chip)
#endif
#ifndef VL_F32_UNR
#define VL_F32_UNR 16
#endif
  if (dtype == DT_F32 && VL_F32_KW != 8) {
    dim3 g4((unsigned)cdiv(K / VL_F32_KW, 128));
#define VLF(ACC, WDW, U) hipLaunchKernelGGL((vlat_bwd_kernel<float, ACC, WDW, VL_MAXB, VL_F32_KW, 128, U>), g4, dim3(128), shm, s, dY, (const float*)X, (const float*)W, (float*)dX, dW, Bm, N, K)
    // (same box, us: overwrite 383 at 8 k x 4 rows in flight per lane, 343 at 4 x 4, 297 at 4 x 8, 275 at 4 x 16; accumulate 424 / 365 / 400 / 415)
    if (!dW) VLF(false, false, VL_F32_UNR); else if (accumulate) VLF(true, true, 4); else VLF(false, true, VL_F32_UNR);
#undef VLF
    return ctclip_check_launch("visual_latent_bwd");
  }
#define VLB(T, ACC, WDW) hipLaunchKernelGGL((vlat_bwd_kernel<T, ACC, WDW>), grid, dim3(128), shm, s, dY, (const T*)X, (const T*)W, (T*)dX, dW, Bm, N, K)
#define VLB_T(T) do { if (!dW) VLB(T, false, false); else if (accumulate) VLB(T, true, true); else VLB(T, false, true); } while (0)
  if (dtype == DT_F32) VLB_T(float);
  else if (dtype == DT_BF16) VLB_T(bf16_t);
#undef VLB_T
#undef VLB
  else return CTCLIP_EUNSUPPORTED;
  return ctclip_check_launch("visual_latent_bwd");
}
// CLIP symmetric InfoNCE forward + backward in ONE launch.  G <= 128, Dl <= 1024 (the G x G logits live in LDS: 16 ranks x batch 8);
// larger gathered batches: ctclip_clip_loss_logits between f32 ctclip_gemm launches (backend._clip_loss_large).  out: [loss, exp(temperature)].
extern "C" int ctclip_clip_loss(const float* text_latents, const float* image_latents, const float* temperature, float* out, float* logits,
                                float* d_text, float* d_image, float* d_temperature, int G, int Dl, hipStream_t s) {
  if (!text_latents || !image_latents || !temperature || !out || G < 1 || G > 128 || Dl > 1024) { ctclip_set_error("clip_loss: G <= 128 and Dl <= 1024 required"); return CTCLIP_EBADARG; }
  const size_t shm = ((size_t)G * G + 4 * G + 16) * sizeof(float);
  if (shm > 48 * 1024) {
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute((const void*)clip_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); raised = true; }
  }
  hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(256), shm, s, text_latents, image_latents, temperature, out, logits, d_text, d_image, d_temperature, G, Dl);
  return ctclip_check_launch("clip_loss");
}
// The middle of the CLIP loss for ANY gathered batch size (ct_clip.py:845-901 with G = world size x batch beyond the single-block
// kernel's 128): S (G, G) f32 holds the cosines <u_t, u_v> of the l2-normalised latents on entry (the host's f32 ctclip_gemm) and
// exp(temperature) * d loss / d logits on return (the factor both latent-gradient GEMMs need); out = [loss, exp(temperature)];
// d_temperature (+=) = sum dS * logits.  lds = row stride of S in elements (>= G; a multiple of 4 keeps the rows 16-byte aligned for the
// GEMMs).  workspace >= 4 * G floats.
extern "C" int ctclip_clip_loss_logits(float* S, int64_t lds, const float* temperature, float* out, float* d_temperature, int G, float* workspace,
                                       int64_t workspace_bytes, hipStream_t s) {
  if (!S || lds < G || !temperature || !out || G < 1 || !workspace || workspace_bytes < (int64_t)4 * G * 4) { ctclip_set_error("clip_loss_logits: bad args / workspace < 16 G bytes"); return CTCLIP_EBADARG; }
  float *rsum = workspace, *csum = rsum + G, *row_loss = csum + G, *row_dth = row_loss + G;
  hipLaunchKernelGGL(clip_rowcol_kernel, dim3((unsigned)((G + 3) / 4 + (G + 255) / 256)), dim3(256), 0, s, S, lds, temperature, G, rsum, csum);
  hipLaunchKernelGGL(clip_ds_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, s, S, lds, temperature, G, rsum, csum, row_loss, row_dth);
  hipLaunchKernelGGL(clip_final_kernel, dim3(1), dim3(256), 0, s, row_loss, row_dth, G, temperature, out, d_temperature);
  return ctclip_check_launch("clip_loss_logits");
}
// Backward of F.normalize on rows (ct_clip.py:49-50,771): out = inv * (du - u <u, du>) with u = raw * inv; all f32, (rows, cols) contiguous.
extern "C" int ctclip_l2norm_bwd_rows(const float* raw, const float* inv, const float* du, float* out, int rows, int cols, hipStream_t s) {
  if (!raw || !inv || !du || !out || rows < 1 || cols < 1) { ctclip_set_error("l2norm_bwd_rows: bad args"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(l2norm_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, raw, inv, du, out, rows, cols);
  return ctclip_check_launch("l2norm_bwd_rows");
}
// dst[i] += src[i] (f32, n % 4 == 0, 16-byte aligned): row blocks of a stacked weight gradient into the flat gradient buffer
extern "C" int ctclip_accumulate_f32(float* dst, const float* src, int64_t n, hipStream_t s) {
  if (!dst || !src || n < 0 || n % 4 || (reinterpret_cast<uintptr_t>(dst) % 16) || (reinterpret_cast<uintptr_t>(src) % 16)) { ctclip_set_error("accumulate_f32: n % 4 == 0, 16-byte aligned"); return CTCLIP_EBADARG; }
  if (n == 0) return CTCLIP_OK;
  int64_t b = cdiv(n / 4, 256); if (b > 4096) b = 4096;
  hipLaunchKernelGGL(accumulate_f32_kernel, dim3((unsigned)b), dim3(256), 0, s, dst, src, n / 4);
  return ctclip_check_launch("accumulate_f32");
}
extern "C" int ctclip_scale_by_scalar(float* x, const float* scalar, int64_t n, hipStream_t s) {
  int64_t b = cdiv(n, 256); if (b > 4096) b = 4096;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)b), dim3(256), 0, s, x, scalar, n);
  return ctclip_check_launch("scale_by_scalar");
}
