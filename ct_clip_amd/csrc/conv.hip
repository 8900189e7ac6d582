// PEG: depthwise 3x3x3 Conv3d over a channels-last (B, D1, D2, D3, C) token grid with causal padding on D1
// (F.pad(x,(1,1,1,1,2,0)) then nn.Conv3d(C, C, 3, groups=C) -- attention.py:56-84), fused with the residual
// add of Transformer.forward (attention.py:324).  HBM-bound streaming: 8 channels (16 B of bf16) per lane,
// neighbouring taps come from L1/L2, weights staged per block in LDS as [27][64 channels].
#include "common.h"

namespace {

constexpr int CCH = 64;  // channels per block

// forward (dir = +1):  y = x + bias + sum_tap w[c][tap] * x[a + d1 - 2, b + d2 - 1, g + d3 - 1]
// grad-in (dir = -1):  y = x +        sum_tap w[c][tap] * x[a - d1 + 2, b - d2 + 1, g - d3 + 1]   (x = dy)
template <typename T>
__global__ __launch_bounds__(256) void peg_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  T* __restrict__ y, int64_t npos, int D1, int D2, int D3, int C, int dir) {
  __shared__ float ws[27][CCH];
  const int c0 = blockIdx.y * CCH;
  for (int i = threadIdx.x; i < 27 * CCH; i += 256) {
    const int tap = i / CCH, cc = i % CCH;
    ws[tap][cc] = (c0 + cc < C) ? w[(int64_t)(c0 + cc) * 27 + tap] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 7;
  const int64_t pos = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int ch = c0 + cg * 8;
  if (pos >= npos || ch >= C) return;
  const int g = pos % D3; const int bb = (pos / D3) % D2; const int a = (pos / ((int64_t)D3 * D2)) % D1;
  const int64_t batch = pos / ((int64_t)D3 * D2 * D1);
  float acc[8], xin[8];
  load8(x + pos * C + ch, xin);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = xin[e] + ((dir > 0 && bias) ? bias[ch + e] : 0.f);
  // all 27 neighbour loads are UNCONDITIONAL (coordinates clamped, out-of-range taps weighted by 0): branches around the
  // loads would make the compiler wait for each one in turn.  Only the innermost 3 taps are unrolled: 27 loads in flight
  // cost 216 VGPRs and the occupancy that hides the latency.
#pragma unroll 1
  for (int d1 = 0; d1 < 3; ++d1) {
    const int aa = a + dir * (d1 - 2);
    const bool ok1 = aa >= 0 && aa < D1;
    const int aac = aa < 0 ? 0 : (aa >= D1 ? D1 - 1 : aa);
#pragma unroll 1
    for (int d2 = 0; d2 < 3; ++d2) {
      const int b2 = bb + dir * (d2 - 1);
      const bool ok2 = ok1 && b2 >= 0 && b2 < D2;
      const int b2c = b2 < 0 ? 0 : (b2 >= D2 ? D2 - 1 : b2);
#pragma unroll
      for (int d3 = 0; d3 < 3; ++d3) {
        const int g2 = g + dir * (d3 - 1);
        const bool ok = ok2 && g2 >= 0 && g2 < D3;
        const int g2c = g2 < 0 ? 0 : (g2 >= D3 ? D3 - 1 : g2);
        float v[8];
        load8(x + ((((batch * D1 + aac) * D2 + b2c) * D3) + g2c) * C + ch, v);
        const int tap = (d1 * 3 + d2) * 3 + d3;
        const float m = ok ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (ws[tap][cg * 8 + e] * m) * v[e];
      }
    }
  }
  store8(y + pos * C + ch, acc);
}

// dw[c][tap] += sum_pos dy[pos,c] * x[pos shifted by tap, c] ;  db[c] += sum_pos dy[pos,c]
// grid: (position chunks of 1024, channel chunks of 64, d1 = 0..2)
template <typename T>
__global__ __launch_bounds__(256) void peg_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw,
                                                        float* __restrict__ db, int64_t npos, int D1, int D2, int D3, int C) {
  __shared__ float red[10][CCH];
  const int c0 = blockIdx.y * CCH, d1 = blockIdx.z;
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int ch = c0 + cg * 8;
  float acc[9][8], accb[8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) accb[e] = 0.f;
  const int64_t pbeg = (int64_t)blockIdx.x * 1024;
  if (ch < C) {
    for (int64_t pos = pbeg + pl; pos < pbeg + 1024 && pos < npos; pos += 32) {
      const int g = pos % D3; const int bb = (pos / D3) % D2; const int a = (pos / ((int64_t)D3 * D2)) % D1;
      const int64_t batch = pos / ((int64_t)D3 * D2 * D1);
      float gy[8];
      load8(dy + pos * C + ch, gy);
      if (d1 == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) accb[e] += gy[e];
      }
      const int aa = a + d1 - 2;
      const bool ok1 = aa >= 0;
      const int aac = aa < 0 ? 0 : aa;
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) {
        const int b2 = bb + d2 - 1;
        const bool ok2 = ok1 && b2 >= 0 && b2 < D2;
        const int b2c = b2 < 0 ? 0 : (b2 >= D2 ? D2 - 1 : b2);
#pragma unroll
        for (int d3 = 0; d3 < 3; ++d3) {
          const int g2 = g + d3 - 1;
          const bool ok = ok2 && g2 >= 0 && g2 < D3;
          const int g2c = g2 < 0 ? 0 : (g2 >= D3 ? D3 - 1 : g2);
          float v[8];
          load8(x + ((((batch * D1 + aac) * D2 + b2c) * D3) + g2c) * C + ch, v);   // unconditional, masked below
          const float m = ok ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[d2 * 3 + d3][e] += (gy[e] * m) * v[e];
        }
      }
    }
  }
  for (int i = threadIdx.x; i < 10 * CCH; i += 256) red[i / CCH][i % CCH] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&red[t][cg * 8 + e], acc[t][e]);
  if (d1 == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&red[9][cg * 8 + e], accb[e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 10 * CCH; i += 256) {
    const int t = i / CCH, cc = i % CCH;
    if (c0 + cc >= C) continue;
    if (t < 9) atomicAdd(dw + (int64_t)(c0 + cc) * 27 + d1 * 9 + t, red[t][cc]);
    else if (d1 == 0 && db) atomicAdd(db + c0 + cc, red[9][cc]);
  }
}

}  // namespace

// y = PEG(x) + x   (attention.py:63-84 + the residual of attention.py:324).  w: (C,27) f32 = dsconv.weight, bias (C) f32.
extern "C" int ctclip_peg_fwd(const void* x, const float* w, const float* bias, void* y, int64_t B, int D1, int D2, int D3, int C,
                              int dtype, hipStream_t stream) {
  if (!x || !w || !y || C % 8) { ctclip_set_error("peg_fwd: C must be a multiple of 8"); return CTCLIP_EBADARG; }
  const int64_t npos = B * D1 * D2 * D3;
  dim3 grid((unsigned)cdiv(npos, 32), (unsigned)cdiv(C, CCH));
  if (dtype == DT_F32) hipLaunchKernelGGL(peg_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, w, bias, (float*)y, npos, D1, D2, D3, C, 1);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(peg_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, w, bias, (bf16_t*)y, npos, D1, D2, D3, C, 1);
  else return CTCLIP_EUNSUPPORTED;
  return ctclip_check_launch("peg_fwd");
}

// dx = dy + conv^T(dy) ; dw (C,27) and db (C) f32 are ACCUMULATED (+=) when non-null.
extern "C" int ctclip_peg_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* db, int64_t B, int D1, int D2,
                              int D3, int C, int dtype, hipStream_t stream) {
  if (!dy || !x || !w || !dx || C % 8) { ctclip_set_error("peg_bwd: bad args"); return CTCLIP_EBADARG; }
  const int64_t npos = B * D1 * D2 * D3;
  dim3 grid((unsigned)cdiv(npos, 32), (unsigned)cdiv(C, CCH));
  dim3 gridw((unsigned)cdiv(npos, 1024), (unsigned)cdiv(C, CCH), 3);
  if (dtype == DT_F32) {
    hipLaunchKernelGGL(peg_kernel<float>, grid, dim3(256), 0, stream, (const float*)dy, w, nullptr, (float*)dx, npos, D1, D2, D3, C, -1);
    if (dw) hipLaunchKernelGGL(peg_wgrad_kernel<float>, gridw, dim3(256), 0, stream, (const float*)dy, (const float*)x, dw, db, npos, D1, D2, D3, C);
  } else if (dtype == DT_BF16) {
    hipLaunchKernelGGL(peg_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)dy, w, nullptr, (bf16_t*)dx, npos, D1, D2, D3, C, -1);
    if (dw) hipLaunchKernelGGL(peg_wgrad_kernel<bf16_t>, gridw, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, dw, db, npos, D1, D2, D3, C);
  } else return CTCLIP_EUNSUPPORTED;
  return ctclip_check_launch("peg_bwd");
}
