// PEG: depthwise 3x3x3 Conv3d over a channels-last (B, D1, D2, D3, C) token grid with causal padding on D1
// (F.pad(x,(1,1,1,1,2,0)) then nn.Conv3d(C, C, 3, groups=C) -- attention.py:56-84), fused with the residual
// add of Transformer.forward (attention.py:324).  HBM-bound streaming: 4 channels per lane, a walk along the innermost grid axis
// with three rotating output accumulators, neighbouring rows from L1/L2, weights per block in LDS as [27 + 3 zero rows][64 ch].
#include "common.h"
#include "peg_lds.h"
#include <type_traits>

namespace {

constexpr int CCH = 64;  // channels per block = one 128-byte line of bf16
#ifndef PEG_MINB
#define PEG_MINB 2
#endif

// Scatter formulation along the innermost grid axis.  A thread owns one grid row (batch, a, b) and FOUR channels (8 bytes of
// bf16: sixteen lanes cover one 128-byte line) and walks q = 0 .. D3-1.  At each step it loads the nine values x[row_k, q] (one per
// (d1, d2) neighbour row), converts them ONCE and adds them, weighted, into the three outputs they touch (q-1, q, q+1), which live
// in three rotating accumulators.  No input window is kept (the gather form needs 27 x 4 live f32 values = 108 registers and
// ran at 2 waves per SIMD), a step costs 9 loads + 9 conversions + 27 FMAs per channel instead of the 27 loads, 27 conversions and
// 54 multiplies of the first version (213 us per call for 113 MB in + 113 MB out).  Out-of-range neighbour rows read a clamped
// (valid) address and take their weights from three all-zero rows of the LDS weight table -- no per-element masking.
template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> {
  u32x2 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const u32x2*>(p); }
  __device__ __forceinline__ void unpack(float (&f)[4]) const {
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
  }
};
template <> struct Raw4<float> {
  f32x4 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
  __device__ __forceinline__ void unpack(float (&f)[4]) const { f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
};

// forward (DIR = +1):  y = x + bias + sum_tap w[c][tap] * x[a + d1 - 2, b + d2 - 1, g + d3 - 1]
// grad-in (DIR = -1):  y = x +        sum_tap w[c][tap] * x[a - d1 + 2, b - d2 + 1, g - d3 + 1]   (x = dy)
template <typename T, int DIR>
__global__ __launch_bounds__(256, PEG_MINB) void peg_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     T* __restrict__ y, int64_t nrows, int D1, int D2, int D3, int C) {
  __shared__ __attribute__((aligned(16))) float ws[30][CCH];   // rows 27..29 stay zero: the taps of out-of-range neighbour rows
  // 1-D grid, channel chunk fastest: workgroups are dealt to the 8 XCDs round-robin, so with C = 512 (8 chunks) chunk i always runs
  // on XCD i and every line of x is pulled into exactly ONE L2 (the 9 threads that share it sit in different row groups).
  const int nchunk = (C + CCH - 1) / CCH;
  const int c0 = (int)(blockIdx.x % nchunk) * CCH;
  const int64_t rgroup = blockIdx.x / nchunk;
  {
    // all seven loads per thread are issued before the first use (a loop with the load inside a branch waits for each in turn:
    // ~8 serialised L2 round trips per block); w is read contiguously (64 channels x 27 taps = one 6.9-KB run)
    float wreg[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int i = threadIdx.x + q * 256;
      const int cc = (i < 27 * CCH ? i : 0) / 27;
      const int cidx = c0 + cc < C ? c0 + cc : C - 1;
      wreg[q] = w[(int64_t)cidx * 27 + (i < 27 * CCH ? i : 0) % 27];
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int i = threadIdx.x + q * 256;
      if (i < 27 * CCH) ws[i % 27][i / 27] = (c0 + i / 27 < C) ? wreg[q] : 0.f;
    }
    if (threadIdx.x < 3 * CCH) ws[27 + threadIdx.x / CCH][threadIdx.x % CCH] = 0.f;
  }
  __syncthreads();
  const int cl = (threadIdx.x & 15) * 4, ch = c0 + cl;
  const int64_t row = rgroup * 16 + (threadIdx.x >> 4);
  if (row >= nrows || ch >= C) return;
  const int bb = (int)(row % D2), a = (int)((row / D2) % D1);
  const int64_t batch = row / ((int64_t)D2 * D1);

  const T* xrow = x + row * D3 * C + ch;   // the thread's own row at q = 0
  int rofs[9];         // neighbour rows (clamped) relative to the own row, in elements
  int wofs[9];         // float offset of tap (d1, d2, d3 = 0) in ws (+ CCH per d3), or of the zero rows
#pragma unroll
  for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
    for (int d2 = 0; d2 < 3; ++d2) {
      const int aa = a + DIR * (d1 - 2), b2 = bb + DIR * (d2 - 1);
      const bool ok = aa >= 0 && aa < D1 && b2 >= 0 && b2 < D2;
      const int aac = aa < 0 ? 0 : (aa >= D1 ? D1 - 1 : aa), b2c = b2 < 0 ? 0 : (b2 >= D2 ? D2 - 1 : b2);
      rofs[d1 * 3 + d2] = (int)(((((batch * D1 + aac) * D2 + b2c) - row) * D3) * C);
      wofs[d1 * 3 + d2] = (ok ? (d1 * 3 + d2) * 3 : 27) * CCH + cl;
    }
  T* yp = y + row * D3 * C + ch;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (DIR > 0 && bias) load4(bias + ch, bv);
  const float* wsf = &ws[0][0];
  constexpr int KC = 7;              // the neighbour row that is the thread's own row: (d1, d2) = (2, 1)
  // x[q] reaches output q + 1, q, q - 1 through tap d3 = 0, 1, 2 (forward) or 2, 1, 0 (grad-in)
  constexpr int D3P = DIR > 0 ? 0 : 2, D3M = DIR > 0 ? 2 : 0;

  // loads run NST steps ahead of their use.  (NST = 2 spills at 2 waves per SIMD: the 27 x 4 weights the compiler keeps in
  // registers -- they are loop invariant -- plus two stages exceed 256.)
  constexpr int NST = 1;
  Raw4<T> staged[NST][9];
#pragma unroll
  for (int st = 0; st < NST; ++st)
#pragma unroll
    for (int k = 0; k < 9; ++k) staged[st][k].load(xrow + rofs[k] + (int64_t)(st < D3 ? st : D3 - 1) * C);
  float accm[4], acc0[4], accp[4];   // outputs q - 1, q, q + 1
#pragma unroll
  for (int e = 0; e < 4; ++e) { accm[e] = 0.f; acc0[e] = bv[e]; accp[e] = bv[e]; }
  auto step = [&](auto stage, int q) {
    constexpr int S = decltype(stage)::value;
    float xs[9][4];
    const T* xn = xrow + (int64_t)(q + NST < D3 ? q + NST : D3 - 1) * C;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      staged[S][k].unpack(xs[k]);
      staged[S][k].load(xn + rofs[k]);                    // clamped, always issued
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc0[e] += xs[KC][e];     // the residual
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const f32x4 wp = *reinterpret_cast<const f32x4*>(wsf + wofs[k] + D3P * CCH);
      const f32x4 wc = *reinterpret_cast<const f32x4*>(wsf + wofs[k] + CCH);
      const f32x4 wm = *reinterpret_cast<const f32x4*>(wsf + wofs[k] + D3M * CCH);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        accp[e] = fmaf(wp[e], xs[k][e], accp[e]);
        acc0[e] = fmaf(wc[e], xs[k][e], acc0[e]);
        accm[e] = fmaf(wm[e], xs[k][e], accm[e]);
      }
    }
    if (q > 0) store4(yp + (int64_t)(q - 1) * C, accm);
#pragma unroll
    for (int e = 0; e < 4; ++e) { accm[e] = acc0[e]; acc0[e] = accp[e]; accp[e] = bv[e]; }
  };
  for (int q = 0; q < D3; q += NST) {
    step(std::integral_constant<int, 0>{}, q);
    if (NST > 1 && q + 1 < D3) step(std::integral_constant<int, NST - 1>{}, q + 1);
  }
  store4(yp + (int64_t)(D3 - 1) * C, accm);
}

// dw[c][tap] += sum_pos dy[pos,c] * x[pos shifted by tap, c] ;  db[c] += sum_pos dy[pos,c]
// 1-D grid over (chunks of 64 grid rows, d1 = 0..2, channel chunks of 64).  Same walk: x[row_k, q] meets dy[q+1], dy[q], dy[q-1]
// (taps d3 = 0, 1, 2), which are kept in a three-value register window.
constexpr int WG_ROWS = 4;   // grid rows per thread
template <typename T>
__global__ __launch_bounds__(256) void peg_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part,
                                                        int64_t nrows, int D1, int D2, int D3, int C) {
  __shared__ float red[16][10][CCH];        // [row lane][tap | bias][channel]: summed in a fixed order (no atomics)
  const int nchunk = (C + CCH - 1) / CCH;                 // channel chunk fastest (see peg_kernel)
  const int c0 = (int)(blockIdx.x % nchunk) * CCH, d1 = (int)((blockIdx.x / nchunk) % 3);
  const int64_t rgroup = blockIdx.x / (nchunk * 3);
  const int cl = (threadIdx.x & 15) * 4, ch = c0 + cl;
  float tot[9][4], totb[4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) tot[t][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) totb[e] = 0.f;

  if (ch < C) {
    for (int ri = 0; ri < WG_ROWS; ++ri) {
      const int64_t row = (rgroup * WG_ROWS + ri) * 16 + (threadIdx.x >> 4);
      if (row >= nrows) break;
      const int bb = (int)(row % D2), a = (int)((row / D2) % D1);
      const int64_t batch = row / ((int64_t)D2 * D1);
      const int aa = a + d1 - 2;
      if (aa < 0) continue;                              // the whole neighbour plane is causal padding: contributes nothing
      const T* rp[3]; float m[3];
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) {
        const int b2 = bb + d2 - 1;
        m[d2] = (b2 >= 0 && b2 < D2) ? 1.f : 0.f;
        const int b2c = b2 < 0 ? 0 : (b2 >= D2 ? D2 - 1 : b2);
        rp[d2] = x + (((batch * D1 + aa) * D2 + b2c) * D3) * C + ch;
      }
      const T* gp = dy + row * D3 * C + ch;
      float acc[9][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
      // Loads run WST steps ahead of their use (the kernel is bound by load latency x bytes in flight, and it has registers to
      // spare): stage q % WST holds x[row_k, q] and dy[q + 1].
      constexpr int WST = 3;
      Raw4<T> staged[WST][3], gy_st[WST];
#pragma unroll
      for (int st = 0; st < WST; ++st) {
#pragma unroll
        for (int k = 0; k < 3; ++k) staged[st][k].load(rp[k] + (int64_t)(st < D3 ? st : D3 - 1) * C);
        gy_st[st].load(gp + (int64_t)(st + 1 < D3 ? st + 1 : D3 - 1) * C);
      }
      float gm[4] = {0.f, 0.f, 0.f, 0.f}, g0[4], gq[4];   // dy[q - 1], dy[q], dy[q + 1]
      { Raw4<T> r; r.load(gp); r.unpack(g0); }
      auto step = [&](auto stage, int q) {
        constexpr int S = decltype(stage)::value;
        float xs[3][4];
        const int qx = q + WST < D3 ? q + WST : D3 - 1, qg = q + 1 + WST < D3 ? q + 1 + WST : D3 - 1;
        gy_st[S].unpack(gq);
        gy_st[S].load(gp + (int64_t)qg * C);
        if (q + 1 >= D3) {                                 // wave-uniform: past the end of the row
#pragma unroll
          for (int e = 0; e < 4; ++e) gq[e] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          staged[S][k].unpack(xs[k]);
          staged[S][k].load(rp[k] + (int64_t)qx * C);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[k * 3 + 0][e] = fmaf(gq[e], xs[k][e], acc[k * 3 + 0][e]);
            acc[k * 3 + 1][e] = fmaf(g0[e], xs[k][e], acc[k * 3 + 1][e]);
            acc[k * 3 + 2][e] = fmaf(gm[e], xs[k][e], acc[k * 3 + 2][e]);
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) { accb[e] += g0[e]; gm[e] = g0[e]; g0[e] = gq[e]; }
      };
      for (int q = 0; q < D3; q += WST) {
        step(std::integral_constant<int, 0>{}, q);
        if (q + 1 < D3) step(std::integral_constant<int, 1>{}, q + 1);
        if (q + 2 < D3) step(std::integral_constant<int, 2>{}, q + 2);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[t][e] = fmaf(m[t / 3], acc[t][e], tot[t][e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) totb[e] += accb[e];
    }
  }
  {
    const int rl = threadIdx.x >> 4;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[rl][t][cl + e] = tot[t][e];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][9][cl + e] = totb[e];
  }
  __syncthreads();
  // part[rgroup][channel][28]: taps d1*9 .. d1*9+8 from this block, the bias gradient (slot 27) from the d1 = 2 block only
  // (d1 = 2 is the plane a itself: every row is visited exactly once there)
  for (int i = threadIdx.x; i < 10 * CCH; i += 256) {
    const int t = i / CCH, cc = i % CCH;
    if (c0 + cc >= C) continue;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += red[r][t][cc];
    float* dst = part + ((int64_t)rgroup * C + c0 + cc) * 28;
    if (t < 9) dst[d1 * 9 + t] = v;
    else if (d1 == 2) dst[27] = v;
  }
}

// dw[c][tap] += sum over row groups (in order) of part ; db[c] += ... slot 27
__global__ __launch_bounds__(256) void peg_wgrad_reduce_kernel(const float* __restrict__ part, int nrg, int C, float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * 28) return;
  const int c = i / 28, t = i % 28;
  float v = 0.f;
  for (int r = 0; r < nrg; ++r) v += part[((int64_t)r * C + c) * 28 + t];
  if (t < 27) dw[(int64_t)c * 27 + t] += v;
  else if (db) db[c] += v;
}

}  // namespace

// y = PEG(x) + x   (attention.py:63-84 + the residual of attention.py:324).  w: (C,27) f32 = dsconv.weight, bias (C) f32.
extern "C" int ctclip_peg_fwd(const void* x, const float* w, const float* bias, void* y, int64_t B, int D1, int D2, int D3, int C,
                              int dtype, hipStream_t stream) {
  if (!x || !w || !y || C % 8) { ctclip_set_error("peg_fwd: C must be a multiple of 8"); return CTCLIP_EBADARG; }
  if (peg_lds_supported(B, D1, D2, D3, C, dtype) && peg_lds_march(x, w, bias, y, B, D1, D2, D3, C, +1, stream) == 0) return ctclip_check_launch("peg_fwd");
  const int64_t nrows = B * D1 * D2;
  dim3 grid((unsigned)(cdiv(nrows, 16) * cdiv(C, CCH)));
  if (dtype == DT_F32) hipLaunchKernelGGL((peg_kernel<float, 1>), grid, dim3(256), 0, stream, (const float*)x, w, bias, (float*)y, nrows, D1, D2, D3, C);
  else if (dtype == DT_BF16) hipLaunchKernelGGL((peg_kernel<bf16_t, 1>), grid, dim3(256), 0, stream, (const bf16_t*)x, w, bias, (bf16_t*)y, nrows, D1, D2, D3, C);
  else return CTCLIP_EUNSUPPORTED;
  return ctclip_check_launch("peg_fwd");
}

// ctclip_peg_fwd on the COMPENSATED residual stream (see ctclip_gemm_residual_comp): s = x + e_in + conv(x) + bias in f32 (e_in may be NULL:
// zeros), y = bf16(s), e_out = bf16(s - y) the rounding residue -- the PEG residual add (attention.py:323-324) without an accumulating
// rounding.  bf16 grids the LDS-marching kernels serve only (C % 32 == 0, D3 in {8, 16, 24, 32}); CTCLIP_EUNSUPPORTED otherwise.
extern "C" int ctclip_peg_fwd_comp(const void* x, const float* w, const float* bias, const void* e_in, void* y, void* e_out, int64_t B, int D1,
                                   int D2, int D3, int C, int dtype, hipStream_t stream) {
  if (!x || !w || !y || !e_out || C % 8) { ctclip_set_error("peg_fwd_comp: bad args"); return CTCLIP_EBADARG; }
  if (!peg_lds_supported(B, D1, D2, D3, C, dtype)) return CTCLIP_EUNSUPPORTED;
  if (peg_lds_march(x, w, bias, y, B, D1, D2, D3, C, +1, stream, e_in, e_out) != 0) return CTCLIP_EUNSUPPORTED;
  return ctclip_check_launch("peg_fwd_comp");
}

// bytes of workspace ctclip_peg_bwd needs when dw is requested (per-row-group partial weight gradients)
extern "C" int64_t ctclip_peg_bwd_workspace(int64_t B, int D1, int D2, int C) {
  const int64_t g1 = cdiv(B * D1 * D2, 16 * WG_ROWS), g2 = peg_lds_wgrad_groups(B, D2, C);
  return (g1 > g2 ? g1 : g2) * C * 28 * 4;
}

// dx = dy + conv^T(dy) ; dw (C,27) and db (C) f32 are ACCUMULATED (+=) when non-null (two stages, fixed summation order).
// dx may be NULL (weight gradient only) and dw may be NULL (grad-input only): the host launches the two halves on different streams.
extern "C" int ctclip_peg_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* db, int64_t B, int D1, int D2,
                              int D3, int C, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (!dy || !x || !w || (!dx && !dw) || C % 8) { ctclip_set_error("peg_bwd: bad args"); return CTCLIP_EBADARG; }
  if (dw && (!workspace || workspace_bytes < ctclip_peg_bwd_workspace(B, D1, D2, C))) { ctclip_set_error("peg_bwd: workspace too small"); return CTCLIP_EWORKSPACE; }
  if (peg_lds_supported(B, D1, D2, D3, C, dtype)) {
    // marching kernels: grad-in, then the weight gradient (per-workgroup partials + the same ordered second stage)
    int groups = 0;
    const bool dx_done = !dx || peg_lds_march(dy, w, nullptr, dx, B, D1, D2, D3, C, -1, stream) == 0;
    const bool dw_done = dx_done && dw && peg_lds_wgrad(dy, x, (float*)workspace, B, D1, D2, D3, C, &groups, stream) == 0;
    if (dw_done) hipLaunchKernelGGL(peg_wgrad_reduce_kernel, dim3((unsigned)cdiv(C * 28, 256)), dim3(256), 0, stream, (const float*)workspace, groups, C, dw, db);
    if (dx_done && (dw_done || !dw)) return ctclip_check_launch("peg_bwd");
    if (dx_done) {          // grad-in done, weight gradient on the first-generation kernel
      const int64_t nrows = B * D1 * D2;
      dim3 gridw((unsigned)(cdiv(nrows, 16 * WG_ROWS) * cdiv(C, CCH) * 3));
      hipLaunchKernelGGL(peg_wgrad_kernel<bf16_t>, gridw, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, (float*)workspace, nrows, D1, D2, D3, C);
      hipLaunchKernelGGL(peg_wgrad_reduce_kernel, dim3((unsigned)cdiv(C * 28, 256)), dim3(256), 0, stream, (const float*)workspace,
                         (int)cdiv(nrows, 16 * WG_ROWS), C, dw, db);
      return ctclip_check_launch("peg_bwd");
    }
  }
  const int64_t nrows = B * D1 * D2;
  dim3 grid((unsigned)(cdiv(nrows, 16) * cdiv(C, CCH)));
  dim3 gridw((unsigned)(cdiv(nrows, 16 * WG_ROWS) * cdiv(C, CCH) * 3));
  if (dtype == DT_F32) {
    if (dx) hipLaunchKernelGGL((peg_kernel<float, -1>), grid, dim3(256), 0, stream, (const float*)dy, w, nullptr, (float*)dx, nrows, D1, D2, D3, C);
    if (dw) hipLaunchKernelGGL(peg_wgrad_kernel<float>, gridw, dim3(256), 0, stream, (const float*)dy, (const float*)x, (float*)workspace, nrows, D1, D2, D3, C);
  } else if (dtype == DT_BF16) {
    if (dx) hipLaunchKernelGGL((peg_kernel<bf16_t, -1>), grid, dim3(256), 0, stream, (const bf16_t*)dy, w, nullptr, (bf16_t*)dx, nrows, D1, D2, D3, C);
    if (dw) hipLaunchKernelGGL(peg_wgrad_kernel<bf16_t>, gridw, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, (float*)workspace, nrows, D1, D2, D3, C);
  } else return CTCLIP_EUNSUPPORTED;
  if (dw) hipLaunchKernelGGL(peg_wgrad_reduce_kernel, dim3((unsigned)cdiv(C * 28, 256)), dim3(256), 0, stream, (const float*)workspace,
                             (int)cdiv(nrows, 16 * WG_ROWS), C, dw, db);
  return ctclip_check_launch("peg_bwd");
}
