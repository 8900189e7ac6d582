// Common device helpers for the CT-CLIP gfx950 (CDNA4 / MI355X) kernels.
// Written for wave64 + MFMA directly; no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CTCLIP_OK 0
#define CTCLIP_EBADARG (-1)
#define CTCLIP_EUNSUPPORTED (-2)
#define CTCLIP_EWORKSPACE (-3)

enum { DT_F32 = 0, DT_BF16 = 1 };

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

extern "C" void ctclip_set_error(const char* msg);
int ctclip_check_launch(const char* what);
// Device-resident step state (capi.hip: ctclip_set_step_state): null, or two 64-bit words { seed offset, optimiser step } that the dropout /
// attention-dropout / Adam kernels read AT RUN TIME -- what lets a captured hipGraph of the training step draw fresh dropout masks and use the
// right bias correction on every replay (scalars passed by value are frozen into the graph).
const unsigned long long* ctclip_step_state(void);

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16 through the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserving).  The integer
// emulation of the first version cost ~6 VALU per element and made every bf16 store path (GEMM epilogues above all)
// VALU-bound: ~1300 VALU per wave per 256 x 256 tile = as long as the K = 512 main loop.
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = DT_F32;
  static constexpr int kPer16B = 4;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kDtype = DT_BF16;
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 8 consecutive elements <-> 8 floats (vectorised: 16 B for bf16, 2 x 16 B for f32). Pointers 16-B aligned.
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(a[i] << 16); v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u); }
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = a;
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = a[i];
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  const u32x2 a = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(a[0] << 16); v[1] = __uint_as_float(a[0] & 0xffff0000u);
  v[2] = __uint_as_float(a[1] << 16); v[3] = __uint_as_float(a[1] & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  f32x4 a; a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
  *reinterpret_cast<f32x4*>(p) = a;
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  u32x2 a; a[0] = pack2bf(v[0], v[1]); a[1] = pack2bf(v[2], v[3]);
  *reinterpret_cast<u32x2*>(p) = a;
}

// wave64 reductions (all 64 lanes participate; every lane gets the result).  Round 6: on the vector unit only -- lanes 1, 2 apart by DPP quad
// permutations, 4 and 8 apart by the DPP half-row / row mirrors (the partners then hold equal partial results: the mirror IS the butterfly
// partner's value), rows 16 and 32 apart by gfx950's v_permlane16_swap / v_permlane32_swap.  __shfl_xor is a ds_bpermute round trip through
// the LDS crossbar: six of them per reduction were ~400 clocks of latency in every row of the LayerNorm kernels.  Fixed order: deterministic.
template <int CTRL> __device__ __forceinline__ float dpp_lanes(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_lanes<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v += dpp_lanes<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  v += dpp_lanes<0x141>(v);      // row_half_mirror
  v += dpp_lanes<0x140>(v);      // row_mirror
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_lanes<0xB1>(v));
  v = fmaxf(v, dpp_lanes<0x4E>(v));
  v = fmaxf(v, dpp_lanes<0x141>(v));
  v = fmaxf(v, dpp_lanes<0x140>(v));
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// v (+ | max) the value of the lane 32 away (the two halves of a wave hold the two half-rows of a 32 x 32 MFMA tile): one v_permlane32_swap,
// no LDS round trip (was __shfl_xor(v, 32, 64) = ds_bpermute, in the online-softmax chain of every attention tile)
__device__ __forceinline__ float half_sum(float v) {
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float half_max(float v) {
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// sum over the aligned group of 4 lanes (xor 1, xor 2) / of 32 lanes (xor 1 ... 16): the ascending butterfly's tree, on the vector unit
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_lanes<0xB1>(v);
  return v + dpp_lanes<0x4E>(v);
}
__device__ __forceinline__ float half32_sum(float v) {
  v = quad_sum(v);
  v += dpp_lanes<0x141>(v);
  v += dpp_lanes<0x140>(v);
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
// block reductions for blockDim.x a multiple of 64 (<= 1024); `red` = >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for GEMM epilogues: Abramowitz-Stegun 7.1.26 (|error of erf| < 1.5e-7, i.e. exact after rounding to bf16), branch-free,
// ~16 plain instructions + one v_rcp_f32 + one v_exp_f32 (erff from the device library is ~2x that, with branches)
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float s = x * 0.70710678118654752440f, a = fabsf(s);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f); poly = fmaf(poly, t, -0.284496736f); poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f);
  const float erf_abs = fmaf(-poly * t, e, 1.f);
  return 0.5f * x * (1.f + copysignf(erf_abs, s));
}
// gelu(x) = x Phi(x) with the normal tail Phi(-a) = 2^p(a), p a degree-6 polynomial fitted to log2(erfc(a / sqrt 2) / 2) on [0, 6.5] (minimax by
// iterated reweighting; |a| is clamped there: Phi(-6.5) = 4e-11): relative error of gelu <= 7.7e-5, absolute <= 1.1e-5 in f32 -- a fiftieth of a
// bf16 ulp -- for seven FMAs-and-friends and ONE v_exp_f32 (gelu_erf_fast: ~14 plain operations, v_rcp_f32 and v_exp_f32: the GELU of the
// feed-forward in-projection's epilogue cost 55 of its 430 us).  Forward epilogues only; the backward keeps gelu_erf_fast_both.
__device__ __forceinline__ float gelu_tail_fast(float x) {
  const float a = fminf(fabsf(x), 6.5f);
  float p = fmaf(2.0085737560293637e-05f, a, -0.0005614986293949187f);
  p = fmaf(p, a, 0.00688051525503397f); p = fmaf(p, a, -0.050271984189748764f); p = fmaf(p, a, -0.4624553918838501f);
  p = fmaf(p, a, -1.1496199369430542f); p = fmaf(p, a, -1.000106692314148f);
  const float q = __builtin_amdgcn_exp2f(p);
  return x * (x < 0.f ? q : 1.f - q);
}
// the same evaluation returning both gelu(x) and d gelu / dx = Phi(x) + x phi(x) (the exponential is shared: phi(x) = e / sqrt(2 pi))
__device__ __forceinline__ void gelu_erf_fast_both(float x, float& y, float& dy) {
  const float s = x * 0.70710678118654752440f, a = fabsf(s);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f); poly = fmaf(poly, t, -0.284496736f); poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f);
  const float erf_abs = fmaf(-poly * t, e, 1.f);
  const float cdf = 0.5f * (1.f + copysignf(erf_abs, s));
  y = x * cdf;
  dy = fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Counter-based random numbers for dropout: Philox4x32-10 (Salmon et al., SC'11; the generator curand / torch use) keyed by
// the 64-bit seed, counter = (index lo, index hi, stream id, 0).  The mask is a pure function of (seed, stream, element index):
// the backward pass regenerates it instead of storing it, and tests/ref_backend.py reproduces it bit for bit in torch.
__device__ __forceinline__ u32x4 philox4x32(uint64_t seed, uint64_t index, uint32_t stream) {
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32), c2 = stream, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}
// keep-multiplier of one element: 0 with probability p, 1 / (1 - p) otherwise (u = top 24 bits / 2^24 >= p keeps)
__device__ __forceinline__ float dropout_mult(uint32_t word, float p, float inv_keep) { return (float)(word >> 8) * (1.f / 16777216.f) >= p ? inv_keep : 0.f; }

// Attention dropout, second form (round 6).  The first form drew ONE Philox call per probability (word 0 of philox(seed, linear index)): 40 quarter-rate
// integer multiplies per score -- at BERT's T = 512 the three attention kernels spent > 90 % of their time in it (111 / 172 / 196 us per layer).  Now one
// call decides a BLOCK of 2 queries x 4 keys with 16 bits each: block (sh = sequence * H + head, qi >> 1, kj >> 2) -> philox(seed, block index, stream 1);
// element (qi & 1, kj & 3) = half (kj & 1) of word (qi & 1) * 2 + ((kj & 3) >> 1); kept when its 16 bits >= round(p * 65536).  The shape serves both
// register layouts: a lane of the forward / dQ kernels holds runs of 8 keys of ONE query (2 blocks per run), a lane of the dK / dV kernel runs of 8
// queries of ONE key (4 blocks per run).  A pure function of (seed, sh, qi, kj): ctclip_attn_dropout_mask and tests/ref_backend.py reproduce it.
__device__ __forceinline__ u32x4 attn_drop_block(uint64_t seed, int64_t sh, int L, int qi, int kj) {
  const uint64_t nq2 = (uint64_t)((L + 1) >> 1), nk4 = (uint64_t)((L + 3) >> 2);
  return philox4x32(seed, ((uint64_t)sh * nq2 + (uint64_t)(qi >> 1)) * nk4 + (uint64_t)(kj >> 2), 1u);
}
__device__ __forceinline__ uint32_t attn_drop_threshold(float p) { return (uint32_t)(p * 65536.f + 0.5f); }
__device__ __forceinline__ float attn_drop_pick(const u32x4& w, int qi, int kj, uint32_t thr16, float inv_keep) {
  const int wi = (qi & 1) * 2 + ((kj & 3) >> 1);
  const uint32_t word = wi == 0 ? w[0] : (wi == 1 ? w[1] : (wi == 2 ? w[2] : w[3]));
  const uint32_t u = (kj & 1) ? (word >> 16) : (word & 0xffffu);
  return u >= thr16 ? inv_keep : 0.f;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
