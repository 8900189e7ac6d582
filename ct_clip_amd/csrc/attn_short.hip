// Cosine-similarity attention (attention.py:145-178: l2norm(q) * q_scale, l2norm(k) * k_scale, sim * scale, softmax, @ v) for SHORT
// sequences: L <= 32 tokens, d_head = 32, no bias, no mask, bf16 -- CTViT's temporal transformer (4608 sequences x 24 frames x 8 heads
// per volume batch of 8; ctvit.py:205-206, attention.py:280-333).
//
// The general kernels spend this case on layout passes (qk-norm, head transposes, delta, a 128-key tile loop): 615 us per layer
// forward + backward.  A whole (sequence, head) problem is ONE 32 x 32 MFMA tile, so here one wave owns one problem end to end:
//   * it reads the token-major q (M, H*32) and kv (M, 2*H*32) rows directly and writes o / dq / dkv directly: 226 MB forward,
//     ~400 MB backward per layer, nothing else touches HBM (no transposed copies, no lse, no delta, no planar operands);
//   * "layout R": lane (row = lane & 31, half = lane >> 5) holds the 16 head dims 8 j + 4 half + r of its token row (four 8-byte
//     loads).  That is at once the MFMA operand order for contractions over d (slot (t, i) <-> d = 8 (2t + i/4) + 4 half + i%4, the
//     same map on both operands) and the accumulator order of a 32 x 32 result column, so l2norm forward / backward, the softmax
//     and the learned-scale gradients are lane-local plus ONE exchange with lane ^ 32;
//   * contractions over tokens take their B operand straight from the accumulators (P, dS) and their A operand (V^T, K^T, Q^T, dO^T)
//     from a 2-KB row-major LDS tile private to the wave through ds_read_b64_tr_b16; no barriers anywhere;
//   * backward recomputes the 32 x 32 softmax in both orientations (queries as columns for dQ, keys as columns for dK / dV) and
//     passes the row statistics between them through 384 bytes of LDS; delta = rowsum(P o dP) (no O needed);
//   * the gradients of q_scale / k_scale accumulate in registers over all problems a wave visits (persistent grid), are folded over
//     the 32 token lanes once at the end and summed over waves by a second kernel in a fixed order: deterministic, no atomics.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int WPB = 4;                       // waves per workgroup (each independent)

struct Frag { bf16x8 v[2]; };
struct Row { u32x2 w[4]; };                  // 16 bf16 of layout R: w[j] = dims 8 j + 4 half + 0..3

__device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0], b.v[0], acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1], b.v[1], acc, 0, 0, 0);
}
__device__ __forceinline__ Frag frag_of(const Row& r) {
  Frag f;
  f.v[0] = __builtin_bit_cast(bf16x8, u32x4{r.w[0][0], r.w[0][1], r.w[1][0], r.w[1][1]});
  f.v[1] = __builtin_bit_cast(bf16x8, u32x4{r.w[2][0], r.w[2][1], r.w[3][0], r.w[3][1]});
  return f;
}
__device__ __forceinline__ Row pack_row(const float (&x)[16]) {
  Row r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r.w[j][0] = pack2bf(x[4 * j], x[4 * j + 1]); r.w[j][1] = pack2bf(x[4 * j + 2], x[4 * j + 3]); }
  return r;
}
__device__ __forceinline__ void unpack_row(const Row& r, float (&x)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[4 * j] = __uint_as_float(r.w[j][0] << 16); x[4 * j + 1] = __uint_as_float(r.w[j][0] & 0xffff0000u);
    x[4 * j + 2] = __uint_as_float(r.w[j][1] << 16); x[4 * j + 3] = __uint_as_float(r.w[j][1] & 0xffff0000u);
  }
}
// the lane's 16 dims of one token row's 64-byte head slice (zero for rows past the sequence)
__device__ __forceinline__ Row load_row(const bf16_t* slice, int half, bool valid) {
  Row r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.w[j] = *reinterpret_cast<const u32x2*>(slice + 8 * j + 4 * half);
  if (!valid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r.w[j] = u32x2{0u, 0u};
  }
  return r;
}
__device__ __forceinline__ void store_row(bf16_t* slice, int half, const Row& r) {
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x2*>(slice + 8 * j + 4 * half) = r.w[j];
}
// row-major LDS tile [32 tokens][64 B], private to the wave
__device__ __forceinline__ void tile_write(char* tile, int row, int half, const Row& r) {
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x2*>(tile + row * 64 + 16 * j + 8 * half) = r.w[j];
}
// Transposed fragment of a tile: MFMA row m = lane & 31 is head dim m, contraction slot (t, i) is token 8 (2t + i/4) + 4 half + i%4.
// ds_read_b64_tr_b16 (measured, tools/tr_probe.hip): in each 16-lane group, output lane i element j = element i & 3 of the 8 bytes
// addressed by lane 4 j + (i >> 2).  Lane t16 of a group therefore points at token kbase + (t16 >> 2), dims 16 grp + 4 (t16 & 3) .. + 3,
// and receives tokens kbase .. kbase + 3 at dim 16 grp + t16.  4 token rows x 64 B per 32 lanes: every bank once.
__device__ __forceinline__ Frag tile_cols(const char* tile, int lane) {
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)tile;
  const int t16 = lane & 15, grp = (lane >> 4) & 1, half = lane >> 5;
  const uint32_t a0 = base + (uint32_t)((4 * half + (t16 >> 2)) * 64 + (16 * grp + 4 * (t16 & 3)) * 2);
  u32x2 r00, r01, r10, r11;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r00) : "v"(a0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(r01) : "v"(a0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(r10) : "v"(a0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1536" : "=v"(r11) : "v"(a0) : "memory");
  // the reads are asynchronous and the compiler does not know it: the wait carries the registers
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r00), "+v"(r01), "+v"(r10), "+v"(r11) :: "memory");
  Frag f;
  f.v[0] = __builtin_bit_cast(bf16x8, u32x4{r00[0], r00[1], r01[0], r01[1]});
  f.v[1] = __builtin_bit_cast(bf16x8, u32x4{r10[0], r10[1], r11[0], r11[1]});
  return f;
}
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float pair_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }

struct ShortParams {
  const bf16_t* q; const bf16_t* kv; const float* q_scale; const float* k_scale;
  int64_t ldq, ldkv;
  int nseq, H, L;
  float scale;
  bf16_t* out; int64_t ldo;                         // forward
  const bf16_t* dout; int64_t lddo;                 // backward
  bf16_t* dq; bf16_t* dkv; int64_t lddq, lddkv;
  float* part;                                      // [nwaves][2][32] scale-gradient partials
};

// l2norm of the lane's half row + the learned scale: unit row xn, inverse norm, and xn * s * mult packed for the MFMA
__device__ __forceinline__ Row norm_row(const Row& raw, const float (&s)[16], float mult, float (&xn)[16], float& inv) {
  float x[16];
  unpack_row(raw, x);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) ss = fmaf(x[i], x[i], ss);
  ss = pair_sum(ss);
  inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);             // F.normalize eps
  float y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { xn[i] = x[i] * inv; y[i] = xn[i] * s[i] * mult; }
  return pack_row(y);
}
// gradient through y = xn * s (xn = x / |x|): g = dL/dy -> dL/dx ; dscale[i] += g[i] * xn[i]
__device__ __forceinline__ Row norm_row_bwd(const float (&g)[16], const float (&xn)[16], float inv, const float (&s)[16], float (&dscale)[16]) {
  float gn[16], dot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dscale[i] = fmaf(g[i], xn[i], dscale[i]); gn[i] = g[i] * s[i]; dot = fmaf(gn[i], xn[i], dot); }
  dot = pair_sum(dot);
  float dx[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) dx[i] = inv * (gn[i] - xn[i] * dot);
  return pack_row(dx);
}

__device__ __forceinline__ void load_scales(const float* v, int half, float (&s)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(v + 8 * j + 4 * half);
    s[4 * j] = t[0]; s[4 * j + 1] = t[1]; s[4 * j + 2] = t[2]; s[4 * j + 3] = t[3];
  }
}

__global__ __launch_bounds__(WPB * 64) void attn_short_fwd_kernel(ShortParams p) {
  __shared__ __attribute__((aligned(16))) char lds[WPB][2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, half = lane >> 5;
  char* tv = lds[wave];
  float sq[16], sk[16];
  load_scales(p.q_scale, half, sq); load_scales(p.k_scale, half, sk);
  const int HD = p.H * 32;
  const bool valid = row < p.L;
  const int crow = valid ? row : p.L - 1;
  const int64_t nitems = (int64_t)p.nseq * p.H;
  for (int64_t it = (int64_t)blockIdx.x * WPB + wave; it < nitems; it += (int64_t)gridDim.x * WPB) {
    const int64_t s = it / p.H; const int h = (int)(it % p.H);
    const int64_t tok = s * p.L + crow;
    const Row rq = load_row(p.q + tok * p.ldq + h * 32, half, valid);
    const Row rk = load_row(p.kv + tok * p.ldkv + h * 32, half, valid);
    const Row rv = load_row(p.kv + tok * p.ldkv + HD + h * 32, half, valid);
    float xn[16], inv;
    const Frag Qt = frag_of(norm_row(rq, sq, p.scale * LOG2E, xn, inv));
    const Frag Ks = frag_of(norm_row(rk, sk, 1.f, xn, inv));
    tile_write(tv, row, half, rv);
    // S^T: rows = keys 8 j + 4 half + r, column = the lane's query (log2 domain)
    const f32x16 sacc = mma(Ks, Qt);
    float pr[16], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = 8 * (i >> 2) + 4 * half + (i & 3);
      pr[i] = key < p.L ? sacc[i] : -INFINITY;
      m = fmaxf(m, pr[i]);
    }
    m = pair_max(m);
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { pr[i] = __builtin_amdgcn_exp2f(pr[i] - m); l += pr[i]; }
    l = pair_sum(l);
    const Frag P = frag_of(pack_row(pr));
    const Frag Vt = tile_cols(tv, lane);
    const f32x16 oacc = mma(Vt, P);                  // O^T: rows = dims of layout R, column = the lane's query
    if (valid) {
      const float il = 1.f / l;
      float o[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = oacc[i] * il;
      store_row(p.out + tok * p.ldo + h * 32, half, pack_row(o));
    }
  }
}

// (three waves per SIMD: the learned scales live in LDS and are re-read at each use instead of holding 32 registers)
__global__ __launch_bounds__(WPB * 64, 3) void attn_short_bwd_kernel(ShortParams p) {
  __shared__ __attribute__((aligned(16))) char lds[WPB][3 * 2048 + 3 * 128];
  __shared__ __attribute__((aligned(16))) float scales[2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, half = lane >> 5;
  char* tk = lds[wave]; char* tq = tk + 2048; char* tdo = tk + 4096;
  float* st = reinterpret_cast<float*>(tk + 6144);   // [3][32]: row max (log2), 1 / row sum, delta of each query
  if (threadIdx.x < 64) scales[threadIdx.x >> 5][threadIdx.x & 31] = (threadIdx.x < 32 ? p.q_scale : p.k_scale)[threadIdx.x & 31];
  __syncthreads();
  float dsq[16], dsk[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dsq[i] = 0.f; dsk[i] = 0.f; }
  const int HD = p.H * 32;
  const bool valid = row < p.L;
  const int crow = valid ? row : p.L - 1;
  const int64_t nitems = (int64_t)p.nseq * p.H;
  for (int64_t it = (int64_t)blockIdx.x * WPB + wave; it < nitems; it += (int64_t)gridDim.x * WPB) {
    const int64_t s = it / p.H; const int h = (int)(it % p.H);
    const int64_t tok = s * p.L + crow;
    const Row rq = load_row(p.q + tok * p.ldq + h * 32, half, valid);
    const Row rk = load_row(p.kv + tok * p.ldkv + h * 32, half, valid);
    const Row rv = load_row(p.kv + tok * p.ldkv + HD + h * 32, half, valid);
    const Row rdo = load_row(p.dout + tok * p.lddo + h * 32, half, valid);
    float qn[16], kn[16], invq, invk;
    Row rqt, rks;
    { float sc[16]; load_scales(scales[0], half, sc); rqt = norm_row(rq, sc, p.scale * LOG2E, qn, invq); }
    { float sc[16]; load_scales(scales[1], half, sc); rks = norm_row(rk, sc, 1.f, kn, invk); }
    const Frag Qt = frag_of(rqt), Ks = frag_of(rks), Vf = frag_of(rv), dOf = frag_of(rdo);
    tile_write(tk, row, half, rks); tile_write(tq, row, half, rqt); tile_write(tdo, row, half, rdo);

    // ---- queries as columns: P, dP, delta, dS -> dQ
    {
      const f32x16 sacc = mma(Ks, Qt);
      float pr[16], m = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 8 * (i >> 2) + 4 * half + (i & 3);
        pr[i] = key < p.L ? sacc[i] : -INFINITY;
        m = fmaxf(m, pr[i]);
      }
      m = pair_max(m);
      float l = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { pr[i] = __builtin_amdgcn_exp2f(pr[i] - m); l += pr[i]; }
      l = pair_sum(l);
      const float il = 1.f / l;
      const f32x16 dp = mma(Vf, dOf);               // dP^T[key][query] = sum_d V[key][d] dO[query][d]
      float delta = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { pr[i] *= il; delta = fmaf(pr[i], dp[i], delta); }
      delta = pair_sum(delta);
      float dz[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) dz[i] = pr[i] * (dp[i] - delta);
      if (half == 0) { st[row] = m; st[32 + row] = il; st[64 + row] = delta; }
      const Frag KsT = tile_cols(tk, lane);
      const f32x16 gacc = mma(KsT, frag_of(pack_row(dz)));     // (dS K^s)^T: rows = dims, column = the lane's query
      float g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = gacc[i] * p.scale;    // dL / d(q^ * q_scale)
      float sc[16];
      load_scales(scales[0], half, sc);
      const Row rdq = norm_row_bwd(g, qn, invq, sc, dsq);
      if (valid) store_row(p.dq + tok * p.lddq + h * 32, half, rdq);
    }
    // ---- keys as columns: P, dP, dS -> dK, dV
    {
      const f32x16 sacc = mma(Qt, Ks);              // S[query][key]: rows = queries 8 j + 4 half + r, column = the lane's key
      const f32x16 dp = mma(dOf, Vf);
      float p2[16], dz[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 mq = *reinterpret_cast<const f32x4*>(st + 8 * j + 4 * half);
        const f32x4 ilq = *reinterpret_cast<const f32x4*>(st + 32 + 8 * j + 4 * half);
        const f32x4 dq4 = *reinterpret_cast<const f32x4*>(st + 64 + 8 * j + 4 * half);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * j + r, query = 8 * j + 4 * half + r;
          const float pv = (valid && query < p.L) ? __builtin_amdgcn_exp2f(sacc[i] - mq[r]) * ilq[r] : 0.f;
          p2[i] = pv;
          dz[i] = pv * (dp[i] - dq4[r]);
        }
      }
      const Frag QtT = tile_cols(tq, lane), dOT = tile_cols(tdo, lane);
      const f32x16 gk = mma(QtT, frag_of(pack_row(dz)));        // (dS^T Q~)^T: Q~ carries scale * log2 e
      const f32x16 gv = mma(dOT, frag_of(pack_row(p2)));        // (P^T dO)^T
      float g[16], dv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { g[i] = gk[i] * (1.f / LOG2E); dv[i] = gv[i]; }
      float sc[16];
      load_scales(scales[1], half, sc);
      const Row rdk = norm_row_bwd(g, kn, invk, sc, dsk);
      if (valid) {
        store_row(p.dkv + tok * p.lddkv + h * 32, half, rdk);
        store_row(p.dkv + tok * p.lddkv + HD + h * 32, half, pack_row(dv));
      }
    }
  }
  // fold the scale gradients over the 32 token lanes of each half, one partial row per wave
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { dsq[i] += __shfl_xor(dsq[i], o, 64); dsk[i] += __shfl_xor(dsk[i], o, 64); }
  }
  if (row == 0) {
    float* dst = p.part + ((int64_t)blockIdx.x * WPB + wave) * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = 8 * (i >> 2) + 4 * half + (i & 3);
      dst[d] = dsq[i]; dst[32 + d] = dsk[i];
    }
  }
}

// dq_scale[d] += sum over waves (in order) of part[w][0][d]; dk_scale likewise
__global__ __launch_bounds__(64) void attn_short_scale_sum_kernel(const float* __restrict__ part, int nwaves, float* __restrict__ dqs, float* __restrict__ dks) {
  const int t = threadIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int w = 0;
  for (; w + 4 <= nwaves; w += 4) {
    a0 += part[(int64_t)w * 64 + t]; a1 += part[(int64_t)(w + 1) * 64 + t]; a2 += part[(int64_t)(w + 2) * 64 + t]; a3 += part[(int64_t)(w + 3) * 64 + t];
  }
  for (; w < nwaves; ++w) a0 += part[(int64_t)w * 64 + t];
  const float v = (a0 + a1) + (a2 + a3);
  if (t < 32) { if (dqs) dqs[t] += v; } else if (dks) dks[t - 32] += v;
}

int short_grid(int nseq, int H) {
  static int cus = [] { hipDeviceProp_t pr; int dev = 0; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256; }();
  const int64_t items = (int64_t)nseq * H, want = (items + WPB - 1) / WPB;
  const int64_t cap = (int64_t)cus * 4;            // four workgroups (16 waves) per CU
  return (int)(want < cap ? want : cap);
}

bool short_args_ok(const void* q, const void* kv, int64_t ldq, int64_t ldkv, int nseq, int H, int L) {
  return q && kv && nseq >= 1 && H >= 1 && L >= 1 && L <= 32 && ldq % 4 == 0 && ldkv % 4 == 0 && ldq >= H * 32 && ldkv >= 2 * H * 32 &&
         reinterpret_cast<uintptr_t>(q) % 8 == 0 && reinterpret_cast<uintptr_t>(kv) % 8 == 0;
}

}  // namespace

// 1 when ctclip_attn_short_* serves this shape (bf16, d_head 32, L <= 32, no bias / mask)
extern "C" int ctclip_attn_short_supported(int L, int D, int dtype) { return dtype == DT_BF16 && D == 32 && L >= 1 && L <= 32; }

// out[(s L + i), h*32 + :] = softmax_j(scale * <l2norm(q_i) q_scale, l2norm(k_j) k_scale>) v_j   (attention.py:145-178)
// q (nseq*L, ldq >= H*32), kv (nseq*L, ldkv >= 2*H*32) = [k | v], out (nseq*L, ldo) bf16; q_scale, k_scale (32) f32.
extern "C" int ctclip_attn_short_fwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const float* q_scale, const float* k_scale,
                                     void* out, int64_t ldo, int nseq, int H, int L, float scale, hipStream_t stream) {
  if (!short_args_ok(q, kv, ldq, ldkv, nseq, H, L) || !out || !q_scale || !k_scale || ldo % 4 || ldo < H * 32) { ctclip_set_error("attn_short_fwd: bad args (L <= 32, 8-byte aligned rows)"); return CTCLIP_EBADARG; }
  ShortParams p{};
  p.q = (const bf16_t*)q; p.kv = (const bf16_t*)kv; p.q_scale = q_scale; p.k_scale = k_scale; p.ldq = ldq; p.ldkv = ldkv;
  p.nseq = nseq; p.H = H; p.L = L; p.scale = scale; p.out = (bf16_t*)out; p.ldo = ldo;
  hipLaunchKernelGGL(attn_short_fwd_kernel, dim3((unsigned)short_grid(nseq, H)), dim3(WPB * 64), 0, stream, p);
  return ctclip_check_launch("attn_short_fwd");
}

extern "C" int64_t ctclip_attn_short_bwd_workspace(int nseq, int H) { return (int64_t)short_grid(nseq, H) * WPB * 64 * 4; }

// dq (nseq*L, lddq), dkv (nseq*L, lddkv) = [dk | dv] bf16 are overwritten; dq_scale, dk_scale (32) f32 are ACCUMULATED (+=) when non-null.
extern "C" int ctclip_attn_short_bwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const float* q_scale, const float* k_scale,
                                     const void* dout, int64_t lddo, void* dq, int64_t lddq, void* dkv, int64_t lddkv, float* dq_scale,
                                     float* dk_scale, int nseq, int H, int L, float scale, void* workspace, int64_t workspace_bytes,
                                     hipStream_t stream) {
  if (!short_args_ok(q, kv, ldq, ldkv, nseq, H, L) || !dout || !dq || !dkv || !q_scale || !k_scale || lddo % 4 || lddq % 4 || lddkv % 4 ||
      lddo < H * 32 || lddq < H * 32 || lddkv < 2 * H * 32) { ctclip_set_error("attn_short_bwd: bad args"); return CTCLIP_EBADARG; }
  if (!workspace || workspace_bytes < ctclip_attn_short_bwd_workspace(nseq, H)) { ctclip_set_error("attn_short_bwd: workspace too small"); return CTCLIP_EWORKSPACE; }
  ShortParams p{};
  p.q = (const bf16_t*)q; p.kv = (const bf16_t*)kv; p.q_scale = q_scale; p.k_scale = k_scale; p.ldq = ldq; p.ldkv = ldkv;
  p.nseq = nseq; p.H = H; p.L = L; p.scale = scale; p.dout = (const bf16_t*)dout; p.lddo = lddo;
  p.dq = (bf16_t*)dq; p.dkv = (bf16_t*)dkv; p.lddq = lddq; p.lddkv = lddkv; p.part = (float*)workspace;
  const int grid = short_grid(nseq, H);
  hipLaunchKernelGGL(attn_short_bwd_kernel, dim3((unsigned)grid), dim3(WPB * 64), 0, stream, p);
  if (dq_scale || dk_scale) hipLaunchKernelGGL(attn_short_scale_sum_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, grid * WPB, dq_scale, dk_scale);
  return ctclip_check_launch("attn_short_bwd");
}
