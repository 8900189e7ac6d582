// Fused (flash-style) attention for gfx950, forward and backward, for the two attention shapes on the
// CT-CLIP hot path:
//   * CTViT cosine attention (attention.py:145-181): d_head 32, L = 576 (spatial, + continuous position
//     bias (H,L,L)) or L = 24 (temporal), sim = 8 * q^ k^T, q^/k^ = l2norm * learned scale (prep kernels below)
//   * BERT self-attention (HF modeling_bert): d_head 64, additive key-padding mask, scale 1/8.
//
// One wave64 owns 32 query rows (or 32 key rows in the dK/dV kernel) of one (sequence, head) and walks the
// other axis in 32-wide tiles with mfma 32x32 (bf16: 32x32x16, f32 parity mode: 32x32x2).  No LDS, no
// barriers, no atomics except the optional dBias accumulation:
//   - S^T = K Q^T is computed with K as the A operand, so each lane owns ONE query column and 16 keys in
//     registers: softmax row max/sum are in-lane + one cross-half shuffle.
//   - The A-operand lane a loads row pi(a) (bits 2,3 swapped).  With that permutation register r of a lane
//     in half h holds row 16*(r>>3) + 8*h + (r&7): two runs of 8 contiguous indices.  P / dS therefore feed
//     the second product directly from registers (as the B operand), and the matching A operand (V^T, K^T,
//     dO^T, Q^T rows) is two 16-byte loads per lane from per-(sequence, head) transposed copies produced by
//     ctclip_head_transpose.  (The hardware pairs element e of lane (i,h) of A with element e of lane (j,h)
//     of B, so any assignment of contraction indices to slots is valid if A and B agree.)
//   - O^T = V^T P^T keeps the query as the lane index, so the online-softmax rescale is a per-lane scalar.
#include "common.h"

namespace {

__device__ __forceinline__ int pi32(int c) { return (c & 3) | ((c & 4) << 1) | ((c & 8) >> 1) | (c & 16); }
__device__ __forceinline__ int slot_index(int r, int half) { return 16 * (r >> 3) + 8 * half + (r & 7); }

template <typename T, int KD> struct Frag;
template <int KD> struct Frag<bf16_t, KD> { bf16x8 v[KD / 16]; };
template <int KD> struct Frag<float, KD> { float v[KD / 2]; };

template <int KD>
__device__ __forceinline__ void frag_zero(Frag<bf16_t, KD>& f) {
#pragma unroll
  for (int g = 0; g < KD / 16; ++g) f.v[g] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
}
template <int KD>
__device__ __forceinline__ void frag_zero(Frag<float, KD>& f) {
#pragma unroll
  for (int e = 0; e < KD / 2; ++e) f.v[e] = 0.f;
}
// p points at contraction index 0 of this lane's row; groups of 8 at 16g + 8*half.  Loads are UNCONDITIONAL (a branch
// around a load makes hipcc serialise the round trips): groups at or past `limit` (a multiple of 8, >= 8) are clamped to the
// last valid group.  Every caller multiplies such slots by an exact zero (masked probability / dS) or never stores them, and
// the clamped data is real, finite tensor data.
template <int KD>
__device__ __forceinline__ void frag_load(Frag<bf16_t, KD>& f, const bf16_t* p, int half, int limit) {
#pragma unroll
  for (int g = 0; g < KD / 16; ++g) {
    int off = 16 * g + 8 * half;
    off = off < limit ? off : limit - 8;
    f.v[g] = *reinterpret_cast<const bf16x8*>(p + off);
  }
}
template <int KD>
__device__ __forceinline__ void frag_load(Frag<float, KD>& f, const float* p, int half, int limit) {
#pragma unroll
  for (int g = 0; g < KD / 16; ++g) {
    int off = 16 * g + 8 * half;
    off = off < limit ? off : limit - 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + off), b = *reinterpret_cast<const f32x4*>(p + off + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.v[8 * g + e] = a[e]; f.v[8 * g + 4 + e] = b[e]; }
  }
}
template <int KD>
__device__ __forceinline__ f32x16 mma(f32x16 acc, const Frag<bf16_t, KD>& a, const Frag<bf16_t, KD>& b) {
#pragma unroll
  for (int g = 0; g < KD / 16; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[g], b.v[g], acc, 0, 0, 0);
  return acc;
}
template <int KD>
__device__ __forceinline__ f32x16 mma(f32x16 acc, const Frag<float, KD>& a, const Frag<float, KD>& b) {
#pragma unroll
  for (int e = 0; e < KD / 2; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[e], b.v[e], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void frag_from_regs(Frag<bf16_t, 32>& f, const float (&p)[16]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack2bf(p[8 * g + 2 * e], p[8 * g + 2 * e + 1]);
    f.v[g] = __builtin_bit_cast(bf16x8, w);
  }
}
__device__ __forceinline__ void frag_from_regs(Frag<float, 32>& f, const float (&p)[16]) {
#pragma unroll
  for (int e = 0; e < 16; ++e) f.v[e] = p[e];
}

struct AttnParams {
  const void *q, *k, *v, *qt, *kt, *vt, *o, *dout, *dot;
  const float *bias, *keymask, *lse, *delta;
  void *out, *dq, *dk, *dv;
  float *lse_out, *dbias;
  int nseq, H, L, Lp;
  int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  float scale;
};

// scores of one 32x32 tile in "lane = column c, regs = rows slot_index(r, half)" layout -> logits
template <bool ROWS_ARE_KEYS>
__device__ __forceinline__ void tile_logits(float (&val)[16], const f32x16& s, const AttnParams& p, int seq, int h, int col_idx,
                                            int row_base, int half) {
  // ROWS_ARE_KEYS: column = query (col_idx), rows = keys.  else: column = key, rows = queries.
  const int L = p.L;
  float add[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) add[r] = 0.f;
  if (p.bias) {   // wave-uniform branch; the loads inside are unconditional (indices clamped)
    if (ROWS_ARE_KEYS && (L & 7) == 0) {
      const int qc = col_idx < L ? col_idx : L - 1;
      const float* brow = p.bias + ((int64_t)h * L + qc) * L;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int k0 = row_base + 16 * g + 8 * half;
        k0 = k0 < L ? k0 : L - 8;               // runs are 8-aligned: wholly valid or wholly past L (masked below)
        const f32x4 a = *reinterpret_cast<const f32x4*>(brow + k0), b = *reinterpret_cast<const f32x4*>(brow + k0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { add[8 * g + e] = a[e]; add[8 * g + 4 + e] = b[e]; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = row_base + slot_index(r, half);
        int qi = ROWS_ARE_KEYS ? col_idx : ri, kj = ROWS_ARE_KEYS ? ri : col_idx;
        qi = qi < L ? qi : L - 1; kj = kj < L ? kj : L - 1;
        add[r] = p.bias[((int64_t)h * L + qi) * L + kj];
      }
    }
  }
  if (p.keymask) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int kj = ROWS_ARE_KEYS ? row_base + slot_index(r, half) : col_idx;
      kj = kj < L ? kj : L - 1;
      add[r] += p.keymask[(int64_t)seq * L + kj];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ri = row_base + slot_index(r, half);
    const int qi = ROWS_ARE_KEYS ? col_idx : ri, kj = ROWS_ARE_KEYS ? ri : col_idx;
    val[r] = (qi < L && kj < L) ? s[r] * p.scale + add[r] : -INFINITY;
  }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qb = blockIdx.x * 4 + wave, h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  if (qb * 32 >= L) return;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* Vt = reinterpret_cast<const T*>(p.vt);

  const int qic = qi < L ? qi : L - 1;   // clamped: loads stay unconditional, invalid lanes are never stored
  Frag<T, D> qf;
  frag_load(qf, Q + ((int64_t)seq * L + qic) * p.ldq + h * D, half, D);

  float m = -INFINITY, lsum = 0.f;
  f32x16 oacc[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;

  const int nkb = (L + 31) / 32;
  // register double-buffering: the next tile's K and V^T fragments are in flight while the current tile is computed
  auto load_k = [&](int kb, Frag<T, D>& kf) {
    int krow = kb * 32 + ar;
    krow = krow < L ? krow : L - 1;       // keys past L are masked to -inf in tile_logits
    frag_load(kf, K + ((int64_t)seq * L + krow) * p.ldk + h * D, half, D);
  };
  auto load_v = [&](int kb, Frag<T, 32> (&vf)[D / 32]) {
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
      frag_load(vf[i], Vt + (((int64_t)seq * p.H + h) * D + i * 32 + ar) * Lp + kb * 32, half, Lp - kb * 32);
  };
  Frag<T, D> kf;
  Frag<T, 32> vf[D / 32];
  load_k(0, kf);
  load_v(0, vf);
  for (int kb = 0; kb < nkb; ++kb) {
    Frag<T, D> kn;
    Frag<T, 32> vn[D / 32];
    const int kbn = kb + 1 < nkb ? kb + 1 : kb;
    load_k(kbn, kn);
    load_v(kbn, vn);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    s = mma(s, kf, qf);
    float val[16];
    tile_logits<true>(val, s, p, seq, h, qi, kb * 32, half);
    float mx = val[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, val[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float mnew = fmaxf(m, mx);
    if (mnew == -INFINITY) mnew = 0.f;
    const float alpha = __expf(m - mnew);
    float pr[16], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { pr[r] = __expf(val[r] - mnew); ps += pr[r]; }
    lsum = lsum * alpha + ps;
    m = mnew;
    Frag<T, 32> pf;
    frag_from_regs(pf, pr);
#pragma unroll
    for (int i = 0; i < D / 32; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      oacc[i] = mma(oacc[i], vf[i], pf);
    }
    kf = kn;
#pragma unroll
    for (int i = 0; i < D / 32; ++i) vf[i] = vn[i];
  }
  const float l = lsum + __shfl_xor(lsum, 32, 64);
  if (qi < L) {
    const float inv = 1.f / l;
    T* O = reinterpret_cast<T*>(p.out) + ((int64_t)seq * L + qi) * p.ldo + h * D;
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = oacc[i][8 * g + e] * inv;
        store8(O + i * 32 + 16 * g + 8 * half, o8);
      }
    if (half == 0 && p.lse_out) p.lse_out[((int64_t)seq * p.H + h) * L + qi] = m + __logf(l);
  }
}

// dQ (and dBias): one wave per 32-query block, loop over key tiles.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qb = blockIdx.x * 4 + wave, h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  if (qb * 32 >= L) return;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  const T* Kt = reinterpret_cast<const T*>(p.kt);
  const T* dO = reinterpret_cast<const T*>(p.dout);

  const int qic = qi < L ? qi : L - 1;
  Frag<T, D> qf, dof;
  frag_load(qf, Q + ((int64_t)seq * L + qic) * p.ldq + h * D, half, D);
  frag_load(dof, dO + ((int64_t)seq * L + qic) * p.lddo + h * D, half, D);
  const float lse = p.lse[((int64_t)seq * p.H + h) * L + qic];
  const float delta = p.delta[((int64_t)seq * p.H + h) * L + qic];

  f32x16 dqacc[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  const int nkb = (L + 31) / 32;
  auto load = [&](int kb, Frag<T, D>& kf, Frag<T, D>& vf, Frag<T, 32> (&ktf)[D / 32]) {
    int krow = kb * 32 + ar;
    krow = krow < L ? krow : L - 1;
    frag_load(kf, K + ((int64_t)seq * L + krow) * p.ldk + h * D, half, D);
    frag_load(vf, V + ((int64_t)seq * L + krow) * p.ldv + h * D, half, D);
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
      frag_load(ktf[i], Kt + (((int64_t)seq * p.H + h) * D + i * 32 + ar) * Lp + kb * 32, half, Lp - kb * 32);
  };
  Frag<T, D> kf, vf;
  Frag<T, 32> ktf[D / 32];
  load(0, kf, vf, ktf);
  for (int kb = 0; kb < nkb; ++kb) {
    Frag<T, D> kn, vn;
    Frag<T, 32> ktn[D / 32];
    load(kb + 1 < nkb ? kb + 1 : kb, kn, vn, ktn);   // next key tile in flight under this tile's math
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, kf, qf);
    dp = mma(dp, vf, dof);
    float val[16], ds[16];
    tile_logits<true>(val, s, p, seq, h, qi, kb * 32, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = kb * 32 + slot_index(r, half);
      const float pr = (qi < L && kj < L) ? __expf(val[r] - lse) : 0.f;
      ds[r] = pr * (dp[r] - delta);
      if (p.dbias && qi < L && kj < L) atomicAdd(p.dbias + ((int64_t)h * L + qi) * L + kj, ds[r]);
    }
    Frag<T, 32> dsf;
    frag_from_regs(dsf, ds);
#pragma unroll
    for (int i = 0; i < D / 32; ++i) dqacc[i] = mma(dqacc[i], ktf[i], dsf);
    kf = kn; vf = vn;
#pragma unroll
    for (int i = 0; i < D / 32; ++i) ktf[i] = ktn[i];
  }
  if (qi < L) {
    T* dQ = reinterpret_cast<T*>(p.dq) + ((int64_t)seq * L + qi) * p.lddq + h * D;
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = dqacc[i][8 * g + e] * p.scale;
        store8(dQ + i * 32 + 16 * g + 8 * half, o8);
      }
  }
}

// dBias without atomics: one wave owns ONE (query tile, key tile) pair of one head and walks a strided subset of the
// sequences, recomputing S and dP and accumulating dS in 16 registers; the bias tile is loop invariant (registers).
// Per-split partial slabs are written with plain stores and summed by dbias_reduce_kernel (deterministic).  The first
// version of this path used f32 atomics from the dQ kernel: 510 M atomics per layer cost 11 ms.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_dbias_kernel(AttnParams p, float* __restrict__ dbias_part, int nsplit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.L;
  const int nkb = (L + 31) / 32;
  const int pair = blockIdx.x * 4 + wave, h = blockIdx.y, split = blockIdx.z;
  if (pair >= nkb * nkb) return;
  const int qb = pair / nkb, kb = pair % nkb;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  const int qic = qi < L ? qi : L - 1;
  int krow = kb * 32 + ar;
  krow = krow < L ? krow : L - 1;
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  const T* dO = reinterpret_cast<const T*>(p.dout);

  // loop-invariant additive logits (bias tile) and validity
  float add[16];
  bool ok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kj = kb * 32 + slot_index(r, half);
    ok[r] = qi < L && kj < L;
    const int kc = kj < L ? kj : L - 1;
    add[r] = p.bias ? p.bias[((int64_t)h * L + qic) * L + kc] : 0.f;
  }
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  auto load = [&](int seq, Frag<T, D>& qf, Frag<T, D>& dof, Frag<T, D>& kf, Frag<T, D>& vf, float& lse, float& delta) {
    const int64_t srow = (int64_t)seq * L;
    frag_load(qf, Q + (srow + qic) * p.ldq + h * D, half, D);
    frag_load(dof, dO + (srow + qic) * p.lddo + h * D, half, D);
    frag_load(kf, K + (srow + krow) * p.ldk + h * D, half, D);
    frag_load(vf, V + (srow + krow) * p.ldv + h * D, half, D);
    lse = p.lse[((int64_t)seq * p.H + h) * L + qic];
    delta = p.delta[((int64_t)seq * p.H + h) * L + qic];
  };
  Frag<T, D> qf, dof, kf, vf;
  float lse, delta;
  int seq = split;
  if (seq < p.nseq) load(seq, qf, dof, kf, vf, lse, delta);
  for (; seq < p.nseq; seq += nsplit) {
    Frag<T, D> qn, don, kn, vn;
    float lsen, deltan;
    const int sn = seq + nsplit < p.nseq ? seq + nsplit : seq;
    load(sn, qn, don, kn, vn, lsen, deltan);     // next sequence in flight
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, kf, qf);
    dp = mma(dp, vf, dof);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t = s[r] * p.scale + add[r];
      if (p.keymask) { const int kj = kb * 32 + slot_index(r, half); t += p.keymask[(int64_t)seq * L + (kj < L ? kj : L - 1)]; }
      const float pr = ok[r] ? __expf(t - lse) : 0.f;
      acc[r] += pr * (dp[r] - delta);
    }
    qf = qn; dof = don; kf = kn; vf = vn; lse = lsen; delta = deltan;
  }
  if (qi < L) {
    float* dst = dbias_part + (((int64_t)split * p.H + h) * L + qi) * L;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = kb * 32 + slot_index(r, half);
      if (kj < L) dst[kj] = acc[r];
    }
  }
}

__global__ void dbias_reduce_kernel(const float* __restrict__ part, float* __restrict__ dbias, int nsplit, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += part[(int64_t)s * n + i];
    dbias[i] += t;
  }
}

// dK, dV: one wave per 32-key block, loop over query tiles.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jb = blockIdx.x * 4 + wave, h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  if (jb * 32 >= L) return;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int kj = jb * 32 + c;
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  const T* Qt = reinterpret_cast<const T*>(p.qt);
  const T* dO = reinterpret_cast<const T*>(p.dout);
  const T* dOt = reinterpret_cast<const T*>(p.dot);

  const int kjc = kj < L ? kj : L - 1;
  Frag<T, D> kf, vf;
  frag_load(kf, K + ((int64_t)seq * L + kjc) * p.ldk + h * D, half, D);
  frag_load(vf, V + ((int64_t)seq * L + kjc) * p.ldv + h * D, half, D);

  f32x16 dkacc[D / 32], dvacc[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkacc[i][r] = 0.f; dvacc[i][r] = 0.f; }

  const int64_t statbase = ((int64_t)seq * p.H + h) * L;
  const int nqb = (L + 31) / 32;
  auto load_q = [&](int qb, Frag<T, D>& qf, Frag<T, D>& dof, Frag<T, 32> (&dotf)[D / 32], Frag<T, 32> (&qtf)[D / 32]) {
    int qrow = qb * 32 + ar;
    qrow = qrow < L ? qrow : L - 1;
    frag_load(qf, Q + ((int64_t)seq * L + qrow) * p.ldq + h * D, half, D);
    frag_load(dof, dO + ((int64_t)seq * L + qrow) * p.lddo + h * D, half, D);
#pragma unroll
    for (int i = 0; i < D / 32; ++i) {
      const int64_t trow = (((int64_t)seq * p.H + h) * D + i * 32 + ar) * Lp + qb * 32;
      frag_load(dotf[i], dOt + trow, half, Lp - qb * 32);
      frag_load(qtf[i], Qt + trow, half, Lp - qb * 32);
    }
  };
  Frag<T, D> qf, dof;
  Frag<T, 32> dotf[D / 32], qtf[D / 32];
  load_q(0, qf, dof, dotf, qtf);
  for (int qb = 0; qb < nqb; ++qb) {
    Frag<T, D> qn, don;
    Frag<T, 32> dotn[D / 32], qtn[D / 32];
    load_q(qb + 1 < nqb ? qb + 1 : qb, qn, don, dotn, qtn);   // next tile in flight under this tile's math
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, qf, kf);     // D[query rho][key c]
    dp = mma(dp, dof, vf);
    float val[16], pr[16], ds[16];
    tile_logits<false>(val, s, p, seq, h, kj, qb * 32, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = qb * 32 + slot_index(r, half);
      const int qc = qi < L ? qi : L - 1;
      const float lse = p.lse[statbase + qc], delta = p.delta[statbase + qc];
      pr[r] = (qi < L && kj < L) ? __expf(val[r] - lse) : 0.f;
      ds[r] = pr[r] * (dp[r] - delta);
    }
    Frag<T, 32> pf, dsf;
    frag_from_regs(pf, pr);
    frag_from_regs(dsf, ds);
#pragma unroll
    for (int i = 0; i < D / 32; ++i) {
      dvacc[i] = mma(dvacc[i], dotf[i], pf);
      dkacc[i] = mma(dkacc[i], qtf[i], dsf);
    }
    qf = qn; dof = don;
#pragma unroll
    for (int i = 0; i < D / 32; ++i) { dotf[i] = dotn[i]; qtf[i] = qtn[i]; }
  }
  if (kj < L) {
    T* dK = reinterpret_cast<T*>(p.dk) + ((int64_t)seq * L + kj) * p.lddk + h * D;
    T* dV = reinterpret_cast<T*>(p.dv) + ((int64_t)seq * L + kj) * p.lddv + h * D;
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float a8[8], b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a8[e] = dkacc[i][8 * g + e] * p.scale; b8[e] = dvacc[i][8 * g + e]; }
        store8(dK + i * 32 + 16 * g + 8 * half, a8);
        store8(dV + i * 32 + 16 * g + 8 * half, b8);
      }
  }
}

// delta[(seq,h), pos] = sum_d dO * O
template <typename T, int D>
__global__ void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta, int64_t M, int H,
                                  int L, int64_t ldo, int64_t lddo) {
  constexpr int G = D / 8;  // lanes per head
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = tid / (H * G);
  const int hc = (int)(tid % (H * G));
  float s = 0.f;
  if (row < M) {
    float a[8], b[8];
    load8(o + row * ldo + hc * 8, a);
    load8(dout + row * lddo + hc * 8, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += a[e] * b[e];
  }
#pragma unroll
  for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off, 64);
  if (row < M && (hc % G) == 0) {
    const int h = hc / G;
    const int64_t seq = row / L; const int pos = (int)(row % L);
    delta[(seq * H + h) * L + pos] = s;
  }
}

// xt[((seq*H + h)*D + d)*Lp + pos] = x[(seq*L + pos)*ldx + h*D + d], zero for pos in [L, Lp)
template <typename T, int D>
__global__ __launch_bounds__(256) void head_transpose_kernel(const T* __restrict__ x, T* __restrict__ xt, int H, int L, int Lp,
                                                             int64_t ldx) {
  __shared__ float tile[D][65];
  const int pc = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
  const int p0 = pc * 64;
  constexpr int G = D / 8;
  for (int idx = threadIdx.x; idx < 64 * G; idx += 256) {
    const int pos = idx / G, dg = idx % G;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (p0 + pos < L) load8(x + ((int64_t)seq * L + p0 + pos) * ldx + h * D + dg * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[dg * 8 + e][pos] = v[e];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < D * 8; idx += 256) {
    const int d = idx / 8, pg = idx % 8;
    if (p0 + pg * 8 < Lp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[d][pg * 8 + e];
      store8(xt + (((int64_t)seq * H + h) * D + d) * Lp + p0 + pg * 8, v);
    }
  }
}

// cosine-attention prep (attention.py:152-154): xhat = x / max(||x_head||, 1e-12) * scale_vec ; inv norms saved
template <typename T, int D>
__global__ void qk_norm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale_vec, T* __restrict__ y,
                                   float* __restrict__ inv_out, int64_t M, int H, int64_t ldx, int64_t ldy) {
  constexpr int G = D / 8;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = tid / (H * G);
  const int hc = (int)(tid % (H * G));
  float v[8];
  float s = 0.f;
  if (row < M) {
    load8(x + row * ldx + hc * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e] * v[e];
  }
#pragma unroll
  for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off, 64);
  if (row >= M) return;
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  const int d0 = (hc % G) * 8;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = v[e] * inv * scale_vec[d0 + e];
  store8(y + row * ldy + hc * 8, o);
  if ((hc % G) == 0) inv_out[row * H + hc / G] = inv;
}

// dx = inv * (g - u (u.g)),  g = dy * scale_vec, u = x * inv ;  dscale[d] += sum dy * u
template <typename T, int D>
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ inv_in, const float* __restrict__ scale_vec,
                                                          T* __restrict__ dx, float* __restrict__ dscale, int64_t M, int H,
                                                          int64_t lddy, int64_t ldx, int64_t lddx) {
  constexpr int G = D / 8;
  __shared__ float red[D];
  const int HG = H * G;
  // each thread keeps a fixed column group (grid stride is a multiple of HG)
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int hc = (int)(tid % HG);
  const int d0 = (hc % G) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int64_t rows_per_pass = nthreads / HG;
  for (int64_t base = 0; base < M; base += rows_per_pass) {
    const int64_t row = base + tid / HG;
    float g[8], u[8];
    float dot = 0.f, inv = 0.f;
    const bool ok = row < M;
    if (ok) {
      float a[8], b[8];
      load8(dy + row * lddy + hc * 8, a);
      load8(x + row * ldx + hc * 8, b);
      inv = inv_in[row * H + hc / G];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        u[e] = b[e] * inv;
        g[e] = a[e] * scale_vec[d0 + e];
        dot += u[e] * g[e];
        acc[e] += a[e] * u[e];
      }
    }
#pragma unroll
    for (int off = 1; off < G; off <<= 1) dot += __shfl_xor(dot, off, 64);
    if (ok) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = inv * (g[e] - u[e] * dot);
      store8(dx + row * lddx + hc * 8, o);
    }
  }
  if (!dscale) return;
  if (threadIdx.x < D) red[threadIdx.x] = 0.f;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(&red[d0 + e], acc[e]);
  __syncthreads();
  if (threadIdx.x < D) atomicAdd(dscale + threadIdx.x, red[threadIdx.x]);
}

template <typename T, int D>
int launch_attn(int which, const AttnParams& p, hipStream_t stream) {
  dim3 grid((unsigned)cdiv(cdiv(p.L, 32), 4), p.H, p.nseq), block(256);
  if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<T, D>), grid, block, 0, stream, p);
  else if (which == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D>), grid, block, 0, stream, p);
  return ctclip_check_launch("attention");
}

int dispatch_attn(int which, const AttnParams& p, int D, int dtype, hipStream_t stream) {
  if (dtype == DT_BF16 && D == 32) return launch_attn<bf16_t, 32>(which, p, stream);
  if (dtype == DT_BF16 && D == 64) return launch_attn<bf16_t, 64>(which, p, stream);
  if (dtype == DT_F32 && D == 32) return launch_attn<float, 32>(which, p, stream);
  if (dtype == DT_F32 && D == 64) return launch_attn<float, 64>(which, p, stream);
  ctclip_set_error("attention: head dim must be 32 or 64, dtype f32/bf16");
  return CTCLIP_EUNSUPPORTED;
}

bool bad_ld(int64_t ld) { return ld % 8 != 0; }

}  // namespace

#define DISPATCH_TD(NAME, ...)                                                            \
  if (dtype == DT_BF16 && D == 32) NAME<bf16_t, 32> __VA_ARGS__;                          \
  else if (dtype == DT_BF16 && D == 64) NAME<bf16_t, 64> __VA_ARGS__;                     \
  else if (dtype == DT_F32 && D == 32) NAME<float, 32> __VA_ARGS__;                       \
  else if (dtype == DT_F32 && D == 64) NAME<float, 64> __VA_ARGS__;                       \
  else { ctclip_set_error("head dim must be 32 or 64, dtype f32/bf16"); return CTCLIP_EUNSUPPORTED; }

// Per-(sequence, head) transposed copy: xt[seq][h][d][Lp] (Lp = L rounded up to 8, zero padded).
extern "C" int ctclip_head_transpose(const void* x, void* xt, int nseq, int H, int L, int Lp, int D, int64_t ldx, int dtype,
                                     hipStream_t stream) {
  if (!x || !xt || Lp % 8 || Lp < L || bad_ld(ldx)) { ctclip_set_error("head_transpose: Lp must be a multiple of 8 >= L, ld % 8 == 0"); return CTCLIP_EBADARG; }
  dim3 grid((unsigned)cdiv(Lp, 64), H, nseq);
#define ARGS (x_, xt_, H, L, Lp, ldx)
  if (dtype == DT_BF16) { auto x_ = (const bf16_t*)x; auto xt_ = (bf16_t*)xt;
    if (D == 32) hipLaunchKernelGGL((head_transpose_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, x_, xt_, H, L, Lp, ldx);
    else if (D == 64) hipLaunchKernelGGL((head_transpose_kernel<bf16_t, 64>), grid, dim3(256), 0, stream, x_, xt_, H, L, Lp, ldx);
    else return CTCLIP_EUNSUPPORTED;
  } else if (dtype == DT_F32) { auto x_ = (const float*)x; auto xt_ = (float*)xt;
    if (D == 32) hipLaunchKernelGGL((head_transpose_kernel<float, 32>), grid, dim3(256), 0, stream, x_, xt_, H, L, Lp, ldx);
    else if (D == 64) hipLaunchKernelGGL((head_transpose_kernel<float, 64>), grid, dim3(256), 0, stream, x_, xt_, H, L, Lp, ldx);
    else return CTCLIP_EUNSUPPORTED;
  } else return CTCLIP_EUNSUPPORTED;
#undef ARGS
  return ctclip_check_launch("head_transpose");
}

// attention.py:152-154 (l2norm(q) * q_scale per head); y may alias a different buffer with its own ld.
extern "C" int ctclip_qk_norm_fwd(const void* x, const float* scale_vec, void* y, float* inv, int64_t M, int H, int D, int64_t ldx,
                                  int64_t ldy, int dtype, hipStream_t stream) {
  if (!x || !y || !inv || !scale_vec || bad_ld(ldx) || bad_ld(ldy)) { ctclip_set_error("qk_norm_fwd: bad args"); return CTCLIP_EBADARG; }
  const int64_t nthreads = M * H * (D / 8);
  dim3 grid((unsigned)cdiv(nthreads, 256));
  if (dtype == DT_BF16 && D == 32) hipLaunchKernelGGL((qk_norm_fwd_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, (const bf16_t*)x, scale_vec, (bf16_t*)y, inv, M, H, ldx, ldy);
  else if (dtype == DT_BF16 && D == 64) hipLaunchKernelGGL((qk_norm_fwd_kernel<bf16_t, 64>), grid, dim3(256), 0, stream, (const bf16_t*)x, scale_vec, (bf16_t*)y, inv, M, H, ldx, ldy);
  else if (dtype == DT_F32 && D == 32) hipLaunchKernelGGL((qk_norm_fwd_kernel<float, 32>), grid, dim3(256), 0, stream, (const float*)x, scale_vec, (float*)y, inv, M, H, ldx, ldy);
  else if (dtype == DT_F32 && D == 64) hipLaunchKernelGGL((qk_norm_fwd_kernel<float, 64>), grid, dim3(256), 0, stream, (const float*)x, scale_vec, (float*)y, inv, M, H, ldx, ldy);
  else { ctclip_set_error("qk_norm: head dim must be 32 or 64"); return CTCLIP_EUNSUPPORTED; }
  return ctclip_check_launch("qk_norm_fwd");
}

extern "C" int ctclip_qk_norm_bwd(const void* dy, const void* x, const float* inv, const float* scale_vec, void* dx, float* dscale,
                                  int64_t M, int H, int D, int64_t lddy, int64_t ldx, int64_t lddx, int dtype, hipStream_t stream) {
  if (!dy || !x || !dx || !inv || !scale_vec || bad_ld(lddy) || bad_ld(ldx) || bad_ld(lddx)) { ctclip_set_error("qk_norm_bwd: bad args"); return CTCLIP_EBADARG; }
  const int HG = H * (D / 8);
  if (256 % (D / 8) != 0) return CTCLIP_EUNSUPPORTED;
  // grid stride must be a multiple of HG so that each thread keeps its column group: use lcm via blocks multiple
  int64_t want_rows = M < 4096 ? M : 4096;
  int64_t nthreads = want_rows * HG;
  int64_t nblocks = cdiv(nthreads, 256);
  // make nblocks*256 a multiple of HG
  while ((nblocks * 256) % HG) ++nblocks;
  dim3 grid((unsigned)nblocks);
  if (dtype == DT_BF16 && D == 32) hipLaunchKernelGGL((qk_norm_bwd_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, inv, scale_vec, (bf16_t*)dx, dscale, M, H, lddy, ldx, lddx);
  else if (dtype == DT_BF16 && D == 64) hipLaunchKernelGGL((qk_norm_bwd_kernel<bf16_t, 64>), grid, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, inv, scale_vec, (bf16_t*)dx, dscale, M, H, lddy, ldx, lddx);
  else if (dtype == DT_F32 && D == 32) hipLaunchKernelGGL((qk_norm_bwd_kernel<float, 32>), grid, dim3(256), 0, stream, (const float*)dy, (const float*)x, inv, scale_vec, (float*)dx, dscale, M, H, lddy, ldx, lddx);
  else if (dtype == DT_F32 && D == 64) hipLaunchKernelGGL((qk_norm_bwd_kernel<float, 64>), grid, dim3(256), 0, stream, (const float*)dy, (const float*)x, inv, scale_vec, (float*)dx, dscale, M, H, lddy, ldx, lddx);
  else { ctclip_set_error("qk_norm: head dim must be 32 or 64"); return CTCLIP_EUNSUPPORTED; }
  return ctclip_check_launch("qk_norm_bwd");
}

// softmax(scale * q k^T + bias[h] + keymask[seq]) v   (attention.py:156-178 / HF BertSelfAttention).
// q,k: (nseq*L, >= H*D) row-major views with row strides ldq/ldk; vt: ctclip_head_transpose of v.
// out (nseq*L, ldo); lse (nseq,H,L) f32 (may be null for inference).
extern "C" int ctclip_attn_fwd(const void* q, const void* k, const void* vt, const float* bias, const float* keymask, void* out,
                               float* lse, int nseq, int H, int L, int Lp, int D, int64_t ldq, int64_t ldk, int64_t ldo, float scale,
                               int dtype, hipStream_t stream) {
  if (!q || !k || !vt || !out || bad_ld(ldq) || bad_ld(ldk) || bad_ld(ldo) || Lp % 8 || Lp < L) { ctclip_set_error("attn_fwd: bad args"); return CTCLIP_EBADARG; }
  AttnParams p{};
  p.q = q; p.k = k; p.vt = vt; p.bias = bias; p.keymask = keymask; p.out = out; p.lse_out = lse;
  p.nseq = nseq; p.H = H; p.L = L; p.Lp = Lp; p.ldq = ldq; p.ldk = ldk; p.ldo = ldo; p.scale = scale;
  return dispatch_attn(0, p, D, dtype, stream);
}

static int dbias_nsplit(int nseq, int H, int L) {
  const int nkb = (L + 31) / 32;
  const int blocks = ((nkb * nkb + 3) / 4) * H;     // 4 tile pairs per block
  int ns = (2048 + blocks - 1) / blocks;
  if (ns > nseq) ns = nseq;
  if (ns > 8) ns = 8;
  if (ns < 1) ns = 1;
  return ns;
}
// bytes of workspace ctclip_attn_bwd needs when dbias is requested (per-split partial dBias slabs)
extern "C" int64_t ctclip_attn_bwd_workspace(int nseq, int H, int L) { return (int64_t)dbias_nsplit(nseq, H, L) * H * L * L * 4; }

// Backward.  Needs the transposed copies qt, kt (of q, k) and dot (of dout) and delta = rowsum(dO*O) (computed here
// into `delta`, (nseq,H,L) f32 scratch).  dbias (H,L,L) f32 is ACCUMULATED with atomics when non-null.
extern "C" int ctclip_attn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* o,
                               const void* dout, const void* dot, const float* lse, const float* bias, const float* keymask,
                               float* delta, void* dq, void* dk, void* dv, float* dbias, int nseq, int H, int L, int Lp, int D,
                               int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk,
                               int64_t lddv, float scale, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (!q || !k || !v || !qt || !kt || !o || !dout || !dot || !lse || !delta || !dq || !dk || !dv) { ctclip_set_error("attn_bwd: null arg"); return CTCLIP_EBADARG; }
  if (bad_ld(ldq) || bad_ld(ldk) || bad_ld(ldv) || bad_ld(ldo) || bad_ld(lddo) || bad_ld(lddq) || bad_ld(lddk) || bad_ld(lddv) || Lp % 8 || Lp < L) { ctclip_set_error("attn_bwd: strides must be multiples of 8"); return CTCLIP_EBADARG; }
  const int64_t M = (int64_t)nseq * L;
  {
    const int64_t nthreads = M * H * (D / 8);
    dim3 grid((unsigned)cdiv(nthreads, 256));
    if (dtype == DT_BF16 && D == 32) hipLaunchKernelGGL((attn_delta_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, (const bf16_t*)o, (const bf16_t*)dout, delta, M, H, L, ldo, lddo);
    else if (dtype == DT_BF16 && D == 64) hipLaunchKernelGGL((attn_delta_kernel<bf16_t, 64>), grid, dim3(256), 0, stream, (const bf16_t*)o, (const bf16_t*)dout, delta, M, H, L, ldo, lddo);
    else if (dtype == DT_F32 && D == 32) hipLaunchKernelGGL((attn_delta_kernel<float, 32>), grid, dim3(256), 0, stream, (const float*)o, (const float*)dout, delta, M, H, L, ldo, lddo);
    else if (dtype == DT_F32 && D == 64) hipLaunchKernelGGL((attn_delta_kernel<float, 64>), grid, dim3(256), 0, stream, (const float*)o, (const float*)dout, delta, M, H, L, ldo, lddo);
    else { ctclip_set_error("attention: head dim must be 32 or 64"); return CTCLIP_EUNSUPPORTED; }
    int rc = ctclip_check_launch("attn_delta");
    if (rc) return rc;
  }
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.qt = qt; p.kt = kt; p.o = o; p.dout = dout; p.dot = dot; p.lse = lse; p.delta = delta;
  p.bias = bias; p.keymask = keymask; p.dq = dq; p.dk = dk; p.dv = dv; p.dbias = dbias;
  p.nseq = nseq; p.H = H; p.L = L; p.Lp = Lp; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo;
  p.lddq = lddq; p.lddk = lddk; p.lddv = lddv; p.scale = scale;
  int rc;
  if (dbias) {
    // deterministic dBias: dedicated kernel (registers + per-split slabs), then the plain dQ kernel
    const int ns = dbias_nsplit(nseq, H, L);
    if (!workspace || workspace_bytes < ctclip_attn_bwd_workspace(nseq, H, L)) { ctclip_set_error("attn_bwd: workspace too small for the dBias slabs"); return CTCLIP_EWORKSPACE; }
    const int nkb = (L + 31) / 32;
    dim3 grid((unsigned)((nkb * nkb + 3) / 4), H, ns);
    float* ws = (float*)workspace;
    if (dtype == DT_BF16 && D == 32) hipLaunchKernelGGL((attn_bwd_dbias_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, p, ws, ns);
    else if (dtype == DT_BF16 && D == 64) hipLaunchKernelGGL((attn_bwd_dbias_kernel<bf16_t, 64>), grid, dim3(256), 0, stream, p, ws, ns);
    else if (dtype == DT_F32 && D == 32) hipLaunchKernelGGL((attn_bwd_dbias_kernel<float, 32>), grid, dim3(256), 0, stream, p, ws, ns);
    else if (dtype == DT_F32 && D == 64) hipLaunchKernelGGL((attn_bwd_dbias_kernel<float, 64>), grid, dim3(256), 0, stream, p, ws, ns);
    else return CTCLIP_EUNSUPPORTED;
    rc = ctclip_check_launch("attn_bwd_dbias");
    if (rc) return rc;
    const int64_t n = (int64_t)H * L * L;
    int64_t nb = cdiv(n, 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dbias_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)workspace, dbias, ns, n);
    rc = ctclip_check_launch("dbias_reduce");
    if (rc) return rc;
    p.dbias = nullptr;
  }
  rc = dispatch_attn(1, p, D, dtype, stream);
  if (rc) return rc;
  return dispatch_attn(2, p, D, dtype, stream);
}
