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
#include <stdlib.h>

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
  float drop_p, drop_inv_keep;   // attention-probability dropout (HF BertSelfAttention, train mode); 0 = off
  uint64_t drop_seed; const unsigned long long* drop_state;      // (drop_state: device-resident seed offset, ctclip_set_step_state)
  const float* bias_tab;   // relative-position form of the bias: (nclass, H) table, class(i, j) below; replaces `bias`
  int gh, gw;              // token grid of the sequence (L = gh * gw) when bias_tab is set
  void *out, *dq, *dk, *dv;
  float *lse_out, *dbias;
  int nseq, H, L, Lp;
  int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  float scale;
};

// Relative-position bias (attention.py:257-276): bias[h][i][j] = tab[class(i, j)][h] with class(i, j) = (iy - jy + gh - 1) (2 gw - 1)
// + (ix - jx + gw - 1) = u(i) - u(j) + const for u(t) = (t / gw) (2 gw - 1) + t % gw.  The expanded (H, L, L) f32 matrix was half of
// all the L2 traffic of the attention kernels (4 KB per 32 x 32 score tile and wave, re-read for every sequence); the table of one
// head is 8.8 KB at 24 x 24 tokens and lives in LDS, one gather per score.
constexpr int REL_MAXCLS = 4096, REL_MAXL = 1024;
struct RelLds {
  float tab[REL_MAXCLS];
  __attribute__((aligned(16))) uint16_t u[REL_MAXL + 32];
};
// whole workgroup; call before any early return
__device__ __forceinline__ void rel_stage(RelLds& rel, const AttnParams& p, int h, float mul = 1.f) {
  if (!p.bias_tab) return;                       // workgroup-uniform
  const int ncls = (2 * p.gh - 1) * (2 * p.gw - 1);
  for (int i = threadIdx.x; i < ncls; i += blockDim.x) rel.tab[i] = p.bias_tab[(int64_t)i * p.H + h] * mul;
  for (int i = threadIdx.x; i < p.L + 32; i += blockDim.x) {
    const int t = i < p.L ? i : p.L - 1;         // positions past L alias the last token: any valid class (they are masked)
    rel.u[i] = (uint16_t)((t / p.gw) * (2 * p.gw - 1) + t % p.gw);
  }
  __syncthreads();
}

// multipliers of the attention probabilities under dropout (common.h attn_drop_block: one Philox call per 2 queries x 4 keys; ctclip_attn_dropout_mask
// materialises the same values), so forward, dQ and dK / dV regenerate identical masks.
// Every block is needed by several lanes of a quad: each lane draws its share and the words travel by DPP quad permutes (a Philox call is ~40
// quarter-rate multiplies, a permute one vector move).
template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
// drop_keys: lane = ONE query qi (clamped; lane c <-> query block * 32 + c), registers = keys kb * 32 + slot_index(r, half).  The tile has four
// key quads per lane; the two lanes of a query PAIR (c, c ^ 1) need the same four blocks: each draws two (quads 2 s + (c & 1)) and swaps: 2 calls per tile.
__device__ __forceinline__ void drop_keys(float (&dm)[16], const AttnParams& p, int seq, int h, int qi, int kb, int half) {
  const uint64_t seed = p.drop_seed + (p.drop_state ? p.drop_state[0] : 0ull);
  const uint32_t thr = attn_drop_threshold(p.drop_p);
  const int64_t sh = (int64_t)seq * p.H + h;
  const int par = (int)(threadIdx.x & 1);          // = qi & 1 for unclamped rows (a clamped row is never stored)
  u32x4 own[2], oth[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    const int g = 2 * s_ + par;                    // key quad 16 (g >> 1) + 8 half + 4 (g & 1) of the tile
    int k0 = kb * 32 + 16 * (g >> 1) + 8 * half + 4 * (g & 1);
    k0 = k0 < p.L ? k0 : ((p.L - 1) & ~3);         // (keys past L are masked by the callers: any block in range will do)
    own[s_] = attn_drop_block(seed, sh, p.L, qi, k0);
#pragma unroll
    for (int w = 0; w < 4; ++w) oth[s_][w] = quad_perm<0xB1>(own[s_][w]);      // the neighbour's block: quad 2 s + (1 - par)
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int s_ = g >> 1;
    const bool mine = (g & 1) == par;
    // this query's two words of block g: (qi & 1) * 2 + {0, 1}
    const uint32_t lo = mine ? (par ? own[s_][2] : own[s_][0]) : (par ? oth[s_][2] : oth[s_][0]);
    const uint32_t hi = mine ? (par ? own[s_][3] : own[s_][1]) : (par ? oth[s_][3] : oth[s_][1]);
    const int r0 = 8 * (g >> 1) + 4 * (g & 1);
    dm[r0 + 0] = (lo & 0xffffu) >= thr ? p.drop_inv_keep : 0.f;
    dm[r0 + 1] = (lo >> 16) >= thr ? p.drop_inv_keep : 0.f;
    dm[r0 + 2] = (hi & 0xffffu) >= thr ? p.drop_inv_keep : 0.f;
    dm[r0 + 3] = (hi >> 16) >= thr ? p.drop_inv_keep : 0.f;
  }
}
// drop_queries: lane = ONE key kj (clamped; lane c <-> key block * 32 + c), registers = queries qb * 32 + slot_index(r, half).  The tile has eight
// query pairs per lane; the four lanes of a key QUAD need the same eight blocks: each draws two (pairs 2 (c & 3) + s) and broadcasts: 2 calls per tile.
__device__ __forceinline__ void drop_queries(float (&dm)[16], const AttnParams& p, int seq, int h, int kj, int qb, int half) {
  const uint64_t seed = p.drop_seed + (p.drop_state ? p.drop_state[0] : 0ull);
  const uint32_t thr = attn_drop_threshold(p.drop_p);
  const int64_t sh = (int64_t)seq * p.H + h;
  const int cq = (int)(threadIdx.x & 3);           // = kj & 3 for unclamped keys
  u32x4 own[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    const int g = 2 * cq + s_;                     // query pair 16 (g >> 2) + 8 half + 2 (g & 3) of the tile
    int q0 = qb * 32 + 16 * (g >> 2) + 8 * half + 2 * (g & 3);
    q0 = q0 < p.L ? q0 : ((p.L - 1) & ~1);
    own[s_] = attn_drop_block(seed, sh, p.L, q0, kj);
  }
  const bool hiword = (cq >> 1) != 0, hihalf = (cq & 1) != 0;       // this key's word (kj & 3) >> 1 of a query row, its half kj & 1
#define CTCLIP_DROPQ(G, CTRL)                                                                                         \
  {                                                                                                                   \
    const uint32_t a0 = quad_perm<CTRL>(own[(G) & 1][0]), a1 = quad_perm<CTRL>(own[(G) & 1][1]);                       \
    const uint32_t b0 = quad_perm<CTRL>(own[(G) & 1][2]), b1 = quad_perm<CTRL>(own[(G) & 1][3]);                       \
    const uint32_t w0 = hiword ? a1 : a0, w1 = hiword ? b1 : b0;       /* query row 0 / 1 of the pair */              \
    const int r0 = 8 * ((G) >> 2) + 2 * ((G) & 3);                                                                    \
    dm[r0] = (hihalf ? (w0 >> 16) : (w0 & 0xffffu)) >= thr ? p.drop_inv_keep : 0.f;                                   \
    dm[r0 + 1] = (hihalf ? (w1 >> 16) : (w1 & 0xffffu)) >= thr ? p.drop_inv_keep : 0.f;                               \
  }
  // pair g is drawn by lane g >> 1 of the quad: quad_perm [o, o, o, o] = o * 0x55
  CTCLIP_DROPQ(0, 0x00) CTCLIP_DROPQ(1, 0x00) CTCLIP_DROPQ(2, 0x55) CTCLIP_DROPQ(3, 0x55)
  CTCLIP_DROPQ(4, 0xAA) CTCLIP_DROPQ(5, 0xAA) CTCLIP_DROPQ(6, 0xFF) CTCLIP_DROPQ(7, 0xFF)
#undef CTCLIP_DROPQ
}

// scores of one 32x32 tile in "lane = column c, regs = rows slot_index(r, half)" layout -> logits
template <bool ROWS_ARE_KEYS>
__device__ __forceinline__ void tile_logits(float (&val)[16], const f32x16& s, const AttnParams& p, const RelLds& rel, int seq, int h,
                                            int col_idx, int row_base, int half) {
  // ROWS_ARE_KEYS: column = query (col_idx), rows = keys.  else: column = key, rows = queries.
  const int L = p.L;
  float add[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) add[r] = 0.f;
  if (p.bias_tab) {
    const int ucol = rel.u[col_idx < L ? col_idx : L - 1];
    const int c0 = (p.gh - 1) * (2 * p.gw - 1) + (p.gw - 1);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(rel.u + row_base + 16 * g + 8 * half);   // 8 consecutive rows (broadcast read)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int urow = (int)((w[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        add[8 * g + e] = rel.tab[ROWS_ARE_KEYS ? ucol - urow + c0 : urow - ucol + c0];
      }
    }
  } else if (p.bias) {   // wave-uniform branch; the loads inside are unconditional (indices clamped)
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
  // L <= 32 (CTViT temporal attention, L = 24): one row block per sequence, so the four waves take four sequences (the first
  // mapping left three of the four waves of every workgroup idle)
  const bool shortseq = p.L <= 32;
  const int qb = shortseq ? 0 : blockIdx.x * 4 + wave, h = blockIdx.y, seq = shortseq ? blockIdx.z * 4 + wave : blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  __shared__ RelLds rel;
  rel_stage(rel, p, h);
  if (qb * 32 >= L || seq >= p.nseq) return;
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
    tile_logits<true>(val, s, p, rel, seq, h, qi, kb * 32, half);
    float mx = val[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, val[r]);
    mx = half_max(mx);
    float mnew = fmaxf(m, mx);
    if (mnew == -INFINITY) mnew = 0.f;
    const float alpha = __expf(m - mnew);
    float pr[16], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { pr[r] = __expf(val[r] - mnew); ps += pr[r]; }
    lsum = lsum * alpha + ps;
    m = mnew;
    if (p.drop_p > 0.f) {   // dropout acts on the normalised probabilities: the row sum above stays undropped
      float dm[16];
      drop_keys(dm, p, seq, h, qic, kb, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] *= dm[r];
    }
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
  const float l = half_sum(lsum);
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
  // L <= 32 (CTViT temporal attention, L = 24): one row block per sequence, so the four waves take four sequences (the first
  // mapping left three of the four waves of every workgroup idle)
  const bool shortseq = p.L <= 32;
  const int qb = shortseq ? 0 : blockIdx.x * 4 + wave, h = blockIdx.y, seq = shortseq ? blockIdx.z * 4 + wave : blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  __shared__ RelLds rel;
  rel_stage(rel, p, h);
  if (qb * 32 >= L || seq >= p.nseq) return;
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
    tile_logits<true>(val, s, p, rel, seq, h, qi, kb * 32, half);
    float dm[16];
    if (p.drop_p > 0.f) drop_keys(dm, p, seq, h, qic, kb, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = kb * 32 + slot_index(r, half);
      const float pr = (qi < L && kj < L) ? __expf(val[r] - lse) : 0.f;
      const float dpr = p.drop_p > 0.f ? dp[r] * dm[r] : dp[r];   // d(dropout(P)) / dP
      ds[r] = pr * (dpr - delta);
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
    if (p.bias_tab) {
      const int cls = (qic / p.gw - kc / p.gw + p.gh - 1) * (2 * p.gw - 1) + (qic % p.gw - kc % p.gw + p.gw - 1);
      add[r] = p.bias_tab[(int64_t)cls * p.H + h];
    } else add[r] = p.bias ? p.bias[((int64_t)h * L + qic) * L + kc] : 0.f;
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

__global__ void dbias_reduce_kernel(const float* __restrict__ part, float* __restrict__ dbias, int nsplit, int64_t n, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += part[(int64_t)s * n + i];
    dbias[i] = accumulate ? dbias[i] + t : t;
  }
}

// dK, dV: one wave per 32-key block, loop over query tiles.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool shortseq = p.L <= 32;                          // see attn_fwd_kernel
  const int jb = shortseq ? 0 : blockIdx.x * 4 + wave, h = blockIdx.y, seq = shortseq ? blockIdx.z * 4 + wave : blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  __shared__ RelLds rel;
  rel_stage(rel, p, h);
  if (jb * 32 >= L || seq >= p.nseq) return;
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
    tile_logits<false>(val, s, p, rel, seq, h, kj, qb * 32, half);
    float dmq[16];
    if (p.drop_p > 0.f) drop_queries(dmq, p, seq, h, kjc, qb, half);
    // the tile's row statistics: two runs of eight consecutive queries per lane -- four 16-byte loads each when L is a multiple of 8 (BERT: 128 / 512)
    // instead of 32 scalar ones (round 6)
    float lse16[16], del16[16];
    if ((L & 7) == 0) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int q0 = qb * 32 + 16 * g + 8 * half;
        q0 = q0 < L ? q0 : L - 8;                  // (rows past L come in whole runs of eight: their probabilities are zeroed below)
        const f32x4 l0 = *reinterpret_cast<const f32x4*>(p.lse + statbase + q0), l1 = *reinterpret_cast<const f32x4*>(p.lse + statbase + q0 + 4);
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(p.delta + statbase + q0), d1 = *reinterpret_cast<const f32x4*>(p.delta + statbase + q0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { lse16[8 * g + e] = l0[e]; lse16[8 * g + 4 + e] = l1[e]; del16[8 * g + e] = d0[e]; del16[8 * g + 4 + e] = d1[e]; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qb * 32 + slot_index(r, half);
        const int qc = qi < L ? qi : L - 1;
        lse16[r] = p.lse[statbase + qc]; del16[r] = p.delta[statbase + qc];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = qb * 32 + slot_index(r, half);
      const int qc = qi < L ? qi : L - 1;
      const float lse = lse16[r], delta = del16[r];
      pr[r] = (qi < L && kj < L) ? __expf(val[r] - lse) : 0.f;
      const float dm = p.drop_p > 0.f ? dmq[r] : 1.f;
      ds[r] = pr[r] * (dp[r] * dm - delta);
      pr[r] *= dm;                                           // dV sees the dropped probabilities
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

// FAST path of the LDS-shared kernels (decided per launch): L % 32 == 0 (every tile is full: no masks), no key-padding mask, bias
// absent or a relative-position table with gw % 8 == 0.  PMC showed these kernels VALU-bound (291 VALU instructions per 32 x 32 tile
// and wave in the forward, ~4 clk each, SQ_ACTIVE_INST_VALU = 64 % of the wall time), so the fast path is a VALU diet:
//  * logits in the log2 domain (scale and table pre-multiplied by log2 e): exp2 is the hardware instruction, no multiply per element;
//  * the 8 keys of a fragment run lie in one image row when gw % 8 == 0, so their bias classes are CONSECUTIVE: one index per run
//    and eight LDS reads with immediate offsets instead of an unpack, a subtract and a shift per element;
//  * no bounds selects; the accumulator rescale is skipped while no lane's running maximum moved;
//  * MFMA results stay in VGPRs (-mllvm -amdgpu-mfma-vgpr-form for this file): the AGPR form cost ~96 copies per tile.
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
template <bool ROWS_ARE_KEYS>
__device__ __forceinline__ void tile_logits_fast(float (&val)[16], const f32x16& s, const AttnParams& p, const RelLds& rel, int col_idx,
                                                 int row_base, int half, float scale2) {
  if (p.bias_tab) {
    const int ucol = rel.u[col_idx];
    const int c0 = (p.gh - 1) * (2 * p.gw - 1) + (p.gw - 1);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int urow0 = rel.u[row_base + 16 * g + 8 * half];
      if (ROWS_ARE_KEYS) {
        const float* b = rel.tab + (ucol - urow0 + c0 - 7);      // class of key e = class of key 0 - e
#pragma unroll
        for (int e = 0; e < 8; ++e) val[8 * g + e] = fmaf(s[8 * g + e], scale2, b[7 - e]);
      } else {
        const float* b = rel.tab + (urow0 - ucol + c0);          // class of query e = class of query 0 + e
#pragma unroll
        for (int e = 0; e < 8; ++e) val[8 * g + e] = fmaf(s[8 * g + e], scale2, b[e]);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) val[r] = s[r] * scale2;
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// LDS-shared variants for the CTViT shape (bf16, d_head 32, L >= 128).  The four waves of a workgroup own four neighbouring
// 32-row blocks of the SAME (sequence, head) and walk the other axis in lockstep, so the operand tiles of a step (32 rows x 64 B
// each) are fetched ONCE per workgroup -- one 16-byte load per thread -- into a double-buffered LDS ring and read from there by all
// four waves.  The register-only kernels above re-read every tile per wave: 4-6 KB of L2 traffic and ~200 line lookups per wave
// and step made them L2 / address-path bound (358 us for the 65-GFLOP spatial forward without bias).
// LDS rows are 80 B apart (64 B of data): the sixteen rows one ds_read_b128 service group touches then start in sixteen different
// 16-byte bank groups (20 r mod 64 is a permutation for the row sets of the fragment layout).
// ----------------------------------------------------------------------------------------------------------------------
constexpr int SROW = 80, STILE = 32 * SROW;

__device__ __forceinline__ void lds_frag(Frag<bf16_t, 32>& f, const char* tile, int ar, int half) {
  const char* r = tile + ar * SROW + half * 16;
  f.v[0] = *reinterpret_cast<const bf16x8*>(r);
  f.v[1] = *reinterpret_cast<const bf16x8*>(r + 32);
}
// source of this thread's 16 bytes of a staged tile.  Row-major operand (rows = tokens of the tile): `row0` = first token.
__device__ __forceinline__ const bf16_t* src_rows(const bf16_t* base, int64_t seq_row0, int tok0, int L, int64_t ld, int hcol, int srow, int schunk) {
  int t = tok0 + srow;
  t = t < L ? t : L - 1;                                   // clamped (never branch around a load); masked by the consumer
  return base + (seq_row0 + t) * ld + hcol + schunk * 8;
}
// transposed copy (rows = head dims, columns = tokens): xt[(seq, h)][d][Lp]
__device__ __forceinline__ const bf16_t* src_cols(const bf16_t* base_sh, int tok0, int Lp, int srow, int schunk) {
  int c = tok0 + schunk * 8;
  c = c + 8 <= Lp ? c : Lp - 8;
  return base_sh + (int64_t)srow * Lp + c;
}

template <bool FAST>
__global__ __launch_bounds__(256) void attn_fwd_lds_kernel(AttnParams p) {
  typedef bf16_t T;
  constexpr int D = 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  const int nkb = (L + 31) / 32;
  __shared__ RelLds rel;
  __shared__ __attribute__((aligned(16))) char tiles[2][2][STILE];      // [buffer][K, V^T]
  rel_stage(rel, p, h, FAST ? LOG2E : 1.f);
  const float scale2 = p.scale * LOG2E;
  const int qb_raw = blockIdx.x * 4 + wave;
  const bool active = qb_raw * 32 < L;
  const int qb = active ? qb_raw : nkb - 1;                // idle waves shadow the last block: they must keep the barriers
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* Vt = reinterpret_cast<const T*>(p.vt) + ((int64_t)seq * p.H + h) * D * Lp;
  const int qic = qi < L ? qi : L - 1;
  Frag<T, D> qf;
  frag_load(qf, Q + ((int64_t)seq * L + qic) * p.ldq + h * D, half, D);

  // staging role of this thread: tile (0 = K rows, 1 = V^T rows), row, 16-byte chunk
  const int stile = threadIdx.x >> 7, srow = (threadIdx.x & 127) >> 2, schunk = threadIdx.x & 3;
  auto gsrc = [&](int kb) {
    const T* a = src_rows(K, (int64_t)seq * L, kb * 32, L, p.ldk, h * D, srow, schunk);
    const T* b = src_cols(Vt, kb * 32, Lp, srow, schunk);
    return stile == 0 ? a : b;
  };
  char* sdst = &tiles[0][stile][0] + srow * SROW + schunk * 16;
  u32x4 staged = *reinterpret_cast<const u32x4*>(gsrc(0));
  *reinterpret_cast<u32x4*>(sdst) = staged;
  __syncthreads();

  float m = -INFINITY, lsum = 0.f;
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const int buf = kb & 1;
    staged = *reinterpret_cast<const u32x4*>(gsrc(kb + 1 < nkb ? kb + 1 : kb));     // next tiles in flight under this step
    Frag<T, D> kf, vf;
    lds_frag(kf, tiles[buf][0], ar, half);
    lds_frag(vf, tiles[buf][1], ar, half);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    s = mma(s, kf, qf);
    float val[16];
    if (FAST) tile_logits_fast<true>(val, s, p, rel, qi, kb * 32, half, scale2);
    else tile_logits<true>(val, s, p, rel, seq, h, qi, kb * 32, half);
    float mx = val[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, val[r]);
    mx = half_max(mx);
    float mnew = fmaxf(m, mx);
    if (!FAST && mnew == -INFINITY) mnew = 0.f;
    const float alpha = FAST ? __builtin_amdgcn_exp2f(m - mnew) : __expf(m - mnew);
    float pr[16], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { pr[r] = FAST ? __builtin_amdgcn_exp2f(val[r] - mnew) : __expf(val[r] - mnew); ps += pr[r]; }
    lsum = lsum * alpha + ps;
    Frag<T, 32> pf;
    frag_from_regs(pf, pr);
    if (!FAST || __builtin_amdgcn_ballot_w64(mnew != m) != 0) {        // wave-uniform: skip the rescale while no maximum moved
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    }
    m = mnew;
    oacc = mma(oacc, vf, pf);
    *reinterpret_cast<u32x4*>(sdst + (buf ^ 1) * 2 * STILE) = staged;   // every wave finished reading that buffer before the last barrier
    __syncthreads();
  }
  const float l = half_sum(lsum);
  if (active && qi < L) {
    const float inv = 1.f / l;
    T* O = reinterpret_cast<T*>(p.out) + ((int64_t)seq * L + qi) * p.ldo + h * D;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = oacc[8 * g + e] * inv;
      store8(O + 16 * g + 8 * half, o8);
    }
    if (half == 0 && p.lse_out) p.lse_out[((int64_t)seq * p.H + h) * L + qi] = FAST ? (m + __log2f(l)) * LN2 : m + __logf(l);
  }
}

// (A fused table-gradient variant of this kernel -- dS added to an LDS copy of the table with ds_add_f32, one flush per workgroup
// -- was measured and dropped: LDS float atomics retire about one LANE per 3 clk per CU, 510 M of them per layer cost 2.8 ms against
// 0.75 ms for the separate attn_bwd_dbias_kernel pass.)
template <bool FAST>
__global__ __launch_bounds__(256) void attn_bwd_dq_lds_kernel(AttnParams p) {
  typedef bf16_t T;
  constexpr int D = 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y;
  const int L = p.L, Lp = p.Lp;
  const int nkb = (L + 31) / 32;
  __shared__ RelLds rel;
  __shared__ __attribute__((aligned(16))) char tiles[2][3][STILE];      // [buffer][K, V, K^T]
  rel_stage(rel, p, h, FAST ? LOG2E : 1.f);
  const float scale2 = p.scale * LOG2E;
  const int seq = blockIdx.z;
  const int qb_raw = blockIdx.x * 4 + wave;
  const bool active = qb_raw * 32 < L;
  const int qb = active ? qb_raw : nkb - 1;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  const T* Kt = reinterpret_cast<const T*>(p.kt) + ((int64_t)seq * p.H + h) * D * Lp;
  const T* dO = reinterpret_cast<const T*>(p.dout);
  const int qic = qi < L ? qi : L - 1;
  Frag<T, D> qf, dof;
  frag_load(qf, Q + ((int64_t)seq * L + qic) * p.ldq + h * D, half, D);
  frag_load(dof, dO + ((int64_t)seq * L + qic) * p.lddo + h * D, half, D);
  const float lse = p.lse[((int64_t)seq * p.H + h) * L + qic] * (FAST ? LOG2E : 1.f);
  const float delta = p.delta[((int64_t)seq * p.H + h) * L + qic];

  const int stile = threadIdx.x >> 7, srow = (threadIdx.x & 127) >> 2, schunk = threadIdx.x & 3;
  auto gsrc0 = [&](int kb) {                                  // pass 0: K (threads 0-127) and V (128-255)
    const T* a = src_rows(K, (int64_t)seq * L, kb * 32, L, p.ldk, h * D, srow, schunk);
    const T* b = src_rows(V, (int64_t)seq * L, kb * 32, L, p.ldv, h * D, srow, schunk);
    return stile == 0 ? a : b;
  };
  auto gsrc1 = [&](int kb) { return src_cols(Kt, kb * 32, Lp, srow, schunk); };   // pass 1: K^T, by every thread (the upper half's copy is unused)
  char* sdst0 = &tiles[0][stile][0] + srow * SROW + schunk * 16;
  char* sdst1 = &tiles[0][2][0] + srow * SROW + schunk * 16;
  u32x4 st0 = *reinterpret_cast<const u32x4*>(gsrc0(0)), st1 = *reinterpret_cast<const u32x4*>(gsrc1(0));
  *reinterpret_cast<u32x4*>(sdst0) = st0;
  if (stile == 0) *reinterpret_cast<u32x4*>(sdst1) = st1;
  __syncthreads();

  f32x16 dqacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const int buf = kb & 1, kbn = kb + 1 < nkb ? kb + 1 : kb;
    st0 = *reinterpret_cast<const u32x4*>(gsrc0(kbn));
    st1 = *reinterpret_cast<const u32x4*>(gsrc1(kbn));
    Frag<T, D> kf, vf, ktf;
    lds_frag(kf, tiles[buf][0], ar, half);
    lds_frag(vf, tiles[buf][1], ar, half);
    lds_frag(ktf, tiles[buf][2], ar, half);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, kf, qf);
    dp = mma(dp, vf, dof);
    float val[16], ds[16];
    if (FAST) {
      tile_logits_fast<true>(val, s, p, rel, qi, kb * 32, half, scale2);
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(val[r] - lse) * (dp[r] - delta);
    } else {
      tile_logits<true>(val, s, p, rel, seq, h, qi, kb * 32, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kj = kb * 32 + slot_index(r, half);
        const float pr = (qi < L && kj < L) ? __expf(val[r] - lse) : 0.f;
        ds[r] = pr * (dp[r] - delta);
      }
    }
    Frag<T, 32> dsf;
    frag_from_regs(dsf, ds);
    dqacc = mma(dqacc, ktf, dsf);
    *reinterpret_cast<u32x4*>(sdst0 + (buf ^ 1) * 3 * STILE) = st0;
    if (stile == 0) *reinterpret_cast<u32x4*>(sdst1 + (buf ^ 1) * 3 * STILE) = st1;
    __syncthreads();
  }
  if (active && qi < L) {
    T* dQ = reinterpret_cast<T*>(p.dq) + ((int64_t)seq * L + qi) * p.lddq + h * D;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = dqacc[8 * g + e] * p.scale;
      store8(dQ + 16 * g + 8 * half, o8);
    }
  }
}

template <bool FAST>
__global__ __launch_bounds__(256) void attn_bwd_dkv_lds_kernel(AttnParams p) {
  typedef bf16_t T;
  constexpr int D = 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp;
  const int nqb = (L + 31) / 32;
  __shared__ RelLds rel;
  __shared__ __attribute__((aligned(16))) char tiles[2][4][STILE];      // [buffer][Q, dO, Q^T, dO^T]
  __shared__ float stats[2][2][32];                                      // [buffer][lse, delta][query of the tile]
  rel_stage(rel, p, h, FAST ? LOG2E : 1.f);
  const float scale2 = p.scale * LOG2E;
  const int jb_raw = blockIdx.x * 4 + wave;
  const bool active = jb_raw * 32 < L;
  const int jb = active ? jb_raw : nqb - 1;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int kj = jb * 32 + c;
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  const T* dO = reinterpret_cast<const T*>(p.dout);
  const T* Qt = reinterpret_cast<const T*>(p.qt) + ((int64_t)seq * p.H + h) * D * Lp;
  const T* dOt = reinterpret_cast<const T*>(p.dot) + ((int64_t)seq * p.H + h) * D * Lp;
  const int kjc = kj < L ? kj : L - 1;
  Frag<T, D> kf, vf;
  frag_load(kf, K + ((int64_t)seq * L + kjc) * p.ldk + h * D, half, D);
  frag_load(vf, V + ((int64_t)seq * L + kjc) * p.ldv + h * D, half, D);
  const int64_t statbase = ((int64_t)seq * p.H + h) * L;

  const int stile = threadIdx.x >> 7, srow = (threadIdx.x & 127) >> 2, schunk = threadIdx.x & 3;
  auto gsrc0 = [&](int qb) {                                  // pass 0: Q (threads 0-127) and dO (128-255)
    const T* a = src_rows(Q, (int64_t)seq * L, qb * 32, L, p.ldq, h * D, srow, schunk);
    const T* b = src_rows(dO, (int64_t)seq * L, qb * 32, L, p.lddo, h * D, srow, schunk);
    return stile == 0 ? a : b;
  };
  auto gsrc1 = [&](int qb) {                                  // pass 1: Q^T and dO^T
    const T* a = src_cols(Qt, qb * 32, Lp, srow, schunk);
    const T* b = src_cols(dOt, qb * 32, Lp, srow, schunk);
    return stile == 0 ? a : b;
  };
  auto gstat = [&](int qb) {                                  // threads 0-31: lse, 32-63: delta of the tile's 32 queries
    int q = qb * 32 + (threadIdx.x & 31);
    q = q < L ? q : L - 1;
    return ((threadIdx.x & 32) ? p.delta : p.lse) + statbase + q;
  };
  char* sdst0 = &tiles[0][stile][0] + srow * SROW + schunk * 16;
  char* sdst1 = &tiles[0][2 + stile][0] + srow * SROW + schunk * 16;
  float* sstat = &stats[0][(threadIdx.x >> 5) & 1][threadIdx.x & 31];
  u32x4 st0 = *reinterpret_cast<const u32x4*>(gsrc0(0)), st1 = *reinterpret_cast<const u32x4*>(gsrc1(0));
  const float smul = (FAST && !(threadIdx.x & 32)) ? LOG2E : 1.f;   // lse is kept in the log2 domain on the fast path
  float sv = *gstat(0) * smul;
  *reinterpret_cast<u32x4*>(sdst0) = st0;
  *reinterpret_cast<u32x4*>(sdst1) = st1;
  if (threadIdx.x < 64) *sstat = sv;
  __syncthreads();

  f32x16 dkacc, dvacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkacc[r] = 0.f; dvacc[r] = 0.f; }
  for (int qb = 0; qb < nqb; ++qb) {
    const int buf = qb & 1, qbn = qb + 1 < nqb ? qb + 1 : qb;
    st0 = *reinterpret_cast<const u32x4*>(gsrc0(qbn));
    st1 = *reinterpret_cast<const u32x4*>(gsrc1(qbn));
    sv = *gstat(qbn) * smul;
    Frag<T, D> qf, dof, qtf, dotf;
    lds_frag(qf, tiles[buf][0], ar, half);
    lds_frag(dof, tiles[buf][1], ar, half);
    lds_frag(qtf, tiles[buf][2], ar, half);
    lds_frag(dotf, tiles[buf][3], ar, half);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, qf, kf);     // D[query rho][key c]
    dp = mma(dp, dof, vf);
    float val[16], pr[16], ds[16];
    if (FAST) tile_logits_fast<false>(val, s, p, rel, kj, qb * 32, half, scale2);
    else tile_logits<false>(val, s, p, rel, seq, h, kj, qb * 32, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int sl = slot_index(r, half);
      const int qi = qb * 32 + sl;
      const float lse = stats[buf][0][sl], delta = stats[buf][1][sl];
      if (FAST) pr[r] = __builtin_amdgcn_exp2f(val[r] - lse);
      else pr[r] = (qi < L && kj < L) ? __expf(val[r] - lse) : 0.f;
      ds[r] = pr[r] * (dp[r] - delta);
    }
    Frag<T, 32> pf, dsf;
    frag_from_regs(pf, pr);
    frag_from_regs(dsf, ds);
    dvacc = mma(dvacc, dotf, pf);
    dkacc = mma(dkacc, qtf, dsf);
    *reinterpret_cast<u32x4*>(sdst0 + (buf ^ 1) * 4 * STILE) = st0;
    *reinterpret_cast<u32x4*>(sdst1 + (buf ^ 1) * 4 * STILE) = st1;
    if (threadIdx.x < 64) sstat[(buf ^ 1) * 64] = sv;
    __syncthreads();
  }
  if (active && kj < L) {
    T* dK = reinterpret_cast<T*>(p.dk) + ((int64_t)seq * L + kj) * p.lddk + h * D;
    T* dV = reinterpret_cast<T*>(p.dv) + ((int64_t)seq * L + kj) * p.lddv + h * D;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float a8[8], b8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a8[e] = dkacc[8 * g + e] * p.scale; b8[e] = dvacc[8 * g + e]; }
      store8(dK + 16 * g + 8 * half, a8);
      store8(dV + 16 * g + 8 * half, b8);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// LDS-shared variants for the BERT shape (bf16, d_head 64, L % 32 == 0, optional key-padding mask and attention dropout; round 6).
// The register-only kernels above let every lane fetch its own operand row of every tile (32 cache lines per load instruction, 64 for the
// transposed copies) and every wave of a workgroup fetch the same tiles again: at T = 512 they ran at the L1 line rate (fwd 52 / dQ 98 /
// dK,dV 123 us per layer for 0.8 / 2 / 2 GFLOP x 96 heads).  Here the four waves of a workgroup own four neighbouring 32-row blocks of one
// (sequence, head) and walk the other axis in lockstep; each operand tile of a step is fetched once per workgroup with one 16-byte load
// per thread (8 threads per 128-byte row) into a double-buffered LDS ring.  Row-major tiles (32 tokens x 128 B) are 144 B apart in LDS,
// transposed ones (64 dims x 64 B) 80 B: 9 r and 5 r mod 16 are permutations, so the sixteen rows of a ds_read_b128 group start in
// sixteen different 16-byte bank groups.  The key mask (x log2 e, clamped finite) and the row statistics of the backward sit in LDS for
// the whole kernel; logits are in the log2 domain.
// ----------------------------------------------------------------------------------------------------------------------
constexpr int SROW64 = 144, STILE64 = 32 * SROW64, STILE64T = 64 * SROW;

__device__ __forceinline__ void lds_frag64(Frag<bf16_t, 64>& f, const char* tile, int ar, int half) {
  const char* r = tile + ar * SROW64 + half * 16;
#pragma unroll
  for (int g = 0; g < 4; ++g) f.v[g] = *reinterpret_cast<const bf16x8*>(r + 32 * g);
}
// key mask of the sequence -> LDS, in the log2 domain; HF's mask is finfo.min: keep it finite so that a fully masked tile stays NaN-free
__device__ __forceinline__ void stage_keymask(float* km, const AttnParams& p, int seq) {
  if (!p.keymask) return;
  for (int i = threadIdx.x; i < p.L; i += blockDim.x) km[i] = fmaxf(p.keymask[(int64_t)seq * p.L + i] * LOG2E, -3.0e38f);
}
// two runs of eight consecutive f32 (rows 16 g + 8 half of a tile) from an LDS array: four broadcast ds_read_b128
__device__ __forceinline__ void lds_runs(float (&out)[16], const float* arr, int base, int half) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(arr + base + 16 * g + 8 * half), b = *reinterpret_cast<const f32x4*>(arr + base + 16 * g + 8 * half + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { out[8 * g + e] = a[e]; out[8 * g + 4 + e] = b[e]; }
  }
}

__global__ __launch_bounds__(256) void attn64_fwd_kernel(AttnParams p) {
  typedef bf16_t T;
  constexpr int D = 64, BUF = STILE64 + STILE64T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp, nkb = L / 32;
  __shared__ __attribute__((aligned(16))) char tiles[2 * BUF];          // [buffer][K rows | V^T rows]
  __shared__ __attribute__((aligned(16))) float km[REL_MAXL];
  stage_keymask(km, p, seq);
  const float scale2 = p.scale * LOG2E;
  const int qb_raw = blockIdx.x * 4 + wave;
  const bool active = qb_raw < nkb;
  const int qb = active ? qb_raw : nkb - 1;                // idle waves shadow the last block: they must keep the barriers
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  Frag<T, D> qf;
  frag_load(qf, reinterpret_cast<const T*>(p.q) + ((int64_t)seq * L + qi) * p.ldq + h * D, half, D);

  const T* ksrc = reinterpret_cast<const T*>(p.k) + ((int64_t)seq * L + (tid >> 3)) * p.ldk + h * D + (tid & 7) * 8;
  const T* vsrc = reinterpret_cast<const T*>(p.vt) + (((int64_t)seq * p.H + h) * D + (tid >> 2)) * Lp + (tid & 3) * 8;
  char* kdst = tiles + (tid >> 3) * SROW64 + (tid & 7) * 16;
  char* vdst = tiles + STILE64 + (tid >> 2) * SROW + (tid & 3) * 16;
  const int64_t kstep = 32 * p.ldk;
  u32x4 sk = *reinterpret_cast<const u32x4*>(ksrc), sv = *reinterpret_cast<const u32x4*>(vsrc);
  *reinterpret_cast<u32x4*>(kdst) = sk;
  *reinterpret_cast<u32x4*>(vdst) = sv;
  __syncthreads();

  float m = -INFINITY, lsum = 0.f;
  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const char* buf = tiles + (kb & 1) * BUF;
    const int kbn = kb + 1 < nkb ? kb + 1 : kb;
    sk = *reinterpret_cast<const u32x4*>(ksrc + kbn * kstep);          // next tiles in flight under this step
    sv = *reinterpret_cast<const u32x4*>(vsrc + kbn * 32);
    Frag<T, D> kf;
    Frag<T, 32> vf[2];
    lds_frag64(kf, buf, ar, half);
    lds_frag(vf[0], buf + STILE64, ar, half);
    lds_frag(vf[1], buf + STILE64 + 32 * SROW, ar, half);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    s = mma(s, kf, qf);
    float val[16];
    if (p.keymask) {
      float add[16];
      lds_runs(add, km, kb * 32, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) val[r] = fmaf(s[r], scale2, add[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) val[r] = s[r] * scale2;
    }
    float mx = val[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, val[r]);
    mx = half_max(mx);
    const float mnew = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    float pr[16], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(val[r] - mnew); ps += pr[r]; }
    lsum = lsum * alpha + ps;
    if (p.drop_p > 0.f) {   // dropout acts on the normalised probabilities: the row sum above stays undropped
      float dm[16];
      drop_keys(dm, p, seq, h, qi, kb, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] *= dm[r];
    }
    Frag<T, 32> pf;
    frag_from_regs(pf, pr);
    if (__builtin_amdgcn_ballot_w64(mnew != m) != 0) {                  // wave-uniform: skip the rescale while no maximum moved
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    m = mnew;
    oacc[0] = mma(oacc[0], vf[0], pf);
    oacc[1] = mma(oacc[1], vf[1], pf);
    const int nb = ((kb & 1) ^ 1) * BUF;                                // every wave finished reading that buffer before the last barrier
    *reinterpret_cast<u32x4*>(kdst + nb) = sk;
    *reinterpret_cast<u32x4*>(vdst + nb) = sv;
    __syncthreads();
  }
  const float l = half_sum(lsum);
  if (active) {
    const float inv = 1.f / l;
    T* O = reinterpret_cast<T*>(p.out) + ((int64_t)seq * L + qi) * p.ldo + h * D;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = oacc[i][8 * g + e] * inv;
        store8(O + i * 32 + 16 * g + 8 * half, o8);
      }
    if (half == 0 && p.lse_out) p.lse_out[((int64_t)seq * p.H + h) * L + qi] = (m + __log2f(l)) * LN2;
  }
}

__global__ __launch_bounds__(256) void attn64_bwd_dq_kernel(AttnParams p) {
  typedef bf16_t T;
  constexpr int D = 64, BUF = 2 * STILE64 + STILE64T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp, nkb = L / 32;
  __shared__ __attribute__((aligned(16))) char tiles[2 * BUF];          // [buffer][K rows | V rows | K^T rows]
  __shared__ __attribute__((aligned(16))) float km[REL_MAXL];
  stage_keymask(km, p, seq);
  const float scale2 = p.scale * LOG2E;
  const int qb_raw = blockIdx.x * 4 + wave;
  const bool active = qb_raw < nkb;
  const int qb = active ? qb_raw : nkb - 1;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  Frag<T, D> qf, dof;
  frag_load(qf, reinterpret_cast<const T*>(p.q) + ((int64_t)seq * L + qi) * p.ldq + h * D, half, D);
  frag_load(dof, reinterpret_cast<const T*>(p.dout) + ((int64_t)seq * L + qi) * p.lddo + h * D, half, D);
  const float lse2 = p.lse[((int64_t)seq * p.H + h) * L + qi] * LOG2E;
  const float delta = p.delta[((int64_t)seq * p.H + h) * L + qi];

  const T* ksrc = reinterpret_cast<const T*>(p.k) + ((int64_t)seq * L + (tid >> 3)) * p.ldk + h * D + (tid & 7) * 8;
  const T* vsrc = reinterpret_cast<const T*>(p.v) + ((int64_t)seq * L + (tid >> 3)) * p.ldv + h * D + (tid & 7) * 8;
  const T* tsrc = reinterpret_cast<const T*>(p.kt) + (((int64_t)seq * p.H + h) * D + (tid >> 2)) * Lp + (tid & 3) * 8;
  char* kdst = tiles + (tid >> 3) * SROW64 + (tid & 7) * 16;
  char* vdst = kdst + STILE64;
  char* tdst = tiles + 2 * STILE64 + (tid >> 2) * SROW + (tid & 3) * 16;
  const int64_t kstep = 32 * p.ldk, vstep = 32 * p.ldv;
  u32x4 sk = *reinterpret_cast<const u32x4*>(ksrc), sv = *reinterpret_cast<const u32x4*>(vsrc), st = *reinterpret_cast<const u32x4*>(tsrc);
  *reinterpret_cast<u32x4*>(kdst) = sk;
  *reinterpret_cast<u32x4*>(vdst) = sv;
  *reinterpret_cast<u32x4*>(tdst) = st;
  __syncthreads();

  f32x16 dqacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const char* buf = tiles + (kb & 1) * BUF;
    const int kbn = kb + 1 < nkb ? kb + 1 : kb;
    sk = *reinterpret_cast<const u32x4*>(ksrc + kbn * kstep);
    sv = *reinterpret_cast<const u32x4*>(vsrc + kbn * vstep);
    st = *reinterpret_cast<const u32x4*>(tsrc + kbn * 32);
    Frag<T, D> kf, vf;
    Frag<T, 32> ktf[2];
    lds_frag64(kf, buf, ar, half);
    lds_frag64(vf, buf + STILE64, ar, half);
    lds_frag(ktf[0], buf + 2 * STILE64, ar, half);
    lds_frag(ktf[1], buf + 2 * STILE64 + 32 * SROW, ar, half);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, kf, qf);
    dp = mma(dp, vf, dof);
    float ds[16], add[16];
    if (p.keymask) lds_runs(add, km, kb * 32, half);
    float dm[16];
    if (p.drop_p > 0.f) drop_keys(dm, p, seq, h, qi, kb, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v2 = p.keymask ? fmaf(s[r], scale2, add[r]) : s[r] * scale2;
      const float pr = __builtin_amdgcn_exp2f(fminf(v2 - lse2, 0.f));   // (p <= 1; the clamp keeps a fully masked sequence -- |lse| ~ 3e38 -- finite)
      const float dpr = p.drop_p > 0.f ? dp[r] * dm[r] : dp[r];         // d(dropout(P)) / dP
      ds[r] = pr * (dpr - delta);
    }
    Frag<T, 32> dsf;
    frag_from_regs(dsf, ds);
    dqacc[0] = mma(dqacc[0], ktf[0], dsf);
    dqacc[1] = mma(dqacc[1], ktf[1], dsf);
    const int nb = ((kb & 1) ^ 1) * BUF;
    *reinterpret_cast<u32x4*>(kdst + nb) = sk;
    *reinterpret_cast<u32x4*>(vdst + nb) = sv;
    *reinterpret_cast<u32x4*>(tdst + nb) = st;
    __syncthreads();
  }
  if (active) {
    T* dQ = reinterpret_cast<T*>(p.dq) + ((int64_t)seq * L + qi) * p.lddq + h * D;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = dqacc[i][8 * g + e] * p.scale;
        store8(dQ + i * 32 + 16 * g + 8 * half, o8);
      }
  }
}

__global__ __launch_bounds__(256) void attn64_bwd_dkv_kernel(AttnParams p) {
  typedef bf16_t T;
  constexpr int D = 64, BUF = 2 * STILE64 + 2 * STILE64T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, seq = blockIdx.z;
  const int L = p.L, Lp = p.Lp, nqb = L / 32;
  __shared__ __attribute__((aligned(16))) char tiles[2 * BUF];          // [buffer][Q rows | dO rows | Q^T rows | dO^T rows]
  __shared__ __attribute__((aligned(16))) float stat[2][REL_MAXL];      // lse x log2 e, delta of every query of this (sequence, head)
  const int64_t statbase = ((int64_t)seq * p.H + h) * L;
  for (int i = tid; i < L; i += 256) { stat[0][i] = p.lse[statbase + i] * LOG2E; stat[1][i] = p.delta[statbase + i]; }
  const float scale2 = p.scale * LOG2E;
  const int jb_raw = blockIdx.x * 4 + wave;
  const bool active = jb_raw < nqb;
  const int jb = active ? jb_raw : nqb - 1;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int kj = jb * 32 + c;
  Frag<T, D> kf, vf;
  frag_load(kf, reinterpret_cast<const T*>(p.k) + ((int64_t)seq * L + kj) * p.ldk + h * D, half, D);
  frag_load(vf, reinterpret_cast<const T*>(p.v) + ((int64_t)seq * L + kj) * p.ldv + h * D, half, D);
  const float km2 = p.keymask ? fmaxf(p.keymask[(int64_t)seq * L + kj] * LOG2E, -3.0e38f) : 0.f;

  const T* qsrc = reinterpret_cast<const T*>(p.q) + ((int64_t)seq * L + (tid >> 3)) * p.ldq + h * D + (tid & 7) * 8;
  const T* osrc = reinterpret_cast<const T*>(p.dout) + ((int64_t)seq * L + (tid >> 3)) * p.lddo + h * D + (tid & 7) * 8;
  const int64_t toff = (((int64_t)seq * p.H + h) * D + (tid >> 2)) * Lp + (tid & 3) * 8;
  const T* qtsrc = reinterpret_cast<const T*>(p.qt) + toff;
  const T* otsrc = reinterpret_cast<const T*>(p.dot) + toff;
  char* qdst = tiles + (tid >> 3) * SROW64 + (tid & 7) * 16;
  char* odst = qdst + STILE64;
  char* qtdst = tiles + 2 * STILE64 + (tid >> 2) * SROW + (tid & 3) * 16;
  char* otdst = qtdst + STILE64T;
  const int64_t qstep = 32 * p.ldq, ostep = 32 * p.lddo;
  u32x4 sq = *reinterpret_cast<const u32x4*>(qsrc), so = *reinterpret_cast<const u32x4*>(osrc);
  u32x4 sqt = *reinterpret_cast<const u32x4*>(qtsrc), sot = *reinterpret_cast<const u32x4*>(otsrc);
  *reinterpret_cast<u32x4*>(qdst) = sq;
  *reinterpret_cast<u32x4*>(odst) = so;
  *reinterpret_cast<u32x4*>(qtdst) = sqt;
  *reinterpret_cast<u32x4*>(otdst) = sot;
  __syncthreads();

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkacc[i][r] = 0.f; dvacc[i][r] = 0.f; }
  for (int qb = 0; qb < nqb; ++qb) {
    const char* buf = tiles + (qb & 1) * BUF;
    const int qbn = qb + 1 < nqb ? qb + 1 : qb;
    sq = *reinterpret_cast<const u32x4*>(qsrc + qbn * qstep);
    so = *reinterpret_cast<const u32x4*>(osrc + qbn * ostep);
    sqt = *reinterpret_cast<const u32x4*>(qtsrc + qbn * 32);
    sot = *reinterpret_cast<const u32x4*>(otsrc + qbn * 32);
    Frag<T, D> qf, dof;
    lds_frag64(qf, buf, ar, half);
    lds_frag64(dof, buf + STILE64, ar, half);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, qf, kf);     // D[query rho][key c]
    dp = mma(dp, dof, vf);
    float lse16[16], del16[16], pr[16], ds[16], dmq[16];
    lds_runs(lse16, stat[0], qb * 32, half);
    lds_runs(del16, stat[1], qb * 32, half);
    if (p.drop_p > 0.f) drop_queries(dmq, p, seq, h, kj, qb, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pr[r] = __builtin_amdgcn_exp2f(fminf(fmaf(s[r], scale2, km2) - lse16[r], 0.f));   // (clamp: see attn64_bwd_dq_kernel)
      const float dm = p.drop_p > 0.f ? dmq[r] : 1.f;
      ds[r] = pr[r] * (dp[r] * dm - del16[r]);
      pr[r] *= dm;                                           // dV sees the dropped probabilities
    }
    Frag<T, 32> pf, dsf, tf[2];
    frag_from_regs(pf, pr);
    frag_from_regs(dsf, ds);
    lds_frag(tf[0], buf + 2 * STILE64 + STILE64T, ar, half);
    lds_frag(tf[1], buf + 2 * STILE64 + STILE64T + 32 * SROW, ar, half);
    dvacc[0] = mma(dvacc[0], tf[0], pf);
    dvacc[1] = mma(dvacc[1], tf[1], pf);
    lds_frag(tf[0], buf + 2 * STILE64, ar, half);
    lds_frag(tf[1], buf + 2 * STILE64 + 32 * SROW, ar, half);
    dkacc[0] = mma(dkacc[0], tf[0], dsf);
    dkacc[1] = mma(dkacc[1], tf[1], dsf);
    const int nb = ((qb & 1) ^ 1) * BUF;
    *reinterpret_cast<u32x4*>(qdst + nb) = sq;
    *reinterpret_cast<u32x4*>(odst + nb) = so;
    *reinterpret_cast<u32x4*>(qtdst + nb) = sqt;
    *reinterpret_cast<u32x4*>(otdst + nb) = sot;
    __syncthreads();
  }
  if (active) {
    T* dK = reinterpret_cast<T*>(p.dk) + ((int64_t)seq * L + kj) * p.lddk + h * D;
    T* dV = reinterpret_cast<T*>(p.dv) + ((int64_t)seq * L + kj) * p.lddv + h * D;
#pragma unroll
    for (int i = 0; i < 2; ++i)
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


// dBias for the CTViT shape, workgroup-shared: a workgroup of EIGHT waves owns two neighbouring query blocks and four neighbouring
// key blocks of one head (one (query, key) tile pair per wave) and walks a strided subset of the sequences.  Per sequence the twelve
// operand tiles (Q, dO of the two query blocks; K, V of the four key blocks) are staged once through LDS with three coalesced
// 16-byte loads per thread -- 1.5 tiles of traffic per tile pair (the 1 x 4 arrangement of the first LDS version moved 2.5 and sat
// at the L2 bandwidth; the register-only attn_bwd_dbias_kernel issues eight scattered loads per lane and sequence and waits for
// them more than half of its time, SQ_WAIT_INST_ANY 54 % of SQ_WAVE_CYCLES).  Same fast-path conventions as above (log2 domain,
// full tiles only).
__global__ __launch_bounds__(512) void attn_bwd_dbias_lds_kernel(AttnParams p, float* __restrict__ dbias_part, int nsplit) {
  typedef bf16_t T;
  constexpr int D = 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.L, nkb = L / 32, ngrp = (nkb + 3) / 4;
  const int qg = blockIdx.x / ngrp, kg = blockIdx.x % ngrp, h = blockIdx.y, split = blockIdx.z;
  const int qsel = wave >> 2, ksel = wave & 3;
  const int qb_raw = qg * 2 + qsel, kb_raw = kg * 4 + ksel;
  const bool active = qb_raw < nkb && kb_raw < nkb;
  const int qb = qb_raw < nkb ? qb_raw : nkb - 1, kb = kb_raw < nkb ? kb_raw : nkb - 1;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int qi = qb * 32 + c;
  __shared__ __attribute__((aligned(16))) char tiles[2][12][STILE];     // [buffer][Q x2, dO x2, K x4, V x4]
  __shared__ float stats[2][2][2][32];                                   // [buffer][query block][lse * log2 e, delta][query]
  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  const T* dO = reinterpret_cast<const T*>(p.dout);

  float add[16], acc[16];                                                // loop-invariant additive logits of this wave's tile, log2 domain
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kj = kb * 32 + slot_index(r, half);
    float a = 0.f;
    if (p.bias_tab) {
      const int cls = (qi / p.gw - kj / p.gw + p.gh - 1) * (2 * p.gw - 1) + (qi % p.gw - kj % p.gw + p.gw - 1);
      a = p.bias_tab[(int64_t)cls * p.H + h];
    } else if (p.bias) a = p.bias[((int64_t)h * L + qi) * L + kj];
    add[r] = a * LOG2E;
    acc[r] = 0.f;
  }
  const float scale2 = p.scale * LOG2E;

  // staging: pass i (0..2) fills tiles 4i .. 4i+3 (128 threads each): pass 0 = Q0 Q1 dO0 dO1, pass 1 = K0..K3, pass 2 = V0..V3
  const int stile = threadIdx.x >> 7, srow = (threadIdx.x & 127) >> 2, schunk = threadIdx.x & 3;
  const T* sbase[3] = {stile < 2 ? Q : dO, K, V};
  const int64_t sld[3] = {stile < 2 ? p.ldq : p.lddo, p.ldk, p.ldv};
  const int stok[3] = {(qg * 2 + (stile & 1)) * 32, (kg * 4 + stile) * 32, (kg * 4 + stile) * 32};
  auto gstat = [&](int seq) {    // threads 0-127: [query block][lse | delta][32 queries]
    int q = (qg * 2 + ((threadIdx.x >> 6) & 1)) * 32 + (threadIdx.x & 31);
    q = q < L ? q : L - 1;
    return ((threadIdx.x & 32) ? p.delta : p.lse) + ((int64_t)seq * p.H + h) * L + q;
  };
  const float smul = (threadIdx.x & 32) ? 1.f : LOG2E;
  char* sdst = &tiles[0][stile][0] + srow * SROW + schunk * 16;          // + 4 * i * STILE per pass, + 12 * STILE per buffer
  float* sstat = &stats[0][0][0][0] + (threadIdx.x & 127);

  int seq = split;
  if (seq >= p.nseq) return;                                             // workgroup-uniform
  u32x4 st[3];
  float sv;
#pragma unroll
  for (int i = 0; i < 3; ++i) st[i] = *reinterpret_cast<const u32x4*>(src_rows(sbase[i], (int64_t)seq * L, stok[i], L, sld[i], h * D, srow, schunk));
  sv = *gstat(seq) * smul;
#pragma unroll
  for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(sdst + 4 * i * STILE) = st[i];
  if (threadIdx.x < 128) *sstat = sv;
  __syncthreads();
  for (int it = 0; seq < p.nseq; seq += nsplit, ++it) {
    const int buf = it & 1;
    const int sn = seq + nsplit < p.nseq ? seq + nsplit : seq;
#pragma unroll
    for (int i = 0; i < 3; ++i) st[i] = *reinterpret_cast<const u32x4*>(src_rows(sbase[i], (int64_t)sn * L, stok[i], L, sld[i], h * D, srow, schunk));
    sv = *gstat(sn) * smul;
    Frag<T, D> qf, dof, kf, vf;
    lds_frag(qf, tiles[buf][qsel], c, half);          // B operands: the lane's own query row (no pi32)
    lds_frag(dof, tiles[buf][2 + qsel], c, half);
    lds_frag(kf, tiles[buf][4 + ksel], ar, half);
    lds_frag(vf, tiles[buf][8 + ksel], ar, half);
    const float lse2 = stats[buf][qsel][0][c], delta = stats[buf][qsel][1][c];
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    s = mma(s, kf, qf);
    dp = mma(dp, vf, dof);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(__builtin_amdgcn_exp2f(fmaf(s[r], scale2, add[r]) - lse2), dp[r] - delta, acc[r]);
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(sdst + (buf ^ 1) * 12 * STILE + 4 * i * STILE) = st[i];
    if (threadIdx.x < 128) sstat[(buf ^ 1) * 128] = sv;
    __syncthreads();
  }
  if (active) {
    float* dst = dbias_part + (((int64_t)split * p.H + h) * L + qi) * L;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[kb * 32 + slot_index(r, half)] = acc[r];
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
                                                          T* __restrict__ dx, float* __restrict__ dscale_part, int64_t M, int H,
                                                          int64_t lddy, int64_t ldx, int64_t lddx) {
  constexpr int G = D / 8;
  __shared__ float red[4][D];
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
  if (!dscale_part) return;
  // deterministic: lanes that share a column group (equal lane mod G) by a fixed xor tree, the four waves in order, one partial
  // row per workgroup (summed in block order by qk_scale_sum_kernel)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = G; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < G) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][d0 + e] = acc[e];          // lanes 0 .. G-1 hold the G distinct column groups
  }
  __syncthreads();
  if (threadIdx.x < D) dscale_part[(int64_t)blockIdx.x * D + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <int D>
__global__ void qk_scale_sum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dscale) {
  __shared__ float red[16][D];
  const int d = threadIdx.x % D, sl = threadIdx.x / D;      // blockDim = 16 * D
  float t = 0.f;
  for (int b = sl; b < nblk; b += 16) t += part[(int64_t)b * D + d];
  red[sl][d] = t;
  __syncthreads();
  if (sl == 0) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += red[i][d];
    dscale[d] += a;
  }
}

template <typename T, int D>
int launch_attn(int which, const AttnParams& p, hipStream_t stream) {
  dim3 grid((unsigned)cdiv(cdiv(p.L, 32), 4), p.H, p.L <= 32 ? (unsigned)cdiv(p.nseq, 4) : p.nseq), block(256);
  if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<T, D>), grid, block, 0, stream, p);
  else if (which == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D>), grid, block, 0, stream, p);
  return ctclip_check_launch("attention");
}

bool attn_lds_enabled() {
  static int use_lds = -1;
  if (use_lds < 0) { const char* e = getenv("CTCLIP_ATTN_LDS"); use_lds = (e && e[0] == '0') ? 0 : 1; }
  return use_lds != 0;
}

int dispatch_attn(int which, const AttnParams& p, int D, int dtype, hipStream_t stream) {
  const bool use_lds = attn_lds_enabled();
  if (use_lds && dtype == DT_BF16 && D == 32 && p.L >= 128 && !p.dbias && p.drop_p == 0.f) {   // CTViT spatial shape: workgroup-shared operand tiles
    dim3 grid((unsigned)cdiv(cdiv(p.L, 32), 4), p.H, p.nseq), block(256);
    const bool fast = (p.L % 32) == 0 && !p.keymask && !p.bias && (!p.bias_tab || (p.gw % 8) == 0);
    if (fast) {
      if (which == 0) hipLaunchKernelGGL(attn_fwd_lds_kernel<true>, grid, block, 0, stream, p);
      else if (which == 1) hipLaunchKernelGGL(attn_bwd_dq_lds_kernel<true>, grid, block, 0, stream, p);
      else hipLaunchKernelGGL(attn_bwd_dkv_lds_kernel<true>, grid, block, 0, stream, p);
    } else {
      if (which == 0) hipLaunchKernelGGL(attn_fwd_lds_kernel<false>, grid, block, 0, stream, p);
      else if (which == 1) hipLaunchKernelGGL(attn_bwd_dq_lds_kernel<false>, grid, block, 0, stream, p);
      else hipLaunchKernelGGL(attn_bwd_dkv_lds_kernel<false>, grid, block, 0, stream, p);
    }
    return ctclip_check_launch("attention (lds)");
  }
  if (use_lds && dtype == DT_BF16 && D == 64 && p.L % 32 == 0 && p.L >= 64 && p.L <= REL_MAXL && p.Lp % 8 == 0 && !p.bias && !p.bias_tab && !p.dbias) {
    dim3 grid((unsigned)cdiv(p.L / 32, 4), p.H, p.nseq), block(256);   // BERT shape: workgroup-shared operand tiles, key mask, dropout
    if (which == 0) hipLaunchKernelGGL(attn64_fwd_kernel, grid, block, 0, stream, p);
    else if (which == 1) hipLaunchKernelGGL(attn64_bwd_dq_kernel, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(attn64_bwd_dkv_kernel, grid, block, 0, stream, p);
    return ctclip_check_launch("attention (lds, d_head 64)");
  }
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

static int64_t qk_norm_bwd_blocks(int64_t M, int H, int D) {
  const int HG = H * (D / 8);
  const int64_t want_rows = M < 4096 ? M : 4096;
  int64_t nblocks = cdiv(want_rows * HG, 256);
  while ((nblocks * 256) % HG) ++nblocks;      // the grid stride must be a multiple of HG so that each thread keeps its column group
  return nblocks;
}
extern "C" int64_t ctclip_qk_norm_bwd_workspace(int64_t M, int H, int D) { return qk_norm_bwd_blocks(M, H, D) * D * 4; }

// dscale (D) is ACCUMULATED (two stages, fixed summation order; workspace >= ctclip_qk_norm_bwd_workspace when dscale is given).
extern "C" int ctclip_qk_norm_bwd(const void* dy, const void* x, const float* inv, const float* scale_vec, void* dx, float* dscale,
                                  int64_t M, int H, int D, int64_t lddy, int64_t ldx, int64_t lddx, int dtype, void* workspace,
                                  int64_t workspace_bytes, hipStream_t stream) {
  if (!dy || !x || !dx || !inv || !scale_vec || bad_ld(lddy) || bad_ld(ldx) || bad_ld(lddx)) { ctclip_set_error("qk_norm_bwd: bad args"); return CTCLIP_EBADARG; }
  if (256 % (D / 8) != 0) return CTCLIP_EUNSUPPORTED;
  if (dscale && (!workspace || workspace_bytes < ctclip_qk_norm_bwd_workspace(M, H, D))) { ctclip_set_error("qk_norm_bwd: workspace too small"); return CTCLIP_EWORKSPACE; }
  const int64_t nblocks = qk_norm_bwd_blocks(M, H, D);
  dim3 grid((unsigned)nblocks);
  float* part = dscale ? (float*)workspace : nullptr;
  if (dtype == DT_BF16 && D == 32) hipLaunchKernelGGL((qk_norm_bwd_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, inv, scale_vec, (bf16_t*)dx, part, M, H, lddy, ldx, lddx);
  else if (dtype == DT_BF16 && D == 64) hipLaunchKernelGGL((qk_norm_bwd_kernel<bf16_t, 64>), grid, dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, inv, scale_vec, (bf16_t*)dx, part, M, H, lddy, ldx, lddx);
  else if (dtype == DT_F32 && D == 32) hipLaunchKernelGGL((qk_norm_bwd_kernel<float, 32>), grid, dim3(256), 0, stream, (const float*)dy, (const float*)x, inv, scale_vec, (float*)dx, part, M, H, lddy, ldx, lddx);
  else if (dtype == DT_F32 && D == 64) hipLaunchKernelGGL((qk_norm_bwd_kernel<float, 64>), grid, dim3(256), 0, stream, (const float*)dy, (const float*)x, inv, scale_vec, (float*)dx, part, M, H, lddy, ldx, lddx);
  else { ctclip_set_error("qk_norm: head dim must be 32 or 64"); return CTCLIP_EUNSUPPORTED; }
  int rc = ctclip_check_launch("qk_norm_bwd");
  if (rc || !dscale) return rc;
  if (D == 32) hipLaunchKernelGGL(qk_scale_sum_kernel<32>, dim3(1), dim3(16 * 32), 0, stream, (const float*)part, (int)nblocks, dscale);
  else hipLaunchKernelGGL(qk_scale_sum_kernel<64>, dim3(1), dim3(16 * 64), 0, stream, (const float*)part, (int)nblocks, dscale);
  return ctclip_check_launch("qk_scale_sum");
}

// softmax(scale * q k^T + bias[h] + keymask[seq]) v   (attention.py:156-178 / HF BertSelfAttention).
// q,k: (nseq*L, >= H*D) row-major views with row strides ldq/ldk; vt: ctclip_head_transpose of v.
// out (nseq*L, ldo); lse (nseq,H,L) f32 (may be null for inference).
// bias: (H, L, L) f32 additive logits, or -- when bias_gh > 0 -- the relative-position table (nclass, H) of a bias_gh x bias_gw token
// grid (L = bias_gh * bias_gw; see RelLds), which the kernels gather from LDS instead of streaming the expanded matrix.
static int set_bias(AttnParams& p, const float* bias, int bias_gh, int bias_gw, int L) {
  if (bias_gh <= 0) { p.bias = bias; return 0; }
  if (!bias || bias_gw <= 0 || bias_gh * bias_gw != L || L > REL_MAXL || (2 * bias_gh - 1) * (2 * bias_gw - 1) > REL_MAXCLS) {
    ctclip_set_error("attention: relative bias table needs L = gh * gw <= 1024 and (2gh-1)(2gw-1) <= 4096");
    return CTCLIP_EBADARG;
  }
  p.bias = nullptr; p.bias_tab = bias; p.gh = bias_gh; p.gw = bias_gw;
  return 0;
}

// dropout_p > 0: HF attention-probability dropout (train mode), mask = philox(dropout_seed, linear score index); the same
// (dropout_p, dropout_seed) must be passed to ctclip_attn_bwd.
static int set_dropout(AttnParams& p, float dropout_p, uint64_t dropout_seed) {
  if (dropout_p < 0.f || dropout_p >= 1.f) { ctclip_set_error("attention: 0 <= dropout_p < 1"); return CTCLIP_EBADARG; }
  p.drop_p = dropout_p; p.drop_inv_keep = 1.f / (1.f - dropout_p); p.drop_seed = dropout_seed; p.drop_state = ctclip_step_state();
  return 0;
}

extern "C" int ctclip_attn_fwd(const void* q, const void* k, const void* vt, const float* bias, int bias_gh, int bias_gw,
                               const float* keymask, void* out, float* lse, int nseq, int H, int L, int Lp, int D, int64_t ldq,
                               int64_t ldk, int64_t ldo, float scale, float dropout_p, uint64_t dropout_seed, int dtype,
                               hipStream_t stream) {
  if (!q || !k || !vt || !out || bad_ld(ldq) || bad_ld(ldk) || bad_ld(ldo) || Lp % 8 || Lp < L) { ctclip_set_error("attn_fwd: bad args"); return CTCLIP_EBADARG; }
  AttnParams p{};
  if (int rc = set_bias(p, bias, bias_gh, bias_gw, L)) return rc;
  if (int rc = set_dropout(p, dropout_p, dropout_seed)) return rc;
  p.q = q; p.k = k; p.vt = vt; p.keymask = keymask; p.out = out; p.lse_out = lse;
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
// (+ two slabs of scratch for the class bins of the deterministic fold in the relative-position mode)
extern "C" int64_t ctclip_attn_bwd_workspace(int nseq, int H, int L) { return (int64_t)(dbias_nsplit(nseq, H, L) + 2) * H * L * L * 4; }
extern "C" int ctclip_cpb_reduce(const float* dbias, float* dtab, int H, int gh, int gw, hipStream_t s);
int ctclip_dbias_fold(const float* part, int nsplit, float* bins_ws, float* dtab, int H, int gh, int gw, hipStream_t stream);   // attn2.hip

// Backward.  Needs the transposed copies qt, kt (of q, k) and dot (of dout) and delta = rowsum(dO*O) (computed here
// into `delta`, (nseq,H,L) f32 scratch).  dbias (H,L,L) f32 is ACCUMULATED when non-null; in the relative-position mode
// (bias_gh > 0) dbias is the table gradient (nclass, H) and is OVERWRITTEN.
extern "C" int ctclip_attn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* o,
                               const void* dout, const void* dot, const float* lse, const float* bias, int bias_gh, int bias_gw,
                               const float* keymask, float* delta, void* dq, void* dk, void* dv, float* dbias, int nseq, int H, int L, int Lp, int D,
                               int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk,
                               int64_t lddv, float scale, float dropout_p, uint64_t dropout_seed, int dtype, void* workspace,
                               int64_t workspace_bytes, hipStream_t stream) {
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
  if (int rc0 = set_bias(p, bias, bias_gh, bias_gw, L)) return rc0;
  if (int rc0 = set_dropout(p, dropout_p, dropout_seed)) return rc0;
  if (dropout_p > 0.f && dbias) { ctclip_set_error("attn_bwd: dropout together with a bias gradient is not implemented"); return CTCLIP_EUNSUPPORTED; }
  p.q = q; p.k = k; p.v = v; p.qt = qt; p.kt = kt; p.o = o; p.dout = dout; p.dot = dot; p.lse = lse; p.delta = delta;
  p.keymask = keymask; p.dq = dq; p.dk = dk; p.dv = dv; p.dbias = dbias;
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
    if (dtype == DT_BF16 && D == 32 && L >= 128 && (L % 32) == 0 && !keymask && attn_lds_enabled()) {
      dim3 grid2((unsigned)(((nkb + 1) / 2) * ((nkb + 3) / 4)), H, ns);
      hipLaunchKernelGGL(attn_bwd_dbias_lds_kernel, grid2, dim3(512), 0, stream, p, ws, ns);
    } else if (dtype == DT_BF16 && D == 32) hipLaunchKernelGGL((attn_bwd_dbias_kernel<bf16_t, 32>), grid, dim3(256), 0, stream, p, ws, ns);
    else if (dtype == DT_BF16 && D == 64) hipLaunchKernelGGL((attn_bwd_dbias_kernel<bf16_t, 64>), grid, dim3(256), 0, stream, p, ws, ns);
    else if (dtype == DT_F32 && D == 32) hipLaunchKernelGGL((attn_bwd_dbias_kernel<float, 32>), grid, dim3(256), 0, stream, p, ws, ns);
    else if (dtype == DT_F32 && D == 64) hipLaunchKernelGGL((attn_bwd_dbias_kernel<float, 64>), grid, dim3(256), 0, stream, p, ws, ns);
    else return CTCLIP_EUNSUPPORTED;
    rc = ctclip_check_launch("attn_bwd_dbias");
    if (rc) return rc;
    const int64_t n = (int64_t)H * L * L;
    int64_t nb = cdiv(n, 256); if (nb > 4096) nb = 4096;
    if (p.bias_tab) {   // table mode: slabs -> (nclass, H) in one pass (overwrites dbias)
      // deterministic class bins (attn2.hip dbias_bin / dbias_sum); the bins live in the spare slab behind the ns partial slabs
      rc = ctclip_dbias_fold((const float*)workspace, ns, (float*)workspace + (int64_t)ns * n, dbias, H, p.gh, p.gw, stream);
    } else {
      hipLaunchKernelGGL(dbias_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)workspace, dbias, ns, n, 1);
      rc = ctclip_check_launch("dbias_reduce");
    }
    if (rc) return rc;
    p.dbias = nullptr;
  }
  rc = dispatch_attn(1, p, D, dtype, stream);
  if (rc) return rc;
  return dispatch_attn(2, p, D, dtype, stream);
}
