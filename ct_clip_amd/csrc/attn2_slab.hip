// Slab-resident, persistent variants of the three main attention kernels (forward, dQ pass, dK/dV pass) for sequences whose two
// operand slabs (L x 64 B each) fit in LDS twice: L <= 576, the CTViT spatial shape (24 x 24 tokens).
//
// Why (measured, profiles/r02_attn_*.md): the ring kernels of attn2.hip synchronise the workgroup every 32 keys; their waves
// spend 43-63 % of their cycles parked in s_waitcnt / s_barrier (SQ_WAIT_ANY) with the vector ALU 22-47 % busy.  Here one workgroup
// of eight waves per CU walks a contiguous run of (head, sequence) items; both operand slabs of the current item sit in LDS, the tile
// loop has NO barrier, and while item k is computed every thread copies its share of item k+1's slabs into the other LDS buffer
// (three 16-byte pieces per thread every fourth step, written three steps later so the loads have landed).  One barrier per item.
// Eight waves = two per SIMD with 256 registers each (nine waves -- two blocks per wave -- would cap the budget at 168: spills).
//
//   forward : a wave owns query blocks w, w + 8 (two independent chains that share the K^ / V fragment reads) and, waves 0-1, w + 16
//   dQ pass : a wave owns query blocks w, w + 8 (, w + 16), one after the other, rows requested one block ahead
//   dK/dV   : a wave owns key blocks w, w + 8 (, w + 16), one after the other; slabs = Q~ and dO', -delta' of the item in LDS
//
// Bias table in LDS with row stride S = 56 instead of 2 gw - 1 = 47 (S = gw mod 32): the 32 queries of a block then gather from 32
// distinct banks (the stride-47 table of the ring kernels loses half of its LDS cycles to 2-way conflicts: SQ_LDS_BANK_CONFLICT =
// 15.9 M of 38 M LDS cycles).  Token -> table offset is arithmetic (no lookup table: the LDS budget is 160 KiB to the byte).
//
// The exp2 of the softmax costs 16 cycles per wave instruction on this chip (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU on the forward:
// 65 vector instructions per 32 x 32 tile of which 16 exp2 account for 3/4 of the busy cycles): 510 M scores per layer at batch 8
// are 56 us of transcendental issue per pass whatever else happens -- the ceiling of this operator at d_head = 32 is the
// transcendental unit, not the matrix cores (which need 26 us).
#include "attn2_common.h"

// Compile-time ablation mask for the slab-resident forward (tools/build_ablation.py attn2_slab.hip:ATTN2_ABL ...; never set in the
// product build): 1 = no exp2, 2 = no row sums, 4 = no P V MFMAs, 8 = no K Q^T MFMAs, 16 = no bias gather, 32 = no copy of the next item.
#ifndef ATTN2_ABL
#define ATTN2_ABL 0
#endif
#ifndef ATTN2_FWD_TRB
#define ATTN2_FWD_TRB 1      // 1: the forward reads V^T through the builtin transposing read (compiler-counted waits); 0: the inline-asm form + lgkmcnt(0)
#endif

namespace {

constexpr int SW = 8;                        // waves per workgroup: two per SIMD, 256 registers each
constexpr int NTH = SW * 64;
constexpr int STAB_MAX = 2688;               // (2 gh - 1) * S entries: 47 * 56 = 2632 for 24 x 24 tokens
constexpr int SL_MAX = 576;

struct SRel {
  float tab[STAB_MAX];
  float red[2][16];
  float m2; int safe;
};
constexpr int SREL_BYTES = (int)((sizeof(SRel) + 255) / 256 * 256);

struct Geo {                                 // token -> offset arithmetic of the bias table
  int gw, S, c0, n, magic;                   // magic = ceil(2^16 / gw): t / gw = (t * magic) >> 16 for t * gw < 2^16
  __device__ __forceinline__ int u(int t) const { const int r = (t * magic) >> 16; return r * S + (t - r * gw); }
};
__host__ __device__ inline int table_stride(int gw) { int S = 2 * gw - 1; while ((S & 31) != (gw & 31)) ++S; return S; }

// stage the table of head h in the orientation the kernel gathers in (REVERSED: entry n - 1 - i holds offset class i) and derive the logit bound
template <bool REVERSED>
__device__ __forceinline__ void stage_srel(SRel& rel, const Params& p, const Geo& g, int h) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY, mn = INFINITY;
  if (p.tab) {
    const int ncls = (2 * p.gh - 1) * (2 * p.gw - 1), W = 2 * p.gw - 1;
    for (int i = tid; i < g.n; i += NTH) rel.tab[i] = 0.f;                  // the gaps of the padded rows are never gathered
    __syncthreads();
    for (int i = tid; i < ncls; i += NTH) {
      const float t = p.tab[(int64_t)i * p.H + h] * LOG2E;
      const int j = (i / W) * g.S + i % W;
      rel.tab[REVERSED ? g.n - 1 - j : j] = t;
      mx = fmaxf(mx, t); mn = fminf(mn, t);
    }
  } else {
    mx = 0.f; mn = 0.f;
    if (tid == 0) rel.tab[0] = 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
  if (lane == 0) { rel.red[0][wave] = mx; rel.red[1][wave] = mn; }
  __syncthreads();
  if (wave == 0) {
    float a = lane < 32 ? fabsf(p.q_scale[lane]) : fabsf(p.k_scale[lane - 32]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
    const float qk = a * __shfl_xor(a, 32, 64) * p.c;
    float tmx = -INFINITY, tmn = INFINITY;
    for (int w = 0; w < SW; ++w) { tmx = fmaxf(tmx, rel.red[0][w]); tmn = fminf(tmn, rel.red[1][w]); }
    if (lane == 0) {
      const float span = 2.f * qk + (tmx - tmn);
      rel.safe = (span <= SAFE_SPAN && span == span) ? 1 : 0;
      rel.m2 = qk + tmx;
    }
  }
  __syncthreads();
  if (rel.safe) {
    const float m2 = rel.m2;
    for (int i = tid; i < (p.tab ? g.n : 1); i += NTH) rel.tab[i] -= m2;
  }
  __syncthreads();
}

// accumulator input of one 32 x 32 tile: rows (registers) = keys (reversed table, ascending addresses) or queries (natural table)
template <bool ROWS_ARE_KEYS, bool TAB>
__device__ __forceinline__ f32x16 sbias(const SRel& rel, const Geo& g, int ucol, int row_base, int half) {
  f32x16 cb;
  if (!TAB) {
    const float t = rel.tab[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) cb[r] = t;
    return cb;
  }
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    const int urow0 = g.u(row_base + 16 * gq + 8 * half);
    const float* b = rel.tab + (ROWS_ARE_KEYS ? g.n - 1 - (ucol - urow0 + g.c0) : urow0 - ucol + g.c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) cb[8 * gq + e] = b[e];
  }
  return cb;
}

// ---- copy of the next item's slabs, spread over the tile loop: every fourth step CP_BATCH 16-byte pieces per thread are requested,
// three steps later they are written (by then they have landed: the compiler's vmcnt(0) in front of the write costs nothing).
constexpr int CP_BATCH = 3;
struct Copier {
  const char* g0; const char* g1;            // global bases of the two slabs of the next item (null: nothing to copy)
  char* dst; int sb;                         // LDS base of the other buffer, bytes per slab
  int nchunk, per;                           // pieces in total (2 slabs), pieces per slab
  u32x4 st[CP_BATCH];
  __device__ __forceinline__ void issue(int t) {
    if (!g0 || (ATTN2_ABL & 32)) return;
#pragma unroll
    for (int k = 0; k < CP_BATCH; ++k) {
      const int chunk = ((t >> 2) * CP_BATCH + k) * NTH + (int)threadIdx.x;
      if (chunk < nchunk) {
        const int tens = chunk >= per ? 1 : 0, rc = chunk - tens * per;
        st[k] = *reinterpret_cast<const u32x4*>((tens ? g1 : g0) + (int64_t)rc * 16);
      }
    }
  }
  __device__ __forceinline__ void commit(int t) {
    if (!g0 || (ATTN2_ABL & 32)) return;
#pragma unroll
    for (int k = 0; k < CP_BATCH; ++k) {
      const int chunk = ((t >> 2) * CP_BATCH + k) * NTH + (int)threadIdx.x;
      if (chunk < nchunk) {
        const int tens = chunk >= per ? 1 : 0, rc = chunk - tens * per, row = rc >> 2, pc = rc & 3;
        *reinterpret_cast<u32x4*>(dst + tens * sb + (row >> 5) * TILE + swz(row & 31, pc)) = st[k];
      }
    }
  }
  // call once per step
  __device__ __forceinline__ void step(int t, int nsteps) {
    if ((t & 3) == 0) issue(t);
    if ((t & 3) == 3 || t == nsteps - 1) commit(t);
  }
};
// synchronous copy of the first item (all requests first, then all writes)
__device__ __forceinline__ void copy_now(const char* g0, const char* g1, char* dst, int sb, int L) {
  constexpr int MAXPER = (2 * SL_MAX * 4 + NTH - 1) / NTH;
  const int per = L * 4, nchunk = 2 * per;
  u32x4 v[MAXPER];
#pragma unroll
  for (int k = 0; k < MAXPER; ++k) {
    const int chunk = k * NTH + (int)threadIdx.x;
    const int tens = chunk >= per ? 1 : 0, rc = chunk - tens * per;
    v[k] = u32x4{0, 0, 0, 0};
    if (chunk < nchunk) v[k] = *reinterpret_cast<const u32x4*>((tens ? g1 : g0) + (int64_t)rc * 16);
  }
#pragma unroll
  for (int k = 0; k < MAXPER; ++k) {
    const int chunk = k * NTH + (int)threadIdx.x;
    const int tens = chunk >= per ? 1 : 0, rc = chunk - tens * per, row = rc >> 2, pc = rc & 3;
    if (chunk < nchunk) *reinterpret_cast<u32x4*>(dst + tens * sb + (row >> 5) * TILE + swz(row & 31, pc)) = v[k];
  }
}

struct Item { int seq, h; };
struct Run {                                 // the contiguous run of (head, sequence) items of this workgroup; head = slow index
  int first, last, nseq;
  __device__ __forceinline__ Item at(int i) const { return Item{i % nseq, i / nseq}; }
};
__device__ __forceinline__ int64_t slab_off(const Params& p, const Item& it) { return ((int64_t)it.h * p.M + (int64_t)it.seq * p.L) * D; }

// ================================================================================================================== forward
// One work unit = NCH query blocks of the current item as independent chains sharing the K^ / V fragment reads.
template <bool SAFE, bool TAB, int NCH>
__device__ __forceinline__ void fwd_unit(const Params& p, const SRel& rel, const Geo& g, const char* kslab, const char* vslab, Copier& cp, bool do_copy, const Item& it,
                                         int qb0, int lane, const Frag (&qf)[2]) {
  const int L = p.L, nkb = L / 32;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const TrOff tr = tr_offsets(lane);
  int qi[NCH], ucol[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) { qi[ch] = (qb0 + ch * SW) * 32 + c; ucol[ch] = g.u(qi[ch]); }
  float ls[NCH][4], m[NCH];
  f32x16 oacc[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    m[ch] = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) ls[ch][e] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[ch][r] = 0.f;
  }
#ifndef ATTN2_FWD_PIPE
#define ATTN2_FWD_PIPE 0     // 1: K^ / V^T fragments and the bias tile of key tile t + 1 are requested behind the Q K^T products of tile t (round 6: measured 145-149 against 143.6-145 us, 256 registers + 12 B of scratch -- off)
#endif
  Frag kf_n, vf_n;
  f32x16 s_n[NCH];
  if (ATTN2_FWD_PIPE) {
    kf_n = lds_rows(kslab, ar, half);
    vf_n = lds_cols_b(vslab, tr);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) s_n[ch] = sbias<true, TAB && !(ATTN2_ABL & 16)>(rel, g, ucol[ch], 0, half);
  }
  for (int t = 0; t < nkb; ++t) {
    const char* ktile = kslab + t * TILE;
    const char* vtile = vslab + t * TILE;
    const Frag kf = ATTN2_FWD_PIPE ? kf_n : lds_rows(ktile, ar, half);
    f32x16 s[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) s[ch] = ATTN2_FWD_PIPE ? s_n[ch] : sbias<true, TAB && !(ATTN2_ABL & 16)>(rel, g, ucol[ch], t * 32, half);
    if (!(ATTN2_ABL & 8)) {     // the chains' dependent MFMA pairs interleaved: a0 b0 a1 b1
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) s[ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v[0], qf[ch].v[0], s[ch], 0, 0, 0);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) s[ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v[1], qf[ch].v[1], s[ch], 0, 0, 0);
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) s[ch][0] += __builtin_bit_cast(float, (uint32_t)kf.v[0][0] << 16) * 1e-30f;
    }
    const Frag vf = ATTN2_FWD_PIPE ? vf_n : (ATTN2_FWD_TRB ? lds_cols_b(vtile, tr) : lds_cols(vtile, tr));
    if (ATTN2_FWD_PIPE) {                                        // the next key tile's operands (the last tile re-reads itself: no branch)
      const int tn = t + 1 < nkb ? t + 1 : t;
      kf_n = lds_rows(kslab + tn * TILE, ar, half);
      vf_n = lds_cols_b(vslab + tn * TILE, tr);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) s_n[ch] = sbias<true, TAB && !(ATTN2_ABL & 16)>(rel, g, ucol[ch], tn * 32, half);
      __builtin_amdgcn_sched_barrier(0);
    }
    Frag pf[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      float pr[16];
      if (SAFE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = (ATTN2_ABL & 1) ? s[ch][r] : __builtin_amdgcn_exp2f(s[ch][r]);
        if (!(ATTN2_ABL & 2)) {
#pragma unroll
          for (int r = 0; r < 16; r += 4) { ls[ch][0] += pr[r]; ls[ch][1] += pr[r + 1]; ls[ch][2] += pr[r + 2]; ls[ch][3] += pr[r + 3]; }
        } else ls[ch][0] += pr[0];
      } else {
        float mx = s[ch][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[ch][r]);
        mx = half_max(mx);
        const float mnew = fmaxf(m[ch], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[ch] - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(s[ch][r] - mnew); ps += pr[r]; }
        ls[ch][0] = ls[ch][0] * alpha + ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[ch][r] *= alpha;
        m[ch] = mnew;
      }
      pf[ch] = pack(pr);
    }
    if (!(ATTN2_ABL & 4)) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) oacc[ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v[0], pf[ch].v[0], oacc[ch], 0, 0, 0);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) oacc[ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v[1], pf[ch].v[1], oacc[ch], 0, 0, 0);
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) oacc[ch][0] += __builtin_bit_cast(float, (uint32_t)(vf.v[0][0] ^ vf.v[1][0] ^ pf[ch].v[0][0] ^ pf[ch].v[1][1]) << 16) * 1e-30f;
    }
    if (do_copy) cp.step(t, nkb);
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const float lsum = (ls[ch][0] + ls[ch][1]) + (ls[ch][2] + ls[ch][3]);
    const float l = half_sum(lsum);
    const float inv = 1.f / l;
    bf16_t* O = p.out + ((int64_t)it.seq * L + qi[ch]) * p.ldo + it.h * D;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = oacc[ch][8 * gq + e] * inv;
      store8(O + 16 * gq + 8 * half, o8);
    }
    if (half == 0 && p.lse2) p.lse2[(int64_t)it.h * p.M + (int64_t)it.seq * L + qi[ch]] = (SAFE ? rel.m2 : m[ch]) + __log2f(l);
  }
}

template <bool TAB>
__global__ __launch_bounds__(NTH) void fwd_slab_kernel(Params p, Geo g, int items_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  SRel& rel = *reinterpret_cast<SRel*>(dyn);
  char* slabs = dyn + SREL_BYTES;
  const int L = p.L, sb = L * 64, nkb = L / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  Run run{(int)blockIdx.x * items_per_wg, min((int)blockIdx.x * items_per_wg + items_per_wg, p.nseq * p.H), p.nseq};
  if (run.first >= run.last) return;
  Item it = run.at(run.first);
  copy_now(reinterpret_cast<const char*>(p.kh + slab_off(p, it)), reinterpret_cast<const char*>(p.vh + slab_off(p, it)), slabs, sb, L);
  const int mine = wave < nkb ? (nkb - wave + SW - 1) / SW : 0;       // query blocks of this wave per item: w, w + SW, ...
  // query fragments of a unit (blocks qb0, qb0 + SW; the second only when it exists): requested one unit ahead
  auto load_q = [&](Frag (&q)[2], const Item& item, int qb0) {
    const int64_t so = slab_off(p, item);
    q[0] = global_row(p.qh + so + (int64_t)(qb0 * 32 + c) * D, half);
    const int qb1 = qb0 + SW < nkb ? qb0 + SW : qb0;
    q[1] = global_row(p.qh + so + (int64_t)(qb1 * 32 + c) * D, half);
  };
  Frag qf[2];
  if (mine > 0) load_q(qf, it, wave);
  int staged_h = -1;
  for (int i = run.first; i < run.last; ++i) {
    it = run.at(i);
    __syncthreads();        // every wave is done with the previous item; the copies of this item's slabs are complete
    if (it.h != staged_h) { stage_srel<true>(rel, p, g, it.h); staged_h = it.h; }
    const int buf = (i - run.first) & 1;
    char* cur = slabs + buf * 2 * sb;
    const bool more = i + 1 < run.last;
    const Item nx = more ? run.at(i + 1) : it;
    Copier cp{more ? reinterpret_cast<const char*>(p.kh + slab_off(p, nx)) : nullptr, reinterpret_cast<const char*>(p.vh + slab_off(p, nx)),
              slabs + (buf ^ 1) * 2 * sb, sb, 2 * L * 4, L * 4, {}};
    if (mine == 0) { for (int t = 0; t < nkb; ++t) cp.step(t, nkb); continue; }
    // units: pairs (qb, qb + SW) while two blocks remain, then a single block
    for (int k = 0; k < mine; k += 2) {
      const int qb0 = wave + k * SW;
      const bool pair = k + 1 < mine;
      drain_vmem();        // this unit's query fragments (requested during the previous unit) have landed
      const Frag qc[2] = {qf[0], qf[1]};
      // request the next unit's fragments: the next unit of this item, else the first unit of the next item
      if (k + 2 < mine) load_q(qf, it, wave + (k + 2) * SW);
      else if (more) load_q(qf, nx, wave);
      const bool cpy = k == 0;
      if (rel.safe) { if (pair) fwd_unit<true, TAB, 2>(p, rel, g, cur, cur + sb, cp, cpy, it, qb0, lane, qc); else fwd_unit<true, TAB, 1>(p, rel, g, cur, cur + sb, cp, cpy, it, qb0, lane, qc); }
      else { if (pair) fwd_unit<false, TAB, 2>(p, rel, g, cur, cur + sb, cp, cpy, it, qb0, lane, qc); else fwd_unit<false, TAB, 1>(p, rel, g, cur, cur + sb, cp, cpy, it, qb0, lane, qc); }
    }
  }
}

// ================================================================================================================== dQ pass
// slabs: K^ and V.  A wave takes its query blocks one after the other.  Publishes dO' = w dO and delta' = w delta (see attn2.hip).
struct QRow { Frag q; u32x4 dlo, dhi, olo, ohi; float lse2; };     // the lane's own rows of one query block, raw bf16 (prefetched)
__device__ __forceinline__ void load_qrow(QRow& r, const Params& p, const Item& item, int qi, int half) {
  const int64_t slab = slab_off(p, item);
  r.q = global_row(p.qh + slab + (int64_t)qi * D, half);
  const int64_t tok = (int64_t)item.seq * p.L + qi;
  const bf16_t* dsrc = p.dout + tok * p.lddo + item.h * D + 8 * half;
  const bf16_t* osrc = p.o + tok * p.ldo + item.h * D + 8 * half;
  r.dlo = *reinterpret_cast<const u32x4*>(dsrc); r.dhi = *reinterpret_cast<const u32x4*>(dsrc + 16);
  r.olo = *reinterpret_cast<const u32x4*>(osrc); r.ohi = *reinterpret_cast<const u32x4*>(osrc + 16);
  r.lse2 = p.lse2[(int64_t)item.h * p.M + tok];
}
__device__ __forceinline__ void unpack8(const u32x4& a, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(a[i] << 16); v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u); }
}

template <bool SAFE, bool TAB>
__device__ __forceinline__ void dq_block(const Params& p, const SRel& rel, const Geo& g, const char* kslab, const char* vslab, Copier& cp, bool do_copy, const Item& it,
                                         int qb, int lane, const QRow& row) {
  const int L = p.L, nkb = L / 32;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const TrOff tr = tr_offsets(lane);
  const int qi = qb * 32 + c;
  const int ucol = g.u(qi);
  const int64_t slab = slab_off(p, it);
  const int64_t tok = (int64_t)it.seq * L + qi;
  float dov[16], ov[16];
  unpack8(row.dlo, dov); unpack8(row.dhi, dov + 8); unpack8(row.olo, ov); unpack8(row.ohi, ov + 8);
  float delta = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) delta += dov[e] * ov[e];
  delta = half_sum(delta);
  const float w = SAFE ? __builtin_amdgcn_exp2f(rel.m2 - row.lse2) : 1.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) dov[e] *= w;
  const Frag dof = pack(dov);
  const float deltap = delta * w;
  {
    bf16_t* dst = p.dop + slab + (int64_t)qi * D;
    *reinterpret_cast<bf16x8*>(dst + 8 * half) = dof.v[0];
    *reinterpret_cast<bf16x8*>(dst + 16 + 8 * half) = dof.v[1];
    if (half == 0) p.deltap[(int64_t)it.h * p.M + tok] = deltap;
  }
  f32x16 cdel, dqacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { cdel[r] = -deltap; dqacc[r] = 0.f; }
  const float nlse = -row.lse2;
  for (int t = 0; t < nkb; ++t) {
    const char* ktile = kslab + t * TILE;
    const char* vtile = vslab + t * TILE;
    const f32x16 cb = sbias<true, TAB>(rel, g, ucol, t * 32, half);
    const Frag kf = lds_rows(ktile, ar, half);
    const Frag vf = lds_rows(vtile, ar, half);
    // S and dP are independent: interleave their dependent MFMA pairs
    f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v[0], row.q.v[0], cb, 0, 0, 0);
    f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v[0], dof.v[0], cdel, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v[1], row.q.v[1], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v[1], dof.v[1], dp, 0, 0, 0);
    const Frag ktf = lds_cols(ktile, tr);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(SAFE ? s[r] : s[r] + nlse) * dp[r];
    const Frag dsf = pack(ds);
    dqacc = mma(dqacc, ktf, dsf);
    if (do_copy) cp.step(t, nkb);
  }
  {                                   // dq^ = scale * dS k^ ; c = scale * log2 e
    const float sc = p.c * LN2;
    bf16_t* dst = p.dqh + slab + (int64_t)qi * D;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = dqacc[8 * gq + e] * sc;
      store8(dst + 16 * gq + 8 * half, o8);
    }
  }
}

template <bool TAB>
__global__ __launch_bounds__(NTH) void dq_slab_kernel(Params p, Geo g, int items_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  SRel& rel = *reinterpret_cast<SRel*>(dyn);
  char* slabs = dyn + SREL_BYTES;
  const int L = p.L, sb = L * 64, nkb = L / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  Run run{(int)blockIdx.x * items_per_wg, min((int)blockIdx.x * items_per_wg + items_per_wg, p.nseq * p.H), p.nseq};
  if (run.first >= run.last) return;
  Item it = run.at(run.first);
  copy_now(reinterpret_cast<const char*>(p.kh + slab_off(p, it)), reinterpret_cast<const char*>(p.vh + slab_off(p, it)), slabs, sb, L);
  const int mine = wave < nkb ? (nkb - wave + SW - 1) / SW : 0;
  QRow nxt;            // rows of the NEXT block this wave will process (requested one block ahead)
  if (mine > 0) load_qrow(nxt, p, it, wave * 32 + c, half);
  int staged_h = -1;
  for (int i = run.first; i < run.last; ++i) {
    it = run.at(i);
    __syncthreads();
    if (it.h != staged_h) { stage_srel<true>(rel, p, g, it.h); staged_h = it.h; }
    const int buf = (i - run.first) & 1;
    char* cur = slabs + buf * 2 * sb;
    const bool more = i + 1 < run.last;
    const Item nx = more ? run.at(i + 1) : it;
    Copier cp{more ? reinterpret_cast<const char*>(p.kh + slab_off(p, nx)) : nullptr, reinterpret_cast<const char*>(p.vh + slab_off(p, nx)),
              slabs + (buf ^ 1) * 2 * sb, sb, 2 * L * 4, L * 4, {}};
    if (mine == 0) { for (int t = 0; t < nkb; ++t) cp.step(t, nkb); continue; }
    for (int k = 0; k < mine; ++k) {
      const int qb = wave + k * SW;
      drain_vmem();
      const QRow row = nxt;
      if (k + 1 < mine) load_qrow(nxt, p, it, (qb + SW) * 32 + c, half);
      else if (more) load_qrow(nxt, p, nx, wave * 32 + c, half);
      const bool cpy = k == 0;
      if (rel.safe) dq_block<true, TAB>(p, rel, g, cur, cur + sb, cp, cpy, it, qb, lane, row);
      else dq_block<false, TAB>(p, rel, g, cur, cur + sb, cp, cpy, it, qb, lane, row);
    }
  }
}

// ================================================================================================================== dK, dV pass
// slabs: Q~ and dO' (head-planar, written by the dQ pass); -delta' of the item's queries in LDS.  lane = key.
// TOK: the un-prep of k / v folded into the epilogue (Params::dk_tok): row-major dk = l2norm backward of dk^ (ctclip_attn2_unprep's arithmetic on
// the bf16-rounded dk^), row-major dv, and the lane's share of the k_scale gradient accumulated in ksacc over the whole kernel.
template <bool SAFE, bool TAB, bool TOK>
__device__ __forceinline__ void dkv_block(const Params& p, const SRel& rel, const Geo& g, const char* qslab, const char* doslab, const float* ndelta,
                                          const float* lse_it, Copier& cp, bool do_copy, const Item& it, int jb, int lane, const Frag& kf, const Frag& vf,
                                          float (&ksacc)[16]) {
  const int L = p.L, nqb = L / 32;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const TrOff tr = tr_offsets(lane);
  const int kj = jb * 32 + c;
  const int ucol = g.u(kj);
  f32x16 dkacc, dvacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkacc[r] = 0.f; dvacc[r] = 0.f; }
  for (int t = 0; t < nqb; ++t) {
    const char* qtile = qslab + t * TILE;
    const char* dotile = doslab + t * TILE;
    f32x16 cb = sbias<false, TAB>(rel, g, ucol, t * 32, half);
    const Frag qf = lds_rows(qtile, ar, half);
    const Frag dof = lds_rows(dotile, ar, half);
    f32x16 cdel;        // -delta' of the tile's queries in register order: queries 8 half + e and 16 + 8 half + e
    {
      const float* sp = ndelta + t * 32 + 8 * half;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp), a1 = *reinterpret_cast<const f32x4*>(sp + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sp + 16), b1 = *reinterpret_cast<const f32x4*>(sp + 20);
#pragma unroll
      for (int e = 0; e < 4; ++e) { cdel[e] = a0[e]; cdel[4 + e] = a1[e]; cdel[8 + e] = b0[e]; cdel[12 + e] = b1[e]; }
    }
    if (!SAFE) {        // slow path: lse2 of the tile's queries straight from global memory (L1 / L2 resident)
      const float* sp = lse_it + t * 32 + 8 * half;
#pragma unroll
      for (int e = 0; e < 8; ++e) { cb[e] -= sp[e]; cb[8 + e] -= sp[16 + e]; }
    }
    f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf.v[0], kf.v[0], cb, 0, 0, 0);
    f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof.v[0], vf.v[0], cdel, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf.v[1], kf.v[1], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof.v[1], vf.v[1], dp, 0, 0, 0);
    const Frag dotf = lds_cols(dotile, tr);
    const Frag qtf = lds_cols(qtile, tr);
    float pr[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(s[r]); ds[r] = pr[r] * dp[r]; }
    const Frag pf = pack(pr), dsf = pack(ds);
    // the two accumulations are independent: interleave
    dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf.v[0], pf.v[0], dvacc, 0, 0, 0);
    dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf.v[0], dsf.v[0], dkacc, 0, 0, 0);
    dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf.v[1], pf.v[1], dvacc, 0, 0, 0);
    dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf.v[1], dsf.v[1], dkacc, 0, 0, 0);
    if (do_copy) cp.step(t, nqb);
  }
  if (TOK) {                          // un-prep in place: u = k^ / k_scale, g = dk^ k_scale, dk = kinv (g - u (u . g)); dscale += dk^ u
    const int64_t tok = (int64_t)it.seq * L + kj;
    const float ik = p.kinv[tok * p.H + it.h];
    float kx[16], dk[16];
    unpack8(__builtin_bit_cast(u32x4, kf.v[0]), kx); unpack8(__builtin_bit_cast(u32x4, kf.v[1]), kx + 8);
    float part[2] = {0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = 8 * gq + e;
        const float ks = p.k_scale[16 * gq + 8 * half + e];
        const float rk = fabsf(ks) > 1e-30f ? 1.f / ks : 0.f;
        const float gk0 = bf2f(f2bf(dkacc[i] * LN2));            // the bf16 value the planar path stores and the un-prep kernel reads
        const float uk = kx[i] * rk;
        ksacc[i] += gk0 * uk;
        const float gk = gk0 * ks;
        part[gq] += uk * gk;
        kx[i] = uk; dk[i] = gk;
      }
    // (u . g) over the 32 head dims = the four 8-dim chunks in the un-prep kernel's order: (c0 + c1) + (c2 + c3); this lane holds chunks
    // half and 2 + half, its partner (lane ^ 32) the other two
    const float dot = (half_sum(part[0])) + (half_sum(part[1]));
    bf16_t* dK = p.dk_tok + tok * p.ldk_tok + it.h * D;
    bf16_t* dV = p.dv_tok + tok * p.ldv_tok + it.h * D;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      float a8[8], b8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a8[e] = ik * (dk[8 * gq + e] - kx[8 * gq + e] * dot); b8[e] = dvacc[8 * gq + e]; }
      store8(dK + 16 * gq + 8 * half, a8);
      store8(dV + 16 * gq + 8 * half, b8);
    }
  } else {                            // dk^ = scale * dS^T q^ = ln 2 * dS^T q~
    const int64_t slab = slab_off(p, it);
    bf16_t* dK = p.dkh + slab + (int64_t)kj * D;
    bf16_t* dV = p.dvh + slab + (int64_t)kj * D;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      float a8[8], b8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a8[e] = dkacc[8 * gq + e] * LN2; b8[e] = dvacc[8 * gq + e]; }
      store8(dK + 16 * gq + 8 * half, a8);
      store8(dV + 16 * gq + 8 * half, b8);
    }
  }
}

template <bool TAB, bool TOK = false>
__global__ __launch_bounds__(NTH) void dkv_slab_kernel(Params p, Geo g, int items_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  SRel& rel = *reinterpret_cast<SRel*>(dyn);
  const int L = p.L, sb = L * 64, nqb = L / 32;
  float* stats = reinterpret_cast<float*>(dyn + SREL_BYTES);           // [buffer][SL_MAX]: -delta'
  char* slabs = dyn + SREL_BYTES + 2 * SL_MAX * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  Run run{(int)blockIdx.x * items_per_wg, min((int)blockIdx.x * items_per_wg + items_per_wg, p.nseq * p.H), p.nseq};
  if (run.first >= run.last) return;
  Item it = run.at(run.first);
  copy_now(reinterpret_cast<const char*>(p.qh + slab_off(p, it)), reinterpret_cast<const char*>(p.dop + slab_off(p, it)), slabs, sb, L);
  for (int i = threadIdx.x; i < L; i += NTH) stats[i] = -p.deltap[(int64_t)it.h * p.M + (int64_t)it.seq * L + i];
  const int mine = wave < nqb ? (nqb - wave + SW - 1) / SW : 0;
  float ksacc[16];     // TOK: this lane's share of the k_scale gradient (its 16 head dims), over every key row it visits
#pragma unroll
  for (int i = 0; i < 16; ++i) ksacc[i] = 0.f;
  auto load_kv = [&](Frag& k, Frag& v, const Item& item, int jb) {
    const int64_t so = slab_off(p, item) + (int64_t)(jb * 32 + c) * D;
    k = global_row(p.kh + so, half); v = global_row(p.vh + so, half);
  };
  Frag kn, vn;         // key / value rows of the NEXT block this wave will process
  if (mine > 0) load_kv(kn, vn, it, wave);
  int staged_h = -1;
  for (int i = run.first; i < run.last; ++i) {
    it = run.at(i);
    __syncthreads();
    if (it.h != staged_h) { stage_srel<false>(rel, p, g, it.h); staged_h = it.h; }
    const int buf = (i - run.first) & 1;
    char* cur = slabs + buf * 2 * sb;
    const float* ndelta = stats + buf * SL_MAX;
    const bool more = i + 1 < run.last;
    const Item nx = more ? run.at(i + 1) : it;
    Copier cp{more ? reinterpret_cast<const char*>(p.qh + slab_off(p, nx)) : nullptr, reinterpret_cast<const char*>(p.dop + slab_off(p, nx)),
              slabs + (buf ^ 1) * 2 * sb, sb, 2 * L * 4, L * 4, {}};
    // -delta' of the next item: requested now (up to two values per thread), written after this wave's blocks
    float sv[2] = {0.f, 0.f};
    if (more) {
#pragma unroll
      for (int k = 0; k < 2; ++k) { const int q = k * NTH + (int)threadIdx.x; if (q < L) sv[k] = -p.deltap[(int64_t)nx.h * p.M + (int64_t)nx.seq * L + q]; }
    }
    const float* lse_it = p.lse2 + (int64_t)it.h * p.M + (int64_t)it.seq * L;        // only read on the unbounded path
    if (mine == 0) { for (int t = 0; t < nqb; ++t) cp.step(t, nqb); }
    for (int k = 0; k < mine; ++k) {
      const int jb = wave + k * SW;
      drain_vmem();
      const Frag kf = kn, vf = vn;
      if (k + 1 < mine) load_kv(kn, vn, it, jb + SW);
      else if (more) load_kv(kn, vn, nx, wave);
      const bool cpy = k == 0;
      if (rel.safe) dkv_block<true, TAB, TOK>(p, rel, g, cur, cur + sb, ndelta, lse_it, cp, cpy, it, jb, lane, kf, vf, ksacc);
      else dkv_block<false, TAB, TOK>(p, rel, g, cur, cur + sb, ndelta, lse_it, cp, cpy, it, jb, lane, kf, vf, ksacc);
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < 2; ++k) { const int q = k * NTH + (int)threadIdx.x; if (q < L) stats[(buf ^ 1) * SL_MAX + q] = sv[k]; }
    }
  }
  if (TOK) {           // k_scale gradient of this workgroup, in a fixed order: the 32 key lanes of a half by an xor tree, then the eight waves
#pragma unroll
    for (int i = 0; i < 16; ++i)
      ksacc[i] = half32_sum(ksacc[i]);
    __syncthreads();                                           // the slabs are free now
    float* red = reinterpret_cast<float*>(slabs);              // [SW][32]
    if (c == 0) {
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * 32 + 16 * gq + 8 * half + e] = ksacc[8 * gq + e];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < SW; ++w) t += red[w * 32 + threadIdx.x];
      p.kpart[(int64_t)blockIdx.x * 32 + threadIdx.x] = t;
    }
  }
}

// ================================================================================================================== dBias
// dBias[h][i][j] = sum over sequences of dS.  A workgroup of eight waves owns 4 query blocks x 4 key blocks of one head; wave (a, b)
// holds the two tile pairs {2a, 2a+1} x {b} (shared K^ / V fragments, 32 accumulators) and walks a strided subset of the sequences.
// Per sequence the 16 operand tiles (Q~, dO', K^, V of the four blocks each: 32 KB) are staged once, double-buffered, one barrier per
// sequence and two independent tile pairs per wave between barriers.  (The first version: one pair per wave and barrier,
// SQ_WAIT_ANY 61 %, 302 us.  128 registers -- two workgroups per CU -- spill: 300 B per lane.)
constexpr int DB_W = 8, DB_NTH = DB_W * 64, DB_G = 4;
constexpr int DB_TILES = 4 * DB_G;                                  // Q~ x4, dO' x4, K^ x4, V x4
constexpr int DB_PIECES = (DB_TILES * 128 + DB_NTH - 1) / DB_NTH;   // 16-byte pieces per thread and sequence (128 per tile)

template <bool SAFE>
__device__ __forceinline__ void dbias_item(const Params& p, const SRel& rel, const Geo& g, char* tiles, float* stats, int h, int qg, int kg, int split) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.L, nkb = L / 32;
  const int wa = wave >> 2, wb = wave & 3;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  int qb[2]; bool qok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int qr = qg * DB_G + 2 * wa + i; qok[i] = qr < nkb; qb[i] = qok[i] ? qr : nkb - 1; }
  const int kr = kg * DB_G + wb;
  const bool kok = kr < nkb;
  const int kb = kok ? kr : nkb - 1;
  const int ucol[2] = {g.u(qb[0] * 32 + c), g.u(qb[1] * 32 + c)};
  float acc[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // staging: piece = (tile, row, 16-byte chunk); tiles 0..3 Q~, 4..7 dO', 8..11 K^, 12..15 V of blocks group * 4 + (tile % 4)
  const int64_t hbase = (int64_t)h * p.M * D;
  auto piece_src = [&](int pc, int seq) -> const u32x4* {
    const int tile = pc >> 7, row = (pc >> 2) & 31, chunk = pc & 3;
    const int kind = tile / DB_G, blk = tile % DB_G;
    const bf16_t* base = kind == 0 ? p.qh : kind == 1 ? p.dop : kind == 2 ? p.kh : p.vh;
    int b = (kind < 2 ? qg : kg) * DB_G + blk;
    b = b < nkb ? b : nkb - 1;
    return reinterpret_cast<const u32x4*>(base + hbase + ((int64_t)seq * L + b * 32 + row) * D + chunk * 8);
  };
  auto piece_dst = [&](int pc) { const int tile = pc >> 7, row = (pc >> 2) & 31, chunk = pc & 3; return tile * TILE + swz(row, chunk); };
  // statistics: threads 0 .. 4*64-1: [query block][lse2 | delta'][32]
  const bool sth = (int)threadIdx.x < DB_G * 64;
  auto stat_src = [&](int seq) {
    int b = qg * DB_G + ((int)threadIdx.x >> 6);
    b = b < nkb ? b : nkb - 1;
    return (((int)threadIdx.x & 32) ? p.deltap : p.lse2) + (int64_t)h * p.M + (int64_t)seq * L + b * 32 + (threadIdx.x & 31);
  };
  int seq = split;
  u32x4 st[DB_PIECES];
  float sv = 0.f;
#pragma unroll
  for (int k = 0; k < DB_PIECES; ++k) st[k] = *piece_src(k * DB_NTH + (int)threadIdx.x, seq);
  if (sth) sv = *stat_src(seq);
#pragma unroll
  for (int k = 0; k < DB_PIECES; ++k) *reinterpret_cast<u32x4*>(tiles + piece_dst(k * DB_NTH + (int)threadIdx.x)) = st[k];
  if (sth) stats[threadIdx.x] = sv;
  __syncthreads();
  for (int it = 0; seq < p.nseq; seq += p.nsplit, ++it) {
    const int buf = it & 1;
    const int sn = seq + p.nsplit < p.nseq ? seq + p.nsplit : seq;
#pragma unroll
    for (int k = 0; k < DB_PIECES; ++k) st[k] = *piece_src(k * DB_NTH + (int)threadIdx.x, sn);
    if (sth) sv = *stat_src(sn);
    const char* tb = tiles + buf * DB_TILES * TILE;
    const float* sb = stats + buf * DB_G * 64;
    const Frag kf = lds_rows(tb + (2 * DB_G + wb) * TILE, ar, half);
    const Frag vf = lds_rows(tb + (3 * DB_G + wb) * TILE, ar, half);
    f32x16 s[2], dp[2];
    Frag qf[2], dof[2];
    float lse2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qt = 2 * wa + i;
      qf[i] = lds_rows(tb + qt * TILE, c, half);                               // B operands: the lane's own query row
      dof[i] = lds_rows(tb + (DB_G + qt) * TILE, c, half);
      lse2[i] = sb[qt * 64 + c];
      const float deltap = sb[qt * 64 + 32 + c];
      s[i] = sbias<true, true>(rel, g, ucol[i], kb * 32, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[i][r] = -deltap;
    }
    // four independent dependent-pairs of MFMAs, interleaved
#pragma unroll
    for (int i = 0; i < 2; ++i) { s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v[0], qf[i].v[0], s[i], 0, 0, 0); dp[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v[0], dof[i].v[0], dp[i], 0, 0, 0); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v[1], qf[i].v[1], s[i], 0, 0, 0); dp[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v[1], dof[i].v[1], dp[i], 0, 0, 0); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = fmaf(__builtin_amdgcn_exp2f(SAFE ? s[i][r] : s[i][r] - lse2[i]), dp[i][r], acc[i][r]);
    char* tn = tiles + (buf ^ 1) * DB_TILES * TILE;
#pragma unroll
    for (int k = 0; k < DB_PIECES; ++k) *reinterpret_cast<u32x4*>(tn + piece_dst(k * DB_NTH + (int)threadIdx.x)) = st[k];
    if (sth) stats[(buf ^ 1) * DB_G * 64 + threadIdx.x] = sv;
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!(qok[i] && kok)) continue;
    float* dst = p.dbias_part + (((int64_t)split * p.H + h) * L + qb[i] * 32 + c) * L + kb * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[slot_index(r, half)] = acc[i][r];
  }
}

__global__ __launch_bounds__(DB_NTH, 2) void dbias_slab_kernel(Params p, Geo g) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  SRel& rel = *reinterpret_cast<SRel*>(dyn);
  float* stats = reinterpret_cast<float*>(dyn + SREL_BYTES);                       // [2][4][lse2 | delta'][32]
  char* tiles = dyn + SREL_BYTES + 2 * DB_G * 64 * 4;                              // [2][16][TILE]
  const int nkb = p.L / 32, ngrp = (nkb + DB_G - 1) / DB_G;
  // XCD-aware work map.  The ngrp^2 tile groups of one (head, sequence subset) "family" read the same Q~ / dO' / K^ / V planes (each
  // tile by ngrp of them), so a family belongs on ONE XCD, in adjacent dispatch slots: workgroup ids are dealt round-robin to the 8
  // XCDs, hence xcd = id % 8 picks the family column and id / 8 walks (family row, group).  (With the plain 3-D grid the groups of a
  // family landed on eight different L2s and every plane was fetched five times: 1.26 GB per launch, 258 us.)
  const int ngg = ngrp * ngrp, nfam = p.H * p.nsplit;
  const int slot = blockIdx.x >> 3, fam = (slot / ngg) * 8 + (blockIdx.x & 7), grp = slot % ngg;
  if (fam >= nfam) return;                                                         // (whole workgroup: before any barrier)
  const int qg = grp / ngrp, kg = grp % ngrp, h = fam % p.H, split = fam / p.H;
  stage_srel<true>(rel, p, g, h);                                                  // (NTH == DB_NTH)
  if (split >= p.nseq) return;
  if (rel.safe) dbias_item<true>(p, rel, g, tiles, stats, h, qg, kg, split);
  else dbias_item<false>(p, rel, g, tiles, stats, h, qg, kg, split);
}

int num_cus() {
  static int ncu = 0;
  if (!ncu) { int dev = 0; hipDeviceProp_t prop; (void)hipGetDevice(&dev); ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
  return ncu;
}
bool slab_enabled() {
  static int use = -1;
  if (use < 0) { const char* e = getenv("CTCLIP_ATTN_SLAB"); use = (e && e[0] == '0') ? 0 : 1; }
  return use != 0;
}
bool make_geo(const Params& p, Geo& g, size_t extra_lds, size_t& shm) {
  if (!slab_enabled() || p.L > SL_MAX || p.L % 32) return false;
  const int nkb = p.L / 32, cpt = (2 * p.L * 4 + NTH - 1) / NTH;
  if (cpt > CP_BATCH * ((nkb + 3) / 4)) return false;          // the copy schedule: CP_BATCH pieces per thread every fourth step
  g = Geo{1, 1, 0, 1, 65536};
  if (p.tab) {
    if (p.gh * p.gw != p.L || p.gw % 8 || p.gw > 64) return false;
    const int S = table_stride(p.gw);
    g = Geo{p.gw, S, (p.gh - 1) * S + (p.gw - 1), (2 * p.gh - 1) * S, (65536 + p.gw - 1) / p.gw};
    if (g.n > STAB_MAX) return false;
  }
  shm = (size_t)SREL_BYTES + extra_lds + 4 * (size_t)p.L * 64;
  return shm <= 160 * 1024;
}
template <class K>
bool raise_lds(K kern) { return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }

}  // namespace

#define SLAB_LAUNCH(KERN, EXTRA, WHAT)                                                                              \
  Geo g; size_t shm;                                                                                                \
  if (!make_geo(p, g, EXTRA, shm)) return 1;                                                                        \
  static bool raised = false;                                                                                       \
  if (!raised) {                                                                                                    \
    if (!raise_lds(KERN<true>) || !raise_lds(KERN<false>)) { ctclip_set_error(WHAT ": cannot raise the LDS limit"); return CTCLIP_EBADARG; } \
    raised = true;                                                                                                  \
  }                                                                                                                 \
  const int ncu = num_cus(), total = p.nseq * p.H;                                                                  \
  const int ipw = (total + ncu - 1) / ncu, nwg = (total + ipw - 1) / ipw;                                           \
  if (p.tab) hipLaunchKernelGGL(KERN<true>, dim3((unsigned)nwg), dim3(NTH), shm, stream, p, g, ipw);                \
  else hipLaunchKernelGGL(KERN<false>, dim3((unsigned)nwg), dim3(NTH), shm, stream, p, g, ipw);                     \
  return ctclip_check_launch(WHAT);

int attn2_slab_fwd(const ctclip_attn2::Params& p, hipStream_t stream) { SLAB_LAUNCH(fwd_slab_kernel, 0, "attn2_fwd (slab)") }
int attn2_slab_bwd_dq(const ctclip_attn2::Params& p, hipStream_t stream) { SLAB_LAUNCH(dq_slab_kernel, 0, "attn2_bwd_dq (slab)") }
int attn2_slab_bwd_dkv(const ctclip_attn2::Params& p, hipStream_t stream, int* nwg_out) {
  Geo g; size_t shm;
  if (!make_geo(p, g, 2 * SL_MAX * 4, shm)) return 1;
  static bool raised = false;
  if (!raised) {
    if (!raise_lds(dkv_slab_kernel<true, false>) || !raise_lds(dkv_slab_kernel<false, false>) || !raise_lds(dkv_slab_kernel<true, true>) ||
        !raise_lds(dkv_slab_kernel<false, true>)) { ctclip_set_error("attn2_bwd_dkv (slab): cannot raise the LDS limit"); return CTCLIP_EBADARG; }
    raised = true;
  }
  const int ncu = num_cus(), total = p.nseq * p.H;
  const int ipw = (total + ncu - 1) / ncu, nwg = (total + ipw - 1) / ipw;
  if (nwg_out) *nwg_out = nwg;
  const bool tok = p.dk_tok != nullptr;
  if (p.tab) { if (tok) hipLaunchKernelGGL((dkv_slab_kernel<true, true>), dim3((unsigned)nwg), dim3(NTH), shm, stream, p, g, ipw);
               else hipLaunchKernelGGL((dkv_slab_kernel<true, false>), dim3((unsigned)nwg), dim3(NTH), shm, stream, p, g, ipw); }
  else { if (tok) hipLaunchKernelGGL((dkv_slab_kernel<false, true>), dim3((unsigned)nwg), dim3(NTH), shm, stream, p, g, ipw);
         else hipLaunchKernelGGL((dkv_slab_kernel<false, false>), dim3((unsigned)nwg), dim3(NTH), shm, stream, p, g, ipw); }
  return ctclip_check_launch("attn2_bwd_dkv (slab)");
}

// dBias slabs (p.dbias_part, p.nsplit must be set): returns 1 when the shape is not eligible
int attn2_slab_bwd_dbias(const ctclip_attn2::Params& p, hipStream_t stream) {
  Geo g; size_t shm0;
  if (!p.tab || !make_geo(p, g, 0, shm0)) return 1;
  const size_t shm = (size_t)SREL_BYTES + 2 * DB_G * 64 * 4 + 2 * (size_t)DB_TILES * TILE;
  static bool raised = false;
  if (!raised) {
    if (!raise_lds(dbias_slab_kernel)) { ctclip_set_error("attn2_bwd_dbias (slab): cannot raise the LDS limit"); return CTCLIP_EBADARG; }
    raised = true;
  }
  const int nkb = p.L / 32, ngrp = (nkb + DB_G - 1) / DB_G;
  const int nfam8 = (p.H * p.nsplit + 7) / 8;                                       // family rows of 8 (one family per XCD)
  hipLaunchKernelGGL(dbias_slab_kernel, dim3((unsigned)(nfam8 * ngrp * ngrp * 8)), dim3(DB_NTH), shm, stream, p, g);
  return ctclip_check_launch("attn2_bwd_dbias (slab)");
}
