// Shared device helpers of the second-generation attention kernels (attn2.hip: ring kernels, prep / un-prep, dBias, C ABI;
// attn2_slab.hip: the persistent slab-resident kernels).  See the header comment of attn2.hip for the design.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace ctclip_attn2 {
struct Params {
  const bf16_t *qh, *kh, *vh;              // head-planar [H][M][32]: q~, k^, v
  const float* tab;                        // (ncls, H) position-bias table (natural units), or null
  const float *q_scale, *k_scale;          // (32) learned scales (for the logit bound)
  int gh, gw, H, L, nseq;
  int64_t M;                               // nseq * L
  float c;                                 // scale * log2 e (already folded into q~)
  // forward
  bf16_t* out; int64_t ldo;                // (M, >= H*32) row-major
  float* lse2;                             // [H][M] log2-domain log-sum-exp
  // backward
  const bf16_t* o; const bf16_t* dout; int64_t lddo;
  bf16_t* dop;                             // [H][M][32]: dO' = w dO (written by the query pass, read by the key pass and dBias)
  float* deltap;                           // [H][M]: delta' = w delta
  bf16_t *dqh, *dkh, *dvh;                 // head-planar gradients
  float* dbias_part; int nsplit;           // dBias slabs [nsplit][H][L][L]
  // fused k / v un-prep of the slab key pass (ctclip_attn2_bwd_tok): when dk_tok is set, the pass applies the l2norm backward of k itself and
  // writes ROW-MAJOR dk (M, ldk_tok) / dv (M, ldv_tok) instead of the head-planar dkh / dvh; kinv = the inverse norms (M, H) of the forward,
  // kpart = per-workgroup partial sums [gridDim][32] of the k_scale gradient (summed in workgroup order by the host's reduce launch)
  bf16_t* dk_tok; bf16_t* dv_tok; int64_t ldk_tok, ldv_tok; const float* kinv; float* kpart;
};
// arguments of the four-wave one-pass backward (attn2_bwd2.hip) beyond Params; partial layouts as in attn2_bwd1.hip
struct Bwd2Args {
  const float* qinv; bf16_t* dq_tok; int64_t lddq;
  float* qpart;                            // [nwg][32] q_scale gradient partials (k: Params::kpart)
  float* dtpart;                           // [nseq][H][ncls] table-gradient partials, one per (workgroup, item), or null
  int ipw, wph;                            // items per workgroup, workgroups per head
  unsigned long long* stamps;              // profiling aid (tools/bench_attn2_bwd.py --stamps): 100-MHz clock at the phase boundaries of workgroup 0, or null
};
}  // namespace ctclip_attn2

namespace {

using ctclip_attn2::Params;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int D = 32;                       // head dim
constexpr int TILE = 32 * 64;               // bytes of one 32-row operand tile
constexpr int MAXCLS = 4096, MAXL = 1024;
constexpr float SAFE_SPAN = 100.f;          // log2 units: exp2(-100) ~ 8e-31 is a normal f32 / bf16 number

__device__ __forceinline__ int pi32(int c) { return (c & 3) | ((c & 4) << 1) | ((c & 8) >> 1) | (c & 16); }
__device__ __forceinline__ int slot_index(int r, int half) { return 16 * (r >> 3) + 8 * half + (r & 7); }
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

struct Frag { bf16x8 v[2]; };               // 32 contraction slots of one lane: slots 8 half + e (v[0]) and 16 + 8 half + e (v[1])

__device__ __forceinline__ f32x16 mma(f32x16 acc, const Frag& a, const Frag& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0], b.v[0], acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1], b.v[1], acc, 0, 0, 0);
}
__device__ __forceinline__ Frag pack(const float (&p)[16]) {
  Frag f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack2bf(p[8 * g + 2 * e], p[8 * g + 2 * e + 1]);
    f.v[g] = __builtin_bit_cast(bf16x8, w);
  }
  return f;
}
// row-major fragment of LDS tile row `row` (64 B, swizzled): contraction slots = the lane's 16 head dims 8 half + e, 16 + 8 half + e
__device__ __forceinline__ Frag lds_rows(const char* tile, int row, int half) {
  Frag f;
  f.v[0] = *reinterpret_cast<const bf16x8*>(tile + swz(row, half));
  f.v[1] = *reinterpret_cast<const bf16x8*>(tile + swz(row, 2 + half));
  return f;
}
// the same from global memory: the lane's own token row of a head-planar operand (64 B)
__device__ __forceinline__ Frag global_row(const bf16_t* row, int half) {
  Frag f;
  f.v[0] = *reinterpret_cast<const bf16x8*>(row + 8 * half);
  f.v[1] = *reinterpret_cast<const bf16x8*>(row + 16 + 8 * half);
  return f;
}
// Transposed fragment of a row-major LDS tile [token][32 dims] with ds_read_b64_tr_b16: output row (MFMA A row) i = lane & 31 is
// head dim pi32(i), contraction slots are the tile's tokens 8 half + e (v[0]) and 16 + 8 half + e (v[1]).  Measured semantics
// (tools/tr_probe.hip): in each 16-lane group, output lane t element j = element t & 3 of the 8 bytes addressed by lane
// 4 j + (t >> 2).  Lane t' therefore points at token k0 + (t' >> 2), dims cbase + 4 sigma(t' & 3) .. + 3 (sigma swaps 1 and 2, which
// realises pi32 inside the group), cbase = 16 * ((lane >> 4) & 1).  troff[j0] = the lane's byte offset for k0 = 8 half + 4 j0; the second
// half of the tile (tokens 16..31) is 1024 B further on and has the same swizzle phase.
struct TrOff { uint32_t o[2]; };
__device__ __forceinline__ TrOff tr_offsets(int lane) {
  const int t = lane & 15, grp = (lane >> 4) & 1, half = lane >> 5;
  const int sg = ((t & 1) << 1) | ((t >> 1) & 1);          // sigma(t & 3)
  TrOff r;
#pragma unroll
  for (int j0 = 0; j0 < 2; ++j0) {
    const int row = 8 * half + 4 * j0 + (t >> 2);
    r.o[j0] = (uint32_t)(swz(row, 2 * grp + (sg >> 1)) + 8 * (sg & 1));
  }
  return r;
}
__device__ __forceinline__ u32x2 tr_read(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ Frag lds_cols(const char* tile, const TrOff& tr) {
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)tile;
  u32x2 a0 = tr_read(base + tr.o[0]), a1 = tr_read(base + tr.o[1]);
  u32x2 b0 = tr_read(base + 1024 + tr.o[0]), b1 = tr_read(base + 1024 + tr.o[1]);
  // the compiler does not know that the asm reads above are asynchronous: the wait must CARRY the registers ("+v"), otherwise the
  // consumer (an MFMA) may be scheduled in front of it -- first hardware run: forward row sums right, outputs garbage
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) :: "memory");
  Frag f;
  f.v[0] = __builtin_bit_cast(bf16x8, u32x4{a0[0], a0[1], a1[0], a1[1]});
  f.v[1] = __builtin_bit_cast(bf16x8, u32x4{b0[0], b0[1], b1[0], b1[1]});
  return f;
}


// The same transposed fragment through the compiler's builtin (round 6): the loads are visible to the scheduler and to the wait-count insertion --
// no full `lgkmcnt(0)` drain in front of the consumer, and the reads can be hoisted above unrelated work.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Frag lds_cols_b(const char* tile, const TrOff& tr) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + tr.o[0]));
  const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + tr.o[1]));
  const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + 1024 + tr.o[0]));
  const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + 1024 + tr.o[1]));
  Frag f;
  f.v[0] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  f.v[1] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  return f;
}

// ---- per-workgroup preamble: stage the bias table of head h (log2 domain) and the token -> offset-class index, and derive the
// logit bound.  All threads of the workgroup must call it.  Returns M2 (log2-domain bound folded into the staged table when safe).
template <int NC, int NL>
struct RelT {
  float tab[NC];
  __attribute__((aligned(16))) uint16_t u[NL];
  float red[2][16];
  float m2; int safe;
};
using Rel = RelT<MAXCLS, MAXL>;
constexpr int SLAB_MAXCLS = 2304, SLAB_MAXL = 576;          // 24 x 24 tokens: 47^2 = 2209 classes
using RelS = RelT<SLAB_MAXCLS, SLAB_MAXL>;
// REVERSED: entry i holds class ncls - 1 - i, so that the descending classes of a key run are ASCENDING addresses and land in
// consecutive registers without moves (the kernels whose tile rows are keys); the key pass keeps the natural order.
template <bool REVERSED, class R>
__device__ __forceinline__ void stage_rel(R& rel, const Params& p, int h) {
  const int ncls = p.tab ? (2 * p.gh - 1) * (2 * p.gw - 1) : 0;
  const int nth = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
  float mx = -INFINITY, mn = INFINITY;
  for (int i = tid; i < ncls; i += nth) {
    const float t = p.tab[(int64_t)(REVERSED ? ncls - 1 - i : i) * p.H + h] * LOG2E;
    rel.tab[i] = t;
    mx = fmaxf(mx, t); mn = fminf(mn, t);
  }
  if (ncls == 0) { mx = 0.f; mn = 0.f; if (tid == 0) rel.tab[0] = 0.f; }
  for (int i = tid; i < p.L; i += nth) rel.u[i] = p.tab ? (uint16_t)((i / p.gw) * (2 * p.gw - 1) + i % p.gw) : (uint16_t)0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
  if (lane == 0) { rel.red[0][wave] = mx; rel.red[1][wave] = mn; }
  __syncthreads();
  if (wave == 0) {
    float a = lane < 32 ? fabsf(p.q_scale[lane]) : fabsf(p.k_scale[lane - 32]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));     // lanes 0-31: max|q_scale|, 32-63: max|k_scale|
    const float qk = a * __shfl_xor(a, 32, 64) * p.c;
    float tmx = -INFINITY, tmn = INFINITY;
    for (int w = 0; w < nw; ++w) { tmx = fmaxf(tmx, rel.red[0][w]); tmn = fminf(tmn, rel.red[1][w]); }
    if (lane == 0) {
      const float span = 2.f * qk + (tmx - tmn);
      rel.safe = (span <= SAFE_SPAN && span == span) ? 1 : 0;
      rel.m2 = qk + tmx;
    }
  }
  __syncthreads();
  if (rel.safe) {
    const float m2 = rel.m2;
    for (int i = tid; i < (ncls ? ncls : 1); i += nth) rel.tab[i] -= m2;
    __syncthreads();
  }
}

// offset-class gather of one 32 x 32 tile as the MFMA accumulator input.  Rows (registers) are KEYS, lane = query: class of key
// run (8 consecutive tokens of one image row; needs gw % 8 == 0) descends by one per key.  Rows are QUERIES, lane = key: ascends.
template <bool ROWS_ARE_KEYS, bool TAB, class R>
__device__ __forceinline__ f32x16 bias_tile(const R& rel, const Params& p, int ucol, int row_base, int half) {
  f32x16 cb;
  if (!TAB) {
    const float t = rel.tab[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) cb[r] = t;
    return cb;
  }
  const int c0 = (p.gh - 1) * (2 * p.gw - 1) + (p.gw - 1);
  const int ncls = (2 * p.gh - 1) * (2 * p.gw - 1);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int urow0 = rel.u[row_base + 16 * g + 8 * half];
    // keys: class(e) = ucol - urow0 + c0 - e, stored reversed at ncls - 1 - class; queries: class(e) = urow0 - ucol + c0 + e
    const float* b = rel.tab + (ROWS_ARE_KEYS ? ncls - 1 - (ucol - urow0 + c0) : urow0 - ucol + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) cb[8 * g + e] = b[e];
  }
  return cb;
}

// work item of a workgroup: XCD-aware decode.  Hardware places workgroup b on XCD b % 8; the `per` consecutive items handled by one
// XCD are the row-block groups of the same (sequence, head), which then share their operand slab through that XCD's L2.
__device__ __forceinline__ bool decode_item(int ngroups, int nitems, int& grp, int& sh) {
  const int per = (nitems + 7) >> 3;
  const int w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (w >= nitems) return false;
  sh = w / ngroups; grp = w % ngroups;
  return true;
}

// loader: thread `lt` of the 256 loader threads moves 16 B of a pair of tiles per step: tile (lt >> 7), row (lt & 127) >> 2,
// source chunk lt & 3  ->  LDS position swz(row, chunk).
struct Loader {
  const char* src;          // global address of this thread's chunk of tile 0 (advance by TILE bytes per tile)
  int dst;                  // byte offset inside a ring slot
};

// every global load issued so far has landed (s_waitcnt vmcnt(0), lgkmcnt / expcnt untouched).  Placed in front of the tile loops: the
// loop-invariant operand fragments come from global loads, and without a visible wait the compiler re-waits for them -- vmcnt(0),
// i.e. for the tile prefetches too -- in front of the first MFMA of EVERY step.
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

}  // namespace

// launchers of the slab-resident kernels (attn2_slab.hip); each returns 1 when the shape is not eligible, else a C-ABI status
int attn2_slab_fwd(const ctclip_attn2::Params& p, hipStream_t stream);
int attn2_slab_bwd_dq(const ctclip_attn2::Params& p, hipStream_t stream);
int attn2_slab_bwd_dkv(const ctclip_attn2::Params& p, hipStream_t stream, int* nwg_out = nullptr);
int attn2_slab_bwd_dbias(const ctclip_attn2::Params& p, hipStream_t stream);
// the four-wave form of the one-pass backward (attn2_bwd2.hip): L = 576 (24 x 24 tokens) only
bool attn2_bwd2_eligible(int nseq, int H, int L, int gh, int gw, bool tab);
int attn2_bwd2_launch(const ctclip_attn2::Params& p, const ctclip_attn2::Bwd2Args& x, int nwg, hipStream_t stream);
