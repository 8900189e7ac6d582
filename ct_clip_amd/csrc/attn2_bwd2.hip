// ONE-PASS backward of the CTViT spatial cosine attention, second form (round 6): FOUR waves x 512 registers, dQ^T in REGISTERS.
// Same mathematics, operands, outputs and fixed-point table gradient as attn2_bwd1.hip (read its header first: attention.py:145-178 with the
// position bias of attention.py:257-276 differentiated, l2norm backward of attention.py:152-154 applied; dq / dk / dv row-major, both learned-scale
// gradients, the position-bias TABLE gradient) -- for the one shape the CTViT spatial layers have: L = 576 tokens = 24 x 24, d_head 32.
//
// Why a second form (profiles/r05_attn_pmc.md: bwd1_kernel 643 us, matrix pipe 11.9 % busy, waves parked 47 %, LDS bank conflicts 20.5 %).
// bwd1 keeps the f32 dQ^T accumulators of an item in LDS (72 KB) and every score tile read-modify-writes 4 KB of them behind a per-tile
// counter: a serial chain  LDS -> MFMA -> exp -> pack -> LDS -> transposing read -> MFMA -> LDS  per tile on two waves per SIMD, and an LDS
// that is full (no room to stage the next item, a class table at its natural stride 47: 2-way bank conflicts, the bias table in global memory).
//
// Decomposition here.  One persistent workgroup of FOUR waves (one per SIMD, the whole 512-entry register file each) walks a run of (sequence,
// head) items of one head.  The 18 x 18 score tiles of an item are split 2 x 2: wave w owns QUERY half qh = w & 1 (tiles 9 qh .. 9 qh + 8) and
// KEY half kh = w >> 1 (blocks 9 kh .. 9 kh + 8): 81 tiles per wave, no remainder.  Per key block a wave sweeps its nine query tiles with
//   * dK^T / dV^T of the block in registers (as bwd1),
//   * dQ^T of ALL NINE query tiles in registers (9 x 16 accumulator registers; the tile index is a template parameter: static register names)
//     -- no LDS accumulators, no tile counters, no ordering between waves inside the tile loop;
//   * dS transposed through a wave-private 2-KB LDS scratch (ds_read_b64_tr_b16), the K^T operand of the block likewise, once per block.
// The sweep is software-pipelined in the source: the S / dP products of tile T + 1 are issued in front of the exponentials of tile T, the dQ^T
// product of tile T - 1 behind them (its transposed operand has landed by then); the row operands are requested two tiles ahead.
// The two waves that share a key block (the two query halves) combine their dK^T / dV^T through an 8-KB LDS slot: one parks, the other adds,
// applies the l2norm backward and stores (alternating by block, a sequence-numbered flag: the pair never drifts more than one block apart).
// After the tile loop the two waves that share a query half combine their dQ^T partials through LDS (the slabs' space: one barrier), apply
// the l2norm backward of q and store.
//
// LDS (145 KB): Q~ slab 36 KB, dO slab 36 KB, the two per-query terms as bf16 triples (16 B per query: -delta and log2 K - lse2, A operands
// of two 3-term matrix products as in bwd1 but pre-split in the load phase: one ds_read_b128 and no VALU per tile), the BIAS TABLE of the head
// (f32, log2 domain) and the fixed-point CLASS TABLE, both at row stride 56 instead of 47 (56 = 24 mod 32: the 32 keys of a block gather from /
// scatter to 32 distinct banks), scratch 8 KB, park slots 32 KB.  All tile-dependent addresses are compile-time immediates on per-lane bases:
// the tile loop has no address arithmetic.
#include "attn2_common.h"

namespace {

constexpr int NW2 = 4, NTH2 = NW2 * 64;
constexpr int NKB2 = 18, HT = 9;             // score tiles per side; tiles per half
constexpr int L576 = 576, GW = 24;
constexpr int TS = 56, TROWS = 2 * GW - 1;   // table row stride / rows (47)
constexpr int TN = TROWS * TS;               // 2632 entries
constexpr int C0 = (GW - 1) * TS + (GW - 1); // class of offset (0, 0)
constexpr int NP2 = 9;                       // 16-byte pieces per thread and slab: 4 L = NP2 * NTH2
constexpr float MAGIC2 = 12582912.f;         // 1.5 * 2^23
constexpr uint32_t MAGIC2_BITS = 0x4B400000u;
constexpr int FIX_BITS2 = 21;

// LDS map (bytes)
constexpr int OFF_QS = 0;
constexpr int OFF_DOS = OFF_QS + L576 * 64;          //  36 864
constexpr int OFF_TRIP = OFF_DOS + L576 * 64;        //  73 728: [L][4] u32 = {nd hi|lo, nd ll, nl hi|lo, nl ll}
constexpr int OFF_TAB = OFF_TRIP + L576 * 16;        //  82 944: bias table, entry i at OFF_TAB + 4 i (8-byte aligned for even i)
constexpr int OFF_TAB1 = OFF_TAB + TN * 4 + 100;     //  93 572: a second copy, entry i at OFF_TAB1 + 4 i (8-byte aligned for ODD i): a lane's eight
                                                     //          consecutive entries start at an index of either parity -- four ds_read_b64 from the copy that aligns
                                                     //          them.  Neighbouring lanes (index i, i - 1) use different copies: the copies are 33 (mod 64) dwords
                                                     //          apart so that the odd lanes' banks are the even lanes' + 32 (PMC: 24.5 % conflict cycles at + 9)
constexpr int OFF_DTAB = OFF_TAB1 + TN * 4 + 12;     // 104 112
constexpr int OFF_SCR = OFF_DTAB + TN * 4;           // 114 640: [NW2][2048]
constexpr int OFF_PARK = OFF_SCR + NW2 * 2048;       // 122 832: [NW2][8192] slot of parking wave w
constexpr int OFF_MISC = OFF_PARK + NW2 * 8192;      // 155 600
// misc (floats): [0,4) max |dO|^2 per wave, [4,8) max |v|^2, [8,12) park sequence numbers (int), [32,96) k_scale | 1 / k_scale,
// [96,352) scale-gradient sums [2][NW2][32]
constexpr int MISC_FLOATS = 352;
constexpr int SHM2 = OFF_MISC + MISC_FLOATS * 4;     // 157 008 of 163 840
static_assert(OFF_TAB % 8 == 0 && OFF_TAB1 % 8 == 4 && ((OFF_TAB1 - OFF_TAB) / 4) % 64 == 33 && OFF_DTAB % 16 == 0 && OFF_SCR % 16 == 0 && OFF_MISC % 16 == 0 && SHM2 <= 160 * 1024, "LDS map");
constexpr int OFF_RED = OFF_QS;                      // dQ^T exchange [18][4096] over the two slabs (after the tile loop)

__host__ __device__ constexpr int U56(int t) { return (t / GW) * TS + t % GW; }

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define LDS_AS(T, ptr) ((__attribute__((address_space(3))) T*)(ptr))
#define GLB(T, ptr) ((__attribute__((address_space(1))) T*)(ptr))

// transposed fragment (see lds_cols of attn2_common.h) through the builtin: the compiler sees the loads, counts their waits and schedules
// around them (the asm form needs a full lgkmcnt(0) in front of every consumer)
__device__ __forceinline__ Frag cols2(const char* tile, const TrOff& tr) {
  const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, tile + tr.o[0]));
  const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, tile + tr.o[1]));
  const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, tile + 1024 + tr.o[0]));
  const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, tile + 1024 + tr.o[1]));
  Frag f;
  f.v[0] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  f.v[1] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  return f;
}

template <class T>
__device__ __forceinline__ T* uni2(T* ptr) {
  const uint64_t v = (uint64_t)ptr;
  return (T*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
__device__ __forceinline__ int uni2(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni2(int64_t v) {
  return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
__device__ __forceinline__ float uni2(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }

__device__ __forceinline__ void unpack8v(const u32x4& a, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(a[i] << 16); v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u); }
}
__device__ __forceinline__ void gstore8(bf16_t* ptr, const float (&v)[8]) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *GLB(u32x4, ptr) = a;
}

// Compile-time ablation mask (tools/build_variant.py attn2_bwd2.hip:BWD2_ABL=<mask>; never set in the product build -- results are WRONG, timing
// only): 1 = no class-table atomics, 2 = no dQ^T product and no dS transposition, 4 = no exponential, 16 = no dV / dK products and no transposing
// reads of Q~ / dO, 32 = no bias gather
#ifndef BWD2_ABL
#define BWD2_ABL 0
#endif
// BWD2_TSTAMPS (variant builds only): shader-clock stamps of wave 0 at the phase boundaries of the nine tiles of key block 1 of item 0 (every stamp
// drains the LDS queue: an SMEM read is counted on lgkmcnt -- the durations are upper bounds of the undisturbed ones)
#ifndef BWD2_TSTAMPS
#define BWD2_TSTAMPS 0
#endif

struct Steps2 {
  const bf16_t *k, *v, *q;                   // the item's K^ / V / Q~ slabs (head-planar, global)
  const float *kinv, *qinv;                  // inverse norms at (row 0 of the item, this head); row stride H
  const float* q_scale;
  bf16_t *dk, *dv, *dq; int64_t ldk, ldv, ldq;      // row 0 at this head of the three outputs
  float invK, c; int H, seq0;                // seq0: park sequence number base of this item
  unsigned long long* stamps;                // profiling aid: this item's 16 phase clocks + 32 block clocks of wave 0 (workgroup 0), or null
  unsigned long long* bstamps;
  // the NEXT item's load-phase operands: one dword per 128-byte line is requested per thread and key block (L2 prefetch: the load phase of the
  // next item then reads L2, not HBM -- 10.5 us of exposed latency per item without)
  const bf16_t *nq, *nv, *nk, *ndout, *no; const float* nlse; int64_t lddo, ldo;
};

// per-lane constants of the tile loop
struct Lane2 {
  int ra, rb;                                // row fragment: swz(pi32(c), half), swz(pi32(c), 2 + half) (+ the query half's tile offset)
  TrOff tr;                                  // transposed fragment offsets (+ the query half's tile offset for the slabs: trq)
  TrOff trq;
  int trip;                                  // (9 qh 32 + pi32(c)) * 16
  bf16x8 onesA, onesB;                       // B operands of the two 3-term products (ones in contraction slots 0-2 / 4-6 of half 0)
};

struct Ops2 {                                // row operands of one tile: A fragments of S and dP, the per-query triples, the bias (-> S accumulator)
  Frag qf, dof;
  bf16x8 trp;
  f32x16 cb;
};

template <int T, bool TAB>
__device__ __forceinline__ void req_rows(const char* dyn, const Lane2& ln, const float* tbA, const float* tbB, Ops2& o) {
  o.qf.v[0] = *reinterpret_cast<const bf16x8*>(dyn + OFF_QS + T * TILE + ln.ra);
  o.qf.v[1] = *reinterpret_cast<const bf16x8*>(dyn + OFF_QS + T * TILE + ln.rb);
  o.dof.v[0] = *reinterpret_cast<const bf16x8*>(dyn + OFF_DOS + T * TILE + ln.ra);
  o.dof.v[1] = *reinterpret_cast<const bf16x8*>(dyn + OFF_DOS + T * TILE + ln.rb);
  o.trp = *reinterpret_cast<const bf16x8*>(dyn + OFF_TRIP + T * 512 + ln.trip);
  if (TAB && !(BWD2_ABL & 32)) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int q0 = 32 * T + 16 * gq;                           // (relative to the query half: 288 = 12 image rows, folded into the bases)
      const f32x2* b = reinterpret_cast<const f32x2*>(((q0 % GW) == 16 ? tbB : tbA) + U56(q0));      // 8-byte aligned (the copy is chosen by the index parity)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) { const f32x2 v2 = b[e2]; o.cb[8 * gq + 2 * e2] = v2[0]; o.cb[8 * gq + 2 * e2 + 1] = v2[1]; }
    }
  } else {
    const float tv = TAB ? 0.f : tbA[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) o.cb[r] = tv;
  }
}

// The tile steps of one item for this wave (81 tiles), the exchange of the dK^T / dV^T and dQ^T partials and all three un-preps.  A call (its own
// register allocation: 512 registers; the load phase of the caller has twenty-seven 16-byte loads in flight per thread).
template <bool TAB, bool DTAB>
__device__ __noinline__ void bwd2_steps(Steps2 a_) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  Steps2 a = a_;
  a.k = uni2(a.k); a.v = uni2(a.v); a.q = uni2(a.q); a.kinv = uni2(a.kinv); a.qinv = uni2(a.qinv); a.q_scale = uni2(a.q_scale);
  a.dk = uni2(a.dk); a.dv = uni2(a.dv); a.dq = uni2(a.dq); a.ldk = uni2(a.ldk); a.ldv = uni2(a.ldv); a.ldq = uni2(a.ldq);
  a.invK = uni2(a.invK); a.c = uni2(a.c); a.H = uni2(a.H); a.seq0 = uni2(a.seq0); a.stamps = uni2(a.stamps); a.bstamps = uni2(a.bstamps);
  a.nq = uni2(a.nq); a.nv = uni2(a.nv); a.nk = uni2(a.nk); a.ndout = uni2(a.ndout); a.no = uni2(a.no); a.nlse = uni2(a.nlse); a.lddo = uni2(a.lddo); a.ldo = uni2(a.ldo);
#define BWD2_ST(i) do { if (a.stamps && threadIdx.x == 0) *GLB(unsigned long long, a.stamps + (i)) = wall_clock64(); } while (0)
#define BWD2_BST(i) do { if (a.bstamps && threadIdx.x == 0) *GLB(unsigned long long, a.bstamps + (i)) = wall_clock64(); } while (0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qh = wave & 1, kh = wave >> 1;
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  float* misc = reinterpret_cast<float*>(dyn + OFF_MISC);
  const float* ksr = misc + 32;
  float* sred = misc + 96;
  volatile int* pseq = reinterpret_cast<volatile int*>(misc + 8);
  char* scr = dyn + OFF_SCR + wave * 2048;

  Lane2 ln;
  ln.ra = qh * HT * TILE + swz(ar, half);
  ln.rb = qh * HT * TILE + swz(ar, 2 + half);
  ln.tr = tr_offsets(lane);
  ln.trq = ln.tr; ln.trq.o[0] += qh * HT * TILE; ln.trq.o[1] += qh * HT * TILE;
  ln.trip = (qh * HT * 32 + ar) * 16;
  ln.onesA = half ? bf16x8{0, 0, 0, 0, 0, 0, 0, 0} : bf16x8{0x3F80, 0x3F80, 0x3F80, 0, 0, 0, 0, 0};
  ln.onesB = half ? bf16x8{0, 0, 0, 0, 0, 0, 0, 0} : bf16x8{0, 0, 0, 0, 0x3F80, 0x3F80, 0x3F80, 0};

  f32x16 dq[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[j][r] = 0.f;
  float ksacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ksacc[i] = 0.f;

  Frag kn, vn; float ikn;
  auto load_kv = [&](int jb) {
    const int64_t o2 = (int64_t)(jb * 32 + c) * D;
    kn.v[0] = *GLB(const bf16x8, a.k + o2 + 8 * half); kn.v[1] = *GLB(const bf16x8, a.k + o2 + 16 + 8 * half);
    vn.v[0] = *GLB(const bf16x8, a.v + o2 + 8 * half); vn.v[1] = *GLB(const bf16x8, a.v + o2 + 16 + 8 * half);
    ikn = *GLB(const float, a.kinv + (int64_t)(jb * 32 + c) * a.H);
  };
  load_kv(kh * HT);

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  uint32_t touch = 0u;
  // this lane's key: class offsets.  Bias / class entry of (query q, key k) = U56(q) - U56(k) + C0; for the 8 consecutive queries of register
  // group gq of tile T: U56(32 T + 16 gq) + half * (8, or 8 + 32 when the eight sit behind an image-row end) + e  -- immediates on two bases
  const float *tbA, *tbB;
  uint32_t *dbA, *dbB;
  auto block_bases = [&](int kb_) {
    const int tk = kb_ * 32 + c;
    const int trow = (tk * 2731) >> 16;                          // tk / 24 (tk < 576)
    const int ucol = trow * TS + (tk - trow * GW);
    const int cbase = C0 - ucol + qh * (HT * 32 / GW) * TS;      // (+ the query half: 288 tokens = 12 image rows)
    const float* tb0 = reinterpret_cast<const float*>(dyn + ((cbase & 1) ? OFF_TAB1 : OFF_TAB));     // (all of a lane's gather indices have the parity of cbase)
    tbA = tb0 + (TAB ? cbase + half * 8 : 0);
    tbB = tb0 + (TAB ? cbase + half * 40 : 0);
    dbA = reinterpret_cast<uint32_t*>(dyn + OFF_DTAB) + cbase + half * 8;
    dbB = reinterpret_cast<uint32_t*>(dyn + OFF_DTAB) + cbase + half * 40;
  };
  Ops2 o0, o1;                                                   // row operands of the next two tiles (alternating)
  f32x16 sc, dp;                                                 // S and dP of the CURRENT tile (computed one tile ahead)
  auto m1 = [&](Ops2& o, const Frag& kf_, const Frag& vf_, f32x16& s_, f32x16& d_) {      // S = Q~ K^^T + bias + (log2 K - lse2), dP = dO V^T - delta
    f32x16 cd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.trp, ln.onesA, zero16, 0, 0, 0);
    f32x16 cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.trp, ln.onesB, o.cb, 0, 0, 0);
    cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.qf.v[0], kf_.v[0], cs, 0, 0, 0);
    cd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.dof.v[0], vf_.v[0], cd, 0, 0, 0);
    s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.qf.v[1], kf_.v[1], cs, 0, 0, 0);
    d_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.dof.v[1], vf_.v[1], cd, 0, 0, 0);
  };
  // The PROLOGUE of a key block -- row operands of its first two tiles, S / dP of its first tile -- is issued at the END of the previous block, in
  // front of that block's park / merge code (two LDS round trips and six matrix instructions that nothing else covered: ~0.2 us per block)
  auto block_prologue = [&](int kb_) {
    block_bases(kb_);
    req_rows<0, TAB>(dyn, ln, tbA, tbB, o0);
    req_rows<1, TAB>(dyn, ln, tbA, tbB, o1);
    m1(o0, kn, vn, sc, dp);
  };
  block_prologue(kh * HT);
  for (int kbi = 0; kbi < HT; ++kbi) {
    const int kb = kh * HT + kbi;
    BWD2_BST(3 * kbi);
    const Frag kf = kn, vf = vn;
    const float ik = ikn;
    asm volatile("" :: "v"(touch));                              // (the previous block's touch: its destination register stays reserved until it has landed)
    if (kbi + 1 < HT) load_kv(kb + 1);                           // a block ahead (consumed at the top of the next block)
    {   // L2 touch of the next item: line kbi * 256 + tid of  Q~ | V | K^ (288 lines each) | dout rows | o rows (576 each: 64 B of a row) | lse2 (18)
      int li = kbi * NTH2 + tid;
      li = li < 2034 ? li : 2033;
      const int sg = li < 288 ? 0 : (li < 576 ? 1 : (li < 864 ? 2 : (li < 1440 ? 3 : (li < 2016 ? 4 : 5))));
      const int r = li - (sg == 0 ? 0 : (sg == 1 ? 288 : (sg == 2 ? 576 : (sg == 3 ? 864 : (sg == 4 ? 1440 : 2016)))));
      const char* base = reinterpret_cast<const char*>(sg == 0 ? a.nq : (sg == 1 ? a.nv : (sg == 2 ? a.nk : (sg == 3 ? a.ndout : a.no))));
      if (sg == 5) base = reinterpret_cast<const char*>(a.nlse);
      const int64_t stride = sg == 3 ? a.lddo * 2 : (sg == 4 ? a.ldo * 2 : 128);
      touch = *GLB(const uint32_t, base + (int64_t)r * stride);
    }
    // K^T of the block (A operand of dQ^T = K^T dS^T): through the wave's scratch
    Frag ktf;
    if (!(BWD2_ABL & 2)) {
      *reinterpret_cast<bf16x8*>(scr + swz(c, half)) = kf.v[0];
      *reinterpret_cast<bf16x8*>(scr + swz(c, 2 + half)) = kf.v[1];
      ktf = cols2(scr, ln.tr);
    }
    f32x16 dkacc = zero16, dvacc = zero16;

    // ---- the sweep over the nine query tiles of this wave, software-pipelined (see the file header); o0, o1, sc, dp were prepared by block_prologue

#define BWD2_SB() __builtin_amdgcn_sched_barrier(0)
    // Three phases per tile, pinned by scheduling barriers.  Why: ONE wave per SIMD issues an instruction every ~5 cycles at best
    // (tools/ubench/issue_rates.hip: v_fma 5.06, v_exp 8.75 cycles; a matrix instruction occupies its pipe for 32), it is in-order, and left
    // alone the machine scheduler sinks every LDS read to its consumer to save registers: a full LDS round trip six times per tile.
    //   A  issue the transposing reads this tile's dV / dK / dQ^T products need (they land under phase B);
    //   B  S / dP of tile T + 1 on the matrix pipe UNDER this tile's exponentials, products and class-table atomics; row operands of tile T + 2 requested;
    //   C  packs, then dV^T, dK^T of tile T and dQ^T of tile T - 1; dS of tile T into the scratch.
    // (Prescribing the instruction mix of B and C with sched_group_barrier -- one matrix instruction per six vector and five LDS instructions --
    // measured 3-5 % SLOWER than leaving the order inside a phase to the scheduler: profiles/r06_attn_bwd2.md.)
#define BWD2_TST(i) do { if (BWD2_TSTAMPS && a.bstamps && kbi == 1 && a.seq0 == 0 && threadIdx.x == 0) *GLB(unsigned long long, a.bstamps + 192 + (i)) = __builtin_readcyclecounter(); } while (0)
#ifndef BWD2_DPP_SUMS
#define BWD2_DPP_SUMS 0           // 1: the un-preps' lane sums by DPP / permlane swaps instead of __shfl_xor (A/B: profiles/r06_ab_experiments.md)
#endif
#ifndef BWD2_DQ_IN_B
#define BWD2_DQ_IN_B 0             // 1: the two dQ^T matrix instructions of tile T - 1 are issued in phase B of tile T (8 + 4 instead of 6 + 6 per phase): 480-491 against 444-445 us
#endif
#ifndef BWD2_SPLIT_ATOMICS
#define BWD2_SPLIT_ATOMICS 0       // 1: the 16 class-table atomics of a tile are issued 8 in phase B, 8 in phase C (measured 456 against 449 us); 0: all 16 in phase B
#endif
#define BWD2_ATOMICS(T, PART)                                                                                                      \
      if (DTAB && !(BWD2_ABL & 1)) {                                                                                               \
        _Pragma("unroll") for (int gq = (PART); gq < (BWD2_SPLIT_ATOMICS ? (PART) + 1 : ((PART) == 0 ? 2 : 0)); ++gq) {             \
          const int q0 = 32 * (T) + 16 * gq;                                                                                       \
          uint32_t* b = ((q0 % GW) == 16 ? dbB : dbA) + U56(q0);                                                                   \
          _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                                            \
            (void)__hip_atomic_fetch_add(b + e, __float_as_uint(fx[4 * gq + (e >> 1)][e & 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        }                                                                                                                          \
      }
#define BWD2_TILE(T, OCUR, ONEXT)                                                                                                  \
    {                                                                                                                              \
      BWD2_TST(3 * (T));                                                                                                           \
      Frag dotf, qtf, dstf;                                                                                                        \
      if (!(BWD2_ABL & 16)) { dotf = cols2(dyn + OFF_DOS + (T) * TILE, ln.trq); qtf = cols2(dyn + OFF_QS + (T) * TILE, ln.trq); }   \
      if (!(BWD2_ABL & 2) && (T) > 0) dstf = cols2(scr, ln.tr);                                                                    \
      BWD2_SB();                                                                                                                   \
      f32x16 sc2, dp2;                                                                                                             \
      if ((T) + 1 < HT) m1(ONEXT, kf, vf, sc2, dp2);                                                                               \
      float pr[16], ds[16];                                                                                                        \
      f32x2 fx[8];                                                 /* packed f32 arithmetic: the single wave of a SIMD is ISSUE-bound */ \
      _Pragma("unroll") for (int r2 = 0; r2 < 8; ++r2) {                                                                           \
        const f32x2 p2 = {(BWD2_ABL & 4) ? sc[2 * r2] : __builtin_amdgcn_exp2f(sc[2 * r2]),                                        \
                          (BWD2_ABL & 4) ? sc[2 * r2 + 1] : __builtin_amdgcn_exp2f(sc[2 * r2 + 1])};                                \
        const f32x2 d2 = {dp[2 * r2], dp[2 * r2 + 1]};                                                                             \
        const f32x2 s2 = p2 * d2;                                                                                                  \
        fx[r2] = __builtin_elementwise_fma(p2, d2, f32x2{MAGIC2, MAGIC2});                                                         \
        pr[2 * r2] = p2[0]; pr[2 * r2 + 1] = p2[1]; ds[2 * r2] = s2[0]; ds[2 * r2 + 1] = s2[1];                                     \
      }                                                                                                                            \
      BWD2_ATOMICS(T, 0)                                           /* first half of the class-table scatter here, second half in phase C */ \
      if ((T) + 2 < HT) req_rows<((T) + 2 < HT ? (T) + 2 : 0), TAB>(dyn, ln, tbA, tbB, OCUR);                                      \
      if (BWD2_DQ_IN_B && !(BWD2_ABL & 2) && (T) > 0) dq[(T) > 0 ? (T) - 1 : 0] = mma(dq[(T) > 0 ? (T) - 1 : 0], ktf, dstf);       \
      BWD2_SB();                                                                                                                   \
      BWD2_TST(3 * (T) + 1);                                                                                                       \
      const Frag pf = pack(pr), dsf = pack(ds);                                                                                    \
      BWD2_ATOMICS(T, 1)                                                                                                           \
      if (!(BWD2_ABL & 16)) {                                                                                                      \
        dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf.v[0], pf.v[0], dvacc, 0, 0, 0);                                       \
        dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf.v[0], dsf.v[0], dkacc, 0, 0, 0);                                       \
        dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf.v[1], pf.v[1], dvacc, 0, 0, 0);                                       \
        dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf.v[1], dsf.v[1], dkacc, 0, 0, 0);                                       \
      } else {                                                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { dvacc[r] += pr[r]; dkacc[r] += ds[r]; }                                   \
      }                                                                                                                            \
      if (!(BWD2_ABL & 2)) {                                                                                                       \
        if (!BWD2_DQ_IN_B && (T) > 0) dq[(T) > 0 ? (T) - 1 : 0] = mma(dq[(T) > 0 ? (T) - 1 : 0], ktf, dstf);                        \
        *reinterpret_cast<bf16x8*>(scr + swz(c, half)) = dsf.v[0];                                                                 \
        *reinterpret_cast<bf16x8*>(scr + swz(c, 2 + half)) = dsf.v[1];                                                             \
      }                                                                                                                            \
      BWD2_SB();                                                                                                                   \
      BWD2_TST(3 * (T) + 2);                                                                                                       \
      if ((T) + 1 < HT) { sc = sc2; dp = dp2; }                                                                                    \
    }
    BWD2_TILE(0, o0, o1)
    BWD2_TILE(1, o1, o0)
    BWD2_TILE(2, o0, o1)
    BWD2_TILE(3, o1, o0)
    BWD2_TILE(4, o0, o1)
    BWD2_TILE(5, o1, o0)
    BWD2_TILE(6, o0, o1)
    BWD2_TILE(7, o1, o0)
    BWD2_TILE(8, o0, o1)
#undef BWD2_TILE
    if (!(BWD2_ABL & 2)) {                                       // dQ^T of the last tile
      const Frag dstf = cols2(scr, ln.tr);
      dq[HT - 1] = mma(dq[HT - 1], ktf, dstf);
    }

    if (kbi + 1 < HT) block_prologue(kb + 1);                    // (kn / vn = the next block's rows, requested at the top of this block)
    BWD2_BST(3 * kbi + 1);
    // ---- end of the key block: the two query halves combine.  Blocks alternate: (kbi + kh) odd -> the qh = 1 wave parks and the qh = 0 wave
    // merges, even -> the other way round (each wave un-preps 4 or 5 of its 9 blocks)
    const int merger_qh = (kbi + kh) & 1;
    const int seqno = a.seq0 + kbi + 1;
    if (qh != merger_qh) {
      float* pk = reinterpret_cast<float*>(dyn + OFF_PARK + wave * 8192) + lane * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<f32x4*>(pk + j * 256) = f32x4{dkacc[4 * j], dkacc[4 * j + 1], dkacc[4 * j + 2], dkacc[4 * j + 3]};
        *reinterpret_cast<f32x4*>(pk + 1024 + j * 256) = f32x4{dvacc[4 * j], dvacc[4 * j + 1], dvacc[4 * j + 2], dvacc[4 * j + 3]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (LDS executes a wave's operations in order: the slot is written when the number is seen)
      if (lane == 0) pseq[wave] = seqno;
    } else {
      const int partner = wave ^ 1;
      while (__builtin_amdgcn_readfirstlane(pseq[partner]) < seqno) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const float* pk = reinterpret_cast<const float*>(dyn + OFF_PARK + partner * 8192) + lane * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(pk + j * 256), w = *reinterpret_cast<const f32x4*>(pk + 1024 + j * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) { dkacc[4 * j + e] += u[e]; dvacc[4 * j + e] += w[e]; }
      }
      // un-prep in place (as bwd1): u = k^ / k_scale, g = dk^ k_scale, dk = kinv (g - u (u . g)); dscale += dk^ u
      const int row = kb * 32 + c;
      bf16_t* dV = a.dv + (int64_t)row * a.ldv;
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        float b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = dvacc[8 * gq + e] * a.invK;
        gstore8(dV + 16 * gq + 8 * half, b8);
      }
      const float kmul = LN2 * a.invK;
      const u32x4 kw0 = __builtin_bit_cast(u32x4, kf.v[0]), kw1 = __builtin_bit_cast(u32x4, kf.v[1]);
      float part[2] = {0.f, 0.f};
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const float* sp = ksr + 16 * gq + 8 * half;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(sp + 32), r1 = *reinterpret_cast<const f32x4*>(sp + 36);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = 8 * gq + e;
          const uint32_t kwd = gq ? kw1[e >> 1] : kw0[e >> 1];
          const float kx = (e & 1) ? __uint_as_float(kwd & 0xffff0000u) : __uint_as_float(kwd << 16);
          const float ks = e < 4 ? s0[e & 3] : s1[e & 3], rk = e < 4 ? r0[e & 3] : r1[e & 3];
          const float gk0 = bf2f(f2bf(dkacc[i] * kmul));
          const float uk = kx * rk;
          ksacc[i] += gk0 * uk;
          part[gq] += uk * (gk0 * ks);
          dkacc[i] = gk0 * ks;
        }
      }
      const float dot = (half_sum(part[0])) + (half_sum(part[1]));
      bf16_t* dK = a.dk + (int64_t)row * a.ldk;
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const float* sp = ksr + 16 * gq + 8 * half;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(sp + 32), r1 = *reinterpret_cast<const f32x4*>(sp + 36);
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t kwd = gq ? kw1[e >> 1] : kw0[e >> 1];
          const float kx = (e & 1) ? __uint_as_float(kwd & 0xffff0000u) : __uint_as_float(kwd << 16);
          const float rk = e < 4 ? r0[e & 3] : r1[e & 3];
          a8[e] = ik * (dkacc[8 * gq + e] - (kx * rk) * dot);
        }
        gstore8(dK + 16 * gq + 8 * half, a8);
      }
    }
    BWD2_BST(3 * kbi + 2);
  }
  asm volatile("" :: "v"(touch));
  BWD2_ST(3);

  // this item's k_scale-gradient sums: the 32 lanes of a half by an xor tree, then added to the wave's row of the LDS accumulator
#pragma unroll
  for (int i = 0; i < 16; ++i)
#if BWD2_DPP_SUMS
    ksacc[i] = half32_sum(ksacc[i]);
#else
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) ksacc[i] += __shfl_xor(ksacc[i], o, 64);
#endif
  if (c == 0) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int e = 0; e < 8; ++e) sred[wave * 32 + 16 * gq + 8 * half + e] += ksacc[8 * gq + e];
  }

  // ---- dQ: the two key halves combine through LDS (the slabs are dead: everybody is past the tile loop after this barrier).  Of the nine
  // tiles of a query half the kh = 0 wave finishes the even ones, the kh = 1 wave the odd ones: each wave hands over the tiles it does not finish.
  // The global operands of the q un-prep (q~ rows, inverse norms, q_scale) are requested HERE, in front of the two barriers, not behind them.
  u32x4 pqa[5], pqb[5];
  float piq[5];
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const int j = kh + 2 * m < HT ? kh + 2 * m : kh;             // (kh = 1 finishes four tiles: the fifth slot re-reads its first)
    const int qrow = (qh * HT + j) * 32 + ar;
    const bf16_t* qp = a.q + (int64_t)qrow * D;
    pqa[m] = *GLB(const u32x4, qp + 8 * half); pqb[m] = *GLB(const u32x4, qp + 16 + 8 * half);
    piq[m] = *GLB(const float, a.qinv + (int64_t)qrow * a.H);
  }
  float qsv[16], rqv[16], qsacc[16];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float qs_ = *GLB(const float, a.q_scale + 16 * gq + 8 * half + e);
      const float qc = qs_ * a.c;
      qsv[8 * gq + e] = qs_;
      rqv[8 * gq + e] = fabsf(qc) > 1e-30f ? 1.f / qc : 0.f;
      qsacc[8 * gq + e] = 0.f;
    }
  __syncthreads();
  BWD2_ST(4);
  char* red = dyn + OFF_RED;
#pragma unroll
  for (int j = 0; j < HT; ++j) {
    if ((j & 1) != kh) {                                         // (wave-uniform)
      float* dst = reinterpret_cast<float*>(red + (qh * HT + j) * 4096) + lane * 4;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<f32x4*>(dst + q4 * 256) = f32x4{dq[j][4 * q4], dq[j][4 * q4 + 1], dq[j][4 * q4 + 2], dq[j][4 * q4 + 3]};
    }
  }
  __syncthreads();
  BWD2_ST(5);
  // un-prep of q (attn_unprep_kernel of attn2.hip / bwd1_unprep_q): u = q~ / (q_scale c), g = dq^ q_scale, dq = qinv (g - u (u . g)), dscale += dq^ u on
  // the bf16-rounded dq^.  Lane n of the transposed product holds query pi32(n & 31), registers = head dims 16 gq + 8 half + e.
  const float scq = a.c * LN2 * a.invK;
#pragma unroll
  for (int j = 0; j < HT; ++j) {
    if ((j & 1) == kh) {
      const int tq = qh * HT + j;
      const float* src = reinterpret_cast<const float*>(red + tq * 4096) + lane * 4;
      const int qrow = tq * 32 + ar;
      const float iq = piq[j >> 1];                                // (j = kh + 2 m: slot m = j >> 1 for either parity)
      const u32x4 qa = pqa[j >> 1], qb = pqb[j >> 1];
      float qx[16], g16[16];
      unpack8v(qa, qx); unpack8v(qb, qx + 8);
      float part[2] = {0.f, 0.f};
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(src + q4 * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * q4 + e;
          const float gq0 = bf2f(f2bf((dq[j][i] + o4[e]) * scq));
          const float uq = qx[i] * rqv[i];
          qsacc[i] += gq0 * uq;
          const float gv = gq0 * qsv[i];
          part[i >> 3] += uq * gv;
          qx[i] = uq; g16[i] = gv;
        }
      }
      const float dot = (half_sum(part[0])) + (half_sum(part[1]));
      bf16_t* dQ = a.dq + (int64_t)qrow * a.ldq;
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[e] = iq * (g16[8 * gq + e] - qx[8 * gq + e] * dot);
        gstore8(dQ + 16 * gq + 8 * half, a8);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
#if BWD2_DPP_SUMS
    qsacc[i] = half32_sum(qsacc[i]);
#else
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) qsacc[i] += __shfl_xor(qsacc[i], o, 64);
#endif
  if (c == 0) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int e = 0; e < 8; ++e) sred[NW2 * 32 + wave * 32 + 16 * gq + 8 * half + e] += qsacc[8 * gq + e];
  }
  BWD2_ST(6);
}

struct Load2 {
  const bf16_t *qsl, *vsl, *dsl, *osl; const float* lsl;       // the item's Q~ / V slabs (head-planar), dout / o rows at this head, lse2
  int64_t lddo, ldo;
  unsigned long long* stp;
};

// Load phase of one item: Q~ and dO slabs -> LDS (plain copies), -delta = -sum_d dO O and log2 K - lse2 per query as bf16 triples, and K -- the
// fixed-point scale of the item, a power of two from the rigorous bound |dS| <= 2 max|dO_q| max|v_k|.  Returns 1 / K.  A call of its own: inlined,
// its 36 + 9 loads per thread shared the kernel function's register allocation with everything that lives across the 512-register tile call,
// and in some builds the allocator spilled the loads (load phase 20 us instead of 6).
__device__ __noinline__ float bwd2_load(Load2 a_) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  Load2 a = a_;
  a.qsl = uni2(a.qsl); a.vsl = uni2(a.vsl); a.dsl = uni2(a.dsl); a.osl = uni2(a.osl); a.lsl = uni2(a.lsl); a.lddo = uni2(a.lddo); a.ldo = uni2(a.ldo);
  a.stp = uni2(a.stp);
  const bf16_t *qsl = a.qsl, *vsl = a.vsl, *dsl = a.dsl, *osl = a.osl; const float* lsl = a.lsl;
  char* qs = dyn + OFF_QS;
  char* dos = dyn + OFF_DOS;
  uint32_t* trip = reinterpret_cast<uint32_t*>(dyn + OFF_TRIP);
  float* misc = reinterpret_cast<float*>(dyn + OFF_MISC);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float invK = 1.f;
    auto split3 = [](float v, uint32_t& w0, uint32_t& w1) {      // 24 mantissa bits as three bf16 terms
      const uint32_t hi = pack2bf(v, 0.f) & 0xffffu;
      const float r1 = v - __uint_as_float(hi << 16);
      const uint32_t lo = pack2bf(r1, 0.f) & 0xffffu;
      const float r2 = r1 - __uint_as_float(lo << 16);
      w0 = hi | (lo << 16); w1 = pack2bf(r2, 0.f) & 0xffffu;
    };
    float mxd = 0.f, mxv = 0.f;
    float ols[NP2];
    // three batches of three pieces (twelve 16-byte loads in flight per thread).  All 36 in flight at once measured SLOWER (6.8-8.8 us against
    // 4.6-5.2: `profiles/r06_attn_bwd2.md`); inlined into the kernel, whose allocator has what the 512-register tile function leaves, they were
    // spilled in some builds (20 us) -- hence a function of its own.
#pragma unroll
    for (int k0 = 0; k0 < NP2; k0 += 3) {
      u32x4 oq[3], ov_[3], od[3], oo[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int pc = (k0 + k) * NTH2 + tid, row = pc >> 2, ch = pc & 3;
        oq[k] = *reinterpret_cast<const u32x4*>(qsl + row * D + ch * 8);
        ov_[k] = *reinterpret_cast<const u32x4*>(vsl + row * D + ch * 8);
        od[k] = *reinterpret_cast<const u32x4*>(dsl + (int64_t)row * a.lddo + ch * 8);
        oo[k] = *reinterpret_cast<const u32x4*>(osl + (int64_t)row * a.ldo + ch * 8);
        ols[k0 + k] = lsl[row];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int pc = (k0 + k) * NTH2 + tid, row = pc >> 2, ch = pc & 3;
        *reinterpret_cast<u32x4*>(qs + (row >> 5) * TILE + swz(row & 31, ch)) = oq[k];
        *reinterpret_cast<u32x4*>(dos + (row >> 5) * TILE + swz(row & 31, ch)) = od[k];
        float x8[8], b[8], v8[8];
        unpack8v(od[k], x8); unpack8v(oo[k], b); unpack8v(ov_[k], v8);
        float ds = 0.f, dn = 0.f, vnn = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ds += x8[e] * b[e]; dn += x8[e] * x8[e]; vnn += v8[e] * v8[e]; }
#if BWD2_DPP_SUMS
        ds = quad_sum(ds); dn = quad_sum(dn); vnn = quad_sum(vnn);
#else
        ds += __shfl_xor(ds, 1, 64); ds += __shfl_xor(ds, 2, 64);
        dn += __shfl_xor(dn, 1, 64); dn += __shfl_xor(dn, 2, 64);
        vnn += __shfl_xor(vnn, 1, 64); vnn += __shfl_xor(vnn, 2, 64);
#endif
        uint32_t w0, w1;
        split3(-ds, w0, w1);
        if (ch < 2) trip[row * 4 + ch] = ch ? w1 : w0;              // (the four chunk threads of a row hold the same value)
        mxd = fmaxf(mxd, dn); mxv = fmaxf(mxv, vnn);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mxd = wave_max(mxd); mxv = wave_max(mxv);
    if (lane == 0) { misc[wave] = mxd; misc[4 + wave] = mxv; }
    __syncthreads();
    if (a.stp && tid == 0) a.stp[1] = wall_clock64();
    float bd = 0.f, bv = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < NW2; ++w4) { bd = fmaxf(bd, misc[w4]); bv = fmaxf(bv, misc[4 + w4]); }
    const float B = 2.f * sqrtf(bd) * sqrtf(bv);                  // |dS| <= P 2 |dO_q| |v_k| with the probability P <= 1
    int kk = 0;
    if (B > 0.f && B < 3.0e38f) {
      int e; (void)frexpf(B, &e);
      kk = FIX_BITS2 - e; kk = kk > 100 ? 100 : (kk < -100 ? -100 : kk);
      invK = ldexpf(1.f, -kk);
    } else if (!(B < 3.0e38f)) {
      invK = __builtin_nanf("");
    }
    const float lgK = (float)kk;
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
      const int pc = k * NTH2 + tid, row = pc >> 2, ch = pc & 3;
      uint32_t w0, w1;
      split3(lgK - ols[k], w0, w1);
      if (ch >= 2) trip[row * 4 + ch] = (ch & 1) ? w1 : w0;
    }
  return invK;
}

template <bool TAB, bool DTAB>
__global__ __launch_bounds__(NTH2) void bwd2_kernel(Params p, ctclip_attn2::Bwd2Args x) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  char* qs = dyn + OFF_QS;
  char* dos = dyn + OFF_DOS;
  uint32_t* trip = reinterpret_cast<uint32_t*>(dyn + OFF_TRIP);
  float* tabl = reinterpret_cast<float*>(dyn + OFF_TAB);
  uint32_t* dtab = reinterpret_cast<uint32_t*>(dyn + OFF_DTAB);
  float* misc = reinterpret_cast<float*>(dyn + OFF_MISC);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = L576;
  const int h = (int)blockIdx.x / x.wph, wgh = (int)blockIdx.x % x.wph;
  const int seq0 = wgh * x.ipw;
  constexpr int NCLS = TROWS * TROWS;
  // ---- once per workgroup: the head's bias table (log2 domain, stride 56), a zeroed class table, the scale vectors, the accumulators
  float* tabl1 = reinterpret_cast<float*>(dyn + OFF_TAB1);
  for (int i = tid; i < TN; i += NTH2) { tabl[i] = 0.f; tabl1[i] = 0.f; dtab[i] = 0u; }
  __syncthreads();
  if (TAB) {
    for (int i = tid; i < NCLS; i += NTH2) {
      const float t = p.tab[(int64_t)i * p.H + h] * LOG2E;
      tabl[(i / TROWS) * TS + i % TROWS] = t; tabl1[(i / TROWS) * TS + i % TROWS] = t;
    }
  }
  if (tid < 32) { const float ks = p.k_scale[tid]; misc[32 + tid] = ks; misc[64 + tid] = fabsf(ks) > 1e-30f ? 1.f / ks : 0.f; }
  if (tid < 2 * NW2 * 32) misc[96 + tid] = 0.f;
  if (tid < NW2) reinterpret_cast<int*>(misc + 8)[tid] = 0;
  __syncthreads();

  for (int it = 0; it < x.ipw; ++it) {
    const int seq = seq0 + it;
    const int64_t so = ((int64_t)h * p.M + (int64_t)seq * L) * D;
    const int64_t tok0 = (int64_t)seq * L;
    unsigned long long* stp = (x.stamps && blockIdx.x == 0 && it < 6) ? x.stamps + it * 16 : nullptr;
#define BWD2_KST(i) do { if (stp && tid == 0) stp[i] = wall_clock64(); } while (0)
    BWD2_KST(0);
    // ---------------------------------------------------------------------------------------------- load phase: the item's operands -> LDS
    const float invK = bwd2_load(Load2{p.qh + so, p.vh + so, p.dout + tok0 * p.lddo + h * D, p.o + tok0 * p.ldo + h * D, p.lse2 + (int64_t)h * p.M + tok0,
                                       p.lddo, p.ldo, stp});
    __syncthreads();
    BWD2_KST(2);
    // ---------------------------------------------------------------------------------------------- tile steps, exchanges, un-preps
    const bool more = it + 1 < x.ipw;
    const int64_t so2 = more ? so + (int64_t)L * D : so, tok2 = more ? tok0 + L : tok0;      // (no next item: this item's lines again, no branch)
    bwd2_steps<TAB, DTAB>(Steps2{p.kh + so, p.vh + so, p.qh + so, p.kinv + tok0 * p.H + h, x.qinv + tok0 * p.H + h, p.q_scale,
                                 p.dk_tok + tok0 * p.ldk_tok + h * D, p.dv_tok + tok0 * p.ldv_tok + h * D, x.dq_tok + tok0 * x.lddq + h * D,
                                 p.ldk_tok, p.ldv_tok, x.lddq, invK, p.c, p.H, it * 16, stp, stp ? x.stamps + 128 + it * 32 : nullptr,
                                 p.qh + so2, p.vh + so2, p.kh + so2, p.dout + tok2 * p.lddo + h * D, p.o + tok2 * p.ldo + h * D,
                                 p.lse2 + (int64_t)h * p.M + tok2, p.lddo, p.ldo});
    // ---------------------------------------------------------------------------------------------- flush the class table
    if (DTAB) {
      float* dst = x.dtpart + (((int64_t)wgh * x.ipw + it) * p.H + h) * NCLS;
      for (int dyi = wave; dyi < TROWS; dyi += NW2) {                 // a wave per table row, a lane per column: no integer division
        const int dxi = lane;
        if (dxi < TROWS) {
          const int pi = dyi * TS + dxi;
          const int ady = dyi - (GW - 1), adx = dxi - (GW - 1);
          const uint32_t cnt = (uint32_t)((GW - (ady < 0 ? -ady : ady)) * (GW - (adx < 0 ? -adx : adx)));
          const int32_t v = (int32_t)(dtab[pi] - cnt * MAGIC2_BITS);
          dst[dyi * TROWS + dxi] = (float)v * invK;
          dtab[pi] = 0u;
        }
      }
    }
    __syncthreads();
    BWD2_KST(7);
  }
  // scale gradients of this workgroup: the four waves in order
  if (tid < 64) {
    const int which = tid >> 5, d = tid & 31;
    float tsum = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < NW2; ++w4) tsum += misc[96 + which * NW2 * 32 + w4 * 32 + d];
    (which ? x.qpart : p.kpart)[(int64_t)blockIdx.x * 32 + d] = tsum;
  }
}

}  // namespace

bool attn2_bwd2_eligible(int nseq, int H, int L, int gh, int gw, bool tab) {
  const char* e = getenv("CTCLIP_ATTN_BWD2");
  if (e && e[0] == '0') return false;
  if (L != L576 || nseq <= 0 || H <= 0) return false;
  if (tab && (gh != GW || gw != GW)) return false;
  return true;
}

// the main kernel of ctclip_attn2_bwd_fused (attn2_bwd1.hip) in its four-wave form; x.dtpart null: no table gradient.  The caller runs the
// scale-sum and table-sum launches behind it (same partial layouts as bwd1_kernel).
int attn2_bwd2_launch(const ctclip_attn2::Params& p, const ctclip_attn2::Bwd2Args& x, int nwg, hipStream_t stream) {
  static bool raised_dev[64] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  bool& raised = raised_dev[(dev >= 0 && dev < 64) ? dev : 0];
  if (!raised) {
    bool ok = true;
    ok = ok && hipFuncSetAttribute((const void*)bwd2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SHM2) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)bwd2_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SHM2) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)bwd2_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SHM2) == hipSuccess;
    if (!ok) { ctclip_set_error("attn2_bwd_fused (four-wave form): cannot raise the LDS limit"); return CTCLIP_EBADARG; }
    raised = true;
  }
  const dim3 grid((unsigned)nwg), block(NTH2);
  if (p.tab && x.dtpart) hipLaunchKernelGGL((bwd2_kernel<true, true>), grid, block, SHM2, stream, p, x);
  else if (p.tab) hipLaunchKernelGGL((bwd2_kernel<true, false>), grid, block, SHM2, stream, p, x);
  else hipLaunchKernelGGL((bwd2_kernel<false, false>), grid, block, SHM2, stream, p, x);
  return ctclip_check_launch("attn2_bwd_fused (four-wave form)");
}
