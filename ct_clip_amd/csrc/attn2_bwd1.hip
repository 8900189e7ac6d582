// ONE-PASS backward of the CTViT spatial cosine attention (attention.py:145-178 with the position bias of attention.py:257-276; bf16,
// d_head 32, 256 <= L <= 576 tokens): dq, dk, dv (row-major, l2norm backward of attention.py:152-154 applied), the two learned-scale
// gradients and the position-bias TABLE gradient from a single sweep over the score tiles.  Replaces the query pass + key pass + dBias
// pass + dBias fold + q un-prep of attn2_slab.hip / attn2.hip (round 3: 178 + 238 + 273 + 68 + 35 us per layer), which computed
// S, P = exp2(S), dP and dS three times per tile.
//
// Work decomposition.  One persistent workgroup of eight waves per CU walks a run of (sequence, head) items of ONE head.  Per item the
// LDS holds the head-planar Q~ slab and a plain copy of the item's dO rows (both L x 64 B, the swizzled 32-row tiles of attn2_common.h),
// two f32 values per query (-delta_q = -sum_d dO O and log2 K - lse2_q), the f32 accumulators of dQ^T (L x 128 B, stored by accumulator
// register: [tile][register / 4][lane][register % 4], 16-byte accesses) and an integer class table for the bias gradient.  The
// L/32 x L/32 score tiles are enumerated row-major, g = kb * nkb + t (key block kb, query tile t), and wave w owns the contiguous
// positions [P w, P w + P): it keeps dK^T / dV^T of the current key block in registers and, per tile, computes
//     s = Q~ K^^T + bias + (log2 K - lse2_q),  dp = dO V^T - delta_q   (rows = queries in registers, lane = key; the two per-query terms
//                                                                      are matrix products of their own: three bf16 terms x ones)
//     p = exp2(s) = K x probability, ds = p dp;   dV^T += dO^T p,  dK^T += Q~^T ds   (transposing LDS reads, as the key pass did)
//     dQ^T[t] += K^^T ds^T                                            (ds transposed through LDS: written row-major, read with
//                                                                      ds_read_b64_tr_b16; the tile's own accumulator block is the scratch)
//     table[class(q, k)] += round(ds)                                  (ds_add_u32, see below)
// dQ^T[t] is a read-modify-write of LDS by whichever wave works on query tile t.  P is chosen so that P (w - w') != 0 (mod nkb): in every
// step the eight waves are on eight different query tiles, and the updates of one tile are ordered by a per-tile LDS counter whose expected
// value per (wave, step) is tabulated at kernel start (each wave keeps its row in ONE register, lane = step): no workgroup barrier in the
// tile loop, every tile sees its addends in a FIXED order (deterministic; no floating-point atomic anywhere).  A key block whose positions
// straddle two waves (7 of 18 at L = 576) has two partial dK^T / dV^T: the holder of the second part PARKS its accumulators in global
// scratch (L2) and raises an LDS flag, the holder of the first part adds them and applies the l2norm backward in the loop (k_scale and its
// reciprocal staged in LDS, the rows' inverse norms prefetched with the K^ / V rows: no division and no global load there).
//
// Bias-table gradient.  ds_add_f32 costs 161 cycles per wave instruction on gfx950 (tools/ubench/lds_atomic_rates.hip), ds_add_u32 4 --
// the same as a plain store.  So the scatter is done in FIXED POINT: per item a power of two K is chosen from the rigorous bound
// |dS| <= 2 max_q |dO_q| max_k |v_k| such that |K dS| < 2^21.  K enters through the LOGITS (log2 K is an integer: exact): p is K times the
// true probability, every product of the item is scaled by K and un-scaled by 1/K in the epilogues, and round(K dS) is obtained as the low
// bits of fma(p, dp, 1.5 * 2^23).  Integer addition is associative: the table is bit-identical from run to run whatever the order in which
// the waves' atomics retire.  576 addends per class and item stay below 2^31; the table is flushed (x 1/K, f32) into a per-(workgroup,
// item) partial after every item.  Because the row's lse2 is subtracted inside the exponent, p <= K whatever the logit span: one code path
// (the slab kernels' bounded-logit / row-maximum distinction does not exist here) and dO is never re-rounded.
//
// The bias itself comes from a per-head f32 table in GLOBAL memory (8.8 KB: L1-resident), log2 e applied by a one-workgroup-per-head
// stage kernel; a tile's 16 values per lane are requested at the end of the previous tile.  The LDS budget (160 KiB nearly to the byte)
// has no room for a second table.
#include "attn2_common.h"

namespace {

constexpr int NW1 = 8, NTH1 = NW1 * 64;
constexpr int NPIECE = 5;                    // 16-byte pieces per thread and slab: 4 L <= NPIECE * NTH1
constexpr int NTOUCH = 5;                    // L2 touches per thread and item: 3 L / 2 + 2 L + L / 32 lines <= NTOUCH * NTH1
constexpr float MAGIC = 12582912.f;          // 1.5 * 2^23: fma(x, y, MAGIC) has round(x y) in its low mantissa bits for |x y| < 2^22
constexpr uint32_t MAGIC_BITS = 0x4B400000u;
constexpr int FIX_BITS = 21;

struct G1 {                                  // token -> class arithmetic (table row stride S = 2 gw - 1: the natural class index)
  int gw, S, c0, magic, ncls, gh;
  __device__ __forceinline__ int u(int t) const { const int r = (t * magic) >> 16; return r * S + (t - r * gw); }
};

struct X1 {                                  // arguments of the fused backward beyond ctclip_attn2::Params
  const float* qinv; bf16_t* dq_tok; int64_t lddq;
  float* qpart;                              // [nwg][32] q_scale gradient partials (kpart: Params)
  const float* tabadj;                       // [H][ncls] staged table (log2 domain)
  float* dtpart;                             // [nseq][H][ncls] table-gradient partials, one per (workgroup, item), or null
  float* park;                               // [nwg][NW1][64][32] a wave's parked dK^T / dV^T accumulators (second part of a split key block)
  int P, ipw, wph;                           // positions per wave, items per workgroup, workgroups per head
  unsigned long long* stamps;                // profiling aid (tools/bench_attn2_bwd.py): 100-MHz clock at the phase boundaries of workgroup 0, or null
};

__device__ __forceinline__ void unpack8u(const u32x4& a, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(a[i] << 16); v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u); }
}
// every LDS operation of this wave has completed, then the workgroup barrier; global loads stay in flight
__device__ __forceinline__ void step_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Global-memory accesses through pointers that arrived as FUNCTION arguments: the compiler cannot tell their address space and emits FLAT
// instructions (counted on vmcnt AND lgkmcnt: every LDS wait then waits for them too) unless the access says "global"
#define GPTR(T, ptr) ((__attribute__((address_space(1))) T*)(ptr))
__device__ __forceinline__ Frag g_row(const bf16_t* row, int half) {
  Frag f;
  f.v[0] = *GPTR(const bf16x8, row + 8 * half);
  f.v[1] = *GPTR(const bf16x8, row + 16 + 8 * half);
  return f;
}
__device__ __forceinline__ void g_store8(bf16_t* ptr, const float (&v)[8]) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *GPTR(u32x4, ptr) = a;
}
// one dword of LDS, read now / written in program order (asm: a volatile generic pointer becomes a FLAT access that drains vmcnt)
__device__ __forceinline__ int lds_peek(uint32_t addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_poke(uint32_t addr, int v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void put_rows(char* tile, int row, int half, const Frag& f) {
  *reinterpret_cast<bf16x8*>(tile + swz(row, half)) = f.v[0];
  *reinterpret_cast<bf16x8*>(tile + swz(row, 2 + half)) = f.v[1];
}

// per head: the position-bias table in the log2 domain, contiguous ([H][ncls]; the model's table is [ncls][H])
__global__ __launch_bounds__(256) void bwd1_stage_kernel(Params p, float* __restrict__ tabadj, int ncls) {
  const int h = blockIdx.x, tid = threadIdx.x;
  if (p.tab) {
    for (int i = tid; i < ncls; i += 256) tabadj[(int64_t)h * ncls + i] = p.tab[(int64_t)i * p.H + h] * LOG2E;
  } else if (tid == 0) tabadj[(int64_t)h * ncls] = 0.f;
}

template <class T>
__device__ __forceinline__ T* uni(T* ptr) {                     // a wave-uniform pointer that arrived in vector registers (a function argument)
  const uint64_t v = (uint64_t)ptr;
  return (T*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v) {
  return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
__device__ __forceinline__ float uni(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }

// Compile-time ablation mask (tools/build_variant.py; never set in the product build -- results are WRONG, timing only): 1 = no class-table
// atomics, 2 = no dQ^T update at all (no wait, no read-modify-write), 4 = no exponential, 8 = no L2 touches, 16 = no dV / dK products and no
// transposing reads of Q~ / dO, 32 = no bias loads, 64 = dQ^T update without the tile-counter wait
#ifndef BWD1_ABL
#define BWD1_ABL 0
#endif
struct StepArgs {
  const bf16_t *k, *v;                       // the item's K^ / V slabs (head-planar)
  const float* tabh;                         // the head's staged table
  const float *kinv, *k_scale;               // inverse norms at (row 0, this head); learned scale
  bf16_t *dk, *dv; int64_t ldk, ldv;         // row 0 at this head of the two outputs
  float* park;                               // this workgroup's parked accumulators [NW1][64][32]
  float invK; int H, L, P;
  G1 g;
  int etrow;                                 // PER LANE: this wave's row of the step table (lane = step): updates of the step's query tile made before it
  unsigned long long* wstamp;                // profiling aid: spin time of wave 0 (100-MHz ticks), or null
  unsigned long long* sstamp;                // profiling aid: per-step sums [wave < 7][48] of the tile wait, then [48] step durations of wave 0; or null
};

// The tile steps of one item (see the file header): a call, so that the loop has the whole register file to itself -- inlined into the kernel
// it shared an allocation with the load phase (twenty 16-byte loads in flight per thread) and the un-prep arithmetic, and one side or the other
// spilled.  Ends with this item's k_scale-gradient sums in the waves' LDS rows; the caller's barrier publishes the dQ^T accumulators.
template <bool TAB, bool DTAB>
__device__ __noinline__ void bwd1_steps(StepArgs a_) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  StepArgs a = a_;
  a.k = uni(a.k); a.v = uni(a.v); a.tabh = uni(a.tabh); a.kinv = uni(a.kinv); a.k_scale = uni(a.k_scale);
  a.dk = uni(a.dk); a.dv = uni(a.dv); a.ldk = uni(a.ldk); a.ldv = uni(a.ldv); a.park = uni(a.park);
  a.invK = uni(a.invK); a.H = uni(a.H); a.L = uni(a.L); a.P = uni(a.P);
  a.g.gw = uni(a.g.gw); a.g.S = uni(a.g.S); a.g.c0 = uni(a.g.c0); a.g.magic = uni(a.g.magic); a.g.ncls = uni(a.g.ncls); a.g.gh = uni(a.g.gh);
  a.wstamp = uni(a.wstamp); a.sstamp = uni(a.sstamp);
  const G1& g = a.g;
  const int L = a.L, nkb = L / 32, NT = nkb * nkb, P = a.P;
  char* qs = dyn;
  char* dos = dyn + L * 64;
  char* dqa = dyn + L * 128;
  float* nd = reinterpret_cast<float*>(dyn + L * 256);           // -delta per query
  float* nl = reinterpret_cast<float*>(dyn + L * 260);           // log2 K - lse2 per query
  uint32_t* dtab = reinterpret_cast<uint32_t*>(dyn + L * 264);
  float* misc = reinterpret_cast<float*>(dyn + L * 264 + ((g.ncls * 4 + 15) & ~15));
  float* sred = misc + 64;
  const uint32_t tcnt_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(misc + 16);    // [32] tile counters
  const uint32_t pflag_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(misc + 48);   // [NW1] "wave w has parked its part"
  const float* ksr = sred + 2 * NW1 * 32;                        // [2][32]: k_scale and its guarded reciprocal (staged once per workgroup)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // (scalar: every position / key-block decision below is wave-uniform)
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const TrOff tr = tr_offsets(lane);
  const int g0 = P * wave;                                       // first position of this wave
  if (g0 >= NT) return;
  const int etrow = a.etrow;                                     // (a per-lane argument: lane = step)
  const int last = (g0 + P < NT ? g0 + P : NT) - 1;              // last position of this wave
  int kb = g0 / nkb, t = g0 - kb * nkb;
  float ksacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ksacc[i] = 0.f;
  float ik = 0.f, ikn;                                           // inverse norm of this lane's key row: current block, next block
  auto load_kv = [&](Frag& kk, Frag& vv, int jb) {
    const int64_t o2 = (int64_t)(jb * 32 + c) * D;
    kk = g_row(a.k + o2, half); vv = g_row(a.v + o2, half);
    ikn = *GPTR(const float, a.kinv + (int64_t)(jb * 32 + c) * a.H);
  };
  auto class0 = [&](int tt, int gq, int uc) { return g.u(tt * 32 + 16 * gq + 8 * half) - uc + g.c0; };
  auto bias_req = [&](f32x16& cb, int tt, int uc) {
    if (BWD1_ABL & 32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cb[r] = 0.f;
    } else if (TAB) {
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const __attribute__((address_space(1))) float* b = GPTR(const float, a.tabh + class0(tt, gq, uc));
#pragma unroll
        for (int e = 0; e < 8; ++e) cb[8 * gq + e] = b[e];
      }
    } else {
      const float tv = *GPTR(const float, a.tabh);
#pragma unroll
      for (int r = 0; r < 16; ++r) cb[r] = tv;
    }
  };
  f32x16 dkacc, dvacc;
  Frag kf, vf, ktf, kn, vn;
  load_kv(kn, vn, kb);
  int ucol = 0;
  bool need_ktf = false, newblk = false;
  f32x16 cbn;
  bias_req(cbn, t, g.u(kb * 32 + c));
  unsigned long long tspin = 0;
  unsigned long long tstep = a.sstamp ? wall_clock64() : 0ull;
  for (int s = 0; s < P; ++s) {
    const int gpos = g0 + s;
    if (gpos > last) break;
    if (a.sstamp && s > 0 && wave == 0 && lane == 0) {
      const unsigned long long now = wall_clock64();
      *GPTR(unsigned long long, a.sstamp + 7 * 48 + s - 1) += now - tstep;
      tstep = now;
    }
    if (s == 0 || t == 0) {                                      // a new key block starts here
      kf = kn; vf = vn; ik = ikn;
      ucol = g.u(kb * 32 + c);
      need_ktf = true; newblk = true;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkacc[r] = 0.f; dvacc[r] = 0.f; }
    }
    // ---- part A: everything that does not touch dQ^T (no ownership of query tile t needed)
    f32x16 cb = cbn;
    const char* qtile = qs + t * TILE;
    const char* dotile = dos + t * TILE;
    const Frag qf = lds_rows(qtile, ar, half);
    const Frag dof = lds_rows(dotile, ar, half);
    // The two per-QUERY terms of a tile enter as matrix products of their own instead of as C operands: row q of the A operand is the f32 value
    // split into three bf16 terms (hi + lo + lolo: 24 mantissa bits), the B operand is ones in those three contraction slots.
    //   * log2 K - lse2_q on the logits: p = exp2(s) is then K times the TRUE probability (<= K whatever the logit span: no bounded-logit
    //     special case, no per-row factor folded into dO, K never touches dO: the dO slab is a plain copy);
    //   * -delta_q on dP.
    // One 4-byte LDS read per lane each instead of 16-byte broadcast reads: the LDS pipe is this loop's bottleneck, the matrix pipe is 90 % idle.
    auto split3 = [&](float v) -> bf16x8 {
      const uint32_t hi = pack2bf(v, 0.f) & 0xffffu;
      const float r1 = v - __uint_as_float(hi << 16);
      const uint32_t lo = pack2bf(r1, 0.f) & 0xffffu;
      const float r2 = r1 - __uint_as_float(lo << 16);
      const uint32_t ll = pack2bf(r2, 0.f) & 0xffffu;
      const u32x4 w4 = half ? u32x4{0u, 0u, 0u, 0u} : u32x4{hi | (lo << 16), ll, 0u, 0u};
      return __builtin_bit_cast(bf16x8, w4);
    };
    const bf16x8 dlt = split3(nd[t * 32 + ar]), lgt = split3(nl[t * 32 + ar]);
    const bf16x8 ones3 = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x00003F80u, 0u, 0u});
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x16 cdel = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dlt, ones3, zero16, 0, 0, 0);
    cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lgt, ones3, cb, 0, 0, 0);
    f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf.v[0], kf.v[0], cb, 0, 0, 0);
    f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof.v[0], vf.v[0], cdel, 0, 0, 0);
    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf.v[1], kf.v[1], sc, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof.v[1], vf.v[1], dp, 0, 0, 0);
    Frag dotf, qtf;
    if (!(BWD1_ABL & 16)) { dotf = lds_cols(dotile, tr); qtf = lds_cols(qtile, tr); }
    float pr[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { pr[r] = (BWD1_ABL & 4) ? sc[r] : __builtin_amdgcn_exp2f(sc[r]); ds[r] = pr[r] * dp[r]; }
    if (DTAB && !(BWD1_ABL & 1)) {
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        uint32_t* b = dtab + class0(t, gq, ucol);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          (void)__hip_atomic_fetch_add(b + e, __float_as_uint(__builtin_fmaf(pr[8 * gq + e], dp[8 * gq + e], MAGIC)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    const Frag pf = pack(pr), dsf = pack(ds);
    if (!(BWD1_ABL & 16)) {
      dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf.v[0], pf.v[0], dvacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf.v[0], dsf.v[0], dkacc, 0, 0, 0);
      dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf.v[1], pf.v[1], dvacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf.v[1], dsf.v[1], dkacc, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { dvacc[r] += pr[r]; dkacc[r] += ds[r]; }
    }
    // ---- part B: the update of dQ^T[t] -- wait for the tile, then read-modify-write its accumulator block
    // wait until every earlier update of query tile t is complete: the steps of ALL waves order the updates of a tile (in one step the eight
    // waves are on eight different tiles), so the number of updates before step s is a closed form (tabulated by the kernel: etab) -- no
    // workgroup barrier
    const int texp = __builtin_amdgcn_readlane(etrow, s) + 1;
    if (!(BWD1_ABL & 2)) {
      float* dqt = reinterpret_cast<float*>(dqa + t * 4096);
      f32x16 dqc;                                                  // block layout [4][64 lanes][4]: four 16-byte accesses per lane each way
      {
        // ONE LDS round trip for "is the tile mine yet" and its accumulators: the counter is read FIRST and the LDS executes a wave's operations in
        // order (and the updater's poke after its writes), so accumulators read behind a counter that shows texp - 1 updates are complete; if it
        // does not, the reads are repeated
        const unsigned long long tb0 = a.wstamp ? wall_clock64() : 0ull;
        const uint32_t ca = tcnt_a + 4 * t, da = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)dqt + lane * 16;
        f32x4 q0, q1, q2, q3;
        for (;;) {
          int cnt;
          asm volatile("ds_read_b32 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:1024\n\tds_read_b128 %3, %6 offset:2048\n\t"
                       "ds_read_b128 %4, %6 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(cnt), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(ca), "v"(da) : "memory");
          if ((BWD1_ABL & 64) || __builtin_amdgcn_readfirstlane(cnt) >= texp - 1) break;
          // not yet: poll the counter alone (4 bytes instead of 4 KB per look), then read the block again
          do { __builtin_amdgcn_s_sleep(1); } while (__builtin_amdgcn_readfirstlane(lds_peek(ca)) < texp - 1);
        }
        if (a.wstamp) {
          const unsigned long long dt = wall_clock64() - tb0;
          tspin += dt;
          if (a.sstamp && wave < 7 && lane == 0) *GPTR(unsigned long long, a.sstamp + wave * 48 + s) += dt;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { dqc[e] = q0[e]; dqc[4 + e] = q1[e]; dqc[8 + e] = q2[e]; dqc[12 + e] = q3[e]; }
      }
      asm volatile("" ::: "memory");                               // (the block is re-used as bf16 scratch below: keep these reads in front of those stores)
      char* scratch = reinterpret_cast<char*>(dqt);                // this wave owns query tile t now; its accumulators are in dqc
      if (need_ktf) { put_rows(scratch, c, half, kf); ktf = lds_cols(scratch, tr); need_ktf = false; }
      put_rows(scratch, c, half, dsf);
      const Frag dstf = lds_cols(scratch, tr);
      dqc = mma(dqc, ktf, dstf);
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dqt + j * 256 + lane * 4) = f32x4{dqc[4 * j], dqc[4 * j + 1], dqc[4 * j + 2], dqc[4 * j + 3]};
      asm volatile("" ::: "memory");
      if (lane == 0) lds_poke(tcnt_a + 4 * t, texp);               // (LDS executes a wave's operations in order: the accumulators are written when this is seen)
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) dkacc[r] += __builtin_bit_cast(float, (uint32_t)dsf.v[r >> 3][(r & 7)] << 16);
      need_ktf = false;
    }
    {                                                            // the next tile's bias (L1-resident table): requested at the END of the tile, when
      int tn = t + 1, kbn = kb;                                  // the tile's temporaries are dead, consumed at the start of the next step
      if (tn == nkb) { tn = 0; kbn = kb + 1; }
      if (gpos + 1 <= last) bias_req(cbn, tn, g.u(kbn * 32 + c));
    }
    if (newblk) {                                                // the next block's K^ / V rows, a block ahead -- requested AFTER the bias (the bias wait
      newblk = false;                                            // of the next step is then a counted one that leaves these in flight)
      if ((kb + 1) * nkb <= last) load_kv(kn, vn, kb + 1);
    }
    if (t == nkb - 1 || gpos == last) {                          // this wave's tiles of key block kb are done
      float* pk = a.park + ((int64_t)wave * 64 + lane) * 32;
      if (kb * nkb < g0) {                                       // the block began in the previous wave's range: park this part for it
        __attribute__((address_space(1))) f32x4* pv = GPTR(f32x4, pk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pv[j] = f32x4{dkacc[4 * j], dkacc[4 * j + 1], dkacc[4 * j + 2], dkacc[4 * j + 3]};
          pv[4 + j] = f32x4{dvacc[4 * j], dvacc[4 * j + 1], dvacc[4 * j + 2], dvacc[4 * j + 3]};
        }
        wait_stores();
        if (lane == 0) lds_poke(pflag_a + 4 * wave, 1);
      } else {
        if ((kb + 1) * nkb - 1 > gpos) {                         // the rest of the block belongs to the next wave: add the part it parked (first
          while (__builtin_amdgcn_readfirstlane(lds_peek(pflag_a + 4 * (wave + 1))) == 0) __builtin_amdgcn_s_sleep(1);     // part + second part)
          asm volatile("" ::: "memory");
          const __attribute__((address_space(1))) f32x4* pv = GPTR(const f32x4, pk + 64 * 32);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 u = pv[j], w = pv[4 + j];
#pragma unroll
            for (int e = 0; e < 4; ++e) { dkacc[4 * j + e] += u[e]; dvacc[4 * j + e] += w[e]; }
          }
        }
        // un-prep in place (attn_unprep_kernel of attn2.hip): u = k^ / k_scale, g = dk^ k_scale, dk = kinv (g - u (u . g)); dscale += dk^ u.  Written
        // for few live registers (dV out first; the dot product in one pass, the outputs in a second one that recomputes u and g)
        const int row = kb * 32 + c;
        bf16_t* dV = a.dv + (int64_t)row * a.ldv;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          float b8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) b8[e] = dvacc[8 * gq + e] * a.invK;
          g_store8(dV + 16 * gq + 8 * half, b8);
        }
        __builtin_amdgcn_sched_barrier(0);
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
            dkacc[i] = gk0;
          }
        }
        const float dot = (half_sum(part[0])) + (half_sum(part[1]));
        __builtin_amdgcn_sched_barrier(0);
        bf16_t* dK = a.dk + (int64_t)row * a.ldk;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          const float* sp = ksr + 16 * gq + 8 * half;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(sp + 32), r1 = *reinterpret_cast<const f32x4*>(sp + 36);
          float a8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t kwd = gq ? kw1[e >> 1] : kw0[e >> 1];
            const float kx = (e & 1) ? __uint_as_float(kwd & 0xffff0000u) : __uint_as_float(kwd << 16);
            const float ks = e < 4 ? s0[e & 3] : s1[e & 3], rk = e < 4 ? r0[e & 3] : r1[e & 3];
            a8[e] = ik * (dkacc[8 * gq + e] * ks - (kx * rk) * dot);
          }
          g_store8(dK + 16 * gq + 8 * half, a8);
        }
      }
    }
    if (++t == nkb) { t = 0; ++kb; }
  }
  if (a.wstamp && wave == 0 && lane == 0) *GPTR(unsigned long long, a.wstamp) = tspin;
  // this item's k_scale-gradient sums: the 32 lanes of a half by an xor tree, then added to the wave's row of the LDS accumulator
#pragma unroll
  for (int i = 0; i < 16; ++i)
    ksacc[i] = half32_sum(ksacc[i]);
  if (c == 0) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int e = 0; e < 8; ++e) sred[wave * 32 + 16 * gq + 8 * half + e] += ksacc[8 * gq + e];
  }
}

struct UnprepArgs {
  const float* qinv;                         // inverse norms at (row 0 of the item, this head); row stride H
  bf16_t* dq; int64_t lddq;                  // row 0 of the item at this head
  const float* q_scale; float c, invK;
  int H, L, ncls;
  const bf16_t *nq, *nv, *nk, *ndout, *no;   // the NEXT item's load-phase operands (touched line by line)
  const float* nlse; int64_t lddo, ldo;
};

// dQ of one item: un-prep of q in place (attn_unprep_kernel of attn2.hip: u = q~ / (q_scale c), g = dq^ q_scale, dq = qinv (g - u (u . g)),
// dscale += dq^ u on the bf16-rounded dq^) from the finished LDS accumulators, and the L2 prefetch of the NEXT item's operands (Q~, V, K^
// slabs, its dout / o rows, lse2): one dword per 128-byte line, requested behind this function's own loads and consumed at its end.  A call for the
// same reason as the steps: inlined, the kernel's long-lived values sit in scratch here, every reload is a VMEM operation and -- loads retire in
// order -- waits for the touches in front of it (15 us per item instead of 8).  (Inside the tile steps the touches cost more than they saved:
// every step waits for its bias behind them.)
__device__ __noinline__ void bwd1_unprep_q(UnprepArgs a_) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  UnprepArgs a = a_;
  a.qinv = uni(a.qinv); a.dq = uni(a.dq); a.lddq = uni(a.lddq); a.q_scale = uni(a.q_scale); a.c = uni(a.c); a.invK = uni(a.invK);
  a.H = uni(a.H); a.L = uni(a.L); a.ncls = uni(a.ncls); a.nq = uni(a.nq); a.nv = uni(a.nv); a.nk = uni(a.nk); a.ndout = uni(a.ndout);
  a.no = uni(a.no); a.nlse = uni(a.nlse); a.lddo = uni(a.lddo); a.ldo = uni(a.ldo);
  const int L = a.L, nkb = L / 32;
  const char* qs = dyn;
  const char* dqa = dyn + L * 128;
  float* sred = reinterpret_cast<float*>(dyn + L * 264 + ((a.ncls * 4 + 15) & ~15)) + 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  constexpr int MAXB = 3;                                        // query tiles per wave: ceil(18 / 8)
  float iq[MAXB], qsv[16];
#pragma unroll
  for (int j = 0; j < MAXB; ++j) {
    const int tq = wave + j * NW1 < nkb ? wave + j * NW1 : nkb - 1;
    iq[j] = *GPTR(const float, a.qinv + (int64_t)(tq * 32 + ar) * a.H);
  }
#pragma unroll
  for (int gq = 0; gq < 2; ++gq)
#pragma unroll
    for (int e = 0; e < 8; ++e) qsv[8 * gq + e] = *GPTR(const float, a.q_scale + 16 * gq + 8 * half + e);
  // the touches, after this function's own loads (loads retire in order: the waits for those are counted ones that leave the touches in flight).
  // Wave-uniform chunks of 64 lines: chunk j * 8 + wave walks the segments Q~ | V | K^ (L / 2 lines each, 128 B apart), dout | o (L rows, one
  // line per row), lse2 (L / 32 lines), each padded to whole chunks: segment, base and stride are scalar selects, the lane part one clamped multiply
  uint32_t tch[NTOUCH];
#pragma unroll
  for (int j = 0; j < NTOUCH; ++j) tch[j] = 0u;
  if (!(BWD1_ABL & 8)) {
    const int hl = L / 2, cq = (hl + 63) >> 6, cr = (L + 63) >> 6;
#pragma unroll
    for (int j = 0; j < NTOUCH; ++j) {
      int r = j * NW1 + wave, sg = 0;
      if (r >= cq) { r -= cq; sg = 1; }
      if (sg == 1 && r >= cq) { r -= cq; sg = 2; }
      if (sg == 2 && r >= cq) { r -= cq; sg = 3; }
      if (sg == 3 && r >= cr) { r -= cr; sg = 4; }
      if (sg == 4 && r >= cr) { r -= cr; sg = 5; }
      const char* base = reinterpret_cast<const char*>(sg == 0 ? a.nq : (sg == 1 ? a.nv : (sg == 2 ? a.nk : (sg == 3 ? a.ndout : a.no))));
      if (sg == 5) base = reinterpret_cast<const char*>(a.nlse);
      const uint32_t stride = sg == 3 ? (uint32_t)a.lddo * 2u : (sg == 4 ? (uint32_t)a.ldo * 2u : 128u);
      const int nl = sg < 3 ? hl : (sg < 5 ? L : L / 32);
      int ln = r * 64 + lane;
      ln = ln < nl ? ln : nl - 1;
      tch[j] = *GPTR(const uint32_t, base + (uint32_t)ln * stride);
    }
  }
  float qsacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) qsacc[i] = 0.f;
  const float scq = a.c * LN2 * a.invK;
#pragma unroll
  for (int j = 0; j < MAXB; ++j) {
    const int tq = wave + j * NW1;
    if (tq < nkb) {                                              // (wave-uniform)
      const float* dqt = reinterpret_cast<const float*>(dqa + tq * 4096);
      const Frag qrow = lds_rows(qs + tq * TILE, ar, half);      // lane n of the transposed product holds query pi32(n & 31)
      float qx[16], dq[16];
      unpack8u(__builtin_bit_cast(u32x4, qrow.v[0]), qx); unpack8u(__builtin_bit_cast(u32x4, qrow.v[1]), qx + 8);
      float part[2] = {0.f, 0.f};
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = 8 * gq + e;
          const float qc = qsv[i] * a.c;
          const float rq = fabsf(qc) > 1e-30f ? 1.f / qc : 0.f;
          const float gq0 = bf2f(f2bf(dqt[(i >> 2) * 256 + lane * 4 + (i & 3)] * scq));
          const float uq = qx[i] * rq;
          qsacc[i] += gq0 * uq;
          const float gv = gq0 * qsv[i];
          part[gq] += uq * gv;
          qx[i] = uq; dq[i] = gv;
        }
      const float dot = (half_sum(part[0])) + (half_sum(part[1]));
      bf16_t* dQ = a.dq + (int64_t)(tq * 32 + ar) * a.lddq;
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[e] = iq[j] * (dq[8 * gq + e] - qx[8 * gq + e] * dot);
        g_store8(dQ + 16 * gq + 8 * half, a8);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    qsacc[i] = half32_sum(qsacc[i]);
  if (c == 0) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int e = 0; e < 8; ++e) sred[NW1 * 32 + wave * 32 + 16 * gq + 8 * half + e] += qsacc[8 * gq + e];
  }
#pragma unroll
  for (int j = 0; j < NTOUCH; ++j) asm volatile("" :: "v"(tch[j]));
}

template <bool TAB, bool DTAB>
__device__ __forceinline__ void bwd1_body(const Params& p, const X1& x, const G1& g, char* dyn) {
  const int L = p.L, nkb = L / 32;
  char* qs = dyn;                                               // Q~ slab
  char* dos = dyn + L * 64;                                     // dO slab (a plain copy)
  char* dqa = dyn + L * 128;                                    // dQ^T accumulators [tile][4][64 lanes][4] f32 (register r of lane l at (r >> 2, l, r & 3))
  float* nd = reinterpret_cast<float*>(dyn + L * 256);          // -delta per query
  float* nl = reinterpret_cast<float*>(dyn + L * 260);          // log2 K - lse2 per query (before the first item: the step table, see below)
  uint32_t* dtab = reinterpret_cast<uint32_t*>(dyn + L * 264);  // class table (fixed point)
  float* misc = reinterpret_cast<float*>(dyn + L * 264 + ((g.ncls * 4 + 15) & ~15));       // [0,16) reductions, [16,48) tile counters, [48,56) park flags
  float* sred = misc + 64;                                      // [2][NW1][32] scale-gradient sums of the waves (k, q) over the items
  int* cnts = reinterpret_cast<int*>(misc + 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5, ar = pi32(c);
  const int h = (int)blockIdx.x / x.wph, wgh = (int)blockIdx.x % x.wph;
  const int seq0 = wgh * x.ipw;
  if (DTAB) { for (int i = tid; i < g.ncls; i += NTH1) dtab[i] = 0u; }
  if (tid < 2 * NW1 * 32) sred[tid] = 0.f;
  {   // etab[w][s] = number of updates of query tile t(w, s) = (P w + s) mod nkb made in steps < s by all waves (see bwd1_steps)
    float* ksr = sred + 2 * NW1 * 32;                           // k_scale and 1 / k_scale for the in-loop un-prep of dK (no division, no global load there)
    if (tid < 32) { const float ks = p.k_scale[tid]; ksr[tid] = ks; ksr[32 + tid] = fabsf(ks) > 1e-30f ? 1.f / ks : 0.f; }
    int* etab = reinterpret_cast<int*>(nl);                     // (lives where log2 K - lse2 will: every wave takes its row into a register below, before the first item writes there)
    const int NT = nkb * nkb;
    for (int i = tid; i < NW1 * x.P; i += NTH1) {
      const int w = i / x.P, sidx = i - w * x.P, tt = (x.P * w + sidx) % nkb;
      int n = 0;
      for (int w8 = 0; w8 < NW1; ++w8) {
        const int len = NT - x.P * w8 < x.P ? NT - x.P * w8 : x.P;              // positions of wave w8 (may be <= 0)
        const int lim = sidx < len ? sidx : len;
        int f = (tt - x.P * w8) % nkb; f = f < 0 ? f + nkb : f;                  // its first step on tile tt
        if (f < lim) n += (lim - 1 - f) / nkb + 1;
      }
      etab[i] = n;
    }
  }
  __syncthreads();
  const int etrow = reinterpret_cast<const int*>(nl)[wave * x.P + (lane < x.P ? lane : 0)];      // lane = step (P <= 64); nl is first written behind the load phase's barrier

  // One item's load-phase operands (twenty 16-byte pieces per thread)
  u32x4 oq[NPIECE], ov_[NPIECE], od[NPIECE], oo[NPIECE];
  float ols[NPIECE];
  auto issue_item = [&](int seq) {
    const int64_t so = ((int64_t)h * p.M + (int64_t)seq * L) * D, tok0 = (int64_t)seq * L;
    const bf16_t* qsl = p.qh + so;
    const bf16_t* vsl = p.vh + so;
    const bf16_t* dsl = p.dout + tok0 * p.lddo + h * D;
    const bf16_t* osl = p.o + tok0 * p.ldo + h * D;
    const float* lsl = p.lse2 + (int64_t)h * p.M + tok0;
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      const int pc = k * NTH1 + tid;
      const int pcc = pc < 4 * L ? pc : 4 * L - 4 + (pc & 3), row = pcc >> 2, ch = pcc & 3;      // (clamped: the last row's four chunks, as a quad)
      oq[k] = *reinterpret_cast<const u32x4*>(qsl + row * D + ch * 8);
      ov_[k] = *reinterpret_cast<const u32x4*>(vsl + row * D + ch * 8);
      od[k] = *reinterpret_cast<const u32x4*>(dsl + (int64_t)row * p.lddo + ch * 8);
      oo[k] = *reinterpret_cast<const u32x4*>(osl + (int64_t)row * p.ldo + ch * 8);
      ols[k] = lsl[row];
    }
  };

#define BWD1_STAMP(i) do { if (x.stamps && blockIdx.x == 0 && tid == 0) x.stamps[it * 16 + (i)] = wall_clock64(); } while (0)
  for (int it = 0; it < x.ipw; ++it) {
    const int seq = seq0 + it;
    BWD1_STAMP(0);
    const int64_t so = ((int64_t)h * p.M + (int64_t)seq * L) * D;
    const int64_t tok0 = (int64_t)seq * L;
    // ------------------------------------------------------------------------------------------------ load phase: the item's operands -> LDS
    // Q~ slab and dO slab (plain copies), -delta = -sum_d dO O, log2 K - lse2 (K: the fixed-point scale of the item, a power of two), zeroed dQ^T
    // accumulators, tile counters and park flags
    float invK = 1.f;
    issue_item(seq);      // (requesting these an item ahead, under the previous item's dQ un-prep, makes hipcc spill every piece as it lands: 20 serial HBM round trips)
    {
      float lsr[NPIECE];
      float mxd = 0.f, mxv = 0.f;
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) {
        const int pc = k * NTH1 + tid;
        const int pcc = pc < 4 * L ? pc : 4 * L - 4 + (pc & 3), row = pcc >> 2, ch = pcc & 3;      // (clamped: the last row's four chunks, as a quad)
        const u32x4 qv = oq[k], vv = ov_[k], dv = od[k], ov = oo[k];
        lsr[k] = ols[k];
        // (no branch in this loop: a branch would end the basic block and the next piece's loads would wait for this piece's -- five HBM
        // round trips instead of one; a clamped piece re-writes the last piece's bytes)
        *reinterpret_cast<u32x4*>(qs + (row >> 5) * TILE + swz(row & 31, ch)) = qv;
        *reinterpret_cast<u32x4*>(dos + (row >> 5) * TILE + swz(row & 31, ch)) = dv;
        float x8[8], b[8], v8[8];
        unpack8u(dv, x8); unpack8u(ov, b); unpack8u(vv, v8);
        float ds = 0.f, dn = 0.f, vnn = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ds += x8[e] * b[e]; dn += x8[e] * x8[e]; vnn += v8[e] * v8[e]; }
        ds = quad_sum(ds);
        dn = quad_sum(dn);
        vnn = quad_sum(vnn);
        nd[row] = -ds;                                                // (the four chunk threads of a row hold the same value)
        mxd = fmaxf(mxd, dn); mxv = fmaxf(mxv, vnn);                  // |dS| <= P 2 |dO_q| |v_k| with the probability P <= 1
      }
      BWD1_STAMP(1);
      mxd = wave_max(mxd); mxv = wave_max(mxv);
      if (lane == 0) { misc[wave] = mxd; misc[8 + wave] = mxv; }
      for (int i = tid; i < L * 8; i += NTH1) *reinterpret_cast<u32x4*>(dqa + (int64_t)i * 16) = u32x4{0u, 0u, 0u, 0u};
      if (tid < 40) cnts[tid] = 0;
      __syncthreads();
      float bd = 0.f, bv = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < NW1; ++w8) { bd = fmaxf(bd, misc[w8]); bv = fmaxf(bv, misc[8 + w8]); }
      const float B = 2.f * sqrtf(bd) * sqrtf(bv);
      int kk = 0;
      if (B > 0.f && B < 3.0e38f) {
        int e; (void)frexpf(B, &e);                              // B < 2^e
        kk = FIX_BITS - e; kk = kk > 100 ? 100 : (kk < -100 ? -100 : kk);
        invK = ldexpf(1.f, -kk);
      } else if (!(B < 3.0e38f)) {
        invK = __builtin_nanf("");                               // a non-finite dO: NaN in, NaN out (dq / dk / dv carry it by arithmetic; the
      }                                                          // fixed-point table gradient would otherwise come out as finite garbage)
      const float lgK = (float)kk;                               // K = 2^kk enters through the logits: exp2(s + kk - lse2) = K x probability
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) {
        const int pc = k * NTH1 + tid;
        const int pcc = pc < 4 * L ? pc : 4 * L - 4 + (pc & 3), row = pcc >> 2;
        nl[row] = lgK - lsr[k];
      }
    }
    __syncthreads();
    BWD1_STAMP(2);
    // ------------------------------------------------------------------------------------------------ tile steps (+ dK / dV un-prep inside)
    {
      bwd1_steps<TAB, DTAB>(StepArgs{p.kh + so, p.vh + so, x.tabadj + (int64_t)h * g.ncls, p.kinv + tok0 * p.H + h,
                                           p.k_scale, p.dk_tok + tok0 * p.ldk_tok + h * D, p.dv_tok + tok0 * p.ldv_tok + h * D, p.ldk_tok, p.ldv_tok,
                                           x.park + (int64_t)blockIdx.x * NW1 * 64 * 32, invK, p.H, L, x.P, g, etrow, (x.stamps && blockIdx.x == 0) ? x.stamps + it * 16 + 9 : nullptr,
                                           (x.stamps && blockIdx.x == 0) ? x.stamps + 128 : nullptr});
    }
    BWD1_STAMP(3);
    __syncthreads();
    BWD1_STAMP(4);
    // ------------------------------------------------------------------------------------------------ dQ: un-prep of q in place + L2 touches of the next item
    {
      const bool more = it + 1 < x.ipw;
      const int64_t so2 = more ? so + (int64_t)L * D : so, tok2 = more ? tok0 + L : tok0;      // (no next item: this item's lines again, no branch)
      bwd1_unprep_q(UnprepArgs{x.qinv + tok0 * p.H + h, x.dq_tok + tok0 * x.lddq + h * D, x.lddq, p.q_scale, p.c, invK, p.H, L, g.ncls,
                               p.qh + so2, p.vh + so2, p.kh + so2, p.dout + tok2 * p.lddo + h * D, p.o + tok2 * p.ldo + h * D,
                               p.lse2 + (int64_t)h * p.M + tok2, p.lddo, p.ldo});
    }
    BWD1_STAMP(7);
    // ------------------------------------------------------------------------------------------------ flush the class table (stores only)
    if (DTAB) {
      const int W = 2 * g.gw - 1;
      float* dst = x.dtpart + (((int64_t)wgh * x.ipw + it) * p.H + h) * g.ncls;
      for (int i = tid; i < g.ncls; i += NTH1) {
        const int dyi = i / W, dxi = i - dyi * W;
        const int ady = dyi - (g.gh - 1), adx = dxi - (g.gw - 1);
        const uint32_t cnt = (uint32_t)((g.gh - (ady < 0 ? -ady : ady)) * (g.gw - (adx < 0 ? -adx : adx)));
        const int32_t v = (int32_t)(dtab[i] - cnt * MAGIC_BITS);
        dst[i] = (float)v * invK;
        dtab[i] = 0u;
      }
    }
    __syncthreads();
    BWD1_STAMP(8);
  }

  // scale gradients of this workgroup: the eight waves in order
  if (tid < 64) {
    const int which = tid >> 5, d = tid & 31;
    float tsum = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < NW1; ++w8) tsum += sred[which * NW1 * 32 + w8 * 32 + d];
    (which ? x.qpart : p.kpart)[(int64_t)blockIdx.x * 32 + d] = tsum;
  }
}

template <bool TAB, bool DTAB>
__global__ __launch_bounds__(NTH1) void bwd1_kernel(Params p, X1 x, G1 g) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  bwd1_body<TAB, DTAB>(p, x, g, dyn);
}

// part[nblk][32] -> dst (+=), two vectors per launch (blockIdx.x: 0 = k, 1 = q): 32 interleaved slices, then the slices in a fixed order
__global__ __launch_bounds__(1024) void bwd1_scale_sum_kernel(const float* __restrict__ kpart, const float* __restrict__ qpart, int nblk,
                                                              float* __restrict__ dks, float* __restrict__ dqs) {
  __shared__ float red[32][32];
  const float* part = blockIdx.x ? qpart : kpart;
  float* dst = blockIdx.x ? dqs : dks;
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  float t = 0.f;
  for (int b = sl; b < nblk; b += 32) t += part[(int64_t)b * 32 + o];
  red[sl][o] = t;
  __syncthreads();
  if (threadIdx.x < 32) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) a += red[i][o];
    dst[o] += a;
  }
}
// dtpart[nblk][H][ncls] -> dtab (ncls, H), overwritten; the partials in index order (eight at a time in flight, summed left to right)
__global__ __launch_bounds__(256) void bwd1_dtab_sum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dtab, int H, int ncls) {
  // 64 outputs per workgroup x 4 interleaved slices of the nblk partials (a thread: 8 loads in flight, nblk / 32 round trips instead of
  // nblk / 8: 15 -> 6 us on the critical path of every layer); the four slice sums are combined in a fixed order
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;                            // i = h * ncls + cls: neighbours read neighbouring addresses
  const bool live = i < ncls * H;
  const int ic = live ? i : 0;
  const int64_t stride = (int64_t)H * ncls;
  float t = 0.f;
  int b = sl;
  for (; b + 28 < nblk; b += 32) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(int64_t)(b + 4 * k) * stride + ic];
#pragma unroll
    for (int k = 0; k < 8; ++k) t += v[k];
  }
  for (; b < nblk; b += 4) t += part[(int64_t)b * stride + ic];
  red[sl][o] = t;
  __syncthreads();
  if (sl == 0 && live) {
    const int h = i / ncls, cls = i - h * ncls;
    dtab[(int64_t)cls * H + h] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
  }
}

constexpr int MAXDEV1 = 64;
int cur_dev1() { int dev = 0; (void)hipGetDevice(&dev); return (dev >= 0 && dev < MAXDEV1) ? dev : 0; }
int ncus1() {        // cached per DEVICE (one process may drive several)
  static int ncu[MAXDEV1] = {};
  const int dev = cur_dev1();
  if (!ncu[dev]) { hipDeviceProp_t prop; ncu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
  return ncu[dev];
}
inline int64_t a256(int64_t v) { return (v + 255) / 256 * 256; }

struct Plan1 { G1 g; int P, ipw, wph, nwg, ncls; size_t shm; };
bool plan1(int nseq, int H, int L, int gh, int gw, bool tab, Plan1& pl) {
  const char* e = getenv("CTCLIP_ATTN_BWD1");
  if (e && e[0] == '0') return false;
  if (L % 32 || L < 256 || 4 * L > NPIECE * NTH1 || nseq <= 0 || H <= 0) return false;
  const int nkb = L / 32, NT = nkb * nkb;
  pl.ncls = 1;
  pl.g = G1{1, 1, 0, 65536, 1, 1};
  if (tab) {
    if (gh * gw != L || gw % 8 || gw > 64) return false;
    const int S = 2 * gw - 1;
    pl.ncls = (2 * gh - 1) * S;
    pl.g = G1{gw, S, (gh - 1) * S + (gw - 1), (65536 + gw - 1) / gw, pl.ncls, gh};
  }
  pl.shm = (size_t)L * 264 + (size_t)((pl.ncls * 4 + 15) & ~15) + 256 + 2 * NW1 * 32 * 4 + 64 * 4;
  // (+ the step table, sized after P below)
  if (pl.shm > 160 * 1024) return false;
  int P = (NT + NW1 - 1) / NW1;
  if (P < nkb) P = nkb;
  for (;; ++P) {
    bool ok = true;
    for (int d = 1; d < NW1; ++d) ok = ok && (P * d) % nkb != 0;
    if (ok) break;
  }
  if (P > 64) return false;                                     // (a wave keeps its row of the step table in ONE register, lane = step)
  pl.P = P;
  if (NW1 * P > L) return false;                                // (the step table is built where the L per-query floats log2 K - lse2 live later)
  const int ncu = ncus1(), total = nseq * H;
  int ipw = (total + ncu - 1) / ncu;
  if (ipw > nseq) ipw = nseq;                                   // (H above the CU count with few sequences: one workgroup per head walks them all)
  while (nseq % ipw) ++ipw;                                     // a workgroup stays inside one head (terminates at ipw = nseq at the latest)
  pl.ipw = ipw; pl.wph = nseq / ipw; pl.nwg = pl.wph * H;
  return true;
}

}  // namespace

extern "C" int ctclip_attn2_bwd_fused_supported(int nseq, int H, int L, int D_, int bias_gh, int bias_gw, int has_bias) {
  Plan1 pl;
  return D_ == D && plan1(nseq, H, L, bias_gh, bias_gw, has_bias != 0, pl) ? 1 : 0;
}
extern "C" int64_t ctclip_attn2_bwd_fused_workspace(int nseq, int H, int L, int bias_gh, int bias_gw) {
  Plan1 pl;
  if (!plan1(nseq, H, L, bias_gh, bias_gw, bias_gh > 0, pl)) return 0;
  return a256((int64_t)H * pl.ncls * 4) + a256((int64_t)nseq * H * pl.ncls * 4) + 2 * a256((int64_t)pl.nwg * 32 * 4) +
         a256((int64_t)pl.nwg * NW1 * 64 * 32 * 4) + 4096;
}

// Backward of ctclip_attn2_fwd in ONE pass over the score tiles (attention.py:145-178 differentiated; the l2norm / learned-scale backward of
// attention.py:152-154 and the position-bias table gradient of attention.py:257-276 included): row-major dq (M, lddq), dk (M, lddk), dv (M, lddv)
// w.r.t. the projections q, k, v; dq_scale / dk_scale (32) ACCUMULATED; dtab (ncls, H) OVERWRITTEN when non-null.  qh / kh / vh: the head-planar
// operands of ctclip_attn2_prep (or ctclip_gemm_headnorm), qinv / kinv (M, H) their inverse norms, o / dout (M, ldo / lddo), lse2 [H][M] of the
// forward.  Deterministic (no floating-point atomics; the table gradient is accumulated in fixed point, see the file header).
// CTCLIP_EUNSUPPORTED when the shape is not served (ctclip_attn2_bwd_fused_supported): callers fall back to ctclip_attn2_bwd_tok.
extern "C" int ctclip_attn2_bwd_fused(const void* qh, const void* kh, const void* vh, const float* tab, int bias_gh, int bias_gw, const float* q_scale,
                                      const float* k_scale, float scale, const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse2,
                                      const float* qinv, const float* kinv, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                      float* dq_scale, float* dk_scale, float* dtab, int nseq, int H, int L, void* workspace, int64_t workspace_bytes,
                                      hipStream_t stream) {
  if (!qh || !kh || !vh || !o || !dout || !lse2 || !qinv || !kinv || !dq || !dk || !dv || !dq_scale || !dk_scale || !q_scale || !k_scale || ldo % 8 ||
      lddo % 8 || lddq % 8 || lddk % 8 || lddv % 8) { ctclip_set_error("attn2_bwd_fused: bad args"); return CTCLIP_EBADARG; }
  if (dtab && !tab) { ctclip_set_error("attn2_bwd_fused: dtab without a table"); return CTCLIP_EBADARG; }
  Plan1 pl;
  if (!plan1(nseq, H, L, bias_gh, bias_gw, tab != nullptr, pl)) return CTCLIP_EUNSUPPORTED;
  if (!workspace || workspace_bytes < ctclip_attn2_bwd_fused_workspace(nseq, H, L, tab ? bias_gh : 0, bias_gw)) { ctclip_set_error("attn2_bwd_fused: workspace too small"); return CTCLIP_EWORKSPACE; }
  const int64_t M = (int64_t)nseq * L;
  Params p{};
  p.qh = (const bf16_t*)qh; p.kh = (const bf16_t*)kh; p.vh = (const bf16_t*)vh; p.tab = tab; p.q_scale = q_scale; p.k_scale = k_scale;
  p.gh = bias_gh; p.gw = bias_gw; p.H = H; p.L = L; p.nseq = nseq; p.M = M; p.c = scale * LOG2E;
  p.o = (const bf16_t*)o; p.ldo = ldo; p.dout = (const bf16_t*)dout; p.lddo = lddo; p.lse2 = const_cast<float*>(lse2);
  p.dk_tok = (bf16_t*)dk; p.dv_tok = (bf16_t*)dv; p.ldk_tok = lddk; p.ldv_tok = lddv; p.kinv = kinv;
  char* w = (char*)workspace;
  float* tabadj = (float*)w; w += a256((int64_t)H * pl.ncls * 4);
  float* dtpart = (float*)w; w += a256((int64_t)nseq * H * pl.ncls * 4);
  p.kpart = (float*)w; w += a256((int64_t)pl.nwg * 32 * 4);
  float* qpart = (float*)w; w += a256((int64_t)pl.nwg * 32 * 4);
  float* park = (float*)w; w += a256((int64_t)pl.nwg * NW1 * 64 * 32 * 4);
  static const bool stamp = getenv("CTCLIP_BWD1_STAMPS") != nullptr;          // the last 4 KB of the workspace: phase clocks of workgroup 0
  X1 x{qinv, (bf16_t*)dq, lddq, qpart, tabadj, dtab ? dtpart : nullptr, park, pl.P, pl.ipw, pl.wph, stamp ? (unsigned long long*)w : nullptr};
  int rc;
  if (attn2_bwd2_eligible(nseq, H, L, bias_gh, bias_gw, tab != nullptr)) {
    // the four-wave form (attn2_bwd2.hip: dQ^T in registers, tables in LDS); same partial layouts, the two sum launches below are shared
    const ctclip_attn2::Bwd2Args x2{qinv, (bf16_t*)dq, lddq, qpart, dtab ? dtpart : nullptr, pl.ipw, pl.wph, stamp ? (unsigned long long*)w : nullptr};
    rc = attn2_bwd2_launch(p, x2, pl.nwg, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(bwd1_scale_sum_kernel, dim3(2), dim3(1024), 0, stream, (const float*)p.kpart, (const float*)qpart, pl.nwg, dk_scale, dq_scale);
    rc = ctclip_check_launch("attn2_bwd_fused (scale sums)");
    if (rc || !dtab) return rc;
    hipLaunchKernelGGL(bwd1_dtab_sum_kernel, dim3((unsigned)cdiv((int64_t)pl.ncls * H, 64)), dim3(256), 0, stream, (const float*)dtpart, nseq, dtab, H, pl.ncls);
    return ctclip_check_launch("attn2_bwd_fused (table sum)");
  }
  hipLaunchKernelGGL(bwd1_stage_kernel, dim3((unsigned)H), dim3(256), 0, stream, p, tabadj, pl.ncls);
  rc = ctclip_check_launch("attn2_bwd_fused (stage)");
  if (rc) return rc;
  static bool raised_dev[MAXDEV1] = {};                         // the attribute is per device
  bool& raised = raised_dev[cur_dev1()];
  if (!raised) {
    bool ok = true;
    ok = ok && hipFuncSetAttribute((const void*)bwd1_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)bwd1_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)bwd1_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (!ok) { ctclip_set_error("attn2_bwd_fused: cannot raise the LDS limit"); return CTCLIP_EBADARG; }
    raised = true;
  }
  const dim3 grid((unsigned)pl.nwg), block(NTH1);
  if (tab && dtab) hipLaunchKernelGGL((bwd1_kernel<true, true>), grid, block, pl.shm, stream, p, x, pl.g);
  else if (tab) hipLaunchKernelGGL((bwd1_kernel<true, false>), grid, block, pl.shm, stream, p, x, pl.g);
  else hipLaunchKernelGGL((bwd1_kernel<false, false>), grid, block, pl.shm, stream, p, x, pl.g);
  rc = ctclip_check_launch("attn2_bwd_fused");
  if (rc) return rc;
  hipLaunchKernelGGL(bwd1_scale_sum_kernel, dim3(2), dim3(1024), 0, stream, (const float*)p.kpart, (const float*)qpart, pl.nwg, dk_scale, dq_scale);
  rc = ctclip_check_launch("attn2_bwd_fused (scale sums)");
  if (rc || !dtab) return rc;
  hipLaunchKernelGGL(bwd1_dtab_sum_kernel, dim3((unsigned)cdiv((int64_t)pl.ncls * H, 64)), dim3(256), 0, stream, (const float*)dtpart, nseq, dtab, H, pl.ncls);
  return ctclip_check_launch("attn2_bwd_fused (table sum)");
}
