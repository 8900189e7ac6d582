// bf16 "NT" GEMM for gfx950: C[M x N] = alpha * A[M x K] * B[N x K]^T (+ bias) (+ residual) (+ C), both operands k-contiguous.
// Serves every forward linear (y = x W^T) and, with the transposed weight shadows, the input gradients (dx = dy (W^T)^T).
//
// Design, each point backed by a measured ablation on the FF in-projection (M=110592, N=2816, K=512; profiles/r01_gemm_ablation.md):
//
//  * FULL CACHE LINES PER LOAD.  A k-step is TK = 64 bf16 = 128 B per operand row = exactly one L2/TCP line.  The previous kernel
//    used TK = 32 (64-B rows): every line was fetched from L2 twice, in two different stages, and the loads alone (no MFMA, no
//    stores) ran at ~24 B/clk/CU of useful data, i.e. ~48 B/clk/CU of line traffic against the 64 B/clk/CU TCP port -- the loads
//    took as long as the MFMAs and could not hide under them.  It was not latency: neither a fifth stage nor L2-resident operands
//    changed the load-only time.
//  * RING OF FIVE 32-KiB PANELS (160 KiB, the whole LDS): a panel is one operand of one k-step (256 rows x 128 B).  Step t consumes
//    panels 2t (A) and 2t+1 (B); after the barrier of step t the two panels of step t-1 are free and are refilled with panels
//    2t+3 and 2t+4, so three panels (96 KiB) are in flight under every step's 64 MFMAs per wave.
//  * ONE CONTINUOUS PANEL STREAM ACROSS TILES: the persistent workgroup's loader does not stop at a tile boundary; the first three
//    panels of the next tile are already in flight when the epilogue of the current tile runs.
//  * global_load_lds (LDS-DMA, 16 B per lane) with COUNTED s_waitcnt vmcnt: vmcnt retires in order, so "all but the 4 youngest"
//    (= the pieces of the panel issued last) is exact; an epilogue's stores are older than that panel and are simply covered.
//  * NOTHING BUT MFMAs IN THE MFMA STREAM'S GAPS: every ds_read_b128 and every LDS-DMA piece is issued between two MFMAs of the
//    same wave (rolling in-place fragment reload, see the main loop); a batched "read 12 fragments, then 32 MFMAs" body left the
//    matrix pipe idle for both read phases of every step (MFMA + reads alone: 226 us against 160 us of pure MFMA issue).
//  * COALESCED EPILOGUE WITHOUT LDS: fragment b of the B operand holds tile columns li*8 + b (a free permutation: it is applied
//    to the per-lane SOURCE row of the LDS-DMA), so accumulator register r of fragment (a, b) is
//    C[a*16 + lg*4 + r][li*8 + b]: a lane owns eight consecutive columns (one 16-byte store in bf16), sixteen lanes two full
//    128-B lines, and one store instruction writes four complete row segments.  (The first version stored 16 partial lines per instruction and spent as long
//    in the epilogue as in the main loop.)
//  * LDS rows are 128 B = 8 chunks of 16 B; chunk' = chunk ^ ((row >> 1) & 7) makes every 16-lane service group of a
//    ds_read_b128 fragment read hit 16 distinct 16-B slots of the 256-B bank row (groups are {rows 0-3, 12-15} at chunk c plus
//    {rows 4-11} at chunk c^1, see MI355X LDS notes); the swizzle too is applied on the DMA source side.
//
// Tile 256 x 256, 8 waves (4 x 2), wave tile 64 x 128 = 4 x 8 fragments of mfma_f32_16x16x32_bf16, two k-sub-steps per step.
#include "common.h"
#include <type_traits>

// Compile-time ablation masks (tools/build_ablation.py; never set in the product build): 8 = no panel waits in the first two steps of a
// tile (timing only, wrong results), 1 = no global loads after the first
// prologue, 4 = epilogue unreachable, 64 = always non-temporal stores.  (Run-time switches are useless here:
// the compiler unswitches the loop and the extra branches perturb the production code.)
#ifndef NT_ABL
#define NT_ABL 0
#endif
#ifndef NT_STAGGER
#define NT_STAGGER 1     // 0 = off, 1 = phase by XCD, 2 = phase by groups of four workgroups
#endif
#ifndef NT_STREAM_MB
#define NT_STREAM_MB 128     // outputs larger than this use non-temporal stores
#endif
#ifndef NT_GELU_TAIL
#define NT_GELU_TAIL 1         // GEGLU forward epilogue: 1 = gelu_tail_fast (one transcendental), 0 = gelu_erf_fast (A/B builds)
#endif
#ifndef NT_GEGLU_ABL
#define NT_GEGLU_ABL 0       // ablations of the GEGLU forward epilogue: 1 = x * gate instead of x * gelu(gate), 2 = no stores
#endif
#ifndef NT_DG_ROWS
#define NT_DG_ROWS 2         // rows of u in flight per lane in the out-projection grad-input + GEGLU backward epilogue (1, 2 or 4)
#endif
#ifndef NT_PLANAR_TEST
#define NT_PLANAR_TEST 0
#endif
#ifndef NT_DEPHASE
#define NT_DEPHASE 0
#endif
#ifndef NT_COUNTED_EPI
// 1 = the first step of a tile waits with vmcnt(GL + n) for its panels only, not for the previous tile's epilogue stores (n = the VMEM
// instructions of that epilogue).  Measured +1..3 % on the GEMMs (profiles/r02_gemm_epilogue_experiments.md) and bit-identical results in
// 60 of 60 full-geometry training steps -- but it needs loads and stores to RETIRE IN ISSUE ORDER RELATIVE TO EACH OTHER (the panels it
// waits for are OLDER than the stores it no longer waits for), which the old vmcnt(GL) does not.  OFF until that order is settled (a rare
// run-to-run difference of the bf16 step was open then; located and fixed in round 3: HISTORY.md section 4).
#define NT_COUNTED_EPI 0
#endif

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int ROWB = 128;                       // bytes per LDS row (one k-step of one operand row)
constexpr int PANEL = 256 * ROWB;               // 32 KiB
constexpr int NPANEL = 5;
constexpr int NTH = 512;
constexpr int GL = 4;                           // LDS-DMA instructions per wave per panel (32 pieces of 1 KiB / 8 waves)

struct NtParams {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias; const void* residual;
  int64_t M, N, K, lda, ldb, ldc, ldr;
  int out_dtype, res_dtype, accumulate;
  float alpha;
  int ntm, ntn;
  // arg-max epilogue (vector-quantiser code search): no C; per (row, tile column half) partial (max, lowest index of the max)
  float* part_val; int32_t* part_idx; int nparts;
  int a_wrap;              // epilogue family 5 (ctclip_gemm_argmax_hilo): A has a_wrap k-steps and is read again from its first column for the k-steps behind them: [x | x] without the copy
  // GEGLU epilogue (feed-forward in-projection, attention.py:39-48): B's rows are interleaved in groups of four (output column
  // 8 q + r = x feature 4 q + r, 8 q + 4 + r = its gate), so a lane's eight consecutive columns are four (x, gate) pairs; the epilogue
  // stores u = [x | gate] in the split layout the backward reads AND g = x * gelu(gate).  geglu_hp = padded hidden width (0 = off).
  bf16_t* geglu_g; int64_t ldg; int geglu_hp;
  // GEGLU backward by recomputation (ctclip_gemm_geglu_bwd): the same GEMM recomputes (x, gate) in f32, the epilogue loads dg and
  // stores du = [dg * gelu(gate) | dg * x * gelu'(gate)] to C -- the forward then has no u to store and the backward no u to read.
  const bf16_t* geglu_dg; int64_t lddg;
  // Grad-input GEMM of the feed-forward OUT-projection with the GEGLU backward in its epilogue (ctclip_gemm_dgeglu): the accumulators
  // are dg = dy W_out (column j = hidden feature j); the epilogue loads u = [x | gate] (row stride dgeglu_ldu, gate at + dgeglu_hp) and
  // stores du = [dg * gelu(gate) | dg * x * gelu'(gate)] to C (gate half at + dgeglu_hp).  dg itself never reaches memory.
  const bf16_t* dgeglu_u; int64_t dgeglu_ldu; int dgeglu_hp;
  // Compensated residual stream (ctclip_gemm_residual_comp, epilogue family 3): the residual stream of the transformer is carried as a
  // bf16 pair (x, e): x = the rounded value every consumer reads, e = the rounding residue of the last add(s).  The epilogue forms
  // s = A B^T + residual + comp1 in f32 and stores C = bf16(s) AND comp_out = bf16(s - C): the 72 bf16 roundings of a 24-layer
  // residual stream no longer accumulate (profiles/r03_bf16_error_budget.md).  comp rows use ldr, comp_out rows use ldc.
  const bf16_t* comp1; bf16_t* comp_out;
  // Attention-operand epilogue (ctclip_gemm_headnorm, epilogue family 4): the q / k|v projections of the spatial attention write the
  // operands the attention kernels read -- head-planar [H][M][32] bf16 -- straight from the accumulators: per 256-column section s = n0 / 256
  // (8 heads of 32): hn_inv[s] != null: x~ = bf16(acc) / max(|row of the head|, 1e-12) * hn_scale[s][d] * hn_mult[s] and the inverse norm
  // to hn_inv[s][row * 8 + head] (exactly ctclip_attn2_prep's arithmetic on the bf16-rounded projection); null: plain copy (v).
  bf16_t* hn_out[3]; float* hn_inv[3]; const float* hn_scale[3]; float hn_mult[3];
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// a wave-uniform 64-bit value the compiler computed with VALU (there is no scalar 64-bit multiply) back into SGPRs
__device__ __forceinline__ const char* to_sgpr(const char* ptr) {
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

#ifndef STRICT_VMCNT
#define STRICT_VMCNT 0      // 1 = every counted vmcnt wait becomes vmcnt(0) (determinism bisection builds, tools/trace_determinism.py)
#endif
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(STRICT_VMCNT ? 0 : N) : "memory"); }

// Fragment reads go through inline asm with the kernel's own counted lgkmcnt waits (LDS returns in order): left to the compiler
// every half sub-step began with s_waitcnt lgkmcnt(0) right after two fresh reads were issued, i.e. exposed their latency.
// (gemm_tn.hip needs the asm form for a harder reason: see there.)  The waits carry the fragments as operands to pin the order.
template <int OFF> __device__ __forceinline__ u32x4 lds_read16(uint32_t vaddr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(vaddr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void wait_lds(u32x4& x, u32x4& y) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N)); }
template <int N> __device__ __forceinline__ void wait_lds3(u32x4& x, u32x4& y, u32x4& z) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(x), "+v"(y), "+v"(z) : "n"(N)); }

template <bool NONTEMPORAL>
__device__ __forceinline__ void store16(void* c, u32x4 d) {
  if (NONTEMPORAL) __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(c));
  else *reinterpret_cast<u32x4*>(c) = d;
}

// EPI selects the epilogue family compiled into the instantiation (each family alone: with all of them in one kernel the allocator
// hoisted lane addresses of every variant out of the tile loop and spilled them): 0 = plain / bias / residual / accumulate / f32 /
// partial tiles / arg-max, 1 = GEGLU forward and its recomputing backward, 2 = out-projection grad-input + GEGLU backward.
template <bool NONTEMPORAL, int EPI>
__global__ __launch_bounds__(NTH) void gemm_nt_kernel(NtParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int ntiles = p.ntm * p.ntn;
  const int nk = (int)(p.K / TK);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // 4 x 2 waves, wave tile 64 x 128
  const int li = lane & 15, lg = lane >> 4;

  // PERSISTENT: one workgroup per CU walks the tile list.  In round i the workgroups of XCD x (blockIdx % 8) take consecutive
  // tile ids, which share A row panels through that XCD's L2.
  const int G = gridDim.x;
#ifndef NT_XCD_CONTIG
#define NT_XCD_CONTIG 0
#endif
#if NT_XCD_CONTIG
  // Round 5 experiment, compiled OFF: every XCD owns ONE contiguous run of tile ids for the whole launch (ntiles / 8 of
  // them, +1 for the first ntiles % 8) and walks it G / 8 tiles per round.  The default map below deals ids [256 i + 32 x, + 32) to XCD x in
  // round i: with 11 column tiles per A row panel (the GEGLU in-projection) the 32-id windows straddle row panels and a straddled panel is
  // fetched by two XCDs' L2s.  Measured (profiles/r05_ab_experiments.md): FETCH 242 -> 214 MB per launch (2.08 x -> 1.84 x the algorithmic
  // 116 MB; traffic / algorithmic 1.12 -> 1.09) -- and the step 84.80 / 84.27 ms against 84.23 / 84.11 with the default map: the kernel is
  // bound by its main loop + store path (3.0 of 8 TB/s), not by what it fetches, and the contiguous runs give up the default map's balance
  // of the last round across XCDs.  Kept as a build option (tools/build_variant.py x gemm_nt.hip:NT_XCD_CONTIG=1).
  const int xcd = blockIdx.x & 7, xidx = blockIdx.x >> 3, xper = G >> 3;       // G is a multiple of 8
  const int t_per = ntiles >> 3, t_rem = ntiles & 7;
  const int x_start = xcd * t_per + (xcd < t_rem ? xcd : t_rem), x_cnt = t_per + (xcd < t_rem ? 1 : 0);
  auto tile_of = [&](int it, int64_t& m0, int64_t& n0) -> bool {
    const int local = it * xper + xidx;
    if (local >= x_cnt) return false;
    const int id = x_start + local;
    m0 = (int64_t)(id / p.ntn) * TM; n0 = (int64_t)(id % p.ntn) * TN;
    return true;
  };
#else
  const int slotb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);     // G is a multiple of 8
  auto tile_of = [&](int it, int64_t& m0, int64_t& n0) -> bool {
    const int id = it * G + slotb;
    if (id >= ntiles) return false;
    m0 = (int64_t)(id / p.ntn) * TM; n0 = (int64_t)(id % p.ntn) * TN;
    return true;
  };
#endif
  int64_t m0, n0;
  if (!tile_of(0, m0, n0)) return;

#if NT_STAGGER
  // De-synchronise the CUs: every tile takes the same time, so without this all 256 workgroups reach their epilogue together and
  // the chip writes 32 MB in one burst every tile period, then nothing.  Phase offsets of 1/8 tile period spread the stores.
  {
    const int phase = NT_STAGGER == 1 ? (blockIdx.x & 7) : (int)((blockIdx.x >> 2) & 7);
    for (int z = (phase * nk * 137) >> 10; z > 0; --z) __builtin_amdgcn_s_sleep(32);   // 32 * 64 clk; a step is ~2200 clk
  }
#endif
  // LDS row rho of the B panel holds tile column (rho & 0x80) + (rho & 15) * 8 + ((rho >> 4) & 7): fragment b of wave column wn
  // then covers columns li*8 + b and a lane's eight accumulator fragments are eight CONSECUTIVE output columns (see epilogue).
  // Loader cursors (plain locals so that they stay in registers: bases / counters in SGPRs, the four piece offsets in VGPRs).
  const char* a_base = nullptr; const char* b_base = nullptr;
  int a_it = 0, b_it = 0, a_t = 0, b_t = 0;
  uint32_t a_off[GL], b_off[GL];
  auto enter_a = [&](int it) {
    int64_t tm0, tn0;
    a_t = 0;
    if (!tile_of(it, tm0, tn0)) return;   // past the last tile: stay on it (harmless re-loads into free panels)
    a_it = it;
    a_base = to_sgpr(reinterpret_cast<const char*>(p.A) + tm0 * p.lda * 2);
    const int last = (int)(p.M - 1 - tm0);
#pragma unroll
    for (int j = 0; j < GL; ++j) {
      const int rho = (wave * GL + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
      const int trow = rho < last ? rho : last;   // rows past the end are clamped (never branch around a load); never stored
      a_off[j] = (uint32_t)trow * (uint32_t)(p.lda * 2) + (uint32_t)(chunk * 16);
    }
  };
  auto enter_b = [&](int it) {
    int64_t tm0, tn0;
    b_t = 0;
    if (!tile_of(it, tm0, tn0)) return;
    b_it = it;
    b_base = to_sgpr(reinterpret_cast<const char*>(p.B) + tn0 * p.ldb * 2);
    const int last = (int)(p.N - 1 - tn0);
#pragma unroll
    for (int j = 0; j < GL; ++j) {
      const int rho = (wave * GL + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
      int trow = (rho & 0x80) + ((rho & 15) << 3) + ((rho >> 4) & 7);
      trow = trow < last ? trow : last;
      b_off[j] = (uint32_t)trow * (uint32_t)(p.ldb * 2) + (uint32_t)(chunk * 16);
    }
  };
  auto glds = [&](const char* sbase, uint32_t voff, int slot, int j) {
#if !(NT_ABL & 1)
    const char* src = sbase + (uint64_t)voff;   // SGPR base + 32-bit lane offset
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + slot * PANEL + (wave * GL + j) * 1024), 16, 0, 0);
#endif
  };
  auto wrap = [](int s) { return s >= NPANEL ? s - NPANEL : s; };
  // byte offset of A's k-step t.  Family 5: the k-steps behind a_wrap read A from its first column again (scalar arithmetic; compiled into that family only)
  const int a_wrap = EPI == 5 ? p.a_wrap : 0;
  auto a_koff = [&](int t) -> int64_t { return (int64_t)((EPI == 5 && t >= a_wrap) ? t - a_wrap : t) * (TK * 2); };

  // fragment registers, reloaded IN PLACE as soon as their last MFMA of a sub-step has been issued
  u32x4 fa[4], fb[8];
  // lane part of the fragment address for ks = 0, 1 (the swizzle depends on li only; fragment f adds 16 rows = 2048 bytes)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const uint32_t fadr[2] = {(uint32_t)swz(li, lg), (uint32_t)swz(li, 4 + lg)};
  uint32_t pa[2], pb[2];
  auto point_a = [&](int slot) { pa[0] = fadr[0] + lds0 + (uint32_t)(slot * PANEL + wm * 8192); pa[1] = fadr[1] + lds0 + (uint32_t)(slot * PANEL + wm * 8192); };
  auto point_b = [&](int slot) { pb[0] = fadr[0] + lds0 + (uint32_t)(slot * PANEL + wn * 16384); pb[1] = fadr[1] + lds0 + (uint32_t)(slot * PANEL + wn * 16384); };
#define NT_RA(f, ks) fa[f] = lds_read16<(f) * 2048>(pa[ks]);
#define NT_RB(f, ks) fb[f] = lds_read16<(f) * 2048>(pb[ks]);

  // ---- prologue.  Ring position of A(g) is 2g, of B(g) 2g+1 (g = global k-step across tiles), slot = position % 5.
  enter_a(0); enter_b(0);
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(a_base, a_off[j], 0, j);                                  // A(0)
  if (++a_t == nk) enter_a(a_it + 1);
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(b_base, b_off[j], 1, j);                                  // B(0)
  if (++b_t == nk) enter_b(b_it + 1);
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(a_base + a_koff(a_t), a_off[j], 2, j);                    // A(1)
  if (++a_t == nk) enter_a(a_it + 1);
  wait_vm<GL>();
  __builtin_amdgcn_s_barrier();                                                               // "barrier_-1": step 0 is in LDS
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(b_base + (int64_t)b_t * (TK * 2), b_off[j], 3, j);        // B(1)
  if (++b_t == nk) enter_b(b_it + 1);
  point_a(0); point_b(1);
  NT_RA(0, 0) NT_RA(1, 0)
  NT_RB(0, 0) NT_RB(1, 0) NT_RB(2, 0) NT_RB(3, 0) NT_RB(4, 0) NT_RB(5, 0) NT_RB(6, 0) NT_RB(7, 0)
  int cs = 0;             // ring slot of A(g) for the consumer's current step g
  const bool dph = (wave & 4) != 0;      // (NT_DEPHASE) the second wave of every SIMD
  int pn = 0;             // VMEM instructions this wave issued in the previous tile's epilogue (exact or an under-count; 0 = unknown)
  auto wait_prev = [&](int n) {      // wave-uniform
    switch (n) {
      case 8: wait_vm<GL + 8>(); break;
      case 16: wait_vm<GL + 16>(); break;
      case 24: wait_vm<GL + 24>(); break;
      case 32: wait_vm<GL + 32>(); break;
      case 48: wait_vm<GL + 48>(); break;
      default: wait_vm<GL>();
    }
  };
  // GEGLU epilogues: a lane owns FOUR features of a row (8 bytes).  The store path costs ~60-85 clk per store INSTRUCTION per CU whatever
  // its width (measured: 128 dwordx2 stores of a tile take as long as 128 dwordx4 stores), so adjacent lanes swap one row of each row
  // pair and every lane writes 16 bytes of ONE row: even lanes the pair's first row, odd lanes the second.
  int odd_i = li & 1;
  auto pair_rows = [&](u32x2 d0, u32x2 d1) -> u32x4 {      // this lane's 8 bytes of rows r0 (d0), r0 + 1 (d1) -> the 16 bytes it stores
    const bool odd_lane = odd_i != 0;
    const u32x2 snd = odd_lane ? d0 : d1;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd[0], 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd[1], 0xB1, 0xF, 0xF, true);
    return odd_lane ? u32x4{r0, r1, d1[0], d1[1]} : u32x4{d0[0], d0[1], r0, r1};
  };

  for (int it = 0;; ++it) {
    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#define NT_MFMA2(a0, b)                                                                                                              \
    acc[a0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a0]), __builtin_bit_cast(bf16x8, fb[b]),      \
                                                         acc[a0][b], 0, 0, 0);                                                        \
    acc[a0 + 1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a0 + 1]), __builtin_bit_cast(bf16x8, fb[b]), \
                                                             acc[a0 + 1][b], 0, 0, 0);

    // Every fragment read and every LDS-DMA piece is issued BETWEEN MFMAs.  A sub-step (32 MFMAs per wave) runs as two halves:
    // rows a = 0,1 against all eight B fragments, then rows a = 2,3.  While the first half runs, fa[2], fa[3] of the same
    // sub-step are fetched; in the second half fa[0], fa[1] and, one by one as they die, fb[0..7] are re-loaded with the NEXT
    // sub-step's fragments.  No second register set is needed and no read phase is exposed.  Issue order of the reads:
    //   H1: A2 A3 | MFMA(a=0,1 ; b) needs A0 A1 B[b]      H2: A0' A1' | MFMA(a=2,3 ; b) needs A2 A3 ; then B[b]'
    //   H3: A2' A3' | MFMA(0,1) needs A0' A1' B'[b]       barrier      H4: A0'' A1'' | MFMA(2,3) ; then B[b]''
    // In H1 / H3 the reads younger than B[b] are B[b+1..7] and the two A fragments just issued: lgkmcnt(2 + 7 - b).
#define NT_H13(b) wait_lds3<2 + 7 - (b)>(fa[0], fa[1], fb[b]); NT_MFMA2(0, b)
    // One k-step (g = global step across tiles).  H1: (t, ks=0), rows a = 0,1 + the LDS-DMA of A(g+2); H2: (t, 0), rows a = 2,3, fetch
    // (t, ks=1); H3: (t, 1), rows a = 0,1; barrier_g: A(g+1), B(g+1) have landed (outstanding, oldest first: A(g+1), B(g+1), [the
    // previous tile's epilogue], A(g+2)) and every wave holds all of step g in registers; H4: (t, 1), rows a = 2,3, fetch (t+1, ks=0)
    // -- of the next tile after the last step -- and the LDS-DMA of B(g+2) into the slot of A(g), free since barrier_g.
    // WAITVM = the vmcnt wait in front of barrier_g.
    // NT_DEPHASE (round-5 experiment, compiled off): the two waves that share a SIMD (w and w + 4) issue their A-panel LDS-DMA pieces in different
    // half-phases -- waves 0-3 in H1 as always, waves 4-7 in H3 -- so that a SIMD's matrix pipe has one wave issuing MFMAs while the other is held by
    // its DMA issues (60-185 clk each).  Same vmcnt arithmetic: the newest GL pieces at barrier_g are A(g+2) for both classes.
#if NT_DEPHASE
#define NT_DMA_A1(i) if (!dph) glds(a_k, a_off[i], slot_a2, i);
#define NT_DMA_A3(i) if (dph) glds(a_k, a_off[i], slot_a2, i);
#define NT_ADV_A1
#define NT_ADV_A3 if (++a_t == nk) enter_a(a_it + 1);
#else
#define NT_DMA_A1(i) glds(a_k, a_off[i], slot_a2, i);
#define NT_DMA_A3(i)
#define NT_ADV_A1 if (++a_t == nk) enter_a(a_it + 1);
#define NT_ADV_A3
#endif
#define NT_STEP(WAITVM) { \
      const int slot_b2 = cs; \
      const int slot_a2 = wrap(cs + 4); \
      const int slot_na = wrap(cs + 2), slot_nb = wrap(cs + 3); \
      cs = wrap(cs + 2); \
      const char* a_k = a_base + a_koff(a_t); \
      NT_RA(2, 0) NT_RA(3, 0) \
      __builtin_amdgcn_sched_barrier(0); \
      NT_H13(0) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(1) NT_DMA_A1(0) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(2) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(3) NT_DMA_A1(1) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(4) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(5) NT_DMA_A1(2) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(6) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(7) NT_DMA_A1(3) __builtin_amdgcn_sched_barrier(0); \
      NT_ADV_A1 \
      NT_RA(0, 1) NT_RA(1, 1) \
      wait_lds<2>(fa[2], fa[3]); \
      __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 0) NT_RB(0, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 1) NT_RB(1, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 2) NT_RB(2, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 3) NT_RB(3, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 4) NT_RB(4, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 5) NT_RB(5, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 6) NT_RB(6, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 7) NT_RB(7, 1) __builtin_amdgcn_sched_barrier(0); \
      NT_RA(2, 1) NT_RA(3, 1) \
      __builtin_amdgcn_sched_barrier(0); \
      NT_H13(0) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(1) NT_DMA_A3(0) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(2) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(3) NT_DMA_A3(1) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(4) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(5) NT_DMA_A3(2) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(6) __builtin_amdgcn_sched_barrier(0); \
      NT_H13(7) NT_DMA_A3(3) __builtin_amdgcn_sched_barrier(0); \
      NT_ADV_A3 \
      WAITVM; \
      wait_lds<0>(fa[2], fa[3]); \
      __builtin_amdgcn_s_barrier(); \
      const char* b_k = b_base + (int64_t)b_t * (TK * 2); \
      point_a(slot_na); point_b(slot_nb); \
      NT_RA(0, 0) NT_RA(1, 0) \
      __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 0) NT_RB(0, 0) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 1) NT_RB(1, 0) glds(b_k, b_off[0], slot_b2, 0); __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 2) NT_RB(2, 0) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 3) NT_RB(3, 0) glds(b_k, b_off[1], slot_b2, 1); __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 4) NT_RB(4, 0) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 5) NT_RB(5, 0) glds(b_k, b_off[2], slot_b2, 2); __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 6) NT_RB(6, 0) __builtin_amdgcn_sched_barrier(0); \
      NT_MFMA2(2, 7) NT_RB(7, 0) glds(b_k, b_off[3], slot_b2, 3); __builtin_amdgcn_sched_barrier(0); \
      if (++b_t == nk) enter_b(b_it + 1); \
    }
    // The FIRST step of a tile is a separate copy of the body: the previous tile's epilogue issued `pn` VMEM instructions AFTER the
    // panels this step waits for and vmcnt retires in order, so "all but the GL + pn youngest" is the exact wait.  vmcnt(GL) made
    // every tile start by waiting for the COMPLETION of the previous tile's stores; now they have until barrier_(g+1).
    // pn = 0 (unknown / variable count: partial tiles, bias, arg-max) keeps the conservative wait; an under-count is always safe.
    int t = 0;
#if NT_ABL & 8      // TIMING ONLY (results are wrong): the first two steps of a tile do not wait for their panels at all -- what would the
    NT_STEP(wait_vm<63>())      // kernel cost if the epilogue stores' acknowledgements never held back the next tile's loads?
    NT_STEP(wait_vm<63>())
    t = 2;
#elif NT_COUNTED_EPI
    NT_STEP(wait_prev(pn))
    t = 1;
#endif
    for (; t < nk; ++t) NT_STEP(wait_vm<GL>())
#undef NT_STEP
#undef NT_H13
#undef NT_MFMA2
#undef NT_RA
#undef NT_RB

    // ---------------- epilogue, straight from registers: acc[a][b][r] = C[m0 + wm*64 + a*16 + lg*4 + r][n0 + wn*128 + li*8 + b]:
    // a lane owns 8 consecutive columns, 16 lanes one 256-B (bf16) row segment, one 16-byte store instruction writes 4 full rows
    // of the wave tile.  The stores are not waited for here: they retire under the next tile's first one and a half sub-steps.
    pn = 0;
    // the lane coordinates, opaque to the optimiser from here on: every per-lane address of the epilogue is then recomputed per tile
    // (a few VALU operations) -- hoisted out of the tile loop as loop invariants they lived across the main loop and were spilled
    // (and spilled; even li / lg themselves were: hence the lane id is read again from the hardware, inside a volatile asm)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int li_e = lane_e & 15, lg_e = lane_e >> 4;
    if (EPI == 5 || (EPI == 0 && p.part_val)) {     // kernel-uniform: row-wise arg-max over this wave's 128 columns; ties -> lowest column (torch.argmax on CPU)
      const int64_t col0 = n0 + wn * 128 + li_e * 8;
      float keep_v = -INFINITY; int keep_i = 0x7fffffff;     // lane li keeps the result of row (a, r) = (li >> 2, li & 3) of its lane group
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float best = -INFINITY; int bidx = 0x7fffffff;
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            const float v = acc[a][b][r];
            if (col0 + b < p.N && v > best) { best = v; bidx = (int)(col0 + b); }
          }
          // all-reduce over the 16 lanes of the row by DPP rotations (row_ror 8, 4, 2, 1: one VALU move each; the butterfly of __shfl_xor was
          // eight ds_bpermute round trips per row, 128 per wave and tile -- the epilogue cost as much as the tile's eight k-steps)
#define NT_ARG_STEP(CTRL) {                                                                                              \
            const float ov = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(best), CTRL, 0xF, 0xF, true));        \
            const int oi = __builtin_amdgcn_mov_dpp(bidx, CTRL, 0xF, 0xF, true);                                          \
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; } }
          NT_ARG_STEP(0x128) NT_ARG_STEP(0x124) NT_ARG_STEP(0x122) NT_ARG_STEP(0x121)
#undef NT_ARG_STEP
          if (li_e == a * 4 + r) { keep_v = best; keep_i = bidx; }   // (after the rotations every lane of the row holds the result)
        }
      // ONE pair of store instructions per wave and tile (sixteen rows per lane group at once; it was one pair per row: 32 instructions on a
      // store path that takes 60-85 clocks per instruction and CU)
      const int64_t row = m0 + wm * 64 + (li_e >> 2) * 16 + lg_e * 4 + (li_e & 3);
      if (row < p.M) {
        const int64_t slot = row * p.nparts + (n0 / TN) * 2 + wn;
        p.part_val[slot] = keep_v; p.part_idx[slot] = keep_i;
      }
    } else {
      const bool vec_ok = ((p.ldc % 8) == 0) && ((reinterpret_cast<uintptr_t>(p.C) % 16) == 0) &&
                          (!p.residual || (((p.ldr % 8) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) % 16) == 0)));
      const int64_t col = n0 + wn * 128 + li_e * 8;
      const int64_t rbase = m0 + wm * 64 + lg_e * 4;
      // (wave-uniform conditions.  A wave owns 128 of the tile's 256 columns: in the last, half-filled column tile of N = 1408 the
      // waves of the filled half keep the fast path and the others have nothing to store)
      const bool cols_in = n0 + wn * 128 + 128 <= p.N, cols_out = n0 + wn * 128 >= p.N;
      const bool fast = vec_ok && !p.residual && !p.accumulate && (m0 + TM <= p.M) && cols_in;
      // the two residual-stream GEMMs of every layer (to_out, feed-forward out-projection: bf16 in, bf16 residual, bf16 out): the residual
      // rows of the lane are requested eight at a time (the fragment registers are dead here), then added and stored -- the
      // general path below loads each row right before its use (one dependent round trip per row: +37 us on a 48-us launch)
      const bool fast_res = vec_ok && p.residual && p.res_dtype == DT_BF16 && p.out_dtype == DT_BF16 && !p.accumulate && !p.bias &&
                            (m0 + TM <= p.M) && cols_in;
      odd_i = li_e & 1;      // the lane-pair bit of the GEGLU epilogues
      const bool odd_lane = odd_i != 0;
#if NT_ABL & 4
      if (p.alpha == 1234.5f)
#endif
      if (EPI == 1 && p.geglu_hp && p.geglu_dg) {      // kernel-uniform; backward by recomputation
        const int64_t j0 = col >> 1;
        // this lane stores row (pair's first row + odd_lane) at the lane PAIR's eight features
        bf16_t* dup = reinterpret_cast<bf16_t*>(p.C) + (rbase + (odd_lane ? 1 : 0)) * p.ldc + (j0 - (odd_lane ? 4 : 0));
        const bf16_t* dgp = p.geglu_dg + rbase * p.lddg + j0;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {          // eight rows of dg in flight per lane
          u32x2 dv[2][4];
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[a2][r] = *reinterpret_cast<const u32x2*>(dgp + (int64_t)((2 * ah + a2) * 16 + r) * p.lddg);
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
              const int a = 2 * ah + a2;
              u32x2 ox[2], og[2];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const int r = 2 * rp + q;
                float dx[4], dgt[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                  const uint32_t w = dv[a2][r][b >> 1];
                  const float d = __uint_as_float((b & 1) ? (w & 0xffff0000u) : (w << 16));
                  const float x = acc[a][b][r] * p.alpha, gt = acc[a][4 + b][r] * p.alpha;
                  float y, dy;
                  gelu_erf_fast_both(gt, y, dy);
                  dx[b] = d * y; dgt[b] = d * x * dy;
                }
                ox[q] = u32x2{pack2bf(dx[0], dx[1]), pack2bf(dx[2], dx[3])}; og[q] = u32x2{pack2bf(dgt[0], dgt[1]), pack2bf(dgt[2], dgt[3])};
              }
              bf16_t* dst = dup + (int64_t)(a * 16 + 2 * rp) * p.ldc;
              store16<NONTEMPORAL>(dst, pair_rows(ox[0], ox[1]));
              store16<NONTEMPORAL>(dst + p.geglu_hp, pair_rows(og[0], og[1]));
            }
        }
        pn = 32;          // 16 loads of dg + 16 stores
      } else if (EPI == 1 && p.geglu_hp) {      // kernel-uniform; the launcher admits full tiles only.  u (p.C) is optional: training keeps only g
        const int64_t j0 = col >> 1;              // first of the lane's four features
        bf16_t* u = reinterpret_cast<bf16_t*>(p.C);
        // (two copies of the row loop, each ONE basic block: with the `u` test inside it every row was a block of its own and the
        // scheduler could not interleave the dependent chains of different rows)
        bf16_t* up = u + (rbase + (odd_lane ? 1 : 0)) * p.ldc + (j0 - (odd_lane ? 4 : 0));
        bf16_t* gp = p.geglu_g + (rbase + (odd_lane ? 1 : 0)) * p.ldg + (j0 - (odd_lane ? 4 : 0));
        auto rows = [&](auto with_u) {
          constexpr bool WU = decltype(with_u)::value;
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
              u32x2 ux[2], ug[2], gg[2];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const int r = 2 * rp + q;
                float x[4], gt[4], g[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                  x[b] = acc[a][b][r] * p.alpha; gt[b] = acc[a][4 + b][r] * p.alpha;
#if NT_GEGLU_ABL & 1
                  g[b] = x[b] * gt[b];
#else
                  g[b] = x[b] * (NT_GELU_TAIL ? gelu_tail_fast(gt[b]) : gelu_erf_fast(gt[b]));
#endif
                }
                gg[q] = u32x2{pack2bf(g[0], g[1]), pack2bf(g[2], g[3])};
                if (WU) { ux[q] = u32x2{pack2bf(x[0], x[1]), pack2bf(x[2], x[3])}; ug[q] = u32x2{pack2bf(gt[0], gt[1]), pack2bf(gt[2], gt[3])}; }
              }
#if NT_GEGLU_ABL & 2
              if (p.alpha == 1234.5f)
#endif
              {
                const int64_t ro = a * 16 + 2 * rp;
                if (WU) {
                  store16<NONTEMPORAL>(up + ro * p.ldc, pair_rows(ux[0], ux[1]));
                  store16<NONTEMPORAL>(up + ro * p.ldc + p.geglu_hp, pair_rows(ug[0], ug[1]));
                }
                store16<NONTEMPORAL>(gp + ro * p.ldg, pair_rows(gg[0], gg[1]));
              }
            }
        };
        if (u) rows(std::true_type{}); else rows(std::false_type{});
        pn = u ? 24 : 8;          // 8 row pairs x (x, gate, g) or g only, 16-byte stores
      } else if (EPI == 2 && p.dgeglu_u) {      // kernel-uniform; the launcher admits full row tiles and whole 128-column halves only
        if (!cols_out) {
          const bf16_t* up = p.dgeglu_u + rbase * p.dgeglu_ldu + col;
          bf16_t* dp = reinterpret_cast<bf16_t*>(p.C) + rbase * p.ldc + col;
#pragma unroll
          for (int a = 0; a < 4; ++a) {          // NT_DG_ROWS rows of u (x and gate: 16-byte loads) in flight per lane
#pragma unroll
            for (int rh = 0; rh < 4; rh += NT_DG_ROWS) {
              u32x4 xv[NT_DG_ROWS], gv[NT_DG_ROWS];
#pragma unroll
              for (int q = 0; q < NT_DG_ROWS; ++q) {
                xv[q] = *reinterpret_cast<const u32x4*>(up + (int64_t)(a * 16 + rh + q) * p.dgeglu_ldu);
                gv[q] = *reinterpret_cast<const u32x4*>(up + (int64_t)(a * 16 + rh + q) * p.dgeglu_ldu + p.dgeglu_hp);
              }
#pragma unroll
              for (int q = 0; q < NT_DG_ROWS; ++q) {
                const int r = rh + q;
                u32x4 ox, og;
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) {
                  float o1[2], o2[2];
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
                    const int b = 2 * b2 + h;
                    const float d = acc[a][b][r] * p.alpha;
                    const float x = __uint_as_float(h ? (xv[q][b2] & 0xffff0000u) : (xv[q][b2] << 16));
                    const float gt = __uint_as_float(h ? (gv[q][b2] & 0xffff0000u) : (gv[q][b2] << 16));
                    float y, dy;
                    gelu_erf_fast_both(gt, y, dy);
                    o1[h] = d * y; o2[h] = d * x * dy;
                  }
                  ox[b2] = pack2bf(o1[0], o1[1]); og[b2] = pack2bf(o2[0], o2[1]);
                }
                bf16_t* dst = dp + (int64_t)(a * 16 + r) * p.ldc;
                store16<NONTEMPORAL>(dst, ox);
                store16<NONTEMPORAL>(dst + p.dgeglu_hp, og);
              }
            }
          }
          pn = 48;          // 32 loads + 32 stores: more than vmcnt can count, an under-count is safe
        }
      } else if (EPI == 3) {      // residual + its compensation term in, rounded value + rounding residue out (full row tiles only: checked by the launcher)
       if (cols_in) {
        const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.residual) + rbase * p.ldr + col;
        const bf16_t* ep = p.comp1 + rbase * p.ldr + col;
#pragma unroll
        for (int a = 0; a < 4; ++a) {             // four rows of both tensors in flight per lane (32 registers; eight rows spill)
          u32x4 rr[4], ee[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t o = (int64_t)(a * 16 + r) * p.ldr;
            rr[r] = *reinterpret_cast<const u32x4*>(rp + o);
            ee[r] = *reinterpret_cast<const u32x4*>(ep + o);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = rbase + a * 16 + r;
            u32x4 d, e;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const uint32_t w = rr[r][b], u = ee[r][b];
              const float s0 = fmaf(acc[a][2 * b][r], p.alpha, __uint_as_float(w << 16)) + __uint_as_float(u << 16);
              const float s1 = fmaf(acc[a][2 * b + 1][r], p.alpha, __uint_as_float(w & 0xffff0000u)) + __uint_as_float(u & 0xffff0000u);
              const uint32_t y = pack2bf(s0, s1);
              d[b] = y;
              e[b] = pack2bf(s0 - __uint_as_float(y << 16), s1 - __uint_as_float(y & 0xffff0000u));
            }
            store16<NONTEMPORAL>(reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col, d);
            store16<NONTEMPORAL>(p.comp_out + row * p.ldc + col, e);
          }
        }
       }
        pn = 0;
      } else if (EPI == 4) {      // head-planar attention operands (full row tiles, N a multiple of 256: checked by the launcher)
        const int sec = (int)(n0 / TN);                          // tile-uniform: 0 = first 256 columns, ...
        bf16_t* outp = sec == 0 ? p.hn_out[0] : (sec == 1 ? p.hn_out[1] : p.hn_out[2]);
        float* invp = sec == 0 ? p.hn_inv[0] : (sec == 1 ? p.hn_inv[1] : p.hn_inv[2]);
        const float* scp = sec == 0 ? p.hn_scale[0] : (sec == 1 ? p.hn_scale[1] : p.hn_scale[2]);
        const float mult = sec == 0 ? p.hn_mult[0] : (sec == 1 ? p.hn_mult[1] : p.hn_mult[2]);
        const int head = wn * 4 + (li_e >> 2), d0 = (li_e & 3) * 8;      // a head = 32 consecutive columns = 4 lanes x 8
        float sc[8];
        if (invp) {
#pragma unroll
          for (int b = 0; b < 8; ++b) sc[b] = scp[d0 + b];
        }
        bf16_t* obase = outp + ((int64_t)head * p.M + rbase) * 32 + d0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t roff = (int64_t)(a * 16 + r);
            u32x4 d;
#pragma unroll
            for (int b = 0; b < 4; ++b) d[b] = pack2bf(acc[a][2 * b][r] * p.alpha, acc[a][2 * b + 1][r] * p.alpha);
            if (invp) {      // (tile-uniform)
              float v[8];
#pragma unroll
              for (int b = 0; b < 4; ++b) { v[2 * b] = __uint_as_float(d[b] << 16); v[2 * b + 1] = __uint_as_float(d[b] & 0xffff0000u); }
              float ss = 0.f;
#pragma unroll
              for (int b = 0; b < 8; ++b) ss += v[b] * v[b];
              ss = quad_sum(ss);
              const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
              for (int b = 0; b < 8; ++b) v[b] *= inv * sc[b] * mult;
#pragma unroll
              for (int b = 0; b < 4; ++b) d[b] = pack2bf(v[2 * b], v[2 * b + 1]);
              if ((li_e & 3) == 0) invp[(rbase + roff) * 8 + head] = inv;
            }
            store16<NONTEMPORAL>(obase + roff * 32, d);
          }
        pn = 0;
      } else if (EPI != 0) {
        // (the launchers set the parameters of the instantiation's own family)
      } else if (fast_res) {
        const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.residual) + rbase * p.ldr + col;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {          // eight rows in flight per lane (sixteen spilled: the accumulators own half the file)
          u32x4 rr[2][4];
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int r = 0; r < 4; ++r) rr[a2][r] = *reinterpret_cast<const u32x4*>(rp + (int64_t)((2 * ah + a2) * 16 + r) * p.ldr);
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int a = 2 * ah + a2;
              const int64_t row = rbase + a * 16 + r;
              u32x4 d;
#pragma unroll
              for (int b = 0; b < 4; ++b) {
                const uint32_t w = rr[a2][r][b];
                d[b] = pack2bf(fmaf(acc[a][2 * b][r], p.alpha, __uint_as_float(w << 16)), fmaf(acc[a][2 * b + 1][r], p.alpha, __uint_as_float(w & 0xffff0000u)));
              }
              store16<NONTEMPORAL>(reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col, d);
            }
        }
        pn = 32;          // 16 residual loads + 16 stores
      } else if (fast) {
        pn = p.bias ? 0 : (p.out_dtype == DT_F32 ? 32 : 16);
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias) load8(p.bias + col, bv);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = rbase + a * 16 + r;
            float v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = acc[a][b][r] * p.alpha + bv[b];
            if (p.out_dtype == DT_F32) {
              float* c = reinterpret_cast<float*>(p.C) + row * p.ldc + col;
              u32x4 d0, d1;
#pragma unroll
              for (int b = 0; b < 4; ++b) { d0[b] = __float_as_uint(v[b]); d1[b] = __float_as_uint(v[4 + b]); }
              store16<NONTEMPORAL>(c, d0); store16<NONTEMPORAL>(c + 4, d1);
            } else {
              u32x4 d;
#pragma unroll
              for (int b = 0; b < 4; ++b) d[b] = pack2bf(v[2 * b], v[2 * b + 1]);
#if NT_PLANAR_TEST      // TIMING EXPERIMENT ONLY (tools/build_variant.py ... gemm_nt.hip:NT_PLANAR_TEST=1): the head-planar layout [N / 32][M][32] instead
              // of token-major rows -- what would an attention-prep epilogue's 64-byte stores cost?  (profiles/r03_ab_experiments.md)
              store16<NONTEMPORAL>(reinterpret_cast<bf16_t*>(p.C) + ((col >> 5) * p.M + row) * 32 + (col & 31), d);
#else
              store16<NONTEMPORAL>(reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col, d);
#endif
            }
          }
      } else if (!cols_out) {
        const bool colfull = vec_ok && (col + 8 <= p.N);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = rbase + a * 16 + r;
            if (row >= p.M || col >= p.N) continue;
            float v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = acc[a][b][r] * p.alpha;
            if (p.bias) {
#pragma unroll
              for (int b = 0; b < 8; ++b) if (col + b < p.N) v[b] += p.bias[col + b];
            }
            if (colfull) {
              if (p.residual) {
                float rv[8];
                if (p.res_dtype == DT_F32) load8(reinterpret_cast<const float*>(p.residual) + row * p.ldr + col, rv);
                else load8(reinterpret_cast<const bf16_t*>(p.residual) + row * p.ldr + col, rv);
#pragma unroll
                for (int b = 0; b < 8; ++b) v[b] += rv[b];
              }
              if (p.out_dtype == DT_F32) {
                float* c = reinterpret_cast<float*>(p.C) + row * p.ldc + col;
                if (p.accumulate) { float old[8]; load8(c, old);
#pragma unroll
                  for (int b = 0; b < 8; ++b) v[b] += old[b]; }
                store8(c, v);
              } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col;
                if (p.accumulate) { float old[8]; load8(c, old);
#pragma unroll
                  for (int b = 0; b < 8; ++b) v[b] += old[b]; }
                store8(c, v);
              }
            } else {
#pragma unroll
              for (int b = 0; b < 8; ++b) {
                if (col + b >= p.N) continue;
                float x = v[b];
                if (p.residual)
                  x += (p.res_dtype == DT_F32) ? reinterpret_cast<const float*>(p.residual)[row * p.ldr + col + b]
                                               : bf2f(reinterpret_cast<const bf16_t*>(p.residual)[row * p.ldr + col + b]);
                if (p.out_dtype == DT_F32) {
                  float* c = reinterpret_cast<float*>(p.C) + row * p.ldc + col + b;
                  *c = p.accumulate ? *c + x : x;
                } else {
                  bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col + b;
                  *c = f2bf(p.accumulate ? bf2f(*c) + x : x);
                }
              }
            }
          }
      }
    }
    if (!tile_of(it + 1, m0, n0)) break;
  }   // persistent tile loop
  wait_vm<0>();   // the cursors ran ahead: no LDS-DMA may be outstanding when the workgroup releases its LDS
}

}  // namespace

// second form (gemm_nt2.hip): two 4-wave workgroups per CU, a tile's epilogue under the other workgroup's main loop
int ctclip_gemm_nt2_try(const void* A, const void* B, void* C, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                        int64_t ldc, int64_t ldr, float alpha, bool nontemporal, hipStream_t stream);
int ctclip_gemm_nt2_geglu_try(const void* A, const void* B, void* U, void* G, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb, int64_t ldu,
                              int64_t ldg, bool nontemporal, hipStream_t stream);
int ctclip_gemm_nt2_dgeglu_try(const void* A, const void* B, const void* U, void* dU, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb,
                               int64_t ldu, int64_t lddu, bool nontemporal, hipStream_t stream);

static int nt_launch(const NtParams& p, bool nontemporal, hipStream_t stream) {
  const int epi = p.geglu_hp ? 1 : (p.dgeglu_u ? 2 : (p.comp_out ? 3 : (p.hn_out[0] ? 4 : (p.a_wrap ? 5 : 0))));
  static bool raised = false;
  if (!raised) {
    const void* fns[11] = {(const void*)gemm_nt_kernel<false, 0>, (const void*)gemm_nt_kernel<true, 0>, (const void*)gemm_nt_kernel<false, 1>,
                           (const void*)gemm_nt_kernel<true, 1>, (const void*)gemm_nt_kernel<false, 2>, (const void*)gemm_nt_kernel<true, 2>,
                           (const void*)gemm_nt_kernel<false, 3>, (const void*)gemm_nt_kernel<true, 3>, (const void*)gemm_nt_kernel<false, 4>,
                           (const void*)gemm_nt_kernel<true, 4>, (const void*)gemm_nt_kernel<false, 5>};
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, NPANEL * PANEL) != hipSuccess) return 1;
    raised = true;
  }
  static int ncu = 0;
  if (!ncu) { int dev = 0; hipDeviceProp_t prop; (void)hipGetDevice(&dev); ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) ? (prop.multiProcessorCount & ~7) : 256; }
  const dim3 grid((unsigned)ncu), block(NTH);
#define NT_GO(E) do { if (nontemporal) hipLaunchKernelGGL((gemm_nt_kernel<true, E>), grid, block, NPANEL * PANEL, stream, p); \
                      else hipLaunchKernelGGL((gemm_nt_kernel<false, E>), grid, block, NPANEL * PANEL, stream, p); } while (0)
  if (epi == 1) NT_GO(1); else if (epi == 2) NT_GO(2); else if (epi == 3) NT_GO(3); else if (epi == 4) NT_GO(4);
  else if (epi == 5) hipLaunchKernelGGL((gemm_nt_kernel<false, 5>), grid, block, NPANEL * PANEL, stream, p);
  else NT_GO(0);
#undef NT_GO
  return ctclip_check_launch("gemm_nt");
}

// Row-wise arg-max of A B^T on the same kernel (ctclip_gemm_argmax, bf16): partials (M x nparts), nparts = 2 * ceil(N / 256).
// Returns 1 when the shape is not eligible.
int ctclip_gemm_nt_argmax_try(const void* A, const void* B, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, float* part_val,
                              int32_t* part_idx, int* nparts, hipStream_t stream, int64_t a_wrap_k) {
  if (K % TK || K / TK < 2) return 1;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;
  const int64_t ntm = cdiv(M, TM), ntn = cdiv(N, TN);
  if (ntm * ntn < 160) return 1;
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.alpha = 1.f;
  p.ntm = (int)ntm; p.ntn = (int)ntn;
  if (a_wrap_k && (a_wrap_k % TK || 2 * a_wrap_k != K)) return 1;      // [x | x]: exactly two passes over A
  p.part_val = part_val; p.part_idx = part_idx; p.nparts = (int)(2 * ntn); p.a_wrap = (int)(a_wrap_k / TK);
  *nparts = p.nparts;
  return nt_launch(p, false, stream);
}

// Internal entry used by ctclip_gemm's dispatcher (gemm.hip).  Returns 1 when the shape is not eligible.
int ctclip_gemm_nt_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int out_dtype, int res_dtype, int accumulate, float alpha,
                       hipStream_t stream) {
  if (K % TK || K / TK < 2) return 1;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;   // 32-bit in-panel byte offsets
  if (bias && (reinterpret_cast<uintptr_t>(bias) % 16)) return 1;
  const int64_t ntm = cdiv(M, TM), ntn = cdiv(N, TN);
  if (ntm * ntn < 160) return 1;   // needs to fill the chip: small problems stay on the other kernels
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.out_dtype = out_dtype; p.res_dtype = res_dtype; p.accumulate = accumulate; p.alpha = alpha;
  p.ntm = (int)ntm; p.ntn = (int)ntn;
  // outputs that cannot stay in the 32 MiB of L2 anyway are written with the non-temporal hint (measured -9 % on the 623-MB FF
  // hidden activation: they no longer evict the operand panels the other CUs of the XCD are about to re-read)
  const bool nontemporal = ((NT_ABL & 64) != 0) || (M * N * (out_dtype == DT_F32 ? 4 : 2) > ((int64_t)NT_STREAM_MB << 20));
  if (!bias && !accumulate && out_dtype == DT_BF16 && (!residual || res_dtype == DT_BF16)) {
    const int rc2 = ctclip_gemm_nt2_try(A, B, C, residual, M, N, K, lda, ldb, ldc, ldr, alpha, nontemporal, stream);
    if (rc2 != 1) return rc2;
  }
  return nt_launch(p, nontemporal, stream);
}

// out = A B^T + residual + comp, stored as the bf16 pair (C, E = the rounding residue): the residual adds of attention.py:326,331 on a
// compensated residual stream.  Whole 256-row tiles, N a multiple of 128.  Returns 1 when the shape is not eligible.
int ctclip_gemm_nt_rescomp_try(const void* A, const void* B, void* C, void* E, const void* residual, const void* comp,
                               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, hipStream_t stream) {
  if (K % TK || K / TK < 2 || M % TM || N % 128 || ldc % 8 || ldr % 8) return 1;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  for (const void* q : {(const void*)C, (const void*)E, residual, comp})
    if (reinterpret_cast<uintptr_t>(q) % 16) return 1;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;
  const int64_t ntm = M / TM, ntn = cdiv(N, TN);
  if (ntm * ntn < 160) return 1;
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.residual = residual; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.out_dtype = DT_BF16; p.res_dtype = DT_BF16; p.alpha = 1.f; p.ntm = (int)ntm; p.ntn = (int)ntn;
  p.comp1 = (const bf16_t*)comp; p.comp_out = (bf16_t*)E;
  return nt_launch(p, 2 * M * N * 2 > ((int64_t)NT_STREAM_MB << 20), stream);
}

// The q / k|v projections of the spatial attention with ctclip_attn2_prep folded into the epilogue: the N = nsec * 256 output columns go, per
// 256-column section, to out[s] as head-planar [8][M][32] bf16 (normalised + scaled when inv[s] is given, copied otherwise).  Returns 1 when
// the shape is not eligible.
int ctclip_gemm_nt_headnorm_try(const void* A, const void* B, int64_t M, int nsec, int64_t K, int64_t lda, int64_t ldb, void* const* out,
                                float* const* inv, const float* const* scale, const float* mult, hipStream_t stream) {
  const int64_t N = (int64_t)nsec * TN;
  if (nsec < 1 || nsec > 3 || K % TK || K / TK < 2 || M % TM) return 1;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;
  const int64_t ntm = M / TM, ntn = nsec;
  if (ntm * ntn < 160) return 1;
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
  p.out_dtype = DT_BF16; p.alpha = 1.f; p.ntm = (int)ntm; p.ntn = (int)ntn;
  for (int s = 0; s < 3; ++s) {
    const int t = s < nsec ? s : 0;
    if (!out[t] || (reinterpret_cast<uintptr_t>(out[t]) % 16) || (inv[t] && !scale[t])) return 1;
    p.hn_out[s] = (bf16_t*)out[t]; p.hn_inv[s] = inv[t]; p.hn_scale[s] = scale[t]; p.hn_mult[s] = mult[t];
  }
  return nt_launch(p, M * N * 2 > ((int64_t)NT_STREAM_MB << 20), stream);
}

// Feed-forward in-projection with the GEGLU fused into the epilogue (attention.py:39-48).  B = the in-projection weight with its
// rows interleaved in groups of four (ctclip_geglu_weight_interleave), N = 2 * hp.  Forward (dG == nullptr): writes g (M, ldg >= hp) =
// x * gelu(gate) and, when U is given, u (M, ldu >= 2 hp) = [x | gate], all bf16.  Backward by recomputation (dG given): writes
// U = du = [dG * gelu(gate) | dG * x * gelu'(gate)] from the recomputed (x, gate).  Returns 1 when the shape is not eligible.
int ctclip_gemm_nt_geglu_try(const void* A, const void* B, void* U, void* G, const void* dG, int64_t M, int hp, int64_t K, int64_t lda,
                             int64_t ldb, int64_t ldu, int64_t ldg, int64_t lddg, hipStream_t stream) {
  const int64_t N = 2 * (int64_t)hp;
  if (K % TK || K / TK < 2 || M % TM || N % TN || hp % 8 || ldu % 8 || ldg % 8 || lddg % 4) return 1;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  if ((reinterpret_cast<uintptr_t>(U) % 16) || (reinterpret_cast<uintptr_t>(G) % 16) || (reinterpret_cast<uintptr_t>(dG) % 8)) return 1;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;
  const int64_t ntm = M / TM, ntn = N / TN;
  if (ntm * ntn < 160) return 1;
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = U; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldu;
  p.out_dtype = DT_BF16; p.alpha = 1.f; p.ntm = (int)ntm; p.ntn = (int)ntn;
  p.geglu_g = (bf16_t*)G; p.ldg = ldg; p.geglu_hp = hp;
  p.geglu_dg = (const bf16_t*)dG; p.lddg = lddg;
  const int64_t out_bytes = (U ? M * N * 2 : 0) + (G ? M * (int64_t)hp * 2 : 0);
  if (!dG && G) {
    const int rc2 = ctclip_gemm_nt2_geglu_try(A, B, U, G, M, hp, K, lda, ldb, ldu, ldg, out_bytes > ((int64_t)NT_STREAM_MB << 20), stream);
    if (rc2 != 1) return rc2;
  }
  return nt_launch(p, out_bytes > ((int64_t)NT_STREAM_MB << 20), stream);
}

// Grad-input GEMM of the feed-forward out-projection + GEGLU backward (attention.py:39-51, backward): dU (M, lddu >= 2 hp) =
// [dg * gelu(gate) | dg * x * gelu'(gate)] with dg = A B^T (A = dy (M, K), B = the out-projection weight TRANSPOSED (hp, K), hidden
// feature j in row j) and u = [x | gate] (M, ldu >= 2 hp) the tensor ctclip_gemm_geglu stored.  Returns 1 when the shape is not eligible.
int ctclip_gemm_nt_dgeglu_try(const void* A, const void* B, const void* U, void* dU, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb,
                              int64_t ldu, int64_t lddu, hipStream_t stream) {
  const int64_t N = hp;
  if (K % TK || K / TK < 2 || M % TM || N % 128 || ldu % 8 || lddu % 8) return 1;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  if ((reinterpret_cast<uintptr_t>(U) % 16) || (reinterpret_cast<uintptr_t>(dU) % 16)) return 1;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;
  const int64_t ntm = M / TM, ntn = cdiv(N, TN);
  if (ntm * ntn < 160) return 1;
  NtParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = dU; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = lddu;
  p.out_dtype = DT_BF16; p.alpha = 1.f; p.ntm = (int)ntm; p.ntn = (int)ntn;
  p.dgeglu_u = (const bf16_t*)U; p.dgeglu_ldu = ldu; p.dgeglu_hp = hp;
  {
    const int rc2 = ctclip_gemm_nt2_dgeglu_try(A, B, U, dU, M, hp, K, lda, ldb, ldu, lddu, M * 2 * N * 2 > ((int64_t)NT_STREAM_MB << 20), stream);
    if (rc2 != 1) return rc2;
  }
  return nt_launch(p, M * 2 * N * 2 > ((int64_t)NT_STREAM_MB << 20), stream);
}
