// bf16 "NT" GEMM, second form (round 5): C[M x N] = A[M x K] * B[N x K]^T with the EPILOGUE OF ONE TILE UNDER THE MAIN LOOP OF ANOTHER.
//
// gemm_nt.hip runs one workgroup of eight waves per CU on 256 x 256 tiles with the whole LDS as its panel ring: its matrix pipe idles while a
// tile's 128-192 KB of output leave through the CU's store path (16 B/clk: 3.9-5.0 us of every 16.5-17.6-us tile round, measured in rounds
// 2-4) and the store path idles during the main loop.  Deferred stores do not fit its register budget (206 live of 256).
// Here a CU holds TWO independent workgroups of four waves (one per SIMD each), each with half the LDS (80 KB) and its own tile stream:
// while one of them stores, the other one's MFMAs have the matrix pipe to themselves; while both are in their main loops they share it as
// the two waves per SIMD of the first form do.  The hardware's wave scheduler does the overlap -- no deferred stores, no second
// accumulator set.
//
//   * tile 192 x 128, k-step 64 (one full 128-B line per operand row, as in the first form): a step is 24 KB of A + 16 KB of B, two
//     stages = 80 KB.  192 divides the token counts of the image tower (13 824 per volume); N = 128 keeps the "lane owns eight
//     consecutive columns" epilogue (one 16-byte store per row and lane).  Per 256 x 256-equivalent of work a CU loads 1.67x the operand
//     bytes of the first form (both from L2 in the steady state); whether the vector-memory port carries that was argued on paper for two
//     rounds (HISTORY.md section 8) -- this kernel is the measurement.
//   * waves 4 x 1: wave w owns rows [48 w, 48 w + 48) x all 128 columns = 3 x 8 fragments of mfma_f32_16x16x32_bf16 (96 accumulators).
//   * double buffering with ONE barrier per k-step, placed after the first third of the step's second sub-step: the remaining 16 MFMAs
//     cover the fragment reads of the next step and the issue of the step after next's LDS-DMA pieces.  The panel stream is continuous
//     across tiles (the loader's cursors run ahead of the consumer by two steps).
//   * fragment reads through inline asm with explicit lgkmcnt waits (two register sets, no in-place rolling: 184 registers), LDS layout,
//     swizzle, B-row permutation and the register-direct epilogues are those of gemm_nt.hip.
// Eligible shapes: M % 192 == 0, N % 128 == 0, K % 64 == 0 (>= 2 steps), bf16 in / bf16 out, no bias; epilogue families: plain / residual,
// GEGLU forward, out-projection grad-input + GEGLU backward.  Everything else stays on gemm_nt.hip.  CTCLIP_GEMM_NT2 / ctclip_gemm_nt2_select choose the
// families (bit 0 plain / residual, bit 1 GEGLU forward, bit 2 GEGLU backward).  DEFAULT 0: MEASURED SLOWER than the first form on every family
// (1.02 - 1.43 x per launch, profiles/r05_gemm_nt2.md) -- the kernel stays as the measurement, bit-identical to gemm_nt.hip.
#include "common.h"
#ifndef NT_GELU_TAIL
#define NT_GELU_TAIL 1         // as in gemm_nt.hip
#endif
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 192, BN = 128, TK = 64;
constexpr int ROWB = 128;                        // bytes per LDS row (one k-step of one operand row)
constexpr int A_BYTES = BM * ROWB;               // 24 KiB
constexpr int B_BYTES = BN * ROWB;               // 16 KiB
constexpr int STAGE = A_BYTES + B_BYTES;         // 40 KiB
constexpr int LDS_BYTES = 2 * STAGE;             // 80 KiB: two workgroups per CU
constexpr int NTH = 256;
#ifndef NT2_DEFAULT_MASK
#define NT2_DEFAULT_MASK 0      // (set from the same-box A/B of the families, profiles/r05_gemm_nt2.md)
#endif
constexpr int GA = 6, GB = 4;                    // LDS-DMA pieces (1 KiB = 8 rows) per wave and step: A 24 / 4 waves, B 16 / 4

struct Nt2Params {
  const bf16_t* A; const bf16_t* B; void* C; const void* residual;
  int64_t M, N, K, lda, ldb, ldc, ldr;
  float alpha;
  int ntm, ntn;
  bf16_t* geglu_g; int64_t ldg; int geglu_hp;                   // EPI 1 (see gemm_nt.hip NtParams)
  const bf16_t* dgeglu_u; int64_t dgeglu_ldu; int dgeglu_hp;    // EPI 2
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ const char* to_sgpr(const char* ptr) {
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int OFF> __device__ __forceinline__ u32x4 lds_read16(uint32_t vaddr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(vaddr), "n"(OFF));
  return v;
}
// all outstanding LDS reads have returned; the fragments of one register set are the operands (pins the order against their MFMAs)
__device__ __forceinline__ void wait_set(u32x4 (&a)[3], u32x4 (&b)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}

template <bool NONTEMPORAL>
__device__ __forceinline__ void store16(void* c, u32x4 d) {
  if (NONTEMPORAL) __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(c));
  else *reinterpret_cast<u32x4*>(c) = d;
}

// EPI: 0 = plain / residual, 1 = GEGLU forward, 2 = out-projection grad-input + GEGLU backward
template <bool NONTEMPORAL, int EPI>
__global__ __launch_bounds__(NTH, 2) void gemm_nt2_kernel(Nt2Params p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int ntiles = p.ntm * p.ntn;
  const int nk = (int)(p.K / TK);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;

  // PERSISTENT: 2 workgroups per CU walk the tile list; in round i the workgroups of XCD x (blockIdx % 8) take consecutive tile ids (the
  // column tiles of a few row panels: they share A rows through that XCD's L2)
  const int G = gridDim.x;
  const int slotb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);     // G is a multiple of 8
  auto tile_of = [&](int it, int64_t& m0, int64_t& n0) -> bool {
    const int id = it * G + slotb;
    if (id >= ntiles) return false;
    m0 = (int64_t)(id / p.ntn) * BM; n0 = (int64_t)(id % p.ntn) * BN;
    return true;
  };
  int64_t m0, n0;
  if (!tile_of(0, m0, n0)) return;

  // De-synchronise: the second workgroup of every CU (the dispatcher fills each CU once before it doubles up) starts half a tile late, so
  // that its main loop sits under the first one's epilogue; XCDs are phased by eighths of a tile as in the first form (store bursts).
  {
    const int half = blockIdx.x >= (G >> 1) ? 1 : 0;      // (s_sleep 32 = 2048 clk; a k-step of one workgroup is ~3000 clk while both share the pipe)
    for (int z = (half * nk * 1500 + (int)(blockIdx.x & 7) * nk * 190) >> 11; z > 0; --z) __builtin_amdgcn_s_sleep(32);
  }

  // ---- loader state: SGPR bases + per-piece lane offsets (invariant: whole tiles only)
  const char* a_base = nullptr; const char* b_base = nullptr;
  int ld_it = 0, ld_t = 0;
  uint32_t a_off[GA], b_off[GB];
#pragma unroll
  for (int j = 0; j < GA; ++j) {
    const int rho = (wave * GA + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
    a_off[j] = (uint32_t)rho * (uint32_t)(p.lda * 2) + (uint32_t)(chunk * 16);
  }
#pragma unroll
  for (int j = 0; j < GB; ++j) {
    const int rho = (wave * GB + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
    const int trow = ((rho & 15) << 3) + (rho >> 4);          // LDS row rho of the B stage holds tile column (rho & 15) * 8 + (rho >> 4)
    b_off[j] = (uint32_t)trow * (uint32_t)(p.ldb * 2) + (uint32_t)(chunk * 16);
  }
  auto enter = [&](int it) {
    int64_t tm0, tn0;
    ld_t = 0;
    if (!tile_of(it, tm0, tn0)) return;      // past the last tile: stay on it (harmless re-loads into free stages)
    ld_it = it;
    a_base = to_sgpr(reinterpret_cast<const char*>(p.A) + tm0 * p.lda * 2);
    b_base = to_sgpr(reinterpret_cast<const char*>(p.B) + tn0 * p.ldb * 2);
  };
  auto glds = [&](const char* sbase, uint32_t voff, int ldsoff) {
    const char* src = sbase + (uint64_t)voff;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + ldsoff), 16, 0, 0);
  };
  // piece i of this wave's ten for the step the cursor points at, into stage `st`
#define NT2_D(i, st)                                                                                                       \
  do {                                                                                                                     \
    if ((i) < GA) glds(a_base + (int64_t)ld_t * (TK * 2), a_off[(i) < GA ? (i) : 0], (st) * STAGE + (wave * GA + (i)) * 1024);            \
    else glds(b_base + (int64_t)ld_t * (TK * 2), b_off[(i) >= GA ? (i) - GA : 0], (st) * STAGE + A_BYTES + (wave * GB + (i) - GA) * 1024); \
  } while (0)
#define NT2_ADVANCE() do { if (++ld_t == nk) enter(ld_it + 1); } while (0)

  // ---- fragment addressing
  u32x4 fa[2][3], fb[2][8];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  uint32_t pa[2], pb[2];
  pa[0] = lds0 + (uint32_t)(wave * 48 * ROWB) + (uint32_t)swz(li, lg); pa[1] = lds0 + (uint32_t)(wave * 48 * ROWB) + (uint32_t)swz(li, 4 + lg);
  pb[0] = lds0 + (uint32_t)A_BYTES + (uint32_t)swz(li, lg); pb[1] = lds0 + (uint32_t)A_BYTES + (uint32_t)swz(li, 4 + lg);
#define NT2_RA(s, f, ks) fa[s][f] = lds_read16<(f) * 2048>(pa[ks]);
#define NT2_RB(s, f, ks) fb[s][f] = lds_read16<(f) * 2048>(pb[ks]);
#define NT2_SB __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: steps 0 and 1 in flight, step 0 landed, its first sub-step's fragments requested
  enter(0);
#pragma unroll
  for (int i = 0; i < GA + GB; ++i) NT2_D(i, 0);
  NT2_ADVANCE();
#pragma unroll
  for (int i = 0; i < GA + GB; ++i) NT2_D(i, 1);
  NT2_ADVANCE();
  wait_vm<GA + GB>();
  __builtin_amdgcn_s_barrier();
  NT2_RA(0, 0, 0) NT2_RA(0, 1, 0) NT2_RA(0, 2, 0)
  NT2_RB(0, 0, 0) NT2_RB(0, 1, 0) NT2_RB(0, 2, 0) NT2_RB(0, 3, 0) NT2_RB(0, 4, 0) NT2_RB(0, 5, 0) NT2_RB(0, 6, 0) NT2_RB(0, 7, 0)
  int cur = 0;            // stage of the consumer's current step

  int odd_i = li & 1;
  auto pair_rows = [&](u32x2 d0, u32x2 d1) -> u32x4 {      // (gemm_nt.hip: adjacent lanes swap one row of each row pair -> 16-byte stores)
    const bool odd_lane = odd_i != 0;
    const u32x2 snd = odd_lane ? d0 : d1;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd[0], 0xB1, 0xF, 0xF, true);
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd[1], 0xB1, 0xF, 0xF, true);
    return odd_lane ? u32x4{r0, r1, d1[0], d1[1]} : u32x4{d0[0], d0[1], r0, r1};
  };

  for (int it = 0;; ++it) {
    f32x4 acc[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#define NT2_MM(s, a, b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[s][a]), __builtin_bit_cast(bf16x8, fb[s][b]), acc[a][b], 0, 0, 0);
#define NT2_COL(s, b) NT2_MM(s, 0, b) NT2_MM(s, 1, b) NT2_MM(s, 2, b)
#define NT2_COL12(s, b) NT2_MM(s, 1, b) NT2_MM(s, 2, b)
    for (int t = 0; t < nk; ++t) {
      // S0: sub-step 0 from register set 0; set 1 <- sub-step 1 of the same stage
      wait_set(fa[0], fb[0]);
      NT2_RA(1, 0, 1) NT2_RA(1, 1, 1) NT2_SB
      NT2_COL(0, 0) NT2_RA(1, 2, 1) NT2_SB
      NT2_COL(0, 1) NT2_RB(1, 0, 1) NT2_SB
      NT2_COL(0, 2) NT2_RB(1, 1, 1) NT2_RB(1, 2, 1) NT2_SB
      NT2_COL(0, 3) NT2_RB(1, 3, 1) NT2_RB(1, 4, 1) NT2_SB
      NT2_COL(0, 4) NT2_RB(1, 5, 1) NT2_RB(1, 6, 1) NT2_SB
      NT2_COL(0, 5) NT2_RB(1, 7, 1) NT2_SB
      NT2_COL(0, 6) NT2_SB
      NT2_COL(0, 7) NT2_SB
      // S1a: first fragment row of sub-step 1
      wait_set(fa[1], fb[1]);
      NT2_MM(1, 0, 0) NT2_MM(1, 0, 1) NT2_MM(1, 0, 2) NT2_MM(1, 0, 3) NT2_MM(1, 0, 4) NT2_MM(1, 0, 5) NT2_MM(1, 0, 6) NT2_MM(1, 0, 7) NT2_SB
      // barrier: the next step's stage has landed (this wave's pieces: vmcnt; the others': the barrier) and every wave has READ the current
      // stage completely (lgkmcnt(0) above, no LDS operation since): it is free for the step after next
      wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      {
        const uint32_t d = cur ? (uint32_t)(-STAGE) : (uint32_t)STAGE;
        pa[0] += d; pa[1] += d; pb[0] += d; pb[1] += d;
      }
      // S1b: the other two fragment rows; set 0 <- sub-step 0 of the next stage; the LDS-DMA of the step after next into the freed stage
      NT2_RA(0, 0, 0) NT2_RA(0, 1, 0) NT2_SB
      NT2_COL12(1, 0) NT2_RA(0, 2, 0) NT2_D(0, cur); NT2_SB
      NT2_COL12(1, 1) NT2_RB(0, 0, 0) NT2_D(1, cur); NT2_SB
      NT2_COL12(1, 2) NT2_RB(0, 1, 0) NT2_D(2, cur); NT2_SB
      NT2_COL12(1, 3) NT2_RB(0, 2, 0) NT2_D(3, cur); NT2_SB
      NT2_COL12(1, 4) NT2_RB(0, 3, 0) NT2_D(4, cur); NT2_SB
      NT2_COL12(1, 5) NT2_RB(0, 4, 0) NT2_D(5, cur); NT2_SB
      NT2_COL12(1, 6) NT2_RB(0, 5, 0) NT2_RB(0, 6, 0) NT2_D(6, cur); NT2_D(7, cur); NT2_SB
      NT2_COL12(1, 7) NT2_RB(0, 7, 0) NT2_D(8, cur); NT2_D(9, cur); NT2_SB
      NT2_ADVANCE();
      cur ^= 1;
    }
#undef NT2_COL12
#undef NT2_COL
#undef NT2_MM

    // ---------------- epilogue, straight from registers: acc[a][b][r] = C[m0 + wave*48 + a*16 + lg*4 + r][n0 + li*8 + b]
    int lane_e;      // (lane coordinates re-read from the hardware inside a volatile asm: per-lane addresses are recomputed per tile, not hoisted and spilled)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int li_e = lane_e & 15, lg_e = lane_e >> 4;
    const int64_t col = n0 + li_e * 8;
    const int64_t rbase = m0 + wave * 48 + lg_e * 4;
    odd_i = li_e & 1;
    const bool odd_lane = odd_i != 0;
    if (EPI == 1) {              // GEGLU forward: u = [x | gate] (optional) and g = x gelu(gate)
      const int64_t j0 = col >> 1;
      bf16_t* u = reinterpret_cast<bf16_t*>(p.C);
      bf16_t* up = u + (rbase + (odd_lane ? 1 : 0)) * p.ldc + (j0 - (odd_lane ? 4 : 0));
      bf16_t* gp = p.geglu_g + (rbase + (odd_lane ? 1 : 0)) * p.ldg + (j0 - (odd_lane ? 4 : 0));
      auto rows = [&](auto with_u) {
        constexpr bool WU = decltype(with_u)::value;
#pragma unroll
        for (int a = 0; a < 3; ++a)
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
                g[b] = x[b] * (NT_GELU_TAIL ? gelu_tail_fast(gt[b]) : gelu_erf_fast(gt[b]));
              }
              gg[q] = u32x2{pack2bf(g[0], g[1]), pack2bf(g[2], g[3])};
              if (WU) { ux[q] = u32x2{pack2bf(x[0], x[1]), pack2bf(x[2], x[3])}; ug[q] = u32x2{pack2bf(gt[0], gt[1]), pack2bf(gt[2], gt[3])}; }
            }
            const int64_t ro = a * 16 + 2 * rp;
            if (WU) {
              store16<NONTEMPORAL>(up + ro * p.ldc, pair_rows(ux[0], ux[1]));
              store16<NONTEMPORAL>(up + ro * p.ldc + p.geglu_hp, pair_rows(ug[0], ug[1]));
            }
            store16<NONTEMPORAL>(gp + ro * p.ldg, pair_rows(gg[0], gg[1]));
          }
      };
      if (u) rows(std::true_type{}); else rows(std::false_type{});
    } else if (EPI == 2) {       // accumulators = dg; du = [dg gelu(gate) | dg x gelu'(gate)] from the stored u = [x | gate]
      const bf16_t* up = p.dgeglu_u + rbase * p.dgeglu_ldu + col;
      bf16_t* dp = reinterpret_cast<bf16_t*>(p.C) + rbase * p.ldc + col;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int rh = 0; rh < 4; rh += 2) {
          u32x4 xv[2], gv[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            xv[q] = *reinterpret_cast<const u32x4*>(up + (int64_t)(a * 16 + rh + q) * p.dgeglu_ldu);
            gv[q] = *reinterpret_cast<const u32x4*>(up + (int64_t)(a * 16 + rh + q) * p.dgeglu_ldu + p.dgeglu_hp);
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
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
    } else if (p.residual) {     // kernel-uniform: bf16 residual rows, twelve 16-byte loads in flight per lane, then add and store
      const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.residual) + rbase * p.ldr + col;
      u32x4 rr[3][4];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) rr[a][r] = *reinterpret_cast<const u32x4*>(rp + (int64_t)(a * 16 + r) * p.ldr);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          u32x4 d;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t w = rr[a][r][b];
            d[b] = pack2bf(fmaf(acc[a][2 * b][r], p.alpha, __uint_as_float(w << 16)), fmaf(acc[a][2 * b + 1][r], p.alpha, __uint_as_float(w & 0xffff0000u)));
          }
          store16<NONTEMPORAL>(reinterpret_cast<bf16_t*>(p.C) + (rbase + a * 16 + r) * p.ldc + col, d);
        }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          u32x4 d;
#pragma unroll
          for (int b = 0; b < 4; ++b) d[b] = pack2bf(acc[a][2 * b][r] * p.alpha, acc[a][2 * b + 1][r] * p.alpha);
          store16<NONTEMPORAL>(reinterpret_cast<bf16_t*>(p.C) + (rbase + a * 16 + r) * p.ldc + col, d);
        }
    }
    if (!tile_of(it + 1, m0, n0)) break;
  }
  wait_vm<0>();      // the cursors ran ahead: no LDS-DMA may be outstanding when the workgroup releases its LDS
#undef NT2_D
#undef NT2_ADVANCE
#undef NT2_RA
#undef NT2_RB
#undef NT2_SB
}

int nt2_launch(const Nt2Params& p, int epi, bool nontemporal, hipStream_t stream) {
  static bool raised_dev[64] = {};                        // (per device)
  int dev_ = 0; (void)hipGetDevice(&dev_);
  bool& raised = raised_dev[(dev_ >= 0 && dev_ < 64) ? dev_ : 0];
  if (!raised) {
    const void* fns[6] = {(const void*)gemm_nt2_kernel<false, 0>, (const void*)gemm_nt2_kernel<true, 0>, (const void*)gemm_nt2_kernel<false, 1>,
                          (const void*)gemm_nt2_kernel<true, 1>, (const void*)gemm_nt2_kernel<false, 2>, (const void*)gemm_nt2_kernel<true, 2>};
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 1;
    raised = true;
  }
  static int ncu = 0;
  if (!ncu) { int dev = 0; hipDeviceProp_t prop; (void)hipGetDevice(&dev); ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) ? (prop.multiProcessorCount & ~7) : 256; }
  const dim3 grid((unsigned)(2 * ncu)), block(NTH);
#define NT2_GO(E) do { if (nontemporal) hipLaunchKernelGGL((gemm_nt2_kernel<true, E>), grid, block, LDS_BYTES, stream, p); \
                       else hipLaunchKernelGGL((gemm_nt2_kernel<false, E>), grid, block, LDS_BYTES, stream, p); } while (0)
  if (epi == 1) NT2_GO(1); else if (epi == 2) NT2_GO(2); else NT2_GO(0);
#undef NT2_GO
  return ctclip_check_launch("gemm_nt2");
}

bool nt2_shape_ok(const void* A, const void* B, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb) {
  if (M % BM || N % BN || K % TK || K / TK < 2) return false;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return false;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return false;          // 32-bit in-panel byte offsets
  return (M / BM) * (N / BN) >= 512;                              // two workgroups on every CU at least once
}

}  // namespace

// which epilogue families go to this kernel: CTCLIP_GEMM_NT2 = bit mask (1 plain / residual, 2 GEGLU forward, 4 GEGLU backward)
static int g_nt2_mask = -1;
int ctclip_gemm_nt2_mask() {
  if (g_nt2_mask < 0) { const char* e = getenv("CTCLIP_GEMM_NT2"); g_nt2_mask = e ? atoi(e) : NT2_DEFAULT_MASK; }
  return g_nt2_mask;
}
// Tuning / test knob of the nn.Linear replacement (attention.py:48,51,119,120,125): selects which epilogue families of ctclip_gemm /
// ctclip_gemm_geglu / ctclip_gemm_dgeglu run on the two-workgroups-per-CU kernel (bit 0 plain / residual, bit 1 GEGLU forward, bit 2 GEGLU
// backward; negative = back to the environment / built-in default).  Returns the previous mask.  Results are bit-identical either way.
extern "C" int ctclip_gemm_nt2_select(int mask) {
  const int prev = ctclip_gemm_nt2_mask();
  g_nt2_mask = mask;
  return prev;
}

// Internal entries (tried first by the gemm_nt.hip launchers).  Return 1 when the shape is not eligible.
int ctclip_gemm_nt2_try(const void* A, const void* B, void* C, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                        int64_t ldc, int64_t ldr, float alpha, bool nontemporal, hipStream_t stream) {
  if (!(ctclip_gemm_nt2_mask() & 1) || !nt2_shape_ok(A, B, M, N, K, lda, ldb)) return 1;
  if ((reinterpret_cast<uintptr_t>(C) % 16) || (ldc % 8) || (residual && ((reinterpret_cast<uintptr_t>(residual) % 16) || (ldr % 8)))) return 1;
  Nt2Params p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.residual = residual; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.alpha = alpha; p.ntm = (int)(M / BM); p.ntn = (int)(N / BN);
  return nt2_launch(p, 0, nontemporal, stream);
}

int ctclip_gemm_nt2_geglu_try(const void* A, const void* B, void* U, void* G, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb, int64_t ldu,
                              int64_t ldg, bool nontemporal, hipStream_t stream) {
  const int64_t N = 2 * (int64_t)hp;
  if (!(ctclip_gemm_nt2_mask() & 2) || !nt2_shape_ok(A, B, M, N, K, lda, ldb)) return 1;
  Nt2Params p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = U; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldu;
  p.alpha = 1.f; p.ntm = (int)(M / BM); p.ntn = (int)(N / BN);
  p.geglu_g = (bf16_t*)G; p.ldg = ldg; p.geglu_hp = hp;
  return nt2_launch(p, 1, nontemporal, stream);
}

int ctclip_gemm_nt2_dgeglu_try(const void* A, const void* B, const void* U, void* dU, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb,
                               int64_t ldu, int64_t lddu, bool nontemporal, hipStream_t stream) {
  const int64_t N = hp;
  if (!(ctclip_gemm_nt2_mask() & 4) || !nt2_shape_ok(A, B, M, N, K, lda, ldb)) return 1;
  Nt2Params p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = dU; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = lddu;
  p.alpha = 1.f; p.ntm = (int)(M / BM); p.ntn = (int)(N / BN);
  p.dgeglu_u = (const bf16_t*)U; p.dgeglu_ldu = ldu; p.dgeglu_hp = hp;
  return nt2_launch(p, 2, nontemporal, stream);
}
