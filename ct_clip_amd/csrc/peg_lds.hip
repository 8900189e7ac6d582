// PEG (attention.py:56-84: causal-padded depthwise 3x3x3 Conv3d over the (B, D1, D2, D3, C) token grid + the residual of
// attention.py:324) for bf16 grids, marching along the causal axis with the input planes resident in LDS.
//
// The first-generation kernels (conv.hip) let every thread fetch its nine neighbour rows from L1/L2: nine 8-byte loads per step,
// one memory round trip per step at two waves per SIMD -- latency-bound at 1.9 TB/s (118 us for 113 MB in + 113 MB out).
// Here a workgroup owns (batch, TB rows of D2, 32 channels) and walks a = 0 .. D1-1:
//   * ONE input plane is live in LDS at a time (one halo row either side of the tile, one zero column either side of D3) while the next
//     PEG_DEPTH planes stream in by LDS-DMA (global_load_lds, 16 B per lane): every input byte crosses HBM once, no VGPR staging;
//   * a dedicated loader wave issues the DMA and is the only wave that waits on vmcnt (counted: vmcnt retires in issue order), so
//     the compute waves never wait for the acknowledgement of their output stores; one barrier per plane;
//   * a compute thread owns (row, channel pair, a quarter of D3) and SCATTERS: input plane a is tap d1 = 2 of output plane a, d1 = 1 of
//     a + 1 and d1 = 0 of a + 2, so the thread keeps three sets of L = D3 / 4 output accumulators (rotating by name: the march is unrolled
//     three times) and reads + unpacks each of its 3 rows x (L + 2) input columns ONCE -- 24 ds_read_b32 + 48 unpacks + 162 packed FMAs per
//     thread and plane at D3 = 24.  (Round 2 gathered from three live planes: 72 reads + 144 unpacks for the same 162 FMAs, and was bound by
//     vector-instruction issue at 3.2 TB/s: profiles/r02_peg.md.)  The 27 x 2 weights stay in registers, the residual is folded into the
//     centre tap; rows of a wave are an odd number of 64-byte positions apart -> the two 32-lane halves of a ds_read_b32 hit disjoint banks;
//   * grad-in (DIR = -1) is the same march from a = D1-1 downwards with the D2 / D3 taps mirrored;
//   * the weight gradient keeps the 27 x 2 tap sums of a thread in registers over the whole march and the same economy: dy plane s is
//     unpacked into registers once and kept for three steps, x plane s - 2 is paired with dy planes s - 2, s - 1, s (30 reads + 60 unpacks
//     per step instead of 78 + 156); then rows -> waves -> workgroup are folded in a fixed order into per-workgroup partials that conv.hip's
//     second stage sums in order: no atomics.
// Bytes per launch: forward / grad-in read x (or dy) once and write y once (2 x B*D1*D2*D3*C*2 B; + 2/TB halo rows from L2);
// the weight gradient reads x and dy once.
#include "common.h"
#include "peg_lds.h"

#include <cstdlib>

#ifndef PEG_DEPTH
#define PEG_DEPTH 2    // input planes in flight ahead of the march (forward / grad-in)
#endif
#ifndef PEG_WDEPTH
#define PEG_WDEPTH 2   // x planes in flight ahead of the weight-gradient march (1 or 2)
#endif
#ifndef PEG_ABL
#define PEG_ABL 0      // ablations (tools/build_ablation.py peg_lds.hip:PEG_ABL ...): 1 deeper prefetch, 2 drain every DMA at each step,
                       // 4 no bf16 unpack instructions (timing only)
#endif

namespace {

constexpr int PCC = 32;   // channels per workgroup = 64 bytes of bf16 per grid position
constexpr int PSEG = 4;   // quarters of the innermost axis

template <int TB, int D3>
struct Geo {
  static constexpr int L = D3 / PSEG;             // outputs per thread and plane
  static constexpr int RSP = (D3 + 2) | 1;        // positions per LDS row of the x ring (odd)
  static constexpr int ROWB = RSP * 64;
  static constexpr int SLOT = (TB + 2) * ROWB;
  static constexpr int RSD = D3 | 1;              // positions per LDS row of a dy plane (no halo)
  static constexpr int DROWB = RSD * 64;
  static constexpr int DSLOT = TB * DROWB;
  static constexpr int NCW = TB / 4 * PSEG;       // compute waves
  static constexpr int PIECES = (D3 + 15) / 16;   // 1-KiB DMA pieces per row
  static constexpr int NPX = (TB + 2) * PIECES;   // DMA pieces per x plane (halo rows included), per dy plane
  static constexpr int NPD = TB * PIECES;
  static constexpr int LDS_MAX = 160 * 1024;
  // forward / grad-in: ONE live plane (the march scatters each input plane into the three outputs it belongs to) + PEG_DEPTH planes in
  // flight, + one dump row (vmcnt is a 6-bit counter: the loader wave never has more than 63 pieces outstanding)
  static constexpr int DEPTH_LDS = (LDS_MAX - ROWB) / SLOT - 1;
  static constexpr int DEPTH_VM = 63 / NPX;
  static constexpr int DEPTH_CAP = PEG_DEPTH;
  static constexpr int DEPTH = DEPTH_LDS < DEPTH_VM ? (DEPTH_LDS < DEPTH_CAP ? DEPTH_LDS : DEPTH_CAP) : (DEPTH_VM < DEPTH_CAP ? DEPTH_VM : DEPTH_CAP);
  static constexpr int NSLOT = DEPTH + 1;
  // weight gradient: one live x plane + WDEPTH in flight, the dy planes one ahead
  static constexpr int WDEPTH = PEG_WDEPTH;
  static constexpr int WSLOT = 1 + WDEPTH;
};

// s_waitcnt vmcnt(N) only (lgkmcnt / expcnt untouched); vmcnt retires in issue order
template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  constexpr int M = (PEG_ABL & 2) ? 0 : N;
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(M) : "memory");      // (asm + memory clobber: the optimiser may move the s_waitcnt builtin across LDS accesses)
}

struct Tile { int64_t b; int beta0, c0; };

// 1-D grid -> (channel chunk, row tile, batch).  Workgroups are dealt to the 8 XCDs round-robin; with 16 chunks the two chunks that
// share the 128-byte lines of x (2j, 2j+1) get ids 8 apart = the same XCD, so a line is pulled into one L2 only.
__device__ __forceinline__ Tile tile_of(int nchunk, int ntile, int TBv) {
  const int id = blockIdx.x;
  int chunk, rest;
  if (nchunk == 16) { chunk = (id & 7) * 2 + ((id >> 3) & 1); rest = id >> 4; }
  else { chunk = id % nchunk; rest = id / nchunk; }
  Tile t;
  t.c0 = chunk * PCC;
  t.beta0 = (rest % ntile) * TBv;
  t.b = rest / ntile;
  return t;
}

__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The lane part of a DMA piece's source address: sixteen positions x four 16-byte quarters of the 64 channel bytes.
template <int D3>
struct LaneSrc {
  static constexpr int PIECES = (D3 + 15) / 16;
  uint32_t off[PIECES];    // bytes from the start of the grid row
  bool on[PIECES];         // position < D3
  __device__ __forceinline__ void init(int C, int lane) {
#pragma unroll
    for (int h = 0; h < PIECES; ++h) {
      const int g = 16 * h + (lane >> 2);
      on[h] = g < D3;
      off[h] = (uint32_t)((g < D3 ? g : 0) * C + (lane & 3) * 8) * 2u;
    }
  }
  // (selects, not indexing: a run-time index would put the arrays into scratch memory -- and every scratch load is followed by a
  // vmcnt(0) that drains the DMA queue)
  __device__ __forceinline__ uint32_t off_of(int h) const {
    uint32_t v = off[0];
#pragma unroll
    for (int k = 1; k < PIECES; ++k) v = h == k ? off[k] : v;
    return v;
  }
  __device__ __forceinline__ bool on_of(int h) const {
    bool v = on[0];
#pragma unroll
    for (int k = 1; k < PIECES; ++k) v = h == k ? on[k] : v;
    return v;
  }
};

// All pieces wave0, wave0 + stride, ... of one plane: piece q = row q / PIECES of the tile (grid row row_first + q / PIECES), sixteen
// positions from 16 * (q % PIECES), to dst + row * rowb + col0 positions.  Rows outside [0, D2) fetch a clamped row into `dump`
// (their LDS rows stay zero): every plane costs the same number of DMA instructions, so the vmcnt arithmetic is exact.  Everything
// but the lane offset is wave-uniform (scalar registers); the loop is unrolled.
typedef __attribute__((address_space(3))) char* lds_ptr;     // (LDS addresses stay 32-bit: no generic-pointer round trip)

template <int D3, int NPIECE, int STRIDE>
__device__ __forceinline__ void dma_plane(const bf16_t* __restrict__ src_plane, lds_ptr l3, uint32_t dst, uint32_t dump, int rowb, int col0, int row_first,
                                          int q0, int D2, int C, const LaneSrc<D3>& ls) {
  constexpr int PIECES = (D3 + 15) / 16;
  const int64_t row_bytes = (int64_t)D3 * C * 2;
#pragma unroll
  for (int k = 0; k < (NPIECE + STRIDE - 1) / STRIDE; ++k) {
    const int q = q0 + k * STRIDE;
    if (q < NPIECE) {
      const int i = q / PIECES, h = q % PIECES, beta = row_first + i;
      const bool inside = beta >= 0 && beta < D2;
      const int bc = beta < 0 ? 0 : (beta >= D2 ? D2 - 1 : beta);
      const char* row = reinterpret_cast<const char*>(src_plane) + bc * row_bytes;
      const uint32_t to = inside ? dst + (uint32_t)(i * rowb + (col0 + 16 * h) * 64) : dump + (uint32_t)(16 * h * 64);
      if (ls.on_of(h))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + ls.off_of(h)), (__attribute__((address_space(3))) void*)(l3 + to), 16, 0, 0);
    }
  }
}

__device__ __forceinline__ void unpack2(uint32_t v, float& lo, float& hi) {
#if PEG_ABL & 4      // timing only: no unpack instructions (wrong values)
  lo = __uint_as_float(v); hi = __uint_as_float(v);
#else
  lo = __uint_as_float(v << 16); hi = __uint_as_float(v & 0xffff0000u);
#endif
}

// ---------------------------------------------------------------------------------------------------- forward / grad-in
template <int TB, int D3, int DIR>
__global__ __launch_bounds__((Geo<TB, D3>::NCW + 1) * 64) void peg_march_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                                const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                                                const bf16_t* __restrict__ ein, bf16_t* __restrict__ rres, int D1, int D2,
                                                                                int C, int ntile) {
  // DIR: +1 forward, -1 grad-in, +2 forward on the COMPENSATED residual stream (ctclip_peg_fwd_comp, profiles/r03_bf16_error_budget.md):
  // s = x + e_in + conv(x) in f32, y = bf16(s) and the rounding residue e_out = bf16(s - y) are stored (e_in may be null: zeros)
  constexpr bool COMP = DIR == 2;
  using G = Geo<TB, D3>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NT = (G::NCW + 1) * 64;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // (wave: uniform -> scalar registers)
  const Tile t = tile_of(C / PCC, ntile, TB);
  for (int i = threadIdx.x; i < G::NSLOT * G::SLOT / 16; i += NT) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const int64_t plane_elems = (int64_t)D2 * D3 * C;
  const bf16_t* xb = x + t.b * D1 * plane_elems + t.c0;
  auto plane_of = [&](int m) { return DIR > 0 ? m : D1 - 1 - m; };

  if (wave == G::NCW) {                                  // ---- the loader wave: plane m + DEPTH is issued in step m
    const lds_ptr l3 = (lds_ptr)lds;
    constexpr uint32_t dump = G::NSLOT * G::SLOT;
    LaneSrc<D3> ls;
    ls.init(C, lane);
    auto issue = [&](int m) {
      dma_plane<D3, G::NPX, 1>(xb + plane_of(m) * plane_elems, l3, (uint32_t)((m % G::NSLOT) * G::SLOT), dump, G::ROWB, 1, t.beta0 - 1, 0, D2, C, ls);
    };
    for (int m = 0; m < G::DEPTH && m < D1; ++m) issue(m);
    for (int m = 0; m < D1; ++m) {
      // plane m must have landed; the planes issued after it (at most DEPTH - 1 of them) may stay in flight
      const int ahead = D1 - 1 - m < G::DEPTH - 1 ? D1 - 1 - m : G::DEPTH - 1;
      if (ahead >= 2) wait_vm<(G::DEPTH >= 3 ? 2 : 0) * G::NPX>();
      else if (ahead == 1) wait_vm<(G::DEPTH >= 2 ? 1 : 0) * G::NPX>();
      else wait_vm<0>();
      wg_barrier();
      if (m + G::DEPTH < D1) issue(m + G::DEPTH);
    }
    return;
  }

  // ---- compute waves
  const int rg = wave % (TB / 4), seg = wave / (TB / 4);
  const int r = rg * 4 + (lane >> 4), pr = lane & 15, g0 = seg * G::L;
  const int ch = t.c0 + 2 * pr;
  float wk[27][2];
  {
    const float* w0 = w + (int64_t)ch * 27;
#pragma unroll
    for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2)
#pragma unroll
        for (int d3 = 0; d3 < 3; ++d3) {
          const int src = DIR > 0 ? d1 * 9 + d2 * 3 + d3 : d1 * 9 + (2 - d2) * 3 + (2 - d3);
          wk[d1 * 9 + d2 * 3 + d3][0] = w0[src];
          wk[d1 * 9 + d2 * 3 + d3][1] = w0[27 + src];
        }
    wk[2 * 9 + 1 * 3 + 1][0] += 1.f; wk[2 * 9 + 1 * 3 + 1][1] += 1.f;      // the residual
  }
  float bv[2] = {0.f, 0.f};
  if (DIR > 0 && bias) { bv[0] = bias[ch]; bv[1] = bias[ch + 1]; }
  const uint32_t tb = (uint32_t)((r * G::RSP + g0) * 64 + pr * 4);
  // stores: a buffer descriptor over the batch item, a scalar offset per plane and column, one 32-bit lane offset; rows past D2
  // (ragged last tile) get an offset beyond the descriptor and the hardware drops their stores -- no branches in the column loop
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(y + t.b * D1 * plane_elems, 0, (int)(D1 * plane_elems * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rres_d = __builtin_amdgcn_make_buffer_rsrc((COMP ? rres : y) + t.b * D1 * plane_elems, 0, (int)(D1 * plane_elems * 2), 0x00020000);
  // (a null e_in becomes a descriptor of zero records: every load is out of range and returns 0 without touching memory)
  const __amdgpu_buffer_rsrc_t ein_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(COMP && ein ? ein : x) + t.b * D1 * plane_elems, 0,
                                                                         (COMP && ein) ? (int)(D1 * plane_elems * 2) : 0, 0x00020000);
  const uint32_t lane_off = t.beta0 + r < D2 ? (uint32_t)((((t.beta0 + r) * D3 + g0) * C + ch) * 2) : 0x80000000u;

  // One march step = one INPUT plane: plane m (march order) feeds the outputs m (tap d1 = 2: complete after this step -> stored), m + 1
  // (d1 = 1) and m + 2 (d1 = 0: started here), so every input value is read from LDS and unpacked ONCE per thread -- 3 rows x (L + 2)
  // columns per step instead of the 3 planes x 3 rows x (L + 2) of a gather over the three live planes (72 -> 24 ds_read_b32, 144 -> 48
  // unpacks per thread and plane; the 162 packed FMAs are the arithmetic itself).  The three accumulator sets rotate by NAME: the march
  // loop is unrolled three times.
  auto step = [&](int m, float (&P0)[G::L][2], float (&P1)[G::L][2], float (&P2)[G::L][2]) {
    wg_barrier();
    const char* pl = lds + (m % G::NSLOT) * G::SLOT + tb;
    const uint32_t plane_off = (uint32_t)(plane_of(m) * plane_elems * 2);
    uint32_t ev[G::L];      // COMP: the compensation terms of this plane's outputs, requested a whole plane body ahead of their use
    if (COMP) {
#pragma unroll
      for (int j = 0; j < G::L; ++j) ev[j] = __builtin_amdgcn_raw_buffer_load_b32(ein_d, lane_off, plane_off + (uint32_t)(j * C * 2), 0);
    }
#pragma unroll
    for (int c = 0; c < G::L; ++c) { P2[c][0] = bv[0]; P2[c][1] = bv[1]; }
    uint32_t raw[2][3];
    auto fetch = [&](uint32_t (&dst)[3], int jj) {
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) dst[d2] = *reinterpret_cast<const uint32_t*>(pl + d2 * G::ROWB + jj * 64);
    };
    fetch(raw[0], 0);
#pragma unroll
    for (int jj = 0; jj < G::L + 2; ++jj) {
      if (jj + 1 < G::L + 2) fetch(raw[(jj + 1) & 1], jj + 1);   // one column ahead
      asm volatile("" ::: "memory");
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) {
        float x0, x1;
        unpack2(raw[jj & 1][d2], x0, x1);
#pragma unroll
        for (int d3 = 0; d3 < 3; ++d3) {
          const int c = jj - d3;                                  // input column jj (halo coordinates) is tap d3 of output column jj - d3
          if (c >= 0 && c < G::L) {
            P0[c][0] = fmaf(wk[18 + d2 * 3 + d3][0], x0, P0[c][0]); P0[c][1] = fmaf(wk[18 + d2 * 3 + d3][1], x1, P0[c][1]);
            P1[c][0] = fmaf(wk[9 + d2 * 3 + d3][0], x0, P1[c][0]);  P1[c][1] = fmaf(wk[9 + d2 * 3 + d3][1], x1, P1[c][1]);
            P2[c][0] = fmaf(wk[d2 * 3 + d3][0], x0, P2[c][0]);      P2[c][1] = fmaf(wk[d2 * 3 + d3][1], x1, P2[c][1]);
          }
        }
      }
      if (jj >= 2) {                                              // output column jj - 2 of plane m has all its 27 taps
        float o0 = P0[jj - 2][0], o1 = P0[jj - 2][1];
        if (COMP) { o0 += __uint_as_float(ev[jj - 2] << 16); o1 += __uint_as_float(ev[jj - 2] & 0xffff0000u); }
        const uint32_t yv = pack2bf(o0, o1);
        __builtin_amdgcn_raw_buffer_store_b32(yv, yres, lane_off, plane_off + (uint32_t)((jj - 2) * C * 2), 0);
        if (COMP)
          __builtin_amdgcn_raw_buffer_store_b32(pack2bf(o0 - __uint_as_float(yv << 16), o1 - __uint_as_float(yv & 0xffff0000u)), rres_d,
                                                lane_off, plane_off + (uint32_t)((jj - 2) * C * 2), 0);
      }
    }
  };
  float PA[G::L][2], PB[G::L][2], PC[G::L][2];
#pragma unroll
  for (int c = 0; c < G::L; ++c) { PA[c][0] = PB[c][0] = bv[0]; PA[c][1] = PB[c][1] = bv[1]; }
#pragma nounroll
  for (int m = 0; m < D1; m += 3) {
    step(m, PA, PB, PC);
    if (m + 1 < D1) step(m + 1, PB, PC, PA);
    if (m + 2 < D1) step(m + 2, PC, PA, PB);
  }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
// part[(b * ntile + tile)][C][28]: 27 taps + the bias gradient of the workgroup's rows
template <int TB, int D3>
__global__ __launch_bounds__((Geo<TB, D3>::NCW) * 64) void peg_wgrad_march_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                                      float* __restrict__ part, int D1, int D2, int C, int ntile) {
  using G = Geo<TB, D3>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NT = G::NCW * 64;
  constexpr int XR = G::WSLOT * G::SLOT;                  // x ring, then two dy planes, then the dump row
  constexpr int NPX = G::NPX, NPD = G::NPD;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // (wave: uniform -> scalar registers)
  const Tile t = tile_of(C / PCC, ntile, TB);
  for (int i = threadIdx.x; i < (XR + 2 * G::DSLOT) / 16; i += NT) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  const int64_t plane_elems = (int64_t)D2 * D3 * C;
  const bf16_t* xb = x + t.b * D1 * plane_elems + t.c0;
  const bf16_t* gb = dy + t.b * D1 * plane_elems + t.c0;

  float acc[27][2], accb[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 27; ++k) { acc[k][0] = 0.f; acc[k][1] = 0.f; }
  const int rg = wave % (TB / 4), seg = wave / (TB / 4);
  const int r = rg * 4 + (lane >> 4), pr = lane & 15, g0 = seg * G::L;

  // No stores in the march, so every wave fetches its share of the coming planes itself (pieces wave, wave + NCW, ...): no loader wave.
  // Step s (0 .. D1 + 1) takes dy plane s out of LDS into registers (unpacked, kept for three steps: the sets rotate by NAME, the march is
  // unrolled three times) and pairs x plane s - 2 with dy planes s - 2, s - 1, s (taps d1 = 2, 1, 0): every x value is read and unpacked
  // ONCE per thread (3 rows x (L + 2) columns per step, not 3 planes x 3 rows), every dy value once.  dy plane s + 1 is requested in
  // step s, x plane j in step j + 2 - WDEPTH (WDEPTH steps before its use); with WDEPTH = 2 a wave waits before the barrier of step s
  // until only its pieces of x plane s - 1 (the youngest) are still in flight.
  const lds_ptr l3 = (lds_ptr)lds;
  constexpr uint32_t dump = XR + 2 * G::DSLOT;
  LaneSrc<D3> ls;
  ls.init(C, lane);
  auto issue_x = [&](int m) {
    dma_plane<D3, NPX, G::NCW>(xb + m * plane_elems, l3, (uint32_t)((m % G::WSLOT) * G::SLOT), dump, G::ROWB, 1, t.beta0 - 1, wave, D2, C, ls);
  };
  auto issue_g = [&](int m) {
    dma_plane<D3, NPD, G::NCW>(gb + m * plane_elems, l3, (uint32_t)(XR + (m & 1) * G::DSLOT), dump, G::DROWB, 0, t.beta0, wave, D2, C, ls);
  };
  constexpr int XLO = NPX / G::NCW, XEXTRA = NPX % G::NCW;   // x pieces per wave: XLO, + 1 for the first XEXTRA waves
  const uint32_t tb = (uint32_t)((r * G::RSP + g0) * 64 + pr * 4);
  const uint32_t db = (uint32_t)(XR + (r * G::RSD + g0) * 64 + pr * 4);
  auto step = [&](int s, float (&Ga)[G::L][2], float (&Gb)[G::L][2], float (&Gc)[G::L][2]) {      // dy planes s - 2, s - 1, s (Gc: filled here)
    if (G::WDEPTH == 2 && s >= 1 && s <= D1) { if (wave < XEXTRA) wait_vm<XLO + 1>(); else wait_vm<XLO>(); }
    else wait_vm<0>();
    wg_barrier();
    if (s + 1 < D1) issue_g(s + 1);
    if (s + G::WDEPTH - 2 >= 0 && s + G::WDEPTH - 2 < D1) issue_x(s + G::WDEPTH - 2);
    {
      const char* gp = lds + db + (s & 1) * G::DSLOT;
      const uint32_t live = s < D1 ? 0xffffffffu : 0u;        // past the last plane the slot holds an old plane: dy = 0
#pragma unroll
      for (int c = 0; c < G::L; ++c) {
        unpack2(*reinterpret_cast<const uint32_t*>(gp + c * 64) & live, Gc[c][0], Gc[c][1]);
        accb[0] += Gc[c][0]; accb[1] += Gc[c][1];
      }
    }
    if (s < 2) return;
    const char* pl = lds + ((s - 2) % G::WSLOT) * G::SLOT + tb;
    uint32_t raw[2][3];
    auto fetch = [&](uint32_t (&dst)[3], int jj) {
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) dst[d2] = *reinterpret_cast<const uint32_t*>(pl + d2 * G::ROWB + jj * 64);
    };
    fetch(raw[0], 0);
#pragma unroll
    for (int jj = 0; jj < G::L + 2; ++jj) {
      if (jj + 1 < G::L + 2) fetch(raw[(jj + 1) & 1], jj + 1);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int d2 = 0; d2 < 3; ++d2) {
        float x0, x1;
        unpack2(raw[jj & 1][d2], x0, x1);
#pragma unroll
        for (int d3 = 0; d3 < 3; ++d3) {
          const int c = jj - d3;                       // input column jj (halo coordinates) is tap d3 of output column jj - d3
          if (c >= 0 && c < G::L) {
            acc[18 + d2 * 3 + d3][0] = fmaf(Ga[c][0], x0, acc[18 + d2 * 3 + d3][0]); acc[18 + d2 * 3 + d3][1] = fmaf(Ga[c][1], x1, acc[18 + d2 * 3 + d3][1]);
            acc[9 + d2 * 3 + d3][0] = fmaf(Gb[c][0], x0, acc[9 + d2 * 3 + d3][0]);   acc[9 + d2 * 3 + d3][1] = fmaf(Gb[c][1], x1, acc[9 + d2 * 3 + d3][1]);
            acc[d2 * 3 + d3][0] = fmaf(Gc[c][0], x0, acc[d2 * 3 + d3][0]);           acc[d2 * 3 + d3][1] = fmaf(Gc[c][1], x1, acc[d2 * 3 + d3][1]);
          }
        }
      }
    }
  };
  issue_g(0);
  {
    float GA[G::L][2], GB[G::L][2], GC[G::L][2];
#pragma unroll
    for (int c = 0; c < G::L; ++c) { GA[c][0] = GA[c][1] = GB[c][0] = GB[c][1] = 0.f; }
#pragma nounroll
    for (int s = 0; s < D1 + 2; s += 3) {
      step(s, GA, GB, GC);
      if (s + 1 < D1 + 2) step(s + 1, GB, GC, GA);
      if (s + 2 < D1 + 2) step(s + 2, GC, GA, GB);
    }
  }
  // ---- fold: the 4 rows of a wave (lanes 16 apart), then the compute waves in order, then out
  wg_barrier();                                           // every wave is done with the ring
  float* red = reinterpret_cast<float*>(lds);             // [wave][pair][57]
  {
    float v[56];
#pragma unroll
    for (int k = 0; k < 27; ++k) { v[2 * k] = acc[k][0]; v[2 * k + 1] = acc[k][1]; }
    v[54] = accb[0]; v[55] = accb[1];
#pragma unroll
    for (int k = 0; k < 56; ++k) {
      v[k] += __shfl_xor(v[k], 16);
      v[k] += __shfl_xor(v[k], 32);
    }
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < 56; ++k) red[(wave * 16 + lane) * 57 + k] = v[k];
    }
  }
  __syncthreads();
  const int tile = (t.beta0 / TB);
  for (int i = threadIdx.x; i < PCC * 28; i += NT) {
    const int cc = i / 28, tap = i % 28;
    const int idx = tap < 27 ? tap * 2 + (cc & 1) : 54 + (cc & 1);
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < G::NCW; ++wv) s += red[(wv * 16 + (cc >> 1)) * 57 + idx];
    part[((t.b * ntile + tile) * C + t.c0 + cc) * 28 + tap] = s;
  }
}

bool lds_path_enabled() {
  static const bool on = [] { const char* e = getenv("CTCLIP_PEG_LDS"); return !(e && e[0] == '0'); }();
  return on;
}

// TB = 12 when that fills the chip, else TB = 4 (three workgroups per CU).  D3 = 8 always takes TB = 4: at thirteen waves per workgroup (128
// registers) the two-column threads of that geometry spill, at five waves they do not.
int pick_tb(int64_t B, int D2, int D3, int C) {
  static int forced = -1;                                  // CTCLIP_PEG_TB=4|12: A/B timing of the tile height
  if (forced < 0) { const char* e = getenv("CTCLIP_PEG_TB"); forced = e ? atoi(e) : 0; }
  if ((forced == 4 || forced == 12) && D3 != 8) return forced;
  return (D3 != 8 && B * ((D2 + 11) / 12) * (C / PCC) >= 200) ? 12 : 4;
}

template <int TB, int D3, int DIR>
int launch_march(const bf16_t* x, const float* w, const float* bias, bf16_t* y, const bf16_t* ein, bf16_t* rres, int64_t B, int D1, int D2, int C, hipStream_t s) {
  using G = Geo<TB, D3>;
  const int ntile = (D2 + TB - 1) / TB;
  constexpr int SHM = G::NSLOT * G::SLOT + G::ROWB;
  static_assert(G::DEPTH >= 1, "one live plane and at least one in flight");
  const size_t shm = SHM;
  static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&peg_march_kernel<TB, D3, DIR>), hipFuncAttributeMaxDynamicSharedMemorySize, SHM) == hipSuccess; }();
  if (!once) { (void)hipGetLastError(); return 1; }
  hipLaunchKernelGGL((peg_march_kernel<TB, D3, DIR>), dim3((unsigned)(B * ntile * (C / PCC))), dim3((G::NCW + 1) * 64), shm, s, x, w, bias, y, ein, rres, D1, D2, C, ntile);
  return 0;
}

template <int TB, int D3>
int launch_wgrad(const bf16_t* dy, const bf16_t* x, float* part, int64_t B, int D1, int D2, int C, hipStream_t s) {
  using G = Geo<TB, D3>;
  const int ntile = (D2 + TB - 1) / TB;
  constexpr int RING = G::WSLOT * G::SLOT + 2 * G::DSLOT + G::ROWB, FOLD = G::NCW * 16 * 57 * 4;     // (the fold reuses the ring)
  constexpr int SHM = RING > FOLD ? RING : FOLD;
  if (SHM > G::LDS_MAX) return 1;                        // e.g. 12-row tiles of D3 = 32: the first-generation kernel takes over
  static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&peg_wgrad_march_kernel<TB, D3>), hipFuncAttributeMaxDynamicSharedMemorySize, SHM) == hipSuccess; }();
  if (!once) { (void)hipGetLastError(); return 1; }
  hipLaunchKernelGGL((peg_wgrad_march_kernel<TB, D3>), dim3((unsigned)(B * ntile * (C / PCC))), dim3(G::NCW * 64), SHM, s, dy, x, part, D1, D2, C, ntile);
  return 0;
}

}  // namespace

bool peg_lds_supported(int64_t B, int D1, int D2, int D3, int C, int dtype) {
  return lds_path_enabled() && dtype == DT_BF16 && C % PCC == 0 && (D3 == 8 || D3 == 16 || D3 == 24 || D3 == 32) && B >= 1 && D1 >= 1 && D2 >= 1 &&
         (int64_t)D1 * D2 * D3 * C * 2 < (int64_t)1 << 31;       // one buffer descriptor per batch item
}

int64_t peg_lds_wgrad_groups(int64_t B, int D2, int C) { return B * ((D2 + 3) / 4); }   // upper bound over both tile heights

#define PEG_D3_SWITCH(CALL)                 \
  switch (D3) {                             \
    case 8: return CALL(8);                 \
    case 16: return CALL(16);               \
    case 24: return CALL(24);               \
    case 32: return CALL(32);               \
    default: return 1;                      \
  }

int peg_lds_march(const void* x, const float* w, const float* bias, void* y, int64_t B, int D1, int D2, int D3, int C, int dir, hipStream_t s, const void* ein, void* rres) {
  const bf16_t* xp = (const bf16_t*)x; bf16_t* yp = (bf16_t*)y; bf16_t* rp = (bf16_t*)rres; const bf16_t* ep = (const bf16_t*)ein;
  const int tb = pick_tb(B, D2, D3, C);
#define FWD12(D) launch_march<12, D, 1>(xp, w, bias, yp, nullptr, nullptr, B, D1, D2, C, s)
#define FWD4(D) launch_march<4, D, 1>(xp, w, bias, yp, nullptr, nullptr, B, D1, D2, C, s)
#define CMP12(D) launch_march<12, D, 2>(xp, w, bias, yp, ep, rp, B, D1, D2, C, s)
#define CMP4(D) launch_march<4, D, 2>(xp, w, bias, yp, ep, rp, B, D1, D2, C, s)
#define BWD12(D) launch_march<12, D, -1>(xp, w, nullptr, yp, nullptr, nullptr, B, D1, D2, C, s)
#define BWD4(D) launch_march<4, D, -1>(xp, w, nullptr, yp, nullptr, nullptr, B, D1, D2, C, s)
  if (dir > 0 && rp) { if (tb == 12) { PEG_D3_SWITCH(CMP12) } else { PEG_D3_SWITCH(CMP4) } }
  else if (dir > 0) { if (tb == 12) { PEG_D3_SWITCH(FWD12) } else { PEG_D3_SWITCH(FWD4) } }
  else { if (tb == 12) { PEG_D3_SWITCH(BWD12) } else { PEG_D3_SWITCH(BWD4) } }
#undef FWD12
#undef FWD4
#undef CMP12
#undef CMP4
#undef BWD12
#undef BWD4
}

int peg_lds_wgrad(const void* dy, const void* x, float* part, int64_t B, int D1, int D2, int D3, int C, int* groups, hipStream_t s) {
  const bf16_t* gp = (const bf16_t*)dy; const bf16_t* xp = (const bf16_t*)x;
  const int tb = pick_tb(B, D2, D3, C);
  *groups = (int)(B * ((D2 + tb - 1) / tb));
#define WG12(D) launch_wgrad<12, D>(gp, xp, part, B, D1, D2, C, s)
#define WG4(D) launch_wgrad<4, D>(gp, xp, part, B, D1, D2, C, s)
  if (tb == 12) { PEG_D3_SWITCH(WG12) } else { PEG_D3_SWITCH(WG4) }
#undef WG12
#undef WG4
}
