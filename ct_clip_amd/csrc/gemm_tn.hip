// bf16 "TN" GEMM for gfx950: C[M x N] (+)= alpha * A^T B with A = [K][M] and B = [K][N], both stored k-major (a row is one k).
// This is the weight gradient dW = dy^T x of every linear layer: K = tokens (110592 at batch 8), M x N = the weight (<= 2816 x 512), so
// the product is split along K over the whole chip and the partial 256 x 256 tiles are summed by a second kernel (deterministic).
//
// Same machinery as gemm_nt.hip (read that header first): 128-byte global reads into a ring of five 32-KiB panels by LDS-DMA with
// counted vmcnt waits, and a rolling schedule in which every LDS read and every DMA piece is issued between two MFMAs.  What differs:
//  * the operands are k-major, but mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane: fragments are fetched with the
//    gfx950 transposing LDS read ds_read_b64_tr_b16 (two per fragment).  Measured semantics (tools/tr_probe.hip): inside each
//    16-lane group, output lane i, element j = element (i & 3) of the 8 bytes addressed by lane 4j + (i >> 2).  With lane t
//    pointing at row k0 + (t >> 2), columns c0 + 4 (t & 3) of a [k][column] image, lane i receives column c0 + i for k0 .. k0+3.
//  * LDS image of a panel (64 k x 256 columns): [column quarter q][k][128-byte segment]; the four 32-byte sub-chunks of a segment
//    are XOR-swizzled with (k >> 1) & 3, so the eight rows a half-wave reads (32 B each) fall into eight different 32-byte bank
//    groups.  The swizzle is applied on the DMA source side (eight lanes still read one full, merely permuted, 128-byte line).
//  * the k-slots of a fragment are k = 4g + j and 16 + 4g + j (g = lane >> 4); any bijection is valid as long as A and B agree.
//
// The first version of this path (gemm256.hip: 2-stage ring, 8 x 8 register transposes on the way into LDS) ran the dW GEMMs at
// 637 TFLOP/s.
#include "common.h"

#ifndef TN_DEPHASE
#define TN_DEPHASE 0      // round-5 experiment (see gemm_nt.hip NT_DEPHASE): waves 4-7 issue their A-panel LDS-DMA pieces in H3 instead of H1
#endif

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int PANEL = TK * 256 * 2;             // 32 KiB
constexpr int NPANEL = 5;
constexpr int NTH = 512;
constexpr int GL = 4;                           // LDS-DMA pieces per wave per panel

struct TnParams {
  const bf16_t* A; const bf16_t* B; float* slabs;
  int64_t M, N, K, lda, ldb, slab_ld;
  float alpha;
  int ntm, ntn, nsplit, k_per_split;
};

__device__ __forceinline__ const char* to_sgpr(const char* ptr) {
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
#ifndef STRICT_VMCNT
#define STRICT_VMCNT 0      // 1 = every counted vmcnt wait becomes vmcnt(0) (determinism bisection builds, tools/trace_determinism.py)
#endif
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(STRICT_VMCNT ? 0 : N) : "memory"); }

// One fragment (8 k-slots x 16 columns) = two transposing reads 16 k-rows (2048 B) apart.  Inline asm on purpose: through the
// builtin the compiler cannot tell these reads from the LDS-DMA writes in flight and drains vmcnt to 0 before every one of them
// (measured: 3.2 us per k-step instead of 0.9).  The price: the compiler no longer tracks lgkmcnt for them, the kernel places
// its own counted waits (LDS returns in order) and threads the fragment registers through them to pin the order.
template <int OFF>
__device__ __forceinline__ u32x4 read_frag(uint32_t vaddr) {
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(vaddr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(vaddr), "n"(OFF + 2048));
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
// wait until at most N LDS reads are outstanding; x, y = the fragments about to be consumed (dependency only)
template <int N> __device__ __forceinline__ void wait_lds(u32x4& x, u32x4& y) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N)); }
template <int N> __device__ __forceinline__ void wait_lds3(u32x4& x, u32x4& y, u32x4& z) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(x), "+v"(y), "+v"(z) : "n"(N)); }

__global__ __launch_bounds__(NTH) void gemm_tn_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // 4 x 2 waves, wave tile 64 (M) x 128 (N)
  const int li = lane & 15, lg = lane >> 4;
  const int ntiles = p.ntm * p.ntn;
  // split-K block order: the output tiles of one k-range run on ONE XCD at the same time, so the re-reads of that k-range by the
  // other tiles hit its L2 and HBM sees each operand element once.  Work units v = split * ntiles + tile are dealt to the XCDs in
  // contiguous runs of `per` units (workgroups go to XCD blockIdx % 8); the grid is 8 * per, surplus workgroups exit.
  const int per = (ntiles * p.nsplit + 7) >> 3;
  const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (v >= ntiles * p.nsplit) return;
  const int split = v / ntiles, tile = v % ntiles;
  const int64_t m0 = (int64_t)(tile / p.ntn) * TM, n0 = (int64_t)(tile % p.ntn) * TN;
  const int64_t kbeg = (int64_t)split * p.k_per_split;
  int64_t kend = kbeg + p.k_per_split; if (kend > p.K) kend = p.K;
  const int nk = kbeg < kend ? (int)((kend - kbeg) / TK) : 0;

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    // ---- loader: piece = 8 k-rows x 128 B of one column quarter; wave w issues pieces 4w .. 4w+3 of each panel
    const char* a_base = to_sgpr(reinterpret_cast<const char*>(p.A) + (kbeg * p.lda + m0) * 2);
    const char* b_base = to_sgpr(reinterpret_cast<const char*>(p.B) + (kbeg * p.ldb + n0) * 2);
    uint32_t a_off[GL], b_off[GL];
    {
      const int64_t mpad = (p.M + 7) / 8 * 8, npad = (p.N + 7) / 8 * 8;
#pragma unroll
      for (int j = 0; j < GL; ++j) {
        const int piece = wave * GL + j, qd = piece >> 3, r = (piece & 7) * 8 + (lane >> 3);
        const int pos = lane & 7, c = (pos >> 1) ^ ((r >> 1) & 3), col = qd * 64 + c * 16 + (pos & 1) * 8;   // tile column of the chunk
        int64_t ca = m0 + col, cb = n0 + col;
        ca = ca + 8 <= mpad ? ca : mpad - 8;      // chunks past the edge re-read a valid one (their columns are never stored)
        cb = cb + 8 <= npad ? cb : npad - 8;
        a_off[j] = (uint32_t)r * (uint32_t)(p.lda * 2) + (uint32_t)((ca - m0) * 2);
        b_off[j] = (uint32_t)r * (uint32_t)(p.ldb * 2) + (uint32_t)((cb - n0) * 2);
      }
    }
    const int64_t a_step = p.lda * (TK * 2), b_step = p.ldb * (TK * 2);
    int a_t = 0, b_t = 0;                       // loader positions (k-steps), clamped to the last step of the range
    auto glds = [&](const char* sbase, uint32_t voff, int slot, int j) {
      const char* src = sbase + (uint64_t)voff;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + slot * PANEL + (wave * GL + j) * 1024), 16, 0, 0);
    };
    auto wrap = [](int s) { return s >= NPANEL ? s - NPANEL : s; };
    auto adv = [&](int& t) { if (t + 1 < nk) ++t; };

    // fragment addressing: lane t = li supplies row (t >> 2), column piece (t & 3) of the 16-column block; k-slot group lg.
    // The swizzle (k >> 1) & 3 does not depend on the sub-step (+32 rows) nor on the second read (+16 rows): four lane
    // addresses (one per 32-byte sub-chunk c) serve every fragment through immediate offsets.
    const int frow = lg * 4 + (li >> 2);
    uint32_t fadr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fadr[c] = (uint32_t)(frow * 128 + ((c ^ ((frow >> 1) & 3)) << 5) + (li & 3) * 8);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    u32x4 fa[4], fb[8];
    uint32_t pa[4], pb[4];          // fadr + base of the panel being read (A: quarter wm; B: quarter 2 wn, +8192 for b >= 4)
    auto point_a = [&](int slot) {
#pragma unroll
      for (int c = 0; c < 4; ++c) pa[c] = fadr[c] + lds0 + (uint32_t)(slot * PANEL + wm * 8192);
    };
    auto point_b = [&](int slot) {
#pragma unroll
      for (int c = 0; c < 4; ++c) pb[c] = fadr[c] + lds0 + (uint32_t)(slot * PANEL + wn * 16384);
    };
#define TN_RA(f, ks) fa[f] = read_frag<(ks) * 4096>(pa[f]);
#define TN_RB(f, ks) fb[f] = read_frag<(ks) * 4096 + ((f) >> 2) * 8192>(pb[(f) & 3]);

    // ---- prologue.  Ring position of A(g) is 2g, of B(g) 2g+1 (g = k-step), slot = position % 5.
#pragma unroll
    for (int j = 0; j < GL; ++j) glds(a_base, a_off[j], 0, j);                                  // A(0)
    adv(a_t);
#pragma unroll
    for (int j = 0; j < GL; ++j) glds(b_base, b_off[j], 1, j);                                  // B(0)
    adv(b_t);
#pragma unroll
    for (int j = 0; j < GL; ++j) glds(a_base + a_t * a_step, a_off[j], 2, j);                   // A(1)
    adv(a_t);
    wait_vm<GL>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < GL; ++j) glds(b_base + b_t * b_step, b_off[j], 3, j);                   // B(1)
    adv(b_t);
    point_a(0); point_b(1);
    TN_RA(0, 0) TN_RA(1, 0)
    TN_RB(0, 0) TN_RB(1, 0) TN_RB(2, 0) TN_RB(3, 0) TN_RB(4, 0) TN_RB(5, 0) TN_RB(6, 0) TN_RB(7, 0)
    int cs = 0;
    const bool dph = (wave & 4) != 0;      // (TN_DEPHASE)

#define TN_MFMA2(a0, b)                                                                                                              \
    acc[a0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a0]), __builtin_bit_cast(bf16x8, fb[b]),      \
                                                         acc[a0][b], 0, 0, 0);                                                        \
    acc[a0 + 1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a0 + 1]), __builtin_bit_cast(bf16x8, fb[b]), \
                                                             acc[a0 + 1][b], 0, 0, 0);
    // Issue order of the LDS reads (2 instructions per fragment) and where each is consumed:
    //   H1: A2 A3 | MFMA(a=0,1 ; b) needs A0 A1 B[b]      H2: A0' A1' | MFMA(a=2,3 ; b) needs A2 A3 ; then B[b]'
    //   H3: A2' A3' | MFMA(0,1) needs A0' A1' B'[b]       barrier      H4: A0'' A1'' | MFMA(2,3) ; then B[b]''
    // In H1 / H3 the reads younger than B[b] are B[b+1..7] (2 each) and the two A fragments just issued (4): lgkmcnt(4 + 2 (7 - b)),
    // capped at the counter's 15.  In H2 / H4 the first MFMA waits for everything but the 4 reads just issued.
#define TN_H13(b, ks2)                                                                                                             \
    wait_lds3<(4 + 2 * (7 - (b)) > 15 ? 15 : 4 + 2 * (7 - (b)))>(fa[0], fa[1], fb[b]);                                               \
    TN_MFMA2(0, b)
    for (int t = 0; t < nk; ++t) {
      const int slot_b2 = cs, slot_a2 = wrap(cs + 4);
      const int slot_na = wrap(cs + 2), slot_nb = wrap(cs + 3);
      cs = wrap(cs + 2);
      const char* a_k = a_base + a_t * a_step;
      // ---- H1: (t, ks = 0), rows a = 0,1; fetch A2 A3 of the same sub-step; LDS-DMA of A(g+2)
      TN_RA(2, 0) TN_RA(3, 0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b == 0) { TN_H13(0, 0) } else if (b == 1) { TN_H13(1, 0) } else if (b == 2) { TN_H13(2, 0) } else if (b == 3) { TN_H13(3, 0) }
        else if (b == 4) { TN_H13(4, 0) } else if (b == 5) { TN_H13(5, 0) } else if (b == 6) { TN_H13(6, 0) } else { TN_H13(7, 0) }
#if TN_DEPHASE
        if ((b & 1) && !dph) glds(a_k, a_off[b >> 1], slot_a2, b >> 1);
#else
        if (b & 1) glds(a_k, a_off[b >> 1], slot_a2, b >> 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#if !TN_DEPHASE
      adv(a_t);
#endif
      // ---- H2: (t, 0), rows a = 2,3; fetch (t, ks = 1)
      TN_RA(0, 1) TN_RA(1, 1)
      wait_lds<4>(fa[2], fa[3]);
      __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 0) TN_RB(0, 1) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 1) TN_RB(1, 1) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 2) TN_RB(2, 1) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 3) TN_RB(3, 1) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 4) TN_RB(4, 1) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 5) TN_RB(5, 1) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 6) TN_RB(6, 1) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 7) TN_RB(7, 1) __builtin_amdgcn_sched_barrier(0);
      // ---- H3: (t, 1), rows a = 0,1
      TN_RA(2, 1) TN_RA(3, 1)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b == 0) { TN_H13(0, 1) } else if (b == 1) { TN_H13(1, 1) } else if (b == 2) { TN_H13(2, 1) } else if (b == 3) { TN_H13(3, 1) }
        else if (b == 4) { TN_H13(4, 1) } else if (b == 5) { TN_H13(5, 1) } else if (b == 6) { TN_H13(6, 1) } else { TN_H13(7, 1) }
#if TN_DEPHASE
        if ((b & 1) && dph) glds(a_k, a_off[b >> 1], slot_a2, b >> 1);      // (the second wave of every SIMD issues its A pieces here: see gemm_nt.hip NT_DEPHASE)
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#if TN_DEPHASE
      adv(a_t);
#endif
      // ---- barrier_g: A(g+1), B(g+1) landed (A(g+2) may stay in flight); every wave holds all of step g in registers
      wait_vm<GL>();
      wait_lds<0>(fa[2], fa[3]);
      __builtin_amdgcn_s_barrier();
      const char* b_k = b_base + b_t * b_step;
      // ---- H4: (t, 1), rows a = 2,3; fetch (t+1, ks = 0); LDS-DMA of B(g+2)
      point_a(slot_na); point_b(slot_nb);
      TN_RA(0, 0) TN_RA(1, 0)
      __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 0) TN_RB(0, 0) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 1) TN_RB(1, 0) glds(b_k, b_off[0], slot_b2, 0); __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 2) TN_RB(2, 0) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 3) TN_RB(3, 0) glds(b_k, b_off[1], slot_b2, 1); __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 4) TN_RB(4, 0) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 5) TN_RB(5, 0) glds(b_k, b_off[2], slot_b2, 2); __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 6) TN_RB(6, 0) __builtin_amdgcn_sched_barrier(0);
      TN_MFMA2(2, 7) TN_RB(7, 0) glds(b_k, b_off[3], slot_b2, 3); __builtin_amdgcn_sched_barrier(0);
      adv(b_t);
    }
#undef TN_H13
#undef TN_RA
#undef TN_RB
#undef TN_MFMA2
    wait_vm<0>();   // the loader ran ahead: no LDS-DMA may be outstanding when the workgroup releases its LDS
  }

  // ---- partial tile -> this split's f32 slab: acc[a][b][r] = C[m0 + wm*64 + a*16 + lg*4 + r][n0 + wn*128 + b*16 + li]
  float* slab = p.slabs + (int64_t)split * p.M * p.slab_ld;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wm * 64 + a * 16 + lg * 4 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int64_t n = n0 + wn * 128 + b * 16 + li;
        if (n < p.N) slab[m * p.slab_ld + n] = acc[a][b][r] * p.alpha;
      }
    }
}

// out[m][n] (+)= sum_s slab[s][m][n]; VEC: four columns per thread (16-byte loads / stores), f32 output
template <bool VEC>
__global__ void tn_reduce_kernel(const float* __restrict__ slabs, int nsplit, int64_t M, int64_t N, int64_t slab_ld, void* __restrict__ C,
                                 int64_t ldc, int out_dtype, int accumulate) {
  if (VEC) {
    const int64_t n4 = N / 4, total = M * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t m = i / n4, n = (i % n4) * 4;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int s = 0; s < nsplit; ++s) t += *reinterpret_cast<const f32x4*>(slabs + ((int64_t)s * M + m) * slab_ld + n);
      f32x4* c = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + m * ldc + n);
      *c = accumulate ? *c + t : t;
    }
    return;
  }
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i % N;
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += slabs[((int64_t)s * M + m) * slab_ld + n];
    if (out_dtype == DT_F32) {
      float* c = reinterpret_cast<float*>(C) + m * ldc + n;
      *c = accumulate ? *c + t : t;
    } else {
      bf16_t* c = reinterpret_cast<bf16_t*>(C) + m * ldc + n;
      *c = f2bf(accumulate ? bf2f(*c) + t : t);
    }
  }
}

// one 256 x 256 tile per CU (the 160-KiB ring allows one workgroup per CU: more than 256 workgroups would run in two rounds):
// nsplit = floor(256 / tiles), at least 8 k-steps per split
int pick_split(int64_t tiles, int64_t ksteps, int requested) {
  static int64_t wgs = 0;                                  // CTCLIP_TN_WGS=<n>: workgroups (tiles x splits) per launch, A/B timing (default 256 = one per CU)
  if (!wgs) { const char* e = getenv("CTCLIP_TN_WGS"); wgs = e ? atoi(e) : 256; if (wgs < 8 || wgs > 256) wgs = 256; }
  int64_t s = requested > 0 ? requested : wgs / tiles;
  if (s * tiles > wgs) s = wgs / tiles;
  if (s > ksteps / 8) s = ksteps / 8;
  return s < 1 ? 1 : (int)s;
}

}  // namespace

int64_t ctclip_gemm_tn_workspace(int64_t M, int64_t N, int64_t K, int split_k) {
  if (K % TK || K / TK < 32) return 0;
  const int64_t tiles = cdiv(M, TM) * cdiv(N, TN);
  return (int64_t)pick_split(tiles, K / TK, split_k) * M * ((N + 7) / 8 * 8) * 4;
}

// Internal entry used by ctclip_gemm's dispatcher (gemm.hip).  Returns 1 when the shape is not eligible.
int ctclip_gemm_tn_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int accumulate, int split_k, float alpha, void* workspace,
                       int64_t workspace_bytes, hipStream_t stream) {
  if (bias || residual) return 1;
  if (K % TK || K / TK < 32) return 1;                       // a split-K kernel: only worth it for long reductions
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  if (lda < (M + 7) / 8 * 8 || ldb < (N + 7) / 8 * 8 || M < 8 || N < 8) return 1;   // whole 16-byte chunks are read
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;       // 32-bit in-panel byte offsets
  const int64_t ntm = cdiv(M, TM), ntn = cdiv(N, TN), ksteps = K / TK;
  TnParams p{};
  p.nsplit = pick_split(ntm * ntn, ksteps, split_k);
  p.k_per_split = (int)(cdiv(ksteps, p.nsplit) * TK);
  if (p.nsplit < 2 || ntm * ntn * p.nsplit < 128) return 1;
  p.nsplit = (int)cdiv(ksteps * TK, p.k_per_split);          // drop empty trailing splits
  p.slab_ld = (N + 7) / 8 * 8;
  const int64_t need = (int64_t)p.nsplit * M * p.slab_ld * 4;
  if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) % 16)) return 1;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.slabs = reinterpret_cast<float*>(workspace);
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.alpha = alpha;
  p.ntm = (int)ntm; p.ntn = (int)ntn;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NPANEL * PANEL) != hipSuccess) return 1;
    raised = true;
  }
  const int64_t per = (ntm * ntn * p.nsplit + 7) / 8;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(per * 8)), dim3(NTH), NPANEL * PANEL, stream, p);
  int rc = ctclip_check_launch("gemm_tn");
  if (rc) return rc;
  const bool vec = out_dtype == DT_F32 && (N % 4) == 0 && (ldc % 4) == 0 && (reinterpret_cast<uintptr_t>(C) % 16) == 0;
  int64_t nb = cdiv(vec ? M * N / 4 : M * N, 256); if (nb > 4096) nb = 4096;
  if (vec) hipLaunchKernelGGL(tn_reduce_kernel<true>, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)p.slabs, p.nsplit, M, N, p.slab_ld, C, ldc, out_dtype, accumulate);
  else hipLaunchKernelGGL(tn_reduce_kernel<false>, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)p.slabs, p.nsplit, M, N, p.slab_ld, C, ldc, out_dtype, accumulate);
  return ctclip_check_launch("gemm_tn reduce");
}
