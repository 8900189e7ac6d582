// bf16 GEMMs of the TEXT TOWER on gfx950 (round 5): C[M x N] = A B^T with M = B * T = a few hundred to a few thousand rows.
//
// HF BertModel at batch 8 / 128 tokens is 72 forward GEMMs of 1.2 - 4.8 GFLOP (M = 1 024, N in {768, 2 304, 3 072}, K in {768, 3 072}) and
// twice that backward.  gemm_nt.hip / gemm_tn.hip decline them (a handful of 256 x 256 tiles does not fill 256 CUs) and the generic 128 x 128
// kernel of gemm.hip ran them at 27 - 107 TFLOP/s: 45 - 80 us per launch, 11.4 ms of kernel time per step on the text tower's side stream,
// where they take CUs and L2 away from the image tower for most of the step (HISTORY.md section 8 item 6).  This kernel is built for THAT
// size: one tile per workgroup, no persistence, 4 waves (2 x 2), tile 128 x 128 when that still gives >= 120 workgroups, else 64 x 64
// (so that N = 768 at M = 1 024 is 192 workgroups, not 48), operands by LDS-DMA into a ring of 3 (128 x 128: 96 KB) or 4 (64 x 64: 64 KB)
// stages of one k-step (64 bf16 = one 128-byte line per row), ONE barrier per k-step, fragment reads through inline asm with explicit
// lgkmcnt waits (compiler-visible LDS reads next to LDS-DMA in flight make hipcc drain vmcnt to 0 in front of every read).
//   NT  (a_kc, b_kc: forward, and grad-input against the transposed weight shadow): rows are k-contiguous, ds_read_b128 fragments, LDS
//       layout / swizzle of gemm_nt.hip; the B rows are permuted on the DMA source side so that a lane owns FB consecutive output columns.
//   TN  (!a_kc, !b_kc: weight gradients dW = dy^T x, the reduction runs over the M = B * T token rows): operands are k-major; fragments
//       come from the transposing LDS read ds_read_b64_tr_b16 (layout and semantics of gemm_tn.hip).
// Epilogue: alpha, + bias (f32), + residual (f32 or bf16), (+)= f32 accumulate, f32 or bf16 out -- what the mixed-precision text tower needs
// (bf16 operands, f32 residual stream).  Deterministic: no split-K, one workgroup owns an output tile.
//   TN + colsum (ctclip_gemm_dw_db): the workgroups of the first column tile also sum A over k (one more MFMA per A fragment against a fragment of
//       ones): the bias gradient of a Linear rides its weight-gradient launch -- 144 colsum / reduce launches per step gone from the text stream.
// Measured (profiles/r05_gemm_sm.md): 12 - 32 us per launch (80 - 350 TFLOP/s) against 26 - 92 us on the generic kernel; per BERT layer 578 -> 217 us.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TK = 64;
constexpr int ROWB = 128;
constexpr int NTH = 256;

struct SmParams {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias; const void* residual;
  int64_t M, N, K, lda, ldb, ldc, ldr;
  int out_dtype, res_dtype, accumulate;
  float alpha;
  int ntm, ntn;
  float* colsum;      // TN only: colsum[m] (+)= sum_k A[k][m] -- the bias gradient of the Linear whose weight gradient this GEMM is (same `accumulate`)
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
// one TN fragment (8 k-slots x 16 columns) = two transposing reads 16 k-rows apart (gemm_tn.hip; KSTRIDE = bytes per k-row of the LDS image)
template <int OFF, int KSTRIDE> __device__ __forceinline__ u32x4 lds_read_tr(uint32_t vaddr) {
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(vaddr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(vaddr), "n"(OFF + 16 * KSTRIDE));
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
template <int FA, int FB> __device__ __forceinline__ void wait_frags(u32x4 (&a)[FA], u32x4 (&b)[FB]) {
  if constexpr (FA == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
}

// FB (2 or 4) consecutive elements <-> floats; `vec` (kernel-uniform): the pointers / pitches allow one 8- / 16-byte access
template <int FB> __device__ __forceinline__ void ldf(const float* p, float (&v)[FB], bool vec) {
  if (vec) {
    if constexpr (FB == 4) { const f32x4 q = *reinterpret_cast<const f32x4*>(p); v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; }
    else { const hw_f32x2 q = *reinterpret_cast<const hw_f32x2*>(p); v[0] = q[0]; v[1] = q[1]; }
  } else {
#pragma unroll
    for (int b = 0; b < FB; ++b) v[b] = p[b];
  }
}
template <int FB> __device__ __forceinline__ void ldh(const bf16_t* p, float (&v)[FB], bool vec) {
  if (vec) {
    if constexpr (FB == 4) { const u32x2 q = *reinterpret_cast<const u32x2*>(p); v[0] = __uint_as_float(q[0] << 16); v[1] = __uint_as_float(q[0] & 0xffff0000u); v[2] = __uint_as_float(q[1] << 16); v[3] = __uint_as_float(q[1] & 0xffff0000u); }
    else { const uint32_t q = *reinterpret_cast<const uint32_t*>(p); v[0] = __uint_as_float(q << 16); v[1] = __uint_as_float(q & 0xffff0000u); }
  } else {
#pragma unroll
    for (int b = 0; b < FB; ++b) v[b] = bf2f(p[b]);
  }
}
template <int FB> __device__ __forceinline__ void stf(float* p, const float (&v)[FB], bool vec) {
  if (vec) {
    if constexpr (FB == 4) *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    else *reinterpret_cast<hw_f32x2*>(p) = hw_f32x2{v[0], v[1]};
  } else {
#pragma unroll
    for (int b = 0; b < FB; ++b) p[b] = v[b];
  }
}
template <int FB> __device__ __forceinline__ void sth(bf16_t* p, const float (&v)[FB], bool vec) {
  if (vec) {
    if constexpr (FB == 4) *reinterpret_cast<u32x2*>(p) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    else *reinterpret_cast<uint32_t*>(p) = pack2bf(v[0], v[1]);
  } else {
#pragma unroll
    for (int b = 0; b < FB; ++b) p[b] = f2bf(v[b]);
  }
}

// TRANS = false: NT.  A (M, lda) and B (N, ldb) k-contiguous.  LDS stage = [BM rows of A | BN rows of B], 128 B per row (one k-step).
// TRANS = true: TN.  A = [K][M] (lda), B = [K][N] (ldb).  LDS stage = [A image | B image], image = [column block of 64][64 k][128 B].
template <int BM, int BN, int NS, bool TRANS>
__global__ __launch_bounds__(NTH) void gemm_sm_kernel(SmParams p) {
  static_assert(BM == BN && (BM == 128 || BM == 64), "square tiles of 128 or 64");
  constexpr int WM = BM / 2, WN = BN / 2;                 // wave tile
  constexpr int FA = WM / 16, FB = WN / 16;               // fragments per wave and k-sub-step
  constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
  constexpr int PW = (BM + BN) / 32;                      // LDS-DMA pieces (1 KiB) per wave and stage: waves 0,1 load A, waves 2,3 load B
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int nk = (int)(p.K / TK);
  // tile id -> (m, n): consecutive ids share the A row panel; ids are dealt to XCDs round-robin by the hardware, so ids are re-ordered such
  // that one XCD (blockIdx % 8) works on neighbouring tiles
  const int G = gridDim.x;
  const int id = (G & 7) == 0 ? (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;      // (a bijection when G % 8 == 0; plain order otherwise)
  const int64_t m0 = (int64_t)(id / p.ntn) * BM, n0 = (int64_t)(id % p.ntn) * BN;

  // ---- loader: this wave's PW pieces of every stage, all from one operand
  const bool loads_a = wave < 2;
  const int pw0 = (wave & 1) * PW;                        // first piece inside the operand's block
  uint32_t off[PW];
  const char* base;
  if (!TRANS) {
    const int64_t rows = loads_a ? p.M - m0 : p.N - n0;   // valid rows of this tile
    const int64_t ld = loads_a ? p.lda : p.ldb;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int rho = (pw0 + j) * 8 + (lane >> 3);        // LDS row inside the operand's block
      const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
      int trow = rho;
      if (!loads_a) trow = (rho / WN) * WN + (rho & 15) * FB + ((rho % WN) >> 4);      // LDS row rho holds tile column ... (lane owns FB consecutive columns)
      if (trow > rows - 1) trow = (int)(rows - 1);        // rows past the end: clamped (never branch around a load), never stored
      off[j] = (uint32_t)trow * (uint32_t)(ld * 2) + (uint32_t)(chunk * 16);
    }
    base = to_sgpr(reinterpret_cast<const char*>(loads_a ? p.A + m0 * p.lda : p.B + n0 * p.ldb));
  } else {
    // image [cb = column block of 64][k = 0..63][128 B]: a piece = 8 k-rows of one column block; lane l -> k-row (l >> 3), 16-byte chunk (l & 7),
    // the four 32-byte sub-chunks of a row XOR-swizzled with (k >> 1) & 3 on the source side (gemm_tn.hip)
    const int64_t cols = loads_a ? p.M - m0 : p.N - n0;
    const int64_t ld = loads_a ? p.lda : p.ldb;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int piece = pw0 + j;                          // BM / 64 column blocks x 8 pieces
      const int cb = piece >> 3, k = (piece & 7) * 8 + (lane >> 3);
      const int c16 = lane & 7;                           // destination chunk
      const int src16 = (((c16 >> 1) ^ ((k >> 1) & 3)) << 1) | (c16 & 1);
      int64_t col = cb * 64 + src16 * 8;
      if (col > cols - 8) col = cols - 8 > 0 ? cols - 8 : 0;      // (whole 16-byte chunks: M, N multiples of 8; columns past the end are clamped, never stored)
      off[j] = (uint32_t)k * (uint32_t)(ld * 2) + (uint32_t)(col * 2);
    }
    base = to_sgpr(reinterpret_cast<const char*>(loads_a ? p.A + m0 : p.B + n0));
  }
  const int64_t kstep_bytes = TRANS ? (int64_t)TK * (loads_a ? p.lda : p.ldb) * 2 : (int64_t)TK * 2;
  const int ldsbase = (loads_a ? 0 : A_BYTES) + pw0 * 1024;
  auto issue = [&](int stage, int step) {
    const char* src = base + (int64_t)step * kstep_bytes;
#pragma unroll
    for (int j = 0; j < PW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint64_t)off[j]),
                                       (__attribute__((address_space(3))) void*)(lds + stage * STAGE + ldsbase + j * 1024), 16, 0, 0);
  };

  // ---- fragment addresses (lane part; the stage offset is added per step)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  uint32_t pa[2], pb[2];
  uint32_t ta[FA], tb[FB];                                // TN only: per-fragment offsets (column block + swizzled 32-byte sub-chunk)
  if (!TRANS) {
    pa[0] = lds0 + (uint32_t)(wm * WM * ROWB) + (uint32_t)swz(li, lg);
    pa[1] = lds0 + (uint32_t)(wm * WM * ROWB) + (uint32_t)swz(li, 4 + lg);
    pb[0] = lds0 + (uint32_t)(A_BYTES + wn * WN * ROWB) + (uint32_t)swz(li, lg);
    pb[1] = lds0 + (uint32_t)(A_BYTES + wn * WN * ROWB) + (uint32_t)swz(li, 4 + lg);
#pragma unroll
    for (int f = 0; f < FA; ++f) ta[f] = 0;
#pragma unroll
    for (int f = 0; f < FB; ++f) tb[f] = 0;
  } else {
    // ds_read_b64_tr_b16 (gemm_tn.hip): in a 16-lane group, lane t points at k-row k0 + (t >> 2), 8-byte piece (t & 3) of a 32-byte chunk of 16
    // columns; lane i receives column c0 + i for k0 .. k0 + 3.  Lane group g takes k0 = 4 g (first read) and 16 + 4 g (second read), sub-step
    // ks adds 32 rows.  The 32-byte sub-chunk s of a row sits at physical position s ^ ((k >> 1) & 3), identical for k, k + 16, k + 32.
    const int krow = lg * 4 + (li >> 2);
    const int ksw = (krow >> 1) & 3;
    pa[0] = lds0 + (uint32_t)(krow * ROWB + (li & 3) * 8);  pa[1] = pa[0] + 32 * ROWB;
    pb[0] = pa[0] + (uint32_t)A_BYTES;                      pb[1] = pb[0] + 32 * ROWB;
#pragma unroll
    for (int f = 0; f < FA; ++f) { const int c = wm * WM + f * 16; ta[f] = (uint32_t)((c >> 6) * 64 * ROWB + ((((c >> 4) & 3) ^ ksw) << 5)); }
#pragma unroll
    for (int f = 0; f < FB; ++f) { const int c = wn * WN + f * 16; tb[f] = (uint32_t)((c >> 6) * 64 * ROWB + ((((c >> 4) & 3) ^ ksw) << 5)); }
  }

  u32x4 fa0[FA], fb0[FB], fa1[FA], fb1[FB];
  f32x4 acc[FA][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // TN + colsum: the workgroups of the first column tile also sum A over k -- one more MFMA per A fragment against a fragment of ones
  // (D[i][j] = sum_k A[k][i] for every j); only the waves wn == 0 of those workgroups (both wn waves of a row block hold the same A fragments)
  const bool do_colsum = TRANS && p.colsum != nullptr && n0 == 0 && wn == 0;
  f32x4 accb[FA];
#pragma unroll
  for (int a = 0; a < FA; ++a) accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

  auto read_set = [&](u32x4 (&fa_)[FA], u32x4 (&fb_)[FB], int ks, uint32_t so) {      // so: byte offset of the stage
    if constexpr (!TRANS) {
      const uint32_t aa = pa[ks] + so, bb = pb[ks] + so;
      fa_[0] = lds_read16<0>(aa); fa_[1] = lds_read16<2048>(aa);
      if constexpr (FA == 4) { fa_[2] = lds_read16<4096>(aa); fa_[3] = lds_read16<6144>(aa); }
      fb_[0] = lds_read16<0>(bb); fb_[1] = lds_read16<2048>(bb);
      if constexpr (FB == 4) { fb_[2] = lds_read16<4096>(bb); fb_[3] = lds_read16<6144>(bb); }
    } else {
#pragma unroll
      for (int f = 0; f < FA; ++f) fa_[f] = lds_read_tr<0, ROWB>(pa[ks] + so + ta[f]);
#pragma unroll
      for (int f = 0; f < FB; ++f) fb_[f] = lds_read_tr<0, ROWB>(pb[ks] + so + tb[f]);
    }
  };
  auto mfma_set = [&](u32x4 (&fa_)[FA], u32x4 (&fb_)[FB]) {
#pragma unroll
    for (int b = 0; b < FB; ++b)
#pragma unroll
      for (int a = 0; a < FA; ++a)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa_[a]), __builtin_bit_cast(bf16x8, fb_[b]), acc[a][b], 0, 0, 0);
    if (TRANS && do_colsum) {      // (wave-uniform)
#pragma unroll
      for (int a = 0; a < FA; ++a)
        accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa_[a]), __builtin_bit_cast(bf16x8, ones), accb[a], 0, 0, 0);
    }
  };

  // ---- prologue: NS - 1 steps in flight (steps past the end re-load the last one: the vmcnt arithmetic stays uniform)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s, s < nk ? s : nk - 1);
  int stage = 0, fill = NS - 1;
  for (int g = 0; g < nk; ++g) {
    wait_vm<(NS - 2) * PW>();                             // all but the newest NS - 2 steps of this wave's pieces have landed: step g is in LDS
    __builtin_amdgcn_s_barrier();                         // ... everybody's pieces; and every wave has finished reading step g - 1's stage
    {
      const int nxt = g + NS - 1;
      issue(fill, nxt < nk ? nxt : nk - 1);               // into the stage step g - 1 used
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
    const uint32_t so = (uint32_t)(stage * STAGE);
    read_set(fa0, fb0, 0, so);
    __builtin_amdgcn_sched_barrier(0);
    wait_frags<FA, FB>(fa0, fb0);
    read_set(fa1, fb1, 1, so);                            // sub-step 1's fragments arrive under sub-step 0's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    wait_frags<FA, FB>(fa1, fb1);
    mfma_set(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    stage = stage + 1 == NS ? 0 : stage + 1;
  }
  wait_vm<0>();                                           // (re-loads of the last step may still be in flight: not when the LDS is released)

  // ---- epilogue: NT: acc[a][b][r] = C[m0 + wm WM + 16 a + 4 lg + r][n0 + wn WN + li FB + b]; TN: same rows, column n0 + wn WN + 16 b + li
  const int64_t rbase = m0 + wm * WM + lg * 4;
  if (TRANS && do_colsum && li == 0) {
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = rbase + a * 16 + r;
        if (row < p.M) p.colsum[row] = (p.accumulate ? p.colsum[row] : 0.f) + accb[a][r] * p.alpha;
      }
  }
  // (kernel-uniform) every row of C / residual starts FB-element aligned for the widest access used: 16 bytes covers all of them
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(p.C) | (uintptr_t)(p.ldc * 2)) % 16 == 0) && (!p.bias || reinterpret_cast<uintptr_t>(p.bias) % 16 == 0) &&
                      (!p.residual || ((reinterpret_cast<uintptr_t>(p.residual) | (uintptr_t)(p.ldr * 2)) % 16 == 0));
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = rbase + a * 16 + r;
      if (row >= p.M) continue;
      if (!TRANS) {
        const int64_t col = n0 + wn * WN + li * FB;
        if (col >= p.N) continue;                         // (N % 8 == 0 and FB | 8: a lane's FB columns are all inside or all outside)
        float v[FB];
#pragma unroll
        for (int b = 0; b < FB; ++b) v[b] = acc[a][b][r] * p.alpha;
        if (p.bias) {
          float t[FB];
          ldf<FB>(p.bias + col, t, vec_ok);
#pragma unroll
          for (int b = 0; b < FB; ++b) v[b] += t[b];
        }
        if (p.residual) {
          float t[FB];
          if (p.res_dtype == DT_F32) ldf<FB>(reinterpret_cast<const float*>(p.residual) + row * p.ldr + col, t, vec_ok);
          else ldh<FB>(reinterpret_cast<const bf16_t*>(p.residual) + row * p.ldr + col, t, vec_ok);
#pragma unroll
          for (int b = 0; b < FB; ++b) v[b] += t[b];
        }
        if (p.out_dtype == DT_F32) {
          float* c = reinterpret_cast<float*>(p.C) + row * p.ldc + col;
          if (p.accumulate) {
            float t[FB];
            ldf<FB>(c, t, vec_ok);
#pragma unroll
            for (int b = 0; b < FB; ++b) v[b] += t[b];
          }
          stf<FB>(c, v, vec_ok);
        } else {
          sth<FB>(reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col, v, vec_ok);
        }
      } else {
#pragma unroll
        for (int b = 0; b < FB; ++b) {
          const int64_t col = n0 + wn * WN + b * 16 + li;
          if (col >= p.N) continue;
          float v = acc[a][b][r] * p.alpha;
          if (p.out_dtype == DT_F32) {
            float* c = reinterpret_cast<float*>(p.C) + row * p.ldc + col;
            *c = p.accumulate ? *c + v : v;
          } else {
            reinterpret_cast<bf16_t*>(p.C)[row * p.ldc + col] = f2bf(v);
          }
        }
      }
    }
}

template <int BM, int NS, bool TRANS>
int sm_launch(const SmParams& p, hipStream_t stream) {
  constexpr int LDSB = NS * 2 * BM * ROWB;
  static bool raised_dev[64] = {};                        // (the attribute is per device: one process may drive several)
  int dev = 0; (void)hipGetDevice(&dev);
  bool& raised = raised_dev[(dev >= 0 && dev < 64) ? dev : 0];
  if (!raised) {
    if (hipFuncSetAttribute((const void*)gemm_sm_kernel<BM, BM, NS, TRANS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess) return 1;
    raised = true;
  }
  hipLaunchKernelGGL((gemm_sm_kernel<BM, BM, NS, TRANS>), dim3((unsigned)(p.ntm * p.ntn)), dim3(NTH), LDSB, stream, p);
  return ctclip_check_launch("gemm_sm");
}

int sm_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CTCLIP_GEMM_SM"); on = (e && e[0] == '0') ? 0 : 1; }
  return on;
}

}  // namespace

static int sm_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                  int64_t ldc, int64_t ldr, int a_kc, int b_kc, int out_dtype, int res_dtype, int accumulate, float alpha, float* colsum, hipStream_t stream);

// Weight AND bias gradient of a Linear layer in one launch: dW (N_out x K_in, f32, ldw) (+)= dy^T x and db (N_out, f32) (+)= column sums of dy, with dy =
// (T, N_out) and x = (T, K_in) bf16, k-major (a row is one token).  [replaces autograd through nn.Linear with bias: HF BertSelfAttention query / key /
// value, BertSelfOutput.dense, BertIntermediate.dense, BertOutput.dense -- the reference's text tower, ct_clip.py:685-686]  The column sums ride the
// A fragments of the first column tile's workgroups (one more MFMA per fragment against ones): no second pass over dy, no reduce launch.
// CTCLIP_EUNSUPPORTED when the shape is not served by this kernel (callers fall back to ctclip_gemm (0,0) + ctclip_colsum).
extern "C" int ctclip_gemm_dw_db(const void* dy, const void* x, float* dW, float* db, int64_t T, int64_t n_out, int64_t k_in, int64_t lddy, int64_t ldx,
                                 int64_t ldw, int accumulate, hipStream_t stream) {
  if (!dy || !x || !dW || !db) { ctclip_set_error("gemm_dw_db: null argument"); return CTCLIP_EBADARG; }
  const int rc = sm_try(dy, x, dW, nullptr, nullptr, n_out, k_in, T, lddy, ldx, ldw, 0, 0, 0, DT_F32, 0, accumulate, 1.f, db, stream);
  if (rc == 1) { ctclip_set_error("gemm_dw_db: shape not served"); return CTCLIP_EUNSUPPORTED; }
  return rc;
}

// Internal entry used by ctclip_gemm's dispatcher (gemm.hip) for the shapes the persistent kernels decline.  Returns 1 when not eligible.
// a_kc && b_kc: NT; !a_kc && !b_kc: TN (M, N = rows / columns of C; K = the reduction = rows of both operands).
int ctclip_gemm_sm_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int a_kc, int b_kc, int out_dtype, int res_dtype, int accumulate,
                       float alpha, hipStream_t stream) {
  if (!sm_enabled()) return 1;
  return sm_try(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, a_kc, b_kc, out_dtype, res_dtype, accumulate, alpha, nullptr, stream);
}

static int sm_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                  int64_t ldc, int64_t ldr, int a_kc, int b_kc, int out_dtype, int res_dtype, int accumulate, float alpha, float* colsum, hipStream_t stream) {
  if (a_kc != b_kc) return 1;
  const bool trans = !a_kc;
  if (K % TK || K < 2 * TK || M < 64 || N < 64 || (N % 8) || (trans && (M % 8))) return 1;
  if ((reinterpret_cast<uintptr_t>(A) % 16) || (reinterpret_cast<uintptr_t>(B) % 16) || (lda % 8) || (ldb % 8)) return 1;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return 1;
  if (trans && (bias || residual)) return 1;
  if (trans && (lda < M || ldb < N)) return 1;
  if (accumulate && out_dtype != DT_F32) return 1;
  if (out_dtype == DT_BF16 && ((ldc % 2) || (reinterpret_cast<uintptr_t>(C) % 4))) return 1;
  if ((int64_t)M * N > ((int64_t)1 << 26)) return 1;                 // big outputs belong to the persistent kernels
  SmParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.out_dtype = out_dtype; p.res_dtype = res_dtype; p.accumulate = accumulate; p.alpha = alpha; p.colsum = colsum;
  const int64_t t128 = cdiv(M, 128) * cdiv(N, 128);
  if (t128 >= 120) {
    p.ntm = (int)cdiv(M, 128); p.ntn = (int)cdiv(N, 128);
    return trans ? sm_launch<128, 3, true>(p, stream) : sm_launch<128, 3, false>(p, stream);
  }
  p.ntm = (int)cdiv(M, 64); p.ntn = (int)cdiv(N, 64);
  if ((int64_t)p.ntm * p.ntn < 8) return 1;
  return trans ? sm_launch<64, 4, true>(p, stream) : sm_launch<64, 4, false>(p, stream);
}
