// Large-tile bf16 MFMA GEMM for gfx950 (the hot projection / feed-forward / weight-gradient GEMMs of CTViT and BERT).
//
//   C[m,n] = alpha * sum_k A(m,k) B(n,k) (+ bias[n]) (+ residual[m,n]) (+ C_old)      -- same contract as gemm.hip
//
// 256 x 256 block tile, BK = 64, 512 threads = 8 waves (2 x 4, 128 x 64 per wave, 8 x 4 accumulator fragments of
// mfma_f32_16x16x32_bf16), one workgroup per CU (128 KiB of LDS = 2 stages x (A 32 KiB + B 32 KiB)).
// The 128^2 tile of gemm.hip needs 32 KiB of L1->LDS traffic per 2.1 MFLOP, which is as many L1 cycles (64 B/clk) as
// MFMA cycles; this tile halves that.
//
// Staging
//   * k-contiguous operands: __builtin_amdgcn_global_load_lds, 16 B per lane, straight into LDS (no VGPR round trip).
//     The LDS image is lane-linear per wave-instruction (8 rows x 128 B), so the bank-conflict XOR swizzle is applied to
//     the per-lane SOURCE address (chunk ^ (row & 7)) and again on the fragment read.  Rows past M / N are clamped to the
//     last valid row (their results are never stored); K must be a multiple of 64.
//   * operands whose contiguous index is NOT the contraction index (dX = dY W, dW = dY^T X): dword loads of 2 adjacent
//     rows x 8 k, transposed in registers (v_perm-style packs) and written with ds_write_b128 into the same image.
//   The next stage's loads are issued before the current stage's MFMAs; one barrier per k-step.
// Epilogue: accumulators -> LDS (per-wave 128 x 64 tile) -> 16-byte coalesced row stores with bias / residual fused;
//   split-K partial sums go to per-split f32 slabs + an ordered reduce (no atomics).
#include "common.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64, NT256 = 512;
constexpr int ROWB = 128;
constexpr int STAGE_BYTES = (TM + TN) * ROWB;  // 64 KiB

struct Gemm256Params {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias; const void* residual;
  int64_t M, N, K, lda, ldb, ldc, ldr;
  int out_dtype, res_dtype, accumulate, nsplit;
  float alpha;
  int k_per_split, ntm, ntn;
  float* slabs;          // nsplit x (M x slab_ld) f32 partial results when nsplit > 1
  int64_t slab_ld;
};

// 16-byte chunk index XOR-swizzled with row bits 0-2 (fragment reads: 16 consecutive rows, same chunk) and bits 3-5
// (transposing stores: rows 8 apart, same chunk)
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7) ^ ((row >> 3) & 7)) << 4); }

// k-contiguous operand: 256 rows x 128 B = 32 pieces of 1 KiB; wave w issues pieces 4w .. 4w+3
__device__ __forceinline__ void stage_glds(char* lds, const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows,
                                           int64_t k0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = wave * 4 + j;
    const int r = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (r & 7) ^ ((r >> 3) & 7);
    int64_t gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;
    const bf16_t* src = base + gr * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
  }
}

// non-k-contiguous operand (global is [k][row], rows contiguous): the 256-row x 64-k tile is 32 row-groups x 8 k-groups
// = 256 units of (8 rows x 8 k); ONE unit per thread of a 256-thread half of the block: 8 x 16-byte loads (one per k),
// an 8x8 16-bit transpose in registers, 8 x ds_write_b128 (one 16-byte k-chunk per row).
__device__ __forceinline__ void stage_tr_load(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                              int u, uint32_t (&r)[32]) {
  // UNCONDITIONAL loads (a branch around each load makes hipcc wait vmcnt(0) per element and serialises the 8 round trips):
  // row groups past the end are clamped to the last group; the pitch covers nrows rounded up to 8 (checked by the host),
  // so a partial last group reads in-bounds padding.  Whatever lands in rows >= nrows only feeds outputs that are never stored.
  const int rg = u & 31, kg = u >> 5;
  int64_t gr = row0 + rg * 8;
  const int64_t last = (nrows - 1) & ~(int64_t)7;
  gr = gr < last ? gr : last;
  const bf16_t* src = base + (k0 + kg * 8) * ld + gr;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + e * ld);
    r[4 * e + 0] = v[0]; r[4 * e + 1] = v[1]; r[4 * e + 2] = v[2]; r[4 * e + 3] = v[3];
  }
}
__device__ __forceinline__ void stage_tr_store(char* lds, int u, const uint32_t (&r)[32]) {
  const int rg = u & 31, kg = u >> 5;
  // r[4*e + j] holds rows (2j, 2j+1) at k = e.  Row q's chunk: dword d = k (2d, 2d+1).
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x4 lo, hi;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t d0 = r[4 * (2 * d) + j], d1 = r[4 * (2 * d + 1) + j];
      lo[d] = (d0 & 0xffffu) | (d1 << 16);
      hi[d] = (d0 >> 16) | (d1 & 0xffff0000u);
    }
    *reinterpret_cast<u32x4*>(lds + swz(rg * 8 + 2 * j, kg)) = lo;
    *reinterpret_cast<u32x4*>(lds + swz(rg * 8 + 2 * j + 1, kg)) = hi;
  }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(NT256) void gemm256_kernel(Gemm256Params p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int ntiles = p.ntm * p.ntn;
  int bid = blockIdx.x, split = 0;
  if (p.nsplit == 1) {
    // XCD-aware tile order: consecutive tile ids (sharing an A row panel) stay on one XCD / L2
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, within = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  } else if ((p.nsplit & 7) == 0) {
    // split-K: all output tiles of one k-range run on ONE XCD at the same time, so the re-reads of that k-range by the
    // other tiles hit its L2 and HBM sees each operand element once.  XCD x owns splits x, x+8, ...
    const int xcd = bid % 8, j = bid / 8;
    split = xcd + 8 * (j / ntiles);
    bid = j % ntiles;
  } else {
    split = bid / ntiles;
    bid = bid % ntiles;
  }
  const int tm = bid / p.ntn, tn = bid % p.ntn;
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;
  const int64_t kbeg = (int64_t)split * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  const bool empty = kbeg >= kend;   // (possible only for a trailing split: it still has to write a zero slab)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 15, lg = lane >> 4;

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing staging: one (8 rows x 8 k) unit per thread; with both operands transposed, waves 0-3 take A and 4-7 take B
  uint32_t rt[(AKC && BKC) ? 1 : 32];
  const int tu = threadIdx.x & 255;
  const bool lowhalf = wave < 4;
  auto issue = [&](int stage, int64_t k0) {
    char* sA = lds + stage * STAGE_BYTES;
    char* sB = sA + TM * ROWB;
    if constexpr (AKC) stage_glds(sA, p.A, p.lda, m0, p.M, k0, wave, lane);
    if constexpr (BKC) stage_glds(sB, p.B, p.ldb, n0, p.N, k0, wave, lane);
    if constexpr (!AKC && !BKC) {
      if (lowhalf) stage_tr_load(p.A, p.lda, m0, p.M, k0, tu, rt); else stage_tr_load(p.B, p.ldb, n0, p.N, k0, tu, rt);
    } else if constexpr (!BKC) {
      if (lowhalf) stage_tr_load(p.B, p.ldb, n0, p.N, k0, tu, rt);
    }
  };
  auto commit = [&](int stage) {
    char* sA = lds + stage * STAGE_BYTES;
    char* sB = sA + TM * ROWB;
    if constexpr (!AKC && !BKC) {
      stage_tr_store(lowhalf ? sA : sB, tu, rt);
    } else if constexpr (!BKC) {
      if (lowhalf) stage_tr_store(sB, tu, rt);
    }
  };

  if (!empty) {
    issue(0, kbeg);
    commit(0);
  }
  __syncthreads();

  int cur = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += TK) {
    const bool more = (k0 + TK) < kend;
    if (more) issue(cur ^ 1, k0 + TK);
    const char* sA = lds + cur * STAGE_BYTES;
    const char* sB = sA + TM * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 bfr[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) bfr[f] = *reinterpret_cast<const u32x4*>(sB + swz(wn * 64 + f * 16 + li, 2 * lg + ks));
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(sA + swz(wm * 128 + a * 16 + li, 2 * lg + ks));
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[b]),
                                                              acc[a][b], 0, 0, 0);
      }
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---------------- epilogue: acc[a][b][r] = C[m0 + wm*128 + a*16 + lg*4 + r][n0 + wn*64 + b*16 + li]
  // split-K: every split writes its partial tile into its own f32 slab (plain vector stores, deterministic);
  // splitk_reduce_kernel then sums the slabs and applies bias / residual / accumulate.
  if (p.nsplit > 1) {
    p.C = p.slabs + (int64_t)split * p.M * p.slab_ld;
    p.ldc = p.slab_ld; p.out_dtype = DT_F32; p.accumulate = 0; p.bias = nullptr; p.residual = nullptr;
  }
  // stage the wave's 128 x 64 f32 tile through LDS in four quarters of 32 rows (row pitch 65 floats, 8320 B per wave)
  float* wbuf = reinterpret_cast<float*>(lds) + wave * (32 * 68);   // pitch 68 floats: conflict-free column writes, 16-B aligned rows
  const bool vec_ok = ((p.ldc % 8) == 0) && ((reinterpret_cast<uintptr_t>(p.C) % 16) == 0) &&
                      (!p.residual || (((p.ldr % 8) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) % 16) == 0)));
  // (the main loop ended with a block barrier: nobody reads the stage buffers any more.  From here on every wave works in
  //  its PRIVATE wbuf region, so only wave-local ordering of LDS writes -> reads is needed, no block barriers.)
#pragma unroll
  for (int hh = 0; hh < 4; ++hh) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) wbuf[(a * 16 + lg * 4 + r) * 68 + b * 16 + li] = acc[hh * 2 + a][b][r] * p.alpha;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // 32 rows x 64 cols: a lane handles 8 consecutive columns of one row; 8 lanes per row, 8 rows per pass, 4 passes
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rr = pass * 8 + (lane >> 3), cc = (lane & 7) * 8;
      const int64_t row = m0 + wm * 128 + hh * 32 + rr, col = n0 + wn * 64 + cc;
      if (row >= p.M || col >= p.N) continue;
      float v[8];
      load8(wbuf + rr * 68 + cc, v);
      const bool full = vec_ok && (col + 8 <= p.N);
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (col + e < p.N) v[e] += p.bias[col + e];
      }
      if (p.residual) {
        if (full) {
          float rv[8];
          if (p.res_dtype == DT_F32) load8(reinterpret_cast<const float*>(p.residual) + row * p.ldr + col, rv);
          else load8(reinterpret_cast<const bf16_t*>(p.residual) + row * p.ldr + col, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < p.N)
              v[e] += (p.res_dtype == DT_F32) ? reinterpret_cast<const float*>(p.residual)[row * p.ldr + col + e]
                                              : bf2f(reinterpret_cast<const bf16_t*>(p.residual)[row * p.ldr + col + e]);
        }
      }
      if (p.out_dtype == DT_F32) {
        float* c = reinterpret_cast<float*>(p.C) + row * p.ldc + col;
        if (full) {
          if (p.accumulate) { float old[8]; load8(c, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += old[e]; }
          store8(c, v);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (col + e < p.N) c[e] = p.accumulate ? c[e] + v[e] : v[e];
        }
      } else {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col;
        if (full) {
          if (p.accumulate) { float old[8]; load8(c, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += old[e]; }
          store8(c, v);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (col + e < p.N) c[e] = f2bf(p.accumulate ? bf2f(c[e]) + v[e] : v[e]);
        }
      }
    }
  }
}

// out[m][n] (+)= sum_s slab[s][m][n] + bias[n] + residual[m][n]
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int nsplit, int64_t M, int64_t N, int64_t slab_ld, void* __restrict__ C,
                                     int64_t ldc, int out_dtype, int accumulate, const float* __restrict__ bias,
                                     const void* __restrict__ residual, int64_t ldr, int res_dtype) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i % N;
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += slabs[((int64_t)s * M + m) * slab_ld + n];
    if (bias) t += bias[n];
    if (residual) t += (res_dtype == DT_F32) ? reinterpret_cast<const float*>(residual)[m * ldr + n]
                                             : bf2f(reinterpret_cast<const bf16_t*>(residual)[m * ldr + n]);
    if (out_dtype == DT_F32) {
      float* c = reinterpret_cast<float*>(C) + m * ldc + n;
      *c = accumulate ? *c + t : t;
    } else {
      bf16_t* c = reinterpret_cast<bf16_t*>(C) + m * ldc + n;
      *c = f2bf(accumulate ? bf2f(*c) + t : t);
    }
  }
}

template <bool AKC, bool BKC>
int launch256(const Gemm256Params& p, dim3 grid, hipStream_t stream) {
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<AKC, BKC>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES) != hipSuccess) {
      ctclip_set_error("gemm256: cannot raise the dynamic LDS limit to 128 KiB");
      return -1000;
    }
    raised = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<AKC, BKC>), grid, dim3(NT256), 2 * STAGE_BYTES, stream, p);
  return ctclip_check_launch("gemm256");
}

}  // namespace

static bool aligned16(const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0 && (ld % 8) == 0; }

// split-K factor of the large-tile kernel: fill the 256 CUs once (one 256 x 256 tile per CU) when the output is small
static int auto_split(int64_t tiles, int64_t ktiles, int requested) {
  int s = requested;
  if (s <= 0) s = tiles >= 128 ? 1 : (int)(256 / tiles);
  if (s > ktiles / 4) s = (int)(ktiles / 4);
  if (s >= 8) s &= ~7;   // whole XCDs (see the split-K block mapping in the kernel)
  return s < 1 ? 1 : s;
}

// bytes of workspace the large-tile path needs for (M, N, K, split_k) -- 0 when it would not be used / not split
int64_t ctclip_gemm256_workspace(int64_t M, int64_t N, int64_t K, int split_k) {
  if (K % TK) return 0;
  const int64_t tiles = cdiv(M, TM) * cdiv(N, TN);
  const int s = auto_split(tiles, K / TK, split_k);
  return s > 1 ? (int64_t)s * M * ((N + 7) / 8 * 8) * 4 : 0;
}

// Internal entry used by ctclip_gemm's dispatcher (gemm.hip).  Returns 1 when the shape is not eligible.
int ctclip_gemm256_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int a_kc, int b_kc, int out_dtype, int res_dtype,
                       int accumulate, int split_k, float alpha, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (K % TK) return 1;
  if (!a_kc && b_kc) return 1;  // (0,1) layout is not on the hot path
  if (!aligned16(A, lda) || !aligned16(B, ldb)) return 1;
  if (!a_kc && lda < (M + 7) / 8 * 8) return 1;   // transposing loads read whole 8-row groups (see stage_tr_load)
  if (!b_kc && ldb < (N + 7) / 8 * 8) return 1;
  const int64_t ntm = cdiv(M, TM), ntn = cdiv(N, TN);
  const int64_t ktiles = K / TK;
  const int ns = auto_split(ntm * ntn, ktiles, split_k);
  // big tiles only pay when they fill the chip; small problems stay on the 128^2 kernel
  if (ntm * ntn * ns < 160) return 1;
  Gemm256Params p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.out_dtype = out_dtype; p.res_dtype = res_dtype; p.accumulate = accumulate; p.alpha = alpha;
  p.ntm = (int)ntm; p.ntn = (int)ntn;
  p.k_per_split = (int)(cdiv(ktiles, ns) * TK);
  p.nsplit = (int)cdiv(K, p.k_per_split);
  if (p.nsplit >= 8 && (p.nsplit & 7)) {   // keep the whole-XCD property after rounding k_per_split up
    p.nsplit &= ~7;
    p.k_per_split = (int)(cdiv(ktiles, p.nsplit) * TK);
    if ((int64_t)p.nsplit * p.k_per_split < K) return 1;
  }
  if (p.nsplit > 1) {
    p.slab_ld = (N + 7) / 8 * 8;
    const int64_t need = (int64_t)p.nsplit * M * p.slab_ld * 4;
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) % 16)) return 1;   // caller did not provide slabs
    p.slabs = reinterpret_cast<float*>(workspace);
  }
  dim3 grid((unsigned)(ntm * ntn * p.nsplit), 1);
  int rc;
  if (a_kc && b_kc) rc = launch256<true, true>(p, grid, stream);
  else if (a_kc && !b_kc) rc = launch256<true, false>(p, grid, stream);
  else rc = launch256<false, false>(p, grid, stream);
  if (rc || p.nsplit == 1) return rc;
  int64_t nb = cdiv(M * N, 256); if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)p.slabs, p.nsplit, M, N, p.slab_ld, C, ldc,
                     out_dtype, accumulate, bias, residual, ldr, res_dtype);
  return ctclip_check_launch("splitk_reduce");
}
