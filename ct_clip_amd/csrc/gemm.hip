// MFMA GEMM for gfx950:  C[m,n] = alpha * sum_k A(m,k) * B(n,k)  (+ bias[n]) (+ residual[m,n]) (+ C_old)
//
//   A(m,k) = A[m*lda + k]  (a_kc = 1, "k-contiguous")   or  A[k*lda + m]  (a_kc = 0)
//   B(n,k) = B[n*ldb + k]  (b_kc = 1)                    or  B[k*ldb + n]  (b_kc = 0)
//
// so that the three products of a Linear layer share one kernel body:
//   forward  y  = x  W^T   : (a_kc, b_kc) = (1, 1)
//   grad-in  dx = dy W     : (1, 0)
//   grad-w   dW = dy^T x   : (0, 0)   (contraction over the token axis; split-K slabs + ordered reduce)
//
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128, k-tile = 128 bytes per row
// (64 bf16 / 32 f32).  Tiles are staged global -> VGPR -> LDS ([row][8 x 16-B chunks], chunk index
// XOR-swizzled with row&7) with the next tile's global loads in flight under the current tile's
// MFMAs.  When the contraction index is NOT the contiguous one in memory the transposition happens
// in registers on the way to LDS (dword loads of 2 adjacent rows x 8 k, v_perm packs), so the LDS
// image and the MFMA loop are identical for every layout.
//
// MFMA operand mapping: lane (i = l&15, g = l>>4) reads chunks 2g and 2g+1 of its row.  The hardware
// multiplies element e of lane (i,g) of A with element e of lane (j,g) of B, so any k-assignment is
// valid as long as A and B use the same one; bf16 uses 2 x mfma_f32_16x16x32_bf16 per fragment pair,
// f32 uses 8 x mfma_f32_16x16x4f32 (exact f32 FMA chain -- the parity mode).
#include "common.h"
#include <stdlib.h>

int ctclip_gemm256_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int a_kc, int b_kc, int out_dtype, int res_dtype,
                       int accumulate, int split_k, float alpha, void* workspace, int64_t workspace_bytes, hipStream_t stream);
int64_t ctclip_gemm256_workspace(int64_t M, int64_t N, int64_t K, int split_k);
int ctclip_gemm_tn_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int accumulate, int split_k, float alpha, void* workspace,
                       int64_t workspace_bytes, hipStream_t stream);
int64_t ctclip_gemm_tn_workspace(int64_t M, int64_t N, int64_t K, int split_k);
int ctclip_gemm_nt_argmax_try(const void* A, const void* B, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, float* part_val,
                              int32_t* part_idx, int* nparts, hipStream_t stream, int64_t a_wrap_k);
int ctclip_gemm_sm_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int a_kc, int b_kc, int out_dtype, int res_dtype, int accumulate,
                       float alpha, hipStream_t stream);
int ctclip_gemm_nt_try(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int out_dtype, int res_dtype, int accumulate, float alpha,
                       hipStream_t stream);

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int ROWB = 128;  // bytes of k per tile row

enum { EPI_STD = 0, EPI_ARGMAX = 1 };

struct GemmParams {
  const void* A; const void* B; void* C; const float* bias; const void* residual;
  int64_t M, N, K, lda, ldb, ldc, ldr;
  int out_dtype, res_dtype, accumulate;
  float* slab;      // split-K partial results [split][M][N] (deterministic reduce afterwards); null = no split
  float alpha;
  int k_per_split;  // multiple of the k-tile
  int ntm, ntn;
  // argmax epilogue
  float* part_val; int32_t* part_idx; int nparts;
};

template <typename T> struct Tile;
template <> struct Tile<bf16_t> { static constexpr int BK = 64, CE = 8; };
template <> struct Tile<float> { static constexpr int BK = 32, CE = 4; };

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }

// ---- staging: global -> registers (16 dwords per operand per thread)
template <typename T, bool KC>
__device__ __forceinline__ void stage_load(const T* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows,
                                           int64_t k0, int64_t kend, uint32_t (&r)[16]) {
  const int t = threadIdx.x;
  constexpr int CE = Tile<T>::CE;
  if constexpr (KC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = t + NTHREADS * j, row = idx >> 3, chunk = idx & 7;
      const int64_t gr = row0 + row, gk = k0 + chunk * CE;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (gr < nrows && gk < kend) {
        if (gk + CE <= kend) {
          v = *reinterpret_cast<const u32x4*>(base + gr * ld + gk);
        } else {  // K tail inside a 16-byte chunk: element-wise, zero filled
          const T* src = base + gr * ld + gk;
          const int nrem = (int)(kend - gk);
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = e < nrem ? __float_as_uint(reinterpret_cast<const float*>(src)[e]) : 0u;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint32_t h = e < nrem ? (uint32_t)reinterpret_cast<const uint16_t*>(src)[e] : 0u;
              v[e >> 1] |= h << ((e & 1) * 16);
            }
          }
        }
      }
      r[4 * j + 0] = v[0]; r[4 * j + 1] = v[1]; r[4 * j + 2] = v[2]; r[4 * j + 3] = v[3];
    }
  } else if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = t + NTHREADS * j, m = idx & 127, kg = idx >> 7;
      const int64_t gr = row0 + m;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t gk = k0 + kg * 4 + e;
        r[4 * j + e] = (gr < nrows && gk < kend) ? *reinterpret_cast<const uint32_t*>(base + gk * ld + gr) : 0u;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = t + NTHREADS * j, mp = idx & 63, kg = idx >> 6;
      const int64_t gr = row0 + 2 * mp;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t gk = k0 + kg * 8 + e;
        r[8 * j + e] = (gr < nrows && gk < kend) ? *reinterpret_cast<const uint32_t*>(base + gk * ld + gr) : 0u;
      }
    }
  }
}

// ---- staging: registers -> LDS tile
template <typename T, bool KC>
__device__ __forceinline__ void stage_store(char* __restrict__ lds, const uint32_t (&r)[16]) {
  const int t = threadIdx.x;
  if constexpr (KC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = t + NTHREADS * j, row = idx >> 3, chunk = idx & 7;
      u32x4 v = {r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
      *reinterpret_cast<u32x4*>(lds + swz(row, chunk)) = v;
    }
  } else if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = t + NTHREADS * j, m = idx & 127, kg = idx >> 7;
      u32x4 v = {r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
      *reinterpret_cast<u32x4*>(lds + swz(m, kg)) = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = t + NTHREADS * j, mp = idx & 63, kg = idx >> 6;
      u32x4 lo, hi;  // lo: row 2mp (low halves of each dword), hi: row 2mp+1
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t d0 = r[8 * j + 2 * e], d1 = r[8 * j + 2 * e + 1];
        lo[e] = (d0 & 0xffffu) | (d1 << 16);
        hi[e] = (d0 >> 16) | (d1 & 0xffff0000u);
      }
      *reinterpret_cast<u32x4*>(lds + swz(2 * mp, kg)) = lo;
      *reinterpret_cast<u32x4*>(lds + swz(2 * mp + 1, kg)) = hi;
    }
  }
}

template <typename T, bool AKC, bool BKC, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * BM * ROWB];
  char* ldsA = lds;
  char* ldsB = lds + BM * ROWB;
  constexpr int BK = Tile<T>::BK;

  // XCD-aware tile order: consecutive tile ids (which share an A row panel) stay on one XCD / L2.
  const int ntiles = p.ntm * p.ntn;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, within = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tm = bid / p.ntn, tn = bid % p.ntn;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t kbeg = (int64_t)blockIdx.y * p.k_per_split;
  const int64_t kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
  if (kbeg >= kend) return;

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lg = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint32_t ra[16], rb[16];
  stage_load<T, AKC>(A, p.lda, m0, p.M, kbeg, kend, ra);
  stage_load<T, BKC>(B, p.ldb, n0, p.N, kbeg, kend, rb);
  stage_store<T, AKC>(ldsA, ra);
  stage_store<T, BKC>(ldsB, rb);
  __syncthreads();

  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = (k0 + BK) < kend;
    if (more) {
      stage_load<T, AKC>(A, p.lda, m0, p.M, k0 + BK, kend, ra);
      stage_load<T, BKC>(B, p.ldb, n0, p.N, k0 + BK, kend, rb);
    }
    // fragments: lane (li, lg) reads chunks 2lg, 2lg+1 of its row
    u32x4 af[4][2], bfr[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int ar = wm * 64 + f * 16 + li, br = wn * 64 + f * 16 + li;
      af[f][0] = *reinterpret_cast<const u32x4*>(ldsA + swz(ar, 2 * lg));
      af[f][1] = *reinterpret_cast<const u32x4*>(ldsA + swz(ar, 2 * lg + 1));
      bfr[f][0] = *reinterpret_cast<const u32x4*>(ldsB + swz(br, 2 * lg));
      bfr[f][1] = *reinterpret_cast<const u32x4*>(ldsB + swz(br, 2 * lg + 1));
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a][s]),
                                                                __builtin_bit_cast(bf16x8, bfr[b][s]), acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[a][s][e]),
                                                               __uint_as_float(bfr[b][s][e]), acc[a][b], 0, 0, 0);
        }
      }
    __syncthreads();
    if (more) {
      stage_store<T, AKC>(ldsA, ra);
      stage_store<T, BKC>(ldsB, rb);
      __syncthreads();
    }
  }

  // ---- epilogue.  acc[a][b][r] is C[m0 + wm*64 + a*16 + lg*4 + r][n0 + wn*64 + b*16 + li]
  if constexpr (EPI == EPI_STD) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 64 + a * 16 + lg * 4 + r;
        if (row >= p.M) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int64_t col = n0 + wn * 64 + b * 16 + li;
          if (col >= p.N) continue;
          float v = acc[a][b][r] * p.alpha;
          if (blockIdx.y == 0) {
            if (p.bias) v += p.bias[col];
            if (p.residual) {
              v += (p.res_dtype == DT_F32) ? reinterpret_cast<const float*>(p.residual)[row * p.ldr + col]
                                           : bf2f(reinterpret_cast<const bf16_t*>(p.residual)[row * p.ldr + col]);
            }
          }
          if (p.out_dtype == DT_F32) {
            float* c = reinterpret_cast<float*>(p.C) + row * p.ldc + col;
            if (p.slab) p.slab[((int64_t)blockIdx.y * p.M + row) * p.N + col] = v;
            else *c = p.accumulate ? (*c + v) : v;
          } else {
            bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + col;
            *c = f2bf(p.accumulate ? (bf2f(*c) + v) : v);
          }
        }
      }
  } else {
    // row-wise arg-max over this wave's 64 columns -> partial (value, index) per (row, tn*2+wn);
    // ties resolve to the lowest column index (torch.argmax semantics on CPU).
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float best = -INFINITY; int bidx = 0x7fffffff;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int64_t col = n0 + wn * 64 + b * 16 + li;
          const float v = acc[a][b][r];
          if (col < p.N && (v > best || (v == best && (int)col < bidx))) { best = v; bidx = (int)col; }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bidx, o, 64);
          if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        const int64_t row = m0 + wm * 64 + a * 16 + lg * 4 + r;
        if (li == 0 && row < p.M) {
          p.part_val[row * p.nparts + tn * 2 + wn] = best;
          p.part_idx[row * p.nparts + tn * 2 + wn] = bidx;
        }
      }
  }
}

__global__ void argmax_reduce_kernel(const float* __restrict__ pv, const int32_t* __restrict__ pi, int64_t* __restrict__ out,
                                     float* __restrict__ out_val, int64_t M, int nparts) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  float best = -INFINITY; int bidx = 0x7fffffff;
  for (int i = 0; i < nparts; ++i) {
    const float v = pv[row * nparts + i]; const int ix = pi[row * nparts + i];
    if (v > best || (v == best && ix < bidx)) { best = v; bidx = ix; }
  }
  out[row] = bidx == 0x7fffffff ? 0 : bidx;       // a row of NaNs compares false everywhere: keep the index in range for the gather that follows
  if (out_val) out_val[row] = best;
}

// C (+)= sum over splits (in order) of slab[split]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int nsplit, float* __restrict__ C, int64_t ldc,
                                                            int64_t M, int64_t N, int accumulate) {
  const int64_t n = M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float t = 0.f;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {                        // four slabs in flight, added in slab order
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = slab[(int64_t)(s + k) * n + i];
#pragma unroll
      for (int k = 0; k < 4; ++k) t += v[k];
    }
    for (; s < nsplit; ++s) t += slab[(int64_t)s * n + i];
    float* c = C + (i / N) * ldc + i % N;
    *c = accumulate ? *c + t : t;
  }
}

// split factor of the generic kernel for (M, N, K): `split_k` <= 0 = auto (~2 resident 128^2 blocks per CU, f32 output only)
int generic_split(int64_t M, int64_t N, int64_t K, int in_dtype, int out_dtype, int split_k, int* k_per_split) {
  const int bk = in_dtype == DT_F32 ? 32 : 64;
  const int64_t ktiles = cdiv(K, bk);
  if (split_k <= 0) {
    const int64_t tiles = cdiv(M, BM) * cdiv(N, BN);
    split_k = (out_dtype == DT_F32 && tiles < 256) ? (int)(512 / tiles) : 1;
    if (split_k > ktiles / 4) split_k = (int)(ktiles / 4);
  }
  if (split_k < 1) split_k = 1;
  if (split_k > ktiles) split_k = (int)ktiles;
  const int kps = (int)(cdiv(ktiles, split_k) * bk);
  if (k_per_split) *k_per_split = kps;
  return (int)cdiv(K, kps);
}

template <typename T, int EPI>
int launch_layout(const GemmParams& p, int a_kc, int b_kc, dim3 grid, hipStream_t stream) {
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, true, EPI>), grid, dim3(NTHREADS), 0, stream, p);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, false, EPI>), grid, dim3(NTHREADS), 0, stream, p);
  else if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<T, false, false, EPI>), grid, dim3(NTHREADS), 0, stream, p);
  else hipLaunchKernelGGL((gemm_kernel<T, false, true, EPI>), grid, dim3(NTHREADS), 0, stream, p);
  return ctclip_check_launch("gemm");
}

int check_operands(const void* A, const void* B, int64_t lda, int64_t ldb, int64_t M, int64_t N, int64_t K, int a_kc,
                   int b_kc, int in_dtype) {
  if (!A || !B || M <= 0 || N <= 0 || K <= 0) { ctclip_set_error("gemm: null operand or empty shape"); return CTCLIP_EBADARG; }
  if (in_dtype != DT_F32 && in_dtype != DT_BF16) { ctclip_set_error("gemm: unsupported dtype"); return CTCLIP_EUNSUPPORTED; }
  const int es = in_dtype == DT_F32 ? 4 : 2, ce = 16 / es;
  auto bad16 = [&](const void* ptr, int64_t ld) { return ((uintptr_t)ptr % 16) != 0 || (ld % ce) != 0; };
  // non-k-contiguous bf16 operands are read as dwords (2 adjacent rows): the pitch must be even and cover the
  // row count rounded up to even (an odd last row reads one in-bounds element past it and discards the result)
  auto bad4 = [&](const void* ptr, int64_t ld, int64_t rows) { return ((uintptr_t)ptr % 4) != 0 || (es == 2 && ((ld % 2) || (ld < rows + (rows & 1)))); };
  if (a_kc ? bad16(A, lda) : bad4(A, lda, M)) { ctclip_set_error("gemm: A alignment (k-contiguous operands need 16-byte aligned rows; others an even pitch)"); return CTCLIP_EBADARG; }
  if (b_kc ? bad16(B, ldb) : bad4(B, ldb, N)) { ctclip_set_error("gemm: B alignment (k-contiguous operands need 16-byte aligned rows; others an even pitch)"); return CTCLIP_EBADARG; }
  return CTCLIP_OK;
}

}  // namespace

// C-ABI ----------------------------------------------------------------------------------------
// Replaces torch F.linear / nn.Linear forward+backward on the reference hot path
// (attention.py:48,51,119,120,125; ctvit.py:173; ct_clip.py:549,762; HF modeling_bert dense layers).
extern "C" int ctclip_gemm(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M,
                           int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int a_kc, int b_kc,
                           int in_dtype, int out_dtype, int res_dtype, int accumulate, int split_k, float alpha,
                           void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  int rc = check_operands(A, B, lda, ldb, M, N, K, a_kc, b_kc, in_dtype);
  if (rc) return rc;
  if (!C) { ctclip_set_error("gemm: null C"); return CTCLIP_EBADARG; }
  if (in_dtype == DT_BF16 && a_kc && b_kc && split_k <= 1) {   // persistent LDS-DMA ring kernel (gemm_nt.hip) for the big forward / grad-input GEMMs
    rc = ctclip_gemm_nt_try(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, out_dtype, res_dtype, accumulate, alpha, stream);
    if (rc != 1) return rc;
  }
  static int use_tn = -1;
  if (use_tn < 0) { const char* e = getenv("CTCLIP_GEMM_TN"); use_tn = (e && e[0] == '0') ? 0 : 1; }
  if (use_tn && in_dtype == DT_BF16 && !a_kc && !b_kc) {       // weight gradients: split-K kernel with transposing LDS reads (gemm_tn.hip)
    rc = ctclip_gemm_tn_try(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, out_dtype, accumulate, split_k, alpha, workspace,
                            workspace_bytes, stream);
    if (rc != 1) return rc;
  }
  if (in_dtype == DT_BF16 && a_kc == b_kc) {   // the text tower's sizes (M = B * T rows): one-tile-per-workgroup LDS-DMA kernel (gemm_sm.hip)
    rc = ctclip_gemm_sm_try(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, a_kc, b_kc, out_dtype, res_dtype, accumulate, alpha, stream);
    if (rc != 1) return rc;
  }
  if (in_dtype == DT_BF16) {   // large-tile fast path (gemm256.hip) when the shape fills the chip
    rc = ctclip_gemm256_try(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, a_kc, b_kc, out_dtype, res_dtype, accumulate,
                            split_k, alpha, workspace, workspace_bytes, stream);
    if (rc != 1) return rc;
  }
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.residual = residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
  p.out_dtype = out_dtype; p.res_dtype = res_dtype; p.accumulate = accumulate; p.alpha = alpha;
  p.ntm = (int)cdiv(M, BM); p.ntn = (int)cdiv(N, BN);
  if (split_k > 1 && out_dtype != DT_F32) { ctclip_set_error("gemm: split-K needs f32 output"); return CTCLIP_EUNSUPPORTED; }
  split_k = generic_split(M, N, K, in_dtype, out_dtype, split_k, &p.k_per_split);
  const int64_t slab_bytes = (int64_t)split_k * M * N * 4;
  if (split_k > 1) {      // split-K partial sums go to slabs and are reduced in a fixed order: there is no float-atomic path in this library
    if (!workspace || workspace_bytes < slab_bytes) { ctclip_set_error("gemm: split-K needs ctclip_gemm_workspace() bytes of workspace"); return CTCLIP_EWORKSPACE; }
    p.slab = (float*)workspace;
  }
  dim3 grid(p.ntm * p.ntn, split_k);
  rc = in_dtype == DT_F32 ? launch_layout<float, EPI_STD>(p, a_kc, b_kc, grid, stream) : launch_layout<bf16_t, EPI_STD>(p, a_kc, b_kc, grid, stream);
  if (rc || !p.slab) return rc;
  int64_t nb = cdiv(M * N, 256); if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)p.slab, split_k, (float*)C, ldc, M, N, accumulate);
  return ctclip_check_launch("gemm split-K reduce");
}

// bytes of optional workspace for ctclip_gemm (split-K partial slabs of the large-tile bf16 path); split_k <= 0 means "auto"
extern "C" int64_t ctclip_gemm_workspace(int64_t M, int64_t N, int64_t K, int in_dtype, int split_k) {
  const int gs = generic_split(M, N, K, in_dtype, DT_F32, split_k, nullptr);
  const int64_t wgen = gs > 1 ? (int64_t)gs * M * N * 4 : 0;         // generic kernel: split-K slabs
  if (in_dtype != DT_BF16) return wgen;
  const int64_t w256 = ctclip_gemm256_workspace(M, N, K, split_k), wtn = ctclip_gemm_tn_workspace(M, N, K, split_k);
  const int64_t w = w256 > wtn ? w256 : wtn;
  return w > wgen ? w : wgen;       // (layout-agnostic entry point: enough for whichever kernel the dispatcher picks)
}

// Row-wise arg-max of A B^T without materialising the product (vector-quantiser code assignment:
// vector_quantize_pytorch CosineSimCodebook.forward, called at ctvit.py:403).
// workspace: M * nparts * 8 bytes, nparts = 2 * ceil(N / 128).
extern "C" int64_t ctclip_gemm_argmax_workspace(int64_t M, int64_t N) { return M * 2 * cdiv(N, BN) * 8; }

extern "C" int ctclip_gemm_argmax(const void* A, const void* B, int64_t* out_idx, float* out_val, int64_t M, int64_t N,
                                  int64_t K, int64_t lda, int64_t ldb, int in_dtype, void* workspace,
                                  int64_t workspace_bytes, hipStream_t stream) {
  int rc = check_operands(A, B, lda, ldb, M, N, K, 1, 1, in_dtype);
  if (rc) return rc;
  if (workspace_bytes < ctclip_gemm_argmax_workspace(M, N) || !workspace) { ctclip_set_error("gemm_argmax: workspace too small"); return CTCLIP_EWORKSPACE; }
  GemmParams p{};
  p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.alpha = 1.f;
  p.ntm = (int)cdiv(M, BM); p.ntn = (int)cdiv(N, BN);
  const int bk = in_dtype == DT_F32 ? 32 : 64;
  p.k_per_split = (int)(cdiv(K, bk) * bk);
  p.nparts = 2 * p.ntn;
  p.part_val = reinterpret_cast<float*>(workspace);
  p.part_idx = reinterpret_cast<int32_t*>(p.part_val + M * p.nparts);
  rc = 1;
  if (in_dtype == DT_BF16) {   // large code searches run on the persistent NT kernel (2.6 ms -> MFMA-bound for the 110592 x 8192 VQ search)
    int np = 0;
    rc = ctclip_gemm_nt_argmax_try(A, B, M, N, K, lda, ldb, p.part_val, reinterpret_cast<int32_t*>(p.part_val + M * 2 * cdiv(N, 256)), &np, stream, 0);
    if (rc == 0) { p.nparts = np; p.part_idx = reinterpret_cast<int32_t*>(p.part_val + M * np); }
  }
  if (rc == 1) {
    dim3 grid(p.ntm * p.ntn, 1);
    rc = (in_dtype == DT_F32) ? launch_layout<float, EPI_ARGMAX>(p, 1, 1, grid, stream)
                              : launch_layout<bf16_t, EPI_ARGMAX>(p, 1, 1, grid, stream);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(argmax_reduce_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, stream, p.part_val, p.part_idx, out_idx,
                     out_val, M, p.nparts);
  return ctclip_check_launch("argmax_reduce");
}

// The same search on the RAW bf16 tokens: B2 is (C, 2 K), row c = [bf16(e_c) | bf16(e_c - hi)] of the unit code e_c (ctclip_l2norm_split3 order 2),
// A the (M, K) tokens as they are: arg-max_c a . e_c = arg-max_c a^ . e_c (the norm of a row is a positive factor), and a bf16 token has no low
// part -- two products per (token, code, dim) instead of the three of the [hi | hi | lo] x [hi | lo | hi] expansion, and no expanded copy of the
// tokens: the persistent NT kernel reads A a second time for the k-steps behind K (epilogue family 5: [a | a] . [hi | lo] with 2 K / 64 k-steps per
// tile).  Round 6: the 110 592 x 8 192 search 2.3 -> 1.2 ms.  out_val = the winning a . (hi + lo) (NOT normalised by |a|).  bf16 only, shapes the
// persistent NT kernel takes (K % 64 == 0, ceil(M / 256) x ceil(C / 256) >= 160 tiles); others: CTCLIP_EUNSUPPORTED (the caller keeps the
// three-term form).  workspace >= ctclip_gemm_argmax_workspace(M, C).
extern "C" int ctclip_gemm_argmax_hilo(const void* A, const void* B2, int64_t* out_idx, float* out_val, int64_t M, int64_t C, int64_t K,
                                       int64_t lda, int64_t ldb, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  int rc = check_operands(A, B2, lda, ldb, M, C, K, 1, 1, DT_BF16);
  if (rc) return rc;
  if (ldb < 2 * K) { ctclip_set_error("gemm_argmax_hilo: B2 rows hold [hi | lo]: ldb >= 2 K"); return CTCLIP_EBADARG; }
  if (workspace_bytes < ctclip_gemm_argmax_workspace(M, C) || !workspace) { ctclip_set_error("gemm_argmax_hilo: workspace too small"); return CTCLIP_EWORKSPACE; }
  float* part_val = reinterpret_cast<float*>(workspace);
  int np = 0;
  rc = ctclip_gemm_nt_argmax_try(A, B2, M, C, 2 * K, lda, ldb, part_val, reinterpret_cast<int32_t*>(part_val + M * 2 * cdiv(C, 256)), &np, stream, K);
  if (rc == 1) { ctclip_set_error("gemm_argmax_hilo: shape not served by the persistent NT kernel (K % 64 == 0, >= 160 tiles of 256 x 256, 16-byte aligned rows)"); return CTCLIP_EUNSUPPORTED; }
  if (rc) return rc;
  hipLaunchKernelGGL(argmax_reduce_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, stream, part_val, reinterpret_cast<int32_t*>(part_val + M * np), out_idx,
                     out_val, M, np);
  return ctclip_check_launch("argmax_reduce");
}

// ---------------------------------------------------------------------------------------------------- fused GEGLU in-projection
int ctclip_gemm_nt_geglu_try(const void* A, const void* B, void* U, void* G, const void* dG, int64_t M, int hp, int64_t K, int64_t lda,
                             int64_t ldb, int64_t ldu, int64_t ldg, int64_t lddg, hipStream_t stream);
int ctclip_gemm_nt_headnorm_try(const void* A, const void* B, int64_t M, int nsec, int64_t K, int64_t lda, int64_t ldb, void* const* out,
                                float* const* inv, const float* const* scale, const float* mult, hipStream_t stream);
int ctclip_gemm_nt_rescomp_try(const void* A, const void* B, void* C, void* E, const void* residual, const void* comp,
                               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, hipStream_t stream);
int ctclip_gemm_nt_dgeglu_try(const void* A, const void* B, const void* U, void* dU, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb,
                              int64_t ldu, int64_t lddu, hipStream_t stream);

namespace {
__global__ __launch_bounds__(256) void geglu_weight_interleave_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int inner, int hp, int K, int64_t ldo) {
  const int64_t n4 = (int64_t)2 * hp * (K / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / (K / 4)), k = (int)(i % (K / 4)) * 4;
    const int j = 4 * (n >> 3) + (n & 3), part = (n >> 2) & 1;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < inner) load4(w + (int64_t)(part * inner + j) * K + k, v);
    store4(out + (int64_t)n * ldo + k, v);
  }
}
}  // namespace

// FeedForward[1].weight (2 * inner, K) f32 = [x rows | gate rows] (attention.py:48) -> the bf16 operand of ctclip_gemm_geglu:
// (2 * hp, ldo >= K) with row 8 q + r = x row 4 q + r and row 8 q + 4 + r = gate row 4 q + r (r < 4); rows of features >= inner are zero.
extern "C" int ctclip_geglu_weight_interleave(const float* w, void* out, int inner, int hp, int K, int64_t ldo, hipStream_t stream) {
  if (!w || !out || inner < 1 || hp < inner || hp % 4 || K % 4 || ldo < K || ldo % 4) { ctclip_set_error("geglu_weight_interleave: hp % 4 == 0, K % 4 == 0"); return CTCLIP_EBADARG; }
  int64_t nb = cdiv((int64_t)2 * hp * (K / 4), 256); if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(geglu_weight_interleave_kernel, dim3((unsigned)nb), dim3(256), 0, stream, w, (bf16_t*)out, inner, hp, K, ldo);
  return ctclip_check_launch("geglu_weight_interleave");
}

// FeedForward in-projection + GEGLU in one launch (attention.py:39-48): g (M, ldg >= hp) = x * gelu_erf(gate) and, when U is not
// NULL, u (M, ldu >= 2 hp) = [x | gate] = A B^T (the layout ctclip_geglu_bwd reads); A (M, lda) bf16, B = ctclip_geglu_weight_interleave's
// output.  Training passes U = NULL and differentiates with ctclip_gemm_geglu_bwd (nothing but the layer input is kept).
// Returns CTCLIP_EUNSUPPORTED when the shape does not fill whole 256 x 256 tiles of the large-tile kernel: the caller then runs
// ctclip_gemm + ctclip_geglu_fwd.
extern "C" int ctclip_gemm_geglu(const void* A, const void* B, void* U, void* G, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb,
                                 int64_t ldu, int64_t ldg, int dtype, hipStream_t stream) {
  if (!A || !B || !G || M < 1 || hp < 1 || K < 1) { ctclip_set_error("gemm_geglu: bad args"); return CTCLIP_EBADARG; }
  if (dtype != DT_BF16) return CTCLIP_EUNSUPPORTED;
  const int rc = ctclip_gemm_nt_geglu_try(A, B, U, G, nullptr, M, hp, K, lda, ldb, U ? ldu : 2 * (int64_t)hp, ldg, 4, stream);
  return rc == 1 ? CTCLIP_EUNSUPPORTED : rc;
}

// Backward of ctclip_gemm_geglu through the GEGLU by RECOMPUTATION: the same GEMM A B^T rebuilds (x, gate) in f32 and the epilogue
// writes dU (M, lddu >= 2 hp) = [dG * gelu(gate) | dG * x * gelu'(gate)] (the gradient of u in the split layout the grad-input and
// weight-gradient GEMMs read) from dG (M, lddg >= hp).  Replaces the stored u (2 hp values per token and layer) and the streaming
// ctclip_geglu_bwd pass.  Same eligibility as ctclip_gemm_geglu.
extern "C" int ctclip_gemm_geglu_bwd(const void* A, const void* B, const void* dG, void* dU, int64_t M, int hp, int64_t K, int64_t lda,
                                     int64_t ldb, int64_t lddg, int64_t lddu, int dtype, hipStream_t stream) {
  if (!A || !B || !dG || !dU || M < 1 || hp < 1 || K < 1) { ctclip_set_error("gemm_geglu_bwd: bad args"); return CTCLIP_EBADARG; }
  if (dtype != DT_BF16) return CTCLIP_EUNSUPPORTED;
  const int rc = ctclip_gemm_nt_geglu_try(A, B, dU, nullptr, dG, M, hp, K, lda, ldb, lddu, 8, lddg, stream);
  return rc == 1 ? CTCLIP_EUNSUPPORTED : rc;
}

// Backward of the feed-forward block between the out-projection and the GEGLU in ONE launch: dU (M, lddu >= 2 hp) =
// [dg * gelu(gate) | dg * x * gelu'(gate)] where dg = dY W_out is formed in the accumulators only (A = dY (M, K = model width) bf16,
// B = W_out^T (hp, ldb >= K): hidden feature j in row j) and u = [x | gate] (M, ldu >= 2 hp) is what ctclip_gemm_geglu stored.
// Replaces the grad-input GEMM of FeedForward[4] + ctclip_geglu_bwd (dg: one write and one read of M x hp less, one pass over u and du
// instead of two launches).  CTCLIP_EUNSUPPORTED when the shape does not fill whole 256-row tiles / 128-column halves.
extern "C" int ctclip_gemm_dgeglu(const void* A, const void* B, const void* U, void* dU, int64_t M, int hp, int64_t K, int64_t lda,
                                  int64_t ldb, int64_t ldu, int64_t lddu, int dtype, hipStream_t stream) {
  if (!A || !B || !U || !dU || M < 1 || hp < 1 || K < 1) { ctclip_set_error("gemm_dgeglu: bad args"); return CTCLIP_EBADARG; }
  if (dtype != DT_BF16) return CTCLIP_EUNSUPPORTED;
  const int rc = ctclip_gemm_nt_dgeglu_try(A, B, U, dU, M, hp, K, lda, ldb, ldu, lddu, stream);
  return rc == 1 ? CTCLIP_EUNSUPPORTED : rc;
}

// A residual add of the transformer (attention.py:326,331: x + attn(x), x + ff(x)) on a COMPENSATED residual stream: the stream is the bf16
// pair (x, e), x what every consumer reads, e the rounding residue of the last add.  C (M, ldc) = bf16(s), E (M, ldc) = bf16(s - C) with
// s = A B^T + residual + comp in f32; A (M, K), B (N, K) bf16 k-contiguous, residual / comp rows with stride ldr.
// Replaces `x = out_proj(...) + x` where bf16 storage rounds x at every add: 72 roundings over 24 layers are the bf16 mode's error
// (profiles/r03_bf16_error_budget.md); with the residue carried along they no longer accumulate.  CTCLIP_EUNSUPPORTED unless bf16, whole
// 256-row tiles, N % 128 == 0, K % 64 == 0, 16-byte aligned rows.
extern "C" int ctclip_gemm_residual_comp(const void* A, const void* B, void* C, void* E, const void* residual, const void* comp,
                                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int dtype,
                                         hipStream_t stream) {
  if (!A || !B || !C || !E || !residual || !comp || M < 1 || N < 1 || K < 1) { ctclip_set_error("gemm_residual_comp: bad args"); return CTCLIP_EBADARG; }
  if (dtype != DT_BF16) return CTCLIP_EUNSUPPORTED;
  const int rc = ctclip_gemm_nt_rescomp_try(A, B, C, E, residual, comp, M, N, K, lda, ldb, ldc, ldr, stream);
  return rc == 1 ? CTCLIP_EUNSUPPORTED : rc;
}

// The q and k|v projections of the spatial attention (attention.py:141-154: to_q / to_kv, split heads, l2norm, * q_scale / k_scale) with the
// attention kernels' operand layout written by the GEMM itself: replaces nn.Linear + ctclip_attn2_prep.  A (M, K) bf16, B (nsec * 256, K) bf16
// (k-contiguous); per 256-column section s (8 heads x 32): inv_s != NULL -> out_s[h][m][d] = bf16(a_m . b_n) / max(|head row|, 1e-12) *
// scale_s[d] * mult_s (head-planar [8][M][32] bf16), inv_s[m * 8 + h] = the inverse norm (f32); inv_s == NULL -> out_s = the head-planar copy
// (v).  The arithmetic is ctclip_attn2_prep's on the bf16-rounded projection.  CTCLIP_EUNSUPPORTED unless bf16, M % 256 == 0, K % 64 == 0,
// 1 <= nsec <= 3 and enough tiles to fill the chip.
extern "C" int ctclip_gemm_headnorm(const void* A, const void* B, int64_t M, int nsec, int64_t K, int64_t lda, int64_t ldb,
                                    void* out0, float* inv0, const float* scale0, float mult0, void* out1, float* inv1, const float* scale1,
                                    float mult1, void* out2, float* inv2, const float* scale2, float mult2, int dtype, hipStream_t stream) {
  if (!A || !B || !out0 || M < 1 || K < 1) { ctclip_set_error("gemm_headnorm: bad args"); return CTCLIP_EBADARG; }
  if (dtype != DT_BF16) return CTCLIP_EUNSUPPORTED;
  void* const out[3] = {out0, out1, out2};
  float* const inv[3] = {inv0, inv1, inv2};
  const float* const scale[3] = {scale0, scale1, scale2};
  const float mult[3] = {mult0, mult1, mult2};
  const int rc = ctclip_gemm_nt_headnorm_try(A, B, M, nsec, K, lda, ldb, out, inv, scale, mult, stream);
  return rc == 1 ? CTCLIP_EUNSUPPORTED : rc;
}
