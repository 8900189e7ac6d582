// Cosine-similarity attention (attention.py:145-178: l2norm(q) * q_scale, l2norm(k) * k_scale, sim * scale, softmax, @ v) for SHORT
// sequences: L <= 32 tokens, d_head = 32, no bias, no mask, bf16 -- CTViT's temporal transformer (4608 sequences x 24 frames x 8 heads
// per volume batch of 8; ctvit.py:187,303, attention.py:280-333).
//
// The general kernels spend this case on layout passes (qk-norm, head transposes, delta, a 128-key tile loop): 615 us per layer
// forward + backward.  A whole (sequence, head) problem is ONE 32 x 32 MFMA tile, so here one wave owns one problem end to end:
//   * it reads the token-major q (M, H*32) and kv (M, 2*H*32) rows directly and writes o / dq / dkv directly: 226 MB forward,
//     ~400 MB backward per layer, nothing else touches HBM (no transposed copies, no lse, no delta, no planar operands);
//   * HBM <-> LDS in 16-byte lanes: a problem's operand is 32 rows x 64 B at a 512-B (1-KB) stride.  LDS-DMA (global_load_lds, four
//     lanes per row) brings it in with 16 cache lines per instruction, asynchronously, one problem ahead of the arithmetic; results
//     leave through a staging tile as 16-byte buffer stores.  (The first version loaded layout R straight from global memory with
//     8-byte lanes: 32 lines per instruction, 512 line accesses per problem, and ran at the L1's line rate -- 66 us forward, the
//     same as the kernels it replaced.)  Rows past the sequence fetch its last row (finite values under zero weights) and their
//     stores carry an out-of-range buffer offset: every problem issues the same instructions, so the counted vmcnt is exact;
//   * "layout R": lane (row = lane & 31, half = lane >> 5) holds the 16 head dims 8 j + 4 half + r of its token row (four ds_read_b64).
//     That is at once the MFMA operand order for contractions over d (slot (t, i) <-> d = 8 (2t + i/4) + 4 half + i%4, the
//     same map on both operands) and the accumulator order of a 32 x 32 result column, so l2norm forward / backward, the softmax
//     and the learned-scale gradients are lane-local plus ONE exchange with lane ^ 32;
//   * contractions over tokens take their B operand straight from the accumulators (P, dS) and their A operand (V^T, K^T, Q^T, dO^T)
//     from a 2-KB row-major LDS tile private to the wave through ds_read_b64_tr_b16; no barriers in the problem loop.  The 16-byte
//     chunks of a tile row are XOR-swizzled with (row >> 2) & 3 (on the DMA's source side), which keeps the row reads at two
//     accesses per bank and the transposing reads conflict-free;
//   * backward recomputes the 32 x 32 softmax in both orientations (queries as columns for dQ, keys as columns for dK / dV) and
//     passes the row statistics between them through 384 bytes of LDS; delta = rowsum(P o dP) (no O needed);
//   * the gradients of q_scale / k_scale accumulate in registers over all problems a wave visits (persistent grid), are folded over
//     the 32 token lanes and the waves of a workgroup once at the end and summed over workgroups by a second kernel in a fixed
//     order: deterministic, no atomics.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int WPB_F = 4, WPB_B = 3;         // waves per workgroup (each independent): forward 10 KB, backward 16.4 KB of LDS per wave

struct Frag { bf16x8 v[2]; };
struct Row { u32x2 w[4]; };                  // 16 bf16 of layout R: w[j] = dims 8 j + 4 half + 0..3
typedef __attribute__((address_space(3))) char* lds_ptr;

__device__ __forceinline__ f32x16 mma(const Frag& a, const Frag& b) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0], b.v[0], acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1], b.v[1], acc, 0, 0, 0);
}
__device__ __forceinline__ Frag frag_of(const Row& r) {
  Frag f;
  f.v[0] = __builtin_bit_cast(bf16x8, u32x4{r.w[0][0], r.w[0][1], r.w[1][0], r.w[1][1]});
  f.v[1] = __builtin_bit_cast(bf16x8, u32x4{r.w[2][0], r.w[2][1], r.w[3][0], r.w[3][1]});
  return f;
}
__device__ __forceinline__ Row pack_row(const float (&x)[16]) {
  Row r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r.w[j][0] = pack2bf(x[4 * j], x[4 * j + 1]); r.w[j][1] = pack2bf(x[4 * j + 2], x[4 * j + 3]); }
  return r;
}
__device__ __forceinline__ void unpack_row(const Row& r, float (&x)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[4 * j] = __uint_as_float(r.w[j][0] << 16); x[4 * j + 1] = __uint_as_float(r.w[j][0] & 0xffff0000u);
    x[4 * j + 2] = __uint_as_float(r.w[j][1] << 16); x[4 * j + 3] = __uint_as_float(r.w[j][1] & 0xffff0000u);
  }
}

// ---- the 2-KB tile [32 tokens][64 B]: 16-byte chunk c of row r lives at chunk position c ^ ((r >> 2) & 3)
__device__ __forceinline__ int tile_off(int row, int byte) { return row * 64 + ((((byte >> 4) ^ (row >> 2)) & 3) << 4) + (byte & 15); }
__device__ __forceinline__ Row tile_read(const char* tile, int row, int half) {
  Row r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.w[j] = *reinterpret_cast<const u32x2*>(tile + tile_off(row, 16 * j + 8 * half));
  return r;
}
__device__ __forceinline__ void tile_write(char* tile, int row, int half, const Row& r) {
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x2*>(tile + tile_off(row, 16 * j + 8 * half)) = r.w[j];
}
// HBM -> tile, two DMA instructions: lane l of instruction e fills chunk position l & 3 of row 16 e + (l >> 2) with the 16 bytes the
// swizzle assigns to it.  slice = the head's 64-byte slice of the sequence's first token, ld in elements; rows >= L re-read row L - 1.
__device__ __forceinline__ void tile_fetch(char* tile, const bf16_t* slice, int64_t ld, int L, int lane) {
  const lds_ptr t3 = (lds_ptr)tile;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 16 * e + (lane >> 2), c = (lane ^ (row >> 2)) & 3;
    const int rr = row < L ? row : L - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slice + rr * ld + c * 8),
                                     (__attribute__((address_space(3))) void*)(t3 + e * 1024), 16, 0, 0);
  }
}
// tile -> HBM, two 16-byte buffer stores (rows >= L: offset beyond the descriptor, dropped by the hardware)
__device__ __forceinline__ void tile_store(const char* tile, bf16_t* seq_base, int64_t ld, int head_off, int L, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(seq_base, 0, (int)(L * ld * 2), 0x00020000);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 16 * e + (lane >> 2), c = (lane ^ (row >> 2)) & 3;
    const u32x4 v = *reinterpret_cast<const u32x4*>(tile + e * 1024 + lane * 16);
    const uint32_t off = row < L ? (uint32_t)((row * ld + head_off + c * 8) * 2) : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
  }
}
#ifndef STRICT_VMCNT
#define STRICT_VMCNT 0      // 1 = every counted vmcnt wait becomes vmcnt(0) (determinism bisection builds, tools/trace_determinism.py)
#endif
// vmcnt(M) only.  An asm statement with a memory clobber, not the s_waitcnt builtin: the optimiser may move the builtin across LDS reads
// (see rows_landed below)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(STRICT_VMCNT ? 0 : N) : "memory"); }
// "The tile reads have returned": lgkmcnt(0) as an asm statement that CARRIES the rows and clobbers memory.  The next instruction after it
// is the LDS-DMA of the wave's next problem INTO THE SAME TILES (write after read).  The first version was the bare builtin
// __builtin_amdgcn_s_waitcnt(0xC07F): hipcc sank it below the conditional fetch block (the builtin touches no memory as far as the
// optimiser knows), so in the steady-state loop the DMA was issued with the four ds_reads of q / k still in flight.  Almost always
// harmless -- the reads return in ~100 cycles, the DMA needs an L2 round trip -- but about one problem in 10^7 read the NEXT problem's
// bytes: the rare run-to-run difference of the bf16 step in rounds 1-2 (located with tools/trace_determinism.py,
// profiles/r03_determinism.md).  Operands pin the loads in front of the wait, the memory clobber pins the DMA behind it.
__device__ __forceinline__ void rows_landed(Row& a, Row& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]), "+v"(a.w[3]), "+v"(b.w[0]), "+v"(b.w[1]), "+v"(b.w[2]), "+v"(b.w[3]) :: "memory");
}

// Transposed fragment of a tile: MFMA row m = lane & 31 is head dim m, contraction slot (t, i) is token 8 (2t + i/4) + 4 half + i%4.
// ds_read_b64_tr_b16 (measured, tools/tr_probe.hip): in each 16-lane group, output lane i element j = element i & 3 of the 8 bytes
// addressed by lane 4 j + (i >> 2).  Lane t16 of a group therefore points at token kbase + (t16 >> 2), dims 16 grp + 4 (t16 & 3) .. + 3,
// and receives tokens kbase .. kbase + 3 at dim 16 grp + t16.  4 token rows x 64 B per 32 lanes: every bank once.  (Rows 8 apart differ
// in the swizzle, rows 16 apart do not: two addresses, the second pair at +1024.)
__device__ __forceinline__ Frag tile_cols(const char* tile, int lane) {
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)tile;
  const int t16 = lane & 15, grp = (lane >> 4) & 1, half = lane >> 5;
  const int tok = 4 * half + (t16 >> 2), byte = 32 * grp + 8 * (t16 & 3);
  const uint32_t a0 = base + (uint32_t)tile_off(tok, byte), a1 = base + (uint32_t)tile_off(tok + 8, byte);
  u32x2 r00, r01, r10, r11;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r00) : "v"(a0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r01) : "v"(a1) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(r10) : "v"(a0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(r11) : "v"(a1) : "memory");
  // the reads are asynchronous and the compiler does not know it: the wait carries the registers
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r00), "+v"(r01), "+v"(r10), "+v"(r11) :: "memory");
  Frag f;
  f.v[0] = __builtin_bit_cast(bf16x8, u32x4{r00[0], r00[1], r01[0], r01[1]});
  f.v[1] = __builtin_bit_cast(bf16x8, u32x4{r10[0], r10[1], r11[0], r11[1]});
  return f;
}
__device__ __forceinline__ float pair_sum(float v) { return half_sum(v); }
__device__ __forceinline__ float pair_max(float v) { return half_max(v); }

struct ShortParams {
  const bf16_t* q; const bf16_t* kv; const float* q_scale; const float* k_scale;
  int64_t ldq, ldkv;
  int nseq, H, L;
  float scale;
  bf16_t* out; int64_t ldo;                         // forward
  const bf16_t* dout; int64_t lddo;                 // backward
  bf16_t* dq; bf16_t* dkv; int64_t lddq, lddkv;
  float* part;                                      // [workgroups][2][32] scale-gradient partials
};

// l2norm of the lane's half row + the learned scale: unit row xn, inverse norm, and xn * s * mult packed for the MFMA
__device__ __forceinline__ Row norm_row(const Row& raw, const float (&s)[16], float mult, float (&xn)[16], float& inv) {
  float x[16];
  unpack_row(raw, x);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) ss = fmaf(x[i], x[i], ss);
  ss = pair_sum(ss);
  inv = __builtin_amdgcn_rcpf(fmaxf(sqrtf(ss), 1e-12f));   // F.normalize eps; v_rcp_f32 is good to 1 ulp
  float y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { xn[i] = x[i] * inv; y[i] = xn[i] * s[i] * mult; }
  return pack_row(y);
}
// gradient through y = xn * s (xn = x / |x|): g = dL/dy -> dL/dx ; dscale[i] += g[i] * xn[i]
__device__ __forceinline__ Row norm_row_bwd(const float (&g)[16], const float (&xn)[16], float inv, const float (&s)[16], float (&dscale)[16]) {
  float gn[16], dot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dscale[i] = fmaf(g[i], xn[i], dscale[i]); gn[i] = g[i] * s[i]; dot = fmaf(gn[i], xn[i], dot); }
  dot = pair_sum(dot);
  float dx[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) dx[i] = inv * (gn[i] - xn[i] * dot);
  return pack_row(dx);
}

__device__ __forceinline__ void load_scales(const float* v, int half, float (&s)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(v + 8 * j + 4 * half);
    s[4 * j] = t[0]; s[4 * j + 1] = t[1]; s[4 * j + 2] = t[2]; s[4 * j + 3] = t[3];
  }
}

// Per problem (one wave): wait for its tiles, move them into registers, start the DMA of the wave's next problem, compute, store.
// Outstanding VMEM at the top of a problem, oldest first: the DMA of this problem, then the previous problem's stores -- vmcnt(#stores).
__global__ __launch_bounds__(WPB_F * 64) void attn_short_fwd_kernel(ShortParams p) {
  __shared__ __attribute__((aligned(16))) char lds[WPB_F][5 * 2048];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 31, half = lane >> 5;
  char* tq = lds[wave]; char* tk = tq + 2048; char* tvv = tq + 4096; char* stage = tq + 8192;     // tvv: two V tiles (the PV product reads V late)
  float sq[16], sk[16];
  load_scales(p.q_scale, half, sq); load_scales(p.k_scale, half, sk);
  const int HD = p.H * 32;
  const int64_t nitems = (int64_t)p.nseq * p.H, stride = (int64_t)gridDim.x * WPB_F;
  auto fetch = [&](int64_t it, int vb) {
    const int64_t t0 = (it / p.H) * p.L; const int hh = (int)(it % p.H);
    tile_fetch(tq, p.q + t0 * p.ldq + hh * 32, p.ldq, p.L, lane);
    tile_fetch(tk, p.kv + t0 * p.ldkv + hh * 32, p.ldkv, p.L, lane);
    tile_fetch(tvv + vb * 2048, p.kv + t0 * p.ldkv + HD + hh * 32, p.ldkv, p.L, lane);
  };
  int64_t it = (int64_t)blockIdx.x * WPB_F + wave;
  if (it < nitems) fetch(it, 0);
  int vb = 0;
  bool first = true;
  for (; it < nitems; it += stride, vb ^= 1) {
    const int64_t s = it / p.H; const int h = (int)(it % p.H);
    if (first) wait_vm<0>(); else wait_vm<2>();
    first = false;
    Row rq = tile_read(tq, row, half), rk = tile_read(tk, row, half);
    rows_landed(rq, rk);
    if (it + stride < nitems) fetch(it + stride, vb ^ 1);
    float xn[16], inv;
    const Frag Qt = frag_of(norm_row(rq, sq, p.scale * LOG2E, xn, inv));
    const Frag Ks = frag_of(norm_row(rk, sk, 1.f, xn, inv));
    // S^T: rows = keys 8 j + 4 half + r, column = the lane's query (log2 domain)
    const f32x16 sacc = mma(Ks, Qt);
    float pr[16], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = 8 * (i >> 2) + 4 * half + (i & 3);
      pr[i] = key < p.L ? sacc[i] : -INFINITY;
      m = fmaxf(m, pr[i]);
    }
    m = pair_max(m);
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { pr[i] = __builtin_amdgcn_exp2f(pr[i] - m); l += pr[i]; }
    l = pair_sum(l);
    const Frag P = frag_of(pack_row(pr));
    const Frag Vt = tile_cols(tvv + vb * 2048, lane);
    const f32x16 oacc = mma(Vt, P);                  // O^T: rows = dims of layout R, column = the lane's query
    const float il = __builtin_amdgcn_rcpf(l);
    float o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = oacc[i] * il;
    tile_write(stage, row, half, pack_row(o));
    tile_store(stage, p.out + s * p.L * p.ldo, p.ldo, h * 32, p.L, lane);
  }
}

// (the learned scales live in LDS and are re-read at each use instead of holding 32 registers)
__global__ __launch_bounds__(WPB_B * 64, 2) void attn_short_bwd_kernel(ShortParams p) {
  constexpr int PER_WAVE = 8 * 2048 + 512;          // raw q, k, v, dO | k^s, q~, dO for the transposing reads | staging | statistics
  // ONE shared object: with a second one the compiler waits vmcnt(0) before every LDS read that follows a DMA (it drains the prefetch)
  __shared__ __attribute__((aligned(16))) char lds[WPB_B * PER_WAVE + 256 + WPB_B * 256];
  float (*scales)[32] = reinterpret_cast<float (*)[32]>(lds + WPB_B * PER_WAVE);
  float (*fold)[64] = reinterpret_cast<float (*)[64]>(lds + WPB_B * PER_WAVE + 256);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 31, half = lane >> 5;
  char* rawq = lds + wave * PER_WAVE; char* rawk = rawq + 2048; char* rawv = rawq + 4096; char* rawdo = rawq + 6144;
  char* tk = rawq + 8192; char* tq = rawq + 10240; char* tdo = rawq + 12288; char* stage = rawq + 14336;
  float* st = reinterpret_cast<float*>(rawq + 16384);   // [3][32]: row max (log2), 1 / row sum, delta of each query
  if (threadIdx.x < 64) scales[threadIdx.x >> 5][threadIdx.x & 31] = (threadIdx.x < 32 ? p.q_scale : p.k_scale)[threadIdx.x & 31];
  __syncthreads();
  float dsq[16], dsk[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dsq[i] = 0.f; dsk[i] = 0.f; }
  const int HD = p.H * 32;
  const bool valid = row < p.L;
  const int64_t nitems = (int64_t)p.nseq * p.H, stride = (int64_t)gridDim.x * WPB_B;
  auto fetch = [&](int64_t it) {
    const int64_t t0 = (it / p.H) * p.L; const int hh = (int)(it % p.H);
    tile_fetch(rawq, p.q + t0 * p.ldq + hh * 32, p.ldq, p.L, lane);
    tile_fetch(rawk, p.kv + t0 * p.ldkv + hh * 32, p.ldkv, p.L, lane);
    tile_fetch(rawv, p.kv + t0 * p.ldkv + HD + hh * 32, p.ldkv, p.L, lane);
    tile_fetch(rawdo, p.dout + t0 * p.lddo + hh * 32, p.lddo, p.L, lane);
  };
  int64_t it = (int64_t)blockIdx.x * WPB_B + wave;
  if (it < nitems) fetch(it);
  bool first = true;
  for (; it < nitems; it += stride) {
    const int64_t s = it / p.H; const int h = (int)(it % p.H);
    if (first) wait_vm<0>(); else wait_vm<6>();
    first = false;
    Row rq = tile_read(rawq, row, half), rk = tile_read(rawk, row, half), rv = tile_read(rawv, row, half), rdo = tile_read(rawdo, row, half);
    rows_landed(rq, rk); rows_landed(rv, rdo);
    if (it + stride < nitems) fetch(it + stride);
    float qn[16], kn[16], invq, invk;
    Row rqt, rks;
    { float sc[16]; load_scales(scales[0], half, sc); rqt = norm_row(rq, sc, p.scale * LOG2E, qn, invq); }
    { float sc[16]; load_scales(scales[1], half, sc); rks = norm_row(rk, sc, 1.f, kn, invk); }
    const Frag Qt = frag_of(rqt), Ks = frag_of(rks), Vf = frag_of(rv), dOf = frag_of(rdo);
    tile_write(tk, row, half, rks); tile_write(tq, row, half, rqt); tile_write(tdo, row, half, rdo);

    // ---- queries as columns: P, dP, delta, dS -> dQ
    {
      const f32x16 sacc = mma(Ks, Qt);
      float pr[16], m = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 8 * (i >> 2) + 4 * half + (i & 3);
        pr[i] = key < p.L ? sacc[i] : -INFINITY;
        m = fmaxf(m, pr[i]);
      }
      m = pair_max(m);
      float l = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { pr[i] = __builtin_amdgcn_exp2f(pr[i] - m); l += pr[i]; }
      l = pair_sum(l);
      const float il = __builtin_amdgcn_rcpf(l);
      const f32x16 dp = mma(Vf, dOf);               // dP^T[key][query] = sum_d V[key][d] dO[query][d]
      float delta = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { pr[i] *= il; delta = fmaf(pr[i], dp[i], delta); }
      delta = pair_sum(delta);
      float dz[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) dz[i] = pr[i] * (dp[i] - delta);
      if (half == 0) { st[row] = m; st[32 + row] = il; st[64 + row] = delta; }
      const Frag KsT = tile_cols(tk, lane);
      const f32x16 gacc = mma(KsT, frag_of(pack_row(dz)));     // (dS K^s)^T: rows = dims, column = the lane's query
      const float gs = valid ? p.scale : 0.f;                  // (rows past the sequence hold a copy of its last row: no gradient)
      float g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = gacc[i] * gs;         // dL / d(q^ * q_scale)
      float sc[16];
      load_scales(scales[0], half, sc);
      tile_write(stage, row, half, norm_row_bwd(g, qn, invq, sc, dsq));
      tile_store(stage, p.dq + s * p.L * p.lddq, p.lddq, h * 32, p.L, lane);
    }
    // ---- keys as columns: P, dP, dS -> dK, dV
    {
      const f32x16 sacc = mma(Qt, Ks);              // S[query][key]: rows = queries 8 j + 4 half + r, column = the lane's key
      const f32x16 dp = mma(dOf, Vf);
      float p2[16], dz[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 mq = *reinterpret_cast<const f32x4*>(st + 8 * j + 4 * half);
        const f32x4 ilq = *reinterpret_cast<const f32x4*>(st + 32 + 8 * j + 4 * half);
        const f32x4 dq4 = *reinterpret_cast<const f32x4*>(st + 64 + 8 * j + 4 * half);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * j + r, query = 8 * j + 4 * half + r;
          const float pv = (valid && query < p.L) ? __builtin_amdgcn_exp2f(sacc[i] - mq[r]) * ilq[r] : 0.f;
          p2[i] = pv;
          dz[i] = pv * (dp[i] - dq4[r]);
        }
      }
      const Frag QtT = tile_cols(tq, lane), dOT = tile_cols(tdo, lane);
      const f32x16 gk = mma(QtT, frag_of(pack_row(dz)));        // (dS^T Q~)^T: Q~ carries scale * log2 e
      const f32x16 gv = mma(dOT, frag_of(pack_row(p2)));        // (P^T dO)^T
      float g[16], dv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { g[i] = gk[i] * (1.f / LOG2E); dv[i] = gv[i]; }
      float sc[16];
      load_scales(scales[1], half, sc);
      const Row rdk = norm_row_bwd(g, kn, invk, sc, dsk);
      // (the staging tile is free again once the dq stores have READ it: their data leaves LDS at issue)
      tile_write(stage, row, half, rdk);
      tile_store(stage, p.dkv + s * p.L * p.lddkv, p.lddkv, h * 32, p.L, lane);
      tile_write(stage, row, half, pack_row(dv));
      tile_store(stage, p.dkv + s * p.L * p.lddkv, p.lddkv, HD + h * 32, p.L, lane);
    }
  }
  // fold the scale gradients: the 32 token lanes of each half, then the waves of the workgroup in order -> one partial row per workgroup
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    { dsq[i] = half32_sum(dsq[i]); dsk[i] = half32_sum(dsk[i]); }
  }
  if (row == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = 8 * (i >> 2) + 4 * half + (i & 3);
      fold[wave][d] = dsq[i]; fold[wave][32 + d] = dsk[i];
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WPB_B; ++w) v += fold[w][threadIdx.x];
    p.part[(int64_t)blockIdx.x * 64 + threadIdx.x] = v;
  }
}

// dq_scale[d] += sum over workgroups of part[g][0][d]; dk_scale likewise.  1024 threads: 16 strided partial sums per column (four
// independent chains each), folded in a fixed order.  (The first version walked 4096 rows with one wave: 248 us.)
__global__ __launch_bounds__(1024) void attn_short_scale_sum_kernel(const float* __restrict__ part, int nrows, float* __restrict__ dqs, float* __restrict__ dks) {
  __shared__ float red[16][64];
  const int t = threadIdx.x & 63, g = threadIdx.x >> 6;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int w = g; w < nrows; w += 64) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (w + 16 * k < nrows) a[k] += part[(int64_t)(w + 16 * k) * 64 + t];
  }
  red[g][t] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (g == 0) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][t];
    if (t < 32) { if (dqs) dqs[t] += v; } else if (dks) dks[t - 32] += v;
  }
}

int short_grid(int nseq, int H, int wpb, int per_cu) {
  static int cus = [] { hipDeviceProp_t pr; int dev = 0; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256; }();
  const int64_t items = (int64_t)nseq * H, want = (items + wpb - 1) / wpb;
  const int64_t cap = (int64_t)cus * per_cu;
  return (int)(want < cap ? want : cap);
}
int fwd_grid(int nseq, int H) { return short_grid(nseq, H, WPB_F, 3); }     // 40 KB of LDS per workgroup: three per CU
int bwd_grid(int nseq, int H) { return short_grid(nseq, H, WPB_B, 3); }     // 51 KB per workgroup: three per CU

bool short_args_ok(const void* q, const void* kv, int64_t ldq, int64_t ldkv, int nseq, int H, int L) {
  return q && kv && nseq >= 1 && H >= 1 && L >= 1 && L <= 32 && ldq % 4 == 0 && ldkv % 4 == 0 && ldq >= H * 32 && ldkv >= 2 * H * 32 &&
         reinterpret_cast<uintptr_t>(q) % 8 == 0 && reinterpret_cast<uintptr_t>(kv) % 8 == 0;
}

}  // namespace

// 1 when ctclip_attn_short_* serves this shape (bf16, d_head 32, L <= 32, no bias / mask)
extern "C" int ctclip_attn_short_supported(int L, int D, int dtype) { return dtype == DT_BF16 && D == 32 && L >= 1 && L <= 32; }

// out[(s L + i), h*32 + :] = softmax_j(scale * <l2norm(q_i) q_scale, l2norm(k_j) k_scale>) v_j   (attention.py:145-178)
// q (nseq*L, ldq >= H*32), kv (nseq*L, ldkv >= 2*H*32) = [k | v], out (nseq*L, ldo) bf16; q_scale, k_scale (32) f32.
extern "C" int ctclip_attn_short_fwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const float* q_scale, const float* k_scale,
                                     void* out, int64_t ldo, int nseq, int H, int L, float scale, hipStream_t stream) {
  if (!short_args_ok(q, kv, ldq, ldkv, nseq, H, L) || !out || !q_scale || !k_scale || ldo % 4 || ldo < H * 32) { ctclip_set_error("attn_short_fwd: bad args (L <= 32, 8-byte aligned rows)"); return CTCLIP_EBADARG; }
  ShortParams p{};
  p.q = (const bf16_t*)q; p.kv = (const bf16_t*)kv; p.q_scale = q_scale; p.k_scale = k_scale; p.ldq = ldq; p.ldkv = ldkv;
  p.nseq = nseq; p.H = H; p.L = L; p.scale = scale; p.out = (bf16_t*)out; p.ldo = ldo;
  hipLaunchKernelGGL(attn_short_fwd_kernel, dim3((unsigned)fwd_grid(nseq, H)), dim3(WPB_F * 64), 0, stream, p);
  return ctclip_check_launch("attn_short_fwd");
}

extern "C" int64_t ctclip_attn_short_bwd_workspace(int nseq, int H) { return (int64_t)bwd_grid(nseq, H) * 64 * 4; }

// dq (nseq*L, lddq), dkv (nseq*L, lddkv) = [dk | dv] bf16 are overwritten; dq_scale, dk_scale (32) f32 are ACCUMULATED (+=) when non-null.
extern "C" int ctclip_attn_short_bwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const float* q_scale, const float* k_scale,
                                     const void* dout, int64_t lddo, void* dq, int64_t lddq, void* dkv, int64_t lddkv, float* dq_scale,
                                     float* dk_scale, int nseq, int H, int L, float scale, void* workspace, int64_t workspace_bytes,
                                     hipStream_t stream) {
  if (!short_args_ok(q, kv, ldq, ldkv, nseq, H, L) || !dout || !dq || !dkv || !q_scale || !k_scale || lddo % 4 || lddq % 4 || lddkv % 4 ||
      lddo < H * 32 || lddq < H * 32 || lddkv < 2 * H * 32) { ctclip_set_error("attn_short_bwd: bad args"); return CTCLIP_EBADARG; }
  if (!workspace || workspace_bytes < ctclip_attn_short_bwd_workspace(nseq, H)) { ctclip_set_error("attn_short_bwd: workspace too small"); return CTCLIP_EWORKSPACE; }
  ShortParams p{};
  p.q = (const bf16_t*)q; p.kv = (const bf16_t*)kv; p.q_scale = q_scale; p.k_scale = k_scale; p.ldq = ldq; p.ldkv = ldkv;
  p.nseq = nseq; p.H = H; p.L = L; p.scale = scale; p.dout = (const bf16_t*)dout; p.lddo = lddo;
  p.dq = (bf16_t*)dq; p.dkv = (bf16_t*)dkv; p.lddq = lddq; p.lddkv = lddkv; p.part = (float*)workspace;
  const int grid = bwd_grid(nseq, H);
  hipLaunchKernelGGL(attn_short_bwd_kernel, dim3((unsigned)grid), dim3(WPB_B * 64), 0, stream, p);
  if (dq_scale || dk_scale) hipLaunchKernelGGL(attn_short_scale_sum_kernel, dim3(1), dim3(1024), 0, stream, (const float*)workspace, grid, dq_scale, dk_scale);
  return ctclip_check_launch("attn_short_bwd");
}
