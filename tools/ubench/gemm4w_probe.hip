// TIMING-ONLY probe (round 6): the bf16 NT GEMM main loop as ONE workgroup of FOUR waves per CU -- one wave per SIMD, 512
// registers each, wave tile 128 x 128 = 4 x 4 fragments of v_mfma_f32_32x32x16_bf16 (256 accumulator registers), the same ring of five
// 32-KiB LDS panels, the same LDS-DMA loader and the same continuous panel stream across tiles as csrc/gemm_nt.hip (8 waves, two per SIMD,
// wave tile 64 x 128 of 16x16x32).  No epilogue (the accumulators are folded into one value per lane so that nothing is optimised away): the
// number this prints is the MAIN-LOOP rate, to be set against the main-loop share of gemm_nt_kernel (57-61 % matrix-pipe busy).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=0 gemm4w_probe.hip -o gemm4w_probe ; ./gemm4w_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int TM = 256, TN = 256, TK = 64, ROWB = 128, PANEL = 256 * ROWB, NPANEL = 5, NTH = 256, GL = 8;

#ifndef PROBE_NO_DMA
#define PROBE_NO_DMA 0       // 1: no loads after the prologue (matrix pipe + LDS reads only)
#endif
#ifndef PROBE_NO_READS
#define PROBE_NO_READS 0     // 1: no fragment reads after the prologue (matrix pipe + DMA only)
#endif

struct P { const uint16_t* A; const uint16_t* B; float* out; int64_t M, N, K, lda, ldb; int ntm, ntn; };

__device__ __forceinline__ const char* to_sgpr(const char* ptr) {
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__global__ __launch_bounds__(NTH) void gemm4w_probe_kernel(P p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int ntiles = p.ntm * p.ntn, nk = (int)(p.K / TK);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int G = gridDim.x;
  const int slotb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  auto tile_of = [&](int it, int64_t& m0, int64_t& n0) -> bool {
    const int id = it * G + slotb;
    if (id >= ntiles) return false;
    m0 = (int64_t)(id / p.ntn) * TM; n0 = (int64_t)(id % p.ntn) * TN;
    return true;
  };
  int64_t m0, n0;
  if (!tile_of(0, m0, n0)) return;

  // ---- loader cursors (as gemm_nt.hip: bases in SGPRs, piece offsets in VGPRs; the LDS swizzle is applied on the source side)
  const char* a_base = nullptr; const char* b_base = nullptr;
  int a_it = 0, b_it = 0, a_t = 0, b_t = 0;
  uint32_t a_off[GL], b_off[GL];
  auto enter = [&](int it, bool isA) {
    int64_t tm0, tn0;
    if (isA) a_t = 0; else b_t = 0;
    if (!tile_of(it, tm0, tn0)) return;
    if (isA) { a_it = it; a_base = to_sgpr(reinterpret_cast<const char*>(p.A) + tm0 * p.lda * 2); }
    else { b_it = it; b_base = to_sgpr(reinterpret_cast<const char*>(p.B) + tn0 * p.ldb * 2); }
#pragma unroll
    for (int j = 0; j < GL; ++j) {
      const int rho = (wave * GL + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
      const uint32_t off = (uint32_t)rho * (uint32_t)((isA ? p.lda : p.ldb) * 2) + (uint32_t)(chunk * 16);
      if (isA) a_off[j] = off; else b_off[j] = off;
    }
  };
  auto glds = [&](const char* sbase, uint32_t voff, int slot, int j) {
    const char* src = sbase + (uint64_t)voff;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + slot * PANEL + (wave * GL + j) * 1024), 16, 0, 0);
  };
  auto wrap = [](int s) { return s >= NPANEL ? s - NPANEL : s; };

  // ---- fragment addressing: row l31 of a 32-row block, 16-byte chunk 2 ks + half, chunk' = chunk ^ ((row >> 1) & 7)  =>  addr(ks) = base0 ^ (ks << 5)
  const uint32_t base0 = (uint32_t)(l31 * ROWB + ((half ^ ((l31 >> 1) & 7)) << 4));
  bf16x8 fa[2][4], fb[2][4];
  auto read_frags = [&](int buf, int slot_a, int slot_b, int ks) {
    const char* pa = lds + slot_a * PANEL + wm * 16384 + (base0 ^ (uint32_t)(ks << 5));
    const char* pb = lds + slot_b * PANEL + wn * 16384 + (base0 ^ (uint32_t)(ks << 5));
#pragma unroll
    for (int f = 0; f < 4; ++f) { fa[buf][f] = *reinterpret_cast<const bf16x8*>(pa + f * 4096); fb[buf][f] = *reinterpret_cast<const bf16x8*>(pb + f * 4096); }
  };

  // ---- prologue: A(0) B(0) A(1) B(1)
  enter(0, true); enter(0, false);
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(a_base, a_off[j], 0, j);
  if (++a_t == nk) enter(a_it + 1, true);
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(b_base, b_off[j], 1, j);
  if (++b_t == nk) enter(b_it + 1, false);
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(a_base + (int64_t)a_t * (TK * 2), a_off[j], 2, j);
  if (++a_t == nk) enter(a_it + 1, true);
#pragma unroll
  for (int j = 0; j < GL; ++j) glds(b_base + (int64_t)b_t * (TK * 2), b_off[j], 3, j);
  if (++b_t == nk) enter(b_it + 1, false);
  wait_vm<2 * GL>();
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, 1, 0);
  int cs = 0;
  float fold = 0.f;

  for (int it = 0;; ++it) {
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#define MM(BUF, a, b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[BUF][a], fb[BUF][b], acc[a][b], 0, 0, 0);
#define MM16(BUF) MM(BUF, 0, 0) MM(BUF, 0, 1) MM(BUF, 1, 0) MM(BUF, 1, 1) MM(BUF, 0, 2) MM(BUF, 0, 3) MM(BUF, 1, 2) MM(BUF, 1, 3) \
                  MM(BUF, 2, 0) MM(BUF, 2, 1) MM(BUF, 3, 0) MM(BUF, 3, 1) MM(BUF, 2, 2) MM(BUF, 2, 3) MM(BUF, 3, 2) MM(BUF, 3, 3)
    // per sub-step: 16 matrix instructions, under them the 8 fragment reads of the NEXT sub-step and 4 LDS-DMA pieces (A(g+2) in ks = 0, 1; B(g+2) in
    // ks = 2 .. wait: its slot is A(g)'s, free only behind barrier_g -> ks = 3 takes all eight)
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    for (int t = 0; t < nk; ++t) {
      const int slot_a2 = wrap(cs + 4), slot_b2 = cs, slot_na = wrap(cs + 2), slot_nb = wrap(cs + 3);
      const char* a_k = a_base + (int64_t)a_t * (TK * 2);
      // ks = 0
      if (!PROBE_NO_READS) read_frags(1, cs, wrap(cs + 1), 1);
      if (!PROBE_NO_DMA) {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds(a_k, a_off[j], slot_a2, j);
      }
      MM16(0)
#pragma unroll
      for (int i = 0; i < 8; ++i) { SGB(0x8, 2); SGB(0x100, 1); if (i & 1) SGB(0x20, 1); }
      __builtin_amdgcn_sched_barrier(0);
      // ks = 1
      if (!PROBE_NO_READS) read_frags(0, cs, wrap(cs + 1), 2);
      if (!PROBE_NO_DMA) {
#pragma unroll
        for (int j = 4; j < 8; ++j) glds(a_k, a_off[j], slot_a2, j);
      }
      MM16(1)
#pragma unroll
      for (int i = 0; i < 8; ++i) { SGB(0x8, 2); SGB(0x100, 1); if (i & 1) SGB(0x20, 1); }
      __builtin_amdgcn_sched_barrier(0);
      if (++a_t == nk) enter(a_it + 1, true);
      // ks = 2
      if (!PROBE_NO_READS) read_frags(1, cs, wrap(cs + 1), 3);
      MM16(0)
#pragma unroll
      for (int i = 0; i < 8; ++i) { SGB(0x8, 2); SGB(0x100, 1); }
      __builtin_amdgcn_sched_barrier(0);
      // barrier_g: A(g+1), B(g+1) have landed (all but this wave's 8 youngest pieces = A(g+2)); every wave holds all of step g in registers
      if (!PROBE_NO_DMA) wait_vm<GL>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ks = 3
      const char* b_k = b_base + (int64_t)b_t * (TK * 2);
      if (!PROBE_NO_READS) read_frags(0, slot_na, slot_nb, 0);
      if (!PROBE_NO_DMA) {
#pragma unroll
        for (int j = 0; j < 8; ++j) glds(b_k, b_off[j], slot_b2, j);
      }
      MM16(1)
#pragma unroll
      for (int i = 0; i < 8; ++i) { SGB(0x8, 2); SGB(0x100, 1); SGB(0x20, 1); }
      __builtin_amdgcn_sched_barrier(0);
      if (++b_t == nk) enter(b_it + 1, false);
      cs = wrap(cs + 2);
    }
    // no epilogue: fold the accumulators (keeps the matrix instructions alive)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) fold += acc[a][b][0] + acc[a][b][15];
    if (!tile_of(it + 1, m0, n0)) break;
  }
  wait_vm<0>();
  p.out[blockIdx.x * NTH + threadIdx.x] = fold;
}

int main(int argc, char** argv) {
  struct Shape { int64_t M, N, K; const char* what; };
  const Shape shapes[] = {{110592, 512, 2816, "dX of the FF in-projection (N 512, K 2816)"}, {110592, 2816, 512, "FF in-projection (N 2816, K 512; GEGLU epilogue in the product)"},
                          {110592, 512, 1408, "FF out-projection (N 512, K 1408)"}, {110592, 768, 512, "q | k | v projection (N 768, K 512)"}};
  hipFuncSetAttribute((const void*)gemm4w_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NPANEL * PANEL);
  for (const Shape& s : shapes) {
    uint16_t *A, *B; float* out;
    hipMalloc(&A, s.M * s.K * 2); hipMalloc(&B, s.N * s.K * 2); hipMalloc(&out, 256 * NTH * 4);
    std::vector<uint16_t> h(1 << 20);
    const bool zero = getenv("PROBE_ZERO") != nullptr;
    for (auto& v : h) v = zero ? (uint16_t)0 : (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));      // ~ +-(0.008 .. 0.016): random operands (the clock follows the data)
    for (int64_t o = 0; o < s.M * s.K; o += (int64_t)h.size()) hipMemcpy(A + o, h.data(), std::min<int64_t>(h.size(), s.M * s.K - o) * 2, hipMemcpyHostToDevice);
    for (int64_t o = 0; o < s.N * s.K; o += (int64_t)h.size()) hipMemcpy(B + o, h.data(), std::min<int64_t>(h.size(), s.N * s.K - o) * 2, hipMemcpyHostToDevice);
    P p{A, B, out, s.M, s.N, s.K, s.K, s.K, (int)(s.M / TM), (int)(s.N / TN)};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w_probe_kernel, dim3(256), dim3(NTH), NPANEL * PANEL, 0, p);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm4w_probe_kernel, dim3(256), dim3(NTH), NPANEL * PANEL, 0, p);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
    const int tiles = p.ntm * p.ntn;
    printf("%-70s %8.1f us  %7.1f TFLOP/s  (%d tiles = %.3f rounds of 256; main loop only, no epilogue%s%s)\n", s.what, us, tf, tiles, tiles / 256.0,
           PROBE_NO_DMA ? ", NO DMA" : "", PROBE_NO_READS ? ", NO fragment reads" : "");
    hipFree(A); hipFree(B); hipFree(out);
  }
  return 0;
}
