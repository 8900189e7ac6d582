// Throughput of ds_read_b64_tr_b16 under candidate LDS layouts of a k-major [64 k][256 col] bf16 panel (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 tools/tr_bench.hip -o tools/tr_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) s16x4*)(p))

// MODE 0: rows of 128 B per quarter, 32-B sub-chunk XOR (k >> 1) & 3   (gemm_tn v1)
// MODE 1: [8 k][16 col] subtiles of 256 B (row stride 32 B), halves swapped in odd row groups
// MODE 2: same as 1 without the swap
// MODE 3: ds_read_b128 on a 128-B-row image with the 16-B chunk XOR of gemm_nt (reference)
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 15, lg = lane >> 4;
  for (int i = threadIdx.x; i < 16384; i += 512) ((unsigned*)lds)[i] = i;
  __syncthreads();
  unsigned x = 0;
  int addr[12][2];
  for (int f = 0; f < 12; ++f) {                       // 4 A fragments (quarter wm) + 8 B fragments (quarters wn*2 + (b>>2))
    const int q = f < 4 ? wm : (wn * 2 + ((f - 4) >> 2)), c = f < 4 ? f : ((f - 4) & 3);
    for (int j = 0; j < 2; ++j) {
      const int kk = j * 16 + lg * 4 + (li >> 2);
      if (MODE == 0) addr[f][j] = q * 8192 + kk * 128 + ((c ^ ((kk >> 1) & 3)) << 5) + (li & 3) * 8;
      else if (MODE == 1 || MODE == 2) {
        const int rg = kk >> 3, k8 = (kk & 7) ^ (MODE == 1 ? (rg & 1) * 4 : 0);
        addr[f][j] = (q * 8 + rg) * 1024 + c * 256 + k8 * 32 + (li & 3) * 8;
      } else {
        const int row = (f < 4 ? wm * 64 + f * 16 : wn * 128 + (f - 4) * 16) + li;
        addr[f][j] = (f < 4 ? 0 : 32768) + row * 128 + (((j * 4 + lg) ^ ((row >> 1) & 7)) << 4);
      }
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 12; ++f) {
      if (MODE == 3) {
        u32x4 v = *reinterpret_cast<const u32x4*>(lds + addr[f][0]);
        x ^= v[0] ^ v[3];
      } else {
        s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + addr[f][0]));
        s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + addr[f][1]));
        x ^= (unsigned)a[0] ^ (unsigned)b[3];
      }
    }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 512 + threadIdx.x] = x;
}
template <int MODE> void run(const char* name, unsigned* out) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, out, iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 8.0 * 12 * 1024 * iters;   // per CU: 8 waves x 12 fragments x 1 KiB
  printf("%-44s %7.3f ms   %.0f B/clk/CU @2.4GHz   (%.1f clk per fragment)\n", name, ms, bytes / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (8.0 * 12 * iters));
}
int main() {
  unsigned* out; (void)hipMalloc(&out, 256 * 512 * 4);
  run<3>("ds_read_b128, 128-B rows, chunk XOR (gemm_nt)", out);
  run<0>("tr_b16 x2, 128-B rows, 32-B XOR (gemm_tn v1)", out);
  run<1>("tr_b16 x2, [8k][16c] subtiles, halves swapped", out);
  run<2>("tr_b16 x2, [8k][16c] subtiles", out);
  return 0;
}
