// LDS read-modify-write rates on gfx950 (round 4: can the position-bias table gradient be scattered inside the attention backward?).
// One workgroup of 8 waves per CU; every wave issues ITERS x 16 DS instructions of one kind, a `s_waitcnt lgkmcnt(0)` every 16.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_rates.hip -o lds_atomic_rates
// Address patterns: 0 = lane-consecutive dwords (conflict-free), 1 = the offset-class pattern of a 32 x 32 score tile whose lanes are
// keys of a 24-wide token grid (stride-47 table), 2 = the same with the stride-56 table, 3 = every lane the same address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int TAB = 4096;      // dwords of the target table (two per 64-bit entry)

template <int OP>
__device__ __forceinline__ void one(uint32_t addr, uint32_t v, int off) {
  // off: compile-time immediate (element e of the 8-run)
  if (OP == 0) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"(addr), "v"(__uint_as_float(v)), "n"(0) : "memory");
  if (OP == 1) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(0) : "memory");
  if (OP == 2) { uint64_t d = ((uint64_t)v << 32) | v; asm volatile("ds_add_u64 %0, %1 offset:%2" :: "v"(addr), "v"(d), "n"(0) : "memory"); }
  if (OP == 3) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(0) : "memory");
  if (OP == 5) asm volatile("ds_max_u32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(0) : "memory");
  (void)off;
}

template <int OP, int PAT>
__global__ __launch_bounds__(512) void rmw(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t tab[2 * TAB];
  for (int i = threadIdx.x; i < 2 * TAB; i += 512) tab[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, half = lane >> 5;
  const int esz = OP == 2 ? 8 : 4;
  uint32_t base[2];
  for (int g = 0; g < 2; ++g) {
    int idx;
    if (PAT == 0) idx = lane + 64 * g;
    else if (PAT == 3) idx = 7;
    else {
      const int S = PAT == 1 ? 47 : 56;
      const int kj = wave * 32 + c, q0 = 16 * g + 8 * half + 96;          // key token of the lane, first query of the 8-run
      const int uk = (kj / 24) * S + kj % 24, uq = (q0 / 24) * S + q0 % 24;
      idx = uq - uk + 23 * S + 23;                                         // class of (q0, kj); the run ascends by one per query
    }
    base[g] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)tab + (uint32_t)idx * esz;
  }
  uint32_t v = lane * 3 + 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) one<OP>(base[g] + (PAT == 0 ? 128 * e * esz : e * esz), v, e);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    v += 1;
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 2 * TAB; i += 512) s += (float)tab[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// 16-bit reads into the high half of a register that keeps a zero low half: a bf16 table read as f32 without a shift
template <int OP>
__global__ __launch_bounds__(512) void rd(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t tab[2 * TAB];
  for (int i = threadIdx.x; i < 2 * TAB; i += 512) tab[i] = 0x3f803f80u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)tab + lane * (OP == 0 ? 2 : 4);
  uint32_t r[16];
  for (int i = 0; i < 16; ++i) r[i] = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (OP == 0) asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(r[e]) : "v"(a0), "n"(2 * 0) : "memory");
      if (OP == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[e]) : "v"(a0), "n"(4 * 0) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                 "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) :: "memory");
    acc += __uint_as_float(r[it & 15]);
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <class F>
void timeit(const char* name, F launch, int iters) {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 512);
  launch(out, 100);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  launch(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(4); hipMemcpy(h.data(), out, 16, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * 8;      // wave-instructions per CU
  printf("%-44s %8.1f us  %.2f ns per wave-instruction per CU (= %.1f clk at 2.0 GHz)   [check %.3g]\n", name, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n * 2.0, h[1]);
  hipFree(out);
}

#define RMW(OP, PAT, NAME) timeit(NAME, [](float* o, int it) { hipLaunchKernelGGL((rmw<OP, PAT>), dim3(256), dim3(512), 0, 0, o, it); }, 4000)
int main() {
  RMW(3, 0, "ds_write_b32  consecutive");
  RMW(0, 0, "ds_add_f32    consecutive");
  RMW(1, 0, "ds_add_u32    consecutive");
  RMW(2, 0, "ds_add_u64    consecutive");
  RMW(5, 0, "ds_max_u32    consecutive");
  RMW(0, 1, "ds_add_f32    class pattern S=47");
  RMW(1, 1, "ds_add_u32    class pattern S=47");
  RMW(2, 1, "ds_add_u64    class pattern S=47");
  RMW(0, 2, "ds_add_f32    class pattern S=56");
  RMW(1, 2, "ds_add_u32    class pattern S=56");
  RMW(2, 2, "ds_add_u64    class pattern S=56");
  RMW(0, 3, "ds_add_f32    one address");
  RMW(1, 3, "ds_add_u32    one address");
  timeit("ds_read_u16_d16_hi consecutive", [](float* o, int it) { hipLaunchKernelGGL((rd<0>), dim3(256), dim3(512), 0, 0, o, it); }, 4000);
  timeit("ds_read_b32 consecutive", [](float* o, int it) { hipLaunchKernelGGL((rd<1>), dim3(256), dim3(512), 0, 0, o, it); }, 4000);
  return 0;
}
