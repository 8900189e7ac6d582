// Issue rate of the vector instructions the streaming kernels lean on (gfx950): cycles per wave64 instruction, measured with
// s_memtime around long independent chains, at 1 and 3 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void chain(float* out, uint64_t* cycles, int iters) {
  float a[8];
  f32x2 p[8];
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; u[i] = threadIdx.x * 7 + i; }
  const float m = 1.0001f, c = 0.5f;
  const f32x2 pm = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) a[i] = __builtin_fmaf(a[i], m, c);
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
        if (OP == 2) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
        if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
        if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
        if (OP == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
        if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (OP == 13) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 14) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc)); if (i & 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i])); else asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i])); }
        if (OP == 15) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i].x) : "v"(m), "v"(c)); if (i & 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i])); else asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i])); }
        if (OP == 17) asm volatile("v_cvt_f32_bf16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
        if (OP == 18) asm volatile("v_cvt_f32_bf16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i]));
        if (OP == 19) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(0x07060302u));
        if (OP == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 21) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        if (OP == 22) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 23) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 16) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %4, %4, %2, %3" : "+v"(a[i]), "+v"(p[i].x), "+v"(p[i].y) : "v"(m), "v"(c));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  const int iters = 4000;
  for (int waves_per_simd : {1, 2, 3}) {
    const int threads = 256 * waves_per_simd;     // one workgroup per CU: waves spread over the four SIMDs
    const int blocks = 256;
    float* out; uint64_t* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(uint64_t) * blocks * threads / 64);
    hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    const double n = (double)iters * 64;         // instructions per wave
    // s_memtime counts at 100 MHz on gfx9 (constant clock): report wall-clock ns per instruction per SIMD instead
    printf("%-22s waves/SIMD %d: %.3f ns per loop slot per SIMD, %.3f counter ticks per slot per SIMD (kernel %.1f us; ticks per us %.0f)\n", name, waves_per_simd,
           ms * 1e6 / (n * waves_per_simd), avg / n / waves_per_simd, ms * 1e3, avg / (ms * 1e3));
    hipFree(out); hipFree(cyc);
  }
}

__global__ void cvt_check(float* o, const unsigned* x) {
  unsigned v = x[0]; float a, b;
  asm volatile("v_cvt_f32_bf16 %0, %1" : "=v"(a) : "v"(v));
  asm volatile("v_cvt_f32_bf16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(b) : "v"(v));
  o[0] = a; o[1] = b;
}

int main() {
  {
    unsigned hx = 0x40490000u | 0xbf80u;   // hi = 3.140625 (0x4049), lo = -1.0 (0xbf80)
    unsigned* dx; float* dout; float h[2];
    hipMalloc(&dx, 4); hipMalloc(&dout, 8); hipMemcpy(dx, &hx, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_check, dim3(1), dim3(64), 0, 0, dout, dx);
    hipMemcpy(h, dout, 8, hipMemcpyDeviceToHost);
    printf("v_cvt_f32_bf16 of 0x%08x: plain -> %g (expect -1), sdwa WORD_1 -> %g (expect 3.140625)\n", hx, h[0], h[1]);
  }
  run<0>("v_fma_f32 (C)"); run<5>("v_fma_f32 (asm)"); run<1>("v_pk_fma_f32"); run<6>("v_pk_mul_f32"); run<2>("v_lshlrev_b32"); run<4>("v_and_b32");
  run<7>("v_cvt_pk_bf16_f32"); run<9>("v_max_f32"); run<10>("v_mul_f32"); run<11>("v_add_f32"); run<12>("v_fmac_f32"); run<13>("v_lshlrev 2-reg");
  run<17>("v_cvt_f32_bf16"); run<18>("v_cvt_f32_bf16 sdwa hi"); run<19>("v_perm_b32"); run<20>("v_mov_b32"); run<21>("v_pk_add_f32"); run<22>("v_sub_f32"); run<23>("v_and_b32 2-reg");
  run<3>("v_exp_f32"); run<8>("v_rcp_f32"); run<14>("pk_fma + 1 unpack op"); run<15>("2 fma + 1 unpack op"); run<16>("exp + 2 fma");
  return 0;
}
