// How fast can ONE wave per SIMD issue?  (round 6: the four-wave x 512-register attention backward runs ~135 instructions per score tile in ~1100 cycles.)
// One workgroup per CU of 256 threads (1 wave per SIMD) or 512 threads (2 per SIMD); per wave a long unrolled stream of independent instructions;
// s_memtime around it and hipEvent time of the launch.   hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// OP: 0 v_fma x8 | 1 v_exp x8 | 2 v_cvt_pk x8 | 3 1 MFMA + 4 fma | 4 1 MFMA + 8 fma | 5 1 MFMA + 12 fma | 6 MFMA only (2 accumulators)
//     7 ds_read_b128 x4 + 4 fma | 8 ds_add_u32 x8 | 9 1 MFMA + 4 fma + 2 ds_read_b128 + 2 ds_add | 10 v_pk_mul x8 | 11 fma + exp alternating
template <int OP>
__global__ void stream(float* out, uint64_t* cycles, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3f80 + i); fb[i] = (short)(0x3f00 + threadIdx.x); }
  f32x16 acc0 = {}, acc1 = {}, acc2 = {};
  const float m = 1.0001f, c = 0.5f;
  uint32_t* ldsu = reinterpret_cast<uint32_t*>(lds);
  const int lane = threadIdx.x & 63;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == 0) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); }
      if (OP == 1) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); }
      if (OP == 2) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7])); }
      if (OP >= 3 && OP <= 6) {
        if (r & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        constexpr int N = OP == 3 ? 4 : (OP == 4 ? 8 : (OP == 5 ? 12 : 0));
        _Pragma("unroll") for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      }
      if (OP == 7) {
        f32x4 v[4];
        _Pragma("unroll") for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + i * 256 + r * 1024) & 4095));
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(v[i][0]));
      }
      if (OP == 8) { _Pragma("unroll") for (int i = 0; i < 8; ++i) (void)__hip_atomic_fetch_add(ldsu + ((lane + i * 64 + r * 512) & 4095), (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (OP == 9) {
        if (r & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        f32x4 v[2];
        _Pragma("unroll") for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + i * 256 + r * 1024) & 4095));
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        _Pragma("unroll") for (int i = 0; i < 2; ++i) (void)__hip_atomic_fetch_add(ldsu + 2048 + ((lane + i * 64 + r * 128) & 2047), (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        a[8] += v[0][0] + v[1][0];
      }
      if (OP >= 12 && OP <= 14) {      // dependent accumulation chains over 1 / 2 / 3 accumulators, 4 fma between matrix instructions
        if (OP == 12) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        if (OP == 13) { if (r & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0); else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0); }
        if (OP == 14) { if (r % 3 == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0); else if (r % 3 == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0); else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0); }
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      }
      if (OP == 10) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&a[2 * i])) : "v"(*reinterpret_cast<const double*>(&a[(2 * i + 2) & 15]))); }
      if (OP == 11) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 + i])); } }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + acc0[i] + acc1[i] + acc2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_r) {
  for (int nthreads : {256, 512}) {
    const int iters = 2000, nblk = 256;
    float* out; uint64_t* cyc;
    hipMalloc(&out, nblk * nthreads * 4); hipMalloc(&cyc, nblk * 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<OP>, dim3(nblk), dim3(nthreads), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream<OP>, dim3(nblk), dim3(nthreads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(nblk * nthreads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    const double n = (double)iters * 8 * instr_per_r;
    printf("%-44s %d wave(s)/SIMD: %6.2f memtime ticks / instr / wave, %6.2f ns / instr / wave (launch %.3f ms)\n", name, nthreads / 256, avg / n, ms * 1e6 / n, ms);
    hipFree(out); hipFree(cyc);
  }
}

int main() {
  run<0>("v_fma_f32 x8", 8); run<1>("v_exp_f32 x8", 8); run<2>("v_cvt_pk_bf16_f32 x8", 8); run<10>("v_pk_mul_f32 x8", 8); run<11>("fma + exp alternating x4", 8);
  run<6>("mfma 32x32x16 only", 1); run<3>("1 mfma + 4 fma", 5); run<4>("1 mfma + 8 fma", 9); run<5>("1 mfma + 12 fma", 13);
  run<12>("1 mfma (ONE accumulator chain) + 4 fma", 5); run<13>("1 mfma (two chains) + 4 fma", 5); run<14>("1 mfma (three chains, r % 3) + 4 fma", 5);
  run<7>("4 ds_read_b128 + 4 fma", 8); run<8>("ds_add_u32 x8", 8); run<9>("1 mfma + 4 fma + 2 ds_read_b128 + 2 ds_add + 1 add", 10);
  return 0;
}
