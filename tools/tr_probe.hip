// Probe of ds_read_b64_tr_b16 (gfx950): every lane supplies its own 8-byte address; which (lane, element) does each output come from?
// build: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4];
  const int l = threadIdx.x;
  for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (unsigned short)(l * 4 + e);   // lane l supplies address &lds[l*4]: value = lane*4 + elem
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 512); hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf("  (L%2d,e%d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4); printf("\n"); }
  return 0;
}
