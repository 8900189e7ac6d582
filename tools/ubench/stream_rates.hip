// Streaming-kernel rates at the image tower's token-grid size (110 592 rows x 512 bf16 columns = 113 MB per tensor): what a read + write pass
// can reach on this chip, next to the LayerNorm forward / backward access patterns (one wave64 per row, 16 B per lane) in a few variants.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_rates tools/ubench/stream_rates.hip && tools/ubench/stream_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { const hw_f32x2 v = {lo, hi}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2)); }
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(a[i] << 16); v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u); }
}
template <bool NT> __device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  if (NT) __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = a;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- plain copy: U 16-byte vectors per thread in flight, grid-stride
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = i0 + u * stride; v[u] = src[i < n ? i : n - 1]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = i0 + u * stride; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
  }
}
// read-only pass (sum into a sink nobody reads)
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  uint32_t acc = 0;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = i0 + u * stride; v[u] = src[i < n ? i : n - 1]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}

// ---- LayerNorm forward, 512 columns: RU rows per wave in flight.  CONTIG: each wave owns a contiguous run of rows instead of a grid stride.
template <int RU, bool NT, bool CONTIG>
__global__ __launch_bounds__(256) void lnf_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows) {
  constexpr int cols = 512;
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = gamma[lane * 8 + e]; bt[e] = beta[lane * 8 + e]; }
  const int64_t per = (rows + nwaves - 1) / nwaves;
  const int64_t beg = CONTIG ? wid * per : wid * RU, end = CONTIG ? (beg + per < rows ? beg + per : rows) : rows, step = CONTIG ? RU : nwaves * RU;
  for (int64_t row0 = beg; row0 < end; row0 += step) {
    float v[RU][8];
#pragma unroll
    for (int u = 0; u < RU; ++u) { const int64_t row = row0 + u < rows ? row0 + u : rows - 1; load8(x + row * cols + lane * 8, v[u]); }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[u][e];
      const float mean = wave_sum(s) / cols;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[u][e] - mean; q += d * d; }
      const float rstd = rsqrtf(wave_sum(q) / cols + 1e-5f);
      if (row < end) {
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[u][e] - mean) * rstd * gm[e] + bt[e];
        store8<NT>(y + row * cols + lane * 8, o);
      }
    }
  }
}

// ---- LayerNorm backward, 512 columns, ADDS extra same-shape gradients summed into dx.  HOIST: the add operands are requested together with
// dy / x (the library's kernel requests them after the two wave reductions).  RU rows per wave in flight.
template <int ADDS, bool HOIST, int RU, bool NT>
__global__ __launch_bounds__(256) void lnb_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                  float* __restrict__ part, int64_t rows, const bf16_t* __restrict__ add1, const bf16_t* __restrict__ add2) {
  constexpr int cols = 512;
  __shared__ float sm[4 * 2 * cols];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gm[8], dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = gamma[lane * 8 + e]; dg[e] = 0.f; db[e] = 0.f; }
  for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RU; row0 < rows; row0 += (int64_t)gridDim.x * 4 * RU) {
    float a[RU][8], b[RU][8], r1[RU][8], r2[RU][8], mu[RU], rs[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u < rows ? row0 + u : rows - 1;
      mu[u] = mean[row]; rs[u] = rstd[row];
      load8(dy + row * cols + lane * 8, a[u]); load8(x + row * cols + lane * 8, b[u]);
      if (HOIST && ADDS >= 1) load8(add1 + row * cols + lane * 8, r1[u]);
      if (HOIST && ADDS >= 2) load8(add2 + row * cols + lane * 8, r2[u]);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t row = row0 + u < rows ? row0 + u : rows - 1;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (b[u][e] - mu[u]) * rs[u];
        const float dyv = row0 + u < rows ? a[u][e] : 0.f;
        const float g = dyv * gm[e];
        b[u][e] = xh; a[u][e] = g;
        s1 += g; s2 += g * xh;
        dg[e] += dyv * xh; db[e] += dyv;
      }
      s1 = wave_sum(s1) / cols;
      s2 = wave_sum(s2) / cols;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rs[u] * (a[u][e] - s1 - b[u][e] * s2);
      if (!HOIST && ADDS >= 1) load8(add1 + row * cols + lane * 8, r1[u]);
      if (!HOIST && ADDS >= 2) load8(add2 + row * cols + lane * 8, r2[u]);
      if (ADDS >= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += r1[u][e];
      }
      if (ADDS >= 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += r2[u][e];
      }
      if (row0 + u < rows) store8<NT>(dx + row * cols + lane * 8, o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { sm[(wave * 2 + 0) * cols + lane * 8 + e] = dg[e]; sm[(wave * 2 + 1) * cols + lane * 8 + e] = db[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * cols; c += 256) {
    const int which = c / cols, col = c % cols;
    float t = 0.f;
    for (int w = 0; w < 4; ++w) t += sm[(w * 2 + which) * cols + col];
    part[((int64_t)blockIdx.x * 2 + which) * cols + col] = t;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <typename F> static float time_us(F&& launch, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1000.f / reps;
}

int main() {
  const int64_t rows = 110592; const int cols = 512;
  const int64_t n = rows * cols, bytes = n * 2, nvec = bytes / 16;
  // NSET rotating buffer sets (5 x 113 MB each): a launch never finds its operands in the 256-MB MALL left there by the previous launch --
  // inside the training step every one of these tensors is produced or consumed once, gigabytes apart.
  constexpr int NSET = 4;
  bf16_t *xs[NSET], *dys[NSET], *a1s[NSET], *a2s[NSET], *outs[NSET]; float *gamma, *beta, *mean, *rstd, *part; uint32_t* sink;
  for (int k = 0; k < NSET; ++k) { CK(hipMalloc(&xs[k], bytes)); CK(hipMalloc(&dys[k], bytes)); CK(hipMalloc(&a1s[k], bytes)); CK(hipMalloc(&a2s[k], bytes)); CK(hipMalloc(&outs[k], bytes)); }
  int turn = 0;
  bf16_t *x = xs[0], *dy = dys[0], *a1 = a1s[0], *a2 = a2s[0], *out = outs[0];
  auto rot = [&] { turn = (turn + 1) % NSET; x = xs[turn]; dy = dys[turn]; a1 = a1s[turn]; a2 = a2s[turn]; out = outs[turn]; };
  CK(hipMalloc(&gamma, cols * 4)); CK(hipMalloc(&beta, cols * 4)); CK(hipMalloc(&mean, rows * 4)); CK(hipMalloc(&rstd, rows * 4));
  CK(hipMalloc(&part, (size_t)8192 * 2 * cols * 4)); CK(hipMalloc(&sink, 64));
  std::vector<bf16_t> h(n); uint32_t s = 12345;
  for (int64_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (bf16_t)(0x3f00 + ((s >> 16) & 0xff) + ((s >> 31) << 15)); }
  for (int k = 0; k < NSET; ++k) {
    CK(hipMemcpy(xs[k], h.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(dys[k], h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(a1s[k], h.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(a2s[k], h.data(), bytes, hipMemcpyHostToDevice));
  }
  std::vector<float> ones(rows, 1.f);
  CK(hipMemcpy(gamma, ones.data(), cols * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, ones.data(), cols * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(mean, ones.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rstd, ones.data(), rows * 4, hipMemcpyHostToDevice));
  const double MB = bytes / 1e6;
  auto rep = [&](const char* name, double mb, float us) { printf("%-58s %8.1f us  %7.0f MB  %5.2f TB/s\n", name, us, mb, mb / us); fflush(stdout); };

  const int grids[] = {1024, 2048, 4096, 8192, 16384};
  for (int g : grids) {
    char nm[96];
    snprintf(nm, sizeof nm, "copy U=1 grid %d", g); rep(nm, 2 * MB, time_us([&] { rot(); hipLaunchKernelGGL((copy_kernel<1, false>), dim3(g), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)out, nvec); }));
    snprintf(nm, sizeof nm, "copy U=4 grid %d", g); rep(nm, 2 * MB, time_us([&] { rot(); hipLaunchKernelGGL((copy_kernel<4, false>), dim3(g), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)out, nvec); }));
    snprintf(nm, sizeof nm, "copy U=4 nt-store grid %d", g); rep(nm, 2 * MB, time_us([&] { rot(); hipLaunchKernelGGL((copy_kernel<4, true>), dim3(g), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)out, nvec); }));
  }
  rep("copy U=1 one vector per thread (grid = n/256)", 2 * MB, time_us([&] { rot(); hipLaunchKernelGGL((copy_kernel<1, false>), dim3((unsigned)(nvec / 256)), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)out, nvec); }));
  rep("read  U=4 grid 4096", MB, time_us([&] { rot(); hipLaunchKernelGGL((read_kernel<4>), dim3(4096), dim3(256), 0, 0, (const u32x4*)x, sink, nvec); }));
  rep("read  U=8 grid 2048", MB, time_us([&] { rot(); hipLaunchKernelGGL((read_kernel<8>), dim3(2048), dim3(256), 0, 0, (const u32x4*)x, sink, nvec); }));
  rep("write grid 4096", MB, time_us([&] { rot(); hipLaunchKernelGGL((write_kernel<false>), dim3(4096), dim3(256), 0, 0, (u32x4*)out, nvec); }));
  rep("write nt grid 4096", MB, time_us([&] { rot(); hipLaunchKernelGGL((write_kernel<true>), dim3(4096), dim3(256), 0, 0, (u32x4*)out, nvec); }));

  const double fmb = 2 * MB + rows * 8 / 1e6;
#define LNF(RU, NT, CG, G, label) rep(label, fmb, time_us([&] { rot(); hipLaunchKernelGGL((lnf_kernel<RU, NT, CG>), dim3(G), dim3(256), 0, 0, x, gamma, beta, out, mean, rstd, rows); }))
  LNF(4, false, false, 4096, "ln fwd RU=4 grid 4096 (library)");
  LNF(4, false, false, 6912, "ln fwd RU=4 grid 6912 (one sweep)");
  LNF(4, false, false, 2304, "ln fwd RU=4 grid 2304 (3 sweeps)");
  LNF(4, true, false, 4096, "ln fwd RU=4 grid 4096 nt-store");
  LNF(4, true, false, 6912, "ln fwd RU=4 grid 6912 nt-store");
  LNF(1, false, false, 27648, "ln fwd RU=1 grid 27648 (one row per wave)");
  LNF(2, false, false, 13824, "ln fwd RU=2 grid 13824 (one sweep)");
  LNF(4, false, true, 2048, "ln fwd RU=4 contiguous runs grid 2048");
  LNF(4, false, true, 4096, "ln fwd RU=4 contiguous runs grid 4096");
  LNF(8, false, false, 3456, "ln fwd RU=8 grid 3456 (one sweep)");
  CK(hipMemcpy(mean, ones.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rstd, ones.data(), rows * 4, hipMemcpyHostToDevice));

#define LNB(AD, HO, RU, NT, G, label) rep(label, (3 + AD) * MB + rows * 8 / 1e6, time_us([&] { rot(); hipLaunchKernelGGL((lnb_kernel<AD, HO, RU, NT>), dim3(G), dim3(256), 0, 0, dy, x, gamma, mean, rstd, out, part, rows, a1, a2); }))
  LNB(0, false, 1, false, 1024, "ln bwd adds=0 RU=1 grid 1024 (library)");
  LNB(1, false, 1, false, 1024, "ln bwd adds=1 RU=1 grid 1024 (library)");
  LNB(2, false, 1, false, 1024, "ln bwd adds=2 RU=1 grid 1024 (library)");
  LNB(1, true, 1, false, 1024, "ln bwd adds=1 hoisted RU=1 grid 1024");
  LNB(2, true, 1, false, 1024, "ln bwd adds=2 hoisted RU=1 grid 1024");
  LNB(0, true, 2, false, 1024, "ln bwd adds=0 RU=2 grid 1024");
  LNB(1, true, 2, false, 1024, "ln bwd adds=1 hoisted RU=2 grid 1024");
  LNB(2, true, 2, false, 1024, "ln bwd adds=2 hoisted RU=2 grid 1024");
  LNB(2, true, 2, true, 1024, "ln bwd adds=2 hoisted RU=2 nt-store grid 1024");
  LNB(2, true, 1, false, 2048, "ln bwd adds=2 hoisted RU=1 grid 2048");
  LNB(2, true, 2, false, 2048, "ln bwd adds=2 hoisted RU=2 grid 2048");
  LNB(2, true, 1, false, 512, "ln bwd adds=2 hoisted RU=1 grid 512");
  LNB(0, true, 4, false, 1024, "ln bwd adds=0 RU=4 grid 1024");
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
