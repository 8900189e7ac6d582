// Optimiser step of CTClipTrainer.train_step (scripts/CTCLIPTrainer.py:259-263): global gradient-norm clip
// (torch.nn.utils.clip_grad_norm_, max_norm 0.5) followed by Adam(lr, betas=(0.9,0.99), eps=1e-8)
// (transformer_maskgit/optimizer.py:24) over ONE flat f32 parameter / gradient / moment buffer.
// HBM-bound: 16 B of state read + 12 B written per parameter, float4 accesses, no host synchronisation
// (the clip coefficient stays on the device).  The norm reduction is two-stage and deterministic.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float red[16];
  float s = 0.f;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)); `extra_sq` adds a
// precomputed sum of squares (e.g. other ranks' / other buffers' contribution), may be null.
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int nparts, const float* __restrict__ extra_sq,
                                                        float max_norm, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
  float t = block_sum(s, red);
  if (threadIdx.x == 0) {
    if (extra_sq) t += extra_sq[0];
    const float norm = sqrtf(t);
    out[0] = norm;
    out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  }
}
template <bool ZERO_G>      // ZERO_G: the gradient buffer is cleared in the same pass (optimizer.zero_grad() without its own 1.1-GB fill launch)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                                   float bc1, float bc2_sqrt, float weight_decay, const float* __restrict__ clip,
                                                   const uint8_t* __restrict__ decay_mask4, const unsigned long long* __restrict__ st) {
  if (st) {                                                  // the step count lives on the device (a replayed hipGraph): bias corrections from it
    const float t = (float)st[1];
    bc1 = 1.f - powf(beta1, t);
    bc2_sqrt = sqrtf(1.f - powf(beta2, t));
  }
  const float cs = clip ? clip[1] : 1.f;
  const float step = lr / bc1;
  const int64_t n4 = n / 4;
  // ADAM_U float4 groups per thread and trip, all 4 * ADAM_U 16-byte loads issued before the first use, non-temporal both ways: 28 bytes per
  // parameter stream through once (7.95 GB for CT-CLIP's 284 M parameters) and nothing of it is read again before the next step.
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
#ifndef ADAM_UNROLL
#define ADAM_UNROLL 4
#endif
  constexpr int ADAM_U = ADAM_UNROLL;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += ADAM_U * stride) {
    int64_t idx[ADAM_U];
    f32x4 pv[ADAM_U], mv[ADAM_U], vv[ADAM_U], gv[ADAM_U];
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      idx[u] = i0 + u * stride;
      const int64_t i = idx[u] < n4 ? idx[u] : i0;       // (clamped: never branch around a load)
      pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
      mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i);
      vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i);
      gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    }
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      if (idx[u] >= n4) break;
      const int64_t i = idx[u];
      const bool decay = weight_decay != 0.f && (!decay_mask4 || decay_mask4[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = gv[u][e] * cs;
        if (decay) pv[u][e] *= 1.f - lr * weight_decay;
        mv[u][e] = beta1 * mv[u][e] + (1.f - beta1) * gg;
        vv[u][e] = beta2 * vv[u][e] + (1.f - beta2) * gg * gg;
        pv[u][e] -= step * mv[u][e] / (sqrtf(vv[u][e]) / bc2_sqrt + eps);
      }
      __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p) + i);
      __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m) + i);
      __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v) + i);
#ifndef ADAM_ZERO_NT
#define ADAM_ZERO_NT 1      // 1 = the zeros bypass the caches like the three state streams; 0 = ordinary stores (A/B: profiles/r05_ab_experiments.md)
#endif
      if (ZERO_G) {
        if (ADAM_ZERO_NT) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(g) + i);
        else reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gg = g[i] * cs;
      float pv = p[i];
      if (weight_decay != 0.f && (!decay_mask4 || decay_mask4[i / 4])) pv *= 1.f - lr * weight_decay;
      m[i] = beta1 * m[i] + (1.f - beta1) * gg;
      v[i] = beta2 * v[i] + (1.f - beta2) * gg * gg;
      p[i] = pv - step * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
      if (ZERO_G) g[i] = 0.f;
    }
}

}  // namespace

extern "C" int64_t ctclip_grad_norm_workspace() { return 1024 * 4; }
// out[0] = ||g||_2 (with sqrt(extra_sq) folded in), out[1] = clip coefficient.  workspace >= ctclip_grad_norm_workspace().
extern "C" int ctclip_grad_norm_clip(const float* g, int64_t n, const float* extra_sq, float max_norm, float* out, void* workspace,
                                     int64_t workspace_bytes, hipStream_t s) {
  if (!g || !out || !workspace || workspace_bytes < ctclip_grad_norm_workspace() || ((uintptr_t)g % 16)) { ctclip_set_error("grad_norm_clip: bad args"); return CTCLIP_EBADARG; }
  int64_t nb = cdiv(n / 4 + 1, 256); if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, g, n, (float*)workspace);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, (int)nb, extra_sq, max_norm, out);
  return ctclip_check_launch("grad_norm_clip");
}
// torch.optim.Adam / AdamW step over flat buffers; `clip` = the 2-float output of ctclip_grad_norm_clip (or null).
// decay_mask4: one byte per group of FOUR consecutive parameters (ceil(n / 4) bytes), 0 = no weight decay for that group -- the
// "ndim < 2 parameters are not decayed" grouping of transformer_maskgit/optimizer.py:3-8,27-32; null = decay everything.
static int adam_launch(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step, float weight_decay,
                       const float* clip, const uint8_t* decay_mask4, bool zero_grad, hipStream_t s) {
  if (!p || !g || !m || !v || step < 1 || ((uintptr_t)p % 16) || ((uintptr_t)g % 16) || ((uintptr_t)m % 16) || ((uintptr_t)v % 16)) { ctclip_set_error("adam_step: bad args (16-B aligned flat buffers, step >= 1)"); return CTCLIP_EBADARG; }
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  int64_t nb = cdiv(n / 4 + 1, 256); if (nb > 8192) nb = 8192;
  if (zero_grad) hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2s, weight_decay, clip, decay_mask4, ctclip_step_state());
  else hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2s, weight_decay, clip, decay_mask4, ctclip_step_state());
  return ctclip_check_launch("adam_step");
}
extern "C" int ctclip_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                int step, float weight_decay, const float* clip, const uint8_t* decay_mask4, hipStream_t s) {
  return adam_launch(p, const_cast<float*>(g), m, v, n, lr, beta1, beta2, eps, step, weight_decay, clip, decay_mask4, false, s);
}
// The same step followed by optimizer.zero_grad() (scripts/CTCLIPTrainer.py:259-264: optim.step(); optim.zero_grad()) in ONE pass: every gradient is
// overwritten with zero right after it has been read -- 4 more bytes written per parameter instead of a separate 1.1-GB fill launch.
extern "C" int ctclip_adam_step_zero_grad(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                          int step, float weight_decay, const float* clip, const uint8_t* decay_mask4, hipStream_t s) {
  return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, step, weight_decay, clip, decay_mask4, true, s);
}
