// Heads of the two fine-tuning loops that reuse the CT-CLIP towers (BASELINE.json configs[3] and [4]):
//   * ClassFine / CT-LiPro (scripts/ct_lipro_train.py:17-38,79-107): image latents -> ReLU -> Dropout(0.3) -> Linear(512, 18),
//     BCEWithLogitsLoss(pos_weight);
//   * VocabFine (scripts/ct_vocabfine_train.py:96-123): softmax over the (present, absent) prompt pair of each pathology, MSE
//     against (1, 0).
// All tensors here are tiny (batch x 512, batch x 18, pairs x 2): one workgroup each, fused forward + backward, deterministic.
#include "common.h"

namespace {

// y = relu(x) * dropout_mask ; the mask is philox(seed, element group, stream) as in ctclip_dropout
__global__ void relu_dropout_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n4, float p,
                                    float inv_keep, uint64_t seed, uint32_t stream, const unsigned long long* __restrict__ st) {
  if (st) seed += st[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 w = philox4x32(seed, (uint64_t)i, stream);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float keep = p > 0.f ? dropout_mult(w[e], p, inv_keep) : 1.f;
      // forward: relu(x) * keep ; backward (dy given): dy * keep * [x > 0]
      r[e] = dy ? (xv[e] > 0.f ? reinterpret_cast<const f32x4*>(dy)[i][e] * keep : 0.f) : fmaxf(xv[e], 0.f) * keep;
    }
    reinterpret_cast<f32x4*>(out)[i] = r;
  }
}

// BCEWithLogitsLoss(pos_weight), mean over B x C: l = pw y softplus(-x) + (1 - y) softplus(x) ; dl/dx = ((pw - 1) y + 1) sigmoid(x) - pw y
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ logits, const float* __restrict__ targets,
                                                         const float* __restrict__ pos_weight, float* __restrict__ loss, float* __restrict__ dlogits,
                                                         int B, int C) {
  __shared__ float red[16];
  const int n = B * C;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = logits[i], y = targets[i], pw = pos_weight ? pos_weight[i % C] : 1.f;
    const float sp_neg = fmaxf(-x, 0.f) + log1pf(__expf(-fabsf(x)));      // softplus(-x), the stable form torch uses
    s += (1.f - y) * x + (1.f + (pw - 1.f) * y) * sp_neg;
    if (dlogits) {
      const float sig = 1.f / (1.f + __expf(-x));
      dlogits[i] = (((pw - 1.f) * y + 1.f) * sig - pw * y) / (float)n;
    }
  }
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = t / (float)n;
}

// VocabFine objective of one prompt group: sims (n, 2) -> p = softmax over the pair -> mean((p0 - 1)^2 + p1^2) over the 2n values
__global__ __launch_bounds__(64) void pair_softmax_mse_kernel(const float* __restrict__ sims, float* __restrict__ loss, float* __restrict__ dsims, int n) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) {
    const float a = sims[2 * i], b = sims[2 * i + 1];
    const float p0 = 1.f / (1.f + __expf(b - a)), p1 = 1.f - p0;
    s += (p0 - 1.f) * (p0 - 1.f) + p1 * p1;
    if (dsims) {
      // d/dp0 = 2 (p0 - 1), d/dp1 = 2 p1 ; dp0/da = p0 p1 = -dp0/db ; dp1/da = -p0 p1
      const float g = (2.f * (p0 - 1.f) - 2.f * p1) * p0 * p1 / (2.f * n);
      dsims[2 * i] = g; dsims[2 * i + 1] = -g;
    }
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s / (2.f * n);
}

// sims[p] = <t / |t|, v / |v|> * exp(temperature) for pair p of (text row, image row), either side broadcast when it has one row.
// One workgroup walks the pairs in order, each thread owning fixed columns, so the broadcast side's gradient accumulates in a fixed order.
__global__ __launch_bounds__(256) void latent_similarity_kernel(const float* __restrict__ text, const float* __restrict__ image,
                                                                const float* __restrict__ temperature, const float* __restrict__ dsims,
                                                                float* __restrict__ sims, float* __restrict__ dtext, float* __restrict__ dimage,
                                                                float* __restrict__ dtemp, int nt, int ni, int D, float eps) {
  __shared__ float red[16];
  const int n = nt > ni ? nt : ni;
  const float et = __expf(temperature[0]);
  if (dsims) {
    for (int i = threadIdx.x; i < nt * D; i += 256) dtext[i] = 0.f;
    for (int i = threadIdx.x; i < ni * D; i += 256) dimage[i] = 0.f;
  }
  float dtacc = 0.f;
  for (int p = 0; p < n; ++p) {
    const float* t = text + (int64_t)(nt == 1 ? 0 : p) * D;
    const float* v = image + (int64_t)(ni == 1 ? 0 : p) * D;
    float tt = 0.f, vv = 0.f, tv = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) { const float a = t[c], b = v[c]; tt += a * a; vv += b * b; tv += a * b; }
    tt = block_sum(tt, red); vv = block_sum(vv, red); tv = block_sum(tv, red);
    const float it = 1.f / fmaxf(sqrtf(tt), eps), iv = 1.f / fmaxf(sqrtf(vv), eps);
    const float cosv = tv * it * iv;
    if (!dsims) { if (threadIdx.x == 0) sims[p] = cosv * et; continue; }
    const float g = dsims[p] * et;
    dtacc += dsims[p] * cosv * et;
    float* dt = dtext + (int64_t)(nt == 1 ? 0 : p) * D;
    float* dv = dimage + (int64_t)(ni == 1 ? 0 : p) * D;
    for (int c = threadIdx.x; c < D; c += 256) {
      const float a = t[c] * it, b = v[c] * iv;           // unit rows
      dt[c] += g * (b - cosv * a) * it;
      dv[c] += g * (a - cosv * b) * iv;
    }
  }
  if (dsims && threadIdx.x == 0) dtemp[0] = dtacc;
}

}  // namespace

// image_latents = relu(latents) then nn.Dropout(p) in train mode (ct_lipro_train.py:33-36).  dy == null: forward (out = y);
// dy != null: backward (out = dx).  n % 4 == 0; the mask is a function of (seed, stream_id, element).
extern "C" int ctclip_relu_dropout(const float* x, const float* dy, float* out, int64_t n, float p, uint64_t seed, uint32_t stream_id, hipStream_t s) {
  if (!x || !out || n % 4 || p < 0.f || p >= 1.f) { ctclip_set_error("relu_dropout: n % 4 == 0, 0 <= p < 1"); return CTCLIP_EBADARG; }
  int64_t nb = cdiv(n / 4, 256); if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(relu_dropout_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, dy, out, n / 4, p, 1.f / (1.f - p), seed, stream_id, ctclip_step_state());
  return ctclip_check_launch("relu_dropout");
}

// torch.nn.BCEWithLogitsLoss(pos_weight=w) (ct_lipro_train.py:84), mean reduction, forward + gradient w.r.t. the logits.
// logits, targets: (B, C) f32; pos_weight: (C) f32 or null; loss: (1); dlogits: (B, C) or null.
extern "C" int ctclip_bce_logits(const float* logits, const float* targets, const float* pos_weight, float* loss, float* dlogits, int B, int C,
                                 hipStream_t s) {
  if (!logits || !targets || !loss || B < 1 || C < 1) { ctclip_set_error("bce_logits: bad args"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, s, logits, targets, pos_weight, loss, dlogits, B, C);
  return ctclip_check_launch("bce_logits");
}

// F.softmax(output, dim=0) of each (present, absent) similarity pair + MSELoss against (1, 0) over the group
// (ct_vocabfine_train.py:112-121).  sims: (n, 2) f32; loss: (1); dsims: (n, 2) or null.
extern "C" int ctclip_pair_softmax_mse(const float* sims, float* loss, float* dsims, int n, hipStream_t s) {
  if (!sims || !loss || n < 1) { ctclip_set_error("pair_softmax_mse: bad args"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(pair_softmax_mse_kernel, dim3(1), dim3(64), 0, s, sims, loss, dsims, n);
  return ctclip_check_launch("pair_softmax_mse");
}

// CTCLIP.forward without return_loss (ct_clip.py:771,796,805-807): l2norm both latents, einsum('b d, b d -> b') with broadcasting
// (two prompts against one volume in the zero-shot and VocabFine loops), times exp(temperature).
// text (nt, D), image (ni, D) f32 pre-normalisation latents; nt == ni or one of them 1.  dsims == null: forward, sims (max(nt, ni)).
// dsims != null: backward, dtext (nt, D), dimage (ni, D), dtemp (1) are overwritten.
extern "C" int ctclip_latent_similarity(const float* text, const float* image, const float* temperature, const float* dsims, float* sims,
                                        float* dtext, float* dimage, float* dtemp, int nt, int ni, int D, hipStream_t s) {
  if (!text || !image || !temperature || nt < 1 || ni < 1 || D < 1 || (nt != ni && nt != 1 && ni != 1)) { ctclip_set_error("latent_similarity: nt == ni or one side has a single row"); return CTCLIP_EBADARG; }
  if (dsims ? (!dtext || !dimage || !dtemp) : !sims) { ctclip_set_error("latent_similarity: missing output"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(latent_similarity_kernel, dim3(1), dim3(256), 0, s, text, image, temperature, dsims, sims, dtext, dimage, dtemp, nt, ni, D, 1e-12f);
  return ctclip_check_launch("latent_similarity");
}
