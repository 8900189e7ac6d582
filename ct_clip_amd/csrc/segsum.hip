// Deterministic segmented row sum: out[key[r]][:] (+)= scale[r] * x[r][:] summed IN ROW ORDER, for the scatter-add shaped
// reductions of the hot path -- the vector-quantiser EMA statistics (vector_quantize_pytorch 1.1.2 cosine codebook,
// called at ctvit.py:403: embed_sum = onehot^T flatten) and the BERT embedding-table gradients (HF BertEmbeddings backward).
// The first versions used f32 atomics (847 us for the VQ statistics and a summation order that changed from run to run).
//
// Five launches, no float atomics:
//   1. seg_hist    one workgroup per chunk of 1024 rows: integer histogram of the keys in LDS -> hist[chunk][seg]
//   2. seg_chunkscan / seg_first   per segment, exclusive scan over the chunks (in place) and the count; then an exclusive
//                  scan of the counts over the segments -> first[seg]
//   3. seg_place   one workgroup per chunk, one thread per row: rank of a row among the EARLIER rows of its chunk with the same key (counted, not
//                  taken from an atomic's return value) -> order[first + chunk base + rank] = row      (a stable counting sort)
//   4. seg_sum     one wave per segment walks its rows in ascending row order with 16-byte loads and f32 accumulators.
#include "common.h"

namespace {

constexpr int CHUNK = 1024;

__device__ __forceinline__ int key_of(const int64_t* keys, int64_t r, int mod) { return keys ? (int)keys[r] : (int)(r % mod); }

__global__ __launch_bounds__(256) void seg_hist_kernel(const int64_t* __restrict__ keys, int mod, int64_t M, int nseg, int* __restrict__ hist) {
  extern __shared__ int lh[];
  for (int i = threadIdx.x; i < nseg; i += 256) lh[i] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * CHUNK;
  for (int i = threadIdx.x; i < CHUNK; i += 256) {
    const int64_t r = r0 + i;
    if (r < M) {
      const int k = key_of(keys, r, mod);
      if (k >= 0 && k < nseg) atomicAdd(&lh[k], 1);        // integer: the result does not depend on the order
    }
  }
  __syncthreads();
  int* dst = hist + (int64_t)blockIdx.x * nseg;
  for (int i = threadIdx.x; i < nseg; i += 256) dst[i] = lh[i];
}

// per segment: exclusive scan of its per-chunk counts (in place) and the total
__global__ __launch_bounds__(256) void seg_chunkscan_kernel(int* __restrict__ hist, int nchunks, int nseg, int* __restrict__ count,
                                                            float* __restrict__ count_f) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nseg) return;
  int run = 0;
  for (int c = 0; c < nchunks; ++c) {
    int* p = hist + (int64_t)c * nseg + s;
    const int t = *p; *p = run; run += t;
  }
  count[s] = run;
  if (count_f) count_f[s] = (float)run;
}
// first[s] = number of rows in segments < s (one workgroup)
__global__ __launch_bounds__(1024) void seg_first_kernel(const int* __restrict__ count, int nseg, int* __restrict__ first) {
  __shared__ int part[1024];
  const int per = (nseg + 1023) / 1024;
  const int s0 = threadIdx.x * per;
  int local = 0;
  for (int s = s0; s < s0 + per && s < nseg; ++s) local += count[s];
  part[threadIdx.x] = local;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {              // Hillis-Steele inclusive scan of the 1024 partial sums
    const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - local;
  for (int s = s0; s < s0 + per && s < nseg; ++s) { first[s] = run; run += count[s]; }
}

// One thread per row of the chunk.  rank = number of earlier rows of the chunk with the same key: the keys of the earlier WAVES' rows are
// compared four at a time from broadcast ds_read_b128 (a wave-uniform trip count), the 63 rows of the own wave under a lane mask.  (The
// first version gave a thread four rows and one dependent 4-byte LDS read per comparison: 79 us per launch with 14 workgroups on the
// chip, 20 launches per VocabFine step.)
__global__ __launch_bounds__(CHUNK) void seg_place_kernel(const int64_t* __restrict__ keys, int mod, int64_t M, int nseg,
                                                          const int* __restrict__ hist, const int* __restrict__ first, int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) int lk[CHUNK];
  const int i = threadIdx.x;
  const int64_t r = (int64_t)blockIdx.x * CHUNK + i;
  int k = -1;
  if (r < M) { k = key_of(keys, r, mod); if (k < 0 || k >= nseg) k = -1; }
  lk[i] = k;
  __syncthreads();
  if (k < 0) return;
  const int4* lk4 = reinterpret_cast<const int4*>(lk);
  const int wb4 = (i & ~63) >> 2;
  int rank = 0;
#pragma unroll 8
  for (int j = 0; j < wb4; ++j) {
    const int4 v = lk4[j];
    rank += (v.x == k) + (v.y == k) + (v.z == k) + (v.w == k);
  }
  const int li = i & 63;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int4 v = lk4[wb4 + j];
    rank += (v.x == k && 4 * j < li) + (v.y == k && 4 * j + 1 < li) + (v.z == k && 4 * j + 2 < li) + (v.w == k && 4 * j + 3 < li);
  }
  order[first[k] + hist[(int64_t)blockIdx.x * nseg + k] + rank] = (int)r;
}

template <typename T>
__global__ __launch_bounds__(256) void seg_sum_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ rowscale,
                                                      const int* __restrict__ order, const int* __restrict__ first,
                                                      const int* __restrict__ count, float* __restrict__ out, int nseg, int d, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const int n = count[seg], f = first[seg];
  if (n == 0 && accumulate) return;
  constexpr int MAXV = 4;                                  // d <= 2048
  float acc[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  // Rows are added in ascending order (fixed), but eight rows are FETCHED at a time: a popular segment (a vector-quantiser code that
  // holds thousands of tokens early in training) is one wave walking its rows, and with one dependent load per row that tail cost
  // 0.7 ms per step.
  constexpr int UNR = 8;
  for (int t0 = 0; t0 < n; t0 += UNR) {
    int64_t r[UNR]; float sc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + u < n ? t0 + u : n - 1;
      r[u] = order[f + t];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) sc[u] = (t0 + u < n) ? (rowscale ? rowscale[r[u]] : 1.f) : 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < d) {
        float v[UNR][8];
#pragma unroll
        for (int u = 0; u < UNR; ++u) load8(x + r[u] * ldx + c, v[u]);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[i][e] = fmaf(v[u][e], sc[u], acc[i][e]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < d) {
      float* o = out + (int64_t)seg * d + c;
      if (accumulate) {
        float old[8];
        load8(o, old);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] += old[e];
      }
      store8(o, acc[i]);
    }
  }
}

inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// workspace: hist[nchunks][nseg] + first[nseg] + count[nseg] + order[M]   (int32 each)
extern "C" int64_t ctclip_segment_sum_workspace(int64_t M, int nseg) {
  const int64_t nchunks = cdiv(M, CHUNK);
  return align256(nchunks * nseg * 4) + 2 * align256((int64_t)nseg * 4) + align256(M * 4);
}

// out[seg][0:d] (+)= sum over rows r with key r == seg, in ascending r, of rowscale[r] * x[r][0:d]      (f32 accumulate)
// keys: int64 [M] or null (then key(r) = r % key_mod); rowscale: f32 [M] or null; counts_f: optional f32 [nseg] histogram.
// x: (M, >= d) rows of in_dtype with row stride ldx; out: f32 (nseg, d) contiguous; d % 8 == 0, d <= 2048, nseg * 4 <= 160 KiB.
extern "C" int ctclip_segment_sum(const int64_t* keys, int key_mod, const void* x, int64_t ldx, const float* rowscale, float* out,
                                  float* counts_f, int64_t M, int d, int nseg, int accumulate, int in_dtype, void* workspace,
                                  int64_t workspace_bytes, hipStream_t stream) {
  if (!x || !out || M <= 0 || d % 8 || d > 2048 || ldx % 8 || nseg <= 0 || nseg > 40000 || (!keys && key_mod <= 0)) {
    ctclip_set_error("segment_sum: d % 8 == 0, d <= 2048, 0 < nseg <= 40000, keys or key_mod required");
    return CTCLIP_EBADARG;
  }
  if (!workspace || workspace_bytes < ctclip_segment_sum_workspace(M, nseg)) { ctclip_set_error("segment_sum: workspace too small"); return CTCLIP_EWORKSPACE; }
  const int64_t nchunks = cdiv(M, CHUNK);
  char* w = (char*)workspace;
  int* hist = (int*)w; w += align256(nchunks * nseg * 4);
  int* first = (int*)w; w += align256((int64_t)nseg * 4);
  int* count = (int*)w; w += align256((int64_t)nseg * 4);
  int* order = (int*)w;
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute((const void*)seg_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { ctclip_set_error("segment_sum: cannot raise the LDS limit"); return CTCLIP_EBADARG; }
    raised = true;
  }
  hipLaunchKernelGGL(seg_hist_kernel, dim3((unsigned)nchunks), dim3(256), (size_t)nseg * 4, stream, keys, key_mod, M, nseg, hist);
  hipLaunchKernelGGL(seg_chunkscan_kernel, dim3((unsigned)cdiv(nseg, 256)), dim3(256), 0, stream, hist, (int)nchunks, nseg, count, counts_f);
  hipLaunchKernelGGL(seg_first_kernel, dim3(1), dim3(1024), 0, stream, count, nseg, first);
  hipLaunchKernelGGL(seg_place_kernel, dim3((unsigned)nchunks), dim3(CHUNK), 0, stream, keys, key_mod, M, nseg, hist, first, order);
  if (in_dtype == DT_BF16) hipLaunchKernelGGL(seg_sum_kernel<bf16_t>, dim3((unsigned)cdiv(nseg, 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, rowscale, order, first, count, out, nseg, d, accumulate);
  else if (in_dtype == DT_F32) hipLaunchKernelGGL(seg_sum_kernel<float>, dim3((unsigned)cdiv(nseg, 4)), dim3(256), 0, stream, (const float*)x, ldx, rowscale, order, first, count, out, nseg, d, accumulate);
  else return CTCLIP_EUNSUPPORTED;
  return ctclip_check_launch("segment_sum");
}
