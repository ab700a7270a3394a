// Cross-entropy over a wide vocabulary for bf16 logits (gfx950): forward = per-row log-sum-exp + picked logit in ONE read of
// the logits, backward = (softmax - onehot) * scale in one read + one write.  Replaces, on the caller side of the path
// (objectives.py:24-28: F.cross_entropy(mlm_logits.view(-1, vocab), labels, ignore_index=-100) over B*40 x 50265 logits), the
// ATen chain bf16->fp32 copy (2 GB) + log_softmax forward + backward + fp32->bf16, ~3.5 ms and 6 GB of temporaries per step.
// Rows are V = 50265 elements long, so a row starts at an arbitrary 2-byte offset: each row is walked as an unaligned head
// (<= 7 elements), 16-byte vectors, and a tail.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one workgroup (256 threads) per row
__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16* __restrict__ x, const long long* __restrict__ labels,
                                                     float* __restrict__ loss, float* __restrict__ lse, int V, long long ignore) {
  __shared__ float red_m[4], red_s[4];
  const int row = blockIdx.x;
  const long long lab = labels[row];
  if (lab == ignore) {                                   // ignored rows cost nothing (85 % of the MLM rows)
    if (threadIdx.x == 0) { loss[row] = 0.f; lse[row] = 0.f; }
    return;
  }
  const bf16* xr = x + (size_t)row * V;
  const int head = (int)((8 - (((size_t)row * V) & 7)) & 7);          // elements before the first 16-byte boundary
  const int nvec = (V - head) >> 3, tail0 = head + nvec * 8;
  float m = -INFINITY, s = 0.f;                          // online softmax: s = sum exp(x - m)
  auto add = [&](float v) {
    if (v > m) { s = s * __expf(m - v) + 1.f; m = v; } else { s += __expf(v - m); }
  };
  if ((int)threadIdx.x < head) add(bf2f(xr[threadIdx.x]));
  for (int i = threadIdx.x; i < nvec; i += 256) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + head + i * 8);
    float mx = bf2f(v[0]);
#pragma unroll
    for (int e = 1; e < 8; ++e) mx = fmaxf(mx, bf2f(v[e]));
    if (mx > m) { s *= __expf(m - mx); m = mx; }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __expf(bf2f(v[e]) - m);
  }
  if (tail0 + (int)threadIdx.x < V) add(bf2f(xr[tail0 + threadIdx.x]));
  const float wm = wave_max(m);
  s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red_m[wave] = wm; red_s[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float S = 0.f;
    for (int w = 0; w < 4; ++w) S += red_m[w] == -INFINITY ? 0.f : red_s[w] * __expf(red_m[w] - M);
    const float l = M + __logf(S);
    lse[row] = l;
    loss[row] = l - bf2f(xr[lab]);
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const bf16* __restrict__ x, const long long* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ scale,
                                                     bf16* __restrict__ dx, int V, long long ignore) {
  const int row = blockIdx.x;
  const long long lab = labels[row];
  const bf16* xr = x + (size_t)row * V;
  bf16* dr = dx + (size_t)row * V;
  const int head = (int)((8 - (((size_t)row * V) & 7)) & 7);
  const int nvec = (V - head) >> 3, tail0 = head + nvec * 8;
  if (lab == ignore) {                                   // zero gradient row
    const bf16 z = f2bf(0.f);
    if ((int)threadIdx.x < head) dr[threadIdx.x] = z;
    bf16x8 zv;
#pragma unroll
    for (int e = 0; e < 8; ++e) zv[e] = z;
    for (int i = threadIdx.x; i < nvec; i += 256) *reinterpret_cast<bf16x8*>(dr + head + i * 8) = zv;
    if (tail0 + (int)threadIdx.x < V) dr[tail0 + threadIdx.x] = z;
    return;
  }
  const float l = lse[row], sc = scale[0];
  auto g1 = [&](int j) { return (__expf(bf2f(xr[j]) - l) - (j == lab ? 1.f : 0.f)) * sc; };
  if ((int)threadIdx.x < head) dr[threadIdx.x] = f2bf(g1(threadIdx.x));
  for (int i = threadIdx.x; i < nvec; i += 256) {
    const int j0 = head + i * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + j0);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf((__expf(bf2f(v[e]) - l) - (j0 + e == lab ? 1.f : 0.f)) * sc);
    *reinterpret_cast<bf16x8*>(dr + j0) = o;
  }
  if (tail0 + (int)threadIdx.x < V) dr[tail0 + threadIdx.x] = f2bf(g1(tail0 + threadIdx.x));
}

}  // namespace

// loss[r] = logsumexp(x[r, :]) - x[r, labels[r]] (0 where labels[r] == ignore_index); lse[r] saved for the backward.
// x: bf16 [rows, V] contiguous; labels: int64 [rows]; V > 16.
extern "C" int fiber_ce_fwd_bf16(const void* logits, const long long* labels, float* loss, float* lse, int rows, int V,
                                 long long ignore_index, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (V <= 16) return FIBER_EINVAL;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, stream, (const bf16*)logits, labels, loss, lse, V, ignore_index);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// dlogits[r, j] = (exp(x[r, j] - lse[r]) - [j == labels[r]]) * scale[0]  (zero rows where labels[r] == ignore_index);
// scale: one fp32 in device memory (upstream gradient / number of valid rows), so no host synchronisation is needed.
extern "C" int fiber_ce_bwd_bf16(const void* logits, const long long* labels, const float* lse, const float* scale, void* dlogits,
                                 int rows, int V, long long ignore_index, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (V <= 16) return FIBER_EINVAL;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(rows), dim3(256), 0, stream, (const bf16*)logits, labels, lse, scale, (bf16*)dlogits, V,
                     ignore_index);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
