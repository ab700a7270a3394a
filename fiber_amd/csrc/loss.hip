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
// pred (nullable): arg max of the row, smallest index among equal maxima (what torch.argmax returns; bf16 logits DO tie), -1 on ignored rows --
// the accuracy metric of the step (my_metrics.py:5-28 takes logits.argmax over ALL rows, a 1-GB pass, and then drops the ignored ones).
__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16* __restrict__ x, const long long* __restrict__ labels,
                                                     float* __restrict__ loss, float* __restrict__ lse, int* __restrict__ pred, int V, long long ignore) {
  __shared__ float red_m[4], red_s[4];
  __shared__ int red_i[4];
  const int row = blockIdx.x;
  const long long lab = labels[row];
  if (lab == ignore) {                                   // ignored rows cost nothing (85 % of the MLM rows)
    if (threadIdx.x == 0) { loss[row] = 0.f; lse[row] = 0.f; if (pred) pred[row] = -1; }
    return;
  }
  const bf16* xr = x + (size_t)row * V;
  const int head = (int)((8 - (((size_t)row * V) & 7)) & 7);          // elements before the first 16-byte boundary
  const int nvec = (V - head) >> 3, tail0 = head + nvec * 8;
  float m = -INFINITY, s = 0.f;                          // online softmax: s = sum exp(x - m)
  int am = V;                                            // index of this thread's first maximum (its indices are visited in ascending order)
  auto add = [&](float v, int j) {
    if (v > m) { s = s * __expf(m - v) + 1.f; m = v; am = j; } else { s += __expf(v - m); }
  };
  if ((int)threadIdx.x < head) add(bf2f(xr[threadIdx.x]), threadIdx.x);
  for (int i = threadIdx.x; i < nvec; i += 256) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + head + i * 8);
    float mx = bf2f(v[0]);
#pragma unroll
    for (int e = 1; e < 8; ++e) mx = fmaxf(mx, bf2f(v[e]));
    if (mx > m) {
      s *= __expf(m - mx); m = mx;
      int e0 = 7;
#pragma unroll
      for (int e = 6; e >= 0; --e) if (bf2f(v[e]) == mx) e0 = e;
      am = head + i * 8 + e0;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __expf(bf2f(v[e]) - m);
  }
  if (tail0 + (int)threadIdx.x < V) add(bf2f(xr[tail0 + threadIdx.x]), tail0 + threadIdx.x);
  const float wm = wave_max(m);
  int wi = m == wm ? am : V;                             // smallest index among the lanes that hold the wave's maximum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wi = min(wi, __shfl_xor(wi, o));
  s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red_m[wave] = wm; red_s[wave] = s; red_i[wave] = wi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float S = 0.f;
    int I = V;
    for (int w = 0; w < 4; ++w) {
      S += red_m[w] == -INFINITY ? 0.f : red_s[w] * __expf(red_m[w] - M);
      if (red_m[w] == M) I = min(I, red_i[w]);
    }
    const float l = M + __logf(S);
    lse[row] = l;
    loss[row] = l - bf2f(xr[lab]);
    if (pred) pred[row] = I;
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

// Column sums over the LABELLED rows of dlogits (rows whose label is the ignore index are exactly zero: ce_bwd_kernel wrote them): the bias
// gradient of the vocabulary decoder without reading the 85 % zero rows.  Workgroup = 256 columns x one slab of rows; the slab's labelled
// rows are listed in LDS first, then every thread walks its column over that list.  part [slabs][V] fp32; fiber_fold_rows_f32 adds the slabs.
__global__ __launch_bounds__(256) void colsum_labelled_kernel(const bf16* __restrict__ x, const long long* __restrict__ labels, float* __restrict__ part,
                                                              int rows, int V, int rows_per_slab, long long ignore) {
  extern __shared__ int list[];
  __shared__ int count;
  const int r0 = blockIdx.y * rows_per_slab, r1 = min(rows, r0 + rows_per_slab);
  if (threadIdx.x < 64) {                                // wave 0 lists the slab's labelled rows in ascending order (fixed summation order)
    int base = 0;
    for (int c0 = r0; c0 < r1; c0 += 64) {
      const int r = c0 + threadIdx.x;
      const bool keep = r < r1 && labels[r] != ignore;
      const unsigned long long mask = __ballot(keep);
      if (keep) list[base + __popcll(mask & ((1ull << threadIdx.x) - 1ull))] = r;
      base += __popcll(mask);
    }
    if (threadIdx.x == 0) count = base;
  }
  __syncthreads();
  const int n = count;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= V) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = 0;
  for (; i + 3 < n; i += 4) {
    a0 += bf2f(x[(size_t)list[i] * V + col]); a1 += bf2f(x[(size_t)list[i + 1] * V + col]);
    a2 += bf2f(x[(size_t)list[i + 2] * V + col]); a3 += bf2f(x[(size_t)list[i + 3] * V + col]);
  }
  for (; i < n; ++i) a0 += bf2f(x[(size_t)list[i] * V + col]);
  part[(size_t)blockIdx.y * V + col] = (a0 + a1) + (a2 + a3);
}

}  // namespace

// loss[r] = logsumexp(x[r, :]) - x[r, labels[r]] (0 where labels[r] == ignore_index); lse[r] saved for the backward.
// x: bf16 [rows, V] contiguous; labels: int64 [rows]; V > 16.
// pred (nullable): int32 [rows], arg max of every labelled row (smallest index among equal maxima), -1 on ignored rows.
extern "C" int fiber_ce_fwd_bf16(const void* logits, const long long* labels, float* loss, float* lse, int* pred, int rows, int V,
                                 long long ignore_index, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (V <= 16) return FIBER_EINVAL;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, stream, (const bf16*)logits, labels, loss, lse, pred, V, ignore_index);
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

extern "C" int fiber_fold_rows_f32(const float* part, float* out, int rows, int N, hipStream_t stream);

// out[j] = sum over the rows r with labels[r] != ignore_index of x[r, j] (x: bf16 [rows, V], the dlogits fiber_ce_bwd_bf16 wrote: its other
// rows are zero).  workspace: fp32 [fiber_colsum_labelled_slabs(rows) * V].
extern "C" int fiber_colsum_labelled_slabs(int rows) { return rows >= 4096 ? 8 : rows >= 512 ? 4 : 1; }
extern "C" int fiber_colsum_labelled_bf16(const void* x, const long long* labels, float* out, float* workspace, int rows, int V,
                                          long long ignore_index, hipStream_t stream) {
  if (rows <= 0 || V <= 0) return FIBER_EINVAL;
  const int slabs = fiber_colsum_labelled_slabs(rows), rps = cdiv(rows, slabs);
  if ((size_t)rps * sizeof(int) > 60 * 1024) return FIBER_EINVAL;
  float* part = slabs > 1 ? workspace : out;
  if (slabs > 1 && !workspace) return FIBER_EINVAL;
  hipLaunchKernelGGL(colsum_labelled_kernel, dim3(cdiv(V, 256), slabs), dim3(256), (size_t)rps * sizeof(int), stream, (const bf16*)x, labels, part,
                     rows, V, rps, ignore_index);
  FIBER_CHECK_LAUNCH();
  if (slabs > 1) return fiber_fold_rows_f32(part, out, slabs, V, stream);
  return FIBER_OK;
}
