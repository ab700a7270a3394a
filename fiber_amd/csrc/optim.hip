// AdamW over a list of parameter tensors in ONE launch per parameter group, refreshing the bf16 working copies the GEMMs
// read in the same pass (gfx950).
//
// Replaces, on the caller side of the path, transformers==4.6.0 AdamW(correct_bias=True) as used by the reference's
// set_schedule (coarse_grained/fiber/modules/fiber_utils.py:248-252), in ITS form of the rule (not torch.optim.AdamW's):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps);  p -= lr wd p
// i.e. eps is added to the UN-corrected sqrt(v) (an effective eps/sqrt(1-b2^t), ~7x larger at step 1 with b2 = 0.98) and the
// decoupled weight decay is applied to the already updated parameter.
// torch's foreach implementation runs ~12 multi-tensor kernels per group (5.5 ms per step for the 282 M parameters of
// FIBER-Base, 7.9 GB of traffic) and leaves ~230 separate fp32->bf16 cast kernels to the next forward pass; this kernel
// streams p, g, m, v once and writes p, m, v and the bf16 copy (8.4 GB -> HBM-bound at ~1.7 ms).
#include "common.h"

namespace {

constexpr int CHUNK = 4096;   // elements per workgroup

struct AdamArgs {
  const long long* table;     // [n][5] device pointers: param fp32, grad fp32, exp_avg fp32, exp_avg_sq fp32, bf16 copy (or 0)
  const long long* numel;     // [n]
  const int* chunks;          // [nchunks][2] = (tensor index, chunk index within the tensor)
  float lr, wd, b1, b2, eps, step_size;   // step_size = lr * sqrt(1 - b2^t) / (1 - b1^t)
  const float* hyper;                     // optional DEVICE {lr, step_size}: overrides the two by-value fields (hipGraph replay)
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
  m = a.b1 * m + (1.f - a.b1) * g;
  v = a.b2 * v + (1.f - a.b2) * g * g;
  p -= a.step_size * (m / (sqrtf(v) + a.eps));
  p -= a.lr * a.wd * p;                                     // decoupled weight decay, after the Adam update (HF order)
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamArgs a) {
  if (a.hyper) { a.lr = a.hyper[0]; a.step_size = a.hyper[1]; }
  const int t = a.chunks[2 * blockIdx.x], c = a.chunks[2 * blockIdx.x + 1];
  float* p = reinterpret_cast<float*>(a.table[5 * t + 0]);
  const float* g = reinterpret_cast<const float*>(a.table[5 * t + 1]);
  float* m = reinterpret_cast<float*>(a.table[5 * t + 2]);
  float* v = reinterpret_cast<float*>(a.table[5 * t + 3]);
  bf16* w = reinterpret_cast<bf16*>(a.table[5 * t + 4]);
  const long long n = a.numel[t];
  const long long lo = (long long)c * CHUNK, hi = lo + CHUNK < n ? lo + CHUNK : n;
  // parameters / moments / bf16 copies are separate allocations (aligned); gradients may be views into DDP's flat buckets
  // (gradient_as_bucket_view) and then start at any multiple of 4 bytes: they are read with four scalar loads in that case
  const bool vec = (((a.table[5 * t + 0] | a.table[5 * t + 2] | a.table[5 * t + 3]) & 15) == 0) && ((a.table[5 * t + 4] & 7) == 0);
  const bool gvec = (a.table[5 * t + 1] & 15) == 0;
  long long i = lo + threadIdx.x * 4;
  if (vec && gvec && hi - lo == CHUNK) {
    // a full chunk: all sixteen loads of the thread's four float4 columns requested before the first update (one iteration at a time left the
    // step of the 220 M parameters at 4.2 TB/s of its 6.6 GB; tools/optim_bench.py)
    float4 pp[4], gg[4], mm[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pp[u] = *reinterpret_cast<float4*>(p + i + u * 1024); gg[u] = *reinterpret_cast<const float4*>(g + i + u * 1024);
      mm[u] = *reinterpret_cast<float4*>(m + i + u * 1024); vv[u] = *reinterpret_cast<float4*>(v + i + u * 1024);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      adam1(pp[u].x, gg[u].x, mm[u].x, vv[u].x, a); adam1(pp[u].y, gg[u].y, mm[u].y, vv[u].y, a);
      adam1(pp[u].z, gg[u].z, mm[u].z, vv[u].z, a); adam1(pp[u].w, gg[u].w, mm[u].w, vv[u].w, a);
      *reinterpret_cast<float4*>(p + i + u * 1024) = pp[u];
      *reinterpret_cast<float4*>(m + i + u * 1024) = mm[u];
      *reinterpret_cast<float4*>(v + i + u * 1024) = vv[u];
      if (w) {
        bf16x4 o;
        o[0] = f2bf(pp[u].x); o[1] = f2bf(pp[u].y); o[2] = f2bf(pp[u].z); o[3] = f2bf(pp[u].w);
        *reinterpret_cast<bf16x4*>(w + i + u * 1024) = o;
      }
    }
  } else if (vec) {
    for (; i + 3 < hi; i += 1024) {
      float4 pp = *reinterpret_cast<float4*>(p + i);
      const float4 gg = gvec ? *reinterpret_cast<const float4*>(g + i) : float4{g[i], g[i + 1], g[i + 2], g[i + 3]};
      float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      adam1(pp.x, gg.x, mm.x, vv.x, a); adam1(pp.y, gg.y, mm.y, vv.y, a);
      adam1(pp.z, gg.z, mm.z, vv.z, a); adam1(pp.w, gg.w, mm.w, vv.w, a);
      *reinterpret_cast<float4*>(p + i) = pp;
      *reinterpret_cast<float4*>(m + i) = mm;
      *reinterpret_cast<float4*>(v + i) = vv;
      if (w) {
        bf16x4 o;
        o[0] = f2bf(pp.x); o[1] = f2bf(pp.y); o[2] = f2bf(pp.z); o[3] = f2bf(pp.w);
        *reinterpret_cast<bf16x4*>(w + i) = o;
      }
    }
    // ragged tail of the tensor (n % 4 != 0) falls through to the scalar loop below: only the last chunk has one
    i = lo + ((hi - lo) & ~3LL) + threadIdx.x;
    for (; i < hi; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam1(pp, g[i], mm, vv, a);
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (w) w[i] = f2bf(pp);
    }
  } else {
    for (i = lo + threadIdx.x; i < hi; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam1(pp, g[i], mm, vv, a);
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (w) w[i] = f2bf(pp);
    }
  }
}

}  // namespace

extern "C" int fiber_adamw_chunk(void) { return CHUNK; }

// One AdamW step for every tensor of a parameter group.  table: int64[n*5] device pointers (param, grad, exp_avg,
// exp_avg_sq, bf16 working copy or 0), numel: int64[n], chunks: int32[nchunks*2] (tensor, chunk) pairs covering each tensor
// in pieces of fiber_adamw_chunk() elements -- all three arrays in device memory.  step >= 1 is the step being taken.
// hyper (nullable): DEVICE float[2] = {lr, lr * sqrt(1 - beta2^step) / (1 - beta1^step)} read by the kernel instead of the
// by-value lr / step, so that a captured hipGraph follows the learning-rate schedule (the host refreshes it before a replay).
extern "C" int fiber_adamw_multi_f32(const long long* table, const long long* numel, const int* chunks, int nchunks, float lr,
                                     float weight_decay, float beta1, float beta2, float eps, int step, const float* hyper,
                                     hipStream_t stream) {
  if (nchunks <= 0) return FIBER_OK;
  if (step < 1) return FIBER_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  AdamArgs a{table, numel, chunks, lr, weight_decay, beta1, beta2, eps, (float)((double)lr * sqrt(bc2) / bc1), hyper};
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(256), 0, stream, a);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// All transposed bf16 working copies in ONE launch.  The dgrad GEMMs read W^T [K, N] (both operands K-contiguous in the NT kernel);
// those copies were one strided-copy kernel per weight per optimizer step: 233 launches x 8.6 us.  Here a table of
// {src [N, K] bf16, dst [K, N] bf16, N, K, first tile} drives 64 x 64 tiles through LDS: 16-byte row reads, 16-byte row writes.
namespace {

struct TrDesc { const bf16* src; bf16* dst; int N, K, tile0, tiles_k; };   // 32 bytes

__global__ __launch_bounds__(256) void transpose_multi_kernel(const TrDesc* __restrict__ table, int ndesc) {
  __shared__ bf16 tile[64][72];                          // 144-byte rows: the 2-byte column reads of 8 lanes fall on 8 bank pairs
  int lo = 0, hi = ndesc - 1;                            // descriptor whose tile range holds blockIdx.x (tile0 ascending)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const TrDesc d = table[lo];
  const int t = blockIdx.x - d.tile0;
  const int n0 = (t / d.tiles_k) * 64, k0 = (t % d.tiles_k) * 64;
  const int r = threadIdx.x >> 3, c = threadIdx.x & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + r + h * 32, k = k0 + c * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f2bf(0.f);
    if (n < d.N && k < d.K) v = *reinterpret_cast<const bf16x8*>(d.src + (size_t)n * d.K + k);   // K % 8 == 0
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r + h * 32][c * 8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = k0 + r + h * 32, n = n0 + c * 8;
    if (k < d.K && n < d.N) {                            // N % 8 == 0
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[c * 8 + e][r + h * 32];
      *reinterpret_cast<bf16x8*>(d.dst + (size_t)k * d.N + n) = v;
    }
  }
}

}  // namespace

// table: device array of ndesc 32-byte records {src, dst (8-byte pointers), int32 N, K, tile0, tiles_k}, tile0 ascending with
// tile0[0] = 0, tiles of a [N, K] weight = ceil(N/64) * ceil(K/64); ntiles = their total.  dst[k][n] = src[n][k].  N % 8 == K % 8 == 0.
extern "C" int fiber_transpose_multi_bf16(const void* table, int ndesc, int ntiles, hipStream_t stream) {
  if (ndesc <= 0 || ntiles <= 0) return FIBER_OK;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, (const TrDesc*)table, ndesc);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row-permuted bf16 working copies of fp32 weights (the head-major qkv projections of the window-attention blocks, ops._LinearQKVHeadMajor:
// rows reordered to [heads][3][32]) in ONE launch after the optimizer step: dst[n][:] = bf16(src[perm[n]][:]), its transpose
// dst_t[:][n] for the dX GEMM, and the permuted fp32 bias.  Before, each of the 24 projections was refreshed by five ATen launches
// (index, cast, index, clone, strided copy) per step.
namespace {

struct PermDesc { const float* src; const int* perm; bf16* dst; bf16* dst_t; const float* bias; float* bias_dst; int N, K, tile0, tiles_k; };   // 64 bytes

__global__ __launch_bounds__(256) void rowperm_cast_multi_kernel(const PermDesc* __restrict__ table, int ndesc) {
  __shared__ bf16 tile[64][72];
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PermDesc d = table[lo];
  const int t = blockIdx.x - d.tile0;
  const int n0 = (t / d.tiles_k) * 64, k0 = (t % d.tiles_k) * 64;
  const int r = threadIdx.x >> 3, c = threadIdx.x & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + r + h * 32, k = k0 + c * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f2bf(0.f);
    if (n < d.N && k < d.K) {                            // K % 8 == 0
      const float4* sp = reinterpret_cast<const float4*>(d.src + (size_t)d.perm[n] * d.K + k);
      const float4 a = sp[0], b = sp[1];
      v[0] = f2bf(a.x); v[1] = f2bf(a.y); v[2] = f2bf(a.z); v[3] = f2bf(a.w);
      v[4] = f2bf(b.x); v[5] = f2bf(b.y); v[6] = f2bf(b.z); v[7] = f2bf(b.w);
      *reinterpret_cast<bf16x8*>(d.dst + (size_t)n * d.K + k) = v;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r + h * 32][c * 8 + e] = v[e];
  }
  if (k0 == 0 && d.bias && threadIdx.x < 64 && n0 + (int)threadIdx.x < d.N) d.bias_dst[n0 + threadIdx.x] = d.bias[d.perm[n0 + threadIdx.x]];
  __syncthreads();
  if (d.dst_t) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + r + h * 32, n = n0 + c * 8;
      if (k < d.K && n < d.N) {                          // N % 8 == 0
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[c * 8 + e][r + h * 32];
        *reinterpret_cast<bf16x8*>(d.dst_t + (size_t)k * d.N + n) = v;
      }
    }
  }
}

}  // namespace

// table: device array of ndesc 64-byte records {const float* src [N,K]; const int32* perm [N]; bf16* dst [N,K]; bf16* dst_t [K,N] or null;
// const float* bias [N] or null; float* bias_dst [N]; int32 N, K, tile0, tiles_k}, tiles as in fiber_transpose_multi_bf16.  N % 8 == K % 8 == 0.
extern "C" int fiber_rowperm_cast_multi_bf16(const void* table, int ndesc, int ntiles, hipStream_t stream) {
  if (ndesc <= 0 || ntiles <= 0) return FIBER_OK;
  hipLaunchKernelGGL(rowperm_cast_multi_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, (const PermDesc*)table, ndesc);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
