// LayerNorm forward / backward (bf16 activations, fp32 statistics and parameters) for gfx950.
//
// Replaces nn.LayerNorm at every site of the reference hot path (swin_transformer.py:362,391 norm1/norm2; :244
// norm_i2t_i; PatchEmbed norm; roberta.py:485,422 attention.output.LayerNorm / output.LayerNorm) and, in MERGE mode,
// fuses PatchMerging's 2x2 strided gather + concat (swin_transformer.py:420-427, order (0,0),(1,0),(0,1),(1,1)) into
// the LayerNorm's loads so the concatenated 4C tensor is produced once, already normalised.
//
// HBM-bound: one wave per row, 16-byte loads (8 bf16 per lane), the whole row held in registers between the
// statistics and normalisation passes (one read, one write per element).  Backward keeps per-lane fp32 partial
// sums of dgamma/dbeta across the rows a wave visits (grid-stride), writes one partial row per wave; a second tiny kernel folds the partials.
#include "common.h"

namespace {

struct MergeMap {  // PatchMerging gather: output row (b,i,j) of width 4C reads 4 source tokens of width C
  int H, W, C;     // source grid
  __device__ __forceinline__ size_t src(int row, int col) const {
    const int Wh = W >> 1, per = (H >> 1) * Wh;
    const int b = row / per, ij = row - b * per, i = ij / Wh, j = ij - i * Wh;
    const int s = col / C, c = col - s * C;
    return ((size_t)(b * H + 2 * i + (s & 1)) * W + 2 * j + (s >> 1)) * C + c;
  }
};

template <int NV, bool MERGE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int C, float eps, MergeMap mm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 3;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const size_t off = MERGE ? mm.src(row, vi * 8) : (size_t)row * C + vi * 8;
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(t[e]); s += v[i][e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
      }
    }
    const float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nvec)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + vi * 8), g1 = *reinterpret_cast<const float4*>(gamma + vi * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + vi * 8), b1 = *reinterpret_cast<const float4*>(beta + vi * 8 + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf((v[i][e] - mu) * rs * g[e] + b[e]);
        *reinterpret_cast<bf16x8*>(y + (size_t)row * C + vi * 8) = o;
      }
    }
  }
}

// dx = rstd * (dy*g - mean_c(dy*g) - xhat * mean_c(dy*g*xhat)); partial dgamma/dbeta per workgroup.
template <int NV, bool MERGE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, bf16* __restrict__ dx,
                                                     float* __restrict__ part /*[grid*4 waves][2][C]*/, int rows, int C,
                                                     MergeMap mm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 3;
  float ag[NV][8], ab[NV][8], g[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; g[i][e] = vi < nvec ? gamma[vi * 8 + e] : 0.f; }
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NV][8], dg[NV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const size_t off = MERGE ? mm.src(row, vi * 8) : (size_t)row * C + vi * 8;
        const bf16x8 tx = *reinterpret_cast<const bf16x8*>(x + off);
        const bf16x8 td = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * C + vi * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = bf2f(td[e]);
          xh[i][e] = (bf2f(tx[e]) - mu) * rs;
          dg[i][e] = d * g[i][e];
          s1 += dg[i][e];
          s2 += dg[i][e] * xh[i][e];
          ag[i][e] += d * xh[i][e];
          ab[i][e] += d;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[i][e] = 0.f; dg[i][e] = 0.f; }
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(rs * (dg[i][e] - s1 - xh[i][e] * s2));
        const size_t off = MERGE ? mm.src(row, vi * 8) : (size_t)row * C + vi * 8;
        *reinterpret_cast<bf16x8*>(dx + off) = o;
      }
    }
  }
  // one fp32 partial row per wave; folded by ln_bwd_reduce_kernel
  float* my = part + (size_t)(blockIdx.x * 4 + wave) * 2 * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        my[vi * 8 + e] = ag[i][e];
        my[C + vi * 8 + e] = ab[i][e];
      }
  }
}

__global__ void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int nblk, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * C) return;
  const int which = c / C, col = c - which * C;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[((size_t)b * 2 + which) * C + col];
  (which ? dbeta : dgamma)[col] = s;
}

template <bool MERGE>
int launch_fwd(const bf16* x, const float* g, const float* b, bf16* y, float* mean, float* rstd, int rows, int C,
               float eps, MergeMap mm, hipStream_t st) {
  const int grid = rows < 4 * 2048 ? cdiv(rows, 4) : 2048;
  const int nv = cdiv(C >> 3, 64);
#define L(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV, MERGE>), dim3(grid), dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, mm)
  if (nv <= 1) L(1); else if (nv <= 2) L(2); else if (nv <= 4) L(4); else if (nv <= 8) L(8); else return FIBER_EINVAL;
#undef L
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

template <bool MERGE>
int launch_bwd(const bf16* dy, const bf16* x, const float* g, const float* mean, const float* rstd, bf16* dx,
               float* dgamma, float* dbeta, float* ws, int grid, int rows, int C, MergeMap mm, hipStream_t st) {
  const int nv = cdiv(C >> 3, 64);
#define L(NV) hipLaunchKernelGGL((ln_bwd_kernel<NV, MERGE>), dim3(grid), dim3(256), 0, st, dy, x, g, mean, rstd, dx, ws, rows, C, mm)
  if (nv <= 1) L(1); else if (nv <= 2) L(2); else if (nv <= 4) L(4); else if (nv <= 8) L(8); else return FIBER_EINVAL;
#undef L
  FIBER_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, st, ws, dgamma, dbeta, grid * 4, C);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

}  // namespace

// Number of workgroups the backward uses for `rows` rows: the caller sizes the fp32 workspace as grid*8*C floats.
extern "C" int fiber_layernorm_bwd_grid(int rows) {
  int g = cdiv(rows, 4 * 8);
  return g < 1 ? 1 : (g > 256 ? 256 : g);
}

// y = LN(x) * gamma + beta over the last dim C (C % 8 == 0, C <= 4096); saves per-row mean / rstd (fp32).
extern "C" int fiber_layernorm_fwd_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                        float* rstd, int rows, int C, float eps, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (C & 7) return FIBER_EINVAL;
  return launch_fwd<false>((const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, rows, C, eps, MergeMap{0, 0, 0}, stream);
}

extern "C" int fiber_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean,
                                        const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace,
                                        int rows, int C, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (C & 7) return FIBER_EINVAL;
  return launch_bwd<false>((const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (bf16*)dx, dgamma, dbeta, workspace,
                           fiber_layernorm_bwd_grid(rows), rows, C, MergeMap{0, 0, 0}, stream);
}

// PatchMerging front half: y[b,(i,j),:] = LN(concat(x[2i,2j], x[2i+1,2j], x[2i,2j+1], x[2i+1,2j+1])) over 4C.
// x: [B, H*W, C] bf16; y: [B, H*W/4, 4C] bf16; gamma/beta fp32[4C].
extern "C" int fiber_patch_merge_ln_fwd_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                             float* rstd, int B, int H, int W, int C, float eps, hipStream_t stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return FIBER_EINVAL;
  return launch_fwd<true>((const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, B * (H / 2) * (W / 2), 4 * C, eps,
                          MergeMap{H, W, C}, stream);
}

// Backward of the above: dy [B, H*W/4, 4C] -> dx [B, H*W, C] (every source element is written exactly once).
extern "C" int fiber_patch_merge_ln_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean,
                                             const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace,
                                             int B, int H, int W, int C, hipStream_t stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return FIBER_EINVAL;
  const int rows = B * (H / 2) * (W / 2);
  return launch_bwd<true>((const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (bf16*)dx, dgamma, dbeta, workspace,
                          fiber_layernorm_bwd_grid(rows), rows, 4 * C, MergeMap{H, W, C}, stream);
}
