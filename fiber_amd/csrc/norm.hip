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

// sum over the LPR lanes that share a row (LPR = 64 -> whole wave)
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LPR lanes cooperate on one row (64/LPR rows per wave, so narrow rows such as C=128 still use all 64 lanes);
// NV 16-byte vectors per lane (NV > 1 only when LPR == 64).
// U: row groups a wave has IN FLIGHT per iteration.  All U loads are issued before the first reduction: with one row per wave
// and iteration (1 KB outstanding, then a 12-shuffle dependent chain, then the store) the kernels sat at 2.4 TB/s -- the guide's
// streaming regime needs >= 32 KB of loads in flight per CU ALL the time, not only while every wave happens to be loading.
// X32: x is the fp32 residual stream (ops.py "stream pair": fp32 payload + bf16 shadow) instead of a bf16 tensor;  y32 (optional):
// the output is ALSO stored in fp32 -- RoBERTa is post-LN, its LayerNorm output is the next residual (roberta.py:485,422).
template <int LPR, int NV, bool MERGE, bool X32, int U>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ xv, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* __restrict__ y,
                                                     float* __restrict__ y32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int C, float eps, MergeMap mm) {
  const bf16* x = reinterpret_cast<const bf16*>(xv);
  const float* xf = reinterpret_cast<const float*>(xv);
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, gl = lane % LPR;
  const int nvec = C >> 3;
  float g[NV][8], b[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = gl + i * LPR;
#pragma unroll
    for (int e = 0; e < 8; ++e) { g[i][e] = vi < nvec ? gamma[vi * 8 + e] : 0.f; b[i][e] = vi < nvec ? beta[vi * 8 + e] : 0.f; }
  }
  for (int base = (blockIdx.x * 4 + wave) * (U * RPW); base < rows; base += gridDim.x * 4 * (U * RPW)) {
    float v[U][NV][8];
    bf16x8 raw[U][NV];
    // ---- every load of the U row groups first
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * RPW + sub;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int vi = gl + i * LPR;
        if (row < rows && vi < nvec) {
          const size_t off = MERGE ? mm.src(row, vi * 8) : (size_t)row * C + vi * 8;
          if constexpr (X32) {
            const float4 t0 = *reinterpret_cast<const float4*>(xf + off), t1 = *reinterpret_cast<const float4*>(xf + off + 4);
            v[u][i][0] = t0.x; v[u][i][1] = t0.y; v[u][i][2] = t0.z; v[u][i][3] = t0.w;
            v[u][i][4] = t1.x; v[u][i][5] = t1.y; v[u][i][6] = t1.z; v[u][i][7] = t1.w;
          } else {
            raw[u][i] = *reinterpret_cast<const bf16x8*>(x + off);
          }
        } else {
          if constexpr (X32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][i][e] = 0.f;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[u][i][e] = (bf16)0.f;
          }
        }
      }
    }
    // ---- statistics of all U groups (independent shuffle chains interleave)
    float mu[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if constexpr (!X32) v[u][i][e] = bf2f(raw[u][i][e]);
          s += v[u][i][e];
        }
      mu[u] = s;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mu[u] = group_sum<LPR>(mu[u]) / C;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (gl + i * LPR < nvec)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[u][i][e] - mu[u]; q += d * d; }
      rs[u] = q;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) rs[u] = rsqrtf(group_sum<LPR>(rs[u]) / C + eps);
    // ---- normalise + store
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * RPW + sub;
      const bool rok = row < rows;
      if (rok && gl == 0) { mean[row] = mu[u]; rstd[row] = rs[u]; }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int vi = gl + i * LPR;
        if (rok && vi < nvec) {
          bf16x8 o;
          float of[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { of[e] = (v[u][i][e] - mu[u]) * rs[u] * g[i][e] + b[i][e]; o[e] = f2bf(of[e]); }
          *reinterpret_cast<bf16x8*>(y + (size_t)row * C + vi * 8) = o;
          if (y32) {
            float* yp = y32 + (size_t)row * C + vi * 8;
            *reinterpret_cast<float4*>(yp) = float4{of[0], of[1], of[2], of[3]};
            *reinterpret_cast<float4*>(yp + 4) = float4{of[4], of[5], of[6], of[7]};
          }
        }
      }
    }
  }
}

// dx = rstd * (dy*g - mean_c(dy*g) - xhat * mean_c(dy*g*xhat)); one fp32 partial row of dgamma/dbeta per workgroup.
template <int LPR, int NV, bool MERGE, bool X32, int U>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, const void* __restrict__ xv,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const bf16* __restrict__ dres,
                                                     bf16* __restrict__ dx,
                                                     float* __restrict__ part /*[grid][2][C]*/, int rows, int C,
                                                     MergeMap mm, float* __restrict__ zero_g, float* __restrict__ zero_b) {
  constexpr int RPW = 64 / LPR;
  // the fold kernel that follows accumulates into dgamma / dbeta with atomics: they are zeroed here (the fold runs after this
  // kernel in stream order) instead of by two memset launches per LayerNorm backward (164 launches per step)
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += 256) { zero_g[c] = 0.f; zero_b[c] = 0.f; }
  const bf16* x = reinterpret_cast<const bf16*>(xv);
  const float* xf = reinterpret_cast<const float*>(xv);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, gl = lane % LPR;
  const int nvec = C >> 3;
  float ag[NV][8], ab[NV][8], g[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = gl + i * LPR;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; g[i][e] = vi < nvec ? gamma[vi * 8 + e] : 0.f; }
  }
  for (int base = (blockIdx.x * 4 + wave) * (U * RPW); base < rows; base += gridDim.x * 4 * (U * RPW)) {
    float xh[U][NV][8], dg[U][NV][8];
    bf16x8 rx[U][NV], rd[U][NV], rr[U][NV];
    float mu[U], rs[U];
    // ---- every load of the U row groups first (x, dy, the residual-path gradient, the saved statistics)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * RPW + sub;
      const bool rok = row < rows;
      mu[u] = rok ? mean[row] : 0.f;
      rs[u] = rok ? rstd[row] : 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int vi = gl + i * LPR;
        if (rok && vi < nvec) {
          const size_t off = MERGE ? mm.src(row, vi * 8) : (size_t)row * C + vi * 8;
          if constexpr (X32) {
            const float4 t0 = *reinterpret_cast<const float4*>(xf + off), t1 = *reinterpret_cast<const float4*>(xf + off + 4);
            xh[u][i][0] = t0.x; xh[u][i][1] = t0.y; xh[u][i][2] = t0.z; xh[u][i][3] = t0.w;
            xh[u][i][4] = t1.x; xh[u][i][5] = t1.y; xh[u][i][6] = t1.z; xh[u][i][7] = t1.w;
          } else {
            rx[u][i] = *reinterpret_cast<const bf16x8*>(x + off);
          }
          rd[u][i] = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * C + vi * 8);
          if (dres) rr[u][i] = *reinterpret_cast<const bf16x8*>(dres + off);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            rd[u][i][e] = (bf16)0.f;
            if constexpr (X32) xh[u][i][e] = mu[u]; else rx[u][i][e] = (bf16)0.f;     // mu = 0 here: xhat = 0
          }
        }
      }
    }
    // ---- per-row sums of all U groups, then the (independent) shuffle chains
    float s1[U], s2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = bf2f(rd[u][i][e]);
          const float xv1 = X32 ? xh[u][i][e] : bf2f(rx[u][i][e]);
          xh[u][i][e] = (xv1 - mu[u]) * rs[u];
          dg[u][i][e] = d * g[i][e];
          s1[u] += dg[u][i][e];
          s2[u] += dg[u][i][e] * xh[u][i][e];
          ag[i][e] += d * xh[u][i][e];
          ab[i][e] += d;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { s1[u] = group_sum<LPR>(s1[u]) / C; s2[u] = group_sum<LPR>(s2[u]) / C; }
    // ---- dx
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + u * RPW + sub;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int vi = gl + i * LPR;
        if (row < rows && vi < nvec) {
          const size_t off = MERGE ? mm.src(row, vi * 8) : (size_t)row * C + vi * 8;
          bf16x8 o;
          if (dres) {                                      // fused residual-path gradient: dx = LN'(dy) + dres
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(rs[u] * (dg[u][i][e] - s1[u] - xh[u][i][e] * s2[u]) + bf2f(rr[u][i][e]));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(rs[u] * (dg[u][i][e] - s1[u] - xh[u][i][e] * s2[u]));
          }
          *reinterpret_cast<bf16x8*>(dx + off) = o;
        }
      }
    }
  }
  // combine the RPW row-groups of this wave, then one fp32 partial row per wave
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int o = 32; o >= LPR; o >>= 1) { ag[i][e] += __shfl_xor(ag[i][e], o); ab[i][e] += __shfl_xor(ab[i][e], o); }
  // ... and the four waves of the workgroup through LDS: ONE partial row per workgroup (round 4: the fold kernel read four times the rows,
  // 16 MB per LayerNorm backward at C = 512, 82 launches of 13.5 us per step).  The same lanes own the same columns in every wave, so the
  // waves add in turn behind barriers (no atomics, fixed order).
  __shared__ float red[2 * LPR * NV * 8];
  constexpr int CP = LPR * NV * 8;                       // >= C
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w && sub == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int vi = gl + i * LPR;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (w == 0) { red[vi * 8 + e] = ag[i][e]; red[CP + vi * 8 + e] = ab[i][e]; }
          else { red[vi * 8 + e] += ag[i][e]; red[CP + vi * 8 + e] += ab[i][e]; }
        }
      }
    }
    __syncthreads();
  }
  float* my = part + (size_t)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < C; c += 256) { my[c] = red[c]; my[C + c] = red[CP + c]; }
}

// fold the per-wave partial rows: block = 64 columns x 4 row lanes (coalesced along columns); the row range is split
// over blockIdx.y and combined with one fp32 atomic per column (outputs pre-zeroed by the launcher)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int nrows, int C) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;            // column in the [2][C] partial row
  const int per = (nrows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  float s = 0.f;
  if (col < 2 * C)
    for (int r = r0 + rl; r < r1; r += 4) s += part[(size_t)r * 2 * C + col];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && col < 2 * C) {
    s = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    atomicAdd(col < C ? dgamma + col : dbeta + (col - C), s);
  }
}

// row groups in flight per wave, forward / backward (A/B builds: -DLN_UF=1 -DLN_UB=1 is the one-row-at-a-time form).  tools/ln_bench.py,
// 256 images: C = 128 forward 264 -> 230 us (4.6 -> 5.3 TB/s), backward + dres 495 -> 466; C = 512 forward 68.7 -> 62.9; rows of two
// vectors per lane (C = 768 / 1024): forward two groups (30.6 -> 18.2 us at 10240 x 768), backward one (two measured 12 % slower)
#ifndef LN_UF
#define LN_UF 4
#endif
#ifndef LN_UB
#define LN_UB 4
#endif
#define LN_DISPATCH(KERNEL, LN_U, LN_U2, ...)                                                             \
  do {                                                                                      \
    const int nvec = C >> 3;                                                                \
    if (nvec <= 8) KERNEL(8, 1, LN_U, __VA_ARGS__);                                         \
    else if (nvec <= 16) KERNEL(16, 1, LN_U, __VA_ARGS__);                                  \
    else if (nvec <= 32) KERNEL(32, 1, LN_U, __VA_ARGS__);                                  \
    else if (nvec <= 64) KERNEL(64, 1, LN_U, __VA_ARGS__);                                  \
    else if (nvec <= 128) KERNEL(64, 2, LN_U2, __VA_ARGS__);                                \
    else if (nvec <= 256) KERNEL(64, 4, 1, __VA_ARGS__);                                    \
    else if (nvec <= 384) KERNEL(64, 6, 1, __VA_ARGS__);   /* 4C = 3072: Swin-L's last merge */ \
    else if (nvec <= 512) {                                                                 \
      /* (the PatchMerging forms stop at 4C = 3072; their 4096-wide instance spilled 52 bytes and no registered model has it) */ \
      if constexpr (MERGE) return FIBER_EINVAL; else KERNEL(64, 8, 1, __VA_ARGS__);         \
    } else return FIBER_EINVAL;                                                             \
  } while (0)

// rows one wave takes per iteration (row groups in flight x rows per group)
inline int rows_per_wave(int C) {
  const int nvec = C >> 3;
  const int u = nvec <= 64 ? LN_UF : nvec <= 128 ? (LN_UF > 1 ? 2 : 1) : 1;
  return u * (nvec <= 8 ? 8 : nvec <= 16 ? 4 : nvec <= 32 ? 2 : 1);
}

template <bool MERGE, bool X32>
int launch_fwd(const void* x, const float* g, const float* b, bf16* y, float* y32, float* mean, float* rstd, int rows, int C,
               float eps, MergeMap mm, hipStream_t st) {
  const int need = cdiv(rows, 4 * rows_per_wave(C));
  // A wave loads gamma / beta (2 x 4 B per column) once and then walks its rows: with one iteration per wave -- every tensor below ~130 k rows at
  // the 4096-workgroup cap -- that preload is more bytes than the wave's rows (text stack, 20480 x 768: 8 KB of parameters for 3 KB of rows).
  // Four iterations per wave: 34 -> 22 us there, 86 -> 68 us at 73728 x 1024; the large tensors sit at the cap either way
  // (tools/probes/ln_small_bench.py; FIBER_LN_FWD_DIV=1 restores one iteration).
  static const int div = getenv("FIBER_LN_FWD_DIV") ? atoi(getenv("FIBER_LN_FWD_DIV")) : 4;
  const int want = cdiv(need, div < 1 ? 1 : div);
  const int grid = want < 4096 ? want : 4096;
#define FWD(LPR, NV, UU, ...) hipLaunchKernelGGL((ln_fwd_kernel<LPR, NV, MERGE, X32, UU>), dim3(grid), dim3(256), 0, st, __VA_ARGS__)
  LN_DISPATCH(FWD, LN_UF, (LN_UF > 1 ? 2 : 1), x, g, b, y, y32, mean, rstd, rows, C, eps, mm);
#undef FWD
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

template <bool MERGE, bool X32>
int launch_bwd(const bf16* dy, const void* x, const float* g, const float* mean, const float* rstd, const bf16* dres,
               bf16* dx, float* dgamma, float* dbeta, float* ws, int grid, int rows, int C, MergeMap mm, hipStream_t st) {
#define BWD(LPR, NV, UU, ...) hipLaunchKernelGGL((ln_bwd_kernel<LPR, NV, MERGE, X32, UU>), dim3(grid), dim3(256), 0, st, __VA_ARGS__)
  LN_DISPATCH(BWD, LN_UB, 1, dy, x, g, mean, rstd, dres, dx, ws, rows, C, mm, dgamma, dbeta);
#undef BWD
  FIBER_CHECK_LAUNCH();
  const int nrows = grid, ysplit = nrows >= 64 ? 16 : 1;          // one partial row per workgroup
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * C, 64), ysplit), dim3(256), 0, st, ws, dgamma, dbeta, nrows, C);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

}  // namespace

// Number of workgroups the backward uses for `rows` rows: the caller sizes the fp32 workspace as grid*8*C floats.
extern "C" int fiber_layernorm_bwd_grid(int rows) {
  int g = cdiv(rows, 4 * 16);                            // (8 or 4 rows per wave: 32 -> 27 -> ~22 us at the text stack's 20480 x 768, nothing above: kept)
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

// y = LN(x) * gamma + beta over the last dim C (C % 8 == 0, C <= 4096); saves per-row mean / rstd (fp32).
extern "C" int fiber_layernorm_fwd_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                        float* rstd, int rows, int C, float eps, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (C & 7) return FIBER_EINVAL;
  return launch_fwd<false, false>(x, gamma, beta, (bf16*)y, nullptr, mean, rstd, rows, C, eps, MergeMap{0, 0, 0}, stream);
}

// The same for the fp32 residual stream.  flags bit 0: x is fp32 [rows, C] (else bf16).  y32 (optional): fp32 copy of the output
// next to the bf16 one (post-LN text stack: the LayerNorm output is the next residual, roberta.py:485,422).
extern "C" int fiber_layernorm_fwd_stream(const void* x, const float* gamma, const float* beta, void* y, float* y32, float* mean,
                                          float* rstd, int rows, int C, float eps, int flags, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (C & 7) return FIBER_EINVAL;
  if (flags & 1) return launch_fwd<false, true>(x, gamma, beta, (bf16*)y, y32, mean, rstd, rows, C, eps, MergeMap{0, 0, 0}, stream);
  return launch_fwd<false, false>(x, gamma, beta, (bf16*)y, y32, mean, rstd, rows, C, eps, MergeMap{0, 0, 0}, stream);
}

extern "C" int fiber_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean,
                                        const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                        float* workspace, int rows, int C, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (C & 7) return FIBER_EINVAL;
  return launch_bwd<false, false>((const bf16*)dy, x, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, workspace,
                                  fiber_layernorm_bwd_grid(rows), rows, C, MergeMap{0, 0, 0}, stream);
}

// Backward with the SAVED INPUT in fp32 (flags bit 0; the fp32 residual stream): gradients stay bf16 (dy, dres, dx).
extern "C" int fiber_layernorm_bwd_stream(const void* dy, const void* x, const float* gamma, const float* mean,
                                          const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                          float* workspace, int rows, int C, int flags, hipStream_t stream) {
  if (rows <= 0) return FIBER_OK;
  if (C & 7) return FIBER_EINVAL;
  if (flags & 1)
    return launch_bwd<false, true>((const bf16*)dy, x, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, workspace,
                                   fiber_layernorm_bwd_grid(rows), rows, C, MergeMap{0, 0, 0}, stream);
  return launch_bwd<false, false>((const bf16*)dy, x, gamma, mean, rstd, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, workspace,
                                  fiber_layernorm_bwd_grid(rows), rows, C, MergeMap{0, 0, 0}, stream);
}

// PatchMerging front half: y[b,(i,j),:] = LN(concat(x[2i,2j], x[2i+1,2j], x[2i,2j+1], x[2i+1,2j+1])) over 4C.
// x: [B, H*W, C] bf16; y: [B, H*W/4, 4C] bf16; gamma/beta fp32[4C].
extern "C" int fiber_patch_merge_ln_fwd_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                             float* rstd, int B, int H, int W, int C, float eps, hipStream_t stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return FIBER_EINVAL;
  return launch_fwd<true, false>(x, gamma, beta, (bf16*)y, nullptr, mean, rstd, B * (H / 2) * (W / 2), 4 * C, eps,
                                 MergeMap{H, W, C}, stream);
}

// PatchMerging on the fp32 residual stream (flags bit 0: x is fp32 [B, H*W, C]); the output stays bf16 (it feeds the GEMM).
extern "C" int fiber_patch_merge_ln_fwd_stream(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                               float* rstd, int B, int H, int W, int C, float eps, int flags, hipStream_t stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return FIBER_EINVAL;
  const int rows = B * (H / 2) * (W / 2);
  if (flags & 1) return launch_fwd<true, true>(x, gamma, beta, (bf16*)y, nullptr, mean, rstd, rows, 4 * C, eps, MergeMap{H, W, C}, stream);
  return launch_fwd<true, false>(x, gamma, beta, (bf16*)y, nullptr, mean, rstd, rows, 4 * C, eps, MergeMap{H, W, C}, stream);
}

// Backward of the above: dy [B, H*W/4, 4C] -> dx [B, H*W, C] (every source element is written exactly once).
extern "C" int fiber_patch_merge_ln_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean,
                                             const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace,
                                             int B, int H, int W, int C, hipStream_t stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return FIBER_EINVAL;
  const int rows = B * (H / 2) * (W / 2);
  return launch_bwd<true, false>((const bf16*)dy, x, gamma, mean, rstd, nullptr, (bf16*)dx, dgamma, dbeta, workspace,
                                 fiber_layernorm_bwd_grid(rows), rows, 4 * C, MergeMap{H, W, C}, stream);
}

extern "C" int fiber_patch_merge_ln_bwd_stream(const void* dy, const void* x, const float* gamma, const float* mean,
                                               const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace,
                                               int B, int H, int W, int C, int flags, hipStream_t stream) {
  if ((C & 7) || (H & 1) || (W & 1)) return FIBER_EINVAL;
  const int rows = B * (H / 2) * (W / 2);
  if (flags & 1)
    return launch_bwd<true, true>((const bf16*)dy, x, gamma, mean, rstd, nullptr, (bf16*)dx, dgamma, dbeta, workspace,
                                  fiber_layernorm_bwd_grid(rows), rows, 4 * C, MergeMap{H, W, C}, stream);
  return launch_bwd<true, false>((const bf16*)dy, x, gamma, mean, rstd, nullptr, (bf16*)dx, dgamma, dbeta, workspace,
                                 fiber_layernorm_bwd_grid(rows), rows, 4 * C, MergeMap{H, W, C}, stream);
}
