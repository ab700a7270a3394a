// Memory-bound helper kernels of the FIBER fused path (gfx950): everything is 16-byte vectorised bf16 with
// fp32 math, grid-stride, ~2048 workgroups (HBM roofline work; nothing here is GEMM shaped).
//
//   gelu_bwd          dH = dG * gelu'(H)                    (timm Mlp / RobertaIntermediate exact-erf GELU backward)
//   scale_add         out = a + alpha * b                   (x + alpha_i2t*y swin_transformer.py:259; alpha_t2i roberta.py:483)
//   dot_scaled        *out += sum(a*b)                      (d alpha = <dOut, branch>)
//   colsum            db[n] = sum_m dY[m,n]                 (bias gradients of every nn.Linear)
//   dropout           y = keep ? x/(1-p) : 0                (RoBERTa hidden dropout, roberta.py:198,339,420)
//   rowscale_add      out = r + s[row/rows_per_sample] * x  (timm DropPath on the residual branch, swin_transformer.py:390-391)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ dg, const bf16* __restrict__ h,
                                                       bf16* __restrict__ dh, size_t nvec) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const bf16x8 a = reinterpret_cast<const bf16x8*>(dg)[i], b = reinterpret_cast<const bf16x8*>(h)[i];
    reinterpret_cast<bf16x8*>(dh)[i] = gelu_grad_mul8(a, b);
  }
}

__global__ __launch_bounds__(256) void scale_add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                        const float* __restrict__ alpha, float mult,
                                                        bf16* __restrict__ out, size_t nvec) {
  const float s = (alpha ? alpha[0] : 1.f) * mult;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const bf16x8 y = reinterpret_cast<const bf16x8*>(b)[i];
    bf16x8 o;
    if (a) {
      const bf16x8 x = reinterpret_cast<const bf16x8*>(a)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(x[e]) + s * bf2f(y[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(s * bf2f(y[e]));
    }
    reinterpret_cast<bf16x8*>(out)[i] = o;
  }
}

__global__ __launch_bounds__(256) void dot_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                  float* __restrict__ out, size_t nvec) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const bf16x8 x = reinterpret_cast<const bf16x8*>(a)[i], y = reinterpret_cast<const bf16x8*>(b)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f(x[e]) * bf2f(y[e]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// column sums of a [M, N] bf16 matrix -> fp32.  block = 32 column-vectors x 8 row lanes covering 256 columns and one
// row slab; writes out[blockIdx.y][col] (no atomics, no pre-zeroing); a second pass folds the slabs when gridDim.y > 1.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, float* __restrict__ out, int M, int N,
                                                     int ld, int rows_per_block) {
  __shared__ float red[8][32 * 8 + 1];
  const int cv = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cv) * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {                       // 4 independent 16-byte loads in flight per thread
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(x + (size_t)r * ld + col);
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 8) * ld + col);
      const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 16) * ld + col);
      const bf16x8 v3 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 24) * ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (bf2f(v0[e]) + bf2f(v1[e])) + (bf2f(v2[e]) + bf2f(v3[e]));
    }
    for (; r < r1; r += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (size_t)r * ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += bf2f(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cv * 8 + e] = s[e];
  __syncthreads();
  const int c = threadIdx.x;
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) t += red[r][c];
  const int gc = blockIdx.x * 256 + c;
  if (gc < N) out[(size_t)blockIdx.y * N + gc] = t;
}

// dh = dg * gelu'(h) with the column sums of dh (fc1 bias gradient) produced in the same pass: same slab layout as
// colsum_kernel (block = 256 columns x one row slab, 8 row lanes), so the 4C-wide dh is not re-read for the bias grad.
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const bf16* __restrict__ dg, const bf16* __restrict__ h,
                                                              bf16* __restrict__ dh, float* __restrict__ out, int M, int N,
                                                              int rows_per_block) {
  __shared__ float red[8][32 * 8 + 1];
  const int cv = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cv) * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    auto one = [&](int r, const bf16x8& a, const bf16x8& b) {
      const bf16x8 o = gelu_grad_mul8(a, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += bf2f(o[e]);
      *reinterpret_cast<bf16x8*>(dh + (size_t)r * N + col) = o;
    };
    int r = r0 + rl;
    for (; r + 8 < r1; r += 16) {                        // two row groups (4 loads) in flight per thread
      const size_t o0 = (size_t)r * N + col, o1 = (size_t)(r + 8) * N + col;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(dg + o0), b0 = *reinterpret_cast<const bf16x8*>(h + o0);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(dg + o1), b1 = *reinterpret_cast<const bf16x8*>(h + o1);
      one(r, a0, b0); one(r + 8, a1, b1);
    }
    for (; r < r1; r += 8) {
      const size_t off = (size_t)r * N + col;
      one(r, *reinterpret_cast<const bf16x8*>(dg + off), *reinterpret_cast<const bf16x8*>(h + off));
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cv * 8 + e] = s[e];
  __syncthreads();
  const int c = threadIdx.x;
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) t += red[r][c];
  const int gc = blockIdx.x * 256 + c;
  if (gc < N) out[(size_t)blockIdx.y * N + gc] = t;
}

// ds = scale[row / rows_per_sample] * dy (DropPath backward on the branch) with the column sums of ds (bias gradient of the
// proj / fc2 linear) produced in the same pass: same slab layout as colsum_kernel.
__global__ __launch_bounds__(256) void rowscale_colsum_kernel(const bf16* __restrict__ x, const float* __restrict__ scale,
                                                              bf16* __restrict__ y, float* __restrict__ out, int M, int N,
                                                              int rows_per_block, int rows_per_sample) {
  __shared__ float red[8][32 * 8 + 1];
  const int cv = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cv) * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    auto one = [&](int r, const bf16x8& a) {
      const float sc = scale[r / rows_per_sample];
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[e] = f2bf(sc * bf2f(a[e])); s[e] += bf2f(o[e]); }
      *reinterpret_cast<bf16x8*>(y + (size_t)r * N + col) = o;
    };
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {                       // 4 independent 16-byte loads in flight per thread
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(x + (size_t)r * N + col);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 8) * N + col);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 16) * N + col);
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 24) * N + col);
      one(r, a0); one(r + 8, a1); one(r + 16, a2); one(r + 24, a3);
    }
    for (; r < r1; r += 8) one(r, *reinterpret_cast<const bf16x8*>(x + (size_t)r * N + col));
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cv * 8 + e] = s[e];
  __syncthreads();
  const int c = threadIdx.x;
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) t += red[r][c];
  const int gc = blockIdx.x * 256 + c;
  if (gc < N) out[(size_t)blockIdx.y * N + gc] = t;
}

// out[c] = sum_y part[y, c].  The slab count reaches 512 while N can be as small as 128, so the work is spread over 16 row
// lanes x 16 columns per block (N/16 blocks) with four independent loads in flight per thread: the previous 64-column x
// 4-lane shape ran 2 blocks of 128 dependent loads each (25 us of pure latency per call, ~210 calls per step).
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int slabs, int N) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < N) {
    int y = rl;
    for (; y + 48 < slabs; y += 64) {
      s0 += part[(size_t)y * N + c];
      s1 += part[(size_t)(y + 16) * N + c];
      s2 += part[(size_t)(y + 32) * N + c];
      s3 += part[(size_t)(y + 48) * N + c];
    }
    for (; y < slabs; y += 16) s0 += part[(size_t)y * N + c];
  }
  red[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    out[c] = t;
  }
}

__global__ __launch_bounds__(256) void dropout_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t nvec,
                                                      uint64_t seed, const uint64_t* __restrict__ seed_base, uint32_t thresh,
                                                      float inv_keep) {
  if (seed_base) seed += *seed_base;                   // graph replay: the per-step part of the key lives in device memory
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    bf16x8 o;
    const uint32_t h0 = drop_base(seed, i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = drop_keep_e(h0, e, thresh) ? f2bf(bf2f(v[e]) * inv_keep) : f2bf(0.f);
    reinterpret_cast<bf16x8*>(y)[i] = o;
  }
}

// timm DropPath factor per sample: floor(keep + U[0,1)) / keep, U from the counter-based hash of (key, sample)
__global__ __launch_bounds__(256) void droppath_scale_kernel(float* __restrict__ out, int n, float keep, uint64_t seed,
                                                             const uint64_t* __restrict__ seed_base) {
  if (seed_base) seed += *seed_base;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float u = (float)hash_u32(seed, (uint64_t)i) * (1.0f / 4294967296.0f);
    out[i] = floorf(keep + u) / keep;
  }
}

__global__ __launch_bounds__(256) void rowscale_add_kernel(const bf16* __restrict__ r, const bf16* __restrict__ x,
                                                           const float* __restrict__ scale, bf16* __restrict__ out,
                                                           size_t nvec, size_t vec_per_sample) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const float s = scale[i / vec_per_sample];
    const bf16x8 b = reinterpret_cast<const bf16x8*>(x)[i];
    bf16x8 o;
    if (r) {
      const bf16x8 a = reinterpret_cast<const bf16x8*>(r)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(a[e]) + s * bf2f(b[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(s * bf2f(b[e]));
    }
    reinterpret_cast<bf16x8*>(out)[i] = o;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Residual-stream add with everything the reference does between a branch and its residual in ONE pass:
//     out = res + rowscale[sample] * ( drop_a(a) + alpha * drop_b(b) )
//   res:  fp32 (RES = 2, the fp32 residual stream), bf16 (RES = 1) or absent (RES = 0)
//   a, b: bf16 branches (b optional); alpha: device scalar (alpha_i2t / alpha_t2i) or NULL = 1
//   drop_a / drop_b: hidden dropout of the branch (RoBERTa dense -> dropout -> add, roberta.py:337-340,417-423), counter-based
//         keep-masks keyed by (seed, element): thresh 0 = off.  rowscale: per-sample timm DropPath factor or NULL
//   out32 (fp32) and / or out16 (bf16 shadow) -- at least one.
// Replaces scale_add + rowscale_add (Swin fused blocks, swin_transformer.py:259 + :390), dropout + add and scale_add + add
// (RobertaSelfOutput / RobertaOutput / RobertaLayer, roberta.py:339,420,483-485): 2-3 launches and as many passes -> one.
struct StreamAddArgs {
  const void* res; const bf16* a; const bf16* b; const float* alpha; const float* rowscale;
  float* out32; bf16* out16;
  size_t nvec, vec_per_sample;
  uint64_t seed_a, seed_b; const uint64_t* seed_base;
  uint32_t thresh_a, thresh_b; float inv_keep_a, inv_keep_b;
};

template <int RES>
__global__ __launch_bounds__(256) void stream_add_kernel(StreamAddArgs p) {
  const float al = p.alpha ? p.alpha[0] : 1.f;
  uint64_t sa = p.seed_a, sb = p.seed_b;
  if (p.seed_base) { sa += *p.seed_base; sb += *p.seed_base; }
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < p.nvec; i += (size_t)gridDim.x * 256) {
    const float rs = p.rowscale ? p.rowscale[i / p.vec_per_sample] : 1.f;
    const bf16x8 av = reinterpret_cast<const bf16x8*>(p.a)[i];
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(av[e]);
    if (p.thresh_a) {
      const uint32_t ha = drop_base(sa, i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = drop_keep_e(ha, e, p.thresh_a) ? v[e] * p.inv_keep_a : 0.f;
    }
    if (p.b) {
      const bf16x8 bv = reinterpret_cast<const bf16x8*>(p.b)[i];
      const uint32_t hb = drop_base(sb, i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = bf2f(bv[e]);
        if (p.thresh_b) t = drop_keep_e(hb, e, p.thresh_b) ? t * p.inv_keep_b : 0.f;
        v[e] += al * t;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= rs;
    if constexpr (RES == 2) {
      const float* rp = reinterpret_cast<const float*>(p.res) + i * 8;
      const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    } else if constexpr (RES == 1) {
      const bf16x8 rv = reinterpret_cast<const bf16x8*>(p.res)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(rv[e]);
    }
    if (p.out32) {
      float* op = p.out32 + i * 8;
      *reinterpret_cast<float4*>(op) = float4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<float4*>(op + 4) = float4{v[4], v[5], v[6], v[7]};
    }
    if (p.out16) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
      reinterpret_cast<bf16x8*>(p.out16)[i] = o;
    }
  }
}

// Backward of stream_add with respect to the branches: da = rowscale * mask_a * dy, db = rowscale * alpha * mask_b * dy,
// dalpha += sum(rowscale * dy * drop_b(b)) (one fp32 atomic per workgroup; the caller zeroes it).  The residual gradient is dy.
struct StreamAddBwdArgs {
  const bf16* dy; const bf16* b; const float* alpha; const float* rowscale;
  bf16* da; bf16* db; float* dalpha;
  size_t nvec, vec_per_sample;
  uint64_t seed_a, seed_b; const uint64_t* seed_base;
  uint32_t thresh_a, thresh_b; float inv_keep_a, inv_keep_b;
};

__global__ __launch_bounds__(256) void stream_add_bwd_kernel(StreamAddBwdArgs p) {
  __shared__ float red[4];
  const float al = p.alpha ? p.alpha[0] : 1.f;
  uint64_t sa = p.seed_a, sb = p.seed_b;
  if (p.seed_base) { sa += *p.seed_base; sb += *p.seed_base; }
  float acc = 0.f;
  // Four vectors per thread and iteration, every load issued before the first use: with the grid capped at 512 workgroups for the gate's one
  // atomic per workgroup, one 16-byte load per thread in flight left the kernel at 4.1 TB/s of its bytes (LayerNorm: 5.4).
  constexpr int U = 4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < p.nvec; i0 += stride * U) {
    bf16x8 dvs[U], bvs[U];
    float rss[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < p.nvec) {
        dvs[u] = reinterpret_cast<const bf16x8*>(p.dy)[i];
        if (p.db && p.dalpha) bvs[u] = reinterpret_cast<const bf16x8*>(p.b)[i];
        rss[u] = p.rowscale ? p.rowscale[i / p.vec_per_sample] : 1.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i >= p.nvec) break;
      const float rs = rss[u];
      const bf16x8 dv = dvs[u];
      float g[8];
      const uint32_t ha = drop_base(sa, i * 8), hb = drop_base(sb, i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = rs * bf2f(dv[e]);
      if (p.da) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = f2bf(p.thresh_a ? (drop_keep_e(ha, e, p.thresh_a) ? g[e] * p.inv_keep_a : 0.f) : g[e]);
        reinterpret_cast<bf16x8*>(p.da)[i] = o;
      }
      if (p.db) {
        const bf16x8 bv = bvs[u];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float m = p.thresh_b ? (drop_keep_e(hb, e, p.thresh_b) ? p.inv_keep_b : 0.f) : 1.f;
          o[e] = f2bf(al * m * g[e]);
          if (p.dalpha) acc += g[e] * m * bf2f(bv[e]);
        }
        reinterpret_cast<bf16x8*>(p.db)[i] = o;
      }
    }
  }
  if (p.dalpha) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(p.dalpha, red[0] + red[1] + red[2] + red[3]);
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t nvec) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    bf16x8 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w), f2bf(b.x), f2bf(b.y), f2bf(b.z), f2bf(b.w)};
    reinterpret_cast<bf16x8*>(y)[i] = o;
  }
}

inline int ew_grid(size_t nvec) {
  size_t g = (nvec + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int fiber_gelu_bwd_bf16(const void* dgelu, const void* h_pre, void* dh, long n, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if (n & 7) return FIBER_EINVAL;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16*)dgelu, (const bf16*)h_pre, (bf16*)dh, (size_t)n / 8);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// out = a + alpha[0]*mult*b   (a may be NULL -> out = alpha*mult*b; alpha may be NULL -> 1)
extern "C" int fiber_scale_add_bf16(const void* a, const void* b, const float* alpha, float mult, void* out, long n,
                                    hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if (n & 7) return FIBER_EINVAL;
  hipLaunchKernelGGL(scale_add_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, alpha, mult, (bf16*)out, (size_t)n / 8);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// out[0] += sum(a*b)   (out must be zero-initialised by the caller for a plain dot product)
extern "C" int fiber_dot_bf16(const void* a, const void* b, float* out, long n, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if (n & 7) return FIBER_EINVAL;
  size_t nvec = n / 8;
  int grid = (int)((nvec + 255) / 256);
  grid = grid > 512 ? 512 : grid;
  hipLaunchKernelGGL(dot_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, out, nvec);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Row slabs the column-sum uses for an [M, N] input: the caller provides a workspace of slabs*N floats when slabs > 1.
extern "C" int fiber_colsum_slabs(int M, int N) {
  const int gx = cdiv(N, 256);
  int gy = cdiv(M, 512);
  int cap = cdiv(1024, gx);
  cap = cap > 512 ? 512 : cap;
  gy = gy > cap ? cap : gy;
  return gy < 1 ? 1 : gy;
}

// out[n] = sum_m x[m,n]   (out fp32[N], overwritten); N % 8 == 0; workspace fp32[slabs*N] (may be NULL when slabs == 1)
extern "C" int fiber_colsum_bf16(const void* x, float* out, float* workspace, int M, int N, int ld, hipStream_t stream) {
  if (M <= 0 || N <= 0) return FIBER_OK;
  if ((N & 7) || (ld & 7)) return FIBER_EINVAL;
  const int gx = cdiv(N, 256);
  const int gy0 = fiber_colsum_slabs(M, N);
  const int rpb = cdiv(M, gy0), gy = cdiv(M, rpb);
  if (gy > 1 && !workspace) return FIBER_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16*)x, gy > 1 ? workspace : out, M, N, ld, rpb);
  FIBER_CHECK_LAUNCH();
  if (gy > 1) {
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(cdiv(N, 16)), dim3(256), 0, stream, workspace, out, gy, N);
    FIBER_CHECK_LAUNCH();
  }
  return FIBER_OK;
}

// dh = dgelu * gelu'(h_pre) and db[n] = sum_m dh[m,n] in one pass over contiguous [M, N] tensors (N % 8 == 0).
// workspace: fp32[fiber_colsum_slabs(M,N) * N] (may be NULL when that is 1).
extern "C" int fiber_gelu_bwd_colsum_bf16(const void* dgelu, const void* h_pre, void* dh, float* db, float* workspace, int M,
                                          int N, hipStream_t stream) {
  if (M <= 0 || N <= 0) return FIBER_OK;
  if (N & 7) return FIBER_EINVAL;
  const int gx = cdiv(N, 256);
  const int gy0 = fiber_colsum_slabs(M, N);
  const int rpb = cdiv(M, gy0), gy = cdiv(M, rpb);
  if (gy > 1 && !workspace) return FIBER_EINVAL;
  hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16*)dgelu, (const bf16*)h_pre, (bf16*)dh,
                     gy > 1 ? workspace : db, M, N, rpb);
  FIBER_CHECK_LAUNCH();
  if (gy > 1) {
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(cdiv(N, 16)), dim3(256), 0, stream, workspace, db, gy, N);
    FIBER_CHECK_LAUNCH();
  }
  return FIBER_OK;
}

// y = scale[row / rows_per_sample] * x and db[n] = sum_m y[m,n] in one pass over contiguous [M, N] tensors (N % 8 == 0);
// workspace as for fiber_gelu_bwd_colsum_bf16.
extern "C" int fiber_rowscale_colsum_bf16(const void* x, const float* scale, void* y, float* db, float* workspace, int M, int N,
                                          int rows_per_sample, hipStream_t stream) {
  if (M <= 0 || N <= 0) return FIBER_OK;
  if ((N & 7) || rows_per_sample <= 0) return FIBER_EINVAL;
  const int gx = cdiv(N, 256);
  const int gy0 = fiber_colsum_slabs(M, N);
  const int rpb = cdiv(M, gy0), gy = cdiv(M, rpb);
  if (gy > 1 && !workspace) return FIBER_EINVAL;
  hipLaunchKernelGGL(rowscale_colsum_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16*)x, scale, (bf16*)y,
                     gy > 1 ? workspace : db, M, N, rpb, rows_per_sample);
  FIBER_CHECK_LAUNCH();
  if (gy > 1) {
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(cdiv(N, 16)), dim3(256), 0, stream, workspace, db, gy, N);
    FIBER_CHECK_LAUNCH();
  }
  return FIBER_OK;
}

// out[n] = sum_r part[r, n]  (fold of per-tile / per-slab partial rows, fp32)
extern "C" int fiber_fold_rows_f32(const float* part, float* out, int rows, int N, hipStream_t stream) {
  if (rows <= 0 || N <= 0) return FIBER_OK;
  hipLaunchKernelGGL(colsum_fold_kernel, dim3(cdiv(N, 16)), dim3(256), 0, stream, part, out, rows, N);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// y = keep(key, i) ? x / (1-p) : 0 ; the same call with dy as x gives the backward.  key = seed + *seed_base (seed_base:
// optional DEVICE pointer to the per-step part of the key, so that a captured hipGraph draws new masks on every replay).
extern "C" int fiber_dropout_bf16(const void* x, void* y, long n, float p, uint64_t seed, const uint64_t* seed_base,
                                  hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if ((n & 7) || p < 0.f || p >= 1.f) return FIBER_EINVAL;
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, (size_t)n / 8, seed, seed_base, thresh, 1.f / (1.f - p));
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// timm 0.4.12 DropPath per-sample factors (swin_transformer.py:322): out[b] = floor(keep + U_b) / keep, fp32 [n]
extern "C" int fiber_droppath_scale_f32(float* out, int n, float keep, uint64_t seed, const uint64_t* seed_base,
                                        hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if (keep <= 0.f || keep > 1.f) return FIBER_EINVAL;
  hipLaunchKernelGGL(droppath_scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, out, n, keep, seed, seed_base);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// out = r + scale[sample] * x  with `per_sample` contiguous elements per sample (r may be NULL)
extern "C" int fiber_rowscale_add_bf16(const void* r, const void* x, const float* scale, void* out, long n,
                                       long per_sample, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if ((n & 7) || (per_sample & 7)) return FIBER_EINVAL;
  hipLaunchKernelGGL(rowscale_add_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16*)r, (const bf16*)x, scale, (bf16*)out, (size_t)n / 8, (size_t)per_sample / 8);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// out = res + rowscale[sample] * (drop_a(a) + alpha * drop_b(b)); see stream_add_kernel.  res_kind: 0 none, 1 bf16, 2 fp32.
// p_a / p_b: dropout probabilities of the two branches (0 = off) with by-value keys seed_a / seed_b (+ *seed_base if non-NULL).
// out32 / out16: fp32 result and / or its bf16 rounding (at least one).  n % 8 == 0, per_sample % 8 == 0 when rowscale is given.
extern "C" int fiber_stream_add(const void* res, int res_kind, const void* a, const void* b, const float* alpha,
                                const float* rowscale, long per_sample, float p_a, uint64_t seed_a, float p_b, uint64_t seed_b,
                                const uint64_t* seed_base, float* out32, void* out16, long n, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if ((n & 7) || !a || (!out32 && !out16) || (res_kind && !res) || res_kind < 0 || res_kind > 2) return FIBER_EINVAL;
  if (rowscale && (per_sample <= 0 || (per_sample & 7))) return FIBER_EINVAL;
  if (p_a < 0.f || p_a >= 1.f || p_b < 0.f || p_b >= 1.f || (p_b > 0.f && !b)) return FIBER_EINVAL;
  StreamAddArgs p{res, (const bf16*)a, (const bf16*)b, alpha, rowscale, out32, (bf16*)out16, (size_t)n / 8,
                  rowscale ? (size_t)per_sample / 8 : (size_t)1, seed_a, seed_b, seed_base,
                  (uint32_t)((double)p_a * 4294967296.0), (uint32_t)((double)p_b * 4294967296.0), 1.f / (1.f - p_a), 1.f / (1.f - p_b)};
  const int grid = ew_grid(p.nvec);
  if (res_kind == 2) hipLaunchKernelGGL(stream_add_kernel<2>, dim3(grid), dim3(256), 0, stream, p);
  else if (res_kind == 1) hipLaunchKernelGGL(stream_add_kernel<1>, dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(stream_add_kernel<0>, dim3(grid), dim3(256), 0, stream, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Branch gradients of fiber_stream_add: da (optional), db (optional; needs b when dalpha is wanted), dalpha (optional fp32
// scalar, accumulated: zero it first).  dy bf16 [n].
extern "C" int fiber_stream_add_bwd(const void* dy, const void* b, const float* alpha, const float* rowscale, long per_sample,
                                    float p_a, uint64_t seed_a, float p_b, uint64_t seed_b, const uint64_t* seed_base, void* da,
                                    void* db, float* dalpha, long n, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if ((n & 7) || !dy || (dalpha && (!b || !db))) return FIBER_EINVAL;
  if (rowscale && (per_sample <= 0 || (per_sample & 7))) return FIBER_EINVAL;
  if (p_a < 0.f || p_a >= 1.f || p_b < 0.f || p_b >= 1.f) return FIBER_EINVAL;
  StreamAddBwdArgs p{(const bf16*)dy, (const bf16*)b, alpha, rowscale, (bf16*)da, (bf16*)db, dalpha, (size_t)n / 8,
                     rowscale ? (size_t)per_sample / 8 : (size_t)1, seed_a, seed_b, seed_base,
                     (uint32_t)((double)p_a * 4294967296.0), (uint32_t)((double)p_b * 4294967296.0), 1.f / (1.f - p_a), 1.f / (1.f - p_b)};
  int grid = ew_grid(p.nvec);
  if (dalpha && grid > 512) grid = 512;                  // one atomic per workgroup
  hipLaunchKernelGGL(stream_add_bwd_kernel, dim3(grid), dim3(256), 0, stream, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// y (bf16) = x (fp32), n % 8 == 0: the bf16 shadow of an fp32 stream tensor where a GEMM consumes it and no producer wrote one
extern "C" int fiber_cast_f32_bf16(const float* x, void* y, long n, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if (n & 7) return FIBER_EINVAL;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, stream, x, (bf16*)y, (size_t)n / 8);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
