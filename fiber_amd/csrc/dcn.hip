// Modulated deformable convolution (DCNv2) sampling kernels for the DyHead of the fine-grained model (gfx950 / CDNA4).
//
// Replaces the reference's CUDA gather / scatter kernels
//   modulated_deformable_im2col_gpu_kernel        fine_grained/maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu:578-640
//   modulated_deformable_col2im_gpu_kernel        :643-700   (gradient to the input feature map)
//   modulated_deformable_col2im_coord_gpu_kernel  :703-773   (gradients to the offsets and the modulation mask)
// and the per-image cuBLAS addmm loop around them (deform_conv_cuda.cu:497-572, 574-692), behind
// layers/deform_conv.py:300-353 (ModulatedDeformConv, used by layers/dyhead.py:14 and modeling/rpn/vldyhead.py:123).
//
// Not a translation.  The reference keeps NCHW planes, builds a [Cin*kh*kw, Ho*Wo] column matrix per IMAGE (one thread per
// (channel, position), 4 scalar gathers per tap) and calls the library once per image.  Here:
//   * the feature map is channels-last bf16 [B, H, W, C] (the layout the fused backbone already produces): a bilinear corner is
//     one contiguous C-vector, so every gather / scatter is a 16-byte-per-lane row access and the (position, tap) geometry is
//     computed once per 8 channels instead of once per channel;
//   * the column matrix is [M = B*Ho*Wo, taps*C] bf16, tap-major -- the K-contiguous A operand of the hand-written NT / TN MFMA
//     GEMMs (gemm.hip, gemm_tn.hip) over the WHOLE batch: one gather launch + one GEMM launch per layer, the bias and the
//     weight / bias gradients ride in the GEMM epilogues;
//   * backward: ONE pass over the column gradients produces all three gradients: lanes of a (position, tap) group own 8 channels
//     each, the input gradient is scattered with fp32 hardware atomics (global_atomic_add_f32), the offset / mask gradients are
//     reduced across the group with DPP shuffles (the reference runs two kernels, the second re-gathering the image per
//     offset CHANNEL: 2 x taps x C/dg gathers per position).
// Both kernels are HBM / L2 bound: algorithmic bytes per position and tap = 2 C (column write) + 4 x 2 C (corner reads, mostly L2
// hits: neighbouring positions share corners).  groups = deformable_groups = 1, dilation 1 (what DyHead instantiates).
#include "common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct DcnP {
  const bf16* x;        // [B, H, W, C]
  const float* offset;  // [M, 2*taps]  (dy, dx) per tap, or NULL (ordinary convolution)
  const float* mask;    // [M, taps] modulation (already through the sigmoid), or NULL (ones)
  int B, H, W, C, Ho, Wo, kh, kw, stride, pad;
};

// bilinear corner geometry of one (position, tap): the reference's rule (deform_conv_kernel_cuda.cu:474-503, 617-627) -- a sample
// is taken when -1 < h < H and -1 < w < W, corners outside the map contribute zero
struct Tap {
  int h0, w0;
  float lh, lw;
  bool in, v00, v01, v10, v11;
  __device__ __forceinline__ void set(float h, float w, int H, int W) {
    in = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
    const float fh = floorf(h), fw = floorf(w);
    h0 = (int)fh; w0 = (int)fw;
    lh = h - fh; lw = w - fw;
    v00 = in && h0 >= 0 && w0 >= 0;
    v01 = in && h0 >= 0 && w0 + 1 <= W - 1;
    v10 = in && h0 + 1 <= H - 1 && w0 >= 0;
    v11 = in && h0 + 1 <= H - 1 && w0 + 1 <= W - 1;
  }
};

__device__ __forceinline__ bf16x8 ld8(const bf16* p, bool ok) {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = f2bf(0.f);
  return ok ? *reinterpret_cast<const bf16x8*>(p) : z;
}

// cols[m, t*C + c] = mask[m,t] * bilinear(x[b, :, :, c], h_t, w_t).  One thread = 8 channels of one (position, tap).
__global__ __launch_bounds__(256) void dcn_gather_kernel(DcnP p, bf16* cols, long total) {
  const int C8 = p.C >> 3, taps = p.kh * p.kw;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    const long mt = idx / C8;
    const int t = (int)(mt % taps);
    const long m = mt / taps;
    const int wo = (int)(m % p.Wo), ho = (int)((m / p.Wo) % p.Ho), b = (int)(m / ((long)p.Wo * p.Ho));
    float h = (float)(ho * p.stride - p.pad + t / p.kw), w = (float)(wo * p.stride - p.pad + t % p.kw);
    float mk = 1.f;
    if (p.offset) { h += p.offset[m * 2 * taps + 2 * t]; w += p.offset[m * 2 * taps + 2 * t + 1]; }
    if (p.mask) mk = p.mask[m * taps + t];
    Tap g;
    g.set(h, w, p.H, p.W);
    const bf16* base = p.x + (((long)b * p.H + g.h0) * p.W + g.w0) * p.C + c8 * 8;
    const bf16x8 a00 = ld8(base, g.v00), a01 = ld8(base + p.C, g.v01);
    const bf16x8 a10 = ld8(base + (long)p.W * p.C, g.v10), a11 = ld8(base + (long)p.W * p.C + p.C, g.v11);
    const float w00 = (1.f - g.lh) * (1.f - g.lw) * mk, w01 = (1.f - g.lh) * g.lw * mk, w10 = g.lh * (1.f - g.lw) * mk, w11 = g.lh * g.lw * mk;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(w00 * bf2f(a00[e]) + w01 * bf2f(a01[e]) + w10 * bf2f(a10[e]) + w11 * bf2f(a11[e]));
    *reinterpret_cast<bf16x8*>(cols + (mt * p.C + c8 * 8)) = o;
  }
}

// Backward of the gather.  A group of G = min(C/8, 64) lanes owns one (position, tap); each lane 8 channels (looping when
// C > 512).  dx += w_corner * mask * dcol (fp32 atomics), dmask = sum_c dcol * sample, doffset = mask * sum_c dcol * d sample / d(h,w)
// (the analytic derivative the reference spells out in dmcn_get_coordinate_weight, deform_conv_kernel_cuda.cu:536-575).
template <int G>
__global__ __launch_bounds__(256) void dcn_scatter_kernel(DcnP p, const bf16* dcols, float* dx, float* doffset, float* dmask, long groups) {
  const int taps = p.kh * p.kw, C8 = p.C >> 3;
  const int gl = threadIdx.x % G;
  for (long mt = ((long)blockIdx.x * 256 + threadIdx.x) / G; mt < groups; mt += (long)gridDim.x * 256 / G) {
    const int t = (int)(mt % taps);
    const long m = mt / taps;
    const int wo = (int)(m % p.Wo), ho = (int)((m / p.Wo) % p.Ho), b = (int)(m / ((long)p.Wo * p.Ho));
    float h = (float)(ho * p.stride - p.pad + t / p.kw), w = (float)(wo * p.stride - p.pad + t % p.kw);
    float mk = 1.f;
    if (p.offset) { h += p.offset[m * 2 * taps + 2 * t]; w += p.offset[m * 2 * taps + 2 * t + 1]; }
    if (p.mask) mk = p.mask[m * taps + t];
    Tap g;
    g.set(h, w, p.H, p.W);
    const float hh = 1.f - g.lh, hw = 1.f - g.lw;
    const long pix = (((long)b * p.H + g.h0) * p.W + g.w0) * p.C;
    float sm = 0.f, sh = 0.f, sw = 0.f;
    for (int c8 = gl; c8 < C8; c8 += G) {
      const bf16x8 d = *reinterpret_cast<const bf16x8*>(dcols + (mt * p.C + c8 * 8));
      const bf16* base = p.x + pix + c8 * 8;
      const bf16x8 a00 = ld8(base, g.v00), a01 = ld8(base + p.C, g.v01);
      const bf16x8 a10 = ld8(base + (long)p.W * p.C, g.v10), a11 = ld8(base + (long)p.W * p.C + p.C, g.v11);
      float* gx = dx + pix + c8 * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dv = bf2f(d[e]);
        const float v00 = bf2f(a00[e]), v01 = bf2f(a01[e]), v10 = bf2f(a10[e]), v11 = bf2f(a11[e]);
        sm += dv * (hh * hw * v00 + hh * g.lw * v01 + g.lh * hw * v10 + g.lh * g.lw * v11);
        sh += dv * (hw * (v10 - v00) + g.lw * (v11 - v01));
        sw += dv * (hh * (v01 - v00) + g.lh * (v11 - v10));
        const float gm = dv * mk;
        if (dx) {
          if (g.v00) unsafeAtomicAdd(gx + e, hh * hw * gm);
          if (g.v01) unsafeAtomicAdd(gx + p.C + e, hh * g.lw * gm);
          if (g.v10) unsafeAtomicAdd(gx + (long)p.W * p.C + e, g.lh * hw * gm);
          if (g.v11) unsafeAtomicAdd(gx + (long)p.W * p.C + p.C + e, g.lh * g.lw * gm);
        }
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sh += __shfl_xor(sh, o); sw += __shfl_xor(sw, o); }
    if (gl == 0) {
      if (dmask) dmask[m * taps + t] = sm;
      if (doffset) { doffset[m * 2 * taps + 2 * t] = sh * mk; doffset[m * 2 * taps + 2 * t + 1] = sw * mk; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Input gradient without device atomics.  fp32 global atomics on a map that all eight XCDs update resolve at the memory side:
// the one-pass kernel above reaches ~40 G atomics/s (15 ms for the 100 x 168 x 256 level at batch 4, 100 x the gather).  Here a
// workgroup owns a T x T tile of OUTPUT positions and a 16-channel slice; every sample of the tile whose bilinear corner falls
// inside the tile's input WINDOW (the tile's receptive field + a halo of ~3 pixels for the learned offsets) is accumulated in LDS,
// the window is then written -- plain, coalesced stores -- to a per-tile buffer, and a second kernel sums, for every input pixel,
// the <= 2 x 2 windows that cover it (static geometry) and writes the bf16 gradient.  Corners that leave the window (offsets
// beyond the halo) fall back to a device atomic on a separate fp32 map that the second kernel adds in: correct for any offset,
// fast for the offsets a trained head produces, deterministic when nothing leaves the windows.
//   LDS accumulation is FIXED POINT on ds_add_u32: float LDS atomics (ds_add_f32) serialise per lane on this part (the same
//   kernel: 3.3 ms with them, 0.89 with integer atomics, 0.78 with racy plain adds).  Scale 2^(19-e) with 2^e > max |dcol| (a
//   small reduction kernel): one contribution is < 2^19 in magnitude and the bilinear weights of a sample sum to <= 1, so a
//   window cell receives at most T*T*taps = 2304 full-size contributions < 2^31 -- no overflow for any input; resolution
//   2^-19 of the largest column gradient per contribution (bf16 keeps 2^-9 of each value).
constexpr int DCN_CS = 16, DCN_WIN = 24;   // window = T*stride + (k - 1) + halo = 24 for (T 16, stride 1) and (T 8, stride 2), k = 3
constexpr int DCN_HALO = 3;

struct DcnT {
  int T, tiles_y, tiles_x;
};

// largest |value| of a bf16 array, as the bit pattern of a non-negative float (ordered like unsigned integers)
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const bf16* v, long n8, unsigned* out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 a = reinterpret_cast<const bf16x8*>(v)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(bf2f(a[e])));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

__device__ __forceinline__ float dcn_fixed_scale(unsigned maxbits) {
  // 2^(19 - e), 2^e > max: the exponent field of max gives floor(log2 max) = E - 127
  const int E = (int)(maxbits >> 23);
  const int ef = min(max(127 + 19 - (E - 127 + 1), 1), 254);      // clamped: a denormal maximum would ask for 2^146
  return maxbits ? __uint_as_float((unsigned)ef << 23) : 1.f;
}

__global__ __launch_bounds__(256) void dcn_dx_tile_kernel(DcnP p, DcnT g, const bf16* dcols, int* tiles, float* far, const unsigned* maxbits) {
  __shared__ __attribute__((aligned(16))) int win[DCN_WIN * DCN_WIN * DCN_CS];             // 36 KB: [pixel][16 channels], fixed point
  const int taps = p.kh * p.kw, nslice = p.C / DCN_CS;
  const int cs = blockIdx.x % nslice;
  const int tile = blockIdx.x / nslice;
  const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, b = tile / (g.tiles_x * g.tiles_y);
  const int y0 = ty * g.T * p.stride - p.pad - DCN_HALO, x0 = tx * g.T * p.stride - p.pad - DCN_HALO;
  const float scale = dcn_fixed_scale(*maxbits);
  for (int i = threadIdx.x; i < DCN_WIN * DCN_WIN * DCN_CS / 4; i += 256) reinterpret_cast<int4*>(win)[i] = int4{0, 0, 0, 0};
  __syncthreads();
  const int grp = threadIdx.x >> 2, l4 = threadIdx.x & 3;                                   // 4 lanes x 4 channels per (position, tap)
  const int npair = g.T * g.T * taps;
  for (int pr = grp; pr < npair; pr += 64) {
    const int pos = pr / taps, t = pr - pos * taps;
    const int ho = ty * g.T + pos / g.T, wo = tx * g.T + pos % g.T;
    if (ho >= p.Ho || wo >= p.Wo) continue;
    const long m = ((long)b * p.Ho + ho) * p.Wo + wo;
    float h = (float)(ho * p.stride - p.pad + t / p.kw), w = (float)(wo * p.stride - p.pad + t % p.kw);
    float mk = 1.f;
    if (p.offset) { h += p.offset[m * 2 * taps + 2 * t]; w += p.offset[m * 2 * taps + 2 * t + 1]; }
    if (p.mask) mk = p.mask[m * taps + t];
    Tap q;
    q.set(h, w, p.H, p.W);
    if (!q.in) continue;
    const bf16x4 d = *reinterpret_cast<const bf16x4*>(dcols + ((m * taps + t) * p.C + cs * DCN_CS + l4 * 4));
    const float hh = 1.f - q.lh, hw = 1.f - q.lw;
    const float wt[4] = {hh * hw * mk, hh * q.lw * mk, q.lh * hw * mk, q.lh * q.lw * mk};
    const bool ok[4] = {q.v00, q.v01, q.v10, q.v11};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!ok[c]) continue;
      const int yy = q.h0 + (c >> 1), xx = q.w0 + (c & 1);
      const int ry = yy - y0, rx = xx - x0;
      if ((unsigned)ry < (unsigned)DCN_WIN && (unsigned)rx < (unsigned)DCN_WIN) {
        int* dst = win + (ry * DCN_WIN + rx) * DCN_CS + l4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, __float2int_rn(wt[c] * bf2f(d[e]) * scale));   // ds_add_u32
      } else {
        float* dst = far + (((long)b * p.H + yy) * p.W + xx) * p.C + cs * DCN_CS + l4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, wt[c] * bf2f(d[e]));
      }
    }
  }
  __syncthreads();
  int* out = tiles + (long)tile * DCN_WIN * DCN_WIN * p.C + cs * DCN_CS;
  for (int i = threadIdx.x; i < DCN_WIN * DCN_WIN * DCN_CS / 4; i += 256) {
    const int pix = i >> 2, c4 = i & 3;
    *reinterpret_cast<int4*>(out + (long)pix * p.C + c4 * 4) = reinterpret_cast<const int4*>(win)[i];
  }
}

// dx[b,y,x,c] = far + (sum of the windows covering (y,x)) / scale; one thread = 8 channels of one pixel
__global__ __launch_bounds__(256) void dcn_dx_sum_kernel(DcnP p, DcnT g, const int* tiles, const float* far, bf16* dx, long total, const unsigned* maxbits) {
  const int C8 = p.C >> 3, span = g.T * p.stride;
  const float inv = 1.f / dcn_fixed_scale(*maxbits);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    const long pix = idx / C8;
    const int x = (int)(pix % p.W), y = (int)((pix / p.W) % p.H), b = (int)(pix / ((long)p.W * p.H));
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ay = y + p.pad + DCN_HALO, ax = x + p.pad + DCN_HALO;          // window-relative coordinate + tile origin
    const int ty_hi = min(ay / span, g.tiles_y - 1), tx_hi = min(ax / span, g.tiles_x - 1);
    for (int ty = ty_hi; ty >= 0 && ay - ty * span < DCN_WIN; --ty)
      for (int tx = tx_hi; tx >= 0 && ax - tx * span < DCN_WIN; --tx) {
        const long tile = ((long)b * g.tiles_y + ty) * g.tiles_x + tx;
        const int* src = tiles + (tile * DCN_WIN * DCN_WIN + (ay - ty * span) * DCN_WIN + (ax - tx * span)) * p.C + c8 * 8;
        const int4 s0 = *reinterpret_cast<const int4*>(src), s1 = *reinterpret_cast<const int4*>(src + 4);
        acc[0] += s0.x; acc[1] += s0.y; acc[2] += s0.z; acc[3] += s0.w; acc[4] += s1.x; acc[5] += s1.y; acc[6] += s1.z; acc[7] += s1.w;
      }
    const float4 f0 = *reinterpret_cast<const float4*>(far + pix * p.C + c8 * 8), f1 = *reinterpret_cast<const float4*>(far + pix * p.C + c8 * 8 + 4);
    const float fr[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(fr[e] + (float)acc[e] * inv);
    *reinterpret_cast<bf16x8*>(dx + pix * p.C + c8 * 8) = o;
  }
}

int dcn_tiling(const DcnP& p, DcnT* g) {
  if (p.kh != 3 || p.kw != 3 || (p.stride != 1 && p.stride != 2) || (p.C % DCN_CS)) return FIBER_EINVAL;
  g->T = p.stride == 1 ? 16 : 8;
  g->tiles_y = (p.Ho + g->T - 1) / g->T;
  g->tiles_x = (p.Wo + g->T - 1) / g->T;
  return FIBER_OK;
}

int dcn_check(const DcnP& p) {
  if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.C <= 0 || (p.C & 7) || p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0) return FIBER_EINVAL;
  if (p.Ho != (p.H + 2 * p.pad - p.kh) / p.stride + 1 || p.Wo != (p.W + 2 * p.pad - p.kw) / p.stride + 1 || p.Ho <= 0 || p.Wo <= 0) return FIBER_EINVAL;
  return FIBER_OK;
}

}  // namespace

// cols [B*Ho*Wo, kh*kw*C] bf16 <- x [B,H,W,C] bf16 sampled at the (optionally deformed, optionally modulated) taps.
// offset [M, 2*kh*kw] fp32 ((dy, dx) per tap, tap = i*kw + j: the channel order of deform_conv_kernel_cuda.cu:607-608) or NULL;
// mask [M, kh*kw] fp32 or NULL.  C % 8 == 0.  With both NULL this is an ordinary im2col (the offset / non-deformable convs).
extern "C" int fiber_dcn_gather_bf16(const void* x, const float* offset, const float* mask, void* cols, int B, int H, int W, int C,
                                     int Ho, int Wo, int kh, int kw, int stride, int pad, hipStream_t stream) {
  DcnP p{(const bf16*)x, offset, mask, B, H, W, C, Ho, Wo, kh, kw, stride, pad};
  if (int rc = dcn_check(p)) return rc;
  const long total = (long)B * Ho * Wo * kh * kw * (C >> 3);
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(dcn_gather_kernel, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(256), 0, stream, p, (bf16*)cols, total);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Gradients of the gather: dx [B,H,W,C] fp32 is ACCUMULATED into (zero it first; NULL skips it), doffset [M, 2*taps] and
// dmask [M, taps] fp32 are written (either may be NULL).  dcols [M, taps*C] bf16.
extern "C" int fiber_dcn_scatter_bf16(const void* dcols, const void* x, const float* offset, const float* mask, float* dx,
                                      float* doffset, float* dmask, int B, int H, int W, int C, int Ho, int Wo, int kh, int kw,
                                      int stride, int pad, hipStream_t stream) {
  DcnP p{(const bf16*)x, offset, mask, B, H, W, C, Ho, Wo, kh, kw, stride, pad};
  if (int rc = dcn_check(p)) return rc;
  const long groups = (long)B * Ho * Wo * kh * kw;
  const int C8 = C >> 3;
#define FIBER_DCN_SCATTER(G)                                                                                                  \
  do {                                                                                                                        \
    const long blocks = (groups * G + 255) / 256;                                                                             \
    hipLaunchKernelGGL(dcn_scatter_kernel<G>, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(256), 0, stream, p, \
                       (const bf16*)dcols, dx, doffset, dmask, groups);                                                       \
  } while (0)
  if (C8 >= 64) FIBER_DCN_SCATTER(64);
  else if (C8 >= 32) FIBER_DCN_SCATTER(32);
  else if (C8 >= 16) FIBER_DCN_SCATTER(16);
  else if (C8 >= 8) FIBER_DCN_SCATTER(8);
  else if (C8 >= 4) FIBER_DCN_SCATTER(4);
  else if (C8 >= 2) FIBER_DCN_SCATTER(2);
  else FIBER_DCN_SCATTER(1);
#undef FIBER_DCN_SCATTER
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Input gradient of the gather, atomics-free form (3x3 taps, pad 1, stride 1 or 2, C % 16 == 0): dx bf16 [B,H,W,C] is WRITTEN.
// workspace: fiber_dcn_dx_workspace() 4-byte words = per-tile windows + one fp32 map for corners that leave their window + the
// scale word; everything from the map on (the LAST B*H*W*C + 4 words) must be zero on entry.
extern "C" long fiber_dcn_dx_workspace(int B, int H, int W, int C, int Ho, int Wo, int stride) {
  DcnP p{nullptr, nullptr, nullptr, B, H, W, C, Ho, Wo, 3, 3, stride, 1};
  DcnT g;
  if (dcn_tiling(p, &g)) return -1;
  return (long)B * g.tiles_y * g.tiles_x * DCN_WIN * DCN_WIN * C + (long)B * H * W * C + 4;
}

extern "C" int fiber_dcn_dx_bf16(const void* dcols, const float* offset, const float* mask, void* dx, float* workspace, int B, int H,
                                 int W, int C, int Ho, int Wo, int kh, int kw, int stride, int pad, hipStream_t stream) {
  DcnP p{nullptr, offset, mask, B, H, W, C, Ho, Wo, kh, kw, stride, pad};
  if (int rc = dcn_check(p)) return rc;
  DcnT g;
  if (int rc = dcn_tiling(p, &g)) return rc;
  if (pad != 1) return FIBER_EINVAL;
  const long ntile = (long)B * g.tiles_y * g.tiles_x;
  int* tiles = reinterpret_cast<int*>(workspace);
  float* far = workspace + ntile * DCN_WIN * DCN_WIN * C;
  unsigned* maxbits = reinterpret_cast<unsigned*>(far + (long)B * H * W * C);
  const long n8 = (long)B * Ho * Wo * kh * kw * C / 8;
  hipLaunchKernelGGL(absmax_bf16_kernel, dim3((unsigned)(n8 / 256 / 8 + 1 < 2048 ? n8 / 256 / 8 + 1 : 2048)), dim3(256), 0, stream, (const bf16*)dcols, n8, maxbits);
  hipLaunchKernelGGL(dcn_dx_tile_kernel, dim3((unsigned)(ntile * (C / DCN_CS))), dim3(256), 0, stream, p, g, (const bf16*)dcols, tiles, far, maxbits);
  const long total = (long)B * H * W * (C >> 3);
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(dcn_dx_sum_kernel, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(256), 0, stream, p, g, tiles, far, (bf16*)dx, total, maxbits);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
