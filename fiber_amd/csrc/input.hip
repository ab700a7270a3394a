// On-device input pipeline for the fused-backbone path (gfx950): what the reference does per sample on CPU dataloader
// workers, moved behind the H2D copy so that the loader only has to decode and ship raw bytes.
//
//   1. image transform  -- coarse_grained/fiber/transforms/transform.py:10-17 `albef_transform`:
//        torchvision Resize((S,S), interpolation=BICUBIC) on a PIL RGB image -> ToTensor -> Normalize(mean, std)
//      i.e. Pillow's ImagingResample (third party, src/libImaging/Resample.c; stable since Pillow 3.x; not in /root/reference):
//      separable two-pass CONVOLUTION resize (horizontal, then vertical) with an anti-aliasing bicubic kernel (a = -0.5,
//      support 2 x max(scale, 1)), per-output-pixel coefficient windows normalised in double precision and quantised to 22-bit
//      fixed point, 8-bit rounding after EACH pass.  Restated here operation by operation (integer and double arithmetic
//      in Pillow's order, no fused multiply-add) so that the result is bit-identical to PIL's, then x/255, (x - mean)/std in
//      fp32 as ToTensor / Normalize do.  Oracle: oracle/image_ref.py (numpy), pinned against PIL itself.
//   2. MLM masking      -- datamodule_base.py:52 `DataCollatorForLanguageModeling(mlm_probability=0.15)` (transformers 4.6.0
//      `mask_tokens`): 15 % of the non-special tokens become labels; of those 80 % -> <mask>, 10 % -> a random token, 10 % keep.
//      The collator's torch.bernoulli / randint streams cannot be reproduced off the CPU generator; the draws here come from
//      the counter-based hash of common.h (seed, token index), so a batch is a pure function of (ids, seed).
//
// Layouts: source images are uint8 HWC (as PIL / a JPEG decoder leaves them), ragged sizes inside a batch, addressed through
// a descriptor table; the output is the [B, 3, S, S] fp32 tensor FIBERTransformerSS.infer consumes.  Byte / integer work,
// HBM-bound: one thread per output pixel (3 channels), taps read through L1/L2 (neighbouring threads share them), the
// intermediate is the uint8 image Pillow itself materialises between its passes.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Pillow: 22-bit fixed-point coefficients for 8-bit channels

struct ImgDesc {                                // one per image, built by the host (fiber_amd/data.py)
  long long src;                                // device pointer: uint8 [H][W][3] (row stride = src_stride bytes)
  int H, W, src_stride;
  int ksize_h, ksize_v;                         // coefficient window lengths (fiber_resample_ksize)
  int tmp_off;                                  // byte offset of this image's [H][S][3] intermediate in `tmp`
  int coef_h_off, coef_v_off;                   // int offsets of its [S][2 + ksize] tables (bounds + coefficients) in `coef`
};

#pragma clang fp contract(off)
__device__ double bicubic_filter(double x) {
  // Pillow Resample.c bicubic_filter, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output coordinate xx of an axis resized in_size -> out_size.
// table row: [xmin, xmax, k_0 .. k_{ksize-1}]
__device__ void coeff_row(int in_size, int out_size, int ksize, int xx, int* row) {
#pragma clang fp contract(off)
  if (in_size == out_size) {                     // Pillow skips a pass whose size does not change (ImagingResample need_*)
    row[0] = xx; row[1] = 1; row[2] = 1 << PRECISION_BITS;
    for (int x = 1; x < ksize; ++x) row[2 + x] = 0;
    return;
  }
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const double center = (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
  row[0] = xmin;
  row[1] = xmax;
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      w = bicubic_filter((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
    }
    row[2 + x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
  }
}

__global__ __launch_bounds__(128) void resample_coeffs_kernel(const ImgDesc* descs, int* coef, int S) {
  const ImgDesc d = descs[blockIdx.y];
  const int t = blockIdx.x * 128 + threadIdx.x;
  if (t < S) coeff_row(d.W, S, d.ksize_h, t, coef + d.coef_h_off + (size_t)t * (2 + d.ksize_h));
  else if (t < 2 * S) coeff_row(d.H, S, d.ksize_v, t - S, coef + d.coef_v_off + (size_t)(t - S) * (2 + d.ksize_v));
}

__device__ __forceinline__ int clip8(int ss) {
  const int v = ss >> PRECISION_BITS;            // arithmetic shift, as Pillow's lookup index
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// pass 1: tmp[y][xx][c] = clip8(2^21 + sum_x src[y][xmin + x][c] * k[x])
__global__ __launch_bounds__(256) void resample_h_kernel(const ImgDesc* descs, const int* coef, unsigned char* tmp, int S) {
  const ImgDesc d = descs[blockIdx.z];
  const int xx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (y >= d.H || xx >= S) return;
  const int* row = coef + d.coef_h_off + (size_t)xx * (2 + d.ksize_h);
  const int xmin = row[0], xmax = row[1];
  const unsigned char* sp = reinterpret_cast<const unsigned char*>(d.src) + (size_t)y * d.src_stride + (size_t)xmin * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < xmax; ++x) {
    const int k = row[2 + x];
    s0 += sp[3 * x] * k; s1 += sp[3 * x + 1] * k; s2 += sp[3 * x + 2] * k;
  }
  unsigned char* o = tmp + d.tmp_off + ((size_t)y * S + xx) * 3;
  o[0] = (unsigned char)clip8(s0); o[1] = (unsigned char)clip8(s1); o[2] = (unsigned char)clip8(s2);
}

// pass 2 + ToTensor + Normalize: out[b][c][yy][xx] = (clip8(...) / 255 - mean[c]) / std[c]   (fp32, each op rounded)
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const ImgDesc* descs, const int* coef, const unsigned char* tmp,
                                                              float* out, int S, float m0, float m1, float m2, float d0, float d1,
                                                              float d2) {
#pragma clang fp contract(off)
  const ImgDesc d = descs[blockIdx.z];
  const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
  if (xx >= S) return;
  const int* row = coef + d.coef_v_off + (size_t)yy * (2 + d.ksize_v);
  const int ymin = row[0], ymax = row[1];
  const unsigned char* tp = tmp + d.tmp_off + ((size_t)ymin * S + xx) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < ymax; ++y) {
    const int k = row[2 + y];
    const unsigned char* p = tp + (size_t)y * S * 3;
    s0 += p[0] * k; s1 += p[1] * k; s2 += p[2] * k;
  }
  float* o = out + ((size_t)blockIdx.z * 3 * S + yy) * S + xx;
  const size_t plane = (size_t)S * S;
  o[0] = ((float)clip8(s0) / 255.0f - m0) / d0;
  o[plane] = ((float)clip8(s1) / 255.0f - m1) / d1;
  o[2 * plane] = ((float)clip8(s2) / 255.0f - m2) / d2;
}

// MLM masking: one thread per token.  Draws: u_k = hash_u32(seed, 4 * index + k), k = 0 (select), 1 (replace with <mask>),
// 2 (replace with a random token), 3 (which token).  Thresholds are floor(p * 2^32).
__global__ __launch_bounds__(256) void mlm_mask_kernel(const long long* ids, long long* ids_mlm, long long* labels, long n,
                                                       unsigned long long seed, unsigned p_select, int mask_id, int vocab,
                                                       int special_lo, int special_hi) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long id = ids[i];
  const bool special = id >= special_lo && id <= special_hi;       // RoBERTa: <s> = 0, <pad> = 1, </s> = 2
  const bool masked = !special && hash_u32(seed, 4ull * i) < p_select;
  long long out = id;
  if (masked) {
    if (hash_u32(seed, 4ull * i + 1) < 3435973836u) out = mask_id;                       // 0.8 * 2^32
    else if (hash_u32(seed, 4ull * i + 2) < 2147483648u) out = hash_u32(seed, 4ull * i + 3) % (unsigned)vocab;   // 0.5
  }
  ids_mlm[i] = out;
  labels[i] = masked ? id : -100;
}

}  // namespace

// Coefficient window length Pillow allocates for an axis resized in_size -> out_size with the bicubic filter.
extern "C" int fiber_resample_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  int c = (int)support;
  if ((double)c < support) ++c;                  // ceil
  return c * 2 + 1;
}

// descs: device array of n ImgDesc (10 ints + 1 int64 each, see fiber_amd/data.py); coef / tmp: workspaces laid out by the
// host; out: fp32 [n, 3, S, S].  max_h: tallest source image of the batch (grid extent).  mean / std: 3 floats each (host).
extern "C" int fiber_resize_bicubic_norm_u8(const void* descs, int n, int* coef, void* tmp, float* out, int S, int max_h,
                                            const float* mean, const float* std, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if (S <= 0 || max_h <= 0 || !mean || !std) return FIBER_EINVAL;
  const ImgDesc* d = reinterpret_cast<const ImgDesc*>(descs);
  hipLaunchKernelGGL(resample_coeffs_kernel, dim3(cdiv(2 * S, 128), n), dim3(128), 0, stream, d, coef, S);
  hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv(S, 256), max_h, n), dim3(256), 0, stream, d, coef,
                     reinterpret_cast<unsigned char*>(tmp), S);
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3(cdiv(S, 256), S, n), dim3(256), 0, stream, d, coef,
                     reinterpret_cast<const unsigned char*>(tmp), out, S, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// ids [n] int64 -> ids_mlm, labels (int64).  p_select = floor(mlm_probability * 2^32).
extern "C" int fiber_mlm_mask_i64(const long long* ids, long long* ids_mlm, long long* labels, long n, unsigned long long seed,
                                  unsigned p_select, int mask_id, int vocab, int special_lo, int special_hi, hipStream_t stream) {
  if (n <= 0) return FIBER_OK;
  if (vocab <= 0) return FIBER_EINVAL;
  hipLaunchKernelGGL(mlm_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ids, ids_mlm, labels, n, seed,
                     p_select, mask_id, vocab, special_lo, special_hi);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
