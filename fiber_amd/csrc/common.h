// Shared device helpers for the fiber_hip kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FIBER_OK 0
#define FIBER_EINVAL 1
#define FIBER_ELAUNCH 2

#define FIBER_CHECK_LAUNCH()                          \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return FIBER_ELAUNCH;      \
  } while (0)

// hipFuncSetAttribute is per-DEVICE state: a launcher's "attributes already set" flag is indexed by the current device.
// Returns true exactly once per device (and always when the device cannot be identified, which only costs the repeated calls).
inline bool fiber_first_on_device(bool (&seen)[16]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;
  if (seen[dev]) return false;
  seen[dev] = true;
  return true;
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Reductions over the four lanes {l, l^16, l^32, l^48} that share one MFMA-C row: two gfx950 row swaps (v_permlane16_swap / v_permlane32_swap,
// plain VALU) instead of two ds_bpermute round trips through the LDS pipe.  With both operands the same register the swap returns
// (mine-or-partner, partner-or-mine) in every lane, so combining the two halves is the xor-shuffle reduction, bit for bit (max and + commute).
#ifdef FIBER_ROWS4_SHFL   // A/B build only (tools/): the ds_bpermute form
__device__ __forceinline__ float rows4_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float rows4_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
#else
__device__ __forceinline__ float rows4_max(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __builtin_bit_cast(unsigned, fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1])));
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}
#endif

// Exact (erf) GELU, cheap enough for a GEMM epilogue.  Phi(x) = 0.5 (1 + erf(x / sqrt 2)) from Abramowitz-Stegun 7.1.28,
//   erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16  (z >= 0, |err| <= 3e-7),
// with z = |x| / sqrt 2 folded into the coefficients and the 0.5 folded in as a 2^(1/16) scale of the polynomial, so that
//   q = 1 / P(|x|)^16 = 0.5 erfc(|x| / sqrt 2),   Phi(x) = x >= 0 ? 1 - q : q.
// Six FMAs, four squarings, ONE half-rate instruction (v_rcp_f32) and no v_exp, no compare / select: ~10 VALU issue slots per element with
// the FMA/multiply chain on v_pk_*_f32 pairs, against ~25 for the 7.1.26 form (rcp + exp) that was VALU-bound in the fc1
// epilogue (+420 us on a 620-GFLOP GEMM).  Measured |gelu - reference| <= 8e-7 over [-12, 12] (bf16 resolves 4e-3 relative).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// q = 0.5 erfc(ax / sqrt 2) for ax >= 0
__device__ __forceinline__ f32x2 half_erfc2(f32x2 ax) {
  f32x2 p = ax * 5.621299351332709e-06f + 5.105520540382713e-05f;
  p = p * ax + 3.9686136005911976e-05f;
  p = p * ax + 0.0034227389842271805f;
  p = p * ax + 0.02207699790596962f;
  p = p * ax + 0.052075158804655075f;
  p = p * ax + 1.0442737340927124f;
  p = p * p; p = p * p; p = p * p; p = p * p;            // overflow -> inf -> q = 0 (|x| > ~25)
  return f32x2{__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
}
// Phi(x) = 0.5 + copysign(0.5 - q, x): two packed adds and a v_bfi per element instead of subtract + compare + select
__device__ __forceinline__ f32x2 gelu_cdf2(f32x2 x) {
  const f32x2 d = 0.5f - half_erfc2(f32x2{fabsf(x.x), fabsf(x.y)});
  return f32x2{__builtin_copysignf(d.x, x.x), __builtin_copysignf(d.y, x.y)} + 0.5f;
}
// gelu(x) = x Phi(x) = max(x, 0) - |x| q = 0.5 (x + |x|) - |x| q: no select at all (packed add, multiply, fma)
__device__ __forceinline__ f32x2 gelu2_exact(f32x2 x) {
  const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
  return __builtin_elementwise_fma(x + ax, f32x2{0.5f, 0.5f}, -(ax * half_erfc2(ax)));
}
// d/dx gelu(x) = Phi(x) + x phi(x)
__device__ __forceinline__ f32x2 gelu_grad2_exact(f32x2 x) {
  const f32x2 t = x * x * -0.72134752044448170368f;      // -0.5 x^2 log2(e)
  const f32x2 pdf = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} * 0.3989422804014327f;
  return gelu_cdf2(x) + x * pdf;
}
// Backward form without a transcendental: the quarter-rate v_rcp + v_exp of gelu_grad2_exact are 12 of its ~28 issue slots per
// pair and the fused (dY W2^T) gelu'(H) epilogue is VALU-bound.  gelu'(t) - 1/2 = t Q(t^2), an odd minimax polynomial on |t| <= 4.5
// with the argument clamped there and the end value scaled to exactly 1/2, so gelu' = 1 for x >= 4.5 and 0 for x <= -4.5.  fp32
// Horner on packed pairs: max |gelu' - reference| 1.7e-4 over [-8, 8] (the product dy * gelu' is rounded to bf16: 4e-3), 14 slots
// per pair; s2 fused backward GEMM 1311 -> 1174 us.  The FORWARD stays on the exact form: the same trick there (1.6e-4) bought
// nothing (that epilogue is bound by its two 128-KB stores) and moved cancellation-dominated gradients past their test bounds.
__device__ __forceinline__ f32x2 clamp45(f32x2 x) {
  return f32x2{__builtin_amdgcn_fmed3f(x.x, -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x.y, -4.5f, 4.5f)};
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) { return gelu2_exact(x); }   // forward: exact form (activations feed the loss)
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
  const f32x2 t = clamp45(x), u = t * t;
  f32x2 p = u * -2.2107319200e-11f + 2.5212396046e-09f;
  p = p * u + -1.2680140542e-07f;
  p = p * u + 3.7218184976e-06f;
  p = p * u + -7.1219708718e-05f;
  p = p * u + 9.4051189985e-04f;
  p = p * u + -8.8155761709e-03f;
  p = p * u + 5.8607425057e-02f;
  p = p * u + -2.6491721243e-01f;
  p = p * u + 7.9760069086e-01f;
  return __builtin_elementwise_fma(t, p, f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu2(f32x2{x, x}).x; }
__device__ __forceinline__ float gelu_erf_grad(float x) { return gelu_grad2(f32x2{x, x}).x; }
// eight bf16 at a time (the unit of every 16-byte epilogue / element-wise access)
__device__ __forceinline__ bf16x8 gelu8(bf16x8 v, float scale) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 g = gelu2(f32x2{bf2f(v[e]), bf2f(v[e + 1])}) * scale;
    o[e] = f2bf(g.x); o[e + 1] = f2bf(g.y);
  }
  return o;
}
// dy * gelu'(h)
__device__ __forceinline__ bf16x8 gelu_grad_mul8(bf16x8 dy, bf16x8 h, float scale = 1.f) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 g = gelu_grad2(f32x2{bf2f(h[e]), bf2f(h[e + 1])}) * f32x2{bf2f(dy[e]), bf2f(dy[e + 1])} * scale;
    o[e] = f2bf(g.x); o[e + 1] = f2bf(g.y);
  }
  return o;
}

// Counter-based RNG for dropout: one 32-bit hash per (seed, element index).  Forward and backward
// recompute the same keep-mask from (seed, index), so no mask tensor is ever stored.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
// (hash_u32 keys the MLM collate and the DropPath draw -- both pinned by the oracle, oracle/image_ref.py; one draw per token / sample.)

// Attention-probability dropout: keep decision of score (row, key) -- row = ((sample * heads + head) * Lq + query) -- of a launch keyed by
// seed32.  Three 32-bit multiplies per score (two when the row or the key term is lane-constant and hoisted) instead of the three 64-bit
// multiplies of hash_u32 (twelve quarter-rate VALU multiplies): the t2i forward + backward lost 19 % to its dropout (tools/option_ablation.py).
// Forward and both backward passes call this with the same (seed32, row, key); the element-wise hidden dropout and the MLM collate keep hash_u32.
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_seed32(uint64_t seed) { return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u); }
// Element-wise (hidden-state) dropout: keep element idx with probability 1 - p, thresh = p * 2^32.  Round 6: the same 32-bit mix as the attention
// dropout -- hash_u32's three 64-bit multiplies are twelve quarter-rate VALU multiplies per ELEMENT, which made the text stack's residual adds
// (stream_add: 31-MB tensors) VALU-bound at 58 us a call.  A vector's elements share drop_base(seed, index of its first element): one multiply per
// vector, two (fmix32) per element.  drop_keep_e(drop_base(seed, b), e, t) == drop_keep(seed, b + e, t) for vectors that do not straddle 2^32.
__device__ __forceinline__ uint32_t drop_base(uint64_t seed, uint64_t base) {
  return drop_seed32(seed) + (uint32_t)base * 0x9E3779B1u + (uint32_t)(base >> 32) * 0x7FEB352Du;
}
__device__ __forceinline__ bool drop_keep_e(uint32_t h0, int e, uint32_t thresh) { return fmix32(h0 + (uint32_t)e * 0x9E3779B1u) >= thresh; }
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh) { return fmix32(drop_base(seed, idx)) >= thresh; }
__device__ __forceinline__ bool drop_keep_rk(uint32_t seed32, uint32_t row, uint32_t key, uint32_t thresh) {
  return fmix32(seed32 + row * 0x9E3779B1u + key * 0x7FEB352Du) >= thresh;
}

// One LDS-DMA instruction (64 lanes x 16 B -> 1 KB at `lds_wave_base`, lane-linear) issued from inline asm.  The builtin
// form tells the compiler that LDS is being written behind the vmcnt counter, and its waitcnt pass then guards LDS reads
// with vmcnt(0) wherever it loses count (after branches, around other VMEM traffic): in the persistent kernel that put a
// full drain -- including the previous round's output stores -- in front of every staging ds_read.  All ordering of these
// transfers is done by hand (counted s_waitcnt + s_barrier), so the compiler does not need to know.  M0 is written and
// consumed inside the block; nothing else in these kernels uses it.
// Address = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit VGPR address arithmetic.
__device__ __forceinline__ void lds_dma16(const void* base, unsigned lane_byte_off, void* lds_wave_base) {
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m), "v"(lane_byte_off), "s"(base) : "memory");
}
// The same with a 64-bit per-lane source address (used where lanes of one instruction read from unrelated buffers).
__device__ __forceinline__ void lds_dma16_v(const void* lane_src, void* lds_wave_base) {
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m), "v"(lane_src) : "memory");
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
