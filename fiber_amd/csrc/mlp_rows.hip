// LayerNorm + Mlp + residual of a Swin block as ONE kernel per direction, for the narrow stages (C = 128 / 256: Swin-B stages 0-1,
// reference swin_transformer.py:391  x = x + drop_path(mlp(norm2(x))), timm Mlp = fc2(gelu(fc1(.)))), gfx950 / CDNA4.
//
// Why: as separate kernels these stages are HBM-bound (SURVEY.md section 7 "stages 0-1 are HBM-bound unless fused": the 4C-wide hidden
// tensor is written twice and read back by fc2, the LayerNorm output is a pass of its own).  Per block at 512 images the separate
// forward moves 20.5 GB at C = 128 (LN 2.4, fc1 10.9, fc2 7.2); this kernel reads x and writes y: 2.4 GB.
//
// How: a wave owns 32 token rows for the whole chain -- the hidden activations never leave its registers.
//   * all products are formed TRANSPOSED, weights as the MFMA A operand (from LDS), activations as the B operand (registers):
//       H^T[hidden, row]  = W1'[hidden, :] . xhat^T          (32x32x16 MFMA, K = C)
//       Y^T[chan, row]   += W2[chan, hidden-chunk] . G^T     (K = the 64 hidden units of a chunk)
//     The C layout of H^T puts the 16 hidden units {8b + 4h + c} of ONE row in a lane (h = lane >> 5): exactly what a B operand
//     needs, up to the order of the contraction index -- a sum does not care.  So the hidden index of the second product runs in the
//     order "bits 2 and 3 swapped" (position 16a + 8h + 4q + c holds logical index 16a + 8q + 4h + c) and the host stores the weight
//     copies whose K dimension is the hidden one with their columns in that order (ops.py _fa): gelu(H) goes from accumulator registers
//     straight into the next MFMA.
//   * LayerNorm's affine part is folded into fc1 on the host: W1' = W1 diag(gamma), b1' = b1 + W1 beta (fp32 product, one bf16
//     rounding of the weight copy).  The kernel normalises only: xhat = (x - mean) rstd.  The backward returns dW1' and the autograd
//     wrapper unfolds it (dW1 = dW1' diag(gamma) + db1 (x) beta, dgamma = colsum(dW1' * W1), dbeta = W1^T db1).
//   * the weights stream through LDS in 64x64 "units" ([64 rows][64 k] bf16, the NT GEMM's swizzled image, LDS-DMA), one phase
//     (= one product of one 64-wide hidden chunk) ahead of their use, behind one workgroup barrier per phase.
//   * global memory is only touched in whole rows, 16 bytes per lane: a wave's 32 x C tile comes in by LDS-DMA into a wave-private
//     region (16-byte pieces XOR-swizzled by row so that the row-per-lane fragment reads are conflict free) and results leave through
//     the same region (8-byte C-layout pieces in, 16-byte row pieces out).  The first version loaded / stored row-per-lane 8-byte
//     pieces directly: 2.8 ms per call at stage 0, bound by the issue of 64-line memory instructions.
//   * the forward optionally stores G = gelu(H) (bf16) for the backward: the kernel is bound by its GELU arithmetic (PMC: VALU busy 46 %,
//     MFMA busy 25 %, HBM nearly idle), so the store rides along; H itself is never stored.
//   * backward (ln_mlp_bwd_kernel) RECOMPUTES H from x (the matrix pipe is idle in these stages), forms dG^T = W2^T . dY^T, dH = s dG
//     gelu'(H), dXhat^T += W1'^T . dH^T, applies the LayerNorm backward and the residual gradient in registers, and writes dx.  It also
//     writes dH and xhat (bf16) -- with the forward's G the operands of the two weight-gradient GEMMs (gemm_tn.hip), which stay separate
//     kernels: their contraction runs over ALL rows, a row-owning wave cannot hold a [4C, C] accumulator.
//
// Measured and not shipped (profiles/r06_summary.md section 3): the two wave groups of the workgroup half a chunk apart (one wave of a SIMD in its MFMA
// block while the other is in its GELU block: 2279 us against 2180 at stage 0), and a chunk loop skewed by one chunk so that the MFMAs of product 1 of
// chunk j + 1 are issued between slices of GELU(chunk j) (tools/experiments/mlp_rows_skewed_fwd.hip.inc: 3135 us against 2662).  The counters say the
// vector pipe is busy 46 % and the matrix pipe 25 % of the forward kernel at C = 128, one after the other.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

struct MlpP {
  const bf16* x; const bf16* dy;
  const bf16* w1p;   // W1' = W1 diag(gamma)  [4C][C]
  const bf16* w2p;   // FA(W2)    [C][4C]      (forward)
  const bf16* w2tp;  // W2^T      [4C][C]      (backward)
  const bf16* w1tp;  // FA(W1'^T) [C][4C]      (backward)
  const float* b1p;  // [4C]
  const float* b2;   // [C]
  const float* rowscale;
  bf16* y; bf16* dx; bf16* dh; bf16* g; bf16* xhat;
  int M, rps;
  float eps;
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

constexpr int UNIT = 64 * 64;                              // elements of one LDS unit (8 KB)

// one 64 x 64 block of a row-major [R][ld] bf16 matrix -> LDS unit (swizzled like gemm.hip's tiles); NW waves, 8 / NW instructions each
template <int NW>
__device__ __forceinline__ void dma_unit(bf16* unit, const bf16* blk, int ld, int wave, int lane) {
#ifdef MLP_PROBE_NODMA
  return;
#endif
  constexpr int RW = 64 / NW;                              // unit rows per wave
#pragma unroll
  for (int q = 0; q < RW / 8; ++q) {
    const int row = wave * RW + q * 8 + (lane >> 3);
    const bf16* src = blk + (size_t)row * ld + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(unit + (wave * RW + q * 8) * 64), 16, 0, 0);
  }
}

__device__ __forceinline__ void wait_all() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void phase_sync() {
#ifdef MLP_PROBE_NOSYNC
  return;
#endif
  wait_all();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// A fragment (weights): rows 32 i + (lane & 31) of a unit, k-step ks (16 k) -> chunks 2 ks + (lane >> 5)
__device__ __forceinline__ bf16x8 afrag(const bf16* unit, int i, int ks, int n, int h) {
  return *reinterpret_cast<const bf16x8*>(unit + swz(i * 32 + n, ks * 2 + h));
}

// ---- wave-private 32 x C tile region: row r at r * C elements, its 16-byte piece c stored at slot c ^ (r & 15) --------------------------------
// rows of the wave's strip by LDS-DMA (whole rows per instruction; rows past M re-read row M - 1)
template <int C>
__device__ __forceinline__ void dma_rows(bf16* reg, const bf16* g, int row0, int M, int lane) {
  constexpr int PPR = C / 8, RPI = 64 / PPR;
#pragma unroll
  for (int q = 0; q < 32 / RPI; ++q) {
    const int rr = q * RPI + lane / PPR, c = (lane % PPR) ^ (rr & 15);
    const int grow = min(row0 + rr, M - 1);
    __builtin_amdgcn_global_load_lds((glb_ptr)(g + (size_t)grow * C + c * 8), (lds_ptr)(reg + q * 512), 16, 0, 0);
  }
}
// the lane's row as B fragments: k-step t = channels 16 t + 8 h .. + 7 (one 16-byte piece)
template <int C>
__device__ __forceinline__ void read_frags(const bf16* reg, int n, int h, bf16x8 (&f)[C / 16]) {
#pragma unroll
  for (int t = 0; t < C / 16; ++t) f[t] = *reinterpret_cast<const bf16x8*>(reg + n * C + (((2 * t + h) ^ (n & 15)) << 3));
}
template <int C>
__device__ __forceinline__ void write_frags(bf16* reg, int n, int h, const bf16x8 (&f)[C / 16]) {
#pragma unroll
  for (int t = 0; t < C / 16; ++t) *reinterpret_cast<bf16x8*>(reg + n * C + (((2 * t + h) ^ (n & 15)) << 3)) = f[t];
}
// 8-byte piece of the C layout: channels 32 mt + 8 b + 4 h .. + 3 of row n
template <int C>
__device__ __forceinline__ bf16* cpiece(bf16* reg, int n, int h, int mt, int b) {
  return reg + n * C + (((4 * mt + b) ^ (n & 15)) << 3) + h * 4;
}
// the region's rows to memory, 16 bytes per lane, whole rows per instruction
template <int C>
__device__ __forceinline__ void store_rows(const bf16* reg, bf16* g, int row0, int M, int lane) {
  constexpr int PPR = C / 8, RPI = 64 / PPR;
#pragma unroll
  for (int q = 0; q < 32 / RPI; ++q) {
    const int rr = q * RPI + lane / PPR, c = (lane % PPR) ^ (rr & 15);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(reg + q * 512 + lane * 8);
    if (row0 + rr < M) *reinterpret_cast<bf16x8*>(g + (size_t)(row0 + rr) * C + c * 8) = v;
  }
}

__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32); }

// xhat = (x - mean) * rstd of the lane's row (two lanes share a row), two-pass statistics as norm.hip
template <int KT>
__device__ __forceinline__ void normalise(const bf16x8 (&xr)[KT], bf16x8 (&xh)[KT], float eps, float& rstd) {
  constexpr int C = KT * 16;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f(xr[t][e]);
  const float mean = pair_sum(s) * (1.f / C);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = bf2f(xr[t][e]) - mean; q += d * d; }
  rstd = rsqrtf(pair_sum(q) * (1.f / C) + eps);
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) xh[t][e] = f2bf((bf2f(xr[t][e]) - mean) * rstd);
}

// B fragments (lane: row n, 16-byte pieces 2 t + h) -> the C layout's 8-byte pieces (lane: half h of EVERY piece): the two lanes of a row
// trade the halves they do not need.  lo[t] = channels 16 t + 4 h .. + 3, hi[t] = channels 16 t + 8 + 4 h .. + 3.
template <int KT>
__device__ __forceinline__ void frags_to_cpieces(const bf16x8 (&f)[KT], bf16x4 (&lo)[KT], bf16x4 (&hi)[KT]) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const bf16x4 fa = {f[t][0], f[t][1], f[t][2], f[t][3]}, fb = {f[t][4], f[t][5], f[t][6], f[t][7]};
    u32x2 a = __builtin_bit_cast(u32x2, fa), b = __builtin_bit_cast(u32x2, fb);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      // swap(a, b): a[upper lanes] <-> b[lower lanes]; afterwards a = half h of the even piece, b = half h of the odd piece
      const auto r = __builtin_amdgcn_permlane32_swap(a[w], b[w], false, false);
      a[w] = (unsigned)r[0]; b[w] = (unsigned)r[1];
    }
    lo[t] = __builtin_bit_cast(bf16x4, a);
    hi[t] = __builtin_bit_cast(bf16x4, b);
  }
}

// Two rules shape the products (both measured on this kernel, tools/lnmlp_probe.py):
//  * the A fragments are fetched in batches of four, one batch ahead of the MFMAs that use them (left to itself hipcc placed every
//    ds_read directly in front of its MFMA behind an lgkmcnt(0): one LDS latency per MFMA);
//  * consecutive MFMAs go to DIFFERENT accumulators.  A chain on one accumulator only runs at the issue rate while nothing at all is
//    issued between its links (MI355X_MICROARCH.md: +43 cycles for the first foreign instruction); with fragment reads and another
//    wave's GELU in the stream that never holds, and the first version of this file ran its MFMAs at 25 % of their rate.
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

#ifdef MLP_PROBE_NOMFMA
#define MLP_MFMA(A, B, ACC) (ACC)[0] += bf2f((A)[0]) * bf2f((B)[0])
#else
#define MLP_MFMA(A, B, ACC) (ACC) = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (ACC), 0, 0, 0)
#endif

// both H^T tiles (32 hidden x 32 rows each) of the chunk: acc[i] += unit rows 32 i.. . frag^T, K = C; batch = 2 k-steps x 2 tiles
template <int KT>
__device__ __forceinline__ void product_k_c2(const bf16* bank, const bf16x8 (&bf)[KT], f32x16 (&acc)[2], int n, int h) {
  bf16x8 fa[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) fa[0][q] = afrag(bank, q & 1, q >> 1, n, h);
#pragma unroll
  for (int b = 0; b < KT / 2; ++b) {
    if (b + 1 < KT / 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = 2 * (b + 1) + (q >> 1);
        fa[(b + 1) & 1][q] = afrag(bank + (t >> 2) * UNIT, q & 1, t & 3, n, h);
      }
    }
    pin();
#pragma unroll
    for (int q = 0; q < 4; ++q) MLP_MFMA(fa[b & 1][q], bf[2 * b + (q >> 1)], acc[q & 1]);
    pin();
  }
}

// tile i of two products at once (backward: H^T from W1' / xhat, dG^T from W2^T / dy): two accumulators, batch = 2 k-steps x 2 operands
template <int KT, bool DB = true>
__device__ __forceinline__ void product_k_cc(const bf16* bank1, const bf16* bank2, int i, const bf16x8 (&b1)[KT], const bf16x8 (&b2)[KT],
                                             f32x16& acc1, f32x16& acc2, int n, int h) {
  if constexpr (!DB) {                                     // register-starved instance: one batch in flight, the partner wave hides the latency
#pragma unroll
    for (int b = 0; b < KT / 2; ++b) {
      bf16x8 f[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = 2 * b + (q >> 1);
        f[q] = afrag(((q & 1) ? bank2 : bank1) + (t >> 2) * UNIT, i, t & 3, n, h);
      }
      pin();
      MLP_MFMA(f[0], b1[2 * b], acc1);
      MLP_MFMA(f[1], b2[2 * b], acc2);
      MLP_MFMA(f[2], b1[2 * b + 1], acc1);
      MLP_MFMA(f[3], b2[2 * b + 1], acc2);
      pin();
    }
    return;
  }
  bf16x8 fa[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) fa[0][q] = afrag((q & 1) ? bank2 : bank1, i, q >> 1, n, h);
#pragma unroll
  for (int b = 0; b < KT / 2; ++b) {
    if (b + 1 < KT / 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = 2 * (b + 1) + (q >> 1);
        fa[(b + 1) & 1][q] = afrag(((q & 1) ? bank2 : bank1) + (t >> 2) * UNIT, i, t & 3, n, h);
      }
    }
    pin();
    MLP_MFMA(fa[b & 1][0], b1[2 * b], acc1);
    MLP_MFMA(fa[b & 1][1], b2[2 * b], acc2);
    MLP_MFMA(fa[b & 1][2], b1[2 * b + 1], acc1);
    MLP_MFMA(fa[b & 1][3], b2[2 * b + 1], acc2);
    pin();
  }
}

// out^T (C channels x 32 rows) += [C][64-hidden] units . frag^T  (frag[i][u]: hidden 32 i + 16 u .. + 15 of the chunk, fragment order);
// batch = one k-step of four channel tiles (four accumulators)
template <int MT, bool DB = true>
__device__ __forceinline__ void product_k_h(const bf16* bank, const bf16x8 (&gf)[2][2], f32x16 (&acc)[MT], int n, int h) {
  constexpr int NB = MT;                                   // MT / 4 tile groups x 4 k-steps
  if constexpr (!DB) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      bf16x8 f[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mt = (b >> 2) * 4 + q;
        f[q] = afrag(bank + (mt >> 1) * UNIT, mt & 1, b & 3, n, h);
      }
      pin();
#pragma unroll
      for (int q = 0; q < 4; ++q) MLP_MFMA(f[q], gf[(b & 3) >> 1][b & 1], acc[(b >> 2) * 4 + q]);
      pin();
    }
    return;
  }
  bf16x8 fa[2][4];
  auto fetch = [&](int b, bf16x8 (&f)[4]) {
    const int grp = b >> 2, ks = b & 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mt = grp * 4 + q;
      f[q] = afrag(bank + (mt >> 1) * UNIT, mt & 1, ks, n, h);
    }
  };
  fetch(0, fa[0]);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) fetch(b + 1, fa[(b + 1) & 1]);
    pin();
#pragma unroll
    for (int q = 0; q < 4; ++q) MLP_MFMA(fa[b & 1][q], gf[(b & 3) >> 1][b & 1], acc[(b >> 2) * 4 + q]);
    pin();
  }
}

__device__ __forceinline__ void bias_init(f32x16& acc, const float* b1s, int base, int h) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + base + b * 8 + h * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[b * 4 + c] = bv[c];
  }
}

template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void ln_mlp_fwd_kernel(MlpP p) {
  constexpr int HID = 4 * C, KT = C / 16, MT = C / 32, NCH = HID / 64, UPP = C / 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* banks = reinterpret_cast<bf16*>(smem_raw);                       // 2 banks x UPP units
  bf16* regions = banks + 2 * UPP * UNIT;                                // NW wave-private 32 x C tiles
  float* b1s = reinterpret_cast<float*>(regions + NW * 32 * C);          // [HID]
  float* b2s = b1s + HID;                                                 // [C]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * (NW * 32) + wave * 32;
  bf16* reg = regions + wave * 32 * C;

  dma_rows<C>(reg, p.x, row0, p.M, lane);
#pragma unroll
  for (int s = 0; s < UPP; ++s) dma_unit<NW>(banks + s * UNIT, p.w1p + s * 64, C, wave, lane);
  for (int i = tid; i < HID; i += 64 * NW) b1s[i] = p.b1p[i];
  for (int i = tid; i < C; i += 64 * NW) b2s[i] = p.b2[i];
  phase_sync();

  bf16x8 xh[KT], xr[KT];
  {
    float rstd;
    read_frags<C>(reg, n, h, xr);
    normalise<KT>(xr, xh, p.eps, rstd);
  }
  f32x16 yacc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[mt][r] = 0.f;

  // (the raw x fragments stay in registers for the residual; the wave's region is the staging image of G during the loop)
  bf16* bank1 = banks + UPP * UNIT;
  // G rows of a chunk (whole 128-byte rows out of the staging image, 8 per instruction).  Every barrier is preceded by vmcnt(0), which
  // on this target counts stores too: the rows of chunk j - 1 leave at the START of phase 2j, the long one (a product and the GELU to
  // drain under), not in the short phase 2j - 1 (measured: + 0.5 ms per call at stage 0 there).
  auto g_rows_out = [&](int jc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = q * 8 + (lane >> 3), c = (lane & 7) ^ (rr & 7);
      const bf16x8 vg = *reinterpret_cast<const bf16x8*>(reg + q * 512 + lane * 8);
      if (row0 + rr < p.M) *reinterpret_cast<bf16x8*>(p.g + (size_t)(row0 + rr) * HID + jc * 64 + c * 8) = vg;
    }
  };
  for (int j = 0; j < NCH; ++j) {
    // ---- phase 2j: H^T chunk from bank 0, GELU; W2 units of the chunk requested into bank 1
#pragma unroll
    for (int s = 0; s < UPP; ++s) dma_unit<NW>(bank1 + s * UNIT, p.w2p + (size_t)(s * 64) * HID + j * 64, HID, wave, lane);
    if (p.g && j > 0) g_rows_out(j - 1);
    bf16x8 gf[2][2];
    {
      f32x16 hacc[2];
      bias_init(hacc[0], b1s, j * 64, h);
      bias_init(hacc[1], b1s, j * 64 + 32, h);
      product_k_c2<KT>(banks, xh, hacc, n, h);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
#ifdef MLP_PROBE_NOGELU
            const f32x2 gv = f32x2{hacc[i][u * 8 + e], hacc[i][u * 8 + e + 1]};
#else
            const f32x2 gv = gelu2(f32x2{hacc[i][u * 8 + e], hacc[i][u * 8 + e + 1]});
#endif
            gf[i][u][e] = f2bf(gv.x); gf[i][u][e + 1] = f2bf(gv.y);
          }
    }
    if (p.g) {                                             // G pieces (hidden 32 i + 8 b + 4 h .. + 3 of the chunk) into the staging image
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bf16x8& f = gf[i][b >> 1];
          const int e0 = (b & 1) * 4;
          *reinterpret_cast<bf16x4*>(reg + n * 64 + (((4 * i + b) ^ (n & 7)) << 3) + h * 4) = bf16x4{f[e0], f[e0 + 1], f[e0 + 2], f[e0 + 3]};
        }
    }
    phase_sync();
    // ---- phase 2j + 1: next chunk's W1' units into bank 0; Y^T += W2 chunk . G^T from bank 1
    if (j + 1 < NCH) {
#pragma unroll
      for (int s = 0; s < UPP; ++s) dma_unit<NW>(banks + s * UNIT, p.w1p + (size_t)((j + 1) * 64) * C + s * 64, C, wave, lane);
    }
    product_k_h<MT>(bank1, gf, yacc, n, h);
    phase_sync();
  }
  if (p.g) g_rows_out(NCH - 1);

  // ---- y = x + s (Y + b2) on the C layout's 8-byte pieces (x of those channels out of the B fragments by a lane-pair exchange), through
  // the wave's region, whole rows out
  {
    const int row = min(row0 + n, p.M - 1);
    const float s = p.rowscale ? p.rowscale[row / p.rps] : 1.f;
    bf16x4 xlo[KT], xhi[KT];
    frags_to_cpieces<KT>(xr, xlo, xhi);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int t = 2 * mt + (b >> 1);
        const bf16x4 xv = (b & 1) ? xhi[t] : xlo[t];
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b2s + mt * 32 + b * 8 + h * 4);
        bf16x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = f2bf(bf2f(xv[c]) + s * (yacc[mt][b * 4 + c] + bv[c]));
        *reinterpret_cast<bf16x4*>(cpiece<C>(reg, n, h, mt, b)) = o;
      }
    store_rows<C>(reg, p.y, row0, p.M, lane);
  }
}

template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void ln_mlp_bwd_kernel(MlpP p) {
  constexpr int HID = 4 * C, KT = C / 16, MT = C / 32, NCH = HID / 64, UPP = C / 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* banks = reinterpret_cast<bf16*>(smem_raw);
  bf16* bankA = banks;                                     // 2 UPP units: W1' and W2^T rows of a chunk
  bf16* bankB = banks + 2 * UPP * UNIT;                    // UPP units: W1'^T columns of a chunk
  bf16* regions = banks + 3 * UPP * UNIT;
  // during the chunk loop a wave's region holds two 4-KB staging images (dH, G).  The fc1 bias table: an area of its own at C = 128;
  // at C = 256 (164 KB otherwise) the unused second half of wave 0's 16-KB region, written once that wave has read its dy tile
  float* b1s = reinterpret_cast<float*>(C == 128 ? regions + NW * 32 * C : regions + 2 * 32 * 64);   // [HID]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * (NW * 32) + wave * 32;
  bf16* reg = regions + wave * 32 * C;

  dma_rows<C>(reg, p.x, row0, p.M, lane);
#pragma unroll
  for (int sl = 0; sl < UPP; ++sl) {
    dma_unit<NW>(bankA + sl * UNIT, p.w1p + sl * 64, C, wave, lane);
    dma_unit<NW>(bankA + (UPP + sl) * UNIT, p.w2tp + sl * 64, C, wave, lane);
  }
  wait_all();                                              // (the region is wave-private: no barrier needed for it)

  bf16x8 xh[KT], dyr[KT];
  float rstd;
  {
    bf16x8 xr[KT];
    read_frags<C>(reg, n, h, xr);
    normalise<KT>(xr, xh, p.eps, rstd);
  }
  // xhat: operand of the fc1 weight-gradient GEMM -- back into the region in place, whole rows out
  write_frags<C>(reg, n, h, xh);
  store_rows<C>(reg, p.xhat, row0, p.M, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the rows are in registers before dy overwrites the region
  __builtin_amdgcn_sched_barrier(0);
  dma_rows<C>(reg, p.dy, row0, p.M, lane);
  wait_all();
  read_frags<C>(reg, n, h, dyr);
  if (C != 128) phase_sync();
  for (int i = tid; i < HID; i += 64 * NW) b1s[i] = p.b1p[i];
  const float s = p.rowscale ? p.rowscale[min(row0 + n, p.M - 1) / p.rps] : 1.f;
  phase_sync();                                            // chunk 0's units and the bias table resident for everyone

  f32x16 dxacc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dxacc[mt][r] = 0.f;

  // phase 2j: H^T and dG^T from bank A, dH / G out; phase 2j + 1: dXhat^T from bank B.  Each phase requests the other bank's next units.
  auto dh_rows_out = [&](int jc) {                        // (see the forward's g_rows_out for the placement)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = q * 8 + (lane >> 3), c = (lane & 7) ^ (rr & 7);
      const bf16x8 vd = *reinterpret_cast<const bf16x8*>(reg + q * 512 + lane * 8);
      if (row0 + rr < p.M) *reinterpret_cast<bf16x8*>(p.dh + (size_t)(row0 + rr) * HID + jc * 64 + c * 8) = vd;
      pin();
    }
  };
  for (int j = 0; j < NCH; ++j) {
#pragma unroll
    for (int sl = 0; sl < UPP; ++sl) dma_unit<NW>(bankB + sl * UNIT, p.w1tp + (size_t)(sl * 64) * HID + j * 64, HID, wave, lane);
    if (j > 0) dh_rows_out(j - 1);
    bf16x8 hf[2][2];
    // dH pieces go through a staging image [32 rows][64 hidden] in the wave's region (128-byte rows, 16-byte pieces XOR (row & 7))
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 hacc, gacc;
      bias_init(hacc, b1s, j * 64 + i * 32, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
      product_k_cc<KT, (C > 128)>(bankA, bankA + UPP * UNIT, i, xh, dyr, hacc, gacc, n, h);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        bf16x4 od;
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          const f32x2 gp = gelu_grad2(f32x2{hacc[b * 4 + c], hacc[b * 4 + c + 1]});
          od[c] = f2bf(s * gacc[b * 4 + c] * gp.x); od[c + 1] = f2bf(s * gacc[b * 4 + c + 1] * gp.y);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) hf[i][b >> 1][(b & 1) * 4 + c] = od[c];
        *reinterpret_cast<bf16x4*>(reg + n * 64 + (((4 * i + b) ^ (n & 7)) << 3) + h * 4) = od;
      }
    }
    phase_sync();
    // ---- phase 2j + 1: request W1' / W2^T units of the next chunk into bank A; dXhat^T += W1'^T chunk . dH^T
    if (j + 1 < NCH) {
#pragma unroll
      for (int sl = 0; sl < UPP; ++sl) {
        dma_unit<NW>(bankA + sl * UNIT, p.w1p + (size_t)((j + 1) * 64) * C + sl * 64, C, wave, lane);
        dma_unit<NW>(bankA + (UPP + sl) * UNIT, p.w2tp + (size_t)((j + 1) * 64) * C + sl * 64, C, wave, lane);
      }
    }
    product_k_h<MT, (C > 128)>(bankB, hf, dxacc, n, h);
    phase_sync();
  }
  dh_rows_out(NCH - 1);

  // ---- LayerNorm backward (no affine part here) + residual gradient: dx = dy + rstd (a - mean_c(a) - xhat mean_c(a xhat)), on the C
  // layout's pieces; xhat and dy of those channels come out of the B fragments by a lane-pair exchange
  bf16x4 xlo[KT], xhi[KT], dlo[KT], dhi[KT];
  frags_to_cpieces<KT>(xh, xlo, xhi);
  frags_to_cpieces<KT>(dyr, dlo, dhi);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = 2 * mt + (b >> 1);
        const float a = dxacc[mt][b * 4 + c], xv = bf2f((b & 1) ? xhi[t][c] : xlo[t][c]);
        s1 += a; s2 += a * xv;
      }
  s1 = pair_sum(s1) * (1.f / C);
  s2 = pair_sum(s2) * (1.f / C);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int t = 2 * mt + (b >> 1);
      bf16x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xv = bf2f((b & 1) ? xhi[t][c] : xlo[t][c]), dv = bf2f((b & 1) ? dhi[t][c] : dlo[t][c]);
        o[c] = f2bf(dv + rstd * (dxacc[mt][b * 4 + c] - s1 - xv * s2));
      }
      *reinterpret_cast<bf16x4*>(cpiece<C>(reg, n, h, mt, b)) = o;
    }
  store_rows<C>(reg, p.dx, row0, p.M, lane);
}

bool attr_fwd[16] = {}, attr_bwd[16] = {};

template <int C, int NW>
constexpr size_t fwd_smem() { return (size_t)2 * (C / 64) * UNIT * 2 + (size_t)NW * 32 * C * 2 + (size_t)(4 * C + C) * 4; }
template <int C, int NW>
constexpr size_t bwd_smem() { return (size_t)3 * (C / 64) * UNIT * 2 + (size_t)NW * 32 * C * 2 + (C == 128 ? (size_t)(4 * C) * 4 : 0); }
// waves per workgroup: C = 128 fits 256 registers per wave -> 8 waves (two per SIMD: one wave's GELU runs under the other's MFMAs) sharing
// one set of weight units; C = 256 needs the whole register file of a SIMD per wave -> 4 waves
constexpr int NW128 = 8, NW256 = 4;

}  // namespace

// y = x + rowscale[row / rps] * (fc2(gelu(fc1'(xhat))) + b2), xhat = (x - mean) rstd per row.  x, y: bf16 [M, C]; w1p = W1 diag(gamma) [4C, C],
// w2p = FA(W2) [C, 4C] (FA: K columns in the order "bits 2 and 3 of the index swapped"), b1p = b1 + W1 beta, b2: fp32.  C in {128, 256}.
// g (optional, bf16 [M, 4C]): gelu(H), kept for the backward's fc2 weight gradient.
extern "C" int fiber_ln_mlp_fwd_bf16(const void* x, const void* w1p, const float* b1p, const void* w2p, const float* b2,
                                     const float* rowscale, void* y, void* g, int M, int C, int rows_per_sample, float eps,
                                     hipStream_t stream) {
  if (M <= 0 || (C != 128 && C != 256) || (rowscale && (rows_per_sample <= 0 || M % rows_per_sample))) return FIBER_EINVAL;
  MlpP p{};
  p.x = (const bf16*)x; p.w1p = (const bf16*)w1p; p.w2p = (const bf16*)w2p; p.b1p = b1p; p.b2 = b2; p.rowscale = rowscale;
  p.y = (bf16*)y; p.g = (bf16*)g; p.M = M; p.rps = rows_per_sample > 0 ? rows_per_sample : M; p.eps = eps;
  static_assert(fwd_smem<128, NW128>() <= 160 * 1024 && fwd_smem<256, NW256>() <= 160 * 1024, "LDS budget");
  if (fiber_first_on_device(attr_fwd)) {
    hipFuncSetAttribute((const void*)ln_mlp_fwd_kernel<128, NW128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_smem<128, NW128>());
    hipFuncSetAttribute((const void*)ln_mlp_fwd_kernel<256, NW256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_smem<256, NW256>());
  }
  if (C == 128) hipLaunchKernelGGL((ln_mlp_fwd_kernel<128, NW128>), dim3(cdiv(M, NW128 * 32)), dim3(64 * NW128), (fwd_smem<128, NW128>()), stream, p);
  else hipLaunchKernelGGL((ln_mlp_fwd_kernel<256, NW256>), dim3(cdiv(M, NW256 * 32)), dim3(64 * NW256), (fwd_smem<256, NW256>()), stream, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Backward of the above from x and dy (H is recomputed): dx [M, C] (LayerNorm backward + residual gradient included), and the operands of
// the fc1 weight-gradient GEMM: dh = rowscale * (dy W2) * gelu'(H) [M, 4C], xhat [M, C] (bf16).
// w2tp = W2^T [4C, C], w1tp = FA((W1 diag(gamma))^T) [C, 4C].
extern "C" int fiber_ln_mlp_bwd_bf16(const void* x, const void* dy, const void* w1p, const float* b1p, const void* w2tp, const void* w1tp,
                                     const float* rowscale, void* dx, void* dh, void* xhat, int M, int C, int rows_per_sample,
                                     float eps, hipStream_t stream) {
  if (M <= 0 || (C != 128 && C != 256) || (rowscale && (rows_per_sample <= 0 || M % rows_per_sample))) return FIBER_EINVAL;
  MlpP p{};
  p.x = (const bf16*)x; p.dy = (const bf16*)dy; p.w1p = (const bf16*)w1p; p.w2tp = (const bf16*)w2tp; p.w1tp = (const bf16*)w1tp;
  p.b1p = b1p; p.rowscale = rowscale; p.dx = (bf16*)dx; p.dh = (bf16*)dh; p.xhat = (bf16*)xhat;
  p.M = M; p.rps = rows_per_sample > 0 ? rows_per_sample : M; p.eps = eps;
  static_assert(bwd_smem<128, NW128>() <= 160 * 1024 && bwd_smem<256, NW256>() <= 160 * 1024, "LDS budget");
  if (fiber_first_on_device(attr_bwd)) {
    hipFuncSetAttribute((const void*)ln_mlp_bwd_kernel<128, NW128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_smem<128, NW128>());
    hipFuncSetAttribute((const void*)ln_mlp_bwd_kernel<256, NW256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_smem<256, NW256>());
  }
  if (C == 128) hipLaunchKernelGGL((ln_mlp_bwd_kernel<128, NW128>), dim3(cdiv(M, NW128 * 32)), dim3(64 * NW128), (bwd_smem<128, NW128>()), stream, p);
  else hipLaunchKernelGGL((ln_mlp_bwd_kernel<256, NW256>), dim3(cdiv(M, NW256 * 32)), dim3(64 * NW256), (bwd_smem<256, NW256>()), stream, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
