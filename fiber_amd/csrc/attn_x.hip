// One-pass kernels for the three attention sites of the fused layers whose ONE side is tiny (gfx950 / CDNA4; round 5):
//   i2t_bwd_kernel / i2t_fwd_kernel   image -> text cross attention (swin_transformer.py:226-259): 576 / 144 queries x 40 text keys, head_dim 32
//   t2i_bwd_kernel / t2i_fwd_kernel   text -> image cross attention and text self attention (roberta.py:256-326, 474-483): 40 queries x 576 / 144 /
//                                     40 keys, head_dim 64, attention-probability dropout
// The generic kernels of attn.hip treat both sides alike (key chunks staged in LDS per workgroup behind barriers; a delta kernel, a query-strip
// pass and a key-strip pass in the backward, K / V read twice): these sites ran at 1.5-2.7 TB/s of their bytes (0.19-0.34 of the HBM floor).
// Here the small side of a (sample, head) lives in REGISTERS as MFMA operands (in both orientations where needed) and the large side is streamed
// through once, 16 rows per step, with no workgroup barrier inside the stream; fiber_mha_fwd_bf16 / fiber_mha_bwd_bf16 dispatch here when the
// shape fits (FIBER_ATTN_I2T_ONEPASS / FIBER_ATTN_T2I_ONEPASS = 0: the generic kernels, for A/B runs) and fall back otherwise.
//
// ---- image -> text, backward ---------------------------------------------------------------------------------------------------------
// The text side is tiny -- K and V of one (sample, head) are
// 40 x 64 B -- so a wave keeps them in REGISTERS (as MFMA operands in both orientations) and streams the query strips once:
//   * per strip of 16 queries a lane loads 16 bytes of q, dO and O; delta comes out of the dO / O pieces it already holds;
//   * the scores are formed in BOTH orientations (the matrix pipe is idle anyway): S^T[key][query] (lane = query) feeds dQ^T = K^T . dS^T,
//     S[query][key] (lane = key) feeds dK^T += Q^T . dS and dV^T += dO^T . P, whose contraction runs over the strip's 16 queries
//     (K = 16 MFMAs; Q^T / dO^T of the strip are read back transposed from a wave-private 3-KB LDS area);
//   * dK / dV of the (sample, head) accumulate in registers over the strips a wave walks and are folded once at the end;
//   * a workgroup is 4 adjacent heads x 2 halves of the strips: the q / dO / O / dQ rows of the four heads are one 256-byte run touched by
//     four waves at the same time (tools/probes/run_probe.hip: scattered 64-byte runs move at 2.2-3.5 TB/s, neighbours requested together
//     at 4.2-5.5), and dQ leaves as 16-byte stores.
// No barrier inside the strip loop.  Conditions (else the generic passes run): head_dim 32, Lk <= 48, Lq % 16 == 0, heads % 4 == 0, no
// attention dropout (the reference applies none at this site).
#include "common.h"

namespace {

constexpr int XRS = 48;                                  // LDS row stride in elements: 32 + 16 pad = 96 B (conflict-free for both read kinds)

struct XP {
  const bf16* q; const bf16* k; const bf16* v; const bf16* o; const bf16* dout;
  bf16* dq; bf16* dk; bf16* dv;
  const float* lse; const float* kmask;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int H, Lq, Lk, G;
  float scale;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = f2bf(a[e]); o[4 + e] = f2bf(b[e]); }
  return o;
}
__device__ __forceinline__ s16x4 pack4(const f32x4& a) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf(a[e]);
  return __builtin_bit_cast(s16x4, o);
}
// transposed MFMA operand out of a row-major [row][32] image: rows {t0*16 + g*4 ..+3} U {t0*16 + 16 + g*4 ..+3}, column d0 + l
__device__ __forceinline__ bf16x8 trr_frag(const bf16* rm, int d0, int t0, int g, int l) {
  const bf16* a = rm + (t0 * 16 + g * 4 + (l >> 2)) * XRS + d0 + (l & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 16 * XRS));
  return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
// the same for ONE 16-row tile (K = 16 operand): [d0 + l][rows g*4 .. +3]
__device__ __forceinline__ s16x4 tr16(const bf16* rm, int d0, int g, int l) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rm + (g * 4 + (l >> 2)) * XRS + d0 + (l & 3) * 4));
}
__device__ __forceinline__ f32x4 exp2x4(f32x4 a) {
  return f32x4{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1]), __builtin_amdgcn_exp2f(a[2]), __builtin_amdgcn_exp2f(a[3])};
}
// 16-byte row pieces out of the C layout of two 16x16 MFMAs (see win_attn.hip rows4_gather16)
__device__ __forceinline__ u32x4 rows4_gather16(f32x4 a, f32x4 b) {
  const bf16x2 alo = {f2bf(a[0]), f2bf(a[1])}, ahi = {f2bf(a[2]), f2bf(a[3])}, blo = {f2bf(b[0]), f2bf(b[1])}, bhi = {f2bf(b[2]), f2bf(b[3])};
  const unsigned wa[2] = {__builtin_bit_cast(unsigned, alo), __builtin_bit_cast(unsigned, ahi)}, wb[2] = {__builtin_bit_cast(unsigned, blo), __builtin_bit_cast(unsigned, bhi)};
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const auto x = __builtin_amdgcn_permlane32_swap(wa[e], wb[e], false, false);
    const auto y = __builtin_amdgcn_permlane16_swap((unsigned)x[0], (unsigned)x[1], false, false);
    o[e] = (unsigned)y[0]; o[2 + e] = (unsigned)y[1];
  }
  return o;
}

constexpr int X_KV = 4 * 64 * XRS * 2;                   // bytes of the K (or V) images of four heads (64 rows each, rows >= Lk zero)
constexpr int X_STRIP = 16 * XRS * 2;                    // one strip image (Q or dO) of a wave
constexpr int X_SMEM = 2 * X_KV + 8 * 2 * X_STRIP + 8 * 32 * 4 + 4 * 12 * 64 * 16 + 64 * 4;   // (+ the key-mask table)

__global__ __launch_bounds__(512) void i2t_bwd_kernel(XP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = reinterpret_cast<bf16*>(smem + X_KV);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gq = lane >> 4, lq = lane & 15;
  const int hl = wave & 3, half = wave >> 2;
  const int b = blockIdx.x, h0 = blockIdx.y * 4, h = h0 + hl;
  bf16* Qst = reinterpret_cast<bf16*>(smem + 2 * X_KV + wave * 2 * X_STRIP);
  bf16* dOst = Qst + 16 * XRS;
  float* stat = reinterpret_cast<float*>(smem + 2 * X_KV + 8 * 2 * X_STRIP) + wave * 32;
  f32x4* red = reinterpret_cast<f32x4*>(smem + 2 * X_KV + 8 * 2 * X_STRIP + 8 * 32 * 4);
  // additive key mask of the sample in the log2 domain, -inf past Lk: ONE load per thread into an LDS table (as per-lane loads behind
  // `key < Lk` branches hipcc gave every one of the 15 its own vmcnt(0): fifteen DRAM latencies in a row at the head of every workgroup)
  float* mkl = reinterpret_cast<float*>(smem + X_SMEM - 64 * 4);
  if (tid < 64) mkl[tid] = tid < p.Lk ? (p.kmask ? p.kmask[(size_t)b * p.Lk + tid] * 1.4426950408889634f : 0.f) : -INFINITY;
  // ---- K, V of the four heads: row-major images, rows >= Lk zero
  for (int idx = tid; idx < 4 * 64 * 4; idx += 512) {
    const int hh = idx >> 8, r = (idx >> 2) & 63, c = idx & 3;
    bf16x8 kv, vv;
#pragma unroll
    for (int e = 0; e < 8; ++e) { kv[e] = f2bf(0.f); vv[e] = f2bf(0.f); }
    if (r < p.Lk) {
      const size_t row = (size_t)b * p.Lk + r;
      kv = *reinterpret_cast<const bf16x8*>(p.k + row * p.ldk + (h0 + hh) * 32 + c * 8);
      vv = *reinterpret_cast<const bf16x8*>(p.v + row * p.ldv + (h0 + hh) * 32 + c * 8);
    }
    *reinterpret_cast<bf16x8*>(Ks + (hh * 64 + r) * XRS + c * 8) = kv;
    *reinterpret_cast<bf16x8*>(Vs + (hh * 64 + r) * XRS + c * 8) = vv;
  }
  __syncthreads();
  const bf16* Kh = Ks + hl * 64 * XRS;
  const bf16* Vh = Vs + hl * 64 * XRS;
  bf16x8 kfr[3], vfr[3], ktf[2][2];                      // K, V rows of key tile kt (operand of both orientations); K^T of tile pairs (0,1), (2,3)
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    kfr[kt] = *reinterpret_cast<const bf16x8*>(Kh + (kt * 16 + lq) * XRS + gq * 8);
    vfr[kt] = *reinterpret_cast<const bf16x8*>(Vh + (kt * 16 + lq) * XRS + gq * 8);
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) ktf[dt][pr] = trr_frag(Kh, dt * 16, 2 * pr, gq, lq);
  // additive key mask in the log2 domain: lane = query form (keys kt*16 + gq*4 + r) and lane = key form (key kt*16 + lq); -inf past Lk
  f32x4 mkT[3];
  float mkS[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    mkT[kt] = *reinterpret_cast<const f32x4*>(mkl + kt * 16 + gq * 4);
    mkS[kt] = mkl[kt * 16 + lq];
  }
  const float c2 = p.scale * 1.4426950408889634f, inv_c2 = 1.f / c2;
  f32x4 dkT[3][2], dvT[3][2];                            // dK^T / dV^T[d = dt*16 + gq*4 + r][key = kt*16 + lq], summed over this wave's strips
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { dkT[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nstrips = p.Lq >> 4;
  // Strips are requested PD iterations ahead (registers): one iteration ahead left every strip waiting ~3 us for its rows (24 KB in flight
  // per CU: 2.5 TB/s); the strip arithmetic itself is ~0.4 us.  (The rotating queue below costs part of that distance: hipcc's waitcnt pass
  // waits for all but the newest one or two requests at the loop top -- see t2i_bwd_kernel for the form that does not; 3.6 TB/s as it is.)
  constexpr int PD = 3;
  bf16x8 qn[PD], don[PD], on[PD];
  float lsen[PD];
  auto prefetch = [&](int s, int slot) {
    const size_t row = (size_t)b * p.Lq + s * 16 + lq;
    qn[slot] = *reinterpret_cast<const bf16x8*>(p.q + row * p.ldq + h * 32 + gq * 8);
    don[slot] = *reinterpret_cast<const bf16x8*>(p.dout + row * p.lddo + h * 32 + gq * 8);
    on[slot] = *reinterpret_cast<const bf16x8*>(p.o + row * p.ldo + h * 32 + gq * 8);
    lsen[slot] = p.lse[row * p.H + h];
  };
#pragma unroll
  for (int d = 0; d < PD; ++d) {
    const int s = half + 2 * d;
    prefetch(s < nstrips ? s : half, d);
  }
  for (int s = half; s < nstrips; s += 2) {
    const bf16x8 qf = qn[0], dof = don[0], of = on[0];
    const float lse2 = lsen[0] * 1.4426950408889634f;
    const size_t row = (size_t)b * p.Lq + s * 16 + lq;
#pragma unroll
    for (int d = 0; d + 1 < PD; ++d) { qn[d] = qn[d + 1]; don[d] = don[d + 1]; on[d] = on[d + 1]; lsen[d] = lsen[d + 1]; }
    { const int sn = s + 2 * PD; prefetch(sn < nstrips ? sn : s, PD - 1); }     // (past the end: a row that is resident anyway, never used)
    // delta[query] = sum_d dO * O: the lane's 8-channel pieces, then the four lanes of the query
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    float dpart = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
      dpart = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, bf16x2{dof[e], dof[e + 1]}), __builtin_bit_cast(bf16x2_t, bf16x2{of[e], of[e + 1]}), dpart, false);
    const float dlt = rows4_sum(dpart);
    // the strip's Q / dO rows and its per-query seeds for the lane = key orientation (wave-private LDS: in-order, no barrier)
    *reinterpret_cast<bf16x8*>(Qst + lq * XRS + gq * 8) = qf;
    *reinterpret_cast<bf16x8*>(dOst + lq * XRS + gq * 8) = dof;
    if (gq == 0) { stat[lq] = -lse2 * inv_c2; stat[16 + lq] = -dlt; }
    // ---- lane = query: S^T[key][query], dS^T -> dQ
    f32x4 dsT[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[kt], qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const f32x4 dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[kt], dof, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const f32x4 pr = exp2x4(__builtin_elementwise_fma(st, f32x4{c2, c2, c2, c2}, mkT[kt] - lse2));
      dsT[kt] = pr * (dpt - dlt);
    }
    {
      const bf16x8 d01 = pack8(dsT[0], dsT[1]), d2 = pack8(dsT[2], f32x4{0.f, 0.f, 0.f, 0.f});
      f32x4 dq[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[dt][0], d01, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[dt][1], d2, dq[dt], 0, 0, 0);
      }
      *reinterpret_cast<u32x4*>(p.dq + row * p.lddq + h * 32 + gq * 8) = rows4_gather16(dq[0] * p.scale, dq[1] * p.scale);
    }
    // ---- lane = key: S[query][key], dS, P -> dK^T, dV^T (contraction over the strip's 16 queries)
    const f32x4 seedL = *reinterpret_cast<const f32x4*>(stat + gq * 4), seedD = *reinterpret_cast<const f32x4*>(stat + 16 + gq * 4);
    s16x4 qt[2], dot[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { qt[dt] = tr16(Qst, dt * 16, gq, lq); dot[dt] = tr16(dOst, dt * 16, gq, lq); }
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kfr[kt], seedL, 0, 0, 0);       // q.k - lse / (scale log2e)
      const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vfr[kt], seedD, 0, 0, 0);      // dP - delta
      const float mk = mkS[kt];
      const f32x4 pr = exp2x4(__builtin_elementwise_fma(sv, f32x4{c2, c2, c2, c2}, f32x4{mk, mk, mk, mk}));
      const s16x4 dsb = pack4(pr * dp), pb = pack4(pr);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dkT[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt[dt], dsb, dkT[kt][dt], 0, 0, 0);
        dvT[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dot[dt], pb, dvT[kt][dt], 0, 0, 0);
      }
    }
  }
  // ---- fold the two halves of a head and write dK, dV: lane holds [key = kt*16 + lq][d = dt*16 + gq*4 .. +3]
  f32x4* mine = red + (size_t)hl * 12 * 64 + lane;
  if (half == 1) {
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) { mine[(kt * 2 + dt) * 64] = dkT[kt][dt]; mine[(6 + kt * 2 + dt) * 64] = dvT[kt][dt]; }
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int key = kt * 16 + lq;
      if (key < p.Lk) {
        const size_t row = (size_t)b * p.Lk + key;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f32x4 a = (dkT[kt][dt] + mine[(kt * 2 + dt) * 64]) * p.scale, c = dvT[kt][dt] + mine[(6 + kt * 2 + dt) * 64];
          bf16x4 ok, ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) { ok[r] = f2bf(a[r]); ov[r] = f2bf(c[r]); }
          *reinterpret_cast<bf16x4*>(p.dk + row * p.lddk + h * 32 + dt * 16 + gq * 4) = ok;
          *reinterpret_cast<bf16x4*>(p.dv + row * p.lddv + h * 32 + dt * 16 + gq * 4) = ov;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Text -> image cross attention (roberta.py:256-326 with encoder_hidden_states = image tokens, 474-483) and text self attention, backward
// in ONE pass: head_dim 64, at most 48 QUERIES (40 text tokens), any number of keys (576 / 144 image tokens, 40 text tokens).
//
// The generic passes read K and V twice (once per pass) plus q / dO per key strip: 40 x 576 x d64 ran at 2.1 TB/s of its algorithmic
// bytes (0.26 of the floor).  Here the roles of attn_x's i2t kernel are mirrored: the QUERY side is the small one, so Q and dO of a
// (sample, head) live in registers (MFMA operands) and as LDS images (for the transposed operands), dQ accumulates in registers, and a
// wave streams 16-key tiles ONCE: K / V rows are loaded straight into MFMA operand registers (a lane = 16 bytes of a key row), two tiles
// ahead; per tile
//   S^T[key][query] = K Q^T, dP^T = V dO^T                 (lane = query: 12 MFMAs K = 32)
//   P, dropout (the forward's (row, key) hash), dS^T        (VALU, once -- the lane = key orientation is NOT recomputed: with dropout the
//                                                           VALU is the scarce pipe; P^T / dS^T go through a wave-private 3-KB LDS slab as
//                                                           [query][key] rows and come back transposed by ds_read_b64_tr_b16)
//   dQ^T[d][query] += K^T dS^T                              (K^T: transposed read of the tile's wave-private LDS image; 12 MFMAs K = 16)
//   dK^T[d][key] = Q^T dS, dV^T[d][key] = dO^T P_dropped    (contraction over the 48 queries: 24 MFMAs K = 16) -> 16-byte stores
// No barrier inside the tile loop; a workgroup = one (sample, head) = 4 waves taking every fourth tile; the four dQ partials are summed
// in a fixed order through LDS at the end (deterministic).
constexpr int TRS = 80;                                  // LDS row stride of the [row][64] images: 64 + 16 pad (as attn.hip's D + 16)
constexpr int T_IMG = 48 * TRS * 2;                      // Q (or dO) image
constexpr int T_KT = 16 * TRS * 2;                       // one wave's K tile image
constexpr int T_SLAB = 2 * 3 * 512;                      // one wave's P^T / dS^T slab: [tensor][query tile][16 queries][16 keys]
constexpr int T_MAXK = 1024;                             // keys served (the additive key mask of a sample is tabulated in LDS)
constexpr int T_FIX = 2 * T_IMG + 2 * 48 * 4 + 4 * T_KT + 4 * T_SLAB;    // 38,272 B; reused as 3 x 12 KB of dQ partials at the end
constexpr int T_SMEM = T_FIX + T_MAXK * 4;
static_assert(3 * 12 * 64 * 16 <= T_FIX, "dQ partials must fit the reused LDS");

struct TP {
  const bf16* q; const bf16* k; const bf16* v; const bf16* o; const bf16* dout;
  bf16* dq; bf16* dk; bf16* dv;
  const float* lse; const float* kmask;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int H, Lq, Lk;
  float scale, p_drop; uint64_t seed; const uint64_t* seed_base;
};
__device__ __forceinline__ s16x4 tr16s(const bf16* rm, int stride, int d0, int g, int l) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rm + (g * 4 + (l >> 2)) * stride + d0 + (l & 3) * 4));
}

template <typename T>
__device__ __forceinline__ T* atx(T* base, unsigned elem_off) {        // wave-uniform 64-bit base + 32-bit per-lane offset (win_attn.hip at())
  using C = std::conditional_t<std::is_const<T>::value, const char, char>;
  return reinterpret_cast<T*>(reinterpret_cast<C*>(base) + elem_off * (unsigned)sizeof(T));
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void t2i_bwd_kernel(TP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Qs = reinterpret_cast<bf16*>(smem);
  bf16* dOs = reinterpret_cast<bf16*>(smem + T_IMG);
  float* st_lse = reinterpret_cast<float*>(smem + 2 * T_IMG);
  float* st_dlt = st_lse + 48;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gq = lane >> 4, lq = lane & 15;
  const int b = blockIdx.x, h = blockIdx.y;
  bf16* Kt = reinterpret_cast<bf16*>(smem + 2 * T_IMG + 2 * 48 * 4 + wave * T_KT);
  bf16* slabP = reinterpret_cast<bf16*>(smem + 2 * T_IMG + 2 * 48 * 4 + 4 * T_KT + wave * T_SLAB);
  bf16* slabS = slabP + 3 * 256;
  float* mkl = reinterpret_cast<float*>(smem + T_FIX);
  for (int j = tid; j < ((p.Lk + 15) & ~15); j += 256)   // additive key mask in the log2 domain, -inf past Lk
    mkl[j] = j < p.Lk ? (p.kmask ? p.kmask[(size_t)b * p.Lk + j] * 1.4426950408889634f : 0.f) : -INFINITY;
  // ---- Q, dO images (rows >= Lq zero), delta = rowsum(dO . O), lse in the log2 domain (+inf past Lq: p = 0 there)
  for (int idx = tid; idx < 48 * 8; idx += 256) {
    const int r = idx >> 3, c = idx & 7;
    bf16x8 qv, dv;
#pragma unroll
    for (int e = 0; e < 8; ++e) { qv[e] = f2bf(0.f); dv[e] = f2bf(0.f); }
    float dpart = 0.f;
    const size_t row = (size_t)b * p.Lq + (r < p.Lq ? r : 0);
    if (r < p.Lq) {
      qv = *reinterpret_cast<const bf16x8*>(p.q + row * p.ldq + h * 64 + c * 8);
      dv = *reinterpret_cast<const bf16x8*>(p.dout + row * p.lddo + h * 64 + c * 8);
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.o + row * p.ldo + h * 64 + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dpart += bf2f(dv[e]) * bf2f(ov[e]);
    }
    *reinterpret_cast<bf16x8*>(Qs + r * TRS + c * 8) = qv;
    *reinterpret_cast<bf16x8*>(dOs + r * TRS + c * 8) = dv;
    dpart += __shfl_xor(dpart, 1); dpart += __shfl_xor(dpart, 2); dpart += __shfl_xor(dpart, 4);
    if (c == 0) { st_dlt[r] = dpart; st_lse[r] = r < p.Lq ? p.lse[row * p.H + h] * 1.4426950408889634f : INFINITY; }
  }
  __syncthreads();
  bf16x8 qf[3][2], dof[3][2];
  float lse2[3], dlt[3];
#pragma unroll
  for (int qt = 0; qt < 3; ++qt) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[qt][ks] = *reinterpret_cast<const bf16x8*>(Qs + (qt * 16 + lq) * TRS + ks * 32 + gq * 8);
      dof[qt][ks] = *reinterpret_cast<const bf16x8*>(dOs + (qt * 16 + lq) * TRS + ks * 32 + gq * 8);
    }
    lse2[qt] = st_lse[qt * 16 + lq]; dlt[qt] = st_dlt[qt * 16 + lq];
  }
  f32x4 dqT[4][3];                                       // dQ^T[d = dt*16 + gq*4 + r][query = qt*16 + lq], this wave's key tiles
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) dqT[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float c2 = p.scale * 1.4426950408889634f;
  const uint32_t thresh = DROP ? (uint32_t)((double)p.p_drop * 4294967296.0) : 0u;
  const uint32_t dseed = DROP ? drop_seed32(p.seed + (p.seed_base ? *p.seed_base : 0ull)) : 0u;
  const float inv_keep = DROP ? 1.f / (1.f - p.p_drop) : 1.f;
  const uint32_t drow = (uint32_t)((b * p.H + h) * p.Lq);

  const int ntiles = (p.Lk + 15) >> 4;
  // K / V rows of a key tile go straight into MFMA operand registers (a lane = 16 bytes of key row kt*16 + lq).  ONE register set: the next
  // tile is requested as soon as this tile's first MFMAs have read the set, and the PREVIOUS tile's dK / dV rows (held packed in 16
  // registers) are stored at the same point, so when the loop comes round the only outstanding memory operations are a body old and the
  // compiler's vmcnt(0) costs nothing.  (A rotating queue of register sets made it wait for the NEWEST request at the top of every
  // iteration -- 20 k cycles per tile; stores at the end of the body made every iteration wait for its own stores.)
  const bf16* kbase = p.k + (size_t)b * p.Lk * p.ldk + h * 64 + gq * 8;
  const bf16* vbase = p.v + (size_t)b * p.Lk * p.ldv + h * 64 + gq * 8;
  bf16* dkbase = p.dk + (size_t)b * p.Lk * p.lddk + h * 64 + gq * 8;
  bf16* dvbase = p.dv + (size_t)b * p.Lk * p.lddv + h * 64 + gq * 8;
  bf16x8 kk[2], vv[2];
  auto request = [&](int kt) {
    const unsigned r = (unsigned)min(kt * 16 + lq, p.Lk - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kk[ks] = *reinterpret_cast<const bf16x8*>(atx(kbase, r * (unsigned)p.ldk + ks * 32));
      vv[ks] = *reinterpret_cast<const bf16x8*>(atx(vbase, r * (unsigned)p.ldv + ks * 32));
    }
  };
  u32x4 wk[2], wv[2];                                    // the previous tile's dK / dV rows (16 bytes per lane and half)
  auto flush = [&](int kt) {
    const int key = kt * 16 + lq;
    if (key < p.Lk) {
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        *reinterpret_cast<u32x4*>(atx(dkbase, (unsigned)key * (unsigned)p.lddk + hp * 32)) = wk[hp];
        *reinterpret_cast<u32x4*>(atx(dvbase, (unsigned)key * (unsigned)p.lddv + hp * 32)) = wv[hp];
      }
    }
  };
  if (wave < ntiles) request(wave);
  for (int kt = wave; kt < ntiles; kt += 4) {
    const bf16x8 kf[2] = {kk[0], kk[1]}, vf[2] = {vv[0], vv[1]};
    // the tile's K rows as a wave-private image for the transposed operand of dQ (in-order LDS: no barrier)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) *reinterpret_cast<bf16x8*>(Kt + lq * TRS + ks * 32 + gq * 8) = kf[ks];
    const f32x4 mk = *reinterpret_cast<const f32x4*>(mkl + kt * 16 + gq * 4);   // additive key mask (log2 domain); -inf past Lk
    f32x4 st[3], dp[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
      st[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      st[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1], qf[qt][1], st[qt], 0, 0, 0);
      dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], dof[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1], dof[qt][1], dp[qt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt != wave) flush(kt - 4);
    request(kt + 4 < ntiles ? kt + 4 : kt);              // (past the end: rows that are resident anyway, never used)
    __builtin_amdgcn_sched_barrier(0);
    s16x4 dsb[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
      const float nl = -lse2[qt];
      const f32x4 pr = exp2x4(__builtin_elementwise_fma(st[qt], f32x4{c2, c2, c2, c2}, mk + nl));
      f32x4 prd = pr, dpe = dp[qt];
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool keep = drop_keep_rk(dseed, drow + (uint32_t)min(qt * 16 + lq, p.Lq - 1), (uint32_t)(kt * 16 + gq * 4 + r), thresh);
          dpe[r] = keep ? dpe[r] * inv_keep : 0.f;
          prd[r] = keep ? pr[r] * inv_keep : 0.f;
        }
      }
      const float dl = dlt[qt];
      dsb[qt] = pack4(pr * (dpe - dl));
      *reinterpret_cast<s16x4*>(slabP + qt * 256 + lq * 16 + gq * 4) = pack4(prd);
      *reinterpret_cast<s16x4*>(slabS + qt * 256 + lq * 16 + gq * 4) = dsb[qt];
    }
    // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const s16x4 ktT = tr16s(Kt, TRS, dt * 16, gq, lq);
#pragma unroll
      for (int qt = 0; qt < 3; ++qt) dqT[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ktT, dsb[qt], dqT[dt][qt], 0, 0, 0);
    }
    // dK^T[d][key] = sum_q Q^T[d][q] dS[q][key];  dV^T[d][key] = sum_q dO^T[d][q] P_dropped[q][key]
    s16x4 pB[3], dB[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) { pB[qt] = tr16s(slabP + qt * 256, 16, 0, gq, lq); dB[qt] = tr16s(slabS + qt * 256, 16, 0, gq, lq); }
    f32x4 dkT[4], dvT[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dkT[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qt = 0; qt < 3; ++qt) {
        dkT[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr16s(Qs + qt * 16 * TRS, TRS, dt * 16, gq, lq), dB[qt], dkT[dt], 0, 0, 0);
        dvT[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tr16s(dOs + qt * 16 * TRS, TRS, dt * 16, gq, lq), pB[qt], dvT[dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {                     // (the permlane swaps run with every lane active; only the stores are guarded)
      wk[hp] = rows4_gather16(dkT[2 * hp] * p.scale, dkT[2 * hp + 1] * p.scale);
      wv[hp] = rows4_gather16(dvT[2 * hp], dvT[2 * hp + 1]);
    }
  }
  if (wave < ntiles) flush(wave + ((ntiles - 1 - wave) >> 2) * 4);
  // ---- dQ: waves 1..3 park their partials in the (now dead) LDS, wave 0 adds them in a fixed order and writes
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  if (wave > 0) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 3; ++qt) red[((wave - 1) * 12 + dt * 3 + qt) * 64 + lane] = dqT[dt][qt];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 3; ++qt)
#pragma unroll
        for (int w = 0; w < 3; ++w) dqT[dt][qt] += red[(w * 12 + dt * 3 + qt) * 64 + lane];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
      const int qi = qt * 16 + lq;
      const size_t row = (size_t)b * p.Lq + min(qi, p.Lq - 1);
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        const u32x4 w = rows4_gather16(dqT[2 * hp][qt] * p.scale, dqT[2 * hp + 1][qt] * p.scale);
        if (qi < p.Lq) *reinterpret_cast<u32x4*>(p.dq + row * p.lddq + h * 64 + hp * 32 + gq * 8) = w;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Forward of the same site (head_dim 64, at most 48 queries, any number of keys): a workgroup = one (sample, head), its four waves take
// every fourth 16-key tile with Q in registers and an ONLINE softmax per wave (running max / sum per query, the O^T accumulators rescaled
// only when some query's max moved), V^T through a wave-private LDS image; the four partial (max, sum, O^T) triples are merged pairwise
// through LDS in a fixed order.  K / V rows go from global memory straight into MFMA operand registers, one tile ahead.  The generic
// forward staged whole key chunks per workgroup behind barriers: 40 x 576 x d64 moved its bytes at 2.7 TB/s.
constexpr int F_SLAB = 14 * 64 * 16;                     // one wave's partial: 12 O^T tiles + (max, sum) of its three query tiles
constexpr int F_SMEM = T_MAXK * 4 + 2 * F_SLAB;          // key-mask table + two slabs (the waves' V^T images live in the slab area until the merge)
static_assert(4 * T_KT <= 2 * F_SLAB, "V^T images must fit the slab area");

struct FP {
  const bf16* q; const bf16* k; const bf16* v; bf16* o; float* lse; const float* kmask;
  int ldq, ldk, ldv, ldo;
  int H, Lq, Lk;
  float scale, p_drop; uint64_t seed; const uint64_t* seed_base;
};

template <bool DROP>
__global__ __launch_bounds__(256, 3) void t2i_fwd_kernel(FP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* mkl = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gq = lane >> 4, lq = lane & 15;
  const int b = blockIdx.x, h = blockIdx.y;
  bf16* Vt = reinterpret_cast<bf16*>(smem + T_MAXK * 4 + wave * T_KT);
  for (int j = tid; j < ((p.Lk + 15) & ~15); j += 256)   // additive key mask in the log2 domain, -inf past Lk
    mkl[j] = j < p.Lk ? (p.kmask ? p.kmask[(size_t)b * p.Lk + j] * 1.4426950408889634f : 0.f) : -INFINITY;
  bf16x8 qf[3][2];
#pragma unroll
  for (int qt = 0; qt < 3; ++qt) {
    const size_t row = (size_t)b * p.Lq + min(qt * 16 + lq, p.Lq - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8*>(p.q + row * p.ldq + h * 64 + ks * 32 + gq * 8);
  }
  const float c2 = p.scale * 1.4426950408889634f;
  const uint32_t thresh = DROP ? (uint32_t)((double)p.p_drop * 4294967296.0) : 0u;
  const uint32_t dseed = DROP ? drop_seed32(p.seed + (p.seed_base ? *p.seed_base : 0ull)) : 0u;
  const float inv_keep = DROP ? 1.f / (1.f - p.p_drop) : 1.f;
  const uint32_t drow = (uint32_t)((b * p.H + h) * p.Lq);
  f32x4 oT[4][3];                                        // O^T[d = dt*16 + gq*4 + r][query = qt*16 + lq], un-normalised
  float m[3], l[3];                                      // running max (log2 domain) and this LANE's share of the running sum
#pragma unroll
  for (int qt = 0; qt < 3; ++qt) {
    m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oT[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();                                       // mask table
  const int ntiles = (p.Lk + 15) >> 4;
  const bf16* kbase = p.k + (size_t)b * p.Lk * p.ldk + h * 64 + gq * 8;
  const bf16* vbase = p.v + (size_t)b * p.Lk * p.ldv + h * 64 + gq * 8;
  bf16x8 kk[2], vv[2];
  auto request = [&](int kt) {
    const unsigned r = (unsigned)min(kt * 16 + lq, p.Lk - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kk[ks] = *reinterpret_cast<const bf16x8*>(atx(kbase, r * (unsigned)p.ldk + ks * 32));
      vv[ks] = *reinterpret_cast<const bf16x8*>(atx(vbase, r * (unsigned)p.ldv + ks * 32));
    }
  };
  if (wave < ntiles) request(wave);
  for (int kt = wave; kt < ntiles; kt += 4) {
    const bf16x8 kf[2] = {kk[0], kk[1]};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) *reinterpret_cast<bf16x8*>(Vt + lq * TRS + ks * 32 + gq * 8) = vv[ks];
    f32x4 st[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
      st[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      st[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1], qf[qt][1], st[qt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    request(kt + 4 < ntiles ? kt + 4 : kt);              // (past the end: rows that are resident anyway, never used)
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 mk = *reinterpret_cast<const f32x4*>(mkl + kt * 16 + gq * 4);
    s16x4 pb[3];
    bool moved = false;
    float alpha[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
      const f32x4 sv = __builtin_elementwise_fma(st[qt], f32x4{c2, c2, c2, c2}, mk);
      const float tmax = rows4_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
      const float mnew = fmaxf(m[qt], tmax);
      alpha[qt] = __builtin_amdgcn_exp2f(m[qt] - mnew);  // (first tile: exp2(-inf) = 0)
      moved |= mnew > m[qt];
      m[qt] = mnew;
      const f32x4 pr = exp2x4(sv - mnew);
      l[qt] = l[qt] * alpha[qt] + (pr[0] + pr[1]) + (pr[2] + pr[3]);
      f32x4 prd = pr;
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool keep = drop_keep_rk(dseed, drow + (uint32_t)min(qt * 16 + lq, p.Lq - 1), (uint32_t)(kt * 16 + gq * 4 + r), thresh);
          prd[r] = keep ? pr[r] * inv_keep : 0.f;
        }
      }
      pb[qt] = pack4(prd);
    }
    if (__builtin_amdgcn_ballot_w64(moved) != 0) {       // some query's running max moved: rescale (rare after the first tiles)
#pragma unroll
      for (int qt = 0; qt < 3; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oT[dt][qt] *= alpha[qt];
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const s16x4 vT = tr16s(Vt, TRS, dt * 16, gq, lq);
#pragma unroll
      for (int qt = 0; qt < 3; ++qt) oT[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vT, pb[qt], oT[dt][qt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 3; ++qt) l[qt] = rows4_sum(l[qt]);
  // ---- merge the four partials pairwise: (0, 2) and (1, 3), then (0, 1)
  f32x4* slab = reinterpret_cast<f32x4*>(smem + T_MAXK * 4);
  auto park = [&](int s_) {
    f32x4* dst = slab + (size_t)s_ * 14 * 64 + lane;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qt = 0; qt < 3; ++qt) dst[(dt * 3 + qt) * 64] = oT[dt][qt];
    dst[12 * 64] = f32x4{m[0], m[1], m[2], 0.f};
    dst[13 * 64] = f32x4{l[0], l[1], l[2], 0.f};
  };
  auto merge = [&](int s_) {
    const f32x4* src = slab + (size_t)s_ * 14 * 64 + lane;
    const f32x4 mo = src[12 * 64], lo = src[13 * 64];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
      const float mn = fmaxf(m[qt], mo[qt]);
      // (a wave without a key tile carries max = -inf, sum = 0, O = 0: its weight is 0, never exp2(-inf + inf))
      const float a = m[qt] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m[qt] - mn), c = mo[qt] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mo[qt] - mn);
      m[qt] = mn;
      l[qt] = l[qt] * a + lo[qt] * c;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oT[dt][qt] = oT[dt][qt] * a + src[(dt * 3 + qt) * 64] * c;
    }
  };
  __syncthreads();                                       // every wave is done with its V^T image
  if (wave >= 2) park(wave - 2);
  __syncthreads();
  if (wave < 2) merge(wave);
  __syncthreads();
  if (wave == 1) park(0);
  __syncthreads();
  if (wave == 0) {
    merge(0);
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
      const int qi = qt * 16 + lq;
      const size_t row = (size_t)b * p.Lq + min(qi, p.Lq - 1);
      const float inv = 1.f / l[qt];
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        const u32x4 w = rows4_gather16(oT[2 * hp][qt] * inv, oT[2 * hp + 1][qt] * inv);
        if (qi < p.Lq) *reinterpret_cast<u32x4*>(p.o + row * p.ldo + h * 64 + hp * 32 + gq * 8) = w;
      }
      if (qi < p.Lq && gq == 0 && p.lse) p.lse[row * p.H + h] = m[qt] * 0.6931471805599453f + __logf(l[qt]);
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Forward of image -> text cross attention (head_dim 32, at most 48 keys, Lq % 16 == 0, heads % 4 == 0, no dropout): K rows and V^T of a
// (sample, head) are MFMA operands held in registers for the whole run, a wave streams 16-query strips (q in: 16 bytes per lane; o out: 16
// bytes per lane), the whole softmax of a strip is in registers (48 keys = three MFMA tiles).  Workgroup = 4 adjacent heads x 2 halves of the
// strips, as the backward above.  The generic forward moved 576 x 40 x d32 at 2.4 TB/s of its bytes.
struct XF {
  const bf16* q; const bf16* k; const bf16* v; bf16* o; float* lse; const float* kmask;
  int ldq, ldk, ldv, ldo;
  int H, Lq, Lk;
  float scale;
};
constexpr int XF_SMEM = 2 * X_KV + 64 * 4;                 // K / V images of four heads + the key-mask table

__global__ __launch_bounds__(512, 3) void i2t_fwd_kernel(XF p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = reinterpret_cast<bf16*>(smem + X_KV);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gq = lane >> 4, lq = lane & 15;
  const int hl = wave & 3, half = wave >> 2;
  const int b = blockIdx.x, h0 = blockIdx.y * 4, h = h0 + hl;
  float* mkl = reinterpret_cast<float*>(smem + 2 * X_KV);     // key mask of the sample (log2 domain, -inf past Lk): one load per thread
  if (tid < 64) mkl[tid] = tid < p.Lk ? (p.kmask ? p.kmask[(size_t)b * p.Lk + tid] * 1.4426950408889634f : 0.f) : -INFINITY;
  for (int idx = tid; idx < 4 * 64 * 4; idx += 512) {    // K, V of the four heads: row-major images, rows >= Lk zero
    const int hh = idx >> 8, r = (idx >> 2) & 63, c = idx & 3;
    bf16x8 kv, vv;
#pragma unroll
    for (int e = 0; e < 8; ++e) { kv[e] = f2bf(0.f); vv[e] = f2bf(0.f); }
    if (r < p.Lk) {
      const size_t row = (size_t)b * p.Lk + r;
      kv = *reinterpret_cast<const bf16x8*>(p.k + row * p.ldk + (h0 + hh) * 32 + c * 8);
      vv = *reinterpret_cast<const bf16x8*>(p.v + row * p.ldv + (h0 + hh) * 32 + c * 8);
    }
    *reinterpret_cast<bf16x8*>(Ks + (hh * 64 + r) * XRS + c * 8) = kv;
    *reinterpret_cast<bf16x8*>(Vs + (hh * 64 + r) * XRS + c * 8) = vv;
  }
  __syncthreads();
  const bf16* Kh = Ks + hl * 64 * XRS;
  const bf16* Vh = Vs + hl * 64 * XRS;
  bf16x8 kfr[3], vtf[2][2];                              // K rows of key tile kt; V^T of the tile pairs (0, 1), (2, zero rows)
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) kfr[kt] = *reinterpret_cast<const bf16x8*>(Kh + (kt * 16 + lq) * XRS + gq * 8);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) vtf[dt][pr] = trr_frag(Vh, dt * 16, 2 * pr, gq, lq);
  f32x4 mkT[3];                                          // additive key mask in the log2 domain (keys kt*16 + gq*4 + r), -inf past Lk
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) mkT[kt] = *reinterpret_cast<const f32x4*>(mkl + kt * 16 + gq * 4);
  const float c2 = p.scale * 1.4426950408889634f;
  const int nstrips = p.Lq >> 4;
  const bf16* qbase = p.q + (size_t)b * p.Lq * p.ldq + h * 32 + gq * 8;
  bf16* obase = p.o + (size_t)b * p.Lq * p.ldo + h * 32 + gq * 8;
  // strips are requested two ahead in two FIXED register sets (loop unrolled twice: no register moves, so the compiler's vmcnt waits stay counted)
  bf16x8 qa, qb;
  auto strip = [&](int s_, bf16x8& qreg, int s_next) {
    const bf16x8 qf = qreg;
    f32x4 st[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[kt], qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    qreg = *reinterpret_cast<const bf16x8*>(atx(qbase, (unsigned)((s_next < nstrips ? s_next : s_) * 16 + lq) * (unsigned)p.ldq));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 sv[3];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      sv[kt] = __builtin_elementwise_fma(st[kt], f32x4{c2, c2, c2, c2}, mkT[kt]);
      mx = fmaxf(mx, fmaxf(fmaxf(sv[kt][0], sv[kt][1]), fmaxf(sv[kt][2], sv[kt][3])));
    }
    mx = rows4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) { sv[kt] = exp2x4(sv[kt] - mx); sum += (sv[kt][0] + sv[kt][1]) + (sv[kt][2] + sv[kt][3]); }
    sum = rows4_sum(sum);
    const bf16x8 p01 = pack8(sv[0], sv[1]), p2 = pack8(sv[2], f32x4{0.f, 0.f, 0.f, 0.f});
    f32x4 oT[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      oT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtf[dt][0], p01, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      oT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtf[dt][1], p2, oT[dt], 0, 0, 0);
    }
    const float inv = 1.f / sum;
    *reinterpret_cast<u32x4*>(atx(obase, (unsigned)(s_ * 16 + lq) * (unsigned)p.ldo)) = rows4_gather16(oT[0] * inv, oT[1] * inv);
    if (gq == 0 && p.lse) p.lse[((size_t)b * p.Lq + s_ * 16 + lq) * p.H + h] = mx * 0.6931471805599453f + __logf(sum);
  };
  if (half < nstrips) qa = *reinterpret_cast<const bf16x8*>(atx(qbase, (unsigned)(half * 16 + lq) * (unsigned)p.ldq));
  if (half + 2 < nstrips) qb = *reinterpret_cast<const bf16x8*>(atx(qbase, (unsigned)((half + 2) * 16 + lq) * (unsigned)p.ldq));
  for (int s_ = half; s_ < nstrips; s_ += 4) {
    strip(s_, qa, s_ + 4);
    if (s_ + 2 < nstrips) strip(s_ + 2, qb, s_ + 6);
  }
}

bool x_attr[16] = {};            // per device: function attributes are per-device state

}  // namespace

// One-pass backward of image -> text cross attention; returns FIBER_EINVAL when the shape is not served (the caller then runs the generic
// passes).  Arguments as fiber_mha_bwd_bf16 (attn.hip); no attention dropout.
int fiber_i2t_bwd_launch(const void* q, const void* k, const void* v, const float* kmask, const void* o, const void* dout, const float* lse,
                         void* dq, void* dk, void* dv, int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int lddo,
                         int lddq, int lddk, int lddv, float scale, hipStream_t st) {
  if (Lk > 48 || Lk <= 0 || (Lq & 15) || (heads & 3) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (lddo & 7) || (lddq & 7) || (lddk & 3) ||
      (lddv & 3) || scale <= 0.f)
    return FIBER_EINVAL;
  if (fiber_first_on_device(x_attr)) hipFuncSetAttribute((const void*)i2t_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  XP p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (const bf16*)o; p.dout = (const bf16*)dout;
  p.dq = (bf16*)dq; p.dk = (bf16*)dk; p.dv = (bf16*)dv; p.lse = lse; p.kmask = kmask;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.H = heads; p.Lq = Lq; p.Lk = Lk; p.G = B; p.scale = scale;
  hipLaunchKernelGGL(i2t_bwd_kernel, dim3(B, heads / 4), dim3(512), X_SMEM, st, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// One-pass backward for head_dim 64 and at most 48 queries (text -> image cross attention, text self attention), attention dropout
// included; FIBER_EINVAL when the shape is not served.
int fiber_t2i_bwd_launch(const void* q, const void* k, const void* v, const float* kmask, const void* o, const void* dout, const float* lse,
                         void* dq, void* dk, void* dv, int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int lddo,
                         int lddq, int lddk, int lddv, float scale, float p_drop, uint64_t seed, const uint64_t* seed_base, hipStream_t st) {
  if (Lq > 48 || Lq <= 0 || Lk <= 0 || Lk > T_MAXK || (size_t)Lk * (size_t)(ldk | ldv | lddk | lddv) >= (1u << 30) || ((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 7) || scale <= 0.f || p_drop < 0.f || p_drop >= 1.f)
    return FIBER_EINVAL;
  TP p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (const bf16*)o; p.dout = (const bf16*)dout;
  p.dq = (bf16*)dq; p.dk = (bf16*)dk; p.dv = (bf16*)dv; p.lse = lse; p.kmask = kmask;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.H = heads; p.Lq = Lq; p.Lk = Lk; p.scale = scale; p.p_drop = p_drop; p.seed = seed; p.seed_base = seed_base;
  if (p_drop > 0.f) hipLaunchKernelGGL(t2i_bwd_kernel<true>, dim3(B, heads), dim3(256), T_SMEM, st, p);
  else hipLaunchKernelGGL(t2i_bwd_kernel<false>, dim3(B, heads), dim3(256), T_SMEM, st, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Forward for head_dim 64 and at most 48 queries (text -> image cross attention, text self attention); FIBER_EINVAL: shape not served.
int fiber_t2i_fwd_launch(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int heads, int Lq, int Lk,
                         int ldq, int ldk, int ldv, int ldo, float scale, float p_drop, uint64_t seed, const uint64_t* seed_base, hipStream_t st) {
  if (Lq > 48 || Lq <= 0 || Lk <= 0 || Lk > T_MAXK || (size_t)Lk * (size_t)(ldk | ldv) >= (1u << 30) || ((ldq | ldk | ldv | ldo) & 7) ||
      scale <= 0.f || p_drop < 0.f || p_drop >= 1.f)
    return FIBER_EINVAL;
  FP p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = lse; p.kmask = kmask;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.H = heads; p.Lq = Lq; p.Lk = Lk; p.scale = scale; p.p_drop = p_drop; p.seed = seed; p.seed_base = seed_base;
  if (p_drop > 0.f) hipLaunchKernelGGL(t2i_fwd_kernel<true>, dim3(B, heads), dim3(256), F_SMEM, st, p);
  else hipLaunchKernelGGL(t2i_fwd_kernel<false>, dim3(B, heads), dim3(256), F_SMEM, st, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Forward of image -> text cross attention (head_dim 32, <= 48 keys, no dropout); FIBER_EINVAL: shape not served.
int fiber_i2t_fwd_launch(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int heads, int Lq, int Lk,
                         int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st) {
  if (Lk > 48 || Lk <= 0 || (Lq & 15) || Lq <= 0 || (heads & 3) || ((ldq | ldk | ldv | ldo) & 7) || (size_t)Lq * (size_t)(ldq | ldo) >= (1u << 30) ||
      scale <= 0.f)
    return FIBER_EINVAL;
  XF p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = lse; p.kmask = kmask;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.H = heads; p.Lq = Lq; p.Lk = Lk; p.scale = scale;
  hipLaunchKernelGGL(i2t_fwd_kernel, dim3(B, heads / 4), dim3(512), XF_SMEM, st, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
