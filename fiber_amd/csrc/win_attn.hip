// Swin (shifted-)window attention, specialised + persistent (gfx950 / CDNA4).  head_dim 32, N = ws*ws <= 336.
//
// Same math and MFMA formulation as attn.hip's WINDOW mode (swin_transformer.py:195-219 with roll / partition /
// reverse :99-126,364-387 folded into addressing, bias gather :208-211, shift mask :327-350), restructured for
// throughput after rocprof showed the generic kernel latency-bound (one 9-wave workgroup per CU, exposed global-load
// latency per window, two integer divisions per score for the bias index):
//   * a workgroup is PERSISTENT over a run of windows of one head: per-head bias table, key (row,col) offsets and all
//     per-lane geometry are computed once; per window only K/V (or Q/dO) tiles are re-staged;
//   * the next window's tiles and per-lane Q/dO fragments are PREFETCHED into registers (global loads in flight)
//     while the MFMAs / softmax of the current window run, and written to LDS after the barrier (T14 split staging);
//   * bias index = qoff(i) - keyoff(j): one LDS int4 read per 4 scores + one LDS float read per score, no divisions;
//   * the shift mask is evaluated only for windows that touch the wrapped border (last window row / column);
//   * LDS holds ROW-MAJOR images only (96-B row stride, conflict-free for both access patterns); operands an MFMA needs
//     transposed are read with ds_read_b64_tr_b16;
//   * the kernels are VALU-issue bound (9 waves on 4 SIMDs, ~4.3 cycles per VALU wave-instruction), so bias, -lse/scale and
//     -delta enter as MFMA accumulator seeds and each score costs one packed fma + one v_exp_f32.
// Backward: pass A (wave = query strip) -> dQ and the relative-position-bias gradient, accumulated in registers over
// all windows the workgroup visits; pass B (wave = key strip) -> dK, dV.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

// Tile capacity is a template parameter MTT of every kernel / helper below: 10 tiles (N <= 160: the 12x12 windows of the 384^2
// configurations, bias slice held in registers) or 21 tiles (N <= 336: the 18x18 windows of the 576^2 configuration, bias
// looked up from LDS per score -- 84 more registers per lane do not exist).  MT / MTP / MAXN are derived locally from it.
// Row stride (elements) of the row-major LDS tiles: 32 + 16 pad = 96 B = 24 banks.  Both read patterns are then conflict-free
// (MI355X_MICROARCH.md LDS table): ds_read_b128 serves lanes {0-3,12-15,20-27}... per cycle, whose 16 row starts 24*row + 4*g
// tile the 64 banks, and ds_read_b64_tr_b16 serves 32 lanes = 8 rows x 32 B per cycle, 24*row mod 64 being the 8 multiples
// of 8.  The former 80-B stride was 2-way conflicted on both (SQ_LDS_BANK_CONFLICT = 43 % of SQ_LDS_IDX_ACTIVE).
constexpr int RS = 48;
// (tiles are consumed in pairs by the second MFMA of each pass: an odd capacity is rounded up for the pair loops, the
// per-tile register arrays and the image height, the extra tile being all zeros)
#define WIN_DIMS(MTT) constexpr int MT = (MTT), MTP = ((MTT) + 1) & ~1, MAXN = MTP * 16; (void)MT; (void)MTP; (void)MAXN

#ifdef FIBER_WIN_TRACE
__device__ float g_win_trace[256];
#endif

struct WinP {
  const bf16* qkv; bf16* o; const bf16* dout; bf16* dqkv;
  // lse, delta: [image][head][token] (round 5; [token][head] made every strip's 16 values 16 scattered 4-byte requests); delta: written by the
  // dQ pass, read by the dK/dV pass
  float* lse; float* delta;
  const float* bias_table; float* dbias_part;
  float* colsum_part;   // optional [gridDim.x * gridDim.z, 3C] fp32: per-workgroup column sums of dqkv (bias gradient of the qkv linear)
  int B, Hres, Wres, C, heads, ws, shift, nWw, nWh, nW, G, N, gpb;
  // qkv channel layout: 0 = [3][heads][32] (reference, swin_transformer.py:202); 1 = [heads][3][32] (q|k|v of a head adjacent: one 192-byte run);
  // 2 = [heads][32] q, then [heads][2][32] k|v: the k and v rows of a (token, head) are ONE aligned 128-byte line
  int hmajor;
};
__device__ __forceinline__ int chan_q(const WinP& p, int h) { return p.hmajor == 1 ? h * 96 : h * 32; }
__device__ __forceinline__ int chan_k(const WinP& p, int h) { return p.hmajor == 1 ? h * 96 + 32 : p.hmajor == 2 ? p.C + h * 64 : p.C + h * 32; }
__device__ __forceinline__ int chan_v(const WinP& p, int h) { return p.hmajor == 1 ? h * 96 + 64 : p.hmajor == 2 ? p.C + h * 64 + 32 : 2 * p.C + h * 32; }

__device__ __forceinline__ int region_of(int x, int n, int ws, int shift) { return x < n - ws ? 0 : (x < n - shift ? 1 : 2); }
__device__ __forceinline__ float g4max(float v) { return rows4_max(v); }
__device__ __forceinline__ float g4sum(float v) { return rows4_sum(v); }
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = f2bf(a[e]); o[4 + e] = f2bf(b[e]); }
  return o;
}
// MFMA operand A (transposed fragment) out of a ROW-MAJOR [token][32] image (row stride RS): rows {t0*16+g*4..+3} U
// {t0*16+16+g*4..+3}, column d0 + l.  ds_read_b64_tr_b16 hands lane l of a 16-lane group column l of the 4x16 block whose
// (row l>>2, 4-column piece l&3) address that lane supplies, so a transposed copy of the image never has to be written
// (8 ds_write_b16 per staged chunk less).
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 trr_frag(const bf16* rm, int d0, int t0, int g, int l) {
  const bf16* a = rm + (t0 * 16 + g * 4 + (l >> 2)) * RS + d0 + (l & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 16 * RS));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
  return __builtin_bit_cast(bf16x8, o);
}
// max of three without the canonicalising v_max_f32 x, x that fmaxf() drags in for every operand (inputs are never sNaN here)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// element-wise helpers on f32x4 with the natural (0,1)(2,3) pairing -> v_pk_fma_f32 / v_pk_mul_f32 and aligned v_cvt_pk_bf16_f32
// (the SLP vectoriser otherwise pairs lanes (1,2) and pays v_alignbit/v_perm shuffles before every pack)
__device__ __forceinline__ f32x4 fma4(f32x4 a, float b, f32x4 c) { return __builtin_elementwise_fma(a, f32x4{b, b, b, b}, c); }
__device__ __forceinline__ f32x4 fma4(f32x4 a, float b, float c) { return __builtin_elementwise_fma(a, f32x4{b, b, b, b}, f32x4{c, c, c, c}); }
__device__ __forceinline__ f32x4 exp2x4(f32x4 a) {
  return f32x4{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1]), __builtin_amdgcn_exp2f(a[2]), __builtin_amdgcn_exp2f(a[3])};
}
// shift-mask term: -100 (natural-log logits) on keys of another shift region, as arithmetic on float region labels:
// pen * (region_a - region_b)^2 is 0 when equal and <= pen otherwise (masked probabilities underflow to 0 either way)
__device__ __forceinline__ f32x4 region_mask(f32x4 sv, f32x4 ra, float rb, float pen) {
  const f32x4 d = ra - rb;
  return __builtin_elementwise_fma(d * d, f32x4{pen, pen, pen, pen}, sv);
}
__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = f2bf(0.f);
  return z;
}

// uniform base pointer + 32-bit BYTE offset: the form the backend turns into `global_load/store v, voff, s[base:base+1]`
template <typename T>
__device__ __forceinline__ T* at(T* base, unsigned elem_off) {
  using C = std::conditional_t<std::is_const<T>::value, const char, char>;
  return reinterpret_cast<T*>(reinterpret_cast<C*>(base) + elem_off * (unsigned)sizeof(T));
}

// window g, window-local (pr,pc) -> token row of the [B*H*W] tensors and shift-mask region label
struct Geo {
  int b, wr, wc;
  bool border;
  __device__ __forceinline__ void set(const WinP& p, int g) {
    b = g / p.nW;
    const int w = g - b * p.nW;
    wr = w / p.nWw;
    wc = w - wr * p.nWw;
    border = p.shift > 0 && (wr == p.nWh - 1 || wc == p.nWw - 1);
  }
  // advance to window g+1 without integer divisions (the loop visits consecutive windows)
  __device__ __forceinline__ void next(const WinP& p) {
    if (++wc == p.nWw) { wc = 0; if (++wr == p.nWh) { wr = 0; ++b; } }
    border = p.shift > 0 && (wr == p.nWh - 1 || wc == p.nWw - 1);
  }
  // Token row = img() + pix(): the image base is wave-uniform (scalar 64-bit pointer arithmetic), the pixel offset is a
  // 32-bit per-lane value, so every global access is `saddr + 32-bit voffset` and no lane carries 64-bit addresses.
  __device__ __forceinline__ unsigned pix(const WinP& p, int pr, int pc) const {
    int r = wr * p.ws + pr + p.shift, c = wc * p.ws + pc + p.shift;
    r = (int)min((unsigned)r, (unsigned)(r - p.Hres));   // r >= Hres ? r - Hres : r as one v_min_u32 (no compare + select)
    c = (int)min((unsigned)c, (unsigned)(c - p.Wres));
    return (unsigned)(r * p.Wres + c);
  }
  __device__ __forceinline__ size_t img(const WinP& p) const { return (size_t)b * p.Hres * p.Wres; }
  __device__ __forceinline__ int reg(const WinP& p, int pr, int pc) const {
    return region_of(wr * p.ws + pr, p.Hres, p.ws, p.shift) * 3 + region_of(wc * p.ws + pc, p.Wres, p.ws, p.shift);
  }
};

struct Smem {
  int* koff; int* kreg; float* btab; float* lse; float* dlt;
  bf16* a0; bf16* a1;
};
template <int MTT>
__device__ __forceinline__ Smem carve(char* base, int nb, int n_rm) {
  WIN_DIMS(MTT);
  Smem S;
  S.koff = reinterpret_cast<int*>(base);
  S.kreg = S.koff + MAXN;
  S.lse = reinterpret_cast<float*>(S.kreg + MAXN);
  S.dlt = S.lse + MAXN;
  S.btab = S.dlt + MAXN;
  bf16* img = reinterpret_cast<bf16*>(base + (size_t)(4 * MAXN + ((nb + 3) & ~3)) * 4);
  S.a0 = img; img += (n_rm > 0) * MAXN * RS;
  S.a1 = img; img += (n_rm > 1) * MAXN * RS;
  return S;
}
template <int MTT>
size_t smem_bytes(int nb, int n_rm) {
  WIN_DIMS(MTT);
  return (size_t)(4 * MAXN + ((nb + 3) & ~3)) * 4 + (size_t)n_rm * MAXN * RS * 2;
}

// Column sums of a backward pass's outputs (the bias gradient of the qkv linear), produced where the values already are instead
// of by a second pass over dqkv: each lane adds its 4-channel pieces into private LDS slots once per window (ds_read_b128 +
// add + ds_write_b128 on the thread's own slots: the register file is full; LDS float atomics measured 2.3x slower for the
// whole backward), and at the end of the kernel 32 threads per 32-channel group sum the slots over waves and the 16 query /
// key lanes and write one fp32 row per workgroup; a fold kernel adds the rows.  Thread t = wave*64 + g*16 + l owns channels
// dt*16 + g*4 + 0..3 of query / key l; slot(piece k = grp*2 + dt, thread t) = slots[k * NTH + t] (f32x4), NTH = the kernel's
// launch bound: lane-contiguous (conflict-free) and one address register + immediate offsets.
template <int NV, int NTH>
__device__ __forceinline__ void colsum_zero(f32x4* slots) {
#pragma unroll
  for (int k = 0; k < NV; ++k) slots[k * NTH + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int NV, int NTH>
__device__ __forceinline__ void colsum_flush(const f32x4* slots, float* dst_row, const int* chan0 /*[NV/2]*/) {
  __syncthreads();
  const int t = threadIdx.x, nwaves = blockDim.x >> 6;
  if (t < 32 * (NV / 2)) {
    const int grp = t >> 5, c = t & 31, dt = c >> 4, g = (c >> 2) & 3, r = c & 3;
    const f32x4* col = slots + (grp * 2 + dt) * NTH + g * 16;
    float s = 0.f;
    for (int w = 0; w < nwaves; ++w)
      for (int l = 0; l < 16; ++l) s += col[w * 64 + l][r];
    dst_row[chan0[grp] + c] = s;
  }
}

// common per-block setup: bias column of this head, key offsets, zeroed LDS tiles (padding rows stay zero forever)
template <int MTT>
__device__ __forceinline__ void setup(const WinP& p, const Smem& S, int h, int nb, int n_rm, float bias_mul, float lse_valid) {
  WIN_DIMS(MTT);
  // bias column of this head, pre-multiplied into the domain the kernel adds it in (bias/scale or bias*log2e)
  for (int t = threadIdx.x; t < nb; t += blockDim.x) S.btab[t] = p.bias_table[(size_t)t * p.heads + h] * bias_mul;
  for (int j = threadIdx.x; j < MAXN; j += blockDim.x) {
    const int jj = j < p.N ? j : 0;
    const int pr = jj / p.ws, pc = jj - pr * p.ws;
    S.koff[j] = pr * (2 * p.ws - 1) + pc;
    S.kreg[j] = 0;
    // dK/dV pass: -lse/scale of each query is staged here per window (accumulator seed), padded queries keep -inf -> p = 0;
    // forward / dQ pass: the same table is the additive key-padding mask (0 on valid keys, -inf on padded ones)
    S.lse[j] = j < p.N ? lse_valid : -INFINITY;
    S.dlt[j] = 0.f;
  }
  const int words = n_rm * MAXN * RS / 2;
  uint32_t* z = reinterpret_cast<uint32_t*>(S.a0);
  for (int t = threadIdx.x; t < words; t += blockDim.x) z[t] = 0u;
}

// ================================================================ forward =====================================
// MAXC: 16-byte staging chunks per thread; NTC: key/query tiles if known at compile time (9 for 12x12 windows);
// MTT: tile capacity (10 / 21); BREG: window-invariant bias slice in registers (else LDS look-ups per score)
template <int MAXC, int NTC = 0, int MTT = 10, bool BREG = true>
__global__ __launch_bounds__(MAXC == 1 ? 640 : 448) void win_fwd_kernel(WinP p) {
  WIN_DIMS(MTT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = (2 * p.ws - 1) * (2 * p.ws - 1);
  Smem S = carve<MTT>(smem, nb, 2);
  bf16* Ks = S.a0; bf16* Vs = S.a1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gq = lane >> 4, lq = lane & 15;
  const int h = blockIdx.y, C = p.C;
  // layout 3 = planar: [3 * heads planes][token][32] -- a window row of a head is ONE 768-byte run (tools/probes/run_probe.hip)
  // layout 4: qkv planar, o in token rows (the probe of what a planar qkv GEMM epilogue ALONE would buy)
  const bool planar = p.hmajor == 3 || p.hmajor == 4, planar_o = p.hmajor == 3;
  const size_t pl = (size_t)p.B * p.Hres * p.Wres * 32;
  const int ld = planar ? 32 : 3 * C, ldo = planar_o ? 32 : C;
  const size_t qo = planar ? h * pl : (size_t)chan_q(p, h), ko = planar ? (p.heads + h) * pl : (size_t)chan_k(p, h),
               vo = planar ? (2 * p.heads + h) * pl : (size_t)chan_v(p, h), oo = planar_o ? h * pl : (size_t)h * 32;
  // compile-time for the 12x12 window (guards fold, the 10th tile's code disappears) and for the 21-tile variant (18x18
  // windows fill it; padded tiles of smaller windows carry bias = -inf and zero rows, so running them is only wasted work)
  const int ntile = NTC ? NTC : MT > 10 ? MT : (p.N + 15) >> 4;
  setup<MTT>(p, S, h, nb, 2, 5.656854249492381f, 0.f);

  // this thread's staging chunks: chunk id = tid + c*blockDim -> (row = id>>2 of the window, 16-byte piece id&3)
  // (Round-6 experiment, removed: the three waves of SIMD 0 -- nine waves sit 3 / 2 / 2 / 2 on the SIMDs -- staging nothing, their chunks given to
  // the six waves of SIMDs 1-3: bit-identical, 0-1 % faster at 12 more registers, profiles/r06_win_fwd_s0f_ab.log.  The forward is not bound by
  // SIMD 0's issue slots, whatever the per-wave s_memtime trace suggests.)
  int spr[MAXC], spc[MAXC];
  bool sval[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int id = tid + c * blockDim.x;
    sval[c] = (id >> 2) < p.N;
    const int r = sval[c] ? (id >> 2) : 0;
    spr[c] = r / p.ws; spc[c] = r - spr[c] * p.ws;
  }
  // this lane's query: workgroup z owns strips [z*nw, (z+1)*nw) so that several small workgroups share a CU
  const int i = (blockIdx.z * (blockDim.x >> 6) + wave) * 16 + lq;
  const bool qval = i < p.N;
  const int ic = qval ? i : p.N - 1;
  const int qpr = ic / p.ws, qpc = ic - qpr * p.ws;
  const int qoff = (qpr + p.ws - 1) * (2 * p.ws - 1) + qpc + p.ws - 1;
  const float scale = 0.17677669529663687f;

  const int g0 = blockIdx.x * p.gpb, g1 = min(p.G, g0 + p.gpb);
  if (g0 >= g1) return;
  // The wave owns the same query strip in every window it visits, so its slice of the relative-position bias
  // (bias_table[rel_index(i, j)], swin_transformer.py:208-211) is window-invariant: gather it ONCE into registers
  // (-inf for padded keys), pre-divided by the scale, and hand it to the QK^T MFMA as its accumulator seed.
  __syncthreads();
  f32x4 breg[BREG ? MT : 1];
  // bias / scale of key tile kt for this lane's query, -inf on padded keys
  auto bias_tile = [&](int kt) -> f32x4 {
    if constexpr (BREG) {
      return breg[kt];
    } else {
      // branch-free: padded keys look up entry koff = 0 and get -inf from the additive padding mask
      const int4 ko = *reinterpret_cast<const int4*>(S.koff + kt * 16 + gq * 4);
      const f32x4 b = {S.btab[qoff - ko.x], S.btab[qoff - ko.y], S.btab[qoff - ko.z], S.btab[qoff - ko.w]};
      return b + *reinterpret_cast<const f32x4*>(S.lse + kt * 16 + gq * 4);
    }
  };
  if constexpr (BREG) {
#pragma unroll
    for (int kt = 0; kt < MT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = kt * 16 + gq * 4 + r;
        breg[kt][r] = (kt < ntile && jj < p.N) ? S.btab[qoff - S.koff[jj]] : -INFINITY;
      }
  }
  const float scale2 = scale * 1.4426950408889634f;     // scores kept in the log2 domain: exp is a bare v_exp_f32
  Geo geo;
  geo.set(p, g0);
  bf16x8 kr[MAXC], vr[MAXC], qn;
  unsigned qpix = geo.pix(p, qpr, qpc);
  size_t qimg = geo.img(p);
  auto prefetch = [&]() {
    const bf16* base = p.qkv + qimg * ld;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (sval[c]) {
        const unsigned sc = (tid + c * blockDim.x) & 3;
        const unsigned st = geo.pix(p, spr[c], spc[c]) * ld + sc * 8;
        kr[c] = *reinterpret_cast<const bf16x8*>(at(base + ko, st));
        vr[c] = *reinterpret_cast<const bf16x8*>(at(base + vo, st));
      }
    }
    qn = *reinterpret_cast<const bf16x8*>(at(base + qo, qpix * ld + gq * 8));
  };
  prefetch();
#ifdef FIBER_WIN_TRACE      // tools/win_trace.py fwd: s_memtime ticks per segment and wave into g_win_trace
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define FWD_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tr[i] += t_ - t0; t0 = t_; } while (0)
#else
#define FWD_MARK(i) do { } while (0)
#endif
  for (int g = g0; g < g1; ++g) {
    __syncthreads();                                  // previous window's LDS reads done (also covers setup)
    FWD_MARK(0);
#ifdef FIBER_WIN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    FWD_MARK(1);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (sval[c]) {
        const int id = tid + c * blockDim.x, sr = id >> 2, sc = id & 3;
        *reinterpret_cast<bf16x8*>(Ks + sr * RS + sc * 8) = kr[c];
        *reinterpret_cast<bf16x8*>(Vs + sr * RS + sc * 8) = vr[c];
      }
    }
    const bool border = geo.border;
    float* kregf = reinterpret_cast<float*>(S.kreg);     // shift-region labels as floats (padding rows stay 0)
    if (border) for (int t = tid; t < p.N; t += blockDim.x) { const int pr = t / p.ws; kregf[t] = (float)geo.reg(p, pr, t - pr * p.ws); }
    const float qregf = border ? (float)geo.reg(p, qpr, qpc) : 0.f;
    const bf16x8 qf = qn;
    const unsigned opix = qpix;
    const size_t oimg = qimg;
    FWD_MARK(2);
    __syncthreads();
    FWD_MARK(3);
    if (g + 1 < g1) {                                  // prefetch next window while this one computes
      geo.next(p);
      qpix = geo.pix(p, qpr, qpc);
      qimg = geo.img(p);
      prefetch();
    }
    FWD_MARK(4);
    // Only windows on the wrapped border pay for the region mask (one wave-uniform branch, swin_transformer.py:327-350);
    // padded key tiles carry bias = -inf so exp2 gives exact zeros without per-element selects.
    f32x4 s[MTP];
    float mx = -INFINITY, sum;
    f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    {
      // Explicit software pipeline: the compiler's own schedule put every ds_read right before the MFMA that consumes it
      // (s_waitcnt lgkmcnt(0) per tile), and with ~2 waves per SIMD nothing hid that latency.  All K fragments are
      // requested first, the MT independent QK^T MFMAs issue back to back, and the V^T fragments are requested before the
      // exp2 pass so that they land while the VALU works.  Padded tiles read zero rows (harmless) and carry bias = -inf.
      constexpr int GK = (NTC || MT > 10) ? 5 : 3, NG = (MT + GK - 1) / GK;        // K fragments in flight: GK tiles ahead of the MFMAs
      constexpr int VE = MT > 10 ? 0 : (NTC ? 3 : 1);                // V^T tile pairs requested before the exp2 pass
      bf16x8 kf[2][GK];
#pragma unroll
      for (int u = 0; u < GK; ++u) kf[0][u] = *reinterpret_cast<const bf16x8*>(Ks + (u * 16 + lq) * RS + gq * 8);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        __builtin_amdgcn_sched_barrier(0);
        if (gi + 1 < NG)
#pragma unroll
          for (int u = 0; u < GK; ++u)
            if ((gi + 1) * GK + u < MT) kf[(gi + 1) & 1][u] = *reinterpret_cast<const bf16x8*>(Ks + (((gi + 1) * GK + u) * 16 + lq) * RS + gq * 8);
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          const int kt = gi * GK + u;
          if (kt < MT && kt < ntile) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[gi & 1][u], qf, bias_tile(kt), 0, 0, 0);   // S^T[key][query] + bias / scale
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 vf[MTP / 2][2];
#pragma unroll
      for (int t2 = 0; t2 < VE; ++t2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[t2][dt] = trr_frag(Vs, dt * 16, 2 * t2, gq, lq);
      __builtin_amdgcn_sched_barrier(0);
      // VALU diet (a plain VALU op is ~4.3 cycles per wave-instruction on a SIMD, v_exp_f32 / v_rcp_f32 8.4): the softmax works on
      // t = q.k + bias/scale straight out of the MFMA; p = exp2(t * scale*log2e - max) is ONE packed fma + exp2 per score.
      if (border) {                                       // one wave-uniform branch per window (swin_transformer.py:327-350)
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
          if (kt < ntile) {
            s[kt] = region_mask(s[kt], *reinterpret_cast<const f32x4*>(kregf + kt * 16 + gq * 4), qregf, -565.6854249492381f);   // -100 / scale
          }
        }
      }
      float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < MT; ++kt)
        if (kt < ntile) { m0 = max3(m0, s[kt][0], s[kt][1]); m1 = max3(m1, s[kt][2], s[kt][3]); }
      mx = g4max(max3(m0, m1, m1));
      const float nm = -mx * scale2;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int kt = 0; kt < MTP; ++kt) {
        if (kt < ntile) {
          const f32x2 e0 = __builtin_elementwise_fma(f32x2{s[kt][0], s[kt][1]}, f32x2{scale2, scale2}, f32x2{nm, nm});
          const f32x2 e1 = __builtin_elementwise_fma(f32x2{s[kt][2], s[kt][3]}, f32x2{scale2, scale2}, f32x2{nm, nm});
          s[kt] = f32x4{__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y), __builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y)};
          s0 += s[kt][0] + s[kt][1];
          s1 += s[kt][2] + s[kt][3];
        } else {
          s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      sum = g4sum(s0 + s1);
#pragma unroll
      for (int t2 = 0; t2 < MTP / 2; ++t2) {
        if (t2 + VE < MTP / 2) {                           // keep VE pairs of V^T fragments ahead of the PV MFMAs
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vf[t2 + VE][dt] = trr_frag(Vs, dt * 16, 2 * (t2 + VE), gq, lq);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t2 * 2 < ntile) {
          const bf16x8 pf = pack8(s[2 * t2], s[2 * t2 + 1]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t2][dt], pf, oacc[dt], 0, 0, 0);
        }
      }
    }
    FWD_MARK(5);
    if (qval) {
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(oacc[dt][r] * inv);
        *reinterpret_cast<bf16x4*>(at(p.o + oo + oimg * ldo, opix * ldo + dt * 16 + gq * 4)) = o;
      }
      if (gq == 0) *at(p.lse + oimg * p.heads + (size_t)h * p.Hres * p.Wres, opix) = mx * scale + __logf(sum);
    }
    FWD_MARK(6);
  }
#ifdef FIBER_WIN_TRACE
  if (lane == 0 && blockIdx.x < 2 && h == 0 && blockIdx.z == 0) {
    float* o = g_win_trace + (blockIdx.x * 10 + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = (float)tr[i];
    if (wave == 0) g_win_trace[200 + blockIdx.x] = (float)(g1 - g0);
  }
#endif
#undef FWD_MARK
}

// ================================================================ backward pass A: dQ + dbias =================
// MAXC: 16-byte staging chunks per thread; NTC: key/query tiles if known at compile time (9 for 12x12 windows);
// MTT: tile capacity (10 / 21); BREG: window-invariant bias slice in registers (else LDS look-ups per score)
// The run-time-tile-count instance (NTC = 0, MAXC = 1: windows of at most 128 tokens, e.g. the 7 x 7 windows of Swin-T) is bounded at 8 waves:
// under the 10-wave bound's 168 registers it left 92 bytes of scratch per lane.  NTH stays the slot stride of the LDS column-sum slots.
template <int MAXC, int NTC = 0, int MTT = 10, bool BREG = true>
__global__ __launch_bounds__(MAXC == 1 ? (NTC ? 640 : 512) : 448) void win_bwd_dq_kernel(WinP p) {
  WIN_DIMS(MTT);
  constexpr int NTH = MAXC == 1 ? 640 : 448;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = (2 * p.ws - 1) * (2 * p.ws - 1);
  Smem S = carve<MTT>(smem, nb, 2);
  bf16* Ks = S.a0; bf16* Vs = S.a1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gq = lane >> 4, lq = lane & 15;
  const int h = blockIdx.y, C = p.C, ld = 3 * C;
  const int qo = chan_q(p, h), ko = chan_k(p, h), vo = chan_v(p, h);
  // compile-time for the 12x12 window (guards fold, the 10th tile's code disappears) and for the 21-tile variant (18x18
  // windows fill it; padded tiles of smaller windows carry bias = -inf and zero rows, so running them is only wasted work)
  const int ntile = NTC ? NTC : MT > 10 ? MT : (p.N + 15) >> 4;
  setup<MTT>(p, S, h, nb, 2, 5.656854249492381f, 0.f);
  int spr[MAXC], spc[MAXC];
  bool sval[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int id = tid + c * blockDim.x;
    sval[c] = (id >> 2) < p.N;
    const int r = sval[c] ? (id >> 2) : 0;
    spr[c] = r / p.ws; spc[c] = r - spr[c] * p.ws;
  }
  const int i = (blockIdx.z * (blockDim.x >> 6) + wave) * 16 + lq;
  const bool qval = i < p.N;
  const int ic = qval ? i : p.N - 1;
  const int qpr = ic / p.ws, qpc = ic - qpr * p.ws;
  const int qoff = (qpr + p.ws - 1) * (2 * p.ws - 1) + qpc + p.ws - 1;
  const float scale = 0.17677669529663687f;

  // BREG: this lane's window-invariant bias slice lives in an LDS slab ([wave][tile][lane] x f32x4, one conflict-free
  // ds_read_b128 per tile) -- this kernel also carries the dbias accumulators, and 36 more registers for the slice left
  // nothing for keeping LDS operands in flight.
  f32x4* slab = reinterpret_cast<f32x4*>(S.a1 + MAXN * RS) + wave * MT * 64 + lane;
  // column-sum slots of this kernel's dQ values (after the bias slab when there is one)
  constexpr int NLD = MTT > 10 ? 8 : 0;                   // dbias tiles kept in LDS slots (see below), in front of the column-sum slots
  f32x4* cs_all = reinterpret_cast<f32x4*>(S.a1 + MAXN * RS) + (BREG ? (blockDim.x >> 6) * MT * 64 : 0) + NLD * NTH;
  if (p.colsum_part) colsum_zero<2, NTH>(cs_all);
  // dbias accumulators: one f32x4 per key tile and lane, summed over the windows of the run.  With 21 tiles (18 x 18 windows) 84 registers
  // of them left 156 bytes of scratch per lane; the first NLD tiles live in private LDS slots instead (ds_read_b128 + add + ds_write_b128
  // on the thread's own slot, lane-contiguous), in front of the column-sum slots.
  f32x4* dbl = cs_all - NLD * NTH + threadIdx.x;
  f32x4 dbacc[MTP - NLD];
  auto bias_tile = [&](int kt) -> f32x4 {                 // bias / scale (the seed of the q.k accumulator), -inf on padded keys
    if constexpr (BREG) {
      return slab[kt * 64];
    } else {
      // branch-free: padded keys look up entry koff = 0 and get -inf from the additive padding mask
      const int4 ko = *reinterpret_cast<const int4*>(S.koff + kt * 16 + gq * 4);
      const f32x4 b = {S.btab[qoff - ko.x], S.btab[qoff - ko.y], S.btab[qoff - ko.z], S.btab[qoff - ko.w]};
      return b + *reinterpret_cast<const f32x4*>(S.lse + kt * 16 + gq * 4);
    }
  };
  __syncthreads();
#pragma unroll
  for (int kt = 0; kt < MT; ++kt) {
    if (kt < NLD) dbl[kt * NTH] = f32x4{0.f, 0.f, 0.f, 0.f}; else dbacc[kt < NLD ? 0 : kt - NLD] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (BREG) {
      f32x4 b;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = kt * 16 + gq * 4 + r;
        b[r] = (kt < ntile && jj < p.N) ? S.btab[qoff - S.koff[jj]] : -INFINITY;   // window-invariant bias slice
      }
      slab[kt * 64] = b;                                  // read back by the same lane only: no barrier needed
    }
  }

  const int g0 = blockIdx.x * p.gpb, g1 = min(p.G, g0 + p.gpb);
  Geo geo;
  bf16x8 kr[MAXC], vr[MAXC], qn = zero8(), don = zero8(), on = zero8();
  float lsen = 0.f;
  unsigned qpix = 0;
  size_t qimg = 0;
  auto prefetch = [&]() {
    const bf16* base = p.qkv + qimg * ld;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (sval[c]) {
        const unsigned sc = (tid + c * blockDim.x) & 3;
        const unsigned st = geo.pix(p, spr[c], spc[c]) * ld + sc * 8;
        kr[c] = *reinterpret_cast<const bf16x8*>(at(base, st + ko));
        vr[c] = *reinterpret_cast<const bf16x8*>(at(base, st + vo));
      }
    }
    qn = *reinterpret_cast<const bf16x8*>(at(base, qpix * ld + qo + gq * 8));
    don = *reinterpret_cast<const bf16x8*>(at(p.dout + qimg * C, qpix * C + h * 32 + gq * 8));
    on = *reinterpret_cast<const bf16x8*>(at(static_cast<const bf16*>(p.o) + qimg * C, qpix * C + h * 32 + gq * 8));
    lsen = *at(p.lse + qimg * p.heads + (size_t)h * p.Hres * p.Wres, qpix);
  };
  if (g0 < g1) {
    geo.set(p, g0);
    qpix = geo.pix(p, qpr, qpc);
    qimg = geo.img(p);
    prefetch();
  }
  for (int g = g0; g < g1; ++g) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (sval[c]) {
        const int id = tid + c * blockDim.x, sr = id >> 2, sc = id & 3;
        *reinterpret_cast<bf16x8*>(Ks + sr * RS + sc * 8) = kr[c];
        *reinterpret_cast<bf16x8*>(Vs + sr * RS + sc * 8) = vr[c];
      }
    }
    const bool border = geo.border;
    float* kregf = reinterpret_cast<float*>(S.kreg);     // shift-region labels as floats (padding rows stay 0)
    if (border) for (int t = tid; t < p.N; t += blockDim.x) { const int pr = t / p.ws; kregf[t] = (float)geo.reg(p, pr, t - pr * p.ws); }
    const float qregf = border ? (float)geo.reg(p, qpr, qpc) : 0.f;
    const bf16x8 qf = qn, dof = don;
    // Accumulator seeds: q.k + bias/scale and dP - delta come straight out of the MFMAs, log2 p = that * scale*log2e - lse*log2e
    // is one fma; the lanes of padded queries get lse = +inf so that their p (and with it ds, dbias) is exactly 0.
    // delta[query] = sum_d dO*O (the softmax-backward row term) is computed here from the lane's 8-channel pieces of dO and O
    // (4 v_dot2_f32_bf16 + a 4-lane sum) and written out for the dK/dV pass: no separate pass over O and dO.
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    float dpart = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
      dpart = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, bf16x2{don[e], don[e + 1]}), __builtin_bit_cast(bf16x2_t, bf16x2{on[e], on[e + 1]}), dpart, false);
    const float dlt = g4sum(dpart);
    const unsigned opix = qpix;
    const size_t oimg = qimg;
    if (gq == 0 && qval) *at(p.delta + oimg * p.heads + (size_t)h * p.Hres * p.Wres, opix) = dlt;
    const float nlse = qval ? lsen * -1.4426950408889634f : -INFINITY, dc = -dlt;
    __syncthreads();
    if (g + 1 < g1) {
      geo.next(p);
      qpix = geo.pix(p, qpr, qpc);
      qimg = geo.img(p);
      prefetch();
    }
    f32x4 dqacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // Two copies of the body (with / without the shift-mask compares) behind one wave-uniform branch; the LDS operands of a
    // key-tile pair are requested together (padded tiles read zero rows and carry bias = -inf).
    auto body = [&](auto border_t) {
      constexpr bool BORDER = decltype(border_t)::value;
      constexpr bool PIPE = NTC != 0;                      // explicit load placement only where the register budget allows it
#pragma unroll
      for (int t2 = 0; t2 < MTP / 2; ++t2) {
        if (t2 * 2 < ntile) {
          bf16x8 kf[2], vf[2], kt_[2];
          f32x4 sa[2], sdp[2], kg[2], ds[2];
          auto loads = [&](int u) {
            const int kt = 2 * t2 + u;
            kf[u] = *reinterpret_cast<const bf16x8*>(Ks + (kt * 16 + lq) * RS + gq * 8);
            vf[u] = *reinterpret_cast<const bf16x8*>(Vs + (kt * 16 + lq) * RS + gq * 8);
            sa[u] = kt < MT ? bias_tile(kt) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BORDER) kg[u] = *reinterpret_cast<const f32x4*>(kregf + kt * 16 + gq * 4);
          };
          auto mfmas = [&](int u) {
            sa[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], qf, sa[u], 0, 0, 0);
            sdp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u], dof, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          };
          auto valu = [&](int u) {
            const int kt = 2 * t2 + u;
            ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < ntile) {
              f32x4 sv = fma4(sa[u], scale * 1.4426950408889634f, nlse);   // log2 p; -inf on padded keys
              if constexpr (BORDER) sv = region_mask(sv, kg[u], qregf, -144.26950408889634f);
              const f32x4 d = exp2x4(sv) * (sdp[u] + dc);   // (-delta as a seed of the dP MFMA would pin four more registers)
              ds[u] = d;
              if (kt < NLD) dbl[kt * NTH] += d; else dbacc[kt < NLD ? 0 : kt - NLD] += d;
            }
          };
          if constexpr (PIPE) {
            loads(0); loads(1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0); mfmas(1);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) kt_[dt] = trr_frag(Ks, dt * 16, 2 * t2, gq, lq);   // lands during the VALU pass
            __builtin_amdgcn_sched_barrier(0);
            valu(0); valu(1);
          } else {
            loads(0); mfmas(0); valu(0);
            loads(1); mfmas(1); valu(1);
          }
          const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            if constexpr (!PIPE) kt_[dt] = trr_frag(Ks, dt * 16, 2 * t2, gq, lq);
            dqacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt], dsf, dqacc[dt], 0, 0, 0);
          }
        }
      }
    };
    if (border) body(std::true_type{}); else body(std::false_type{});
    if (qval) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(dqacc[dt][r] * scale);
        *reinterpret_cast<bf16x4*>(at(p.dqkv + oimg * ld, opix * ld + qo + dt * 16 + gq * 4)) = o;
      }
    }
    if (p.colsum_part) {                                 // (padded query lanes hold zeros)
      f32x4* mine = cs_all + threadIdx.x;
      mine[0] += dqacc[0] * scale; mine[NTH] += dqacc[1] * scale;
    }
  }
  if (qval) {                                          // every (z, h, i, j<N) entry is written, zeros included
    float* dst = p.dbias_part + (((size_t)blockIdx.x * p.heads + h) * p.N + i) * p.N;
#pragma unroll
    for (int kt = 0; kt < MT; ++kt)
      if (kt < ntile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt * 16 + gq * 4 + r;
          if (j < p.N) dst[j] = kt < NLD ? dbl[kt * NTH][r] : dbacc[kt < NLD ? 0 : kt - NLD][r];
        }
  }
  if (p.colsum_part) {
    const int chan0[1] = {qo};
    colsum_flush<2, NTH>(cs_all, p.colsum_part + (size_t)(blockIdx.x * gridDim.z + blockIdx.z) * 3 * C, chan0);
  }
}

// ================================================================ backward pass B: dK, dV ======================
// MAXC: 16-byte staging chunks per thread; NTC: key/query tiles if known at compile time (9 for 12x12 windows);
// MTT: tile capacity (10 / 21); BREG: window-invariant bias slice in registers (else LDS look-ups per score)
template <int MAXC, int NTC = 0, int MTT = 10, bool BREG = true>
__global__ __launch_bounds__(MAXC == 1 ? 640 : 448) void win_bwd_dkv_kernel(WinP p) {
  WIN_DIMS(MTT);
  constexpr int NTH = MAXC == 1 ? 640 : 448;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = (2 * p.ws - 1) * (2 * p.ws - 1);
  Smem S = carve<MTT>(smem, nb, 2);
  bf16* Qs = S.a0; bf16* dOs = S.a1;
  f32x4* cs_all = reinterpret_cast<f32x4*>(S.a1 + MAXN * RS);   // column-sum slots of dK / dV (see colsum_flush)
  if (p.colsum_part) colsum_zero<4, NTH>(cs_all);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gq = lane >> 4, lq = lane & 15;
  const int h = blockIdx.y, C = p.C, ld = 3 * C;
  const int qo = chan_q(p, h), ko = chan_k(p, h), vo = chan_v(p, h);
  // compile-time for the 12x12 window (guards fold, the 10th tile's code disappears) and for the 21-tile variant (18x18
  // windows fill it; padded tiles of smaller windows carry bias = -inf and zero rows, so running them is only wasted work)
  const int ntile = NTC ? NTC : MT > 10 ? MT : (p.N + 15) >> 4;
  setup<MTT>(p, S, h, nb, 2, 1.4426950408889634f, -INFINITY);
  int spr[MAXC], spc[MAXC];
  bool sval[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int id = tid + c * blockDim.x;
    sval[c] = (id >> 2) < p.N;
    const int r = sval[c] ? (id >> 2) : 0;
    spr[c] = r / p.ws; spc[c] = r - spr[c] * p.ws;
  }
  const int j = (blockIdx.z * (blockDim.x >> 6) + wave) * 16 + lq;   // this lane's key
  const bool kval = j < p.N;
  const int jc = kval ? j : p.N - 1;
  const int kpr = jc / p.ws, kpc = jc - kpr * p.ws;
  // bias index for (query i, key j) = koff[i] + cst - koff[j]
  const int kconst = (p.ws - 1) * (2 * p.ws - 1) + p.ws - 1 - (kpr * (2 * p.ws - 1) + kpc);
  const float scale = 0.17677669529663687f;

  const int g0 = blockIdx.x * p.gpb, g1 = min(p.G, g0 + p.gpb);
  if (g0 >= g1) return;
  __syncthreads();
  f32x4 breg[BREG ? MT : 1];                            // window-invariant bias slice of this key strip
  auto bias_tile = [&](int qt) -> f32x4 {                 // log2 domain, 0 on padded queries (their lse is +inf)
    if constexpr (BREG) {
      return breg[qt];
    } else {
      // branch-free: padded queries look up entry koff = 0 (any finite value: their accumulator seed is -inf)
      const int4 ko = *reinterpret_cast<const int4*>(S.koff + qt * 16 + gq * 4);
      return f32x4{S.btab[ko.x + kconst], S.btab[ko.y + kconst], S.btab[ko.z + kconst], S.btab[ko.w + kconst]};
    }
  };
  if constexpr (BREG) {
#pragma unroll
    for (int qt = 0; qt < MT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = qt * 16 + gq * 4 + r;
        // log2 domain; -inf on the lanes of padded keys (they contribute nothing), 0 on padded queries (their seed is -inf)
        breg[qt][r] = !kval ? -INFINITY : (qt < ntile && ii < p.N) ? S.btab[S.koff[ii] + kconst] : 0.f;
      }
  }
  Geo geo;
  geo.set(p, g0);
  bf16x8 qr[MAXC], dr[MAXC], kn, vn;
  float lser[MAXC], dltr[MAXC];
  unsigned kpix = geo.pix(p, kpr, kpc);
  size_t kimg = geo.img(p);
  auto prefetch = [&]() {
    const bf16* base = p.qkv + kimg * ld;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (sval[c]) {
        const unsigned sc = (tid + c * blockDim.x) & 3;
        const unsigned st = geo.pix(p, spr[c], spc[c]);
        qr[c] = *reinterpret_cast<const bf16x8*>(at(base, st * ld + qo + sc * 8));
        dr[c] = *reinterpret_cast<const bf16x8*>(at(p.dout + kimg * C, st * C + h * 32 + sc * 8));
        if (sc == 0) { lser[c] = *at(p.lse + kimg * p.heads + (size_t)h * p.Hres * p.Wres, st); dltr[c] = *at(p.delta + kimg * p.heads + (size_t)h * p.Hres * p.Wres, st); }
      }
    }
    kn = *reinterpret_cast<const bf16x8*>(at(base, kpix * ld + ko + gq * 8));
    vn = *reinterpret_cast<const bf16x8*>(at(base, kpix * ld + vo + gq * 8));
  };
  prefetch();
  for (int g = g0; g < g1; ++g) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (sval[c]) {
        const int id = tid + c * blockDim.x, sr = id >> 2, sc = id & 3;
        *reinterpret_cast<bf16x8*>(Qs + sr * RS + sc * 8) = qr[c];
        *reinterpret_cast<bf16x8*>(dOs + sr * RS + sc * 8) = dr[c];
        // seeds of the S and dP accumulators: (q.k - lse/scale) * scale*log2e + bias = log2 p, and dP - delta come out of the MFMAs
        if (sc == 0) { S.lse[sr] = lser[c] * -5.656854249492381f; S.dlt[sr] = -dltr[c]; }
      }
    }
    const bool border = geo.border;
    float* kregf = reinterpret_cast<float*>(S.kreg);     // shift-region labels as floats (padding rows stay 0)
    if (border) for (int t = tid; t < p.N; t += blockDim.x) { const int pr = t / p.ws; kregf[t] = (float)geo.reg(p, pr, t - pr * p.ws); }
    const float kregf_own = border ? (float)geo.reg(p, kpr, kpc) : 0.f;
    const bf16x8 kf = kn, vf = vn;
    const unsigned opix = kpix;
    const size_t oimg = kimg;
    __syncthreads();
    if (g + 1 < g1) {
      geo.next(p);
      kpix = geo.pix(p, kpr, kpc);
      kimg = geo.img(p);
      prefetch();
    }
    f32x4 dkacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dvacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const float gate = kval ? 1.f : 0.f;                 // lanes of padded keys contribute nothing (folded into breg when BREG)
    // The per-window body exists twice (with / without the shift-mask compares, swin_transformer.py:327-350) behind ONE
    // wave-uniform branch, so each copy is a single basic block; every LDS operand of a query-tile pair is requested at the
    // top of the pair (one exposed LDS latency per pair instead of one per MFMA).  Padded tiles read zero rows.
    auto body = [&](auto border_t) {
      constexpr bool BORDER = decltype(border_t)::value;
#pragma unroll
      for (int t2 = 0; t2 < MTP / 2; ++t2) {
        if (t2 * 2 < ntile) {
          bf16x8 qf[2], df[2], qt[2], dt_[2];
          f32x4 l4[2], d4[2], qg[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int qi = 2 * t2 + u;
            qf[u] = *reinterpret_cast<const bf16x8*>(Qs + (qi * 16 + lq) * RS + gq * 8);
            df[u] = *reinterpret_cast<const bf16x8*>(dOs + (qi * 16 + lq) * RS + gq * 8);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int qi = 2 * t2 + u;
            l4[u] = *reinterpret_cast<const f32x4*>(S.lse + qi * 16 + gq * 4);
            d4[u] = *reinterpret_cast<const f32x4*>(S.dlt + qi * 16 + gq * 4);
            if constexpr (BORDER) qg[u] = *reinterpret_cast<const f32x4*>(kregf + qi * 16 + gq * 4);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 sa[2], sdp[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            sa[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[u], kf, l4[u], 0, 0, 0);     // S[query][key] - lse/scale
            sdp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[u], vf, d4[u], 0, 0, 0);    // dP[query][key] - delta
          }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {                   // transposed operands of the dK / dV MFMAs land during the VALU pass
            qt[dt] = trr_frag(Qs, dt * 16, 2 * t2, gq, lq);
            dt_[dt] = trr_frag(dOs, dt * 16, 2 * t2, gq, lq);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 ds[2], pd[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int qi = 2 * t2 + u;
            ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            pd[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qi < ntile) {
              f32x4 sv = fma4(sa[u], scale * 1.4426950408889634f, bias_tile(qi));
              if constexpr (BORDER) sv = region_mask(sv, qg[u], kregf_own, -144.26950408889634f);
              f32x4 pr = exp2x4(sv);
              if constexpr (!BREG) pr *= gate;
              pd[u] = pr;
              ds[u] = pr * sdp[u];
            }
          }
          const bf16x8 dsf = pack8(ds[0], ds[1]), pf = pack8(pd[0], pd[1]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[dt], dsf, dkacc[dt], 0, 0, 0);
            dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt_[dt], pf, dvacc[dt], 0, 0, 0);
          }
        }
      }
    };
    if (border) body(std::true_type{}); else body(std::false_type{});
    if (kval) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x4 ok, ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok[r] = f2bf(dkacc[dt][r] * scale); ov[r] = f2bf(dvacc[dt][r]); }
        *reinterpret_cast<bf16x4*>(at(p.dqkv + oimg * ld, opix * ld + ko + dt * 16 + gq * 4)) = ok;
        *reinterpret_cast<bf16x4*>(at(p.dqkv + oimg * ld, opix * ld + vo + dt * 16 + gq * 4)) = ov;
      }
    }
    if (p.colsum_part) {                                 // (padded key lanes hold zeros)
      f32x4* mine = cs_all + threadIdx.x;
      mine[0] += dkacc[0] * scale; mine[NTH] += dkacc[1] * scale; mine[2 * NTH] += dvacc[0]; mine[3 * NTH] += dvacc[1];
    }
  }
  if (p.colsum_part) {
    const int chan0[2] = {ko, vo};
    colsum_flush<4, NTH>(cs_all, p.colsum_part + (size_t)(blockIdx.x * gridDim.z + blockIdx.z) * 3 * C, chan0);
  }
}

// ================================================================ backward, ONE pass (12x12 windows) ============
// The two-pass backward evaluates the softmax backward twice (S, P, dP, dS once per pass: the kernels are VALU-issue bound, so
// that is the cost) and reads q, k, v, dO twice.  Here each (window, head) is visited once:
//   phase 1 (wave = key strip, the dK/dV pass's formulation: S[query][key] with the query on registers, the key on lanes):
//           dK, dV of the strip complete in registers; dS is ALSO what dbias accumulates (registers, over the windows of the run)
//           and is written -- transposed, bf16 -- into an LDS image DSt[key][query] (one ds_write_b64 per query tile);
//   phase 2 (wave = query strip, after one barrier): dQ^T = K^T . dS^T, ten MFMAs per window whose B operand is read back from
//           DSt with ds_read_b64_tr_b16 (the same fragment trr_frag() builds for K^T: both operands see the same key order).
// delta = rowsum(dO . O) is formed while staging (the four lanes that stage a row hold its four chunks of dO and O).
// LDS: tables 4.7 KB + Q, dO, K images (160 rows: tile 9 stays zero) 46 KB + DSt 144 x 352 B = 50.7 KB + column-sum slots 55 KB.
// Row stride of DSt: 88 dwords = 24 mod 64, the stride class the transposed read is conflict-free for (see RS above).
constexpr int DSS = 176;
__device__ __forceinline__ bf16x8 trr_frag_lohi(const bf16* lo_img, int lo_stride, const bf16* hi_img, int hi_stride, int g, int l) {
  const bf16* a = lo_img + (g * 4 + (l >> 2)) * lo_stride + (l & 3) * 4;
  const bf16* b = hi_img + (g * 4 + (l >> 2)) * hi_stride + (l & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
  return __builtin_bit_cast(bf16x8, o);
}

// SHIFT: compile-time copy for shifted blocks -- every window then runs the masked body (interior windows carry one label: the
// mask term is 0), so the kernel has ONE window body (two copies behind a branch cost 4-6 registers more than the cap allows).
// NWV = 11 (round 6): nine waves on four SIMDs are 3 / 2 / 2 / 2, both phases are issue-bound, so SIMD 0's third wave (key strip 8) was
// the critical path of every window (tools/win_trace.py: phase 1 ends at 4.8 k ticks in waves 0-3, 6.3 k in waves 4-7, 8.1 k in wave 8).
// Key strip 8 is cut by QUERY tiles into three helper waves 8 / 9 / 10 (SIMDs 0 / 1 / 2: 21 / 21 / 21 / 18 tile steps per SIMD instead of
// 27 / 18 / 18 / 18).  Waves 8 and 9 hand their partial dK / dV of the strip to wave 10 through two LDS slots; wave 10, idle in phase 2,
// adds them in a fixed order and stores the strip.  Waves 9 and 10 have no query strip (phase 2) and stage nothing.
template <bool SHIFT, int NWV = 9>
__global__ __launch_bounds__(NWV * 64) void win_bwd_fused_kernel(WinP p) {
  constexpr int MT = 9, MAXN = 160, NTH = NWV * 64, NST = 576;   // 12x12 windows: 9 strips of 16, images padded to 10 tiles; NST staging threads
  constexpr bool HELP = NWV == 11;
  static_assert(NWV == 9 || NWV == 11, "9 waves, or 8 + 3 helpers of key strip 8");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = (2 * p.ws - 1) * (2 * p.ws - 1);
  Smem S = carve<10>(smem, nb, 0);
  bf16* Qs = S.a0; bf16* dOs = Qs + MAXN * RS; bf16* Ks = dOs + MAXN * RS; bf16* DSt = Ks + MAXN * RS;
  // dbias accumulators of query tiles 0 .. NL-1: private LDS slots [tile][thread] (read + add + write per window, conflict-free);
  // tiles NL .. 8 stay in registers.  All nine in registers put the kernel 17-36 registers over its 168-register cap (scratch
  // traffic: shifted stage-2 backward 1820 us against 1330 without spills).
  constexpr int NL = 5;
  f32x4* db_lds = reinterpret_cast<f32x4*>(DSt + 144 * DSS);
  // column sums of dQ | dK | dV (the qkv bias gradient): reduced over the 16 key / query lanes in registers (the kernel is not
  // VALU-bound), then one f32x4 per (wave, lane group, piece) in LDS: [wave][gq][6]
  f32x4* cs_small = db_lds + NL * NST;
  f32x4* pslot = cs_small + NWV * 4 * 6;                 // HELP: partial dK | dV of key strip 8, [helper wave 8 | 9][4 accumulators][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool helper = HELP && wave >= 8;                 // (wave-uniform)
  // A helper runs the code of tiles 6, 7, 8 (first of a pair, second of a pair, alone; register accumulators 1 .. 3) on its query tiles
  // hq0 .. hq0 + 2: every per-tile address below is taken relative to a base moved by trb tiles (0 in the waves that own a whole strip).
  const int hq0 = helper ? (wave - 8) * 3 : 0;
  const int trb = helper ? hq0 - 6 : 0;
  const int gq = lane >> 4, lq = lane & 15;
  const int h = blockIdx.y, C = p.C;
  // layout 3 = planar [3 * heads planes][token][32] for qkv / dqkv and [heads planes][token][32] for o / dout (see win_fwd_kernel)
  // layout 4: qkv planar; o, dout and dqkv in token rows (dqkv in the reference channel order)
  const bool planar = p.hmajor == 3 || p.hmajor == 4, planar_o = p.hmajor == 3;
  const size_t pl = (size_t)p.B * p.Hres * p.Wres * 32;
  const int ld = planar ? 32 : 3 * C, ldo = planar_o ? 32 : C, ldd = p.hmajor == 4 ? 3 * C : ld;
  const size_t qo = planar ? h * pl : (size_t)chan_q(p, h), ko = planar ? (p.heads + h) * pl : (size_t)chan_k(p, h),
               vo = planar ? (2 * p.heads + h) * pl : (size_t)chan_v(p, h), oo = planar_o ? h * pl : (size_t)h * 32;
  const size_t dqo = p.hmajor == 4 ? (size_t)h * 32 : qo, dko = p.hmajor == 4 ? (size_t)C + h * 32 : ko, dvo = p.hmajor == 4 ? (size_t)2 * C + h * 32 : vo;
  setup<10>(p, S, h, nb, 0, 1.4426950408889634f, -INFINITY);
  {                                                      // zero the three images once: rows 144..159 (tile 9) are never staged
    uint32_t* z = reinterpret_cast<uint32_t*>(Qs);
    for (int t = tid; t < 3 * MAXN * RS / 2; t += NTH) z[t] = 0u;
  }
  for (int t = tid; t < NL * NST + NWV * 4 * 6; t += NTH) db_lds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging chunk of this thread: row sr of the window, 16-byte piece sc (576 threads = 144 rows x 4 pieces exactly)
  const int sr = tid >> 2, sc = tid & 3;
  const int spr = sr / p.ws, spc = sr - spr * p.ws;
  const int j = (helper ? 8 : wave) * 16 + lq;           // this lane's key (phase 1) = this lane's query (phase 2)
  const bool stager = !HELP || wave < 9;                 // waves 0-8 = the 576 staging threads
  const int kpr = j / p.ws, kpc = j - kpr * p.ws;
  const int kconst = (p.ws - 1) * (2 * p.ws - 1) + p.ws - 1 - (kpr * (2 * p.ws - 1) + kpc);
  const float scale = 0.17677669529663687f;
  const int g0 = blockIdx.x * p.gpb, g1 = min(p.G, g0 + p.gpb);
  if (g0 >= g1) return;
  __syncthreads();
  // Bias of (query tile qt, rows gq*4 .. +3) for this lane's key: the four queries lie in one window row (12 = 3 x 4), so their
  // table entries are CONSECUTIVE: btab[koff[qt*16 + gq*4] + kconst + 0..3] -- five ds_read_b32 per tile instead of
  // 36 registers held for the whole run (this kernel also carries the dbias accumulators; the register cap is 168 at 9 waves).
  f32x4 dbacc[MT - NL];
#pragma unroll
  for (int qt = 0; qt < MT - NL; ++qt) dbacc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4* db_mine = db_lds + tid;
  // sum over the 16 lanes of a lane group (a DPP row) as four v_add_f32 with DPP operands: quad butterflies, then the mirrored
  // half row and the mirrored row (ds_bpermute shuffles here cost the kernel 15 %: 96 LDS-crossbar operations per window and wave)
  auto lane16_sum = [&](f32x4 v) -> f32x4 {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = v[e];
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));   // row_half_mirror
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));   // row_mirror
      v[e] = x;
    }
    return v;
  };
  f32x4* cs_mine = cs_small + (wave * 4 + gq) * 6;
  const int* koff_b = S.koff + trb * 16 + gq * 4;
  auto bias_tile = [&](int qt) -> f32x4 {                 // (the tile's base offset is re-read as well: nine more registers do not exist)
    const float* b = S.btab + koff_b[qt * 16] + kconst;
    return f32x4{b[0], b[1], b[2], b[3]};
  };
  Geo geo;
  geo.set(p, g0);
  bf16x8 qr, dr, kr, orr, vn;
  float lser = 0.f;
  unsigned kpix = geo.pix(p, kpr, kpc);
  size_t kimg = geo.img(p);
  auto prefetch = [&]() {
    const bf16* base = p.qkv + kimg * ld;
    if (stager) {
      const unsigned st = geo.pix(p, spr, spc);
      qr = *reinterpret_cast<const bf16x8*>(at(base + qo, st * ld + sc * 8));
      kr = *reinterpret_cast<const bf16x8*>(at(base + ko, st * ld + sc * 8));
      dr = *reinterpret_cast<const bf16x8*>(at(p.dout + oo + kimg * ldo, st * ldo + sc * 8));
      orr = *reinterpret_cast<const bf16x8*>(at(static_cast<const bf16*>(p.o) + oo + kimg * ldo, st * ldo + sc * 8));
      if (sc == 0) lser = *at(p.lse + kimg * p.heads + (size_t)h * p.Hres * p.Wres, st);
    }
    vn = *reinterpret_cast<const bf16x8*>(at(base + vo, kpix * ld + gq * 8));
  };
  prefetch();
  // TRACE build (-DFIBER_WIN_TRACE, tools/win_trace.py): s_memtime ticks per segment and wave, summed over the windows of the run, written
  // into the delta workspace (unused by this kernel)
#ifdef FIBER_WIN_TRACE
  unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define WIN_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tr[i] += t_ - t0; t0 = t_; } while (0)
#else
#define WIN_MARK(i) do { } while (0)
#endif
  for (int g = g0; g < g1; ++g) {
    __syncthreads();                                     // phase 2 of the previous window is done with the images and DSt
    WIN_MARK(0);
#ifdef FIBER_WIN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    WIN_MARK(1);
    if (stager) {
    *reinterpret_cast<bf16x8*>(Qs + sr * RS + sc * 8) = qr;
    *reinterpret_cast<bf16x8*>(dOs + sr * RS + sc * 8) = dr;
    *reinterpret_cast<bf16x8*>(Ks + sr * RS + sc * 8) = kr;
    {
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      float dpart = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        dpart = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, bf16x2{dr[e], dr[e + 1]}), __builtin_bit_cast(bf16x2_t, bf16x2{orr[e], orr[e + 1]}), dpart, false);
      dpart += __shfl_xor(dpart, 1);
      dpart += __shfl_xor(dpart, 2);
      // seeds of the S and dP accumulators: (q.k - lse/scale) * scale*log2e + bias = log2 p, and dP - delta come out of the MFMAs
      if (sc == 0) { S.lse[sr] = lser * -5.656854249492381f; S.dlt[sr] = -dpart; }
    }
    }
    float* kregf = reinterpret_cast<float*>(S.kreg);     // shift-region labels as floats
    float kregf_own = 0.f;
    if constexpr (SHIFT) {
      if (tid < p.N) kregf[tid] = (float)geo.reg(p, tid / p.ws, tid % p.ws);
      kregf_own = (float)geo.reg(p, kpr, kpc);
    }
    const bf16x8 vf = vn;
    const unsigned opix = kpix;
    const size_t oimg = kimg;
    WIN_MARK(2);
    __syncthreads();
    WIN_MARK(3);
    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + j * RS + gq * 8);   // this lane's key row out of the staged image
    // The next window's loads are requested from INSIDE phase 1, a third of the waves at tile 0, 3 and 6: all nine waves requesting
    // right behind the barrier stalled each other for 0.5-2 k ticks on the address path (tools/win_trace.py) with nobody computing;
    // the data is needed a whole phase later either way (it used to land ~9 k ticks early).
    const int pf_tile = helper ? 6 : (wave % 3) * 3;
    const bool pf_more = g + 1 < g1;
    auto prefetch_next = [&]() {
      geo.next(p);
      kpix = geo.pix(p, kpr, kpc);
      kimg = geo.img(p);
      prefetch();
    };
    WIN_MARK(7);
    // ---- phase 1: this wave's key strip against every query tile
    f32x4 dkacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dvacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    bf16* dsrow = DSt + j * DSS + trb * 16 + gq * 4;     // DSt[key j][query tile * 16 + gq * 4 .. + 3]
    const bf16* Qb = Qs + trb * 16 * RS; const bf16* dOb = dOs + trb * 16 * RS;
    const float* lse_b = S.lse + trb * 16 + gq * 4; const float* dlt_b = S.dlt + trb * 16 + gq * 4;
    // One query tile per step; the dK / dV MFMAs are the K = 16 form (the operand is the tile's own 16 queries), so a step holds one
    // tile's temporaries instead of a pair's -- this kernel also carries 36 dbias accumulators under a 168-register cap.
    {
      constexpr bool BORDER = SHIFT;
      bf16x4 ds_prev, p_prev;
      // ROLE: 0 = first tile of a pair, 1 = second tile of a pair, 2 = alone.  DB: the tile's dbias accumulator, -1 = the LDS slot of tile QI,
      // else register set DB.  (In a helper wave "tile QI" is query tile QI + trb.)
      // The step of a tile is cut in two: front = operand reads + the S and dP MFMAs (results in sa / sdp), back = everything behind them.
      // The front of tile i + 1 is issued BEFORE the back of tile i (two result sets, A for even tiles, B for odd ones): a SIMD holds two or three
      // waves here and each tile is one dependent chain LDS -> MFMA -> exp -> convert -> MFMA, so the matrix pipe idled under the chain's VALU part
      // (vector + matrix time per SIMD added up to the length of phase 1: tools/win_trace.py).
      auto front = [&](auto qi_t, f32x4& sa, f32x4& sdp) {
        constexpr int qi = decltype(qi_t)::value;
        if constexpr (qi % 3 == 0) { if (pf_more && pf_tile == qi) prefetch_next(); }
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qb + (qi * 16 + lq) * RS + gq * 8);
        const bf16x8 df = *reinterpret_cast<const bf16x8*>(dOb + (qi * 16 + lq) * RS + gq * 8);
        sa = *reinterpret_cast<const f32x4*>(lse_b + qi * 16);
        sdp = *reinterpret_cast<const f32x4*>(dlt_b + qi * 16);
        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, sa, 0, 0, 0);        // S[query][key] - lse/scale
        sdp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf, sdp, 0, 0, 0);      // dP[query][key] - delta
        __builtin_amdgcn_sched_barrier(0);
      };
      auto back = [&](auto qi_t, auto role_t, auto db_t, const f32x4& sa, const f32x4& sdp) {
        constexpr int QI = decltype(qi_t)::value, ROLE = decltype(role_t)::value, DB = decltype(db_t)::value;
        constexpr int qi = QI;
        f32x4 qg;
        if constexpr (BORDER) qg = *reinterpret_cast<const f32x4*>(kregf + trb * 16 + qi * 16 + gq * 4);
        const f32x4 bq = bias_tile(qi);
        // dK / dV: query tiles are consumed in PAIRS by K = 32 MFMAs (the K = 16 form has half the rate per pass: 1.7 k of SIMD 0's
        // 10 k cycles per window); the odd tile of a pair brings both tiles' transposed operands, the last tile (8) runs alone
        constexpr bool PAIR_END = ROLE == 1, ALONE = ROLE == 2;
        bf16x8 qt2[2], dt2[2];
        if constexpr (PAIR_END) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) { qt2[dt] = trr_frag(Qb, dt * 16, qi - 1, gq, lq); dt2[dt] = trr_frag(dOb, dt * 16, qi - 1, gq, lq); }
        } else if constexpr (ALONE) {
          // Q^T / dO^T of this tile and of the image rows behind it (tile 8: the all-zero, never staged rows 144..159; a helper's lone tile:
          // the next tile's rows, finite, against the zero half of dS / P): the lone tile runs as a K = 32 MFMA whose upper half is zero.  NOT the K = 16 opcode (round 5): a 16x16x16 MFMA that takes the D of a 16x16x32 MFMA as
          // its C is padded by hipcc like an accumulate chain of ONE opcode (one wait state) and can read the accumulator before the
          // K = 32 instruction has written it -- seen in a forward kernel as row sums of "last tile + stale registers".
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) { qt2[dt] = trr_frag(Qb, dt * 16, qi, gq, lq); dt2[dt] = trr_frag(dOb, dt * 16, qi, gq, lq); }
        }
        f32x4 sv = fma4(sa, scale * 1.4426950408889634f, bq);
        if constexpr (BORDER) sv = region_mask(sv, qg, kregf_own, -144.26950408889634f);
        const f32x4 pr = exp2x4(sv);
        const f32x4 ds = pr * sdp;
        if constexpr (DB < 0) db_mine[QI * NST] += ds; else dbacc[DB] += ds;
        bf16x4 dsb, pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) { dsb[r] = f2bf(ds[r]); pb[r] = f2bf(pr[r]); }
        *reinterpret_cast<bf16x4*>(dsrow + qi * 16) = dsb;
        if constexpr (PAIR_END) {
          bf16x8 dsf, pf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { dsf[r] = ds_prev[r]; dsf[4 + r] = dsb[r]; pf[r] = p_prev[r]; pf[4 + r] = pb[r]; }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt2[dt], dsf, dkacc[dt], 0, 0, 0);
            dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt2[dt], pf, dvacc[dt], 0, 0, 0);
          }
        } else if constexpr (ALONE) {
          bf16x8 dsf, pf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { dsf[r] = dsb[r]; dsf[4 + r] = f2bf(0.f); pf[r] = pb[r]; pf[4 + r] = f2bf(0.f); }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt2[dt], dsf, dkacc[dt], 0, 0, 0);
            dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt2[dt], pf, dvacc[dt], 0, 0, 0);
          }
        } else {
          ds_prev = dsb; p_prev = pb;
        }
        __builtin_amdgcn_sched_barrier(0);                 // no loads of later tiles hoisted over this one (register cap)
      };
      using std::integral_constant;
      static_assert(NL == 5 && MT == 9, "the call list below spells the accumulator of every tile out");
      f32x4 saA, sdA, saB, sdB;
#define WIN_F(QI, X) front(integral_constant<int, QI>{}, sa##X, sd##X)
#define WIN_B(QI, ROLE, DB, X) back(integral_constant<int, QI>{}, integral_constant<int, ROLE>{}, integral_constant<int, DB>{}, sa##X, sd##X)
      if (!helper) {
        WIN_F(0, A);
        WIN_F(1, B); WIN_B(0, 0, -1, A);
        WIN_F(2, A); WIN_B(1, 1, -1, B);
        WIN_F(3, B); WIN_B(2, 0, -1, A);
        WIN_F(4, A); WIN_B(3, 1, -1, B);
        WIN_F(5, B); WIN_B(4, 0, -1, A);
      }
      WIN_F(6, A);
      if (!helper) WIN_B(5, 1, 0, B);
      WIN_F(7, B); WIN_B(6, 0, 1, A);
      WIN_F(8, A); WIN_B(7, 1, 2, B);
      WIN_B(8, 2, 3, A);
#undef WIN_F
#undef WIN_B
    }
    WIN_MARK(8);
    auto store_dkv = [&]() {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x4 ok, ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok[r] = f2bf(dkacc[dt][r] * scale); ov[r] = f2bf(dvacc[dt][r]); }
        *reinterpret_cast<bf16x4*>(at(p.dqkv + dko + oimg * ldd, opix * ldd + dt * 16 + gq * 4)) = ok;
        *reinterpret_cast<bf16x4*>(at(p.dqkv + dvo + oimg * ldd, opix * ldd + dt * 16 + gq * 4)) = ov;
      }
      if (p.colsum_part) {
        const f32x4 s0 = lane16_sum(dkacc[0] * scale), s1 = lane16_sum(dkacc[1] * scale), s2 = lane16_sum(dvacc[0]), s3 = lane16_sum(dvacc[1]);
        if (lq == 0) { cs_mine[2] += s0; cs_mine[3] += s1; cs_mine[4] += s2; cs_mine[5] += s3; }
      }
    };
    if (!helper) store_dkv();
    else if (wave < 10) {                                // partial sums of key strip 8 over this helper's query tiles -> wave 10
      f32x4* ps = pslot + (wave - 8) * 256 + lane;
      ps[0] = dkacc[0]; ps[64] = dkacc[1]; ps[128] = dvacc[0]; ps[192] = dvacc[1];
    }
    WIN_MARK(4);
    __syncthreads();                                     // DSt (and the partial slots) complete
    WIN_MARK(5);
    if (HELP && wave == 10) {                            // (w8 + w9) + w10, always in this order
      const f32x4* ps = pslot + lane;
      dkacc[0] = (ps[0] + ps[256]) + dkacc[0]; dkacc[1] = (ps[64] + ps[320]) + dkacc[1];
      dvacc[0] = (ps[128] + ps[384]) + dvacc[0]; dvacc[1] = (ps[192] + ps[448]) + dvacc[1];
      store_dkv();
    }
    if (!HELP || wave < 9) {
    // ---- phase 2: dQ^T[d][query] of this wave's query strip = sum over keys K^T[d][key] dS^T[key][query]
    f32x4 dqacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t2 = 0; t2 < 5; ++t2) {
      // key tiles 2 t2 and 2 t2 + 1; tile 9 does not exist in DSt: its half of the fragment is read from the zero rows of the K image
      const bf16* lo = DSt + (2 * t2 * 16) * DSS + wave * 16;
      const bf16x8 dsb = t2 < 4 ? trr_frag_lohi(lo, DSS, lo + 16 * DSS, DSS, gq, lq)
                                : trr_frag_lohi(lo, DSS, Ks + 144 * RS, RS, gq, lq);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 kt_ = trr_frag(Ks, dt * 16, 2 * t2, gq, lq);
        dqacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, dsb, dqacc[dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = f2bf(dqacc[dt][r] * scale);
      *reinterpret_cast<bf16x4*>(at(p.dqkv + dqo + oimg * ldd, opix * ldd + dt * 16 + gq * 4)) = o;
    }
    if (p.colsum_part) {
      const f32x4 s0 = lane16_sum(dqacc[0] * scale), s1 = lane16_sum(dqacc[1] * scale);
      if (lq == 0) { cs_mine[0] += s0; cs_mine[1] += s1; }
    }
    }
    WIN_MARK(6);
  }
#ifdef FIBER_WIN_TRACE
  if (lane == 0 && blockIdx.x < 4 && h == 0) {
    float* o = p.delta + (blockIdx.x * NWV + wave) * 12;
    for (int i = 0; i < 12; ++i) o[i] = (float)tr[i];
    if (wave == 0) p.delta[1000 + blockIdx.x] = (float)(g1 - g0);
  }
#endif
#undef WIN_MARK
  {                                                      // dbias_part[z, h, i, j]: this lane holds column j, rows qt*16 + gq*4 + r
    float* dst = p.dbias_part + ((size_t)blockIdx.x * p.heads + h) * p.N * p.N + j;
    if (!helper) {
#pragma unroll
      for (int qt = 0; qt < MT; ++qt) {
        const f32x4 v = qt < NL ? db_mine[qt * NST] : dbacc[qt < NL ? 0 : qt - NL];
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)(qt * 16 + gq * 4 + r) * p.N] = v[r];
      }
    } else {
#pragma unroll
      for (int li = 0; li < 3; ++li)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)((hq0 + li) * 16 + gq * 4 + r) * p.N] = dbacc[1 + li][r];
    }
  }
  if (p.colsum_part) {                                   // channel c of group grp (q | k | v): piece grp*2 + (c >> 4), lane group (c >> 2) & 3, element c & 3
    __syncthreads();
    if (tid < 96) {
      const int grp = tid >> 5, c = tid & 31;
      float sum = 0.f;
      for (int w = 0; w < NWV; ++w) sum += cs_small[(w * 4 + ((c >> 2) & 3)) * 6 + grp * 2 + (c >> 4)][c & 3];
      const int chan0 = planar ? grp * C + h * 32 : (int)(grp == 0 ? qo : grp == 1 ? ko : vo);   // planar: sums in the reference channel order
      p.colsum_part[(size_t)blockIdx.x * 3 * C + chan0 + c] = sum;
    }
  }
}

// dtable[rel_index(i, j), h] = sum_z part[z, h, i, j], without atomics (round 4; the one-kernel form with one global atomic per (i, j)
// was bound by them: 51 us for 42 MB at stage 2, 1.2 ms per step -- and summed in a run-dependent order):
//   fold:   part[0, h, i, j] = sum_z part[z, h, i, j]      one thread per float4 of the [H, N, N] slice, coalesced over z slices
//   gather: dtable[e, h] = sum over the (ws - |dr|)(ws - |dc|) pairs (i, j) with relative offset e = (dr, dc), read from slice 0
__global__ __launch_bounds__(256) void win_dbias_fold_kernel(float* __restrict__ part, int nz, long n4) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n4) return;
  float4* p = reinterpret_cast<float4*>(part) + t;
  float4 a = p[0], b = {0.f, 0.f, 0.f, 0.f}, c = b, d = b;
  int z = 1;
  for (; z + 3 < nz; z += 4) {
    const float4 v0 = p[(long)z * n4], v1 = p[(long)(z + 1) * n4], v2 = p[(long)(z + 2) * n4], v3 = p[(long)(z + 3) * n4];
    a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;  b.x += v1.x; b.y += v1.y; b.z += v1.z; b.w += v1.w;
    c.x += v2.x; c.y += v2.y; c.z += v2.z; c.w += v2.w;  d.x += v3.x; d.y += v3.y; d.z += v3.z; d.w += v3.w;
  }
  for (; z < nz; ++z) { const float4 v = p[(long)z * n4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  p[0] = float4{(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
}
__global__ __launch_bounds__(256) void win_dbias_fold1_kernel(float* __restrict__ part, int nz, long n) {   // slices that are not float4-sized (odd windows)
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float a = part[t], b = 0.f;
  int z = 1;
  for (; z + 1 < nz; z += 2) { a += part[(long)z * n + t]; b += part[(long)(z + 1) * n + t]; }
  if (z < nz) a += part[(long)z * n + t];
  part[t] = a + b;
}
// One WAVE per table entry (e, h): its (ws - |dr|)(ws - |dc|) pairs are spread over the lanes and added up by a butterfly (a fixed order).  The
// one-thread-per-entry form walked up to 144 pairs with a stride of N + 1 floats per thread: 26 us for 1.3 MB, 24 times per step at any batch.
__global__ __launch_bounds__(256) void win_dbias_gather_kernel(const float* __restrict__ dense, float* __restrict__ dtable, int H, int ws) {
  const int W2 = 2 * ws - 1, N = ws * ws;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= W2 * W2 * H) return;
  const int h = t % H, e = t / H;
  const int dr = e / W2 - (ws - 1), dc = e % W2 - (ws - 1);               // dr = row(i) - row(j), dc = col(i) - col(j)
  const int r0 = dr > 0 ? dr : 0, r1 = dr < 0 ? ws + dr : ws, c0 = dc > 0 ? dc : 0, c1 = dc < 0 ? ws + dc : ws;
  const int ncol = c1 - c0, cnt = (r1 - r0) * ncol;
  const float* src = dense + (size_t)h * N * N;
  float s = 0.f;
  for (int q = lane; q < cnt; q += 64) {
    const int pri = r0 + q / ncol, pci = c0 + q % ncol;
    s += src[(size_t)(pri * ws + pci) * N + (pri - dr) * ws + pci - dc];   // element (i = pri*ws + pci, j = (pri-dr)*ws + pci - dc)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) dtable[(size_t)e * H + h] = s;
}

bool attrs_set[16] = {};
void ensure_attrs() {
  if (!fiber_first_on_device(attrs_set)) return;
  const int big = 160 * 1024;
  hipFuncSetAttribute((const void*)win_fwd_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_dq_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_dkv_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_fwd_kernel<1, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_fwd_kernel<3, 0, 21, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_dq_kernel<3, 0, 21, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_dkv_kernel<3, 0, 21, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_fused_kernel<false, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)win_bwd_fused_kernel<true, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
}

// waves per workgroup / strip groups of the three passes
void strip_geometry(int N, int& nw, int& sg) {
  // N <= 160: one workgroup per (window run, head) with one wave per 16-query strip: each thread stages exactly one 16-byte
  // chunk (MAXC = 1).  A split into <=4-wave strip groups (MAXC = 3, several workgroups per CU) measured no faster.
  // 160 < N <= 336 (18x18 windows: 21 strips): strip groups of 7 waves, each staging the whole window (MAXC = 3).
  if (N <= 160) { nw = cdiv(N, 16); sg = 1; }
  else { nw = 7; sg = cdiv(cdiv(N, 16), 7); }
}

// which kernels use the compile-time tile count for 12x12 windows (bit 0 forward, 1 dQ pass, 2 dK/dV pass; FIBER_WIN_NTC).
// Forward 1828 -> 1542 us at 512 images, stage 0, when it was introduced.  The dQ pass could not afford it while its bias slice
// lived in registers (168-VGPR cap, spills inside the window loop: backward 7441 -> 8837 us); with the slice in an LDS slab all
// three passes use it.
int ntc_mask() {
  static const int m = getenv("FIBER_WIN_NTC") ? atoi(getenv("FIBER_WIN_NTC")) : 7;
  return m;
}

int blocks_for(int G, int heads) {
  // One 9-wave workgroup fits per CU.  One round of 256 workgroups (256/heads window runs per head) when there are few heads:
  // rocprof ablation showed ~45 % of the kernel was per-workgroup setup (bias column, key offsets, LDS clear, bias-slice
  // gather) + per-window geometry when 768 short-lived workgroups ran in 3 rounds.  From 8 heads on (stages 1-3: fewer windows
  // per run, so the end of the single round is ragged) two rounds of half-length runs measured 3-8 % faster
  // (tools/op_bench.py 512 attn: stage 2 forward 399 -> 368 us, backward 1790 -> 1716 us; stage 0 backward 5880 -> 5966 us).
  static const int forced = getenv("FIBER_WIN_BLOCKS") ? atoi(getenv("FIBER_WIN_BLOCKS")) : 0;
  const int per = forced > 0 ? forced : (heads >= 8 ? 512 : 256);
  int nz = per / heads;
  nz = nz < 1 ? 1 : nz;
  return nz > G ? G : nz;
}

bool big_window(int N) { return N > 160; }
// bias-table gradient from the per-workgroup-run partials [nz, H, N, N] (slice 0 is overwritten with the sum)
int dbias_fold_gather(float* part, float* dtable, int nz, int H, int ws, hipStream_t st) {
  const int N = ws * ws, W2 = 2 * ws - 1;
  const long n = (long)H * N * N, n4 = n / 4;
  if (n % 4 == 0) hipLaunchKernelGGL(win_dbias_fold_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, nz, n4);
  else hipLaunchKernelGGL(win_dbias_fold1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, nz, n);   // (odd windows: slices are not float4-sized)
  FIBER_CHECK_LAUNCH();
  hipLaunchKernelGGL(win_dbias_gather_kernel, dim3((unsigned)((W2 * W2 * H + 3) / 4)), dim3(256), 0, st, part, dtable, H, ws);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

WinP make(const void* qkv, int B, int Hres, int Wres, int C, int heads, int ws, int shift, int hmajor) {
  WinP p{};
  p.qkv = (const bf16*)qkv;
  p.B = B; p.Hres = Hres; p.Wres = Wres; p.C = C; p.heads = heads; p.ws = ws; p.shift = shift;
  p.hmajor = hmajor;
  p.nWw = Wres / ws; p.nWh = Hres / ws; p.nW = p.nWw * p.nWh; p.G = B * p.nW; p.N = ws * ws;
  int nw, sg;
  strip_geometry(p.N, nw, sg);
  const int nz = blocks_for(p.G, heads * sg);            // about one workgroup per CU in total
  p.gpb = cdiv(p.G, nz);
  return p;
}

#ifdef FIBER_WIN_TRACE
}  // namespace
extern "C" int fiber_win_trace_read(float* host256) { return (int)hipMemcpyFromSymbol(host256, HIP_SYMBOL(g_win_trace), 256 * sizeof(float)); }
namespace {
#endif
}  // namespace

// Used by attn.hip's C entry points when the window fits the specialised path (head_dim 32, N <= 160).
int fiber_win_fwd_launch(const void* qkv, const float* bias_table, void* o, float* lse, int B, int Hres, int Wres, int C,
                         int heads, int ws, int shift, int hmajor, hipStream_t st) {
  // channel layouts 0..2 everywhere; the planar probe layouts 3 / 4 exist only in the 12 x 12 forward and fused backward
  if (hmajor < 0 || hmajor > 4 || (hmajor >= 3 && ws * ws != 144)) return FIBER_EINVAL;
  ensure_attrs();
  WinP p = make(qkv, B, Hres, Wres, C, heads, ws, shift, hmajor);
  p.o = (bf16*)o; p.lse = lse; p.bias_table = bias_table;
  const int nb = (2 * ws - 1) * (2 * ws - 1);
  int nw, sg;
  strip_geometry(p.N, nw, sg);
  if (big_window(p.N)) hipLaunchKernelGGL((win_fwd_kernel<3, 0, 21, false>), dim3(cdiv(p.G, p.gpb), heads, sg), dim3(64 * nw), smem_bytes<21>(nb, 2), st, p);
  else if (p.N == 144 && (ntc_mask() & 1)) hipLaunchKernelGGL((win_fwd_kernel<1, 9>), dim3(cdiv(p.G, p.gpb), heads, sg), dim3(64 * nw), smem_bytes<10>(nb, 2), st, p);
  else hipLaunchKernelGGL((win_fwd_kernel<1, 0>), dim3(cdiv(p.G, p.gpb), heads, sg), dim3(64 * nw), smem_bytes<10>(nb, 2), st, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

int fiber_win_bwd_slices(int n_windows, int heads) {
  const int nz = blocks_for(n_windows, heads);
  return cdiv(n_windows, cdiv(n_windows, nz));
}

int fiber_win_colsum_rows(int n_windows, int heads, int N) {
  int nw, sg;
  strip_geometry(N, nw, sg);
  return fiber_win_bwd_slices(n_windows, heads) * sg;
}

extern "C" int fiber_fold_rows_f32(const float* part, float* out, int rows, int N, hipStream_t stream);

int fiber_win_bwd_launch(const void* qkv, const float* bias_table, const void* o, const void* dout, const float* lse,
                         void* dqkv, float* dbias_table, float* delta_ws, float* dbias_ws, float* dqkv_colsum, float* colsum_ws,
                         int B, int Hres, int Wres, int C, int heads, int ws, int shift, int hmajor, hipStream_t st) {
  if (hmajor < 0 || hmajor > 4 || (hmajor >= 3 && ws * ws != 144)) return FIBER_EINVAL;
  ensure_attrs();
  WinP p = make(qkv, B, Hres, Wres, C, heads, ws, shift, hmajor);
  p.o = (bf16*)o; p.lse = (float*)lse; p.bias_table = bias_table; p.dout = (const bf16*)dout; p.dqkv = (bf16*)dqkv;
  p.delta = delta_ws; p.dbias_part = dbias_ws; p.colsum_part = colsum_ws;
  const int nb = (2 * ws - 1) * (2 * ws - 1);
  int nw, sg;
  strip_geometry(p.N, nw, sg);
  // (delta[query, head] = sum_d dO*O is produced by the dQ pass itself and read back by the dK/dV pass)
  const int gz = cdiv(p.G, p.gpb);
  const size_t slab = (size_t)nw * 10 * 64 * sizeof(float) * 4;   // dQ pass: per-lane bias slices [wave][tile][lane] x f32x4
  const size_t nth = big_window(p.N) ? 448 : 640;        // launch bounds = slot stride of the column-sum slots
  const size_t csq = colsum_ws ? nth * 2 * 16 : 0, cskv = colsum_ws ? nth * 4 * 16 : 0;
  if ((colsum_ws == nullptr) != (dqkv_colsum == nullptr)) return FIBER_EINVAL;
  if (p.N == 144 && sg == 1) {                           // 12x12 windows: one pass (delta_ws stays unused).  The two-pass instances for this
    // size (round 3's A/B switch FIBER_WIN_FUSED=0) carried 8 / 24 bytes of scratch and are no longer built.
    static const int nwv = getenv("FIBER_WIN_BWD_WAVES") ? atoi(getenv("FIBER_WIN_BWD_WAVES")) : 9;   // 11: key strip 8 on three helper waves
    const size_t bytes = (size_t)(4 * 160 + ((nb + 3) & ~3)) * 4 + (size_t)3 * 160 * RS * 2 + (size_t)144 * DSS * 2 +
                         (size_t)(5 * 576 + (nwv == 11 ? 11 * 4 * 6 + 2 * 4 * 64 : 9 * 4 * 6)) * 16;
    if (nwv == 11) {
      if (shift > 0) hipLaunchKernelGGL((win_bwd_fused_kernel<true, 11>), dim3(gz, heads, 1), dim3(704), bytes, st, p);
      else hipLaunchKernelGGL((win_bwd_fused_kernel<false, 11>), dim3(gz, heads, 1), dim3(704), bytes, st, p);
    } else if (shift > 0) hipLaunchKernelGGL(win_bwd_fused_kernel<true>, dim3(gz, heads, 1), dim3(576), bytes, st, p);
    else hipLaunchKernelGGL(win_bwd_fused_kernel<false>, dim3(gz, heads, 1), dim3(576), bytes, st, p);
    FIBER_CHECK_LAUNCH();
    if (int rc = dbias_fold_gather(dbias_ws, dbias_table, gz, heads, ws, st)) return rc;
    if (colsum_ws) return fiber_fold_rows_f32(colsum_ws, dqkv_colsum, gz, 3 * C, st);
    return FIBER_OK;
  }
  if (big_window(p.N)) hipLaunchKernelGGL((win_bwd_dq_kernel<3, 0, 21, false>), dim3(gz, heads, sg), dim3(64 * nw), smem_bytes<21>(nb, 2) + csq + 8 * 448 * 16, st, p);   // (+ the 8 dbias tiles kept in LDS: 150 KB in all)
  else if (nw > 8) return FIBER_EINVAL;                  // (N = ws * ws: no square window has 129..160 tokens other than 12 x 12)
  else hipLaunchKernelGGL((win_bwd_dq_kernel<1, 0>), dim3(gz, heads, sg), dim3(64 * nw), smem_bytes<10>(nb, 2) + slab + csq, st, p);
  FIBER_CHECK_LAUNCH();
  if (int rc = dbias_fold_gather(dbias_ws, dbias_table, gz, heads, ws, st)) return rc;
  if (big_window(p.N)) hipLaunchKernelGGL((win_bwd_dkv_kernel<3, 0, 21, false>), dim3(gz, heads, sg), dim3(64 * nw), smem_bytes<21>(nb, 2) + cskv, st, p);
  else hipLaunchKernelGGL((win_bwd_dkv_kernel<1, 0>), dim3(gz, heads, sg), dim3(64 * nw), smem_bytes<10>(nb, 2) + cskv, st, p);
  FIBER_CHECK_LAUNCH();
  if (colsum_ws) return fiber_fold_rows_f32(colsum_ws, dqkv_colsum, gz * sg, 3 * C, st);
  return FIBER_OK;
}
