// Fused multi-head attention for the FIBER fused-backbone path (gfx950 / CDNA4, wave64, MFMA 16x16x32 bf16).
//
// One kernel family serves all four attention sites of the reference hot path:
//   WINDOW mode  Swin (shifted-)window self-attention, swin_transformer.py:195-224 + roll/partition/reverse
//                (:99-126, 367-387): softmax(q.k^T * d^-1/2 + bias_table[rel_index] + shift_mask) . v
//                Windows are never materialised: q/k/v/o live in IMAGE TOKEN ORDER ([B*H*W, 3C] / [B*H*W, C]) and the
//                cyclic shift + window partition are folded into the row addressing; the relative-position bias is
//                gathered from the (2ws-1)^2 x heads table (staged per head in LDS) and the {0,-100} shift mask is
//                recomputed from region labels (swin_transformer.py:327-350) - no N x N tensors touch HBM.
//   PLAIN mode   RoBERTa self-attention (roberta.py:256-326, additive key mask (1-m)*-10000, attention-prob dropout),
//                image->text cross attention (swin_transformer.py:226-256: queries = image tokens in token order, keys =
//                the sample's 40 text tokens, so repeat_interleave over windows disappears) and text->image cross
//                attention (roberta.py:272-276: no mask).
//
// Structure (flash-style, no score tensor in HBM): a wave owns a strip of 16 queries; K and V of up to 160 keys are
// staged in LDS per workgroup as row-major images and shared by its waves (the transposed MFMA operands -- V^T for P.V,
// K^T, Q^T, dO^T in the backward -- are read with ds_read_b64_tr_b16); S^T = K.Q^T is computed with swapped MFMA operands so a lane
// holds 4 consecutive keys of ONE query => row max / row sum are in-lane + two row swaps (v_permlane16/32_swap: VALU, no LDS round trip), and the
// fp32 probabilities convert in-register into the B operand of O^T = V^T.P^T (the key order inside a 32-key MFMA
// k-slot is permuted identically on the V^T side, which is free).  Longer key ranges are walked in chunks with an
// online softmax.  Backward recomputes P from the saved log-sum-exp (one fp32 per query/head) in two passes:
//   pass A (wave = query strip):  dQ, and in WINDOW mode the relative-position-bias gradient accumulated in registers
//                                 over the windows a workgroup visits (no atomics; partials folded by a scatter kernel)
//   pass B (wave = key strip):    dK, dV
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int NKT = 10;          // key (or query) tiles of 16 per LDS chunk  -> 160 rows
constexpr int CH = NKT * 16;     // rows per chunk

struct AttnP {
  const bf16* q; const bf16* k; const bf16* v; bf16* o;       // forward tensors (row-major, head h at column h*D)
  const bf16* dout; bf16* dq; bf16* dk; bf16* dv;              // backward tensors
  float* lse;            // [rows_q, H]  log-sum-exp of each query row (natural log), indexed by q token row
  const float* delta;    // [rows_q, H]  rowsum(dO * O)
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int H, Lq, Lk, G;      // heads, queries / keys per group, number of groups (windows or samples)
  float scale;
  const float* kmask;    // PLAIN: additive key mask [G, Lk] or null
  // WINDOW mode geometry
  int window, Hres, Wres, ws, shift, nWw, nW;
  const float* bias_table;   // [(2ws-1)^2, H] fp32
  float* dbias_part;         // [gridDim.z, H, N, N] fp32 partial bias gradients (pass A, WINDOW)
  int groups_per_block;      // pass A: windows visited by one workgroup
  // LDS chunk geometry of this launch (host-chosen, launch_geometry()): tiles per chunk cap, rows of the row-major images
  int tpc_cap, chrows;
  // attention-probability dropout
  float p_drop; uint64_t seed; const uint64_t* seed_base;   // key = seed + *seed_base (seed_base nullable: hipGraph replay)
};

__device__ __forceinline__ int region_of(int x, int n, int ws, int shift) { return x < n - ws ? 0 : (x < n - shift ? 1 : 2); }

// token row (in the [B*Hres*Wres] token-ordered tensors) and shift-mask region of window-local position i of window g
__device__ __forceinline__ void window_tok(const AttnP& p, int g, int i, int& tok, int& reg) {
  const int b = g / p.nW, w = g - b * p.nW;
  const int wr = w / p.nWw, wc = w - wr * p.nWw;
  const int pr = i / p.ws, pc = i - pr * p.ws;
  const int R = wr * p.ws + pr, C = wc * p.ws + pc;
  int r = R + p.shift, c = C + p.shift;
  r = r >= p.Hres ? r - p.Hres : r;
  c = c >= p.Wres ? c - p.Wres : c;
  tok = (b * p.Hres + r) * p.Wres + c;
  reg = p.shift > 0 ? region_of(R, p.Hres, p.ws, p.shift) * 3 + region_of(C, p.Wres, p.ws, p.shift) : 0;
}

__device__ __forceinline__ float group4_max(float v) { return rows4_max(v); }
__device__ __forceinline__ float group4_sum(float v) { return rows4_sum(v); }

// Head served by block y.  With head_dim 32 a head's slice of a token row is 64 B = half a cache line, and blocks y and y+1 (which
// share every line) land on different XCDs when grid.x == 1 (linear block id = y + H*z, XCD = id % 8): each XCD's L2 then fetches
// the whole line and the i2t kernels read 2 x their algorithmic bytes (PMC: 0.62 vs 0.30 GB forward).  Blocks y and y+8 share an
// XCD and are dispatched in the same round, so they take NEIGHBOURING heads: head = (y % 8) * (H / 8) + y / 8 (a bijection).
template <int D, bool WINDOW>
__device__ __forceinline__ int head_of(int y, int H) {
  if (D == 32 && !WINDOW && gridDim.x == 1 && (H & 7) == 0) return (y & 7) * (H >> 3) + (y >> 3);
  return y;
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = f2bf(a[e]); o[4 + e] = f2bf(b[e]); }
  return o;
}

// Stage `nrows` rows (row r of the chunk <-> global row rowmap[r]) of a [*, ld] bf16 matrix (head slice D wide) into a
// row-major LDS image.  Rows >= valid are zero filled.  Row stride D+16 elements (96 B / 160 B): 24 or 40 banks per row, which
// makes both ds_read_b128 (operand straight) and ds_read_b64_tr_b16 (operand transposed) conflict-free (MI355X_MICROARCH.md,
// LDS table); no transposed copy of an image is ever written.
template <int D, int U = 4>
__device__ __forceinline__ void stage(const bf16* __restrict__ src, int ld, int hcol, const int* rowmap, int nrows,
                                      int valid, bf16* rm) {
  constexpr int CPR = D / 8;
  // Four pieces per thread are REQUESTED before the first is written: the one-piece loop (load, wait, LDS write) exposed one global
  // latency per piece -- the t2i kernels (128-row chunks staged by 192 threads: 11 pieces per thread and chunk) spent ~16 us per chunk there
  // and ran at 1.5-2 TB/s of their bytes.
  const int total = nrows * CPR;
  for (int base = threadIdx.x; base < total; base += U * blockDim.x) {
    bf16x8 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * blockDim.x;
      const int r = idx / CPR, c = idx - r * CPR;
      if (idx < total && r < valid) {
        v[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)rowmap[r] * ld + hcol + c * 8);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = f2bf(0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * blockDim.x;
      const int r = idx / CPR, c = idx - r * CPR;
      if (idx < total) *reinterpret_cast<bf16x8*>(rm + r * (D + 16) + c * 8) = v[u];
    }
  }
}

// MFMA operand A = transposed fragment out of a ROW-MAJOR image: rows {t0*16+g*4..+3} U {t0*16+16+g*4..+3}, column d0 + l.
// ds_read_b64_tr_b16 hands lane l of a 16-lane group column l of the 4x16 block whose (row l>>2, 4-column piece l&3)
// address that lane supplies.
template <int D>
__device__ __forceinline__ bf16x8 trr_frag(const bf16* rm, int d0, int t0, int g, int l) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const bf16* a = rm + (t0 * 16 + g * 4 + (l >> 2)) * (D + 16) + d0 + (l & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 16 * (D + 16)));
  s16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
  return __builtin_bit_cast(bf16x8, o);
}

// ---------------------------------------------------------------------------------------------------------------
// Shared LDS layout (dynamic): [rowmap int CH][reg int CH][addmask float CH][aux float CH][woff int CH][btab float nb] then bf16 images
struct Lds {
  int* rowmap; int* reg; float* addmask; float* aux; int* woff; float* btab;
  bf16* rm0; bf16* rm1;
};
template <int D>
__device__ __forceinline__ Lds carve(char* base, int nbias, int n_rm, int chrows) {
  Lds L;
  L.rowmap = reinterpret_cast<int*>(base);
  L.reg = L.rowmap + CH;
  L.addmask = reinterpret_cast<float*>(L.reg + CH);
  L.aux = L.addmask + CH;
  L.woff = reinterpret_cast<int*>(L.aux + CH);
  L.btab = reinterpret_cast<float*>(L.woff + CH);
  size_t off = (size_t)(5 * CH + ((nbias + 3) & ~3)) * 4;
  bf16* img = reinterpret_cast<bf16*>(base + off);
  L.rm0 = img; img += (n_rm > 0) * chrows * (D + 16);
  L.rm1 = img; img += (n_rm > 1) * chrows * (D + 16);
  return L;
}
template <int D>
size_t lds_bytes(int nbias, int n_rm, int chrows) {
  return (size_t)(5 * CH + ((nbias + 3) & ~3)) * 4 + (size_t)n_rm * chrows * (D + 16) * 2;
}

// WINDOW mode: relative_position_index(i, j) = off(i) - off(j) + wconst with off(x) = row(x) * (2 ws - 1) + col(x)
// (swin_transformer.py:166-176).  off() of every staged row is tabulated once per chunk (Lds::woff) and each lane keeps its
// own, so a score costs two LDS reads instead of two integer divisions (the 576^2 configuration spends half its step here).
__device__ __forceinline__ int win_off(const AttnP& p, int x) {
  const int pr = x / p.ws;
  return pr * (2 * p.ws - 1) + (x - pr * p.ws);
}
__device__ __forceinline__ int win_const(const AttnP& p) { return (p.ws - 1) * (2 * p.ws - 1) + (p.ws - 1); }

// rows of chunk `c0..c0+n` of the KEY (or query) axis -> LDS rowmap/reg/addmask
__device__ __forceinline__ void fill_rowmeta(const AttnP& p, const Lds& L, int g, int base, int n, int len, bool keys) {
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    const int j = base + r;
    int tok = 0, reg = 0;
    float am = 0.f;
    if (j < len) {
      if (p.window) window_tok(p, g, j, tok, reg);
      else { tok = g * len + j; if (keys && p.kmask) am = p.kmask[(size_t)g * len + j] * 1.4426950408889634f; }   // log2 domain
    } else {
      am = -INFINITY;
    }
    L.rowmap[r] = tok; L.reg[r] = reg; L.addmask[r] = am;
    if (p.window) L.woff[r] = win_off(p, j < len ? j : len - 1);     // relative-position offset of this row (see win_off)
  }
}


// ===================================================== forward =================================================
template <int D, bool WINDOW, int NT = NKT>       // NT: key (query) tiles held per chunk -- 3 for the 40-token text side
__global__ __launch_bounds__(768) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = D / 32, DT = D / 16;
  const int nb = WINDOW ? (2 * p.ws - 1) * (2 * p.ws - 1) : 0;
  Lds L = carve<D>(smem, nb, 2, p.chrows);
  bf16* Ks = L.rm0; bf16* Vs = L.rm1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int gq = lane >> 4, lq = lane & 15;
  const int h = head_of<D, WINDOW>(blockIdx.y, p.H), g = blockIdx.z;
  for (int t = threadIdx.x; t < nb; t += blockDim.x) L.btab[t] = p.bias_table[(size_t)t * p.H + h] * 1.4426950408889634f;   // log2 domain
  const int ntiles = (p.Lk + 15) / 16, nchunk = (ntiles + p.tpc_cap - 1) / p.tpc_cap, tpc = (ntiles + nchunk - 1) / nchunk;
  const float sc2 = p.scale * 1.4426950408889634f;
  const uint32_t thresh = (uint32_t)((double)p.p_drop * 4294967296.0);
  const uint32_t dseed = p.p_drop > 0.f ? drop_seed32(p.seed + (p.seed_base ? *p.seed_base : 0ull)) : 0u;
  const float inv_keep = 1.f / (1.f - p.p_drop);
  // A key side that fits one chunk (text keys of the i2t cross-attention, text self-attention) is staged ONCE and the workgroup
  // then walks query strips blockIdx.x*nw + wave, + gridDim.x*nw, ... without any further barrier: with one strip per wave the
  // kernel was all per-workgroup set-up (row metadata, two barriers, staging) around a few MFMAs.
  const bool once = nchunk == 1;
  const int nstrips = (p.Lq + 15) / 16;
  if (once) {
    fill_rowmeta(p, L, g, 0, tpc * 16, p.Lk, true);
    __syncthreads();
    const int valid = min(tpc * 16, p.Lk);
    stage<D>(p.k, p.ldk, h * D, L.rowmap, tpc * 16, valid, Ks);
    stage<D>(p.v, p.ldv, h * D, L.rowmap, (tpc * 16 + 31) & ~31, valid, Vs);
    __syncthreads();
  }
  for (int strip = blockIdx.x * nw + wave;; strip += gridDim.x * nw) {
  if (once && strip >= nstrips) break;                  // (no barriers below in this mode)
  const int i = strip * 16 + lq;                      // this lane's query (window-local / sample-local)
  const bool qvalid = i < p.Lq;
  const int ioff = WINDOW ? win_off(p, qvalid ? i : 0) + win_const(p) : 0;   // this lane's query in relative-position offsets
  int qtok = 0, qreg = 0;
  {
    const int ic = qvalid ? i : p.Lq - 1;
    if (WINDOW) window_tok(p, g, ic, qtok, qreg); else qtok = g * p.Lq + ic;
  }

  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(p.q + (size_t)qtok * p.ldq + h * D + ks * 32 + gq * 8);

  float m = -INFINITY, lsum = 0.f;
  f32x4 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nchunk; ++c) {
    const int kbase = c * tpc * 16;
    if (!once) {
      __syncthreads();                                 // previous chunk fully consumed
      fill_rowmeta(p, L, g, kbase, tpc * 16, p.Lk, true);
      __syncthreads();
      const int valid = min(tpc * 16, p.Lk - kbase);
      stage<D>(p.k, p.ldk, h * D, L.rowmap, tpc * 16, valid, Ks);
      stage<D>(p.v, p.ldv, h * D, L.rowmap, (tpc * 16 + 31) & ~31, valid, Vs);
      __syncthreads();
    }
    f32x4 s[NT];
    float cmax = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (kt < tpc) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 16 + lq) * (D + 16) + ks * 32 + gq * 8);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], a, 0, 0, 0);   // S^T[key][query]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jl = kt * 16 + gq * 4 + r;
          float v = a[r] * sc2 + L.addmask[jl];               // scores in the log2 domain: exp is a bare v_exp_f32
          if (WINDOW) {
            v += L.btab[ioff - L.woff[jl]];
            if (L.reg[jl] != qreg) v += -144.26950408889634f;   // -100 * log2(e)
          }
          s[kt][r] = v;
          cmax = fmaxf(cmax, v);
        }
      }
    }
    cmax = group4_max(cmax);
    const float mnew = fmaxf(m, cmax);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);   // m = -inf on the first chunk -> 0
    m = mnew;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = kt < tpc ? __builtin_amdgcn_exp2f(s[kt][r] - mnew) : 0.f;
        psum += e;
        if (p.p_drop > 0.f) {
          e = drop_keep_rk(dseed, (uint32_t)((g * p.H + h) * p.Lq + (qvalid ? i : 0)), (uint32_t)(kbase + kt * 16 + gq * 4 + r), thresh) ? e * inv_keep : 0.f;
        }
        s[kt][r] = e;
      }
    }
    lsum = lsum * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
#pragma unroll
    for (int t2 = 0; t2 < NT / 2; ++t2) {
      if (t2 * 2 < tpc) {
        const bf16x8 pf = pack8(s[2 * t2], s[2 * t2 + 1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 vf = trr_frag<D>(Vs, dt * 16, 2 * t2, gq, lq);
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);   // O^T[d][query]
        }
      }
    }
  }
  lsum = group4_sum(lsum);
  if (qvalid) {
    const float inv = 1.f / lsum;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = f2bf(oacc[dt][r] * inv);
      *reinterpret_cast<bf16x4*>(p.o + (size_t)qtok * p.ldo + h * D + dt * 16 + gq * 4) = o;
    }
    if (gq == 0 && p.lse) p.lse[(size_t)qtok * p.H + h] = m * 0.6931471805599453f + __logf(lsum);
  }
  if (!once) break;
  }
}

// ===================================================== delta ===================================================
// delta[row, h] = sum_d dO[row, h*D+d] * O[row, h*D+d]
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o, const bf16* __restrict__ dout,
                                                         float* __restrict__ delta, int rows, int H, int ldo, int lddo) {
  constexpr int LPH = D / 8;                         // lanes per head
  const int vecs = H * LPH;
  for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < (size_t)rows * vecs; idx += (size_t)gridDim.x * 256) {
    const int row = idx / vecs, v = idx - (size_t)row * vecs;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(o + (size_t)row * ldo + v * 8);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(dout + (size_t)row * lddo + v * 8);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f(a[e]) * bf2f(b[e]);
#pragma unroll
    for (int off = 1; off < LPH; off <<= 1) s += __shfl_xor(s, off);
    if ((v & (LPH - 1)) == 0) delta[(size_t)row * H + v / LPH] = s;
  }
}

// ===================================================== backward, pass A: dQ (+ dbias) ==========================
template <int D, bool WINDOW, int NT = NKT>       // NT: key (query) tiles held per chunk -- 3 for the 40-token text side
// (WINDOW mode keeps its bias-gradient slice in registers, the 10-slot form 40 score registers per array: at most 8 waves per workgroup so
//  that the allocator has 256 registers -- the 168 of a 12-wave bound left 24-260 bytes of scratch per lane)
__global__ __launch_bounds__((WINDOW || NT > 6) ? 512 : 768) void attn_bwd_dq_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = D / 32, DT = D / 16;
  const int nb = WINDOW ? (2 * p.ws - 1) * (2 * p.ws - 1) : 0;
  Lds L = carve<D>(smem, nb, 2, p.chrows);
  bf16* Ks = L.rm0; bf16* Vs = L.rm1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int gq = lane >> 4, lq = lane & 15;
  const int h = head_of<D, WINDOW>(blockIdx.y, p.H);
  for (int t = threadIdx.x; t < nb; t += blockDim.x) L.btab[t] = p.bias_table[(size_t)t * p.H + h] * 1.4426950408889634f;   // log2 domain
  const int ntiles = (p.Lk + 15) / 16, nchunk = (ntiles + p.tpc_cap - 1) / p.tpc_cap, tpc = (ntiles + nchunk - 1) / nchunk;
  const uint32_t thresh = (uint32_t)((double)p.p_drop * 4294967296.0);
  const uint32_t dseed = p.p_drop > 0.f ? drop_seed32(p.seed + (p.seed_base ? *p.seed_base : 0ull)) : 0u;
  const float inv_keep = 1.f / (1.f - p.p_drop);

  const float sc2 = p.scale * 1.4426950408889634f;
  const int g0 = blockIdx.z * p.groups_per_block, g1 = min(p.G, g0 + p.groups_per_block);
  // WINDOW mode walks key chunks outermost so that the per-lane dbias accumulator only ever spans one chunk
  // (N = 324 at 576^2 needs three); dQ is then accumulated across chunk passes by the lane that owns it.
  const int nouter = WINDOW ? nchunk : 1;
  for (int co = 0; co < nouter; ++co) {
  f32x4 dbacc[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) dbacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int c_lo = WINDOW ? co : 0, c_hi = WINDOW ? co + 1 : nchunk;
  for (int g = g0; g < g1; ++g) {
    // single-chunk key side without window bias (i2t cross-attention, text self-attention): stage once, then walk the query
    // strips without further barriers (see attn_fwd_kernel)
    const bool once = !WINDOW && nchunk == 1;
    if (once) {
      __syncthreads();
      fill_rowmeta(p, L, g, 0, tpc * 16, p.Lk, true);
      __syncthreads();
      const int valid = min(tpc * 16, p.Lk);
      stage<D, 2>(p.k, p.ldk, h * D, L.rowmap, (tpc * 16 + 31) & ~31, valid, Ks);
      stage<D, 2>(p.v, p.ldv, h * D, L.rowmap, tpc * 16, valid, Vs);
      __syncthreads();
    }
    for (int strip = blockIdx.x * nw + wave;; strip += gridDim.x * nw) {
    if (once && strip >= (p.Lq + 15) / 16) break;
    const int i = strip * 16 + lq;
    const bool qvalid = i < p.Lq;
    const int ioff = WINDOW ? win_off(p, qvalid ? i : 0) + win_const(p) : 0;   // this lane's query in relative-position offsets
    int qtok = 0, qreg = 0;
    {
      const int ic = qvalid ? i : p.Lq - 1;
      if (WINDOW) window_tok(p, g, ic, qtok, qreg); else qtok = g * p.Lq + ic;
    }
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(p.q + (size_t)qtok * p.ldq + h * D + ks * 32 + gq * 8);
      dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + (size_t)qtok * p.lddo + h * D + ks * 32 + gq * 8);
    }
    const float lse = p.lse[(size_t)qtok * p.H + h] * 1.4426950408889634f, dlt = p.delta[(size_t)qtok * p.H + h];   // lse in the log2 domain
    f32x4 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = c_lo; c < c_hi; ++c) {
      const int kbase = c * tpc * 16;
      if (!once) {
        __syncthreads();
        fill_rowmeta(p, L, g, kbase, tpc * 16, p.Lk, true);
        __syncthreads();
        const int valid = min(tpc * 16, p.Lk - kbase);
        stage<D, 2>(p.k, p.ldk, h * D, L.rowmap, (tpc * 16 + 31) & ~31, valid, Ks);
        stage<D, 2>(p.v, p.ldv, h * D, L.rowmap, tpc * 16, valid, Vs);
        __syncthreads();
      }
#pragma unroll
      for (int t2 = 0; t2 < NT / 2; ++t2) {
        if (t2 * 2 < tpc) {
          f32x4 ds[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int kt = 2 * t2 + u;
            ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < tpc) {
              f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 16 + lq) * (D + 16) + ks * 32 + gq * 8);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (kt * 16 + lq) * (D + 16) + ks * 32 + gq * 8);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], a, 0, 0, 0);     // S^T
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dp, 0, 0, 0);  // dP^T
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int jl = kt * 16 + gq * 4 + r;
                float sv = a[r] * sc2 + L.addmask[jl];
                if (WINDOW) {
                  sv += L.btab[ioff - L.woff[jl]];
                  if (L.reg[jl] != qreg) sv += -144.26950408889634f;
                }
                const float pr = __builtin_amdgcn_exp2f(sv - lse);
                float dpe = dp[r];
                if (p.p_drop > 0.f) {
                  dpe = drop_keep_rk(dseed, (uint32_t)((g * p.H + h) * p.Lq + (qvalid ? i : 0)), (uint32_t)(kbase + jl), thresh) ? dpe * inv_keep : 0.f;
                }
                const float d = qvalid ? pr * (dpe - dlt) : 0.f;
                ds[u][r] = d;
              }
              if (WINDOW) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dbacc[kt][r] += ds[u][r];
              }
            }
          }
          const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const bf16x8 ktf = trr_frag<D>(Ks, dt * 16, 2 * t2, gq, lq);
            dqacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dqacc[dt], 0, 0, 0);   // dQ^T[d][query]
          }
        }
      }
    }
    if (qvalid) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4* dst = reinterpret_cast<bf16x4*>(p.dq + (size_t)qtok * p.lddq + h * D + dt * 16 + gq * 4);
        bf16x4 o;
        if (WINDOW && co > 0) {
          const bf16x4 prev = *dst;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(bf2f(prev[r]) + dqacc[dt][r] * p.scale);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(dqacc[dt][r] * p.scale);
        }
        *dst = o;
      }
    }
    if (!once) break;
    }
  }
  const int i = (blockIdx.x * nw + wave) * 16 + lq;     // (window mode: one strip per wave)
  const bool qvalid = i < p.Lq;
  if (WINDOW && p.dbias_part && qvalid) {
    float* dst = p.dbias_part + (((size_t)blockIdx.z * p.H + h) * p.Lq + i) * p.Lk + co * tpc * 16;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (kt < tpc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt * 16 + gq * 4 + r;
          if (co * tpc * 16 + j < p.Lk) dst[j] = dbacc[kt][r];
        }
      }
    }
  }
  }
}

// ===================================================== backward, pass B: dK, dV ================================
template <int D, bool WINDOW, int NT = NKT>       // NT: key (query) tiles held per chunk -- 3 for the 40-token text side
// (the 10-slot form at D = 64 carries 8 more registers for the next strip's K / V: bounded at 8 waves so that it does not spill)
__global__ __launch_bounds__((!WINDOW && NT > 6 && D == 64) ? 512 : 768) void attn_bwd_dkv_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = D / 32, DT = D / 16;
  const int nb = WINDOW ? (2 * p.ws - 1) * (2 * p.ws - 1) : 0;
  Lds L = carve<D>(smem, nb, 2, p.chrows);
  bf16* Qs = L.rm0; bf16* dOs = L.rm1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int gq = lane >> 4, lq = lane & 15;
  const int h = head_of<D, WINDOW>(blockIdx.y, p.H), g = blockIdx.z;
  const int ntiles = (p.Lq + 15) / 16, nchunk = (ntiles + p.tpc_cap - 1) / p.tpc_cap, tpc = (ntiles + nchunk - 1) / nchunk;
  const int nstrips = (p.Lk + 15) / 16;
  // `once`: the query side fits ONE staged chunk (the 40-token text side of t2i / text self-attention): it is staged once per workgroup and
  // the waves walk key strips blockIdx.x*nw + wave, + gridDim.x*nw, ... without any further barrier (round 4; one strip per wave and a
  // re-staged chunk per 9 strips left this pass all set-up: three barriers and 10 KB of staging for ~20 MFMAs per wave).
  const bool once = !WINDOW && nchunk == 1;
  auto stage_chunk = [&](int qbase) {
    __syncthreads();
    fill_rowmeta(p, L, g, qbase, tpc * 16, p.Lq, false);
    __syncthreads();
    const int valid = min(tpc * 16, p.Lq - qbase);
    const int nst = (tpc * 16 + 31) & ~31;
    stage<D>(p.q, p.ldq, h * D, L.rowmap, nst, valid, Qs);
    stage<D>(p.dout, p.lddo, h * D, L.rowmap, nst, valid, dOs);
    for (int r = threadIdx.x; r < tpc * 16; r += blockDim.x) {   // per-query lse (addmask slot) and delta (aux slot)
      const bool ok = r < valid;
      L.addmask[r] = ok ? p.lse[(size_t)L.rowmap[r] * p.H + h] * 1.4426950408889634f : INFINITY;   // log2 domain; +inf -> p = exp2(-inf) = 0 for padded queries
      L.aux[r] = ok ? p.delta[(size_t)L.rowmap[r] * p.H + h] : 0.f;
    }
    __syncthreads();
  };
  for (int t = threadIdx.x; t < nb; t += blockDim.x) L.btab[t] = p.bias_table[(size_t)t * p.H + h] * 1.4426950408889634f;   // log2 domain
  const uint32_t thresh = (uint32_t)((double)p.p_drop * 4294967296.0);
  const uint32_t dseed = p.p_drop > 0.f ? drop_seed32(p.seed + (p.seed_base ? *p.seed_base : 0ull)) : 0u;
  const float inv_keep = 1.f / (1.f - p.p_drop);
  const float sc2 = p.scale * 1.4426950408889634f;
  // K / V rows of this lane's key in strip `st` (the next strip's are requested before the current one is computed: `once` mode)
  auto load_kv = [&](int st, bf16x8* kfo, bf16x8* vfo) {
    if (WINDOW) return;
    const int jn = min(st * 16 + lq, p.Lk - 1);
    const size_t tok = (size_t)g * p.Lk + jn;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kfo[ks] = *reinterpret_cast<const bf16x8*>(p.k + tok * p.ldk + h * D + ks * 32 + gq * 8);
      vfo[ks] = *reinterpret_cast<const bf16x8*>(p.v + tok * p.ldv + h * D + ks * 32 + gq * 8);
    }
  };
  bf16x8 kfn[KS], vfn[KS];
  if (once && (int)(blockIdx.x * nw + wave) < nstrips) load_kv(blockIdx.x * nw + wave, kfn, vfn);
  if (once) stage_chunk(0);
  for (int strip = blockIdx.x * nw + wave; ; strip += gridDim.x * nw) {
    if (once && strip >= nstrips) break;
    const int j = strip * 16 + lq;                      // this lane's key
    const bool kvalid = j < p.Lk;
    const int joff = WINDOW ? win_off(p, kvalid ? j : 0) - win_const(p) : 0;    // this lane's key
    int ktok = 0, kreg = 0;
    float kadd = 0.f;
    {
      const int jc = kvalid ? j : p.Lk - 1;
      if (WINDOW) window_tok(p, g, jc, ktok, kreg);
      else { ktok = g * p.Lk + jc; if (p.kmask) kadd = p.kmask[(size_t)g * p.Lk + jc] * 1.4426950408889634f; }
    }
    bf16x8 kf[KS], vf[KS];
    if (once) {
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) { kf[ks] = kfn[ks]; vf[ks] = vfn[ks]; }
      if (strip + (int)(gridDim.x * nw) < nstrips) load_kv(strip + gridDim.x * nw, kfn, vfn);
    } else {
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(p.k + (size_t)ktok * p.ldk + h * D + ks * 32 + gq * 8);
        vf[ks] = *reinterpret_cast<const bf16x8*>(p.v + (size_t)ktok * p.ldv + h * D + ks * 32 + gq * 8);
      }
    }
    f32x4 dkacc[DT], dvacc[DT];
  #pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int c = 0; c < nchunk; ++c) {
      const int qbase = c * tpc * 16;
      if (!once) stage_chunk(qbase);
  #pragma unroll
      for (int t2 = 0; t2 < NT / 2; ++t2) {
        if (t2 * 2 < tpc) {
          f32x4 ds[2], pd[2];
  #pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int qt = 2 * t2 + u;
            ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            pd[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qt < tpc) {
              f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + (qt * 16 + lq) * (D + 16) + ks * 32 + gq * 8);
                const bf16x8 df = *reinterpret_cast<const bf16x8*>(dOs + (qt * 16 + lq) * (D + 16) + ks * 32 + gq * 8);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], a, 0, 0, 0);    // S[query][key]
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf[ks], dp, 0, 0, 0);  // dP[query][key]
              }
  #pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int il = qt * 16 + gq * 4 + r;      // chunk-local query
                const int ig = min(qbase + il, p.Lq - 1);
                float sv = a[r] * sc2 + kadd;
                if (WINDOW) {
                  sv += L.btab[L.woff[il] - joff];
                  if (L.reg[il] != kreg) sv += -144.26950408889634f;
                }
                float pr = kvalid ? __builtin_amdgcn_exp2f(sv - L.addmask[il]) : 0.f;
                float dpe = dp[r], prd = pr;
                if (p.p_drop > 0.f) {
                  const bool keep = drop_keep_rk(dseed, (uint32_t)((g * p.H + h) * p.Lq + ig), (uint32_t)(kvalid ? j : 0), thresh);
                  dpe = keep ? dpe * inv_keep : 0.f;
                  prd = keep ? pr * inv_keep : 0.f;
                }
                ds[u][r] = pr * (dpe - L.aux[il]);
                pd[u][r] = prd;
              }
            }
          }
          const bf16x8 dsf = pack8(ds[0], ds[1]), pf = pack8(pd[0], pd[1]);
  #pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const bf16x8 qtf = trr_frag<D>(Qs, dt * 16, 2 * t2, gq, lq);
            const bf16x8 dotf = trr_frag<D>(dOs, dt * 16, 2 * t2, gq, lq);
            dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dkacc[dt], 0, 0, 0);   // dK^T[d][key]
            dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotf, pf, dvacc[dt], 0, 0, 0);   // dV^T[d][key]
          }
        }
      }
    }
    if (kvalid) {
  #pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4 ok, ov;
  #pragma unroll
        for (int r = 0; r < 4; ++r) { ok[r] = f2bf(dkacc[dt][r] * p.scale); ov[r] = f2bf(dvacc[dt][r]); }
        *reinterpret_cast<bf16x4*>(p.dk + (size_t)ktok * p.lddk + h * D + dt * 16 + gq * 4) = ok;
        *reinterpret_cast<bf16x4*>(p.dv + (size_t)ktok * p.lddv + h * D + dt * 16 + gq * 4) = ov;
      }
    }
    if (!once) break;
  }
}

// dtable[rel_index(i,j), h] += sum_z part[z, h, i, j]   (swin_transformer.py:166-176 index; dtable pre-zeroed)
// block = (query i, head h): threads stride over keys j, the z-sum reads are coalesced along j.
__global__ __launch_bounds__(256) void dbias_scatter_kernel(const float* __restrict__ part, float* __restrict__ dtable,
                                                            int nz, int H, int ws) {
  const int i = blockIdx.x, h = blockIdx.y, N = ws * ws, W2 = 2 * ws - 1;
  const int pri = i / ws, pci = i - pri * ws;
  for (int j = threadIdx.x; j < N; j += 256) {
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += part[(((size_t)z * H + h) * N + i) * N + j];
    const int prj = j / ws, pcj = j - prj * ws;
    atomicAdd(dtable + (size_t)((pri - prj + ws - 1) * W2 + (pci - pcj + ws - 1)) * H + h, s);
  }
}

// waves (16-row strips) per workgroup.  `staged` = rows of the OTHER operand every workgroup stages into LDS: when that is a
// short text sequence (i2t cross-attention: 40-50 keys) small workgroups win -- three or four of them share a CU and overlap
// each other's staging / softmax phases (op_bench, 576 queries x 40 keys: forward 812 -> 543 us) -- while a long staged side
// (t2i backward: 576 keys per 40 queries) wants few, large workgroups so it is staged fewer times.
int pick_waves(int nstrips, int staged) {
  static const int force = getenv("FIBER_ATTN_WAVES") ? atoi(getenv("FIBER_ATTN_WAVES")) : 0;
  if (force > 0) return nstrips < force ? nstrips : force;
  if (staged <= 64 && nstrips > 4) return nstrips % 3 == 0 ? 3 : 4;
  return nstrips <= 12 ? nstrips : (nstrips % 9 == 0 ? 9 : 8);
}

// LDS chunk geometry for a launch that stages `staged_len` rows per group with `nw` waves per workgroup.  The images are
// sized to the chunk actually used (a 40-token text sequence needs 48 rows, not 160), and small workgroups take chunks of at
// most 5 tiles: with the full 160-row layout a 3-wave workgroup of the D = 64 kernels owned 47-91 KB of LDS, i.e. one or two
// workgroups (3-6 waves) per CU.
void launch_geometry(AttnP& p, int staged_len, int nw, bool backward) {
  const int ntiles = cdiv(staged_len, 16);
  // small workgroups want chunks short enough for three or four of them to share a CU's LDS: two row-major images of
  // chrows x (D+16) plus 3.2 KB of tables.  Forward: 8 tiles (128 rows, 44 KB at D = 64 -> 3 workgroups; the full 160-row chunk
  // leaves 2 and measured 386 -> 486 us on the t2i shape, 5-tile chunks add barriers: 422 us); backward: 5 tiles.
  p.tpc_cap = (nw <= 4 && !p.window && ntiles > (backward ? 5 : 8)) ? (backward ? 5 : 8) : NKT;
  const int nchunk = cdiv(ntiles, p.tpc_cap), tpc = cdiv(ntiles, nchunk);
  p.chrows = (tpc * 16 + 31) & ~31;
}

// grid.x of the query-strip kernels.  When the staged side fits one chunk the kernels stage it once per workgroup and loop over
// strips, so only as many workgroups per (head, group) are launched as it takes to fill the chip (~8 per CU overall).
int strip_blocks(const AttnP& p, int staged_len, int nstrips, int nw, int hg, bool window_ok) {
  const int full = cdiv(nstrips, nw);
  const bool once = cdiv(cdiv(staged_len, 16), p.tpc_cap) == 1 && (window_ok || !p.window);
  if (!once) return full;
  const int want = cdiv(2048, hg);
  return want < 1 ? 1 : (want < full ? want : full);
}

// chunks of at most 4 tiles (a 40-token text side: 3): the kernels instantiated with 4 tile slots instead of 10 hold 24 fewer score
// registers per array and fit more waves per SIMD (the strip-walking kernels are bound by the latency of their per-strip loads)
int chunk_tiles(const AttnP& p, int staged_len) {
  const int ntiles = cdiv(staged_len, 16), nchunk = cdiv(ntiles, p.tpc_cap);
  return cdiv(ntiles, nchunk);
}
bool small_chunk(const AttnP& p, int staged_len) {
  static const int off = getenv("FIBER_ATTN_NOSMALL") ? 1 : 0;
  const int ntiles = cdiv(staged_len, 16), nchunk = cdiv(ntiles, p.tpc_cap);
  return !off && cdiv(ntiles, nchunk) <= 4;
}

template <int D>
int launch_fwd(AttnP& p, hipStream_t st) {
  const int nstrips = cdiv(p.Lq, 16), nw = pick_waves(nstrips, p.Lk);
  const int nb = p.window ? (2 * p.ws - 1) * (2 * p.ws - 1) : 0;
  launch_geometry(p, p.Lk, nw, false);
  const size_t sh = lds_bytes<D>(nb, 2, p.chrows);
  const int gx = strip_blocks(p, p.Lk, nstrips, nw, p.H * p.G, true);
  if (p.window) hipLaunchKernelGGL((attn_fwd_kernel<D, true>), dim3(gx, p.H, p.G), dim3(64 * nw), sh, st, p);
  else if (small_chunk(p, p.Lk)) hipLaunchKernelGGL((attn_fwd_kernel<D, false, 4>), dim3(gx, p.H, p.G), dim3(64 * nw), sh, st, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<D, false>), dim3(gx, p.H, p.G), dim3(64 * nw), sh, st, p);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

template <int D>
int launch_bwd(AttnP& p, float* delta, float* dbias_table, float* dbias_ws, int nz, hipStream_t st) {
  const int rows_q = p.G * p.Lq;
  {
    const size_t total = (size_t)rows_q * p.H * (D / 8);
    int grid = (int)((total + 255) / 256);
    grid = grid > 2048 ? 2048 : grid;
    hipLaunchKernelGGL((attn_delta_kernel<D>), dim3(grid), dim3(256), 0, st, (const bf16*)p.o, p.dout, delta, rows_q, p.H, p.ldo, p.lddo);
    FIBER_CHECK_LAUNCH();
  }
  p.delta = delta;
  const int nb = p.window ? (2 * p.ws - 1) * (2 * p.ws - 1) : 0;
  {
    const int nstrips = cdiv(p.Lq, 16);
    int nw = pick_waves(nstrips, p.Lk);
    launch_geometry(p, p.Lk, nw, true);                  // (the chunking only changes at nw <= 4, the cap below only acts on nw > 8)
    const bool wide_form = p.window || (!small_chunk(p, p.Lk) && chunk_tiles(p, p.Lk) > 6);
    if (wide_form && nw > 8) nw = 8;                     // launch bound of those instantiations (strips beyond 8 go to further workgroups)
    int gz = p.G;
    p.groups_per_block = 1;
    p.dbias_part = nullptr;
    if (p.window) {
      gz = nz;
      p.groups_per_block = cdiv(p.G, nz);
      gz = cdiv(p.G, p.groups_per_block);
      p.dbias_part = dbias_ws;
    }
    launch_geometry(p, p.Lk, nw, true);
    const size_t sh = lds_bytes<D>(nb, 2, p.chrows);
    if (p.window) hipLaunchKernelGGL((attn_bwd_dq_kernel<D, true>), dim3(cdiv(nstrips, nw), p.H, gz), dim3(64 * nw), sh, st, p);
    else if (small_chunk(p, p.Lk)) hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false, 4>), dim3(strip_blocks(p, p.Lk, nstrips, nw, p.H * gz, false), p.H, gz), dim3(64 * nw), sh, st, p);
    else if (chunk_tiles(p, p.Lk) <= 6) hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false, 6>), dim3(strip_blocks(p, p.Lk, nstrips, nw, p.H * gz, false), p.H, gz), dim3(64 * nw), sh, st, p);   // (t2i: 5-tile chunks; the 10-slot form spills)
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false>), dim3(strip_blocks(p, p.Lk, nstrips, nw, p.H * gz, false), p.H, gz), dim3(64 * nw), sh, st, p);
    FIBER_CHECK_LAUNCH();
    if (p.window) {
      if (hipMemsetAsync(dbias_table, 0, (size_t)nb * p.H * sizeof(float), st) != hipSuccess) return FIBER_ELAUNCH;
      hipLaunchKernelGGL(dbias_scatter_kernel, dim3(p.Lq, p.H), dim3(256), 0, st, dbias_ws, dbias_table, gz, p.H, p.ws);
      FIBER_CHECK_LAUNCH();
    }
  }
  {
    const int nstrips = cdiv(p.Lk, 16);
    int nw = pick_waves(nstrips, 1 << 20);               // key-strip pass: measured worse with small workgroups
    launch_geometry(p, p.Lq, nw, true);
    if (!p.window && D == 64 && !small_chunk(p, p.Lq) && nw > 8) nw = 8;   // launch bound of that instantiation (the chunking only changes at nw <= 4)
    const size_t sh = lds_bytes<D>(nb, 2, p.chrows);
    // query side in one staged chunk: the kernel stages it once and walks key strips, so only as many workgroups per (head, sample) as it
    // takes to fill the chip (strip_blocks); otherwise one workgroup per nw strips
    static const int once_env = getenv("FIBER_ATTN_DKV_ONCE") ? atoi(getenv("FIBER_ATTN_DKV_ONCE")) : 1;   // 0: one strip per wave (A/B runs)
    const int gx = once_env ? strip_blocks(p, p.Lq, nstrips, nw, p.H * p.G, false) : cdiv(nstrips, nw);
    if (p.window) hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, true>), dim3(cdiv(nstrips, nw), p.H, p.G), dim3(64 * nw), sh, st, p);
    else if (small_chunk(p, p.Lq)) hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, false, 4>), dim3(gx, p.H, p.G), dim3(64 * nw), sh, st, p);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, false>), dim3(gx, p.H, p.G), dim3(64 * nw), sh, st, p);
    FIBER_CHECK_LAUNCH();
  }
  return FIBER_OK;
}

bool attr_done[16] = {};
void ensure_attrs() {
  if (!fiber_first_on_device(attr_done)) return;
  const int big = 160 * 1024;
  hipFuncSetAttribute((const void*)attn_fwd_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_fwd_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_fwd_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_fwd_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
}

}  // namespace

// specialised persistent window kernels (win_attn.hip)
int fiber_win_fwd_launch(const void* qkv, const float* bias_table, void* o, float* lse, int B, int Hres, int Wres, int C,
                         int heads, int ws, int shift, int hmajor, hipStream_t st);
int fiber_win_bwd_slices(int n_windows, int heads);
int fiber_win_bwd_launch(const void* qkv, const float* bias_table, const void* o, const void* dout, const float* lse,
                         void* dqkv, float* dbias_table, float* delta_ws, float* dbias_ws, float* dqkv_colsum, float* colsum_ws,
                         int B, int Hres, int Wres, int C, int heads, int ws, int shift, int hmajor, hipStream_t st);
int fiber_win_colsum_rows(int n_windows, int heads, int N);
// one-pass backward of image -> text cross attention (attn_x.hip); FIBER_EINVAL = shape not served
int fiber_i2t_bwd_launch(const void* q, const void* k, const void* v, const float* kmask, const void* o, const void* dout, const float* lse,
                         void* dq, void* dk, void* dv, int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int lddo,
                         int lddq, int lddk, int lddv, float scale, hipStream_t st);
// the same for head_dim 64 and at most 48 queries (text -> image cross attention, text self attention), dropout included
int fiber_t2i_bwd_launch(const void* q, const void* k, const void* v, const float* kmask, const void* o, const void* dout, const float* lse,
                         void* dq, void* dk, void* dv, int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int lddo,
                         int lddq, int lddk, int lddv, float scale, float p_drop, uint64_t seed, const uint64_t* seed_base, hipStream_t st);
int fiber_t2i_fwd_launch(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int heads, int Lq, int Lk,
                         int ldq, int ldk, int ldv, int ldo, float scale, float p_drop, uint64_t seed, const uint64_t* seed_base, hipStream_t st);
int fiber_i2t_fwd_launch(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int heads, int Lq, int Lk,
                         int ldq, int ldk, int ldv, int ldo, float scale, hipStream_t st);

// --------------------------------------------------------------------------------------------------- C ABI
// Window attention in image-token order.  qkv: [B*Hres*Wres, 3C] bf16 with channel layout [3][heads][32]
// (swin_transformer.py:202) or, with head_major = 1, [heads][3][32] (the caller permutes the qkv weight rows: q|k|v of a
// head become one contiguous 192-byte run per token, which raises cache-line efficiency of the per-head gathers); o: [B*Hres*Wres, C]; bias_table fp32 [(2ws-1)^2, heads]; lse: fp32, B*Hres*Wres*heads values whose LAYOUT is private to the path a window size takes -- [image][head][token] for N <= 336 (win_attn.hip), [token][head] for the generic path; forward and backward of one size always take the same path, nothing else may read it.
// shift = 0 disables the cyclic shift and the region mask.  head_dim must be 32.
extern "C" int fiber_window_attn_fwd_bf16(const void* qkv, const float* bias_table, void* o, float* lse, int B, int Hres,
                                          int Wres, int C, int heads, int ws, int shift, int head_major, hipStream_t stream) {
  if (C != heads * 32 || Hres % ws || Wres % ws || shift < 0 || shift >= ws) return FIBER_EINVAL;
  if (ws * ws <= 336) return fiber_win_fwd_launch(qkv, bias_table, o, lse, B, Hres, Wres, C, heads, ws, shift, head_major, stream);
  if (head_major) return FIBER_EINVAL;                 // the generic (N > 336) path reads the reference layout only
  ensure_attrs();
  AttnP p{};
  const bf16* base = (const bf16*)qkv;
  p.q = base; p.k = base + C; p.v = base + 2 * C; p.o = (bf16*)o; p.lse = lse;
  p.ldq = p.ldk = p.ldv = 3 * C; p.ldo = C;
  p.H = heads; p.Lq = p.Lk = ws * ws; p.nWw = Wres / ws; p.nW = (Hres / ws) * p.nWw; p.G = B * p.nW;
  p.scale = 0.17677669529663687f;  // 32^-0.5
  p.window = 1; p.Hres = Hres; p.Wres = Wres; p.ws = ws; p.shift = shift; p.bias_table = bias_table;
  return launch_fwd<32>(p, stream);
}

// Number of partial-gradient slices pass A uses for B*nW windows; workspace = slices * heads * N * N floats.
extern "C" int fiber_window_attn_bwd_slices(int n_windows, int heads) { return fiber_win_bwd_slices(n_windows, heads); }
// Rows of the column-sum workspace (0: this window size has no fused column sums, pass NULL for both pointers).
extern "C" int fiber_window_attn_colsum_rows(int n_windows, int heads, int ws) {
  return ws * ws <= 336 ? fiber_win_colsum_rows(n_windows, heads, ws * ws) : 0;
}

// Backward of the above.  dqkv: [B*Hres*Wres, 3C] (fully written); dbias_table fp32 [(2ws-1)^2, heads] (overwritten);
// delta_ws: fp32 [B*Hres*Wres*heads]; dbias_ws: fp32 [slices*heads*N*N].
// Optional: dqkv_colsum fp32 [3C] = column sums of dqkv (the bias gradient of the qkv linear, swin_transformer.py:197),
// produced inside the two passes instead of by another pass over dqkv; colsum_ws fp32 [colsum_rows * 3C].  Both or neither.
extern "C" int fiber_window_attn_bwd_bf16(const void* qkv, const float* bias_table, const void* o, const void* dout,
                                          const float* lse, void* dqkv, float* dbias_table, float* delta_ws,
                                          float* dbias_ws, float* dqkv_colsum, float* colsum_ws, int B, int Hres, int Wres,
                                          int C, int heads, int ws, int shift, int head_major, hipStream_t stream) {
  if (C != heads * 32 || Hres % ws || Wres % ws || shift < 0 || shift >= ws) return FIBER_EINVAL;
  if (ws * ws <= 336)
    return fiber_win_bwd_launch(qkv, bias_table, o, dout, lse, dqkv, dbias_table, delta_ws, dbias_ws, dqkv_colsum, colsum_ws, B,
                                Hres, Wres, C, heads, ws, shift, head_major, stream);
  if (head_major || dqkv_colsum || colsum_ws) return FIBER_EINVAL;   // (fiber_window_attn_colsum_rows() returns 0 for this path)
  ensure_attrs();
  AttnP p{};
  const bf16* base = (const bf16*)qkv;
  bf16* dbase = (bf16*)dqkv;
  p.q = base; p.k = base + C; p.v = base + 2 * C; p.o = (bf16*)o; p.lse = (float*)lse; p.dout = (const bf16*)dout;
  p.dq = dbase; p.dk = dbase + C; p.dv = dbase + 2 * C;
  p.ldq = p.ldk = p.ldv = p.lddq = p.lddk = p.lddv = 3 * C; p.ldo = p.lddo = C;
  p.H = heads; p.Lq = p.Lk = ws * ws; p.nWw = Wres / ws; p.nW = (Hres / ws) * p.nWw; p.G = B * p.nW;
  p.scale = 0.17677669529663687f;
  p.window = 1; p.Hres = Hres; p.Wres = Wres; p.ws = ws; p.shift = shift; p.bias_table = bias_table;
  return launch_bwd<32>(p, delta_ws, dbias_table, dbias_ws, fiber_window_attn_bwd_slices(p.G, heads), stream);
}

// Generic multi-head attention: q [B*Lq, ldq], k/v [B*Lk, ldk/ldv], o [B*Lq, ldo]; head h occupies columns
// [h*D, (h+1)*D) of each; kmask: additive fp32 [B, Lk] or NULL; scale applied to q.k^T; p_drop/seed: attention-prob
// dropout (0 disables).  D in {32, 64}.  lse fp32 [B*Lq, heads].
extern "C" int fiber_mha_fwd_bf16(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse,
                                  int B, int heads, int Lq, int Lk, int D, int ldq, int ldk, int ldv, int ldo,
                                  float scale, float p_drop, uint64_t seed, const uint64_t* seed_base, hipStream_t stream) {
  if ((D != 32 && D != 64) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || Lq <= 0 || Lk <= 0) return FIBER_EINVAL;
  // few queries, head_dim 64 (text -> image cross attention, text self attention): streaming forward with Q in registers (attn_x.hip)
  static const int onepass_t = getenv("FIBER_ATTN_T2I_ONEPASS") ? atoi(getenv("FIBER_ATTN_T2I_ONEPASS")) : 1;   // 0: the generic kernel (A/B runs)
  static const int onepass_i = getenv("FIBER_ATTN_I2T_ONEPASS") ? atoi(getenv("FIBER_ATTN_I2T_ONEPASS")) : 1;
  if (onepass_i && D == 32 && Lk <= 48 && p_drop == 0.f) {   // few keys, head_dim 32 (image -> text cross attention): K / V in registers
    const int rc = fiber_i2t_fwd_launch(q, k, v, kmask, o, lse, B, heads, Lq, Lk, ldq, ldk, ldv, ldo, scale, stream);
    if (rc != FIBER_EINVAL) return rc;
  }
  if (onepass_t && D == 64 && Lq <= 48) {
    const int rc = fiber_t2i_fwd_launch(q, k, v, kmask, o, lse, B, heads, Lq, Lk, ldq, ldk, ldv, ldo, scale, p_drop, seed, seed_base, stream);
    if (rc != FIBER_EINVAL) return rc;
  }
  ensure_attrs();
  AttnP p{};
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = lse;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.H = heads; p.Lq = Lq; p.Lk = Lk; p.G = B; p.scale = scale; p.kmask = kmask; p.p_drop = p_drop; p.seed = seed; p.seed_base = seed_base;
  return D == 32 ? launch_fwd<32>(p, stream) : launch_fwd<64>(p, stream);
}

// Backward: dq/dk/dv have the same row layout as q/k/v with leading dims lddq/lddk/lddv; delta_ws fp32 [B*Lq*heads].
extern "C" int fiber_mha_bwd_bf16(const void* q, const void* k, const void* v, const float* kmask, const void* o,
                                  const void* dout, const float* lse, void* dq, void* dk, void* dv, float* delta_ws,
                                  int B, int heads, int Lq, int Lk, int D, int ldq, int ldk, int ldv, int ldo, int lddo,
                                  int lddq, int lddk, int lddv, float scale, float p_drop, uint64_t seed,
                                  const uint64_t* seed_base, hipStream_t stream) {
  if ((D != 32 && D != 64) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (lddo & 7) || (lddq & 3) || (lddk & 3) ||
      (lddv & 3) || Lq <= 0 || Lk <= 0)
    return FIBER_EINVAL;
  // few keys, head_dim 32, no dropout (image -> text cross attention): one pass with K / V in registers (delta_ws stays unused)
  static const int onepass = getenv("FIBER_ATTN_I2T_ONEPASS") ? atoi(getenv("FIBER_ATTN_I2T_ONEPASS")) : 1;   // 0: the generic three kernels (A/B runs)
  if (onepass && D == 32 && Lk <= 48 && p_drop == 0.f && (Lq & 15) == 0 && (heads & 3) == 0 && !(lddq & 7)) {
    const int rc = fiber_i2t_bwd_launch(q, k, v, kmask, o, dout, lse, dq, dk, dv, B, heads, Lq, Lk, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv,
                                        scale, stream);
    if (rc != FIBER_EINVAL) return rc;
  }
  // few queries, head_dim 64 (text -> image cross attention, text self attention): one pass with Q / dO in registers
  static const int onepass_t = getenv("FIBER_ATTN_T2I_ONEPASS") ? atoi(getenv("FIBER_ATTN_T2I_ONEPASS")) : 1;
  if (onepass_t && D == 64 && Lq <= 48) {
    const int rc = fiber_t2i_bwd_launch(q, k, v, kmask, o, dout, lse, dq, dk, dv, B, heads, Lq, Lk, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv,
                                        scale, p_drop, seed, seed_base, stream);
    if (rc != FIBER_EINVAL) return rc;
  }
  ensure_attrs();
  AttnP p{};
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = (float*)lse;
  p.dout = (const bf16*)dout; p.dq = (bf16*)dq; p.dk = (bf16*)dk; p.dv = (bf16*)dv;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.H = heads; p.Lq = Lq; p.Lk = Lk; p.G = B; p.scale = scale; p.kmask = kmask; p.p_drop = p_drop; p.seed = seed; p.seed_base = seed_base;
  return D == 32 ? launch_bwd<32>(p, delta_ws, nullptr, nullptr, 1, stream) : launch_bwd<64>(p, delta_ws, nullptr, nullptr, 1, stream);
}
