// Weight-gradient GEMM of every nn.Linear on the FIBER fused-backbone path (gfx950 / CDNA4):
//
//   dW[N,K] = dY[M,N]^T . X[M,K]        dY, X row-major bf16 (the layouts the forward pass left them in), fp32 result
//   db[N]   = column sums of dY         (optional: the bias gradient rides in the same pass)
//
// This is the implicit autograd of swin_transformer.py:197,221,233,238,257 (qkv / proj / i2t linears), timm Mlp fc1 / fc2
// (:325), PatchMerging.reduction (:431), PatchEmbed.proj, roberta.py:231-241,337,398,415 and fiber_module.py:349-350, which
// the reference leaves to ATen (addmm backward = a TN GEMM + a column-sum kernel per linear).
//
// "TN": the contraction runs over M, the SLOW dimension of both operands, and M is huge (B*L = 10^4..10^6 rows) while the
// result is small (<= 4096 x 1024).  So
//  * the M reduction is split INSIDE the launch: workgroup = (output tile, M range); every split writes its fp32 tile into
//    its own slab and one small fold kernel sums the slabs (no atomics, deterministic, no [S,N,K] batched-GEMM temporary
//    in front of a library reduction);
//  * K tiles are 64 rows of dY and X, brought HBM -> LDS by global_load_lds_dwordx4 exactly as they lie in memory
//    (row-major, 512-byte rows: fully coalesced), and BOTH MFMA operands are read transposed out of those row-major images
//    with ds_read_b64_tr_b16 (two reads = the 8 consecutive-m values a lane needs for its column) -- no transposed copy of
//    dY or X is ever made, in HBM or in LDS;
//  * bank conflicts: a 16-lane group of a transposed read touches 4 rows x 32 B; with 512-byte (or 256-byte) rows the four
//    rows would sit on the same banks, so 32-byte chunk c of row r is stored at chunk c ^ ((r & 3) << 1) -- applied on the
//    per-lane SOURCE address of the DMA (which writes lane-linear) and on the read address.  The eight 32-byte pieces of a
//    half-wave then tile the 256-byte bank row exactly;
//  * 256x256 output tile, 8 waves as 2 x 4 (128 x 64 per wave = 4 x 2 MFMA 32x32x16 tiles), two 64-KB stages, and the same
//    two-wave-group schedule as the forward GEMM (gemm.hip): the groups run the K loop half a sub-tile apart so that on every
//    SIMD one wave feeds the matrix pipe while its partner reads fragments.  A 128x128 / 4-wave instance covers the narrow
//    weights (stage-0 / stage-1 linears, patch embedding, test configurations);
//  * the bias gradient: in the workgroups of the first K-column tile every wave adds ONE of its dY fragment tiles up with
//    v_dot2_f32_bf16 against (1, 1) -- four VALU issues per MFMA k-step behind a scalar branch -- so no separate pass ever streams dY
//    for its column sums;
//  * DropPath backward (timm 0.4.12, swin_transformer.py:390-391): the branch gradient is s_b dY with a per-sample factor
//    s_b in {0, 1/keep}.  Instead of a pass that writes s_b dY, the kernel skips the K tiles of dropped samples (row_mask)
//    and multiplies the result by 1/keep (scale): dW = (1/keep) sum over kept samples of dY^T X.
#include <stdlib.h>

#include "common.h"

namespace {

struct TnArgs {
  const bf16* A;      // dY [M, lda]  (columns = output rows n)
  const bf16* B;      // X  [M, ldb]  (columns = output columns k)
  float* out;         // slabs [S][N*K] (S > 1) or dW itself (S == 1)
  float* cs;          // slabs [S][N] or db itself; nullable
  int M, N, K, lda, ldb;
  int S, tiles_n, tiles_k, kt_per_split, nk_total;
  const float* row_mask;   // optional fp32 [M / rows_per_sample]: rows of samples whose entry is 0 do not contribute
  int rows_per_sample;     // (a multiple of the K-tile depth: a K tile never straddles two samples)
  float scale;             // the result (and the bias sums) are multiplied by this
};

__device__ __attribute__((aligned(16))) unsigned g_tn_zeros[128];   // 512 zero bytes: the source of dY rows past M

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 tr_frag(const char* p, int hi_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + hi_off));
  s16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
  return __builtin_bit_cast(bf16x8, o);
}

// TS: output tile edge (256: 8 waves, 128: 4 waves).  BKM: rows of dY / X per K tile.  NS: LDS stages.
// STAG (TS = 256, BKM = 64, NS = 2): two wave groups half a sub-tile apart, one workgroup per CU.
// ring (TS = 128, BKM = 32, NS = 4): the narrow weights are pure streaming problems (2 x 256 B per row of M against 128 x 128
// MACs), so what matters is bytes in flight: three K tiles per workgroup stay requested across the (raw) barrier behind a
// COUNTED vmcnt, 64 KB of LDS per workgroup so that two workgroups share a CU and cover each other's fragment reads.
template <int TS, int BKM, int NS, bool STAG>
__global__ __launch_bounds__(TS * 2) void gemm_tn_kernel(TnArgs a) {
  constexpr int NTH = TS * 2;
  constexpr int NW = NTH / 64;                  // 8 | 4 waves
  constexpr int WA = 2, WB = NW / 2;            // wave grid: 2 along n, 4 | 2 along k
  constexpr int WTA = TS / WA, WTB = TS / WB;   // per-wave tile 128 x 64 | 64 x 64
  constexpr int TA = WTA / 32, TB = WTB / 32;
  constexpr int RB = TS * 2;                    // bytes per LDS row (one operand)
  constexpr int CPRW = RB / 16;                 // 16-byte chunks per row
  constexpr int RPI = 64 / CPRW;                // rows per DMA instruction (1 KB)
  constexpr int NDI = BKM / RPI / NW;           // DMA instructions per wave, operand and K tile
  constexpr int OPB = BKM * RB;                 // bytes per operand per stage
  constexpr int STAGEB = 2 * OPB;
  constexpr int NSUB = BKM / 32;                // 32-row sub-tiles (two MFMA k-steps each) per K tile
  static_assert(NDI * RPI * NW == BKM && (!STAG || (NW == 8 && BKM == 64 && NS == 2)) && (BKM == 32 || BKM == 64), "geometry");
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGEB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: scalar branches, SGPR LDS bases
  const int wa = wave / WB, wb = wave % WB;
  const int tiles = a.tiles_n * a.tiles_k, nblk = tiles * a.S;
  int bid = blockIdx.x;
  {  // XCD-aware bijective remap: the tiles of one M range (they share dY / X row panels) run on one XCD's L2
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int s = bid / tiles, tile = bid - s * tiles;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * TS, k0 = tk * TS;
  const int kt0 = s * a.kt_per_split;
  const int nk = min(a.kt_per_split, a.nk_total - kt0);
  const int m0 = kt0 * BKM;

  // ---- DMA addressing: instruction p of this wave covers tile rows (p*NW + wave)*RPI .. +RPI-1, lane -> (row, 16-B chunk).
  // Source = wave-uniform row-panel base (SGPR pair, advanced per K tile) + per-lane 32-bit byte offset; issued from inline
  // asm (common.h lds_dma16) so that the compiler's waitcnt pass does not guard every ds_read with vmcnt(0).
  const int drow = lane / CPRW, dchunk = lane % CPRW;
  unsigned aoff[NDI], boff[NDI];
#pragma unroll
  for (int p = 0; p < NDI; ++p) {
    const int r = (p * NW + wave) * RPI + drow;
    const int gc = dchunk ^ ((r & 3) << 2);                       // source chunk of this LDS slot (32-B chunk ^ (r&3)<<1)
    aoff[p] = (unsigned)(r * a.lda + min(n0 + gc * 8, a.N - 8)) * 2u;
    boff[p] = (unsigned)(r * a.ldb + min(k0 + gc * 8, a.K - 8)) * 2u;
  }
  auto dma = [&](int kt) {
    char* st = smem + (kt % NS) * STAGEB;
    const int mt = m0 + kt * BKM;                                 // first row of this K tile (scalar)
    const bf16* abase = a.A + (size_t)mt * a.lda;
    const bf16* bbase = a.B + (size_t)mt * a.ldb;
    if (mt + BKM <= a.M) {
#pragma unroll
      for (int p = 0; p < NDI; ++p) {
        lds_dma16(abase, aoff[p], st + (p * NW + wave) * RPI * RB);
        lds_dma16(bbase, boff[p], st + OPB + (p * NW + wave) * RPI * RB);
      }
    } else {                                                      // the very last K tile of the problem: rows past M
#pragma unroll
      for (int p = 0; p < NDI; ++p) {
        const int r = (p * NW + wave) * RPI + drow;
        const bool in = mt + r < a.M;
        const char* ap = in ? reinterpret_cast<const char*>(abase) + aoff[p]
                            : reinterpret_cast<const char*>(g_tn_zeros) + dchunk * 16;          // dY row of zeros ...
        const char* bp = reinterpret_cast<const char*>(bbase) + (in ? boff[p] : boff[p] - (unsigned)(r * a.ldb) * 2u);  // ... times row mt (finite)
        lds_dma16_v(ap, st + (p * NW + wave) * RPI * RB);
        lds_dma16_v(bp, st + OPB + (p * NW + wave) * RPI * RB);
      }
    }
  };

  // ---- fragment addressing (transposed reads of the row-major images)
  const int fi = lane & 15, cg = (lane >> 4) & 1, kh = lane >> 5;
  const int rowl = kh * 8 + (fi >> 2), sx = (fi >> 2) << 1;
  int offa[TA], offb[TB];
#pragma unroll
  for (int t = 0; t < TA; ++t) offa[t] = rowl * RB + (((wa * (WTA / 16) + 2 * t + cg) ^ sx) << 5) + (fi & 3) * 8;
#pragma unroll
  for (int u = 0; u < TB; ++u) offb[u] = OPB + rowl * RB + (((wb * (WTB / 16) + 2 * u + cg) ^ sx) << 5) + (fi & 3) * 8;

  f32x16 acc[TA][TB];
#pragma unroll
  for (int t = 0; t < TA; ++t)
#pragma unroll
    for (int u = 0; u < TB; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  // Bias gradient = column sums of dY: in the workgroups of the first K-column tile wave (wa, wb) sums A tile t = wb of its four (two)
  // (TA == WB), i.e. 4 v_dot2 per MFMA k-step behind a SCALAR branch.  Round 3 had the wb = 0 waves sum all their tiles behind an
  // `if` the compiler turned into selects: every wave of every workgroup issued 16 v_dot2 + 8 moves / selects per k-step, which cost the
  // kernel 10-12 % of its rate (1095 TFLOP/s without the code, 980 with it; tools/tn_bias_probe.py).
  static_assert(TA == WB, "column sums: one A tile per wave");
  float csa = 0.f;
  const bool do_cs = a.cs != nullptr && tk == 0;                  // workgroup-uniform
  // K tile kt belongs to sample (m0 + kt*BKM) / rows_per_sample; its MFMAs are skipped when that sample was dropped.  The factors of the
  // next 64 samples are fetched with ONE vector load and kept as a ballot in an SGPR pair: a load in front of every K tile's branch cost
  // the masked form 18 % of its rate (and any vector load inside the loop makes hipcc wait with vmcnt(0), which also drains the LDS-DMA
  // requests of the next K tile -- the compiler does not count what inline asm issued).  The reload every 64 samples sits at the top of
  // an iteration, where no DMA is outstanding.
  const int tps = a.row_mask ? a.rows_per_sample / BKM : 1;       // K tiles per sample
  const int nsamp = a.row_mask ? (a.M + a.rows_per_sample - 1) / a.rows_per_sample : 1;
  int samp0 = a.row_mask ? (m0 / BKM) / tps : 0, srem = a.row_mask ? (m0 / BKM) % tps : 0, sbit = 0;
  unsigned long long keepbits = ~0ull;
  auto reload_mask = [&]() {
    if (a.row_mask) keepbits = __builtin_amdgcn_ballot_w64(a.row_mask[min(samp0 + lane, nsamp - 1)] != 0.f);
  };
  auto kept_then_advance = [&]() {                                 // is the K tile the counters point at kept?  then step to the next tile
    if (sbit == 64) { samp0 += 64; sbit = 0; reload_mask(); }
    const bool k = (keepbits >> sbit) & 1ull;
    if (++srem == tps) { srem = 0; ++sbit; }
    return k;
  };
  reload_mask();

  bf16x8 fa[2][TA], fb[2][TB];
  auto load_phase = [&](int kt, int sub) {
    const char* st = smem + (kt % NS) * STAGEB + sub * 32 * RB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < TA; ++t) fa[ks][t] = tr_frag(st + offa[t] + ks * 16 * RB, 4 * RB);
#pragma unroll
      for (int u = 0; u < TB; ++u) fb[ks][u] = tr_frag(st + offb[u] + ks * 16 * RB, 4 * RB);
    }
  };
  const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
  auto math_phase = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < TA; ++t) {
#pragma unroll
        for (int u = 0; u < TB; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][t], fb[ks][u], acc[t][u], 0, 0, 0);
        if (do_cs && wb == t) {
          asm volatile("" : "+v"(csa));                     // (keeps this a branch: no speculation into selects)
#pragma unroll
          for (int e = 0; e < 8; e += 2)
            csa = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{fa[ks][t][e], fa[ks][t][e + 1]}, ones, csa, false);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (STAG) {
    auto phase_barrier = [&](bool landed) {
      if (landed) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    dma(0);
    phase_barrier(true);
    if (wa == 1) phase_barrier(false);                   // group 1 runs one phase behind (see gemm.hip, wide kernel)
    for (int kt = 0; kt < nk; ++kt) {
      const bool inc = kept_then_advance();
      load_phase(kt, 0);
      if (kt + 1 < nk) dma(kt + 1);
      phase_barrier(false);
      if (inc) math_phase();
      phase_barrier(false);
      load_phase(kt, 1);
      phase_barrier(wa == 1);
      if (inc) math_phase();
      phase_barrier(wa == 0);
    }
    if (wa == 0) phase_barrier(false);
  } else {
    static_assert(STAG || (NS == 4 && NDI == 2), "counted vmcnt below: 3 tiles x 4 DMA instructions per wave in flight");
    for (int pre = 0; pre < NS - 1 && pre < nk; ++pre) dma(pre);
    for (int kt = 0; kt < nk; ++kt) {
      const bool inc = kept_then_advance();
      const int newer = min(nk - 1 - kt, NS - 2);          // K tiles requested after tile kt (2 NDI instructions per wave each)
      if (newer >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                       // tile kt resident for every wave; tile kt-1 fully consumed
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kt + NS - 1 < nk) dma(kt + NS - 1);             // into the stage tile kt-1 just vacated
      if (inc) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
          load_phase(kt, sub);
          math_phase();
        }
      }
    }
  }

  // ---- epilogue: fp32 tile -> this split's slab.  Per accumulator register a half-wave writes 128 contiguous bytes.
  float* outp = a.out + (size_t)s * a.N * a.K;
  const int kcol_l = lane & 31;
#pragma unroll
  for (int t = 0; t < TA; ++t)
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int kk = k0 + wb * WTB + u * 32 + kcol_l;
      const int nb = n0 + wa * WTA + t * 32 + 4 * kh;
      if (kk < a.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nb + (r & 3) + 8 * (r >> 2);
          if (n < a.N) outp[(size_t)n * a.K + kk] = acc[t][u][r] * a.scale;
        }
      }
    }
  if (do_cs) {
    float* csp = a.cs + (size_t)s * a.N;
    const float v = (csa + __shfl_xor(csa, 32)) * a.scale;
    const int n = n0 + wa * WTA + wb * 32 + (lane & 31);
    if (lane < 32 && n < a.N) csp[n] = v;
  }
}

// out[i] = sum_s slab[s][i] over the N*K tile elements (float4) and, optionally, the N column sums
// out = sum over the S slabs, element by element.  A workgroup owns CW = 256 / QL float4 columns of the result and cuts the slabs over QL thread
// rows: thread (q, c) adds slabs q, q + QL, ... of column c (four requested before the first add), the QL partial sums meet in LDS and are added
// in the order 0 .. QL - 1 -- a fixed order, the same in the one-result and the many-results launch.  (Round 6: one thread per float4 walking all S
// slabs ran at 2.2-3.9 TB/s out of the Infinity Cache, 0.7 TB/s at the stage-0 shapes with 170 slabs and 48 workgroups; 187 folds a step.)
// row_map (nullable): element (n, k) of the sum is written at row row_map[n] (a float4 never straddles rows: K % 8 == 0), bias entry n at row_map[n]
__host__ __device__ inline int tn_fold_ql(int S) { return S >= 64 ? 16 : S >= 16 ? 8 : 4; }
__host__ __device__ inline long tn_fold_nblocks(int S, long nk4, int N, bool has_cs) {
  const int cw = 256 / tn_fold_ql(S);
  return (nk4 + cw - 1) / cw + (has_cs ? (N + cw - 1) / cw : 0);
}
template <int QL>
__device__ __forceinline__ void tn_fold_block(const float* ws, const float* ws_cs, float* out, float* cs, int S, long nk4, int N, long blk,
                                              const int* row_map) {
  constexpr int CW = 256 / QL;
  __shared__ float4 part[QL][CW];
  const int c = threadIdx.x % CW, q = threadIdx.x / CW;
  const long nmain = (nk4 + CW - 1) / CW;
  if (blk < nmain) {
    const long i = blk * CW + c;
    float4 t = {0.f, 0.f, 0.f, 0.f};
    if (i < nk4) {
      const float4* p = reinterpret_cast<const float4*>(ws) + i;
      int s = q;
      for (; s + 3 * QL < S; s += 4 * QL) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[(long)(s + u * QL) * nk4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
      }
      for (; s < S; s += QL) {
        const float4 v = p[(long)s * nk4];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
    }
    part[q][c] = t;
    __syncthreads();
    if (q == 0 && i < nk4) {
      float4 r = part[0][c];
#pragma unroll
      for (int k = 1; k < QL; ++k) { const float4 v = part[k][c]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
      long o = i;
      if (row_map) {
        const long k4 = nk4 / N;                           // float4s per row
        const long n = i / k4;
        o = (long)row_map[n] * k4 + (i - n * k4);
      }
      reinterpret_cast<float4*>(out)[o] = r;
    }
  } else if (cs != nullptr) {                              // the bias sums: CW entries per workgroup, the same cut over the slabs
    const long n = (blk - nmain) * CW + c;
    float t = 0.f;
    if (n < N)
      for (int s = q; s < S; s += QL) t += ws_cs[(long)s * N + n];
    float* pf = reinterpret_cast<float*>(&part[0][0]);
    pf[q * CW + c] = t;
    __syncthreads();
    if (q == 0 && n < N) {
      float r = pf[c];
#pragma unroll
      for (int k = 1; k < QL; ++k) r += pf[k * CW + c];
      cs[row_map ? row_map[n] : n] = r;
    }
  }
}

__global__ __launch_bounds__(256) void tn_fold_kernel(const float* ws, const float* ws_cs, float* out, float* cs, int S, long nk4, int N, const int* row_map) {
  const int ql = tn_fold_ql(S);
  if (ql == 16) tn_fold_block<16>(ws, ws_cs, out, cs, S, nk4, N, blockIdx.x, row_map);
  else if (ql == 8) tn_fold_block<8>(ws, ws_cs, out, cs, S, nk4, N, blockIdx.x, row_map);
  else tn_fold_block<4>(ws, ws_cs, out, cs, S, nk4, N, blockIdx.x, row_map);
}

// The folds of MANY weight gradients in one launch (ops.py can defer them to the end of the backward pass; measured slower than the immediate
// folds, profiles/r06_summary.md section 3).  Same per-element summation order as tn_fold_kernel: same bits.
struct FoldDesc { const float* ws; float* out; float* cs; int S, N, nk4, block0; };   // 40 bytes (cs: NULL = no bias sums)

__global__ __launch_bounds__(256) void tn_fold_multi_kernel(const FoldDesc* __restrict__ table, int ndesc) {
  int lo = 0, hi = ndesc - 1;                              // descriptor whose block range holds blockIdx.x (block0 ascending)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FoldDesc d = table[lo];
  const long blk = (long)blockIdx.x - d.block0;
  const int ql = tn_fold_ql(d.S);
  if (ql == 16) tn_fold_block<16>(d.ws, d.ws + (size_t)d.S * d.nk4 * 4, d.out, d.cs, d.S, d.nk4, d.N, blk, nullptr);
  else if (ql == 8) tn_fold_block<8>(d.ws, d.ws + (size_t)d.S * d.nk4 * 4, d.out, d.cs, d.S, d.nk4, d.N, blk, nullptr);
  else tn_fold_block<4>(d.ws, d.ws + (size_t)d.S * d.nk4 * 4, d.out, d.cs, d.S, d.nk4, d.N, blk, nullptr);
}

struct TnPlan { int ts, bkm, tiles_n, tiles_k, S, kt_per_split, nk_total; };

TnPlan tn_plan(int M, int N, int K) {
  static const int force_ts = getenv("FIBER_TN_TILE") ? atoi(getenv("FIBER_TN_TILE")) : 0;
  static const int rounds = getenv("FIBER_TN_ROUNDS") ? atoi(getenv("FIBER_TN_ROUNDS")) : 1;
  TnPlan p;
  p.ts = (N >= 192 && K >= 192) ? 256 : 128;
  if (force_ts == 128 || force_ts == 256) p.ts = force_ts;
  p.bkm = p.ts == 256 ? 64 : 32;
  p.tiles_n = cdiv(N, p.ts);
  p.tiles_k = cdiv(K, p.ts);
  p.nk_total = cdiv(M, p.bkm);
  const int tiles = p.tiles_n * p.tiles_k;
  // Every workgroup of a launch does the same amount of work, so the grid should be just UNDER a whole number of rounds of
  // resident workgroups (1 per CU for the 256 tile, 2 for the 128 tile): 516 workgroups on 256 CUs would take three rounds.
  const int resident = 256 * (p.ts == 256 ? 1 : 2) * (rounds > 0 ? rounds : 1);
  int S = resident / tiles;
  const int min_kt = p.ts == 256 ? 8 : 16;                 // a split shorter than this is all prologue + epilogue
  if (S > p.nk_total / min_kt) S = p.nk_total / min_kt;
  if (S < 1) S = 1;
  p.kt_per_split = cdiv(p.nk_total, S);
  p.S = cdiv(p.nk_total, p.kt_per_split);                  // every split owns >= 1 K tile
  return p;
}

}  // namespace

// C ABI ---------------------------------------------------------------------------------------------------------
// Number of M splits the kernel will use for this problem; workspace = S > 1 ? S * (N*K + N) floats : 0.
extern "C" int fiber_gemm_tn_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return tn_plan(M, N, K).S;
}

// dW[N,K] (fp32, contiguous) = scale * sum over rows m of kept samples of dY[m,:]^T X[m,:];  dbias (nullable, fp32[N]) = the same
// sum of dY rows.  row_mask (nullable): fp32 [M / rows_per_sample], sample b is kept iff row_mask[b] != 0 (the DropPath factors
// of the branch: pass 1/keep as `scale`); rows_per_sample must be a multiple of 64 then.  Without a mask every row counts.
// N % 8 == 0, K % 8 == 0, lddy % 8 == 0, ldx % 8 == 0, 16-byte aligned bases.
static int tn_launch(const void* dY, const void* X, float* dW, float* dbias, float* workspace, int M, int N, int K, int lddy, int ldx,
                     const float* row_mask, int rows_per_sample, float scale, bool fold, const int* row_map, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return FIBER_OK;
  if ((N & 7) || (K & 7) || (lddy & 7) || (ldx & 7)) return FIBER_EINVAL;
  if (row_mask && (rows_per_sample <= 0 || (rows_per_sample & 63))) return FIBER_EINVAL;
  const TnPlan p = tn_plan(M, N, K);
  // The row permutation is applied by the fold (one index per float4 of the result), never by the GEMM's epilogue: a `row_map ? row_map[n] : n`
  // in front of each of its 128 stores per lane cost EVERY weight gradient of the step 2.8 % (profiles/r06_kernel_diff_vs_r05.log).  A
  // permuted result therefore always goes through the workspace, as ONE slab when the reduction is not split.
  const bool via_ws = p.S > 1 || (row_map != nullptr && fold);
  if (via_ws && !workspace) return FIBER_EINVAL;
  if (row_map && !fold) return FIBER_EINVAL;
  TnArgs a;
  a.A = (const bf16*)dY; a.B = (const bf16*)X;
  a.out = via_ws ? workspace : dW;
  a.cs = dbias ? (via_ws ? workspace + (size_t)p.S * N * K : dbias) : nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lddy; a.ldb = ldx;
  a.S = p.S; a.tiles_n = p.tiles_n; a.tiles_k = p.tiles_k; a.kt_per_split = p.kt_per_split; a.nk_total = p.nk_total;
  a.row_mask = row_mask; a.rows_per_sample = rows_per_sample; a.scale = scale;
  const unsigned grid = (unsigned)(p.tiles_n * p.tiles_k * p.S);
  if (p.ts == 256) hipLaunchKernelGGL((gemm_tn_kernel<256, 64, 2, true>), dim3(grid), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((gemm_tn_kernel<128, 32, 4, false>), dim3(grid), dim3(256), 0, stream, a);
  FIBER_CHECK_LAUNCH();
  if (via_ws && fold) {
    const long nk4 = (long)N * K / 4;
    hipLaunchKernelGGL(tn_fold_kernel, dim3((unsigned)tn_fold_nblocks(p.S, nk4, N, dbias != nullptr)), dim3(256), 0, stream, workspace,
                       workspace + (size_t)p.S * N * K, dW, dbias, p.S, nk4, N, row_map);
    FIBER_CHECK_LAUNCH();
  }
  return FIBER_OK;
}

extern "C" int fiber_gemm_tn_bf16(const void* dY, const void* X, float* dW, float* dbias, float* workspace, int M, int N, int K,
                                  int lddy, int ldx, const float* row_mask, int rows_per_sample, float scale,
                                  hipStream_t stream) {
  return tn_launch(dY, X, dW, dbias, workspace, M, N, K, lddy, ldx, row_mask, rows_per_sample, scale, true, nullptr, stream);
}

// The same without the fold when the M reduction is split (fiber_gemm_tn_splits > 1): the slabs stay in `workspace` (S*(N*K) floats of
// weight slabs followed by S*N floats of bias slabs) for fiber_tn_fold_multi; dW / dbias are not written.  With one split it is the call above.
extern "C" int fiber_gemm_tn_slabs_bf16(const void* dY, const void* X, float* dW, float* dbias, float* workspace, int M, int N, int K,
                                        int lddy, int ldx, const float* row_mask, int rows_per_sample, float scale,
                                        hipStream_t stream) {
  return tn_launch(dY, X, dW, dbias, workspace, M, N, K, lddy, ldx, row_mask, rows_per_sample, scale, false, nullptr, stream);
}

// table: device array of ndesc 40-byte records {ws, out, cs (8-byte pointers; cs NULL = no bias sums), int32 S, N, nk4 = N*K/4, block0},
// block0 ascending from 0, blocks of a record = fiber_tn_fold_blocks(S, N, K, cs != NULL); nblocks = their total.
extern "C" int fiber_tn_fold_blocks(int S, int N, int K, int has_cs) { return (int)tn_fold_nblocks(S, (long)N * K / 4, N, has_cs != 0); }
extern "C" int fiber_tn_fold_multi(const void* table, int ndesc, int nblocks, hipStream_t stream) {
  if (ndesc <= 0 || nblocks <= 0) return FIBER_OK;
  hipLaunchKernelGGL(tn_fold_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, (const FoldDesc*)table, ndesc);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// fiber_gemm_tn_bf16 with its output rows permuted on the way out: row n of dW (and entry n of dbias) is written at row_map[n] (int32 [N], a
// permutation; the head-major qkv projection of ops.py trains a weight whose rows are stored in another order than the kernel computes them:
// three index kernels per Swin block and step otherwise).
extern "C" int fiber_gemm_tn_rowmap_bf16(const void* dY, const void* X, float* dW, float* dbias, float* workspace, int M, int N, int K,
                                         int lddy, int ldx, const float* row_mask, int rows_per_sample, float scale, const int* row_map,
                                         hipStream_t stream) {
  return tn_launch(dY, X, dW, dbias, workspace, M, N, K, lddy, ldx, row_mask, rows_per_sample, scale, true, row_map, stream);
}

