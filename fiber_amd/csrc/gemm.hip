// bf16 MFMA GEMM with fused epilogues for the FIBER fused-backbone path (gfx950 / CDNA4).
//
//   Y[M,N] = epilogue( X[M,K] . W[N,K]^T )         X, W, Y row-major bf16, fp32 accumulate
//
// Replaces the separate ATen addmm + bias + GELU + residual-add kernels behind every nn.Linear on the
// reference hot path (swin_transformer.py:197,221,233,238,257 qkv/proj/i2t projections; timm Mlp fc1/fc2 at
// swin_transformer.py:325; roberta.py:231-241,337,398,415 query/key/value/dense layers; PatchMerging.reduction
// swin_transformer.py:431; fiber_module.py:349-350 cross-modal transforms).
//
// Design (CDNA4): 256-thread workgroups (4 waves, 2x2), block tile BMxBNx64, per-wave (BM/2)x(BN/2) built from
// v_mfma_f32_32x32x16_bf16 tiles.  Operands are staged global -> VGPR (16 B/lane, coalesced 128-B rows) -> LDS with a
// 16-byte-chunk XOR swizzle (chunk ^= (row>>1)&7) that makes both the ds_write_b128 staging stores and the
// ds_read_b128 fragment loads bank-conflict free (LDS bank row = 256 B = two 128-B tile rows).  LDS is double
// buffered: loads for tile t+1 are issued before the MFMAs of tile t and written after them, one barrier per K tile.
// The MFMA is issued with swapped operands (D^T = W.X^T) so each lane owns 4 consecutive output columns and the
// epilogue (bias, exact-erf GELU, residual, optional pre-activation copy) reads/writes 8-byte vectors.
// Workgroup ids are remapped so that consecutive tiles of one X row-panel land on the same XCD (shared L2).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BK = 64;

struct GemmArgs {
  const bf16* X; const bf16* W; const float* bias; const bf16* R; bf16* Y; bf16* Ypre;
  const float* rowscale;   // optional per-sample scale (timm DropPath): row m uses rowscale[m / rows_per_sample]
  const bf16* aux;         // act == 2: pre-activation H [M, ldaux]; the output is acc * gelu'(H)  (fused GELU backward)
  float* colpart;          // optional [tilesM, N] fp32: per-row-tile column sums of the stored output (bias gradient)
  int M, N, K, ldx, ldw, ldy, ldr, act, rows_per_sample, ldaux;
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row * 8 + (chunk ^ ((row >> 1) & 7))) * 8; }  // element offset

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
  constexpr int WTM = BM / 2, WTN = BN / 2;      // per-wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;    // 32x32 MFMA tiles per wave
  constexpr int PA = BM / 32, PB = BN / 32;      // staging passes (32 rows x 8 chunks per pass)
  __shared__ __attribute__((aligned(16))) bf16 As[2][BM * BK];
  __shared__ __attribute__((aligned(16))) bf16 Bs[2][BN * BK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tilesN = (a.N + BN - 1) / BN, tilesM = (a.M + BM - 1) / BM;
  const int nblk = tilesM * tilesN;
  // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a contiguous run of tiles.
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm0 = (bid / tilesN) * BM, tn0 = (bid % tilesN) * BN;

  const int srow = tid >> 3, schunk = tid & 7;
  const bf16* xrow[PA];
  const bf16* wrow[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    int r = tm0 + srow + p * 32;
    r = r < a.M ? r : a.M - 1;
    xrow[p] = a.X + (size_t)r * a.ldx + schunk * 8;
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    int r = tn0 + srow + p * 32;
    r = r < a.N ? r : a.N - 1;
    wrow[p] = a.W + (size_t)r * a.ldw + schunk * 8;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[PA], rb[PB];
  const int nk = (a.K + BK - 1) / BK;
  auto gload = [&](int kt) {
    const int k = kt * BK + schunk * 8;
    const bool ok = k < a.K;   // K % 8 == 0 is required, so a chunk is entirely in or out
#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = ok ? *reinterpret_cast<const uint4*>(xrow[p] + kt * BK) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = ok ? *reinterpret_cast<const uint4*>(wrow[p] + kt * BK) : make_uint4(0, 0, 0, 0);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PA; ++p) *reinterpret_cast<uint4*>(&As[buf][swz(srow + p * 32, schunk)]) = ra[p];
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<uint4*>(&Bs[buf][swz(srow + p * 32, schunk)]) = rb[p];
  };

  gload(0);
  lstore(0);
  __syncthreads();

  const int frow = lane & 31, fk = lane >> 5;   // fragment row within a 32-row tile, k-chunk selector (0/1)
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(&As[cur][swz(wm * WTM + i * 32 + frow, ks * 2 + fk)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(&Bs[cur][swz(wn * WTN + j * 32 + frow, ks * 2 + fk)]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D^T[n][m]
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds, for output row m = ..+(lane&31), columns n = ..+8*q+4*(lane>>5)+{0..3}, q = 0..3
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = tm0 + wm * WTM + i * 32 + (lane & 31);
    if (m >= a.M) continue;
    const float rsc = a.rowscale ? a.rowscale[m / a.rows_per_sample] : 1.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = tn0 + wn * WTN + j * 32 + q * 8 + (lane >> 5) * 4;
        if (n >= a.N) continue;   // N % 4 == 0 required
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
        if (a.bias) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (a.act == 1) {
          if (a.Ypre) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
            *reinterpret_cast<bf16x4*>(a.Ypre + (size_t)m * a.ldy + n) = o;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (a.rowscale) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= rsc;
        }
        if (a.R) {
          const bf16x4 r = *reinterpret_cast<const bf16x4*>(a.R + (size_t)m * a.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(r[e]);
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *reinterpret_cast<bf16x4*>(a.Y + (size_t)m * a.ldy + n) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v2: same tile / MFMA structure, but (1) operands go global -> LDS directly (global_load_lds_dwordx4: no VGPR round
// trip, no ds_write issue cost -- rocprof + the LDS budget showed v1 LDS-bound: 32 KB of ds_write_b128 per K tile cost
// ~415 LDS cycles next to 512 MFMA cycles); the XOR swizzle moves to the per-lane SOURCE address because the DMA writes
// lane-linear (wave base + lane*16 B); (2) the epilogue is staged through LDS so that every global store / residual
// load is a full 16-byte, row-contiguous access (v1's per-lane 8-byte stores touched 32 rows per instruction and held
// the HBM-bound stage-0/1 GEMMs at ~50 % of the bandwidth roofline).  Requires K % 64 == 0 and N % 8 == 0.
template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_glds_kernel(GemmArgs a) {
  constexpr int NT = 64 * WM * WN;                     // threads per workgroup
  constexpr int WTM = BM / WM, WTN = BN / WN;          // per-wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPD = NT / 8;                          // tile rows covered by one DMA pass (8 lanes x 16 B per 128-B row)
  constexpr int PA = BM / RPD, PB = BN / RPD;
  constexpr int CLD = BN + 8;                          // epilogue tile row stride (elements)
  static_assert(BM * CLD <= NS * (BM + BN) * BK, "epilogue tile must fit in the operand buffers");
  __shared__ __attribute__((aligned(16))) bf16 smem[NS * (BM + BN) * BK];
  bf16* As = smem;                                     // [NS][BM*BK]
  bf16* Bs = smem + NS * BM * BK;                      // [NS][BN*BK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tilesN = (a.N + BN - 1) / BN, tilesM = (a.M + BM - 1) / BM;
  const int nblk = tilesM * tilesN;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm0 = (bid / tilesN) * BM, tn0 = (bid % tilesN) * BN;

  // DMA slot of this lane in pass p: tile row = p*RPD + wave*8 + (lane>>3), physical chunk = lane&7
  const int srow = wave * 8 + (lane >> 3), spc = lane & 7;
  const bf16* xsrc[PA];
  const bf16* wsrc[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = srow + p * RPD;
    int r = tm0 + row;
    r = r < a.M ? r : a.M - 1;
    xsrc[p] = a.X + (size_t)r * a.ldx + ((spc ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int row = srow + p * RPD;
    int r = tn0 + row;
    r = r < a.N ? r : a.N - 1;
    wsrc[p] = a.W + (size_t)r * a.ldw + ((spc ^ ((row >> 1) & 7)) << 3);
  }
  // one DMA instruction (1 KiB per wave): index d in [0, PA+PB) selects the operand and pass
  auto dma1 = [&](int d, int kt, int buf) {
    if (d < PA)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[d < PA ? d : 0] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(As + buf * BM * BK + (d * RPD + wave * 8) * BK), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[d >= PA ? d - PA : 0] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(Bs + buf * BN * BK + ((d - PA) * RPD + wave * 8) * BK), 16, 0, 0);
  };
  auto dma = [&](int kt, int buf) {
#pragma unroll
    for (int d = 0; d < PA + PB; ++d) dma1(d, kt, buf);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = a.K / BK;
  const int frow = lane & 31, fk = lane >> 5;
  const bool dbg_nodma = a.act & 0x100;                // ablation switch (tools/gemm_probe.py)
  constexpr int KS = BK / 16;
  constexpr int DPK = (PA + PB + KS - 2) / (KS - 1);   // DMA instructions issued per k-step (spread over the first KS-1 steps)
  // K-tile body, software pipelined inside the wave: fragments of k-step ks+1 are requested and a slice of the NEXT
  // tiles' DMA is issued BEFORE the MFMAs of k-step ks, so LDS latency and DMA issue cost sit under matrix-pipe time
  // (in the lockstep version every wave issued all DMA right after the barrier and the three phases simply added up).
  auto compute = [&](int buf, bool prefetch, int kt_next, int nbuf) {
    const bf16* Ac = As + buf * BM * BK;
    const bf16* Bc = Bs + buf * BN * BK;
    bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * WTM + i * 32 + frow, fk));
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * WTN + j * 32 + frow, fk));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      if (ks + 1 < KS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[n][i] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * WTM + i * 32 + frow, (ks + 1) * 2 + fk));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[n][j] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * WTN + j * 32 + frow, (ks + 1) * 2 + fk));
        if (prefetch) {
#pragma unroll
          for (int d = ks * DPK; d < (ks + 1) * DPK && d < PA + PB; ++d) dma1(d, kt_next, nbuf);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[c][j], fa[c][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (NS == 2) {
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      compute(cur, kt + 1 < nk && !dbg_nodma, kt + 1, cur ^ 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // NS-deep ring: NS-1 K tiles stay in flight across the (raw) barrier; the wait is COUNTED (PA+PB DMA instructions
    // per wave per tile), never a drain, so L2/HBM latency of tile kt+NS-1 hides under NS-1 tiles of MFMAs.
    static_assert(NS == 3 && PA + PB == 6, "counted vmcnt below is written for 3 stages x 6 DMA instructions");
    dma(0, 0);
    if (nk > 1) dma(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile kt landed (tile kt+1 may still fly)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // every wave's share of tile kt is in LDS; tile kt-1 fully consumed
      asm volatile("" ::: "memory");
      const int nb = buf == 0 ? 2 : buf - 1;            // (kt+2) % 3: the buffer read during iteration kt-1
      compute(buf, kt + 2 < nk && !dbg_nodma, kt + 2, nb);
      buf = buf == 2 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: registers -> LDS tile (bf16) -> coalesced 16-byte rows ------------------------------------------
  bf16* Cs = smem;
  constexpr int CPR = BN / 8;                           // 16-byte chunks per tile row
  constexpr int RPP = NT / CPR;                         // tile rows per store pass
  const int erow = tid / CPR, echunk = tid % CPR;
  auto stage_tile = [&](bool pre) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ml = wm * WTM + i * 32 + (lane & 31);
      const int m = tm0 + ml;
      const float rsc = (!pre && a.rowscale && m < a.M) ? a.rowscale[m / a.rows_per_sample] : 1.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = wn * WTN + j * 32 + q * 8 + (lane >> 5) * 4;
          const int n = tn0 + nl;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
          if (a.bias && n < a.N) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          }
          if (!pre) {
            if ((a.act & 0xff) == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= rsc;
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
          *reinterpret_cast<bf16x4*>(Cs + ml * CLD + nl) = o;
        }
    }
  };
  // Single staging pass: the tile staged in LDS is the PRE-activation when act=GELU; the coalesced store pass writes it
  // to Ypre (if requested), applies GELU / DropPath scale / residual on 8-wide vectors and writes Y.  (GELU is evaluated
  // on the bf16-rounded pre-activation, i.e. exactly the value the backward pass will differentiate at.)
  const bool gelu = (a.act & 0xff) == 1, gelu_grad = (a.act & 0xff) == 2;
  stage_tile(gelu);
  __syncthreads();
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r0 = 0; r0 < BM; r0 += RPP) {
    const int ml = r0 + erow, m = tm0 + ml, n = tn0 + echunk * 8;
    if (m < a.M && n < a.N) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(Cs + ml * CLD + echunk * 8);
      if (gelu) {
        if (a.Ypre) *reinterpret_cast<bf16x8*>(a.Ypre + (size_t)m * a.ldy + n) = v;
        const float rsc = a.rowscale ? a.rowscale[m / a.rows_per_sample] : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(gelu_erf(bf2f(v[e])) * rsc);
      } else if (gelu_grad) {
        const bf16x8 h = *reinterpret_cast<const bf16x8*>(a.aux + (size_t)m * a.ldaux + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) * gelu_erf_grad(bf2f(h[e])));
      }
      if (a.R) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(a.R + (size_t)m * a.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(r[e]));
      }
      *reinterpret_cast<bf16x8*>(a.Y + (size_t)m * a.ldy + n) = v;
#pragma unroll
      for (int e = 0; e < 8; ++e) csum[e] += bf2f(v[e]);
    }
  }
  if (a.colpart) {                                      // column sums of this tile: RPP row-lanes -> one row, via LDS
    float* red = reinterpret_cast<float*>(Cs + BM * CLD);          // free LDS behind the staged tile
    static_assert((size_t)BM * CLD * 2 + (size_t)RPP * BN * 4 <= sizeof(smem), "no LDS room for the column-sum reduction");
#pragma unroll
    for (int e = 0; e < 8; ++e) red[erow * BN + echunk * 8 + e] = csum[e];
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      float t = 0.f;
      for (int r = 0; r < RPP; ++r) t += red[r * BN + c];
      if (tn0 + c < a.N) a.colpart[(size_t)(tm0 / BM) * a.N + tn0 + c] = t;
    }
  }
}

}  // namespace

// C ABI ---------------------------------------------------------------------------------------------------------
// Row-tile height the dispatcher will use for an [M,N,K] problem (rows of `colpart` = ceil(M / tile)).
extern "C" int fiber_gemm_row_tile(int M, int N, int K) {
  const long big = (long)cdiv(M, 128) * cdiv(N, 128), huge = (long)cdiv(M, 256) * cdiv(N, 128);
  static const int force = getenv("FIBER_GEMM_TILE") ? atoi(getenv("FIBER_GEMM_TILE")) : 0;
  if ((huge >= 400 && K >= 256 && force == 0) || force == 256) return 256;
  if (big >= 192 || force == 128) return 128;
  return 64;
}

// Y = rowscale * act(X.W^T + bias) + residual.  bias: fp32[N] or NULL; residual: bf16[M,ldr] or NULL; act: 0 none, 1 exact GELU;
// rowscale: fp32[M / rows_per_sample] or NULL (per-sample DropPath factor on the branch, swin_transformer.py:390-391)
// (Ypre, if non-NULL with act=1, receives the pre-activation for the backward pass).  K % 8 == 0, N % 4 == 0,
// all leading dimensions multiples of 8 elements (16-byte rows).
extern "C" int fiber_gemm_nt_bf16(const void* X, const void* W, const float* bias, const void* residual, void* Y,
                                  void* Ypre, const float* rowscale, int rows_per_sample, const void* aux, int ldaux,
                                  float* colpart, int M, int N, int K, int ldx, int ldw, int ldy, int ldr, int act,
                                  hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return FIBER_OK;
  if ((K & 7) || (N & 3) || (ldx & 7) || (ldw & 7) || (ldy & 3) || (residual && (ldr & 3))) return FIBER_EINVAL;
  if (rowscale && rows_per_sample <= 0) return FIBER_EINVAL;
  if ((act & 0xff) == 2 && (!aux || (ldaux & 7))) return FIBER_EINVAL;
  GemmArgs a{(const bf16*)X, (const bf16*)W, bias, (const bf16*)residual, (bf16*)Y, (bf16*)Ypre, rowscale,
             (const bf16*)aux, colpart, M, N, K, ldx, ldw, ldy, ldr, act, rows_per_sample, ldaux};
  const long big = (long)cdiv(M, 128) * cdiv(N, 128);
  const long huge = (long)cdiv(M, 256) * cdiv(N, 128);
  if (((act & 0xff) == 2 || colpart) && !((K % 64 == 0) && (N % 8 == 0) && (ldy % 8 == 0))) return FIBER_EINVAL;   // glds kernels only
  const bool v2 = (K % 64 == 0) && (N % 8 == 0) && (ldy % 8 == 0) && (!residual || ldr % 8 == 0) && !getenv("FIBER_GEMM_V1");
  // Tile choice: the 128x128 tile is L2->LDS bandwidth bound (64 flop per staged byte, ~10 TB/s fabric => ~650 TFLOP/s,
  // measured with tools/gemm_probe.py); 256x128 (8 waves, 85 flop/B) lifts that ceiling when there are enough tiles.
  static const int force = getenv("FIBER_GEMM_TILE") ? atoi(getenv("FIBER_GEMM_TILE")) : 0;
  if (v2 && ((huge >= 400 && K >= 256 && force == 0) || force == 256)) {
    hipLaunchKernelGGL((gemm_nt_glds_kernel<256, 128, 4, 2, 3>), dim3((unsigned)huge), dim3(512), 0, stream, a);
  } else if (big >= 192 || force == 128) {
    if (v2) hipLaunchKernelGGL((gemm_nt_glds_kernel<128, 128, 2, 2, 2>), dim3((unsigned)big), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<128, 128>), dim3((unsigned)big), dim3(256), 0, stream, a);
  } else {
    const long small = (long)cdiv(M, 64) * cdiv(N, 64);
    if (v2) hipLaunchKernelGGL((gemm_nt_glds_kernel<64, 64, 2, 2, 2>), dim3((unsigned)small), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<64, 64>), dim3((unsigned)small), dim3(256), 0, stream, a);
  }
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
