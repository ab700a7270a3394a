// bf16 MFMA GEMM with fused epilogues for the FIBER fused-backbone path (gfx950 / CDNA4).
//
//   Y[M,N] = epilogue( X[M,K] . W[N,K]^T )         X, W, Y row-major bf16, fp32 accumulate
//
// Replaces the separate ATen addmm + bias + GELU + residual-add kernels behind every nn.Linear on the
// reference hot path (swin_transformer.py:197,221,233,238,257 qkv/proj/i2t projections; timm Mlp fc1/fc2 at
// swin_transformer.py:325; roberta.py:231-241,337,398,415 query/key/value/dense layers; PatchMerging.reduction
// swin_transformer.py:431; fiber_module.py:349-350 cross-modal transforms).
//
// Kernel family (CDNA4, all NT = both operands K-contiguous, v_mfma_f32_32x32x16_bf16 issued with swapped operands so each
// lane owns 4 consecutive output columns):
//   gemm_nt_wide_persist2_kernel 256x256 tile, 8 waves, two 64-KB LDS-DMA stages, two wave groups half a sub-tile apart,
//                                persistent over output tiles (>= 512 tiles, N % 256 == 0), wave-private epilogue with no
//                                workgroup barrier, bias as accumulator seed                          <- the large shapes
//   gemm_nt_wide_persist_kernel  its predecessor (workgroup-wide staged epilogue): still serves the fused gelu' * aux + column
//                                sums backward and the residual-without-DropPath forms, which spill in the v4 structure
//   gemm_nt_wide_kernel          the same K loop, one tile per workgroup (200..511 tiles)
//   gemm_nt_glds_kernel          256x128 (3-stage ring, counted vmcnt) / 128x128 / 64x64 tiles for N not a multiple of 256 or
//                                few tiles (stage-0 qkv / proj, text layers at small batch, edge configs)
//   gemm_nt_kernel               register-staged fallback for K % 64 != 0 (patch embedding K = 48 -> 64 padded is DMA-able;
//                                this covers odd test shapes)
// Operands go global -> LDS by global_load_lds_dwordx4 with a 16-byte-chunk XOR swizzle (chunk ^= (row>>1)&7) applied on the
// per-lane SOURCE address (the DMA writes lane-linear) and on the ds_read_b128 side: conflict-free for both (LDS bank row =
// 256 B = two 128-B tile rows).  Every epilogue (bias, exact-erf GELU + pre-activation copy, DropPath row scale, residual,
// gelu' * aux + column sums) is a compile-time variant of tile_epilogue<>, staged through LDS so that global stores and
// residual loads are 16-byte row-contiguous.  Workgroup ids are remapped so that consecutive tiles of one X row panel land on
// the same XCD (shared L2).  Measurements behind each choice: profiles/r01_summary.md.
#include <stdlib.h>

#include "gemm_epilogue.h"

namespace {

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
        if ((a.act & 0xff) == 1) {
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
        if (a.act & 0x800) {                              // fp32 residual stream (EPI 3 of gemm_epilogue.h): branch rounded once
          const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.R) + (size_t)m * a.ldr + n);
          const float4 o = {bf2f(f2bf(v[0])) + r.x, bf2f(f2bf(v[1])) + r.y, bf2f(f2bf(v[2])) + r.z, bf2f(f2bf(v[3])) + r.w};
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.Ypre) + (size_t)m * a.ldy + n) = o;
          if (a.Y) {
            bf16x4 ob = {f2bf(o.x), f2bf(o.y), f2bf(o.z), f2bf(o.w)};
            *reinterpret_cast<bf16x4*>(a.Y + (size_t)m * a.ldy + n) = ob;
          }
          continue;
        }
        if (a.R) {
          const bf16x4 r = *reinterpret_cast<const bf16x4*>(a.R + (size_t)m * a.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(r[e]);
        }
        if (a.act & 0x100) {                              // fp32 output (narrow heads whose consumers need full precision)
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.Y) + (size_t)m * a.ldy + n) = float4{v[0], v[1], v[2], v[3]};
          continue;
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
template <int BM, int BN, int WM, int WN, int NS, int EPI, bool HAS_R, bool HAS_RS>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_glds_kernel(GemmArgs a) {
  constexpr int NT = 64 * WM * WN;                     // threads per workgroup
  constexpr int WTM = BM / WM, WTN = BN / WN;          // per-wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPD = NT / 8;                          // tile rows covered by one DMA pass (8 lanes x 16 B per 128-B row)
  constexpr int PA = BM / RPD, PB = BN / RPD;
  constexpr int CLD = BN + 8;                          // epilogue tile row stride (elements)
  static_assert(BM * CLD <= NS * (BM + BN) * BK, "epilogue tile must fit in the operand buffers");
  // operand ring + 1 KB for the bias slice.  ONE LDS object on purpose: with a second __shared__ array next to LDS-DMA
  // traffic the compiler's waitcnt pass starts guarding ds_reads with vmcnt(0), which drains the ring.
  __shared__ __attribute__((aligned(16))) bf16 smem[NS * (BM + BN) * BK + 512];
  bf16* As = smem;                                     // [NS][BM*BK]
  bf16* Bs = smem + NS * BM * BK;                      // [NS][BN*BK]
  float* bias_s = reinterpret_cast<float*>(smem + NS * (BM + BN) * BK);

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

  stage_bias<BN>(a, bias_s, tn0);
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
  if (tm0 + BM <= a.M && tn0 + BN <= a.N)
    tile_epilogue<BM, BN, WM, WN, 1, EPI, HAS_R, HAS_RS, true>(a, acc, smem, bias_s, tm0, tn0);
  else
    tile_epilogue<BM, BN, WM, WN, 1, EPI, HAS_R, HAS_RS, false>(a, acc, smem, bias_s, tm0, tn0);
}

// ---------------------------------------------------------------------------------------------------------------
// v3 ("wide"): 256x256 output tile, 8 waves as 2(M) x 4(N) -> 128x64 per wave, two 64-KB LDS stages of 64-deep K tiles.
//
// What the measurements said (tools/gemm_ab.py / gemm_dbg.py / gemm_trace.py, rocprofv3 PMC, M = 295k..74k):
//  * lock-step waves (all eight in the same phase) add their phases up: MFMA-only 1640 TFLOP/s, +DMA 1016, +ds_read
//    1150, all three 830.  So the eight waves form two groups (wm = 0 / 1: one wave of each per SIMD) that run the same K
//    loop HALF A SUB-TILE APART: a load phase (12 fragment ds_reads) and a math phase (16 MFMAs) per 32-deep sub-tile,
//    a workgroup barrier after each; group 1 enters the loop one barrier late, so on every SIMD one wave feeds the matrix
//    pipe while its partner fetches.
//  * with a 32-deep K tile (64-byte rows) every LDS-DMA instruction costs the texture addresser ~31 cycles (TA_BUSY 54 %,
//    one request per 64-B half line): 1024 TA cycles per 1024 MFMA cycles, and the load phase (670-730 cycles) outlasts
//    the math phase (512).  128-byte rows halve the requests per byte, hence K tiles of 64 and only two stages (three do not
//    fit the 160-KB LDS): the whole next K tile is requested in the first load phase of the current one and has 2.5 phases
//    to land.
template <int WM, int WN, int EPI, bool HAS_R, bool HAS_RS, bool TRACE = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_wide_kernel(GemmArgs a) {
  constexpr int BM = 256, BN = 256, NS = 2;
  constexpr int NT = 64 * WM * WN;
  static_assert(NT == 512 && WM == 2, "two wave groups of four; DMA pass geometry for 8 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPD = NT / 8;                          // 64 tile rows per DMA pass (8 lanes x 16 B per 128-B row)
  constexpr int PA = BM / RPD, PB = BN / RPD;          // 4 + 4 DMA instructions per wave per K tile
  constexpr int NDMA = PA + PB;
  constexpr int STAGE = (BM + BN) * BK;
  constexpr int CLD = BN + 8;
  static_assert((size_t)(BM / 2) * CLD * 2 <= (size_t)NS * STAGE * 2, "epilogue half tile must fit in the ring");
  // ring stages + 1 KB of bias (ONE LDS object: a second __shared__ array next to LDS-DMA makes the compiler's waitcnt
  // pass guard ds_reads with vmcnt(0))
  __shared__ __attribute__((aligned(16))) bf16 smem[NS * STAGE + 512];
  float* bias_s = reinterpret_cast<float*>(smem + NS * STAGE);

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

  const int srow = wave * 8 + (lane >> 3), spc = lane & 7;
  const bf16* xsrc[PA];
  const bf16* wsrc[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = srow + p * RPD;
    xsrc[p] = a.X + (size_t)min(tm0 + row, a.M - 1) * a.ldx + ((spc ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int row = srow + p * RPD;
    wsrc[p] = a.W + (size_t)min(tn0 + row, a.N - 1) * a.ldw + ((spc ^ ((row >> 1) & 7)) << 3);
  }
  auto dma = [&](int kt) {
    bf16* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int p = 0; p < PA; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[p] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(st + (p * RPD + wave * 8) * BK), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < PB; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[p] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(st + BM * BK + (p * RPD + wave * 8) * BK), 16, 0, 0);
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
  bf16x8 fa[2][TM], fb[2][TN];
  // fragments of the 32-deep sub-tile `sub` of K tile kt (two MFMA k-steps)
  auto load_phase = [&](int kt, int sub) {
    const bf16* Ac = smem + (kt & 1) * STAGE;
    const bf16* Bc = Ac + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * WTM + i * 32 + frow, sub * 4 + ks * 2 + fk));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * WTN + j * 32 + frow, sub * 4 + ks * 2 + fk));
    }
  };
  auto math_phase = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // workgroup barrier closing a phase; `landed`: first make this wave's share of the next K tile resident (all of its
  // requests were issued two or more phases ago, nothing newer is in flight, so the wait is a plain vmcnt(0))
  auto phase_barrier = [&](bool landed) {
    if (landed) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  stage_bias<BN>(a, bias_s, tn0);
  dma(0);
  phase_barrier(true);                                    // K tile 0 resident for everyone
  if (wm == 1) phase_barrier(false);                      // stagger: group 1 runs one phase behind
  // TRACE build (tools/gemm_trace.py): s_memtime deltas of the segments of a K tile, summed per wave
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = 0;
  if constexpr (TRACE) t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = t0;
  auto mark = [&](int i) {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tr[i] += t - t0;
      t0 = t;
    }
  };
  // Global phase p of group 0 = 4 kt + {0: load sub 0, 1: math, 2: load sub 1, 3: math}; group 1 does the same at p + 1.
  // Stage (kt+1)&1 was last read in phase 4 kt - 1 (group 1, sub-tile 1 of K tile kt-1), so the requests for K tile kt+1
  // go out in the first load phase of K tile kt; every wave makes its share resident before the barrier that closes global
  // phase 4 kt + 3 -- the end of math(sub 1) for group 0, of load(sub 1) for group 1 -- after which group 0 reads it.
  for (int kt = 0; kt < nk; ++kt) {
    load_phase(kt, 0);
    if (kt + 1 < nk) dma(kt + 1);
    if constexpr (TRACE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mark(0);
    phase_barrier(false);
    mark(1);
    math_phase();
    mark(2);
    phase_barrier(false);
    mark(3);
    load_phase(kt, 1);
    if constexpr (TRACE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mark(4);
    phase_barrier(wm == 1);
    mark(5);
    math_phase();
    mark(6);
    phase_barrier(wm == 0);
    mark(7);
  }
  if (wm == 0) phase_barrier(false);                      // re-align the barrier count of the two groups
  if constexpr (TRACE) {
    if (lane == 0 && blockIdx.x < 8) {
      float* o = a.colpart + (blockIdx.x * 8 + wave) * 16;
      for (int i = 0; i < 8; ++i) o[i] = (float)tr[i];
      o[8] = (float)(__builtin_amdgcn_s_memtime() - tstart);
      o[9] = (float)nk;
    }
    return;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: two 128-row slabs through the (now idle) ring -------------------------------------------------
  if (tm0 + BM <= a.M && tn0 + BN <= a.N)
    tile_epilogue<BM, BN, WM, WN, 2, EPI, HAS_R, HAS_RS, true>(a, acc, smem, bias_s, tm0, tn0);
  else
    tile_epilogue<BM, BN, WM, WN, 2, EPI, HAS_R, HAS_RS, false>(a, acc, smem, bias_s, tm0, tn0);
}


// ---------------------------------------------------------------------------------------------------------------
// v3 persistent: the wide kernel walking a run of output tiles per workgroup (one workgroup per CU).  At K = 512 the
// non-persistent kernel spends ~9 of 23 us per tile outside the K loop, most of it waiting for its 128 KB of output to drain
// to HBM before the workgroup may retire and the next one may even start fetching.  Here the K-tile stream simply continues
// across tiles (the last K tile of tile t requests the first of tile t+1), the epilogue goes through the stage that tile's
// last K tile just vacated (four 64-row rounds of 33 KB), its stores drain under the next tile's MFMAs, and the bias slice
// arrives by LDS-DMA ahead of the tile's first K tile (double-buffered by tile parity).
template <int WM, int WN, int EPI, bool HAS_R, bool HAS_RS>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_wide_persist_kernel(GemmArgs a) {
  constexpr int BM = 256, BN = 256, NS = 2;
  constexpr int NT = 64 * WM * WN;
  static_assert(NT == 512 && WM == 2, "two wave groups of four; DMA pass geometry for 8 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPD = NT / 8;
  constexpr int PA = BM / RPD, PB = BN / RPD;
  constexpr int STAGE = (BM + BN) * BK;
  constexpr int CLD = BN + 8;
  constexpr int ROUNDS = 4;
  static_assert((size_t)(BM / ROUNDS) * CLD * 2 <= (size_t)STAGE * 2, "an epilogue round must fit one ring stage");
  __shared__ __attribute__((aligned(16))) bf16 smem[NS * STAGE + 1024];          // + two 1-KB bias slices (ONE LDS object)
  float* bias_s = reinterpret_cast<float*>(smem + NS * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tilesN = (a.N + BN - 1) / BN, tilesM = (a.M + BM - 1) / BM;
  const int nblk = tilesM * tilesN;
  // XCD x owns a contiguous range of tile ids; its P workgroups stride through it together, so one XCD's L2 serves P
  // neighbouring tiles at any moment (shared X row panels / W column panels)
  const int P = gridDim.x >> 3, xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int first = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + widx;
  const int count = q8 + (xcd < r8 ? 1 : 0);
  const int T = widx < count ? (count - widx + P - 1) / P : 0;
  if (T == 0) return;

  const int srow = wave * 8 + (lane >> 3), spc = lane & 7;
  const bf16* xbase; const bf16* wbase;                  // wave-uniform tile origins (SGPRs)
  unsigned xo[PA], wo[PB];                               // per-lane byte offsets inside the tile (swizzled chunk, clamped row)
  // point the DMA addressing at tile `seq` and request its bias slice (every wave writes the same 1 KB; lanes past the
  // slice fetch a clamped address); ordered before the tile's first K-tile request, so the same waits cover it
  auto aim = [&](int seq) {
    const int id = first + seq * P;
    const int m0 = (id / tilesN) * BM, n0 = (id % tilesN) * BN;
    xbase = a.X + (size_t)m0 * a.ldx;
    wbase = a.W + (size_t)n0 * a.ldw;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int row = srow + p * RPD;
      xo[p] = (unsigned)(min(row, a.M - 1 - m0) * a.ldx + ((spc ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int row = srow + p * RPD;
      wo[p] = (unsigned)(min(row, a.N - 1 - n0) * a.ldw + ((spc ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
    if (a.bias)
      lds_dma16(a.bias, (unsigned)min(n0 + lane * 4, a.N - 4) * 4u, bias_s + (seq & 1) * 256);
  };
  auto dma = [&](int g, int kt) {
    bf16* st = smem + (g & 1) * STAGE;
#pragma unroll
    for (int p = 0; p < PA; ++p)
      lds_dma16(xbase + kt * BK, xo[p], st + (p * RPD + wave * 8) * BK);
#pragma unroll
    for (int p = 0; p < PB; ++p)
      lds_dma16(wbase + kt * BK, wo[p], st + BM * BK + (p * RPD + wave * 8) * BK);
  };

  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  const int nk = a.K / BK;
  const int frow = lane & 31, fk = lane >> 5;
  bf16x8 fa[2][TM], fb[2][TN];
  auto load_phase = [&](int g, int sub) {
    const bf16* Ac = smem + (g & 1) * STAGE;
    const bf16* Bc = Ac + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * WTM + i * 32 + frow, sub * 4 + ks * 2 + fk));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * WTN + j * 32 + frow, sub * 4 + ks * 2 + fk));
    }
  };
  auto math_phase = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_barrier = [&](bool landed) {
    if (landed) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  aim(0);
  dma(0, 0);
  phase_barrier(true);
  int g = 0;                                              // K tiles consumed so far (selects the ring stage)
  for (int seq = 0; seq < T; ++seq) {
    const int id = first + seq * P;
    const int tm0 = (id / tilesN) * BM, tn0 = (id % tilesN) * BN;
    if (wm == 1) phase_barrier(false);                    // stagger: group 1 runs one phase behind (see the wide kernel)
    for (int kt = 0; kt < nk; ++kt, ++g) {
      load_phase(g, 0);
      if (kt + 1 < nk) dma(g + 1, kt + 1);
      else if (seq + 1 < T) { aim(seq + 1); dma(g + 1, 0); }     // the stream continues into the next output tile
      phase_barrier(false);
      math_phase();
      phase_barrier(false);
      load_phase(g, 1);
      phase_barrier(wm == 1);
      math_phase();
      phase_barrier(wm == 0);
    }
    if (wm == 0) phase_barrier(false);                    // re-align the barrier count of the two groups
    bf16* Cs = smem + ((g - 1) & 1) * STAGE;              // the stage the last K tile just vacated
    const float* bias_c = bias_s + (seq & 1) * 256;
    if (tm0 + BM <= a.M && tn0 + BN <= a.N)
      tile_epilogue<BM, BN, WM, WN, ROUNDS, EPI, HAS_R, HAS_RS, true, true>(a, acc, Cs, bias_c, tm0, tn0);
    else
      tile_epilogue<BM, BN, WM, WN, ROUNDS, EPI, HAS_R, HAS_RS, false, true>(a, acc, Cs, bias_c, tm0, tn0);
    zero_acc();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v4 persistent: the same K loop, but nothing between two output tiles is a workgroup-wide event any more.
//  * epilogue = wave_epilogue<>: each wave drains its own 128x64 sub-tile through a private 4-KB LDS slab (dedicated: 8 x 4 KB
//    next to the two ring stages = the whole 160 KB), no barriers -- the v3 epilogue (4 rounds x 2 workgroup barriers, half
//    the waves idle during each staging pass, MFMA pipe idle throughout) cost ~7 of the ~21 us of a K = 512 tile;
//  * the two wave groups keep their stagger ACROSS tiles (one extra barrier for group 1 before the first tile, one for group 0
//    after the last), so one group's epilogue runs under the other group's last / first MFMAs;
//  * bias = accumulator seed: the first MFMAs of a tile take C from 32 VGPRs built out of scalar loads of the wave's 64 bias
//    values (or constant 0) -- no zeroing pass (128 VALU per wave per tile), no bias adds, no bias slice in LDS.
template <int WM, int WN, int EPI, bool HAS_R, bool HAS_RS, bool TRACE = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_wide_persist2_kernel(GemmArgs a) {
  constexpr int BM = 256, BN = 256, NS = 2;
  constexpr int NT = 64 * WM * WN;
  static_assert(NT == 512 && WM == 2 && WN == 4, "two wave groups of four; DMA pass geometry for 8 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TN == 2 && WTN == 64, "wave_epilogue slab = 32 x 64");
  constexpr int RPD = NT / 8;
  constexpr int PA = BM / RPD, PB = BN / RPD;
  constexpr int STAGE = (BM + BN) * BK;
  constexpr int SLAB = 32 * 64;                                                   // elements of one wave's slab (4 KB)
  __shared__ __attribute__((aligned(16))) bf16 smem[NS * STAGE + 8 * SLAB];       // ONE LDS object: 128 KB ring + 32 KB slabs

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  bf16* cw = smem + NS * STAGE + wave * SLAB;
  const int tilesN = (a.N + BN - 1) / BN, tilesM = (a.M + BM - 1) / BM;
  const int nblk = tilesM * tilesN;
  const int P = gridDim.x >> 3, xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int first = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + widx;
  const int count = q8 + (xcd < r8 ? 1 : 0);
  const int T = widx < count ? (count - widx + P - 1) / P : 0;
  if (T == 0) return;

  const int srow = wave * 8 + (lane >> 3), spc = lane & 7;
  const bf16* xbase; const bf16* wbase;
  unsigned xo[PA], wo[PB];
  auto aim = [&](int seq) {
    const int id = first + seq * P;
    const int m0 = (id / tilesN) * BM, n0 = (id % tilesN) * BN;
    xbase = a.X + (size_t)m0 * a.ldx;
    wbase = a.W + (size_t)n0 * a.ldw;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int row = srow + p * RPD;
      xo[p] = (unsigned)(min(row, a.M - 1 - m0) * a.ldx + ((spc ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int row = srow + p * RPD;
      wo[p] = (unsigned)(min(row, a.N - 1 - n0) * a.ldw + ((spc ^ ((row >> 1) & 7)) << 3)) * 2u;
    }
  };
  auto dma = [&](int g, int kt) {
    bf16* st = smem + (g & 1) * STAGE;
#pragma unroll
    for (int p = 0; p < PA; ++p)
      lds_dma16(xbase + kt * BK, xo[p], st + (p * RPD + wave * 8) * BK);
#pragma unroll
    for (int p = 0; p < PB; ++p)
      lds_dma16(wbase + kt * BK, wo[p], st + BM * BK + (p * RPD + wave * 8) * BK);
  };

  f32x16 acc[TM][TN];
  const int nk = a.K / BK;
  const int frow = lane & 31, fk = lane >> 5;
  bf16x8 fa[2][TM], fb[2][TN];
  auto load_phase = [&](int g, int sub) {
    const bf16* Ac = smem + (g & 1) * STAGE;
    const bf16* Bc = Ac + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(Ac + swz(wm * WTM + i * 32 + frow, sub * 4 + ks * 2 + fk));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(Bc + swz(wn * WTN + j * 32 + frow, sub * 4 + ks * 2 + fk));
    }
  };
  auto math_phase = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // first math phase of a tile: C = the bias of the wave's 64 columns in accumulator layout (element q*4+e of block j is
  // column j*32 + q*8 + 4*(lane>>5) + e), read with SCALAR loads (wave-uniform address) and picked per half-wave
  auto math_phase_seeded = [&](int n0w) {
    f32x16 seed[TN];
    if (EPI != 2 && a.bias) {
      typedef __attribute__((address_space(4))) const float cfloat;
      cfloat* bp = (cfloat*)(uintptr_t)(a.bias + n0w);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bp[j * 32 + q * 8 + e], hi = bp[j * 32 + q * 8 + 4 + e];
            seed[j][q * 4 + e] = fk ? hi : lo;
          }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) seed[j][r] = 0.f;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[0][i], seed[j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_barrier = [&](bool landed) {
    if (landed) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  aim(0);
  dma(0, 0);
  phase_barrier(true);
  if (wm == 1) phase_barrier(false);                      // group 1 runs one phase behind group 0 from here to the end
  int g = 0;
  // TRACE build (tools/gemm_trace.py persist): s_memtime ticks per wave, summed over its tiles --
  //   [0] first K tile of a tile  [1] other K tiles  [2] epilogue  [3..6] the four barriers of a K tile (wait + barrier)  [7] load+dma issue
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = 0, tb = 0;
  if constexpr (TRACE) t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = t0;
  auto mark = [&](int i) {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tr[i] += t - t0;
      t0 = t;
    }
  };
  auto bar_t = [&](int i, bool landed) {
    if constexpr (TRACE) tb = __builtin_amdgcn_s_memtime();
    phase_barrier(landed);
    if constexpr (TRACE) tr[3 + i] += __builtin_amdgcn_s_memtime() - tb;
  };
  for (int seq = 0; seq < T; ++seq) {
    const int id = first + seq * P;
    const int tm0 = (id / tilesN) * BM, tn0 = (id % tilesN) * BN;
    for (int kt = 0; kt < nk; ++kt, ++g) {
      if constexpr (TRACE) tb = __builtin_amdgcn_s_memtime();
      load_phase(g, 0);
      if (kt + 1 < nk) dma(g + 1, kt + 1);
      else if (seq + 1 < T) { aim(seq + 1); dma(g + 1, 0); }
      if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tr[7] += __builtin_amdgcn_s_memtime() - tb; }
      bar_t(0, false);
      if (kt == 0) math_phase_seeded(tn0 + wn * WTN); else math_phase();
      bar_t(1, false);
      load_phase(g, 1);
      bar_t(2, wm == 1);
      math_phase();
      bar_t(3, wm == 0);
      mark(kt == 0 ? 0 : 1);
    }
    if (tm0 + BM <= a.M)
      wave_epilogue<TM, EPI, HAS_R, HAS_RS, true>(a, acc, cw, tm0 + wm * WTM, tn0 + wn * WTN);
    else
      wave_epilogue<TM, EPI, HAS_R, HAS_RS, false>(a, acc, cw, tm0 + wm * WTM, tn0 + wn * WTN);
    mark(2);
  }
  if (wm == 0) phase_barrier(false);                      // every wave passes the same number of barriers
  if constexpr (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && blockIdx.x < 8) {
      float* o = a.colpart + (blockIdx.x * 8 + wave) * 16;
      for (int i = 0; i < 8; ++i) o[i] = (float)tr[i];
      o[8] = (float)(__builtin_amdgcn_s_memtime() - tstart);
      o[9] = (float)nk;
      o[10] = (float)T;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v5 ("q8", round 3): the persistent kernel with the K loop re-cut along the OUTPUT instead of along K -- the 8-phase schedule of
// cdna_hip_programming.md section 5 ("The 256^2 8-phase template") on this kernel's LDS images, tile walk and wave-private epilogue.
//
// Why: in v4 a wave requests the whole next K tile (8 LDS-DMA instructions) at the head of one load phase.  The texture addresser
// takes ~31 cycles per 1-KB instruction, so the 4 x 8 instructions of a wave group hold that phase for ~1000 cycles while the
// partner group's 16 MFMAs need 512: two of the four phases of every K tile run at half the matrix pipe's pace (trace: 3.3-3.6 k
// cycles per K tile for 2.05 k of MFMA).  Spreading the same requests over all four phases INCLUDING the math phases made it
// slower (662 vs 568 us at 294912 x 512 x 2048: a wave stalled on a full DMA queue stops feeding the matrix pipe), and a K-split
// sub-tile cannot be re-staged before the whole K tile has been read.  Cutting the wave's 128 x 64 tile into four QUADRANTS
// (64 rows x 32 columns, all of K = 64) instead makes the four half-tiles of a stage (A rows h*64.., W rows j*32.. of every wave)
// die at different times, so ONE half-tile (2 DMA instructions per wave) can be re-staged in EVERY phase, always by the group
// that is in its load half -- 8 instructions = ~250 addresser cycles beside 256 cycles of the partner's MFMAs, continuously.
//   phase  quadrant (A half, W half)   LDS reads (new fragments)      stages (cursor K tile c)     wait
//     P1     (0, 0)                     A0: 8  W0: 4                   W1 of c
//     P2     (0, 1)                     W1: 4                          A1 of c  -> c advances       vmcnt(6)
//     P3     (1, 1)                     A1: 8                          A0 of c
//     P4     (1, 0)                     -  (W0 still in registers)     W0 of c                      vmcnt(6)
// With K tile t in flight the cursor is t+1 in P1/P2 and t+2 in P3/P4 (the buffer of t itself: A0 / W0 were last read in P1, two
// and three phases earlier).  Every half-tile has 3-4 phases to land; a counted vmcnt(6) leaves the three newest half-tiles in
// flight and is followed by a barrier before the NEXT phase reads what it retired (RAW rule of the guide: wait in phase p, read in
// p+1; the group that runs one barrier behind passes its own wait before the leading group's read).  WAR: a half-tile is
// re-staged two or three phases after its last ds_read.  Epilogue stores / residual loads only make the counted waits stricter.
template <int EPI, bool HAS_R, bool HAS_RS, bool TRACE = false>
__global__ __launch_bounds__(512) void gemm_nt_q8_kernel(GemmArgs a) {
  constexpr int BM = 256, BN = 256, WN = 4;
  constexpr int WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int STAGE = (BM + BN) * BK;
  constexpr int SLAB = 32 * 64;
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * STAGE + 8 * SLAB];        // ONE LDS object: 128 KB ring + 32 KB slabs

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  bf16* cw = smem + 2 * STAGE + wave * SLAB;
  const int tilesN = (a.N + BN - 1) / BN, tilesM = (a.M + BM - 1) / BM;
  const int nblk = tilesM * tilesN;
  const int P = gridDim.x >> 3, xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int first = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + widx;
  const int count = q8 + (xcd < r8 ? 1 : 0);
  const int T = widx < count ? (count - widx + P - 1) / P : 0;
  if (T == 0) return;
  const int nk = a.K / BK;

  // ---- DMA side.  A half-tile = 128 tile rows x 64 k = 16 units of 8 rows (one 1-KB DMA instruction each); wave w owns units
  // w and w + 8 of every half-tile:  A half h: rows {0, 128} + h*64 + w*8 .. +8;   W half j: rows {0, 128} + (w>>2)*64 + j*32 + (w&3)*8 .. +8
  const int lr = lane >> 3;
  const unsigned csw = (unsigned)(((lane & 7) ^ (((wave & 1) << 2) + (lr >> 1))) << 4);     // swizzled 16-B chunk, same for all 8 units
  const int ra0 = wave * 8, rb0 = (wave >> 2) * 64 + (wave & 3) * 8;                          // unit base rows (half 0, unit 0)
  const bf16* xbase; const bf16* wbase;                  // origins of the cursor's output tile
  unsigned xo[2][2], wo[2][2];                           // per-lane byte offsets [half][unit]
  int c_seq = 0, c_kt = 0, c_g = 0;                      // cursor: output tile, K tile in it, ring counter (buffer = c_g & 1)
  auto aim = [&]() {
    const int id = first + c_seq * P;
    const int m0 = (id / tilesN) * BM, n0 = (id % tilesN) * BN;
    xbase = a.X + (size_t)m0 * a.ldx;
    wbase = a.W + (size_t)n0 * a.ldw;
    const int xlim = a.M - 1 - m0, wlim = a.N - 1 - n0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xo[h][u] = (unsigned)(min(u * 128 + h * 64 + ra0 + lr, xlim) * a.ldx) * 2u + csw;
        wo[h][u] = (unsigned)(min(u * 128 + h * 32 + rb0 + lr, wlim) * a.ldw) * 2u + csw;
      }
  };
  auto advance = [&]() {                                  // after the last half-tile (A1) of the cursor's K tile
    ++c_g;
    if (++c_kt == nk) {
      if (c_seq + 1 < T) { c_kt = 0; ++c_seq; aim(); }
      else c_kt = nk - 1;                                 // past the end: keep re-requesting the last K tile into buffers that are
    }                                                     // free by construction (no "anything left?" test in the phases)
  };
  auto stageA = [&](int h) {
    bf16* st = smem + (c_g & 1) * STAGE;
    lds_dma16(xbase + c_kt * BK, xo[h][0], st + (h * 64 + ra0) * BK);
    lds_dma16(xbase + c_kt * BK, xo[h][1], st + (128 + h * 64 + ra0) * BK);
  };
  auto stageW = [&](int j) {
    bf16* st = smem + (c_g & 1) * STAGE + BM * BK;
    lds_dma16(wbase + c_kt * BK, wo[j][0], st + (j * 32 + rb0) * BK);
    lds_dma16(wbase + c_kt * BK, wo[j][1], st + (128 + j * 32 + rb0) * BK);
  };

  // ---- MFMA side
  f32x16 acc[TM][TN];
  const int frow = lane & 31, fk = lane >> 5;
  bf16x8 fa[4][2], fb[2][4];                             // A half: [k step][32-row tile];  both W halves: [half][k step]
  auto readA = [&](const bf16* sb, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
        fa[ks][ii] = *reinterpret_cast<const bf16x8*>(sb + swz(wm * WTM + (2 * h + ii) * 32 + frow, ks * 2 + fk));
  };
  auto readW = [&](const bf16* sb, int j) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      fb[j][ks] = *reinterpret_cast<const bf16x8*>(sb + BM * BK + swz(wn * WTN + j * 32 + frow, ks * 2 + fk));
  };
  // bias of column block j in accumulator layout (element q*4+e <-> column j*32 + q*8 + 4*(lane>>5) + e): scalar loads, one select each
  // N need only be a multiple of 64 (a wave's 64 columns are then all inside or all outside): the waves of the last tile column
  // whose columns lie past N run the K loop on clamped W rows and store nothing (Swin stage 0: qkv, N = 384 = 1.5 tiles)
  // Two steps: the scalar loads are REQUESTED in the load half of the phase, the selects that consume them sit behind the phase's barrier
  // (bar_load's s_waitcnt lgkmcnt(0) then finds them landed) -- requested and consumed back to back, each seed exposed a scalar-load
  // latency of ~350 cycles, four times per tile: a bias cost the plain GEMM 6 % (tools/option_ablation.py).
  struct SeedRaw { float v[32]; bool on; };
  auto seed_raw = [&](int n0w, int j) {
    SeedRaw r;
    r.on = EPI != 2 && a.bias && n0w < a.N;
    if (r.on) {
      typedef __attribute__((address_space(4))) const float cfloat;
      cfloat* bp = (cfloat*)(uintptr_t)(a.bias + n0w + j * 32);
#pragma unroll
      for (int i = 0; i < 32; ++i) r.v[i] = bp[i];
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) r.v[i] = 0.f;
    }
    return r;
  };
  auto seed_sel = [&](const SeedRaw& r) {
    f32x16 sd;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) sd[q * 4 + e] = fk ? r.v[q * 8 + 4 + e] : r.v[q * 8 + e];
    return sd;
  };
  auto mma = [&](int h, int j) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
        acc[2 * h + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][ks], fa[ks][ii], acc[2 * h + ii][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma_seeded = [&](int h, int j, const f32x16& sd) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
      acc[2 * h + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][0], fa[0][ii], sd, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
        acc[2 * h + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][ks], fa[ks][ii], acc[2 * h + ii][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // first barrier of a phase: the requests of this load half are out; `counted`: make everything but the three newest half-tiles
  // resident first.  Second barrier: the MFMAs of the phase are issued.
  auto bar_load = [&](bool counted) {
    if (counted) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar_math = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K tile 0 complete, A0 / W0 of K tile 1 requested (what the steady state has issued before P1 of K tile 0)
  aim();
  stageA(0); stageW(0); stageW(1); stageA(1); advance();
  stageA(0); stageW(0);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // Group 1 runs one barrier behind group 0 INSIDE a tile's K loop.  Round 4: the stagger is taken down at the end of every K loop
  // (one extra barrier for group 0) and put up again at the start of the next (one for group 1), so that both groups enter their
  // wave-private epilogues together.  With the stagger kept across tiles (round 3; act bit 0x4000 = that form, for A/B runs) group 1
  // sat at its last P4 barrier for the whole of group 0's epilogue and group 0 at its first P1 barrier for the whole of group 1's:
  // the two epilogues ran one after the other with one wave per SIMD (trace: 2 x 11.2 k ticks per GELU tile, K loop 27 k).
  const bool keep_stagger = (a.act & 0x4000) != 0;
  if (keep_stagger && wm == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  int g = 0;
  // TRACE build (tools/gemm_trace.py q8): s_memtime ticks per wave summed over its K tiles -- for each of the four phases
  // [4p + 0] load half (reads + DMA issued AND the reads returned: s_memtime drains lgkmcnt)  [4p + 1] wait + first barrier
  // [4p + 2] MFMA issue  [4p + 3] second barrier;  [16] epilogue
  unsigned long long tr[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = 0;
  if constexpr (TRACE) t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = t0;
  auto mark = [&](int i) {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tr[i] += t - t0;
      t0 = t;
    }
  };
  for (int seq = 0; seq < T; ++seq) {
    const int id = first + seq * P;
    const int tm0 = (id / tilesN) * BM, tn0 = (id % tilesN) * BN;
    const int n0w = tn0 + wn * WTN;
    if (!keep_stagger && wm == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const bf16* sb = smem + (g & 1) * STAGE;
      // P1: quadrant (0, 0)
      readW(sb, 0);
      __builtin_amdgcn_sched_barrier(0);
      readA(sb, 0);
      stageW(1);
      mark(0);
      if (kt == 0) {
        const SeedRaw raw = seed_raw(n0w, 0);
        bar_load(false);
        mark(1);
        mma_seeded(0, 0, seed_sel(raw));
      } else {
        bar_load(false);
        mark(1);
        mma(0, 0);
      }
      mark(2);
      bar_math();
      mark(3);
      // P2: quadrant (0, 1)
      readW(sb, 1);
      stageA(1);
      advance();
      mark(4);
      if (kt == 0) {
        const SeedRaw raw = seed_raw(n0w, 1);
        bar_load(true);
        mark(5);
        mma_seeded(0, 1, seed_sel(raw));
      } else {
        bar_load(true);
        mark(5);
        mma(0, 1);
      }
      mark(6);
      bar_math();
      mark(7);
      // P3: quadrant (1, 1)
      readA(sb, 1);
      stageA(0);
      mark(8);
      if (kt == 0) {
        const SeedRaw raw = seed_raw(n0w, 1);
        bar_load(false);
        mark(9);
        mma_seeded(1, 1, seed_sel(raw));
      } else {
        bar_load(false);
        mark(9);
        mma(1, 1);
      }
      mark(10);
      bar_math();
      mark(11);
      // P4: quadrant (1, 0) -- W half 0 is still in registers
      stageW(0);
      mark(12);
      if (kt == 0) {
        const SeedRaw raw = seed_raw(n0w, 0);
        bar_load(true);
        mark(13);
        mma_seeded(1, 0, seed_sel(raw));
      } else {
        bar_load(true);
        mark(13);
        mma(1, 0);
      }
      mark(14);
      bar_math();
      mark(15);
    }
    if (!keep_stagger && wm == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    if constexpr (TRACE) t0 = __builtin_amdgcn_s_memtime();
    if (n0w >= a.N) {
    } else if (tm0 + BM <= a.M)
      wave_epilogue<TM, EPI, HAS_R, HAS_RS, true>(a, acc, cw, tm0 + wm * WTM, tn0 + wn * WTN);
    else
      wave_epilogue<TM, EPI, HAS_R, HAS_RS, false>(a, acc, cw, tm0 + wm * WTM, tn0 + wn * WTN);
    mark(16);
  }
  if (keep_stagger && wm == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }   // every wave passes the same number of barriers
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS allocation
  if constexpr (TRACE) {
    if (lane == 0 && blockIdx.x < 8) {
      float* o = a.colpart + (blockIdx.x * 8 + wave) * 24;
      for (int i = 0; i < 17; ++i) o[i] = (float)tr[i];
      o[17] = (float)(__builtin_amdgcn_s_memtime() - tstart);
      o[18] = (float)nk;
      o[19] = (float)T;
    }
  }
}

// Variants the v4 (wave-private epilogue) persistent kernel serves: those it compiles without scratch.  gelu' * aux WITH column sums
// and the residual-without-DropPath forms spill 60-120 VGPRs around its longer live ranges and stay on v3; gelu' * aux without
// column sums (the caller takes the bias gradient from the weight-gradient kernel) is served.
__device__ float g_gemm_one = 1.0f;        // the scale of the residual-without-DropPath form served by the DropPath instantiation

template <int EPI, bool R, bool RS>
constexpr bool kV4Ok = (EPI == 0 && (!R || RS)) || (EPI == 1 && !R) || EPI == 2 || (EPI == 3 && RS);

}  // namespace

#ifdef FIBER_P2_PROBE   // compile-time probe: ONE instantiation (register / scratch check while editing a K loop)
#ifndef FIBER_P2_PROBE_EPI
#define FIBER_P2_PROBE_EPI 0
#endif
extern "C" void fiber_p2_probe(GemmArgs a, hipStream_t st) {
  hipLaunchKernelGGL((gemm_nt_q8_kernel<FIBER_P2_PROBE_EPI, false, false>), dim3(256), dim3(512), 0, st, a);
}
#else
// C ABI ---------------------------------------------------------------------------------------------------------
// Row-tile height the dispatcher will use for an [M,N,K] problem (rows of `colpart` = ceil(M / tile)).
extern "C" int fiber_gemm_row_tile(int M, int N, int K) {
  const long big = (long)cdiv(M, 128) * cdiv(N, 128), huge = (long)cdiv(M, 256) * cdiv(N, 128);
  static const int force = getenv("FIBER_GEMM_TILE") ? atoi(getenv("FIBER_GEMM_TILE")) : 0;
  static const int nowide = getenv("FIBER_GEMM_NOWIDE") ? atoi(getenv("FIBER_GEMM_NOWIDE")) : 0;
  const long wide = (long)cdiv(M, 256) * cdiv(N, 256);
  if (!nowide && force == 0 && wide >= 200 && N % 256 == 0 && K >= 128 && K % 64 == 0) return 256;
  if ((huge >= 400 && K >= 256 && force == 0) || force == 256) return 256;
  if (big >= 192 || force == 128) return 128;
  return 64;
}

// Y = rowscale * act(X.W^T + bias) + residual.  bias: fp32[N] or NULL; residual: bf16[M,ldr] or NULL; act: 0 none, 1 exact GELU;
// act | 0x100 (act 0 only, no residual): Y is fp32 [M, ldy] -- narrow outputs whose consumers need more than bf16 (the
// offset predictor of the deformable convolutions: sampling positions);
// rowscale: fp32[M / rows_per_sample] or NULL (per-sample DropPath factor on the branch, swin_transformer.py:390-391)
// act | 0x800 (act 0, residual required): the fp32 RESIDUAL STREAM form -- `residual` is fp32 [M, ldr], the sum goes to `Ypre`
// viewed as fp32 [M, ldy] and, if Y is non-NULL, once more as bf16 to Y (the shadow GEMM consumers of the stream read);
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
  static const int stagger_env = getenv("FIBER_GEMM_STAGGER") ? atoi(getenv("FIBER_GEMM_STAGGER")) : 0;   // 1: round-3 barrier form (A/B runs)
  if (stagger_env) a.act |= 0x4000;
  if ((size_t)M * N * 2 > ((size_t)256 << 20)) a.act |= 0x2000;   // output larger than the last-level cache: streaming stores (gemm_epilogue.h st_out)
  const long big = (long)cdiv(M, 128) * cdiv(N, 128);
  const long huge = (long)cdiv(M, 256) * cdiv(N, 128);
  const long wide = (long)cdiv(M, 256) * cdiv(N, 256);
  const int mode = act & 0xff;
  if ((mode == 2 || colpart) && !((K % 64 == 0) && (N % 8 == 0) && (ldy % 8 == 0))) return FIBER_EINVAL;   // LDS-DMA kernels only
  if (mode == 2 && residual) return FIBER_EINVAL;
  if ((act & 0x800) && (mode != 0 || !residual || !Ypre || colpart || (act & 0x100))) return FIBER_EINVAL;
  if (colpart && mode != 2 && !(act & 0x1600)) return FIBER_EINVAL;
  const bool v2 = (K % 64 == 0) && (N % 8 == 0) && (ldy % 8 == 0) && (!residual || ldr % 8 == 0) && !getenv("FIBER_GEMM_V1");
  // Tile choice.  256x256 (K step 32, two wave groups half a tile apart) whenever N is a multiple of 256 and there are
  // enough tiles; otherwise 256x128 / 128x128 / 64x64 on the 64-deep ring.  FIBER_GEMM_TILE / FIBER_GEMM_NOWIDE force a
  // choice for A/B runs (tools/gemm_ab.py).
  static const int force = getenv("FIBER_GEMM_TILE") ? atoi(getenv("FIBER_GEMM_TILE")) : 0;
  static const int nowide = getenv("FIBER_GEMM_NOWIDE") ? atoi(getenv("FIBER_GEMM_NOWIDE")) : 0;
  static const int persist_env = getenv("FIBER_GEMM_PERSIST") ? atoi(getenv("FIBER_GEMM_PERSIST")) : 1;
  static const int persist_min = getenv("FIBER_GEMM_PERSIST_MIN") ? atoi(getenv("FIBER_GEMM_PERSIST_MIN")) : 200;
  static const int q8_env = getenv("FIBER_GEMM_Q8") ? atoi(getenv("FIBER_GEMM_Q8")) : 1;   // 0: the v4 K loop (A/B runs)
  // will FIBER_LAUNCH_EPI below pick gemm_nt_q8_kernel for this call?  (run-time mirror of kV4Ok and of the colpart exclusion)
  const bool has_r = residual != nullptr, has_rs = rowscale != nullptr;
  const bool v4ok = (act & 0x800) ? has_rs : mode == 0 ? (!has_r || has_rs) : mode == 1 ? !has_r : mode == 2;
  const bool q8_serves = persist_env == 1 && q8_env && wide >= persist_min && v4ok && !(mode == 2 && colpart) && !(act & 0x1600);
  int shape;                                             // 0 wide, 1 256x128, 2 128x128, 3 64x64, 4/5 register-staged
  if (act & 0x100) {                                      // fp32 output: the register-staged kernels only
    if (mode != 0 || residual || colpart) return FIBER_EINVAL;
    shape = big >= 192 ? 4 : 5;
  } else if (v2 && !nowide && force == 0 && wide >= 200 && K >= 128 && !(mode == 2 && !q8_serves) &&
             (N % 256 == 0 || (N % 64 == 0 && N > 256 && q8_serves))) shape = 0;   // (only the q8 kernel masks a partial tile column; gelu' * aux
                                                                                    //  with column sums / without q8: the 256x128 ring kernel)
  else if (v2 && ((huge >= 400 && K >= 256 && force == 0) || force == 256)) shape = 1;
  else if (big >= 192 || force == 128) shape = v2 ? 2 : 4;
  else shape = v2 ? 3 : 5;
  const long small = (long)cdiv(M, 64) * cdiv(N, 64);
  const bool persist = persist_env && wide >= persist_min;   // (q8 wins from one tile per CU on: 240 tiles 31.5 -> 28.1 us, 97.5 -> 85.1 us at K = 3072)
#define FIBER_LAUNCH_EPI(EPI, R, RS)                                                                                          \
  do {                                                                                                                        \
    if (shape == 0 && EPI == 2 && !(persist && q8_env && !a.colpart)) return FIBER_EINVAL;   /* (the pre-q8 kernels are not built for gelu' * aux: 104-168 B of scratch) */ \
    if (shape == 0 && persist && persist_env == 2) hipLaunchKernelGGL((gemm_nt_wide_persist_kernel<2, 4, (EPI == 2 ? 0 : EPI), R, RS>), dim3(256), dim3(512), 0, stream, a); \
    else if (shape == 0 && persist && q8_env && kV4Ok<EPI, R, RS> && !(EPI == 2 && a.colpart)) hipLaunchKernelGGL((gemm_nt_q8_kernel<kV4Ok<EPI, R, RS> ? EPI : 0, kV4Ok<EPI, R, RS> && R, RS>), dim3(256), dim3(512), 0, stream, a); \
    else if (shape == 0 && persist && kV4Ok<EPI, R, RS> && !(EPI == 2 && a.colpart)) hipLaunchKernelGGL((gemm_nt_wide_persist2_kernel<2, 4, kV4Ok<EPI, R, RS> ? EPI : 0, kV4Ok<EPI, R, RS> && R, RS>), dim3(256), dim3(512), 0, stream, a); \
    else if (shape == 0 && persist) hipLaunchKernelGGL((gemm_nt_wide_persist_kernel<2, 4, (EPI == 2 ? 0 : EPI), R, RS>), dim3(256), dim3(512), 0, stream, a); \
    else if (shape == 0) hipLaunchKernelGGL((gemm_nt_wide_kernel<2, 4, (EPI == 2 ? 0 : EPI), R, RS>), dim3((unsigned)wide), dim3(512), 0, stream, a);   \
    else if (shape == 1) hipLaunchKernelGGL((gemm_nt_glds_kernel<256, 128, 4, 2, 3, EPI, R, RS>), dim3((unsigned)huge), dim3(512), 0, stream, a); \
    else if (shape == 2) hipLaunchKernelGGL((gemm_nt_glds_kernel<128, 128, 2, 2, 2, EPI, R, RS>), dim3((unsigned)big), dim3(256), 0, stream, a);  \
    else hipLaunchKernelGGL((gemm_nt_glds_kernel<64, 64, 2, 2, 2, EPI, R, RS>), dim3((unsigned)small), dim3(256), 0, stream, a);                   \
  } while (0)
  if (shape == 0 && (act & 0x1000)) {                     // tools/gemm_trace.py q8: per-phase timing of the v5 kernel
    a.act &= 0x40ff;
    if ((act & 0xff) == 1) hipLaunchKernelGGL((gemm_nt_q8_kernel<1, false, false, true>), dim3(256), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((gemm_nt_q8_kernel<0, false, false, true>), dim3(256), dim3(512), 0, stream, a);
    FIBER_CHECK_LAUNCH();
    return FIBER_OK;
  }
  if (shape == 0 && (act & 0x400)) {                      // tools/gemm_trace.py persist: per-segment timing of the persistent kernel
    a.act &= 0xff;
    if ((act & 0xff) == 1) hipLaunchKernelGGL((gemm_nt_wide_persist2_kernel<2, 4, 1, false, false, true>), dim3(256), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((gemm_nt_wide_persist2_kernel<2, 4, 0, false, false, true>), dim3(256), dim3(512), 0, stream, a);
    FIBER_CHECK_LAUNCH();
    return FIBER_OK;
  }
  if (shape == 0 && (act & 0x200)) {                      // tools/gemm_trace.py: per-segment timing build of the wide kernel
    hipLaunchKernelGGL((gemm_nt_wide_kernel<2, 4, 0, false, false, true>), dim3((unsigned)wide), dim3(512), 0, stream, a);
    FIBER_CHECK_LAUNCH();
    return FIBER_OK;
  }
  if (shape == 4) hipLaunchKernelGGL((gemm_nt_kernel<128, 128>), dim3((unsigned)big), dim3(256), 0, stream, a);
  else if (shape == 5) hipLaunchKernelGGL((gemm_nt_kernel<64, 64>), dim3((unsigned)small), dim3(256), 0, stream, a);
  else if (act & 0x800) { if (rowscale) FIBER_LAUNCH_EPI(3, true, true); else FIBER_LAUNCH_EPI(3, true, false); }
  else if (mode == 2) { if (rowscale) FIBER_LAUNCH_EPI(2, false, true); else FIBER_LAUNCH_EPI(2, false, false); }
  else if (mode == 1 && residual) { if (rowscale) FIBER_LAUNCH_EPI(1, true, true); else FIBER_LAUNCH_EPI(1, true, false); }
  else if (mode == 1) { if (rowscale) FIBER_LAUNCH_EPI(1, false, true); else FIBER_LAUNCH_EPI(1, false, false); }
  else if (residual && rowscale) FIBER_LAUNCH_EPI(0, true, true);
  else if (residual && shape == 0 && persist && persist_env == 1 && q8_env) {
    // residual without DropPath (text-layer output projections, blocks with drop_path = 0) on the large-shape path: the q8 instantiation
    // for it spills 120 bytes per lane, the DropPath one does not -- run that one with a one-element scale of 1.0f (x * 1.0f is exact)
    // (a __device__ symbol has one address PER DEVICE: cached per device id, not per process)
    static const float* one_of[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return FIBER_ELAUNCH;
    if (one_of[dev] == nullptr) { float* p = nullptr; if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_gemm_one)) != hipSuccess) return FIBER_ELAUNCH; one_of[dev] = p; }
    const float* one = one_of[dev];
    if (one == nullptr) return FIBER_ELAUNCH;
    a.rowscale = one; a.rows_per_sample = a.M;
    FIBER_LAUNCH_EPI(0, true, true);
  }
  else if (residual) FIBER_LAUNCH_EPI(0, true, false);
  else if (rowscale) FIBER_LAUNCH_EPI(0, false, true);
  else FIBER_LAUNCH_EPI(0, false, false);
#undef FIBER_LAUNCH_EPI
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
#endif  // FIBER_P2_PROBE
