// Shared pieces of the NT GEMM kernels (gemm.hip): argument block, LDS swizzle, fused epilogue.
#pragma once
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

// ---------------------------------------------------------------------------------------------------------------
// Epilogue shared by the LDS-DMA kernels: accumulators -> LDS tile (bf16, in ROUNDS row slabs) -> coalesced 16-byte rows.
// Everything that selects code is a template parameter: with the variants as run-time flags the unrolled store passes
// compiled to ~560 instructions each and the epilogue of a 256x256 tile cost 7 us *without* its stores (ablation in
// tools/gemm_dbg.py) -- as much as the tile's MFMAs at K = 512.
//   EPI 0: Y = rowscale * (acc + bias) + R          (HAS_RS / HAS_R)
//   EPI 1: Ypre = acc + bias (optional);  Y = rowscale * gelu(bf16(acc + bias))
//   EPI 3: EPI 0 on the fp32 RESIDUAL STREAM (HAS_R must be set): R is fp32 [M, ldr], the sum rowscale * (acc + bias) + R is
//          stored in fp32 at Ypre (viewed as float [M, ldy]) and, if Y is non-NULL, once more rounded to bf16 at Y -- the
//          "shadow" that GEMM consumers of the stream read (ops.py: stream pair).  The branch goes through the bf16 staging
//          slab like every other output (rounded once, relative to the BRANCH); the stream itself is never rounded.
//   EPI 2: Y = rowscale * acc * gelu'(aux);  optional per-row-tile column sums of Y (colpart)   (HAS_RS: the DropPath factor
//          of the branch whose backward this is -- (s dY) W2^T = s (dY W2^T), so the scale rides in the epilogue)
//   FULL:  the tile lies entirely inside [M, N] (no row / column predicates)
// Y is read by the next kernel and is stored normally: marking it non-temporal made the isolated GEMM faster (fc1 845 ->
// 772 us at M = 295k, less L2 pollution) but the whole step slower (670 -> 658 images/s, the consumer then misses the
// Infinity Cache).  The pre-activation copy is only read again in the backward pass, so it does stream past the caches.
// Output rows: streaming (non-temporal) stores when the whole output is larger than the 256-MB last-level cache (act bit 0x2000, set by
// the dispatcher) -- nothing of it survives until its consumer runs, and not allocating it keeps the K loop's operand panels in L2: same-box
// step 288.7 -> 286.4 ms, fc1 + GELU 879 -> 845 us; outputs that fit (Swin stage 3: 151 MB) measured 2-4 % slower that way and keep plain stores.
__device__ __forceinline__ void st_out(bf16x8* p, bf16x8 v, bool stream) {
  if (stream) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void st_stream(bf16x8* p, bf16x8 v) { __builtin_nontemporal_store(v, p); }

//   RAWBAR: barriers are s_waitcnt lgkmcnt(0) + s_barrier instead of __syncthreads() -- for the persistent kernel, where a
//          __syncthreads() fence would drain the next tile's LDS-DMA and this tile's own stores (vmcnt(0)).
template <bool RAWBAR>
__device__ __forceinline__ void epi_barrier() {
  if constexpr (RAWBAR) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
}

template <int BM, int BN, int WM, int WN, int ROUNDS, int EPI, bool HAS_R, bool HAS_RS, bool FULL, bool RAWBAR = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], bf16* Cs,
                                              const float* bias_s, int tm0, int tn0) {
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int CLD = BN + 8;
  constexpr int HM = BM / ROUNDS;                       // rows staged per round
  constexpr int CPR = BN / 8;                           // 16-byte chunks per tile row
  constexpr int RPP = NT / CPR;                         // rows per store pass
  constexpr int NPH = HM / RPP;                         // store passes per round
  static_assert(HM % 32 == 0 && HM % RPP == 0, "epilogue geometry");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int erow = tid / CPR, echunk = tid % CPR;
  const int n_out = tn0 + echunk * 8;
  const bool col_ok = FULL || n_out < a.N;
  constexpr bool E0 = EPI == 0 || EPI == 3;            // bias / DropPath scale / residual family
  constexpr bool SIDE = (HAS_R && EPI != 3) || EPI == 2;
  const bf16* sidep = EPI == 2 ? a.aux : a.R;
  const size_t sideld = EPI == 2 ? a.ldaux : a.ldr;
  const bool has_bias = a.bias != nullptr;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int h = 0; h < ROUNDS; ++h) {
    const int mbase = tm0 + h * HM + erow;              // this thread's first output row in the round
    // side rows / DropPath scales are requested BEFORE the staging pass so their latency hides behind it
    bf16x8 side[SIDE ? NPH : 1];
    float prs[((EPI == 1 || EPI == 2) && HAS_RS) ? NPH : 1];
    if constexpr (SIDE) {
      const bf16* sp = sidep + (size_t)mbase * sideld + n_out;
#pragma unroll
      for (int pp = 0; pp < NPH; ++pp)
        if (FULL || (mbase + pp * RPP < a.M && col_ok)) side[pp] = *reinterpret_cast<const bf16x8*>(sp + (size_t)pp * RPP * sideld);
    }
    if constexpr ((EPI == 1 || EPI == 2) && HAS_RS) {
#pragma unroll
      for (int pp = 0; pp < NPH; ++pp) prs[pp] = a.rowscale[min(mbase + pp * RPP, a.M - 1) / a.rows_per_sample];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (ROUNDS == 1 || (wm * WTM + i * 32) / HM == h) {          // this wave's 32-row slab i belongs to round h
        const int ml = wm * WTM + i * 32 + (lane & 31);
        float rsc = 1.f;
        if constexpr (E0 && HAS_RS) rsc = a.rowscale[min(tm0 + ml, a.M - 1) / a.rows_per_sample];
        bf16* crow = Cs + (ml - h * HM) * CLD + wn * WTN + (lane >> 5) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = wn * WTN + j * 32 + q * 8 + (lane >> 5) * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
            if (EPI != 2 && has_bias) {
              const float4 bb = *reinterpret_cast<const float4*>(bias_s + nl);
              v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
            }
            if constexpr (E0 && HAS_RS) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= rsc;
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
            *reinterpret_cast<bf16x4*>(crow + j * 32 + q * 8) = o;
          }
      }
    }
    epi_barrier<RAWBAR>();
    {
      bf16* yp = a.Y + (size_t)mbase * a.ldy + n_out;
      bf16* prep = (EPI == 1 && a.Ypre) ? a.Ypre + (size_t)mbase * a.ldy + n_out : nullptr;
      const size_t ystep = (size_t)RPP * a.ldy;
      const bf16* cp = Cs + erow * CLD + echunk * 8;
#pragma unroll
      for (int pp = 0; pp < NPH; ++pp) {
        if (FULL || (mbase + pp * RPP < a.M && col_ok)) {
          bf16x8 v = *reinterpret_cast<const bf16x8*>(cp + pp * RPP * CLD);
          if constexpr (EPI == 1) {
            if (prep) st_stream(reinterpret_cast<bf16x8*>(prep + pp * ystep), v);
            v = gelu8(v, HAS_RS ? prs[pp] : 1.f);
          } else if constexpr (EPI == 2) {
            v = gelu_grad_mul8(v, side[pp], HAS_RS ? prs[pp] : 1.f);
          }
          if constexpr (EPI == 3) {                      // fp32 residual stream: R and the sum in fp32 (+ bf16 shadow)
            const size_t row = (size_t)(mbase + pp * RPP);
            const float* rp = reinterpret_cast<const float*>(a.R) + row * a.ldr + n_out;
            float* op = reinterpret_cast<float*>(a.Ypre) + row * a.ldy + n_out;
            const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
            const float4 o0 = {bf2f(v[0]) + r0.x, bf2f(v[1]) + r0.y, bf2f(v[2]) + r0.z, bf2f(v[3]) + r0.w};
            const float4 o1 = {bf2f(v[4]) + r1.x, bf2f(v[5]) + r1.y, bf2f(v[6]) + r1.z, bf2f(v[7]) + r1.w};
            *reinterpret_cast<float4*>(op) = o0;
            *reinterpret_cast<float4*>(op + 4) = o1;
            if (a.Y) {
              v[0] = f2bf(o0.x); v[1] = f2bf(o0.y); v[2] = f2bf(o0.z); v[3] = f2bf(o0.w);
              v[4] = f2bf(o1.x); v[5] = f2bf(o1.y); v[6] = f2bf(o1.z); v[7] = f2bf(o1.w);
              st_out(reinterpret_cast<bf16x8*>(yp + pp * ystep), v, (a.act & 0x2000) != 0);
            }
            continue;
          }
          if constexpr (HAS_R) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(side[pp][e]));
          }
          st_out(reinterpret_cast<bf16x8*>(yp + pp * ystep), v, (a.act & 0x2000) != 0);
          if constexpr (EPI == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[e] += bf2f(v[e]);
          }
        }
      }
    }
    if (h + 1 < ROUNDS || EPI == 2 || RAWBAR) epi_barrier<RAWBAR>();     // staged slab fully read before it is overwritten / re-used
  }
  if constexpr (EPI == 2) {
    if (a.colpart) {                                    // column sums of the stored tile: bias gradient of the fused backward
      float* red = reinterpret_cast<float*>(Cs);
      constexpr int LPC = 64 / (CPR < 64 ? CPR : 64);   // lanes of one wave that share a column chunk
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = csum[e];
        if constexpr (LPC >= 2 && CPR <= 32) t += __shfl_xor(t, 32);
        if constexpr (LPC >= 4 && CPR <= 16) t += __shfl_xor(t, 16);
        if constexpr (LPC >= 8 && CPR <= 8) t += __shfl_xor(t, 8);
        csum[e] = t;
      }
      static_assert(CPR == 8 || CPR == 16 || CPR == 32, "column-sum shuffle tree");
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * BN + lane * 8 + e] = csum[e];
      }
      epi_barrier<RAWBAR>();
      for (int c = tid; c < BN; c += NT) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WM * WN; ++w) t += red[w * BN + c];
        if (tn0 + c < a.N) a.colpart[(size_t)(tm0 / BM) * a.N + tn0 + c] = t;
      }
      if constexpr (RAWBAR) epi_barrier<RAWBAR>();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-private epilogue (persistent kernel): every wave drains its own 128x64 sub-tile, 32 rows at a time, through a private
// 4-KB LDS slab.  No workgroup barrier anywhere: the two wave groups keep their half-sub-tile stagger across output tiles, one
// group's staging / GELU / stores run under the other group's MFMAs, and a wave's stores drain under the next tile's K loop.
// The bias is already in the accumulators (the tile's first MFMAs are seeded with it), so EPI 0 / 1 start from acc + bias.
//   slab image: 32 rows x 128 B; 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7) -- the A/B tile swizzle: the 8-byte
//   MFMA-layout writes of 16 consecutive rows and the 16-byte row-contiguous reads of 8 lanes per row both tile the banks
//   stores: 8 rows x 128 B per instruction (one whole cache line per row)
// Column sums (EPI 2) are per 128-row wave sub-tile: `colpart` has two rows per output tile (fiber_gemm_row_tile says 128).
template <int TM, int EPI, bool HAS_R, bool HAS_RS, bool FULL>
__device__ __forceinline__ void wave_epilogue(const GemmArgs& a, f32x16 (&acc)[TM][2], bf16* cw, int m0w, int n0w) {
  // The lane index is re-derived here (v_mbcnt, made opaque) instead of taken from threadIdx: everything below that depends on it
  // only -- slab addresses, the 16-byte column of the lane -- is otherwise hoisted out of the persistent tile loop and held in
  // registers through the K loop, which at 248+ VGPRs pushed the gelu' variants into scratch (and a scratch reload makes the
  // compiler wait vmcnt(0): a drain of the LDS-DMA stream).
  int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane));
  const int wr = lane & 31, wh = lane >> 5;              // staging: row of the slab, which 4-column half of an 8-column group
  const int rr = lane >> 3, rc = lane & 7;               // read-back: row inside an 8-row pass, 16-byte chunk of the 128-B row
  constexpr bool E0 = EPI == 0 || EPI == 3;
  constexpr bool SIDE = (HAS_R && EPI != 3) || EPI == 2;
  const bf16* sidep = EPI == 2 ? a.aux : a.R;
  const size_t sideld = EPI == 2 ? a.ldaux : a.ldr;
  const int n_out = n0w + rc * 8;
  bf16* wbase = cw + wr * 64 + wh * 4;
  const int wsw = (wr >> 1) & 7;
  const bf16* rbase = cw + rr * 64;
  // The LDS queue of a wave executes in order, so slab i+1 may be written right behind the reads of slab i (they still see
  // slab i) without waiting for their data: stage(i+1) / read(i+1) are issued before slab i is processed and stored, and the
  // write -> read -> return latency of a slab hides behind the previous slab's VALU work and store issue.
  auto stage = [&](int i) {
    float rsc = 1.f;
    if constexpr (E0 && HAS_RS) rsc = a.rowscale[min(m0w + i * 32 + wr, a.M - 1) / a.rows_per_sample];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf((E0 && HAS_RS) ? acc[i][j][q * 4 + e] * rsc : acc[i][j][q * 4 + e]);
        *reinterpret_cast<bf16x4*>(wbase + (((j * 4 + q) ^ wsw) << 3)) = o;
      }
  };
  auto read_pass = [&](int pp) {
    const int row = pp * 8 + rr;
    return *reinterpret_cast<const bf16x8*>(rbase + pp * 8 * 64 + ((rc ^ ((row >> 1) & 7)) << 3));
  };
  auto load_side = [&](int i, bf16x8 (&sd)[4], float (&prs)[4]) {
    const int mrow = m0w + i * 32 + rr;
    if constexpr (SIDE) {
      const bf16* sp = sidep + (size_t)mrow * sideld + n_out;
#pragma unroll
      for (int pp = 0; pp < 4; ++pp)
        if (FULL || mrow + pp * 8 < a.M) sd[pp] = *reinterpret_cast<const bf16x8*>(sp + (size_t)pp * 8 * sideld);
    }
    if constexpr ((EPI == 1 || EPI == 2) && HAS_RS) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) prs[pp] = a.rowscale[min(mrow + pp * 8, a.M - 1) / a.rows_per_sample];
    }
  };
  auto load_side_pass = [&](int i, int pp, bf16x8& sdp, float& prp) {
    const int mrow = m0w + i * 32 + rr + pp * 8;
    if (FULL || mrow < a.M) sdp = *reinterpret_cast<const bf16x8*>(sidep + (size_t)mrow * sideld + n_out);
    if constexpr (HAS_RS) prp = a.rowscale[min(mrow, a.M - 1) / a.rows_per_sample];
  };
  // fp32 residual rows (EPI 3): two 16-byte loads per pass, requested one pass ahead (rolling registers)
  const float* r32 = reinterpret_cast<const float*>(a.R);
  float* y32 = reinterpret_cast<float*>(a.Ypre);
  float4 rf[EPI == 3 ? 4 : 1][2];
  auto load_r32 = [&](int i, int pp) {
    const int mrow = m0w + i * 32 + rr + pp * 8;
    if (FULL || mrow < a.M) {
      const float* rp = r32 + (size_t)mrow * a.ldr + n_out;
      rf[pp][0] = *reinterpret_cast<const float4*>(rp);
      rf[pp][1] = *reinterpret_cast<const float4*>(rp + 4);
    }
  };
  constexpr bool PREF = EPI != 2 && EPI != 3;            // side rows a whole slab ahead; gelu' * aux / fp32 rows (register budget): pass by pass
  bf16x8 cur[4], sd[4], sdn[PREF ? 4 : 1];
  float prs[4] = {1.f, 1.f, 1.f, 1.f}, prsn[4] = {1.f, 1.f, 1.f, 1.f};
  load_side(0, sd, prs);
  if constexpr (EPI == 3) {
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) load_r32(0, pp);
  }
  stage(0);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) cur[pp] = read_pass(pp);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i + 1 < TM) {
      if constexpr (PREF) load_side(i + 1, sdn, prsn);
      asm volatile("" ::: "memory");
      stage(i + 1);                                      // queued behind the reads of slab i
      asm volatile("" ::: "memory");
    }
    const int mrow = m0w + i * 32 + rr;                  // this lane's first read-back row of the slab
    bf16* yp = a.Y + (size_t)mrow * a.ldy + n_out;
    bf16* prep = (EPI == 1 && a.Ypre) ? a.Ypre + (size_t)mrow * a.ldy + n_out : nullptr;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      if (FULL || mrow + pp * 8 < a.M) {
        bf16x8 v = cur[pp];
        if constexpr (EPI == 1) {
          if (prep) st_stream(reinterpret_cast<bf16x8*>(prep + (size_t)pp * 8 * a.ldy), v);
          v = gelu8(v, HAS_RS ? prs[pp] : 1.f);
        } else if constexpr (EPI == 2) {
          v = gelu_grad_mul8(v, sd[pp], HAS_RS ? prs[pp] : 1.f);
        }
        if constexpr (EPI == 3) {
          const float4 o0 = {bf2f(v[0]) + rf[pp][0].x, bf2f(v[1]) + rf[pp][0].y, bf2f(v[2]) + rf[pp][0].z, bf2f(v[3]) + rf[pp][0].w};
          const float4 o1 = {bf2f(v[4]) + rf[pp][1].x, bf2f(v[5]) + rf[pp][1].y, bf2f(v[6]) + rf[pp][1].z, bf2f(v[7]) + rf[pp][1].w};
          float* op = y32 + (size_t)(mrow + pp * 8) * a.ldy + n_out;
          *reinterpret_cast<float4*>(op) = o0;
          *reinterpret_cast<float4*>(op + 4) = o1;
          if (a.Y) {
            v[0] = f2bf(o0.x); v[1] = f2bf(o0.y); v[2] = f2bf(o0.z); v[3] = f2bf(o0.w);
            v[4] = f2bf(o1.x); v[5] = f2bf(o1.y); v[6] = f2bf(o1.z); v[7] = f2bf(o1.w);
            st_out(reinterpret_cast<bf16x8*>(yp + (size_t)pp * 8 * a.ldy), v, (a.act & 0x2000) != 0);
          }
        } else {
          if constexpr (HAS_R) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(sd[pp][e]));
          }
          st_out(reinterpret_cast<bf16x8*>(yp + (size_t)pp * 8 * a.ldy), v, (a.act & 0x2000) != 0);
        }
      }
      if constexpr (HAS_R || EPI == 2) __builtin_amdgcn_sched_barrier(0);     // one pass at a time (keeps the residual variants out of scratch)
      if (i + 1 < TM) {                                  // the register just drained takes the same pass of the next slab
        asm volatile("" ::: "memory");
        cur[pp] = read_pass(pp);
        if constexpr (EPI == 3) load_r32(i + 1, pp);
        else if constexpr (PREF) { sd[pp] = sdn[pp]; prs[pp] = prsn[pp]; }
        else load_side_pass(i + 1, pp, sd[pp], prs[pp]);  // rolling: the register just consumed takes the same pass of the next slab
      }
    }
  }
}

// bias slice of the tile -> LDS once per workgroup (read by the staging pass many barriers later)
template <int BN>
__device__ __forceinline__ void stage_bias(const GemmArgs& a, float* bias_s, int tn0) {
  if (a.bias && threadIdx.x < BN / 4) {
    const int n = min(tn0 + (int)threadIdx.x * 4, a.N - 4);
    *reinterpret_cast<float4*>(bias_s + threadIdx.x * 4) = *reinterpret_cast<const float4*>(a.bias + n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace
