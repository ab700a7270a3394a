// Input-side kernels of the FIBER fused path (gfx950): RoBERTa embeddings and Swin patch-embed im2col.
//
//   roberta_embed  pos = cumsum(ids != pad) * (ids != pad) + pad ; y = dropout(LN(word[ids] + type[0] + position[pos]))
//                  (roberta.py:169-199, 877-888).  Tables stay fp32 (the master parameters): only B*S rows are gathered.
//   backward       recomputes the pre-LN sum from the tables, LN backward in registers, then scatter-adds fp32 into
//                  the word / position / token-type gradient tables (atomics; only touched rows).
//   im2col         PatchEmbed's Conv2d(3->C, k=4, s=4) is a GEMM over non-overlapping patches (timm 0.4.12 PatchEmbed,
//                  used at swin_transformer.py:588): rows = patches, K = 48 ordered [c][kh][kw], zero padded to 64.
#include "common.h"

namespace {

template <int NV>
__global__ __launch_bounds__(256) void roberta_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                                const float* __restrict__ pos_tab, const float* __restrict__ type_tab,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                bf16* __restrict__ y, int* __restrict__ pos_out,
                                                                float* __restrict__ mean, float* __restrict__ rstd, int S, int C,
                                                                int pad, float eps, float p_drop, uint64_t seed,
                                                                const uint64_t* __restrict__ seed_base) {
  if (seed_base) seed += *seed_base;
  __shared__ int spos[1024];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int s = 0; s < S; ++s) {
      const int m = ids[(size_t)b * S + s] != pad;
      run += m;
      spos[s] = run * m + pad;
    }
  }
  __syncthreads();
  const int nvec = C >> 2;                       // float4 vectors
  const uint32_t thresh = (uint32_t)((double)p_drop * 4294967296.0);
  const float inv_keep = 1.f / (1.f - p_drop);
  for (int s = wave; s < S; s += 4) {
    const size_t row = (size_t)b * S + s;
    const int64_t id = ids[row];
    const int ps = spos[s];
    if (lane == 0) pos_out[row] = ps;
    float v[NV][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const float4 w = *reinterpret_cast<const float4*>(word + (size_t)id * C + vi * 4);
        const float4 pp = *reinterpret_cast<const float4*>(pos_tab + (size_t)ps * C + vi * 4);
        const float4 t = *reinterpret_cast<const float4*>(type_tab + vi * 4);
        v[i][0] = w.x + t.x + pp.x; v[i][1] = w.y + t.y + pp.y; v[i][2] = w.z + t.z + pp.z; v[i][3] = w.w + t.w + pp.w;
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
      } else { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
    }
    const float mu = wave_sum(sum) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nvec)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + vi * 4), bb = *reinterpret_cast<const float4*>(beta + vi * 4);
        const float gg[4] = {g.x, g.y, g.z, g.w}, be[4] = {bb.x, bb.y, bb.z, bb.w};
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float r = (v[i][e] - mu) * rs * gg[e] + be[e];
          if (p_drop > 0.f) r = drop_keep_e(drop_base(seed, (size_t)row * C + vi * 4), e, thresh) ? r * inv_keep : 0.f;
          o[e] = f2bf(r);
        }
        *reinterpret_cast<bf16x4*>(y + row * C + vi * 4) = o;
      }
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void roberta_embed_bwd_kernel(const bf16* __restrict__ dy, const int64_t* __restrict__ ids,
                                                                const int* __restrict__ pos, const float* __restrict__ word,
                                                                const float* __restrict__ pos_tab, const float* __restrict__ type_tab,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ dword,
                                                                float* __restrict__ dpos, float* __restrict__ dtype,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                                int C, int pad, float p_drop, uint64_t seed,
                                                                const uint64_t* __restrict__ seed_base) {
  if (seed_base) seed += *seed_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 2;
  const uint32_t thresh = (uint32_t)((double)p_drop * 4294967296.0);
  const float inv_keep = 1.f / (1.f - p_drop);
  float ag[NV][4], ab[NV][4], at[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) ag[i][e] = ab[i][e] = at[i][e] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const int64_t id = ids[row];
    const int ps = pos[row];
    const float mu = mean[row], rs = rstd[row];
    float xh[NV][4], dg[NV][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const float4 w = *reinterpret_cast<const float4*>(word + (size_t)id * C + vi * 4);
        const float4 pp = *reinterpret_cast<const float4*>(pos_tab + (size_t)ps * C + vi * 4);
        const float4 t = *reinterpret_cast<const float4*>(type_tab + vi * 4);
        const float4 g = *reinterpret_cast<const float4*>(gamma + vi * 4);
        const float x[4] = {w.x + t.x + pp.x, w.y + t.y + pp.y, w.z + t.z + pp.z, w.w + t.w + pp.w};
        const float gg[4] = {g.x, g.y, g.z, g.w};
        const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(dy + (size_t)row * C + vi * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = bf2f(d4[e]);
          if (p_drop > 0.f) d = drop_keep_e(drop_base(seed, (size_t)row * C + vi * 4), e, thresh) ? d * inv_keep : 0.f;
          xh[i][e] = (x[e] - mu) * rs;
          dg[i][e] = d * gg[e];
          s1 += dg[i][e];
          s2 += dg[i][e] * xh[i][e];
          ag[i][e] += d * xh[i][e];
          ab[i][e] += d;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) xh[i][e] = dg[i][e] = 0.f;
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float de = rs * (dg[i][e] - s1 - xh[i][e] * s2);
          // nn.Embedding(padding_idx=pad): the pad row of both tables never receives a gradient (roberta.py:146,163)
          if (id != pad) atomicAdd(dword + (size_t)id * C + vi * 4 + e, de);
          if (ps != pad) atomicAdd(dpos + (size_t)ps * C + vi * 4 + e, de);
          at[i][e] += de;
        }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(dgamma + vi * 4 + e, ag[i][e]);
        atomicAdd(dbeta + vi * 4 + e, ab[i][e]);
        atomicAdd(dtype + vi * 4 + e, at[i][e]);
      }
  }
}

// img fp32 [B,3,H,W] -> cols bf16 [B*(H/4)*(W/4), 64]; column = c*16 + kh*4 + kw for c<3, zeros for 48..63.
// PAIR: the 2B-sample batch of the one-pass MLM + ITM step, [img ; where(sel, img, alt)] (objectives.py:56-61 builds the ITM half with a
// python loop over samples; compute_mlm_itm_fused concatenates the halves), gathered straight from the two sources: sample b < B reads
// img[b], sample B + b reads sel[b] ? img[b] : alt[b] -- no torch.where / torch.cat pass over the fp32 images (3.2 GB of traffic at B = 256).
template <bool PAIR>
__global__ __launch_bounds__(256) void im2col4_kernel(const float* __restrict__ img, const float* __restrict__ alt, const unsigned char* __restrict__ sel,
                                                      bf16* __restrict__ cols, int B, int H, int W) {
  const int Hp = H >> 2, Wp = W >> 2;
  const size_t total = (size_t)(PAIR ? 2 * B : B) * Hp * Wp * 16;   // 16 groups of 4 columns per patch
  for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int grp = idx & 15;
    const size_t patch = idx >> 4;
    const int j = patch % Wp, i = (patch / Wp) % Hp;
    int b = patch / ((size_t)Wp * Hp);
    const float* src = img;
    if constexpr (PAIR) {
      if (b >= B) { b -= B; if (!sel[b]) src = alt; }
    }
    bf16x4 o;
    if (grp < 12) {
      const int c = grp >> 2, kh = grp & 3;
      const float4 v = *reinterpret_cast<const float4*>(src + (((size_t)b * 3 + c) * H + i * 4 + kh) * W + j * 4);
      o[0] = f2bf(v.x); o[1] = f2bf(v.y); o[2] = f2bf(v.z); o[3] = f2bf(v.w);
    } else {
      o[0] = o[1] = o[2] = o[3] = f2bf(0.f);
    }
    *reinterpret_cast<bf16x4*>(cols + patch * 64 + grp * 4) = o;
  }
}

}  // namespace

// ids int64 [B,S]; tables fp32; y bf16 [B*S, C]; pos_out int32 [B*S]; mean/rstd fp32 [B*S].  C % 4 == 0, C <= 2048, S <= 1024
extern "C" int fiber_roberta_embed_fwd(const int64_t* ids, const float* word, const float* pos_tab, const float* type_tab,
                                       const float* gamma, const float* beta, void* y, int* pos_out, float* mean, float* rstd,
                                       int B, int S, int C, int pad, float eps, float p_drop, uint64_t seed, const uint64_t* seed_base,
                                       hipStream_t stream) {
  if (B <= 0) return FIBER_OK;
  if ((C & 3) || S > 1024 || C > 2048) return FIBER_EINVAL;
  const int nv = cdiv(C >> 2, 64);
#define L(NV) hipLaunchKernelGGL((roberta_embed_fwd_kernel<NV>), dim3(B), dim3(256), 0, stream, ids, word, pos_tab, type_tab, gamma, beta, (bf16*)y, pos_out, mean, rstd, S, C, pad, eps, p_drop, seed, seed_base)
  if (nv <= 1) L(1); else if (nv <= 2) L(2); else if (nv <= 4) L(4); else L(8);
#undef L
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// Gradient tables dword/dpos/dtype and dgamma/dbeta are ACCUMULATED into (caller zero-initialises).
extern "C" int fiber_roberta_embed_bwd(const void* dy, const int64_t* ids, const int* pos, const float* word, const float* pos_tab,
                                       const float* type_tab, const float* gamma, const float* mean, const float* rstd,
                                       float* dword, float* dpos, float* dtype, float* dgamma, float* dbeta, int B, int S, int C,
                                       int pad, float p_drop, uint64_t seed, const uint64_t* seed_base, hipStream_t stream) {
  if (B <= 0) return FIBER_OK;
  if ((C & 3) || C > 2048) return FIBER_EINVAL;
  const int rows = B * S, nv = cdiv(C >> 2, 64);
  int grid = cdiv(rows, 4 * 4);
  grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
#define L(NV) hipLaunchKernelGGL((roberta_embed_bwd_kernel<NV>), dim3(grid), dim3(256), 0, stream, (const bf16*)dy, ids, pos, word, pos_tab, type_tab, gamma, mean, rstd, dword, dpos, dtype, dgamma, dbeta, rows, C, pad, p_drop, seed, seed_base)
  if (nv <= 1) L(1); else if (nv <= 2) L(2); else if (nv <= 4) L(4); else L(8);
#undef L
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// img fp32 [B,3,H,W] (H, W multiples of 4) -> cols bf16 [B*(H/4)*(W/4), 64]
extern "C" int fiber_im2col_patch4(const float* img, void* cols, int B, int H, int W, hipStream_t stream) {
  if ((H & 3) || (W & 3)) return FIBER_EINVAL;
  const size_t total = (size_t)B * (H / 4) * (W / 4) * 16;
  size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(im2col4_kernel<false>, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, stream, img, nullptr, nullptr, (bf16*)cols, B, H, W);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}

// img, alt fp32 [B,3,H,W]; sel uint8 [B] -> cols bf16 [2B*(H/4)*(W/4), 64] of the batch [img ; where(sel, img, alt)]
extern "C" int fiber_im2col_patch4_pair(const float* img, const float* alt, const unsigned char* sel, void* cols, int B, int H, int W,
                                        hipStream_t stream) {
  if ((H & 3) || (W & 3)) return FIBER_EINVAL;
  if (B <= 0) return FIBER_OK;
  const size_t total = (size_t)2 * B * (H / 4) * (W / 4) * 16;
  size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(im2col4_kernel<true>, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, stream, img, alt, sel, (bf16*)cols, B, H, W);
  FIBER_CHECK_LAUNCH();
  return FIBER_OK;
}
