/* fiber_hip.h -- C ABI of libfiber_hip.so, the MI355X (gfx950) kernels behind the FIBER coarse-grained
 * fused-backbone forward/backward path.
 *
 * The reference has no FFI: its hot path is Python calling ATen.  Each entry point below replaces the ATen call
 * sequence of one reference function (file:line relative to /root/reference/coarse_grained/fiber/modules/).
 * A maintainer binds them with ctypes (fiber_amd/lib.py is that binding).
 *
 * Conventions: every pointer is a DEVICE pointer; "bf16" buffers are passed as void* (uint16 storage, row-major);
 * fp32 is used for parameters of normalisation/bias, statistics and gradient accumulators; `stream` is a hipStream_t
 * (the caller's current stream -- kernels are stream-ordered, re-entrant, and never synchronise the host);
 * return value 0 = ok, 1 = invalid argument (shape/alignment contract violated), 2 = launch failure.
 * Inputs are borrowed and never mutated; outputs and workspaces are caller-allocated.
 */
#ifndef FIBER_HIP_H
#define FIBER_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ihipStream_t* fiber_stream_t;

/* nn.Linear (+bias, +exact-erf GELU, +per-sample DropPath scale, +residual) : Y = rowscale*act(X.W^T + bias) + R
 * replaces: swin_transformer.py:197,221,233,238,257 (qkv/proj/i2t linears), timm Mlp fc1/fc2 (:325), PatchMerging.reduction
 * (:431), roberta.py:231-241,337,398,415, fiber_module.py:349-350.  act: 0 none, 1 GELU (Ypre, if non-NULL, gets the
 * pre-activation).  Requires K%8==0, N%4==0, ldx/ldw%8==0, ldy/ldr%4==0. */
/* act 2 = fused GELU backward: Y = rowscale * (X.W^T) * gelu'(aux) with aux = saved pre-activation [M, ldaux];
 * colpart (nullable) receives per-row-tile column sums of Y, [ceil(M / fiber_gemm_row_tile(M,N,K)), N] fp32 (bias gradient). */
int fiber_gemm_nt_bf16(const void* X, const void* W, const float* bias, const void* residual, void* Y, void* Ypre,
                       const float* rowscale, int rows_per_sample, const void* aux, int ldaux, float* colpart, int M, int N,
                       int K, int ldx, int ldw, int ldy, int ldr, int act, fiber_stream_t stream);
/* act | 0x800 (act 0, residual required) = the fp32 RESIDUAL STREAM form of the residual epilogue (x = x + branch of
 * swin_transformer.py:388-391, dense + input_tensor of roberta.py:417-423,485 -- the reference keeps these sums in fp32):
 * `residual` is fp32 [M, ldr]; the sum is stored in fp32 at Ypre viewed as float [M, ldy] and, if Y is non-NULL, once more
 * rounded to bf16 at Y (the copy GEMM consumers of the stream read). */
int fiber_gemm_row_tile(int M, int N, int K);

/* Weight (+bias) gradient of nn.Linear: dW[N,K] (fp32, contiguous) = dY[M,lddy]^T . X[M,ldx]; dbias (nullable, fp32[N]) =
 * column sums of dY.  replaces the ATen addmm-backward (TN GEMM + sum(0)) behind every Linear the reference back-propagates
 * through: swin_transformer.py:197,221,233,238,257, timm Mlp (:325), PatchMerging.reduction (:431), PatchEmbed.proj,
 * roberta.py:231-241,337,398,415, fiber_module.py:349-350.  The M reduction is split inside the launch: workspace must hold
 * S*(N*K + N) floats when S = fiber_gemm_tn_splits(M,N,K) > 1 (NULL otherwise).  N%8==0, K%8==0, lddy%8==0, ldx%8==0.
 * DropPath backward (timm 0.4.12 DropPath on the branch, swin_transformer.py:390-391) folded in: row_mask (nullable, fp32
 * [M / rows_per_sample], the per-sample factors in {0, 1/keep}) makes the rows of dropped samples not contribute, `scale` (= 1/keep;
 * 1 without a mask) multiplies dW and dbias.  rows_per_sample % 64 == 0 when a mask is given. */
int fiber_gemm_tn_splits(int M, int N, int K);
int fiber_gemm_tn_bf16(const void* dY, const void* X, float* dW, float* dbias, float* workspace, int M, int N, int K, int lddy,
                       int ldx, const float* row_mask, int rows_per_sample, float scale, fiber_stream_t stream);
/* The same GEMM WITHOUT the fold of its M splits: with fiber_gemm_tn_splits(M,N,K) = S > 1 the S weight slabs (S*N*K floats) and the S
 * bias slabs (S*N floats) stay in `workspace`, dW / dbias are not written; with S = 1 it is the call above. */
int fiber_gemm_tn_slabs_bf16(const void* dY, const void* X, float* dW, float* dbias, float* workspace, int M, int N, int K, int lddy,
                       int ldx, const float* row_mask, int rows_per_sample, float scale, fiber_stream_t stream);
/* fiber_gemm_tn_bf16 with its output rows permuted on the way out: row n of dW (entry n of dbias) is written at row_map[n] (int32 [N], a
 * permutation) -- the gradient of a weight whose working copy has its rows in another order (the head-major qkv projection).  The
 * permutation is applied by the fold of the M splits, so `workspace` is needed for every S: max(S, 1) * (N*K + N) floats. */
int fiber_gemm_tn_rowmap_bf16(const void* dY, const void* X, float* dW, float* dbias, float* workspace, int M, int N, int K, int lddy,
                              int ldx, const float* row_mask, int rows_per_sample, float scale, const int* row_map,
                              fiber_stream_t stream);
/* Folds the slabs of MANY such GEMMs in one launch (ops.py defers them to the end of the backward pass).  table: device array of ndesc
 * 40-byte records {const float* ws; float* dW; float* dbias (NULL = none); int32 S, N, nk4 = N*K/4, block0}, block0 ascending from 0, a
 * record owns fiber_tn_fold_blocks(S, N, K, dbias != NULL) blocks; nblocks = their total.  Same summation order as fiber_gemm_tn_bf16's fold. */
int fiber_tn_fold_blocks(int S, int N, int K, int has_dbias);
int fiber_tn_fold_multi(const void* table, int ndesc, int nblocks, fiber_stream_t stream);

/* nn.LayerNorm over the last dim (C%8==0, C<=4096); saves mean/rstd.  replaces swin_transformer.py:362,391,244; roberta.py:485,422 */
int fiber_layernorm_fwd_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows,
                             int C, float eps, fiber_stream_t stream);
int fiber_layernorm_bwd_grid(int rows); /* workspace = grid*8*C floats */
/* dres (nullable): gradient arriving on the residual path of the same x; fused: dx = LN'(dy) + dres */
int fiber_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                             const void* dres, void* dx, float* dgamma, float* dbeta, float* workspace, int rows, int C,
                             fiber_stream_t stream);

/* The same on the fp32 residual stream.  flags bit 0: x is fp32 [rows, C] (the un-rounded x of swin_transformer.py:362,391 /
 * the LayerNorm input of roberta.py:485,422).  y32 (nullable): the output once more in fp32 -- RoBERTa is post-LN, its LayerNorm
 * output is the next residual.  Gradients (dy, dres, dx) stay bf16. */
int fiber_layernorm_fwd_stream(const void* x, const float* gamma, const float* beta, void* y, float* y32, float* mean, float* rstd,
                               int rows, int C, float eps, int flags, fiber_stream_t stream);
int fiber_layernorm_bwd_stream(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                               const void* dres, void* dx, float* dgamma, float* dbeta, float* workspace, int rows, int C, int flags,
                               fiber_stream_t stream);

/* PatchMerging gather+concat+LayerNorm (swin_transformer.py:411-430): x [B,H*W,C] -> y [B,H*W/4,4C] */
int fiber_patch_merge_ln_fwd_bf16(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B,
                                  int H, int W, int C, float eps, fiber_stream_t stream);
int fiber_patch_merge_ln_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  void* dx, float* dgamma, float* dbeta, float* workspace, int B, int H, int W, int C,
                                  fiber_stream_t stream);
/* flags bit 0: x is the fp32 residual stream [B,H*W,C]; y, dy, dx stay bf16 */
int fiber_patch_merge_ln_fwd_stream(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B,
                                    int H, int W, int C, float eps, int flags, fiber_stream_t stream);
int fiber_patch_merge_ln_bwd_stream(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                    void* dx, float* dgamma, float* dbeta, float* workspace, int B, int H, int W, int C, int flags,
                                    fiber_stream_t stream);

/* LayerNorm + Mlp + DropPath + residual of a Swin block as one kernel per direction (csrc/mlp_rows.hip), C in {128, 256}:
 *   y = x + rowscale[row / rows_per_sample] * (fc2(gelu(fc1(LN(x)))) + b2)
 * replaces norm2 -> timm Mlp -> drop_path -> residual add of SwinTransformerBlock.forward (swin_transformer.py:391; Mlp :325,
 * LayerNorm :320).  The LayerNorm's affine part arrives folded into fc1 (w1p = bf16(W1 diag(gamma)) [4C, C], b1p = b1 + W1 beta
 * [4C], fp32); w2p = FA(W2) [C, 4C]; FA(.) = bf16 copy whose K columns are stored in the order "bits 2 and 3 of the index
 * swapped".  The pre-activation stays on chip; g (optional, bf16 [M, 4C]) receives gelu(H) for the backward.  rowscale NULL = no
 * DropPath. */
int fiber_ln_mlp_fwd_bf16(const void* x, const void* w1p, const float* b1p, const void* w2p, const float* b2, const float* rowscale,
                          void* y, void* g, int M, int C, int rows_per_sample, float eps, fiber_stream_t stream);
/* backward from x and dy (the pre-activation is recomputed): dx [M, C] = dy + LayerNorm-backward of the branch gradient, plus the
 * operands of the fc1 weight-gradient GEMM (fiber_gemm_tn_bf16): dh = rowscale (dy W2) gelu'(H) [M, 4C], xhat = (x - mean) rstd
 * [M, C], bf16.  w2tp = bf16(W2^T) [4C, C], w1tp = FA((W1 diag(gamma))^T) [C, 4C]. */
int fiber_ln_mlp_bwd_bf16(const void* x, const void* dy, const void* w1p, const float* b1p, const void* w2tp, const void* w1tp,
                          const float* rowscale, void* dx, void* dh, void* xhat, int M, int C, int rows_per_sample, float eps,
                          fiber_stream_t stream);

/* Swin (shifted) window attention in image-token order: roll + window_partition + WindowAttention self-attn core +
 * window_reverse + roll (swin_transformer.py:99-126, 195-219, 364-387, mask 327-350).  qkv [B*H*W,3C] -> o [B*H*W,C]. */
/* head_major: 0 = reference channel layout [3][heads][32]; 1 = [heads][3][32] (qkv weight rows permuted by the caller);
 * 2 = [heads][q | k v] line layout; 3, 4 = planar probe layouts, 12 x 12 windows only (anything else -> FIBER_EINVAL).
 * lse: B*Hres*Wres*heads fp32 values saved by the forward for the backward of the SAME call shape; its layout is private
 * ([image][head][token] for windows of <= 336 tokens, [token][head] above) -- no other consumer may interpret it. */
int fiber_window_attn_fwd_bf16(const void* qkv, const float* bias_table, void* o, float* lse, int B, int Hres, int Wres, int C,
                               int heads, int ws, int shift, int head_major, fiber_stream_t stream);
int fiber_window_attn_bwd_slices(int n_windows, int heads);
int fiber_window_attn_colsum_rows(int n_windows, int heads, int ws);   /* rows of colsum_ws; 0 = not available for this ws */
/* dqkv_colsum (optional, fp32[3C]) = column sums of dqkv = bias gradient of the qkv nn.Linear (swin_transformer.py:197),
 * produced by the two passes themselves; colsum_ws fp32[colsum_rows*3C].  Pass both or neither (NULL). */
int fiber_window_attn_bwd_bf16(const void* qkv, const float* bias_table, const void* o, const void* dout, const float* lse,
                               void* dqkv, float* dbias_table, float* delta_ws, float* dbias_ws, float* dqkv_colsum,
                               float* colsum_ws, int B, int Hres, int Wres, int C, int heads, int ws, int shift, int head_major,
                               fiber_stream_t stream);

/* Generic MHA core softmax(q.k^T*scale + kmask).v with optional attention-prob dropout: RoBERTa self-attention
 * (roberta.py:256-326), image->text cross-attention (swin_transformer.py:226-256) and text->image cross-attention
 * (roberta.py:272-276).  D in {32,64}.
 * Every dropout-capable entry point takes the 64-bit key as `seed` + an optional DEVICE pointer `seed_base` (key = seed +
 * *seed_base; NULL = seed alone): a captured hipGraph keeps the per-call-site part by value and re-reads the per-step part from
 * device memory on every replay. */
int fiber_mha_fwd_bf16(const void* q, const void* k, const void* v, const float* kmask, void* o, float* lse, int B, int heads,
                       int Lq, int Lk, int D, int ldq, int ldk, int ldv, int ldo, float scale, float p_drop, uint64_t seed,
                       const uint64_t* seed_base, fiber_stream_t stream);
int fiber_mha_bwd_bf16(const void* q, const void* k, const void* v, const float* kmask, const void* o, const void* dout,
                       const float* lse, void* dq, void* dk, void* dv, float* delta_ws, int B, int heads, int Lq, int Lk, int D,
                       int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale, float p_drop,
                       uint64_t seed, const uint64_t* seed_base, fiber_stream_t stream);

/* RobertaEmbeddings.forward (roberta.py:169-199, 877-888) and its backward (scatter-add into fp32 gradient tables) */
int fiber_roberta_embed_fwd(const int64_t* ids, const float* word, const float* pos_tab, const float* type_tab,
                            const float* gamma, const float* beta, void* y, int* pos_out, float* mean, float* rstd, int B, int S,
                            int C, int pad, float eps, float p_drop, uint64_t seed, const uint64_t* seed_base, fiber_stream_t stream);
int fiber_roberta_embed_bwd(const void* dy, const int64_t* ids, const int* pos, const float* word, const float* pos_tab,
                            const float* type_tab, const float* gamma, const float* mean, const float* rstd, float* dword,
                            float* dpos, float* dtype, float* dgamma, float* dbeta, int B, int S, int C, int pad, float p_drop,
                            uint64_t seed, const uint64_t* seed_base, fiber_stream_t stream);

/* timm PatchEmbed Conv2d(3->C,k=4,s=4) as im2col (K=48 ordered [c][kh][kw], zero padded to 64) feeding fiber_gemm_nt_bf16 */
int fiber_im2col_patch4(const float* img, void* cols, int B, int H, int W, fiber_stream_t stream);
/* ... of the 2B-sample batch [img ; where(sel, img, alt)] that the one-pass MLM + ITM step feeds the backbone (objectives.py:56-61 draws
 * the ITM images, :17-75 are the two passes), gathered from the two fp32 sources [B,3,H,W]; sel uint8 [B]. cols bf16 [2B*(H/4)*(W/4), 64]. */
int fiber_im2col_patch4_pair(const float* img, const float* alt, const unsigned char* sel, void* cols, int B, int H, int W,
                             fiber_stream_t stream);

/* element-wise / reduction helpers (n % 8 == 0) */
int fiber_gelu_bwd_bf16(const void* dgelu, const void* h_pre, void* dh, long n, fiber_stream_t stream);
int fiber_gelu_bwd_colsum_bf16(const void* dgelu, const void* h_pre, void* dh, float* db, float* workspace, int M, int N,
                               fiber_stream_t stream);
int fiber_scale_add_bf16(const void* a, const void* b, const float* alpha, float mult, void* out, long n, fiber_stream_t stream);
int fiber_dot_bf16(const void* a, const void* b, float* out, long n, fiber_stream_t stream);
int fiber_colsum_slabs(int M, int N); /* workspace = slabs*N floats when slabs > 1 */
int fiber_colsum_bf16(const void* x, float* out, float* workspace, int M, int N, int ld, fiber_stream_t stream);
int fiber_fold_rows_f32(const float* part, float* out, int rows, int N, fiber_stream_t stream);
int fiber_dropout_bf16(const void* x, void* y, long n, float p, uint64_t seed, const uint64_t* seed_base, fiber_stream_t stream);
/* timm 0.4.12 DropPath factors (swin_transformer.py:322,390-391): out[b] = floor(keep + U_b) / keep, fp32 [n] */
int fiber_droppath_scale_f32(float* out, int n, float keep, uint64_t seed, const uint64_t* seed_base, fiber_stream_t stream);
int fiber_rowscale_add_bf16(const void* r, const void* x, const float* scale, void* out, long n, long per_sample,
                            fiber_stream_t stream);
/* Everything the reference does between a branch and its residual, in one pass:
 *   out = res + rowscale[sample] * (drop_a(a) + alpha * drop_b(b))
 * = swin_transformer.py:259 (x + alpha_i2t * y) followed by :390 (shortcut + drop_path(x)); roberta.py:339 / :420 (dense ->
 * dropout -> + input_tensor) and :483-485 (alpha_t2i * cross + self, + hidden_states).  res_kind: 0 none, 1 bf16, 2 fp32 (the fp32
 * residual stream); b, alpha, rowscale nullable; p_a / p_b dropout probabilities (0 = off) with keys seed_a / seed_b (+ *seed_base);
 * out32 (fp32) and / or out16 (bf16), at least one.  n % 8 == 0; per_sample % 8 == 0 when rowscale is given. */
int fiber_stream_add(const void* res, int res_kind, const void* a, const void* b, const float* alpha, const float* rowscale,
                     long per_sample, float p_a, uint64_t seed_a, float p_b, uint64_t seed_b, const uint64_t* seed_base,
                     float* out32, void* out16, long n, fiber_stream_t stream);
/* branch gradients of the above: da, db (nullable), dalpha (nullable fp32 scalar, accumulated: zero it first; needs b) */
int fiber_stream_add_bwd(const void* dy, const void* b, const float* alpha, const float* rowscale, long per_sample, float p_a,
                         uint64_t seed_a, float p_b, uint64_t seed_b, const uint64_t* seed_base, void* da, void* db, float* dalpha,
                         long n, fiber_stream_t stream);
/* y (bf16) = x (fp32): the bf16 copy of an fp32 stream tensor for a GEMM that consumes it */
int fiber_cast_f32_bf16(const float* x, void* y, long n, fiber_stream_t stream);
/* y = scale[row / rows_per_sample] * x (DropPath backward, swin_transformer.py:390-391) and db = column sums of y (bias
 * gradient of the proj / fc2 linear) in one pass; workspace as for fiber_gelu_bwd_colsum_bf16 */
int fiber_rowscale_colsum_bf16(const void* x, const float* scale, void* y, float* db, float* workspace, int M, int N,
                               int rows_per_sample, fiber_stream_t stream);

/* Cross-entropy over the vocabulary for bf16 logits (caller side: compute_mlm's F.cross_entropy(..., ignore_index=-100),
 * objectives.py:24-28).  fwd: loss[r] = logsumexp(x[r,:]) - x[r, labels[r]] (0 on ignored rows), lse saved; bwd: dlogits =
 * (softmax - onehot) * scale[0] with scale a device scalar (upstream gradient / number of valid rows). */
int fiber_ce_fwd_bf16(const void* logits, const long long* labels, float* loss, float* lse, int* pred, int rows, int V,
                      long long ignore_index, fiber_stream_t stream);
int fiber_ce_bwd_bf16(const void* logits, const long long* labels, const float* lse, const float* scale, void* dlogits, int rows,
                      int V, long long ignore_index, fiber_stream_t stream);
/* pred (nullable, int32 [rows]) of fiber_ce_fwd_bf16: arg max of every labelled row, smallest index among equal maxima as torch.argmax has
 * it, -1 on ignored rows -- the MLM accuracy of the step (gadgets/my_metrics.py:5-28 via fiber_utils.py) without an argmax pass over all rows.
 * fiber_colsum_labelled_bf16: the decoder's bias gradient, column sums of dlogits over the labelled rows only (fiber_ce_bwd_bf16 wrote zeros into
 * the others; heads.py:47-62 MLMHead.bias); workspace fp32 [fiber_colsum_labelled_slabs(rows) * V]. */
int fiber_colsum_labelled_slabs(int rows);
int fiber_colsum_labelled_bf16(const void* x, const long long* labels, float* out, float* workspace, int rows, int V,
                               long long ignore_index, fiber_stream_t stream);

/* AdamW step of one parameter group in one launch (caller side of the path: transformers 4.6.0 AdamW(correct_bias=True) as
 * configured by fiber_utils.set_schedule, fiber_utils.py:248-252), also refreshing the bf16 working copies of the weights.
 * table: int64[n*5] device pointers (param fp32, grad fp32, exp_avg, exp_avg_sq, bf16 copy or 0); numel: int64[n];
 * chunks: int32[nchunks*2] (tensor, chunk) pairs of fiber_adamw_chunk() elements; all three in device memory; step >= 1.
 * hyper (nullable): device float[2] {lr, lr*sqrt(1-beta2^step)/(1-beta1^step)} read instead of the by-value lr / step (hipGraph). */
int fiber_adamw_chunk(void);
int fiber_adamw_multi_f32(const long long* table, const long long* numel, const int* chunks, int nchunks, float lr,
                          float weight_decay, float beta1, float beta2, float eps, int step, const float* hyper,
                          fiber_stream_t stream);
/* All transposed bf16 working copies of the linear weights (the W^T operand of every dX GEMM) in one launch, after the optimizer
 * step -- caller side, next to fiber_adamw_multi_f32 (which rewrites the plain bf16 copies).  table: device array of ndesc 32-byte
 * records {const bf16* src [N,K]; bf16* dst [K,N]; int32 N, K, tile0, tiles_k}; tile0 ascending from 0, a weight owns
 * ceil(N/64)*ceil(K/64) tiles, tiles_k = ceil(K/64); ntiles = the total.  N % 8 == K % 8 == 0. */
int fiber_transpose_multi_bf16(const void* table, int ndesc, int ntiles, fiber_stream_t stream);
/* Row-permuted bf16 working copies (+ their transposes, + the permuted fp32 bias) of fp32 weights in one launch after the optimizer step:
 * the head-major qkv projections of the window-attention blocks (ops._LinearQKVHeadMajor; the permutation is this implementation's, the
 * weights are WindowAttention.qkv, swin_transformer.py:197).  table: device array of ndesc 64-byte records {const float* src [N,K];
 * const int32* perm [N]; bf16* dst [N,K]; bf16* dst_t [K,N] or null; const float* bias [N] or null; float* bias_dst [N]; int32 N, K, tile0,
 * tiles_k}: dst[n] = bf16(src[perm[n]]), dst_t = dst^T, bias_dst[n] = bias[perm[n]]; tiles as in fiber_transpose_multi_bf16. */
int fiber_rowperm_cast_multi_bf16(const void* table, int ndesc, int ntiles, fiber_stream_t stream);
/* On-device input pipeline.
 * fiber_resize_bicubic_norm_u8 replaces transforms/transform.py:10-17 `albef_transform`: torchvision Resize((S,S), BICUBIC) on a
 * PIL RGB image (= Pillow ImagingResample: anti-aliased separable bicubic, 22-bit fixed-point coefficients, 8-bit rounding after
 * each pass) + ToTensor + Normalize, bit-identical to PIL.  descs: device array of n records {int64 src; int32 H, W, src_stride,
 * ksize_h, ksize_v, tmp_off, coef_h_off, coef_v_off} (40 bytes; src = uint8 [H][W][3], ksize_* = fiber_resample_ksize(in, S),
 * offsets into the int32 workspace `coef` (S*(2+ksize) ints per axis and image) and the byte workspace `tmp` (H*S*3 bytes per
 * image)); out: fp32 [n,3,S,S]; max_h = tallest source; mean / std: HOST pointers to 3 floats.
 * fiber_mlm_mask_i64 replaces datamodule_base.py:52 DataCollatorForLanguageModeling.mask_tokens (transformers 4.6.0): tokens with
 * id outside [special_lo, special_hi] are selected with probability p_select / 2^32; selected tokens become labels (others -100)
 * and are replaced by mask_id (80 %), a random id < vocab (10 %) or kept (10 %); draws = counter-based hash of (seed, index). */
int fiber_resample_ksize(int in_size, int out_size);
int fiber_resize_bicubic_norm_u8(const void* descs, int n, int* coef, void* tmp, float* out, int S, int max_h, const float* mean,
                                 const float* std, fiber_stream_t stream);
int fiber_mlm_mask_i64(const long long* ids, long long* ids_mlm, long long* labels, long n, unsigned long long seed,
                       unsigned p_select, int mask_id, int vocab, int special_lo, int special_hi, fiber_stream_t stream);

/* Modulated deformable convolution (DCNv2) of the fine-grained model's DyHead: the sampling half of
 * layers/deform_conv.py:300-353 `ModulatedDeformConv` (csrc/cuda/deform_conv_kernel_cuda.cu:578-640 im2col, :643-700 col2im,
 * :703-773 col2im_coord; host loop csrc/cuda/deform_conv_cuda.cu:497-692).  Channels-last: x bf16 [B,H,W,C], cols / dcols bf16
 * [B*Ho*Wo, kh*kw*C] (tap-major, tap = i*kw + j), offset fp32 [B*Ho*Wo, 2*kh*kw] ((dy,dx) per tap) or NULL, mask fp32
 * [B*Ho*Wo, kh*kw] (after the sigmoid) or NULL; both NULL = ordinary im2col.  The convolution itself is fiber_gemm_nt_bf16 on
 * cols (weights reordered to [Cout, kh*kw*C]); its gradients are fiber_gemm_nt_bf16 / fiber_gemm_tn_bf16.
 * fiber_dcn_scatter_bf16: dx fp32 [B,H,W,C] is accumulated into (caller zeroes; NULL skips), doffset / dmask are written (NULL skips).
 * groups = deformable_groups = 1, dilation 1, C % 8 == 0. */
int fiber_dcn_gather_bf16(const void* x, const float* offset, const float* mask, void* cols, int B, int H, int W, int C, int Ho,
                          int Wo, int kh, int kw, int stride, int pad, fiber_stream_t stream);
int fiber_dcn_scatter_bf16(const void* dcols, const void* x, const float* offset, const float* mask, float* dx, float* doffset,
                           float* dmask, int B, int H, int W, int C, int Ho, int Wo, int kh, int kw, int stride, int pad,
                           fiber_stream_t stream);
/* Input gradient without device atomics (3x3, stride 1 or 2, pad 1, C % 16 == 0): per-tile fixed-point LDS accumulation + a
 * window sum.  dx bf16 [B,H,W,C] is written; workspace = fiber_dcn_dx_workspace() 4-byte words whose LAST B*H*W*C + 4 must be
 * zero on entry. */
long fiber_dcn_dx_workspace(int B, int H, int W, int C, int Ho, int Wo, int stride);
int fiber_dcn_dx_bf16(const void* dcols, const float* offset, const float* mask, void* dx, float* workspace, int B, int H, int W,
                      int C, int Ho, int Wo, int kh, int kw, int stride, int pad, fiber_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
