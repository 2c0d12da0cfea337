/*
 * vq3_hip.h - C ABI of libvq3hip.so: the MI355X (gfx950) kernels behind the VGGT -> Perceiver -> Qwen3
 * forward/backward path of Sycamorers/vggt-qwen3.
 *
 * The reference has no FFI of its own: its hot path is stock PyTorch modules called from
 * src/models/vggt_qwen3_vlm.py:179-201 (VGGTQwen3VLM.forward). Each entry point below names the reference
 * (or third-party) Python op it replaces, so a maintainer can bind it from the reference side with ctypes
 * (see INTEGRATION.md). Conventions:
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; sizes are element counts;
 *   - bf16 tensors are raw uint16 (round-to-nearest-even), "f32" tensors are IEEE float;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work, they never
 *     synchronise, allocate or free, so they are hipGraph-capturable;
 *   - return value 0 = enqueued, non-zero = argument/launch error, message via vq3_last_error().
 */
#ifndef VQ3_HIP_H
#define VQ3_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQ3_ABI_VERSION 1

int vq3_abi_version(void);
const char* vq3_last_error(void);
/* Name of the code object's target ("gfx950"). */
const char* vq3_target_arch(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM  C = epilogue(alpha * A[M,K] . B[N,K]^T)     (bf16 in, f32 MFMA accumulate)
 * Replaces torch.nn.functional.linear on every projection of the path:
 *   Qwen3 q/k/v/o_proj, gate/up/down_proj, lm_head  (transformers/models/qwen3/modeling_qwen3.py:81-83,241-280,495)
 *   Perceiver in_proj/out_proj/MHA projections/FFN   (src/models/projector_perceiver.py:31-42,59-67)
 *   VGGT aggregator qkv/proj/fc1/fc2/patch-embed      (vggt.models.aggregator, un-vendored)
 * and, with operands swapped or transposed copies, all dgrad / wgrad / attention products.
 * Epilogue order: v = alpha*acc; v += bias[n]; (round to bf16 if C is bf16); v = act(v); v *= colscale[n];
 *                 v += R[m,n]; v += C[m,n] if accumulate; store.
 * Batches: blockIdx.z = b1*nb2 + b2; A += b1*sA1 + b2*sA2; B += b1*sB1 + (b2/b2divB)*sB2; C,R alike (no div).
 * General form: C[m,n] = epilogue(alpha * sum_k opA(A)[m,k] * opB(B)[n,k]); transA/transB select k-major operands, so
 * dgrad (dX = dY . W), wgrad (dW = dY^T . X) and P.V products need no transposed copies.
 * Requirements: K % 8 == 0 (K % 64 == 0 takes the fastest path), lda/ldb % 8 == 0, A/B 16-byte aligned, and the
 * M (resp. N) extent of a k-major A (resp. B) a multiple of 8.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vq3_gemm_desc {
  const void* A;        /* bf16 [M, lda] */
  const void* B;        /* bf16 [N, ldb] */
  void* C;              /* bf16 or f32 [M, ldc] */
  const void* bias;     /* f32 [N] or NULL */
  const void* colscale; /* f32 [N] or NULL */
  const void* R;        /* residual, dtype of C, [M, ldr] or NULL */
  int32_t M, N, K, lda, ldb, ldc, ldr;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
  int32_t nb1, nb2, b2divB;
  int32_t act;        /* 0 none, 1 GELU(erf), 2 SiLU */
  int32_t out_f32;    /* 0: C/R bf16, 1: C/R f32 */
  int32_t accumulate; /* C += */
  float alpha;
  int32_t transA;     /* 0: A is [M, lda] (contraction contiguous); 1: A is [K, lda] ("k-major", element (m,k) at k*lda+m) */
  int32_t transB;     /* 0: B is [N, ldb]; 1: B is [K, ldb] */
  int32_t ksplit;     /* > 1: split the contraction over this many workgroup slices; C must be f32, zero-initialised,
                         no epilogue (partials meet by f32 atomics) - for tiny-M / huge-K shapes (lm_head dgrad) */
} vq3_gemm_desc;

int vq3_gemm_bf16_nt(const vq3_gemm_desc* desc, void* stream);

/* The qkv projection of a VGGT / DINOv2 attention block with its consumer fused into the epilogue (layers/attention.py of the
 * published model, reached from vggt_qwen3_vlm.py:144): qkv = x Wqkv^T + b is never written; the tile leaves as head-major
 * Q, K, V bf16 [M / N, NH, N, 64] with, on q and k, the per-head LayerNorm(64) (use_norm: the aggregator's q_norm / k_norm) and the
 * 2-D rotate-half RoPE (use_rope; token t of a frame of `tokens_per_frame` tokens sits at (y, x) = ((t - patch_start) / Wp + 1,
 * (t - patch_start) % Wp + 1), special tokens at (0, 0); cos / sin bf16 [maxpos + 1, 32]) - same arithmetic and bf16 rounding
 * points as vq3_vit_qkprep on the materialised qkv. desc: plain NT bf16 GEMM (bias allowed), N = 3 * NH * 64, K % 64 == 0, C ignored. */
typedef struct vq3_vit_qkv_epilogue {
  void *Q, *K, *V;
  const float *qn_w, *qn_b, *kn_w, *kn_b;
  const void *cos, *sin;
  int32_t N, NH, tokens_per_frame, patch_start, Wp, use_norm, use_rope;
  float eps;
} vq3_vit_qkv_epilogue;
int vq3_gemm_vit_qkv(const vq3_gemm_desc* desc, const vq3_vit_qkv_epilogue* epi, void* stream);

/* LayerNorm folded into the GEMMs either side of it (the pre-norm blocks of the published VGGT / DINOv2 towers, reached from
 * vggt_qwen3_vlm.py:144: x -> LayerNorm -> Linear). With B = gamma o W, bias = b + W . beta and colsum[n] = sum_k B[n,k]
 *   Linear(LayerNorm(x))[m,n] = rstd_m * ((x . B^T)[m,n] - mu_m * colsum[n]) + bias[n],
 * so the consuming GEMM reads the RAW rows x (desc.A) and applies the row statistics in its epilogue: the normalised activation is
 * never written or read. stats_in: f32 [M, parts_in, 2] = (sum, sum of squares) of x's row over parts_in column groups (K = the
 * normalised width). stats_out: the producing GEMM (the residual add that forms x) leaves the same pairs for ITS output rows,
 * f32 [M, N / 128, 2] over groups of 128 stored bf16 columns - every slot written by exactly one thread (no atomics, no zeroing,
 * bit-identical from run to run). vq3_rowstats128 computes them for rows no GEMM produced. Needs the bf16 whole-row epilogue:
 * bf16 C, one batch, 16-byte aligned rows. */
typedef struct vq3_gemm_ln_fold {
  const float* stats_in; /* or NULL */
  int32_t parts_in;
  float eps;
  const float* colsum;   /* f32 [N], 16-byte aligned (with stats_in) */
  float* stats_out;      /* or NULL; needs N % 128 == 0 */
} vq3_gemm_ln_fold;
int vq3_gemm_bf16_nt_ln(const vq3_gemm_desc* desc, const vq3_gemm_ln_fold* ln, void* stream);
int vq3_gemm_vit_qkv_ln(const vq3_gemm_desc* desc, const vq3_vit_qkv_epilogue* epi, const vq3_gemm_ln_fold* ln, void* stream);
int vq3_rowstats128(const void* x_bf16, float* stats, int64_t rows, int32_t cols, void* stream);

/* The down_proj input-gradient GEMM with the SwiGLU backward in its epilogue (autograd of modeling_qwen3.py:81-83): desc computes
 * d(act) [M, N] = dY . W (any operand layout, no other epilogue, C ignored); instead of being written it is combined with the saved
 * pre-activations gu bf16 [M, 2N] = gate | up into dgu bf16 [M, 2N] = d(gate) | d(up), d(gate) = d(act) * up * silu'(gate),
 * d(up) = d(act) * silu(gate) - same arithmetic and rounding as vq3_silu_mul_bwd on the materialised d(act). */
int vq3_gemm_swiglu_bwd(const vq3_gemm_desc* desc, const void* gu, void* dgu, void* stream);
/* The gate|up projection with the SwiGLU forward in its epilogue (Qwen3MLP.forward, modeling_qwen3.py:81-83): desc is the plain NT GEMM
 * x [M, K] . W_gate|up [2 I, K]^T (N = 2 I, I % 128 == 0, K % 64 == 0, no bias / residual / batch); every tile multiplies a block of gate
 * rows of W and the same block of up rows, so the epilogue holds both halves of a feature: gu bf16 [M, 2 I] = gate | up is written as the
 * backward reads it, and desc->C / ldc receive act bf16 [M, I] = bf16(bf16(silu(gate)) * up) - same arithmetic and rounding as
 * vq3_silu_mul_fwd on the materialised gu. gu == NULL: only act is written (a forward that no backward follows: eval / no_grad). */
int vq3_gemm_swiglu_fwd(const vq3_gemm_desc* desc, void* gu, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Normalisation
 * ---------------------------------------------------------------------------------------------------------- */
/* Qwen3RMSNorm.forward (modeling_qwen3.py:59-64): y = w * bf16(x * rsqrt(mean(x^2) + eps)).
 * x,y bf16 [rows, cols] (row stride = ldx / ldy elements), w bf16 [cols]; rstd f32 [rows] (optional, for bwd). */
int vq3_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int32_t cols, int64_t ldx,
                    int64_t ldy, float eps, void* stream);
/* Backward of the above: dx = [dres +] rstd * (g - xhat * mean(g * xhat)), g = dy*w, xhat = x*rstd;
 * dw_part f32 [ceil(rows/4), cols] receives one partial row of sum(dy * xhat) per workgroup (plain stores, every row
 * written); reduce it with vq3_colsum_f32_to_bf16. dres may be NULL; dx may alias dres. */
int vq3_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                    float* dw_part, int64_t rows, int32_t cols, float eps, void* stream);
/* The same with rows_per_part rows (a multiple of 4, 4..256) per workgroup and partial row: dw_part f32 [ceil(rows/rows_per_part), cols]. */
int vq3_rmsnorm_bwd_rows(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                         float* dw_part, int64_t rows, int32_t cols, int32_t rows_per_part, void* stream);
/* out_bf16[c] (+)= sum_{r < nrows} part[r*cols + c]: column sum of a partial slab into a (bf16) gradient vector. */
int vq3_colsum_f32_to_bf16(const float* part, int32_t nrows, int32_t cols, void* out_bf16, int32_t accumulate,
                           void* stream);
/* Up to 8 of the above in one launch (a decoder layer's four norm-weight gradients). `jobs` is a HOST array. */
typedef struct vq3_colsum_job {
  const float* part;
  void* out_bf16;
  int32_t nrows, cols, accumulate;
} vq3_colsum_job;
int vq3_colsum_multi(const vq3_colsum_job* jobs, int32_t njobs, void* stream);

/* torch.nn.LayerNorm (projector_perceiver.py:39-40; VGGT Block.norm1/norm2, q_norm/k_norm).
 * x: bf16 (x_f32=0) or f32 (x_f32=1) [rows, cols]; w,b f32 [cols]; y_bf16 and/or y_f32 may be NULL.
 * Optional residual: the kernel normalises (x + res) where res has the dtype of x (Perceiver post-norm). */
int vq3_layernorm_fwd(const void* x, const void* res, int32_t x_f32, const float* w, const float* b, void* y_bf16,
                      float* y_f32, int64_t rows, int32_t cols, float eps, void* stream);

/* Backward of torch.nn.LayerNorm over f32 rows (the Perceiver's post-norm LayerNorms, src/models/projector_perceiver.py:39-40,48-50,
 * when the projector is trained - VisionLanguageConfig.train_projector; the reference's @torch.no_grad() never reaches it):
 * x (+ res, when non-NULL: the forward's optional residual) f32 [rows, cols] = the tensor that was normalised, dy f32 =
 * d(loss)/d(output), w f32 [cols].
 * dx f32 [rows, cols]; dw_part, db_part f32 [ceil(rows / 16), cols]: one partial row per workgroup of sum(dy * xhat) / sum(dy)
 * (plain stores, every slot written) - reduce them with vq3_colsum_f32. Statistics are recomputed from x. cols <= 4096, % 4 == 0. */
int vq3_layernorm_bwd(const float* dy, const float* x, const float* res, const float* w, float* dx, float* dw_part, float* db_part,
                      int64_t rows, int32_t cols, float eps, void* stream);
/* out_f32[c] (+)= sum_{r < nrows} part[r*cols + c] (f32 twin of vq3_colsum_f32_to_bf16). */
int vq3_colsum_f32(const float* part, int32_t nrows, int32_t cols, float* out_f32, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Element-wise / data movement
 * ---------------------------------------------------------------------------------------------------------- */
/* Backward of the exact (erf) GELU of projector_perceiver.py:36 on the bf16 pre-activation z the forward GEMM produced:
 * dz = dh * (Phi(z) + z * phi(z)); dh, z, dz bf16 [n], n % 8 == 0. */
int vq3_gelu_bwd(const void* dh, const void* z, void* dz, int64_t n, void* stream);
/* h = bf16(gelu_erf(z)) on a materialised bf16 pre-activation z [n] (n % 8 == 0): the activation the GEMM epilogue applies when it
 * is fused (act = 1), as a pass of its own for the trained-projector path, which keeps z for vq3_gelu_bwd. */
int vq3_gelu_fwd(const void* z, void* h, int64_t n, void* stream);
/* Qwen3MLP (modeling_qwen3.py:81-83): act[m, i] = silu(gu[m, i]) * gu[m, I + i]; gu bf16 [rows, 2I]. */
int vq3_silu_mul_fwd(const void* gu, void* act, int64_t rows, int32_t inter, void* stream);
/* dgu[m, i] = dact * up * silu'(gate); dgu[m, I+i] = dact * silu(gate). */
int vq3_silu_mul_bwd(const void* dact, const void* gu, void* dgu, int64_t rows, int32_t inter, void* stream);

/* Batched 2-D transpose of bf16 tiles with zero padding:
 * for batch (i0,i1,i2): dst[c * ldd + r] = src[r * lds + c], r < R, c < C; dst columns R..Rpad-1 are zeroed.
 * src += i0*s0 + i1*s1 + i2*s2; dst += i0*d0 + i1*d1 + i2*d2. */
int vq3_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int32_t Rpad, int64_t lds, int64_t ldd,
                       int32_t n0, int32_t n1, int32_t n2, int64_t s0, int64_t s1, int64_t s2, int64_t d0, int64_t d1,
                       int64_t d2, void* stream);

/* y = cast(x): f32 -> bf16 (dir 0) or bf16 -> f32 (dir 1), n elements. */
int vq3_cast(const void* x, void* y, int64_t n, int32_t dir, void* stream);
/* acc_bf16[i] (+)= src_f32[i]  (accumulate != 0 adds to the existing bf16 value in f32, rounds once). */
int vq3_f32_to_bf16_acc(const float* src, void* acc_bf16, int64_t n, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Qwen3 attention (modeling_qwen3.py:104-170 RoPE, :237-253 q/k-norm, :185-207 eager attention)
 * ---------------------------------------------------------------------------------------------------------- */
/* qkv bf16 [B*L, (Hq+2*Hkv)*D] (fused q|k|v projection output) ->
 *   Q [B,Hq,L,D], K [B,Hkv,L,D], V [B,Hkv,L,D]  (bf16), with per-head RMSNorm(q_w / k_w, eps) and rotate-half RoPE
 *   using cos/sin bf16 [L, D]. q_rstd f32 [B*L*Hq], k_rstd f32 [B*L*Hkv] (optional, saved for backward). */
int vq3_qwen_qkprep_fwd(const void* qkv, const void* q_w, const void* k_w, const void* cos, const void* sin, void* Q,
                        void* K, void* V, float* q_rstd, float* k_rstd, int32_t B, int32_t L, int32_t Hq, int32_t Hkv,
                        int32_t D, float eps, void* stream);
/* Backward: dQ,dK,dV (layouts as above) + saved qkv, rstd -> dqkv bf16 [B*L,(Hq+2Hkv)*D];
 * dq_w_part, dk_w_part f32 [ceil(B*L / VQ3_QKPREP_BWD_TOKENS_PER_PART), D]: one partial row per workgroup = per 8 consecutive tokens
 * (plain stores); reduce that many rows with vq3_colsum_f32_to_bf16.
 * kv_parts: dK and dV are [kv_parts, B, Hkv, L, D] partial slabs (vq3_qwen_flash_bwd), summed here in f32. */
#define VQ3_QKPREP_BWD_TOKENS_PER_PART 8
int vq3_qwen_qkprep_bwd(const void* dQ, const void* dK, const void* dV, const void* qkv, const void* q_w,
                        const void* k_w, const void* cos, const void* sin, const float* q_rstd, const float* k_rstd,
                        void* dqkv, float* dq_w_part, float* dk_w_part, int32_t kv_parts, int32_t B, int32_t L, int32_t Hq,
                        int32_t Hkv, int32_t D, void* stream);

/* Masked softmax over the last dim. S f32 [nb, Lq, ldS] -> P bf16 [nb, Lq, ldP]; columns >= Lk (up to ldP) zeroed.
 * causal != 0: key j visible to query i iff j <= i. keymask u8 [nb / heads_per_mask, Lk] or NULL (1 = visible).
 * Fully masked rows produce zeros. */
int vq3_softmax_fwd(const float* S, void* P, const uint8_t* keymask, int32_t nb, int32_t heads_per_mask, int32_t Lq,
                    int32_t Lk, int32_t ldS, int32_t ldP, int32_t causal, void* stream);
/* dS = scale * P * (dP - rowsum(P*dP)); P bf16, dP f32 [nb,Lq,ldS], dS bf16 [nb,Lq,ldP] (pad columns zeroed). */
int vq3_softmax_bwd(const void* P, const float* dP, void* dS, int32_t nb, int32_t Lq, int32_t Lk, int32_t ldS,
                    int32_t ldP, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Embedding / splice / loss (vggt_qwen3_vlm.py:190-195; modeling_qwen3.py:495; loss_utils.py:32-71)
 * ---------------------------------------------------------------------------------------------------------- */
/* inputs_embeds build: out[b,l,:] = srcmap[b,l] < 0 ? table[ids[b,l],:] : feat[b, srcmap[b,l], :].
 * srcmap i32 [B,L] is the host-computed image of the reference's python loop
 * `inputs_embeds[b, pos:pos+S] = features[b]` (last writer wins for repeated <image> tokens); the host has
 * already raised for pos+S > L exactly as the reference's index assignment does. feat bf16 [B,S,H]. */
int vq3_embed_splice_fwd(const int64_t* ids, const void* table, const void* feat, const int32_t* srcmap, void* out,
                         int32_t B, int32_t L, int32_t H, int32_t S, void* stream);
/* Backward of the above. sorted_ids/order = sort of ids over all B*L positions (stable): the embedding-table
 * gradient (bf16 [V,H], tied with lm_head) gets += sum of dout rows per distinct id, skipping spliced positions
 * (deterministic, no atomics). dfeat_f32 [B,S,H] += dout rows of spliced positions (f32 atomics; may be NULL). */
int vq3_embed_splice_bwd(const int64_t* sorted_ids, const int64_t* order, const int32_t* srcmap, const void* dout,
                         void* dtable_bf16, float* dfeat_f32, int32_t B, int32_t L, int32_t H, int32_t S, void* stream);
/* out[i,:] = src[idx[i],:] for i < n (rows i >= n up to n_pad zero-filled). bf16 rows of `cols`. */
int vq3_gather_rows(const void* src, const int32_t* idx, void* out, int32_t n, int32_t n_pad, int32_t cols,
                    void* stream);
/* dst[idx[i],:] (+)= src[i,:] for i < n (idx unique). */
int vq3_scatter_rows(const void* src, const int32_t* idx, void* dst, int32_t n, int32_t cols, int32_t accumulate,
                     void* stream);
/* Cross entropy over bf16 logits [n, ldl] with V valid columns, targets i32 [n] (all valid):
 * loss_sum_f32[0] += sum_i (logsumexp_i - logit_i[target_i]); dlogits (in place, bf16) = (softmax - onehot) * gscale,
 * columns V..ldl-1 zeroed. Caller zeroes loss_sum and divides by n. */
int vq3_cross_entropy_fwd_bwd(void* logits, const int32_t* targets, float* loss_sum_f32, int32_t n, int32_t V,
                              int32_t ldl, float gscale, void* stream);
/* The same with a per-row gradient scale and the per-row losses written out (row_loss_f32[i] = logsumexp_i - logit_i[target_i]):
 * rows of several micro-batches in one launch, each micro-batch's mean taken over its own labelled rows (loss_utils.py:49-71 per
 * micro-batch, train_sft.py:213 scaling). */
int vq3_cross_entropy_rows(void* logits, const int32_t* targets, const float* row_scale_f32, float* row_loss_f32, int32_t n, int32_t V,
                           int32_t ldl, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * VGGT aggregator (un-vendored `vggt` package: vggt.models.aggregator.Aggregator, vggt.layers.{block,attention,rope,
 * vision_transformer}); call site in the reference: src/models/vggt_qwen3_vlm.py:144
 * ---------------------------------------------------------------------------------------------------------- */
/* Patch extraction for the 14x14/stride-14 patch-embed conv, fused with the aggregator's ImageNet normalisation:
 * images f32 [NI,3,H,W] in [0,1] -> patches bf16 [NI*(H/p)*(W/p), Kp], k = c*p*p + ky*p + kx, columns >= 3*p*p zeroed
 * (Kp % 64 == 0 so the result feeds vq3_gemm_bf16_nt). mean/std are 3 host floats each. */
int vq3_im2col_norm(const float* images, void* patches, int32_t NI, int32_t H, int32_t W, int32_t p, int32_t Kp,
                    const float* mean3_host, const float* std3_host, void* stream);
/* Attention.forward prologue (vggt/layers/attention.py): qkv bf16 [T, 3*NH*64] -> Q,K,V bf16 [T/N, NH, N, 64];
 * optional per-head LayerNorm on q,k (f32 weight/bias [64]) and 2-D RoPE (vggt/layers/rope.py; cos/sin bf16
 * [maxpos+1, 32]); token n of a group sits in frame position n % tokens_per_frame; the first patch_start tokens of a
 * frame are special tokens at position 0, patch p at (p / Wp + 1, p % Wp + 1). */
int vq3_vit_qkprep(const void* qkv, const float* qn_w, const float* qn_b, const float* kn_w, const float* kn_b,
                   const void* cos, const void* sin, void* Q, void* K, void* V, int64_t T, int32_t N, int32_t NH,
                   int32_t head_dim, int32_t tokens_per_frame, int32_t patch_start, int32_t Wp, int32_t use_norm,
                   int32_t use_rope, float eps, void* stream);
/* F.scaled_dot_product_attention(q, k, v), non-causal, head_dim 64: Q, K, V bf16 [G*NH, N, 64] (V as stored: the kernel
 * reads it transposed out of LDS) -> O bf16 token-major O[(g*N + n)*ldo + h*64 + d]. Q, K, V, O 16-byte aligned, ldo % 8 == 0 (a head's
 * 128 output bytes of a row leave as 16-byte pieces). Workgroups are placed so that the query blocks of one (group, head) pair - which
 * walk the same K / V rows - run on ONE XCD and share its L2 (vggt.hip; VQ3_FLASH_XCD=0: the plain 2-D grid). */
int vq3_flash_attn_fwd(const void* Q, const void* K, const void* V, void* O, int32_t G, int32_t NH, int32_t N,
                       int32_t head_dim, int64_t ldo, float scale, void* stream);
/* The same for the first q_rows queries of every group only (keys / values: all N): O[(g*q_rows + n)*ldo + h*64 + d], n < q_rows.
 * The reference keeps only the first num_vis_tokens rows of the last global block's output (src/models/vggt_qwen3_vlm.py:148-156). */
int vq3_flash_attn_fwd_rows(const void* Q, const void* K, const void* V, void* O, int32_t G, int32_t NH, int32_t N,
                            int32_t q_rows, int32_t head_dim, int64_t ldo, float scale, void* stream);
/* vq3_flash_attn_fwd_rows with a promise from the caller: |scale * q . k| <= score_bound for every (query, key) pair (q_rows = N: all
 * queries). VGGT's frame / global blocks normalise q and k with a LayerNorm over the 64 head channels before the (norm-preserving) RoPE
 * (upstream vggt/layers/attention.py: q_norm / k_norm; not vendored, third_party/README.md:5-11), so |q| <= 8 max|gamma_q| + |beta_q| and
 * likewise |k|: a bound that depends on the WEIGHTS only. score_bound <= 62.4 (= 90 / log2 e) selects kernels without a running maximum
 * (exact - softmax is shift-invariant; no accumulator fill, growth scan or rescale per key tile); larger or negative (= none): the general
 * kernels. A wrong promise can overflow: the caller owns it. VQ3_FLASH_NOMAX=0 ignores the promise (A/B runs). */
int vq3_flash_attn_fwd_bounded(const void* Q, const void* K, const void* V, void* O, int32_t G, int32_t NH, int32_t N,
                               int32_t q_rows, int32_t head_dim, int64_t ldo, float scale, float score_bound, void* stream);

/* Fused cross-attention of a PerceiverLayer (src/models/projector_perceiver.py:33,44: nn.MultiheadAttention(latents, context,
 * context) with attention-weight dropout): O[b N + n, h D + :] = dropout(softmax_t(alpha q[b,n,h] . k[b,t,h])) . v[b,t,h], one launch.
 * q bf16 [B*N, ldq] (head h at columns h*D), kv bf16 [B*T, ldkv] (K at columns h*D, V at v_off + h*D), o bf16 [B*N, ldo].
 * head_dim D in {64, 128, 256, 512}. P / Pd (bf16 [B*H, N, Tp], either may be NULL): the softmax and its dropped-out copy as
 * vq3_softmax_fwd + vq3_dropout would have left them (pad columns zero), for callers that run a backward pass. The dropout decision
 * for P[b, h, n, t] is vq3_dropout's for element ((b H + h) N + n) Tp + t at (seed, offset) whether or not P is kept. p_drop = 0: none. */
int vq3_perceiver_xattn_fwd(const void* q, const void* kv, void* o, void* P, void* Pd, int32_t B, int32_t H, int32_t N, int32_t T,
                            int32_t head_dim, int64_t ldq, int64_t ldkv, int64_t ldo, int64_t v_off, int32_t Tp, float alpha,
                            float p_drop, uint64_t seed, uint64_t offset, void* stream);

/* Inverted dropout in place (the four nn.Dropout sites of a PerceiverLayer, src/models/projector_perceiver.py:33,37,42,46-49,
 * which stay ACTIVE under `model.train()` although encode_images runs under no_grad): element i is zeroed with probability p,
 * else scaled by 1/(1-p); the decision is a counter-based hash of (seed, offset + i). x is bf16, or f32 when is_f32. */
int vq3_dropout(void* x, int32_t is_f32, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimiser (src/train/train_sft.py:147-156 torch.optim.AdamW; DeepSpeed bf16 keeps f32 master weights)
 * ---------------------------------------------------------------------------------------------------------- */
/* master/m/v f32, grad bf16, w_bf16 (compute copy) updated in place. step >= 1. grad is scaled by gscale first.
 * clip_sumsq (device f32[1], may be NULL): global-norm gradient clipping as the reference's default launch applies it
 * (configs/deepspeed_zero3.json:15 "gradient_clipping": 1.0; formula of torch.nn.utils.clip_grad_norm_): the scaled
 * gradient is further multiplied by min(1, max_norm / (sqrt(*clip_sumsq) * gscale + 1e-6)). */
int vq3_adamw_step(float* master, float* m, float* v, const void* grad_bf16, void* w_bf16, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int32_t step, float gscale,
                   const float* clip_sumsq, float max_norm, void* stream);
/* accum[0] += sum_i x[i]^2 over n elements (x bf16, or f32 when is_f32). Deterministic: <= 1024 per-block partials are written
 * to `partials` (scratch, >= 1024 floats) and added in a fixed order by a second one-block kernel. */
int vq3_sumsq(const void* x, int32_t is_f32, int64_t n, float* partials, float* accum, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batch builder: the collator step right before the model (src/dataio/collate_multiview.py)
 * ---------------------------------------------------------------------------------------------------------- */
/* One source image of a vq3_resize_crop_u8 batch. `src` is a DEVICE pointer to uint8 RGB, row-major HWC with `pitch`
 * bytes per row. The *_off fields index the shared `coefs` / `bounds` device arrays (int32 elements) where this
 * image's horizontal (kh, bh) and vertical (kv, bv) resampling plans start; crop_x / crop_y locate the S x S centre
 * crop inside the resized image (torchvision center_crop: int(round((size - S) / 2.0)), round-half-even). */
typedef struct vq3_image_desc {
  const uint8_t* src;
  int32_t h, w, pitch;
  int32_t ksize_h, ksize_v;
  int32_t kh_off, kv_off;
  int32_t bh_off, bv_off;
  int32_t crop_x, crop_y;
} vq3_image_desc;

/* Pillow's bicubic resampling plan for one axis (PIL.Image.resize(..., BICUBIC) as called by
 * torchvision.transforms.Resize, collate_multiview.py:15): taps per output index. HOST function, needs no GPU.
 * vq3_resample_ksize returns the row stride of `coefs` (taps per output, -1 on bad sizes); vq3_resample_plan fills
 * bounds[out_size*2] = {first source index, tap count} and coefs[out_size*ksize] (22-bit fixed point). */
int vq3_resample_ksize(int32_t in_size, int32_t out_size);
int vq3_resample_plan(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs);
/* Resize(S, BICUBIC) -> CenterCrop(S) -> ToTensor() (collate_multiview.py:12-19) for n_images uint8 images in one
 * launch: out f32 [n_images, 3, S, S] in [0,1], bit-identical to the PIL/torchvision pipeline. tile_rows output rows
 * per workgroup; max_src_rows = the largest number of source rows any such tile touches (LDS: max_src_rows*256 B). */
int vq3_resize_crop_u8(const vq3_image_desc* descs_dev, int32_t n_images, const int32_t* coefs_dev,
                       const int32_t* bounds_dev, float* out, int32_t S, int32_t tile_rows, int32_t max_src_rows,
                       void* stream);
/* Token layout of MultiViewCollator.__call__ (collate_multiview.py:56-79): row b = (prompt_b + answer_b)[:max_length]
 * padded with pad_id to L; labels = -100 on the prompt and the padding, the answer ids elsewhere; attention_mask =
 * (input_ids != pad_id). prompt_ids / answer_ids are the concatenated ragged lists (int32), *_off [B+1] their row
 * starts; outputs int64 [B, L]. L (>= every truncated row) is chosen by the caller (:69 max(longest, floor)). */
int vq3_pack_tokens(const int32_t* prompt_ids, const int32_t* prompt_off, const int32_t* answer_ids,
                    const int32_t* answer_off, int32_t B, int32_t L, int32_t max_length, int64_t pad_id,
                    int64_t* input_ids, int64_t* labels, int64_t* attention_mask, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Greedy decoding with a resident KV cache (src/inference/qa_inference.py:207-216, arkit_inference.py:274-284:
 * text_model.generate(inputs_embeds=..., do_sample=False, num_beams=1, repetition_penalty, no_repeat_ngram_size))
 * All per-step state (lens, step, generated, finished) is read from DEVICE memory so a step can be graph-replayed.
 * ---------------------------------------------------------------------------------------------------------- */
/* y[M, N] = f(x)[M, K] . W[N, K]^T (+ residual[M, N] bf16), 1 <= M <= 8: the weight-streaming (HBM-bound) form of
 * nn.Linear for one token per row. x, W bf16, K % 8 == 0; y bf16 (or f32 if out_f32). xmode selects f:
 *   0  f(x) = x
 *   1  f(x) = ln_w * bf16(x * rsqrt(mean(x^2) + eps))       Qwen3RMSNorm (modeling_qwen3.py:49-64) fused in
 *   2  f(x) = bf16(bf16(silu(x[:, :K])) * x[:, K:2K])        SwiGLU (modeling_qwen3.py:81-83) fused into down_proj */
int vq3_skinny_gemm_bf16(const void* x, const void* W, void* y, const void* residual, const void* ln_w, float eps,
                         int32_t xmode, int32_t M, int32_t N, int32_t K, int64_t ldx, int64_t ldw, int64_t ldy,
                         int64_t ldr, int32_t out_f32, void* stream);
/* The same product with e4m3 weights (config C5): Wq uint8 e4m3 [N, K] with one f32 scale per output row; the prepared
 * activation row f(x) is quantised per token in registers exactly as vq3_quant_fp8_rows does, so decode and the fp8
 * GEMM share one numeric contract. M <= 2, K % 16 == 0, K <= 12288; y bf16. */
int vq3_skinny_gemm_fp8(const void* x, const void* Wq, const float* w_scale, void* y, const void* residual,
                        const void* ln_w, float eps, int32_t xmode, int32_t M, int32_t N, int32_t K, int64_t ldx,
                        int64_t ldw, int64_t ldy, int64_t ldr, void* stream);
/* One new token per row: qkv bf16 [B, (Hq+2Hkv)*128] -> q_norm/k_norm + RoPE at position lens[b]
 * (modeling_qwen3.py:237-253); Q bf16 [B, Hq, 128]; K, V written into the caches [B, Hkv, Lmax, 128] at slot lens[b].
 * cos/sin: bf16 [>= Lmax, 128] tables. */
int vq3_qwen_decode_qkprep(const void* qkv, const void* q_w, const void* k_w, const void* cos, const void* sin,
                           const int32_t* lens, void* Q, void* Kcache, void* Vcache, int32_t B, int32_t Hq,
                           int32_t Hkv, int32_t head_dim, int32_t Lmax, float eps, void* stream);
/* O[b, h, :] = softmax(scale * q . K[b, h/G, :lens[b]+1]^T) . V[b, h/G, :lens[b]+1]  (GQA, fp32 softmax). */
int vq3_qwen_decode_attn(const void* Q, const void* Kcache, const void* Vcache, const int32_t* lens, void* O, int32_t B,
                         int32_t Hq, int32_t Hkv, int32_t head_dim, int32_t Lmax, float scale, void* stream);
/* transformers greedy step on bf16 logits [B, ld]: RepetitionPenaltyLogitsProcessor over generated[b, :*step], then
 * NoRepeatNGramLogitsProcessor, argmax (first index on ties), finished rows emit pad_id, eos marks a row finished.
 * Writes generated[b, *step] and next_ids[b]. work: scratch of >= B*128 + 1 four-byte slots; max_new (columns of
 * `generated`, prompt ids included when they take part in the penalty) <= 512. */
int vq3_greedy_pick(const void* logits_bf16, int64_t ld_logits, float* work, int32_t B, int32_t V, int64_t* generated,
                    int32_t max_new, const int32_t* step, int32_t* finished, float repetition_penalty,
                    int32_t no_repeat_ngram, const int64_t* eos_ids, int32_t n_eos, int64_t pad_id, int32_t* next_ids,
                    void* stream);
/* lens[b] += 1 for b < B (B <= 64), *step += 1 (either pointer may be NULL): the last launch of a decode step. */
int vq3_decode_advance(int32_t* lens, int32_t B, int32_t* step, void* stream);

/* All decoder layers of one decode step for ONE row (B = 1, the reference's callers) in one persistent launch: per layer
 * q|k|v[+RMSNorm] -> q/k-norm + RoPE + cache append + attention -> o[+residual] -> gate|up[+RMSNorm] + SwiGLU -> down[+residual]
 * (modeling_qwen3.py:49-83, 185-207, 237-330), the arithmetic of the five entry points above, with the weight stream running through
 * the phase boundaries (256 workgroups, grid barriers in device memory; csrc/decode_layers.hip).
 * `weights`: DEVICE array [layers][8] of device pointers, in the order qkv [(Hq+2Hkv)*128, hidden], o [hidden, Hq*128],
 * gate|up [2*intermediate, hidden] (gate rows first), down [hidden, intermediate], input_layernorm, post_attention_layernorm, q_norm,
 * k_norm - all bf16, rows contiguous. `h` [hidden] bf16 holds the stack's input row and receives its output. `workspace`:
 * vq3_qwen_decode_layers_workspace_bytes() of device memory (16-byte aligned; the rows that travel between workgroups). Kcache /
 * Vcache: layer 0's [Hkv, Lmax, 128]; layer l's at + l * cache_layer_stride elements; the new row is appended at position lens[0]
 * (lens is NOT advanced here).
 * `barrier`: 4096 bytes (1024 uint32) that MUST be zero when the kernel starts (zero them on the same stream before every launch).
 * `status`: one uint32 the kernel ORs into and never clears: 1 = a grid-barrier wait ran out (a workgroup was not co-resident;
 * the outputs are garbage), 2 = cache full (lens[0] >= Lmax). The caller reads it when it next synchronises.
 * vq3_qwen_decode_layers_supported: 1 when this shape runs here (Qwen3-4B's 2560 / 9728 / 32 / 8 x 128, Lmax <= 2048, >= 256 CUs),
 * else 0 - the per-projection launches above cover every other case. */
typedef struct vq3_decode_layers_desc {
  const void* const* weights;
  void* h;
  void* workspace;
  const void* cos;
  const void* sin;
  const int32_t* lens;
  void* Kcache;
  void* Vcache;
  int64_t cache_layer_stride;
  uint32_t* barrier;
  uint32_t* status;
  int32_t layers, hidden, intermediate, Hq, Hkv, head_dim, Lmax;
  float eps, scale;
} vq3_decode_layers_desc;
int64_t vq3_qwen_decode_layers_workspace_bytes(void);
int vq3_qwen_decode_layers_supported(int32_t hidden, int32_t intermediate, int32_t Hq, int32_t Hkv, int32_t head_dim, int32_t Lmax);
int vq3_qwen_decode_layers(const vq3_decode_layers_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * FP8 forward GEMM (BASELINE config C5: Qwen3 linear weights in OCP e4m3, one fp32 scale per output channel;
 * activations quantised per token on the fly; fp32 accumulate). Replaces nn.Linear of modeling_qwen3.py:81-83,241-280.
 * ---------------------------------------------------------------------------------------------------------- */
/* Per-row quantisation: scale[r] = max|x[r,:]| / 448 (1 if the row is zero), q[r,k] = e4m3_rne(x[r,k] * 448 / amax).
 * x bf16 [rows, K] (ldx), q uint8 e4m3 [rows, K] (ldq). Used for activations (row = token) and weights (row = output
 * channel). */
int vq3_quant_fp8_rows(const void* x_bf16, int64_t ldx, int64_t rows, int32_t K, void* q, int64_t ldq, float* scale,
                       void* stream);
/* C[M,N] bf16 = bf16(x_scale[m] * w_scale[n] * sum_k Xq[m,k] Wq[n,k]) (+ residual bf16). Xq [M,K], Wq [N,K] e4m3,
 * K % 128 == 0. Block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales) at 2x the bf16 rate. */
int vq3_gemm_fp8_nt(const void* Xq, const float* x_scale, const void* Wq, const float* w_scale, void* C,
                    const void* residual, int32_t M, int32_t N, int32_t K, int64_t ldx, int64_t ldw, int64_t ldc,
                    int64_t ldr, void* stream);
/* The same quantisation of x[r, k] * colmul[k] (colmul f32 [K], 16-byte aligned): the dgrad GEMMs dX = dY . W contract over W's output
 * channels, so W's per-output-channel scales are folded into dY before its rows are quantised. */
int vq3_quant_fp8_rows_scaled(const void* x_bf16, int64_t ldx, int64_t rows, int32_t K, const float* colmul, void* q, int64_t ldq,
                              float* scale, void* stream);
/* dst[c, r] = src[r, c] for BYTE matrices (e4m3 weights -> the W^T operand of the NT dgrad GEMMs). R, C, lds, ldd multiples of 16. */
int vq3_transpose_u8(const void* src, void* dst, int32_t R, int32_t C, int64_t lds, int64_t ldd, void* stream);
/* e4m3 GEMM with the bf16 path's fused epilogues, on the 256x256 8-phase kernel (gemm6.hip, F8 instantiations), the last round of tiles
 * split along K where that pays. mode 0: C[M,N] = bf16(x_scale[m] w_scale[n] sum_k Xq[m,k] Wq[n,k]) (+ residual); mode 1 (SwiGLU forward,
 * modeling_qwen3.py:81-83): Wq = the gate|up weight [N = 2 I, K], C = act [M, I] = bf16(bf16(silu(gate)) * up), gu (or NULL) = gate|up
 * [M, 2 I]; mode 2 (SwiGLU backward): the product is d(act) [M, N] and is never stored - dgu [M, 2 N] leaves, from the saved gate|up gu.
 * w_scale == NULL means 1 (scales already folded into Xq: vq3_quant_fp8_rows_scaled). No counterpart in the reference (it has no fp8). */
typedef struct vq3_gemm_fp8_desc {
  const void* Xq; const float* x_scale;
  const void* Wq; const float* w_scale;
  void* C; const void* residual;
  int32_t M, N, K;
  int64_t ldx, ldw, ldc, ldr;
  int32_t mode;
  void* gu;
  void* dgu;
} vq3_gemm_fp8_desc;
int vq3_gemm_fp8_ex(const vq3_gemm_fp8_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused causal GQA attention for Qwen3, head_dim 128, 1..4 query heads per kv head (modeling_qwen3.py:185-207,254-280 + autograd):
 * softmax(scale * Q K^T + causal & key-padding mask) V, scores in fp32, probabilities bf16 before the PV product.
 * Q [B,Hq,L,128], K/V [B,Hkv,L,128] bf16 (as vq3_qwen_qkprep_fwd writes them); keymask uint8 [B,L] (0 = padded key);
 * O / dO token-major: row (b*L + q), head h at column h*128, row stride ldo / lddo elements.
 * ---------------------------------------------------------------------------------------------------------- */
/* LSE f32 [B,Hq,L] (+ 4 floats of slack behind it, also behind Delta: the backward reads them 16 bytes at a time):
 * log2-domain log-sum-exp of the scaled scores (+inf for rows with no visible key; their O is 0). */
int vq3_qwen_flash_fwd(const void* Q, const void* K, const void* V, const void* keymask, void* O, float* LSE, int32_t B,
                       int32_t L, int32_t Hq, int32_t Hkv, int32_t head_dim, int64_t ldo, float scale, void* stream);
/* dQ [B,Hq,L,128]; dK/dV bf16 [kv_parts, B,Hkv,L,128]: kv_parts (1..4) workgroups share a key tile, each walking every
 * kv_parts-th query block and writing its own partial slab (summed over the query heads of each kv head in registers/LDS, no
 * atomics); the consumer (vq3_qwen_qkprep_bwd) adds the slabs. Key tiles without an attended key are skipped in all passes
 * (exact zeros). Delta f32 [B,Hq,L] is scratch (rowsum(dO * O), written by the dQ pass, read by the dK/dV pass). */
int vq3_qwen_flash_bwd(const void* Q, const void* K, const void* V, const void* keymask, const void* O, const void* dO,
                       const float* LSE, float* Delta, void* dQ, void* dK, void* dV, int32_t kv_parts, int32_t B, int32_t L,
                       int32_t Hq, int32_t Hkv, int32_t head_dim, int64_t ldo, int64_t lddo, float scale, void* stream);
/* Benchmarking / test hook for vq3_gemm_bf16_nt's kernel choice on NT, K % 64 == 0 shapes: cfg = -3 restores the automatic
 * choice (the default), -1 the register-staged reference kernel, 0..14 a gemm2.hip tile configuration, 20 / 21 / 22 the
 * 8-phase kernels of gemm6.hip (256x256 / 256x128 / 128x256 tiles), 24 the two-workgroups-per-CU 256x128 kernel of gemm7.hip, 25 the
 * 256x256 kernel with the last round's tiles split along K, 30 whole rounds of 256x256 tiles + a row tail; 102 / 103 / 105 pin the schedule of the any-layout kernel
 * (gemm3.hip: 128x128 2-stage, 128x128 loader ring, 256x128 loader ring; -3 releases it too). Process-wide; not meant for concurrent use with launches on other threads. */
int vq3_gemm_force_config(int32_t cfg);

/* Host-only test hook (no launch, no GPU needed): the workgroup -> tile map the GEMM kernels use for an M x N output cut into
 * bm x bn tiles with wg_per_cu workgroups resident per CU. Workgroup id b runs on XCD b % 8; each XCD owns one rectangle of an
 * xm x (8 / xm) blocking of the tile grid and walks it in bands of `band` n-tiles (n-fastest inside a band, then m), so the tiles
 * an XCD has in flight together share few row panels of A and few panels of B in its private 4 MiB L2. Writes the chosen xm / band
 * and, when `order` is non-null, (m-tile, n-tile) of workgroup ids 0 .. mtiles * ntiles - 1 into order[2 * id], order[2 * id + 1].
 * There is no counterpart in the reference (cuBLAS picks its own rasterisation behind torch.nn.functional.linear). */
int vq3_gemm_tile_order(int32_t M, int32_t N, int32_t bm, int32_t bn, int32_t wg_per_cu, int32_t* xm_out, int32_t* band_out,
                        int32_t* order);

/* Kernel-choice hygiene (gemm.hip: tuned_choice). The first call of a new (shape, layout, epilogue kind) times its candidate kernels
 * unless a table entry answers. vq3_gemm_tune_table_load reads "M N K batch flags cfg" lines ('#' comments) into the table (the package
 * ships one for the Stage-1 shapes; VQ3_GEMM_TUNE_FILE=<path> still adds / appends). vq3_gemm_tune_workspace hands the tuner caller-owned
 * device memory for its trial output and cache-flush buffer (>= trial bytes + 320 MiB, 256-byte aligned; NULL, 0 detaches): with a
 * workspace registered the library never allocates for tuning, and a shape that does not fit is not measured. vq3_gemm_autotune_hold(1)
 * stops all measuring (table, then heuristic): multi-rank jobs - a measurement synchronises the device under in-flight collectives and
 * ranks would disagree on near-ties. No counterpart in the reference (cuBLAS heuristics are internal). */
int vq3_gemm_tune_table_load(const char* path, int32_t* entries_out);
int vq3_gemm_tune_workspace(void* ptr, int64_t bytes);
/* Lazy form of the above, and the source of the split-K launches' per-stream workspaces: `fn(bytes, device, kind)` is called from inside
 * a GEMM call - never under graph capture - the first time a measurement (kind 0: trial output + flush buffer, may be called again with a
 * larger size; the previous block may then be released) or a split launch on a new stream (kind 1: 48 MiB + counts, must stay alive for
 * the life of the process) needs device memory on `device`; it returns a 256-byte aligned pointer or NULL ("none available": the shape is
 * not measured / the launch runs unsplit). With a provider registered the library itself never calls hipMalloc. NULL detaches. */
typedef void* (*vq3_ws_provider_t)(int64_t bytes, int32_t device, int32_t kind);
int vq3_gemm_workspace_provider(vq3_ws_provider_t fn);
int vq3_gemm_autotune_hold(int32_t on);

/* Last-round K split of the 256x256 GEMM kernel (cfg 25; chosen by measurement like every other configuration). An M x N output is
 * ceil(M/256) * ceil(N/256) tiles; when the last round of `ncu` CUs would be at most half full, its `rem` tiles are cut into `slices`
 * K ranges run by rem * slices workgroups: all but the last slice of a tile leave f32 partial tiles in a per-stream workspace (allocated on
 * first use outside graph capture) and count themselves in, the last slice adds them and runs the epilogue. No counterpart in the reference
 * (cuBLAS's stream-K is its own). vq3_gemm_split_plan is host-only (no launch): writes the plan, *slices = 0 where the split does not apply.
 * A reducer whose bounded wait expires (a broken launch; its output tile is then incomplete) sets ONE STICKY error word per process,
 * kept in host-mapped memory and never touched by the per-launch memset of the arrival counts: it stays set until it is read with clear.
 * vq3_gemm_split_poll reads it from the host without synchronising anything (1 = a split launch COMPLETED so far on any stream gave up since
 * the last clearing read; `clear` != 0 resets it) - Stage1Trainer polls it once per optimiser step and raises. vq3_gemm_split_status
 * synchronises `stream` first, then reads and clears. vq3_gemm_split_debug_spin_bound (tests only) sets the reducer's wait bound in polls
 * (0 = the production bound, 1 << 23) so that the give-up path can be walked on a healthy device. */
int vq3_gemm_split_plan(int32_t M, int32_t N, int32_t K, int32_t ncu, int32_t* full_out, int32_t* rem_out, int32_t* slices_out);
int vq3_gemm_split_status(void* stream, int32_t* gave_up);
int vq3_gemm_split_poll(int32_t* gave_up, int32_t clear);
int vq3_gemm_split_debug_spin_bound(int64_t polls);

#ifdef __cplusplus
}
#endif
#endif /* VQ3_HIP_H */
