/*
 * goalforce.h — C ABI of libgoalforce_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary underneath the reference's Python call sites
 * (level B5).  The reference (brown-palm/goal-force) has no
 * FFI of its own: every entry point below replaces a *Python* interface, cited
 * as file:line relative to the reference tree.  Abbreviations:
 *   DIT  = diffsynth/models/wan_video_dit.py
 *   GF   = src/goal_force/wan_video_new.py
 *   FM   = diffsynth/schedulers/flow_match.py
 *   VAE  = diffsynth/models/wan_video_vae.py
 *   VRAM = diffsynth/vram_management/layers.py
 *   DS   = src/goal_force/unified_dataset.py
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM) unless a parameter says "host".
 *   - bf16 tensors are passed as `const void*` / `void*` (raw 16-bit storage).
 *   - The caller owns all memory; the library never allocates or frees
 *     user-visible buffers and never synchronises: kernels are enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - Return value: 0 on success, negative gf_status on error; the message is
 *     available from gf_last_error() (thread-local).
 *   - Row-major everywhere; `ld*`/`*_stride` are in ELEMENTS.
 */
#ifndef GOALFORCE_H
#define GOALFORCE_H

#include <stdint.h>

#if defined(GF_BUILD)
#define GF_API __attribute__((visibility("default")))
#else
#define GF_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GF_OK = 0,
    GF_ERR_INVALID_ARG = -1,   /* bad shape / alignment / null pointer */
    GF_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels are built for */
    GF_ERR_LAUNCH = -3         /* hipLaunchKernel / hipFuncSetAttribute failed */
} gf_status;

/* library identity ------------------------------------------------------- */
GF_API const char* gf_version(void);      /* "goalforce-hip <semver> gfx950" */
GF_API const char* gf_last_error(void);   /* thread-local message of the last failure */
GF_API int gf_abi_version(void);          /* bumped on any signature change */
/* Dispatch overrides.  Every kernel in the library ships, each for the shapes its launcher sends it; the parity tests cross-check
 * two kernels on the same operands, which needs a way to route a shape to the one that would not get it by default.  Names:
 * "prefer_8wave" (0/1), "a4_stagger" (>= 0), "a4_group_m" (0 = by K), "conv_nb" (0 = by Cout, 1, 2), "conv_gather" (0/1),
 * "conv_direct" (0/1), "vae_rms3" (0/1).  Values are clamped to their range; an unknown name returns GF_ERR_INVALID_ARG.
 * The library reads NO environment variable: only these calls change the dispatch.  Process-wide, relaxed atomics. */
GF_API int gf_set_option(const char* name, int value);
GF_API int gf_get_option(const char* name, int* value);   /* the value in force (after clamping); a caller that overrides one reads it first to restore it */
GF_API void gf_reset_options(void);   /* back to the shipped dispatch */

/* ------------------------------------------------------------------------
 * gf_layernorm_modulate — LayerNorm over the last dim (fp32 math, one
 * rounding to bf16), optionally affine, optionally followed by the AdaLN
 * modulate y*(1+scale)+shift with the reference's bf16 rounding after each op.
 * Replaces: nn.LayerNorm norm1/norm2/norm3 + modulate() (DIT:64-65, 206-208,
 * 225-228), WanAutoCastLayerNorm.forward (VRAM:78-92), Head.norm (DIT:258,268).
 *   x, out     [rows, dim] bf16 (out may alias x)
 *   weight,bias[dim] bf16 or NULL  (norm3 affine)
 *   scale1p    [dim] bf16 or NULL  — the already-formed (1+scale) vector
 *   shift      [dim] bf16 or NULL
 * dim % 8 == 0, dim <= 8192.
 */
GF_API int gf_layernorm_modulate(const void* x, void* out, const void* weight, const void* bias,
                          const void* scale1p, const void* shift,
                          int64_t rows, int64_t dim, int64_t x_stride, int64_t out_stride,
                          float eps, void* stream);

/* ------------------------------------------------------------------------
 * gf_modulation — forms the AdaLN vectors of one block in bf16 exactly as the
 * reference's eager ops do: out[i,:] = bf16(param[i,:] + t[i % t_rows,:]) and,
 * for rows whose bit is set in onep_mask, out[i,:] = bf16(1 + out[i,:]).
 * Replaces `(self.modulation + t_mod).chunk(6)` and the `1 + scale` of
 * modulate() (DIT:64-65, 218-219) and Head's `(modulation + t).chunk(2)` (DIT:264-268).
 *   param [k, dim] bf16, t [t_rows, dim] bf16, out [k, dim] bf16; k <= 32.
 */
GF_API int gf_modulation(const void* param, const void* t, void* out, int64_t k, int64_t dim,
                         int64_t t_rows, uint32_t onep_mask, void* stream);

/* ------------------------------------------------------------------------
 * gf_rmsnorm_rope — in place: RMSNorm over the FULL row (all heads jointly;
 * fp32 math, round to bf16, then bf16 multiply by weight) followed by 3-D RoPE
 * on adjacent element pairs.  Replaces RMSNorm.forward (DIT:100-111) and
 * rope_apply (DIT:92-97) as used by SelfAttention.forward (DIT:141-145) and
 * CrossAttention.forward (DIT:177-178; cos/sin NULL => no RoPE).
 *   x       [rows, dim] bf16, row stride x_stride
 *   weight  [dim] bf16
 *   cos,sin [rows, head_dim/2] fp32 or NULL
 * dim % 8 == 0, dim <= 8192, head_dim % 8 == 0.
 */
GF_API int gf_rmsnorm_rope(void* x, const void* weight, const float* cos_tab, const float* sin_tab,
                    int64_t rows, int64_t dim, int64_t head_dim, int64_t x_stride,
                    float eps, void* stream);

/* ------------------------------------------------------------------------
 * gf_gemm_bf16 — C[M,N] = epilogue(A[M,K] · W[N,K]^T + bias[N]); bf16 in/out,
 * fp32 MFMA accumulate (v_mfma_f32_16x16x32_bf16).  Replaces nn.Linear /
 * F.linear at DIT:131-134,146 (q,k,v,o), DIT:209-210 (ffn), DIT:309-320
 * (text/time embeddings), DIT:263 (head), the Conv3d patch embeddings
 * (DIT:307-308,342; GF:85,92 — after gf_patchify_im2col) and the ControlNet
 * zero-conv Conv1d(k=1) (GF:113-116, 1565-1570).
 *   epilogue: see gf_epilogue.  resid [M,N] (ldr) and gate [N] are bf16.
 *   C may alias resid (in-place residual update).
 * K % 64 == 0, N % 8 == 0, lda/ldw/ldc/ldr % 8 == 0, 16-byte aligned bases.
 */
typedef enum {
    GF_EPI_BIAS = 0,           /* bf16(acc + bias)                               */
    GF_EPI_BIAS_GELU_TANH = 1, /* bf16(gelu_tanh(bf16(acc + bias)))              */
    GF_EPI_BIAS_GATE_RESID = 2,/* bf16(resid + bf16(gate * bf16(acc + bias)))    */
    GF_EPI_BIAS_RESID = 3,     /* bf16(resid + bf16(acc + bias))                 */
    GF_EPI_BIAS_SILU = 4,      /* bf16(silu(bf16(acc + bias)))                   */
    GF_EPI_BIAS_MUL = 5        /* bf16(bf16(acc + bias) * resid)  — GEGLU: fc1(x) * gelu(gate(x)),
                                  wan_video_text_encoder.py:106                  */
} gf_epilogue;

GF_API int gf_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                 void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                 int epilogue, const void* resid, int64_t ldr, const void* gate,
                 void* stream);

/* gf_gemm_bf16_batched — `batch` independent products C_b = A_b W_b^T (no bias, no epilogue) in ONE launch: problem b reads
 * A + b*strideA ([M, K], row pitch lda), W + b*strideW ([N, K], row pitch ldw) and writes C + b*strideC ([M, N], row pitch ldc); strides in
 * elements, and a stride may be smaller than a matrix (heads side by side in one [L, H*d] tensor: strideA = d, lda = H*d).  The small
 * per-head / per-frame products of the umT5 encoder's attention (wan_video_text_encoder.py:61-93: q k^T and attn v per head) and of the
 * VAE AttentionBlock (VAE:304-342, per frame).  K a multiple of 64, N of 8.  Runs on the 8-wave kernel: bit-identical to `batch`
 * gf_gemm_bf16 calls on that kernel (M < 512, or gf_set_option("prefer_8wave", 1)); the 4-wave kernel gf_gemm_bf16 takes for M >= 512 rotates the K
 * loop's start per column tile, i.e. sums the same products in another order. */
GF_API int gf_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW, void* C,
                                int64_t ldc, int64_t strideC, int64_t M, int64_t N, int64_t K, int64_t batch, void* stream);

/* ------------------------------------------------------------------------
 * gf_flash_attn_fwd — softmax(Q K^T * scale) V per head, non-causal, no mask,
 * no dropout, head_dim 128.  Replaces flash_attention() (DIT:28-61) /
 * AttentionModule (DIT:114-121).  q,k,v,o are [len, heads*128] bf16 with the
 * head h at columns [h*128, (h+1)*128) ("b s (n d)" layout, DIT:56-60); row
 * strides in elements (so q/k/v may be slices of one fused buffer).
 */
GF_API int gf_flash_attn_fwd(const void* q, const void* k, const void* v, void* o,
                      int64_t q_len, int64_t kv_len, int64_t heads, int64_t head_dim,
                      int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                      float scale, void* stream);

/* gf_transpose_v + gf_flash_attn_fwd_vt — the same attention with V handed over pre-transposed: vt [heads][128][kv_pad] bf16
 * (kv_pad a multiple of 64, keys >= kv_len zero, keys permuted 0-3,8-11,4-7,12-15 inside every group of 16 — written by
 * gf_transpose_v).  One LDS read per PV MFMA instead of two; what the self-attention of the DiT blocks uses.  lse may be NULL. */
GF_API int gf_transpose_v(const void* v, int64_t v_stride, void* vt, int64_t kv_len, int64_t kv_pad, int64_t heads, void* stream);
GF_API int gf_flash_attn_fwd_vt(const void* q, const void* k, const void* vt, void* o, float* lse,
                         int64_t q_len, int64_t kv_len, int64_t kv_pad, int64_t heads, int64_t head_dim,
                         int64_t q_stride, int64_t k_stride, int64_t o_stride, float scale, void* stream);

/* gf_transpose_v32 + gf_flash_attn_fwd_vt32 — the self-attention of the DiT blocks (round 2; kv_len >= 128): the same attention
 * (DIT:28-61) on v_mfma_f32_16x16x32_bf16, the MFMA shape the chip sustains a higher clock on under random data.  Same
 * arguments as the pair above; the two V^T copies differ in the key order inside a 64-key tile and are NOT interchangeable
 * (vt32: inside every group of 32 keys position 8 g + i holds key 4 g + i for i < 4, 16 + 4 g + i - 4 for i >= 4). */
GF_API int gf_transpose_v32(const void* v, int64_t v_stride, void* vt, int64_t kv_len, int64_t kv_pad, int64_t heads, void* stream);
GF_API int gf_flash_attn_fwd_vt32(const void* q, const void* k, const void* vt, void* o, float* lse,
                                  int64_t q_len, int64_t kv_len, int64_t kv_pad, int64_t heads, int64_t head_dim,
                                  int64_t q_stride, int64_t k_stride, int64_t o_stride, float scale, void* stream);

/* gf_linear_vt32 — the V projection of SelfAttention.forward (`v = self.v(x)`, DIT:131-146) written DIRECTLY in the layout
 * gf_flash_attn_fwd_vt32 reads: vt[n * kv_pad + pos(s)] = bf16(sum_k x[s,k] * w[n,k] + bias[n]) for n < N, s < kv_len, positions
 * kv_len .. kv_pad-1 zero, pos() = the key order of gf_transpose_v32.  Bit-identical to gf_gemm_bf16(x, w, bias) followed by
 * gf_transpose_v32 (same MFMA kernel with the operands swapped, same summation order); saves the plain V tensor's round trip.
 *   x [kv_len, ldx] bf16, w [N, ldw] bf16 (nn.Linear layout), bias [N] bf16 or NULL, vt [N * kv_pad] bf16;
 *   kv_pad = kv_len rounded up to 64, N >= 512 and a multiple of 128, K a multiple of 64. */
GF_API int gf_linear_vt32(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* vt,
                          int64_t kv_len, int64_t kv_pad, int64_t N, int64_t K, void* stream);

/* gf_flash_attn_fwd_lastmult — gf_flash_attn_fwd in which the LAST key (row kv_len - 1 of k / v) counts `last_key_multiplicity`
 * times in the softmax: softmax over [k_0 .. k_{n-2}, k_{n-1} x m] evaluated on n keys (the key's score gets + log2 m in the exp2
 * domain).  Cross-attention over a prompt whose padded tail is a run of identical context rows (wan_prompter.py:99-109 zeroes the
 * text-encoder output past the prompt: every padded row is text_embedding(0), so its K and V rows are identical, DIT:177-186) is
 * the same function of q evaluated on the distinct keys only.  Key lengths below the V^T threshold only (the 512-token context). */
GF_API int gf_flash_attn_fwd_lastmult(const void* q, const void* k, const void* v, void* o, int64_t q_len, int64_t kv_len,
                                      int64_t heads, int64_t head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                      int64_t o_stride, float scale, float last_key_multiplicity, void* stream);

/* ------------------------------------------------------------------------
 * Training (ControlNet training step): training_loss (GF:180-193) calls loss.backward() through
 * F.scaled_dot_product_attention (DIT:28-61) in every block.
 * gf_flash_attn_fwd_lse — gf_flash_attn_fwd that also returns lse [q_len, heads] fp32, the log2-domain log-sum-exp
 *   of the scaled scores: softmax row = exp2(scale*log2(e)*S - lse).
 * gf_flash_attn_bwd — dq, dk, dv from (q, k, v, o, dout, lse).  Same layouts / strides as the forward; fp32 accumulation,
 *   bf16 results.  `workspace` is caller-owned, 16-byte aligned, gf_flash_attn_bwd_workspace_bytes(q_len, kv_len, heads) bytes:
 *   rowsum(dout*o) [q_len, heads] fp32 and the dK/dV kernel's per-granule (-lse | -delta) records (a -DGF_BWD_QSCALE=1 build adds
 *   a pre-scaled copy of q: always size the buffer with the function); scratch, its contents need not survive the call.  The
 *   log-sum-exp must come from gf_flash_attn_fwd_lse (or gf_flash_attn_fwd_vt32 / _vt with a non-NULL lse) with the same `scale`.
 *   dk and dv may be NULL together: only dq is computed (a frozen block's cross-attention: nobody reads the context gradients).
 */
GF_API int gf_flash_attn_fwd_lse(const void* q, const void* k, const void* v, void* o, float* lse,
                          int64_t q_len, int64_t kv_len, int64_t heads, int64_t head_dim,
                          int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                          float scale, void* stream);
GF_API int64_t gf_flash_attn_bwd_workspace_bytes(int64_t q_len, int64_t kv_len, int64_t heads);
GF_API int gf_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                      const float* lse, void* workspace, void* dq, void* dk, void* dv,
                      int64_t q_len, int64_t kv_len, int64_t heads, int64_t head_dim,
                      int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride, int64_t do_stride,
                      int64_t dq_stride, int64_t dk_stride, int64_t dv_stride, float scale, void* stream);

/* Backward of the row / elementwise ops of a block and the loss / optimiser of the training step (gf_backward.hip).
 * Gradients: fp32 math on the bf16 forward operands, one rounding to bf16.  *_acc are caller-zeroed fp32 [dim] column
 * accumulators (parameter gradients summed over the token rows; NULL = not wanted).
 * gf_layernorm_bwd   — LayerNorm (DIT:206-208, VRAM:78-92) with y = xhat*g (+b): g = affine weight or (1+scale) or NULL;
 *                      dx; dg_acc += dy*xhat, db_acc += dy.
 * gf_rmsnorm_rope_bwd — RMSNorm (DIT:100-111) (+ RoPE DIT:92-97): x is the PRE-norm tensor; dx; dw_acc.
 * gf_colsum          — acc[n] += sum_r a[r,n]*(b ? b[r,n] : 1); optionally out[r,n] = bf16(a[r,n]*gate[n]) (bias gradients;
 *                      the gated residual out = resid + gate*y: a = dout, b = y gives dgate and dy in one pass).
 * gf_act_bwd         — du = df * act'(u), kind 0 = GELU-tanh (DIT:209), 1 = SiLU.
 * gf_mse_loss        — loss[0] = weight * mean((pred-target)^2), dpred = weight*2(pred-target)/n  (GF:190-191).
 * gf_adamw_step      — torch.optim.AdamW update of a bf16 parameter with fp32 moments (utils.py:755); grad_scale
 *                      multiplies the gradient first (gradient clipping).
 * gf_f32_to_bf16     — rounds an accumulator buffer to the bf16 gradient.                                             */
GF_API int gf_layernorm_bwd(const void* x, int64_t x_stride, const void* dy, int64_t dy_stride, const void* g, void* dx,
                            int64_t dx_stride, float* dg_acc, float* db_acc, int64_t rows, int64_t dim, float eps, void* stream);
GF_API int gf_rmsnorm_rope_bwd(const void* x, int64_t x_stride, const void* dy, int64_t dy_stride, const void* weight,
                               const float* cos_tab, const float* sin_tab, void* dx, int64_t dx_stride, float* dw_acc,
                               int64_t rows, int64_t dim, int64_t head_dim, float eps, void* stream);
GF_API int gf_colsum(const void* a, int64_t lda, const void* b, int64_t ldb, const void* gate, void* out, int64_t ldo,
                     float* acc, int64_t rows, int64_t cols, void* stream);
GF_API int gf_act_bwd(const void* u, const void* df, void* du, int64_t n, int kind, void* stream);
GF_API int gf_mse_loss(const void* pred, const void* target, void* dpred, float* loss, int64_t n, float weight, void* stream);
GF_API int gf_adamw_step(void* param, const void* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
GF_API int gf_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* gf_sumsq — acc[0] += sum of squares of a bf16 buffer (fp32): the global gradient norm of clip_grad_norm_ (utils.py:806-808). */
GF_API int gf_sumsq(const void* x, int64_t n, float* acc, void* stream);

/* ------------------------------------------------------------------------
 * gf_patchify_im2col — gathers the (1,2,2) patches of an NCTHW latent into a
 * token-major matrix for the patch-embedding GEMM.  Replaces the data movement
 * of WanModel.patchify (DIT:341-349) / ControlNet_PatchEmbedding.forward
 * (GF:85-94) for the concatenated input x = cat([latents, y]) (GF:1458).
 *   src0 [c0,F,H,W], src1 [c1,F,H,W] (src1 may be NULL with c1 = 0), bf16
 *   out  [F*(H/2)*(W/2), kpad] bf16, column = c*4 + dy*2 + dx (Conv3d weight
 *   order), columns >= (c0+c1)*4 zero-filled.
 */
GF_API int gf_patchify_im2col(const void* src0, int64_t c0, const void* src1, int64_t c1,
                       void* out, int64_t F, int64_t H, int64_t W, int64_t kpad, void* stream);

/* ------------------------------------------------------------------------
 * gf_unpatchify — 'b (f h w) (x y z c) -> b c (f x) (h y) (w z)' with patch
 * (1,2,2).  Replaces WanModel.unpatchify (DIT:351-356).
 *   tokens [f*h*w, 4*c] bf16  ->  out [c, f, 2h, 2w] bf16
 */
GF_API int gf_unpatchify(const void* tokens, void* out, int64_t c, int64_t f, int64_t h, int64_t w,
                  void* stream);

/* ------------------------------------------------------------------------
 * gf_cfg_euler_step — classifier-free guidance + flow-match Euler update in
 * one pass, with the reference's bf16 rounding after every op:
 *   pred = nega + cfg*(posi - nega)            (GF:716)
 *   latents = latents + pred * dsigma          (FM:72-82; dsigma = sigma_next - sigma)
 * nega may be NULL (cfg_scale == 1: pred = posi, GF:717-718).  In place on latents.
 */
GF_API int gf_cfg_euler_step(void* latents, const void* posi, const void* nega,
                      float cfg_scale, float dsigma, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * gf_act — elementwise activation bf16 -> bf16 (fp32 math, one rounding):
 * kind 0 = SiLU (DIT:316,320), 1 = GELU-tanh (DIT:311).
 */
GF_API int gf_act(const void* x, void* out, int64_t n, int kind, void* stream);

/* ------------------------------------------------------------------------
 * gf_add_bf16 — out = bf16(a + b), elementwise (x + controlnet_state,
 * GF:1570).  out may alias a.
 */
GF_API int gf_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------
 * gf_force_map — renders the Goal-Force control-signal video on the GPU.
 * Replaces the pixel work of ControlSignalDataset_Balls._generate_control_video
 * / get_gaussian_blob / get_blob_for_mass (DS:775-940): every blob contributes
 * amp * exp(-((x-cx)^2 + (y-cy)^2) / denom) to its channel (fp32, summed in blob
 * order, optional clamp to [0,1] as DS:887, one rounding to bf16).
 *   out      [frames, H, W, 3] bf16 (THWC, DS:842)
 *   channels [n_blobs] int32      0 direct force, 1 goal force, 2 mass
 *   params   [n_blobs, 2] fp32    {denom = 2*radius^2, amp}
 *   centers  [n_blobs, frames, 2] fp32  {cx, cy} per frame (host computes them
 *            in float64 exactly as DS:817-823 and rounds once to fp32)
 */
GF_API int gf_force_map(void* out, int64_t frames, int64_t H, int64_t W,
                 const int32_t* channels, const float* params, const float* centers,
                 int64_t n_blobs, int clamp01, void* stream);

/* ========================================================================
 * Wan 3-D causal VAE decoder (VAE = diffsynth/models/wan_video_vae.py).
 * Decoder activations are channels-last [T, H, W, C] bf16; every convolution is
 * gf_vae_im2col (patch gather, temporal halo from the 2-frame feature cache)
 * followed by gf_gemm_bf16 on the weight pre-permuted to [Cout, (dt,dy,dx,cin)].
 * ======================================================================== */

/* gf_vae_prep_latent — un-normalise a latent tile and make it channels-last:
 * out[t,y,x,c] = bf16(bf16(z[c,t,y,x] / inv_std[c]) + mean[c]), c >= C zero-filled.
 * Replaces VideoVAE_.decode's scale step (VAE:1014-1020); z is addressed with
 * element strides (sc,st,sy,sx) so a tile slice needs no copy (VAE:1125).     */
GF_API int gf_vae_prep_latent(const void* z, int64_t sc, int64_t st, int64_t sy, int64_t sx,
                              const void* mean, const void* inv_std, void* out,
                              int64_t C, int64_t T, int64_t H, int64_t W, int64_t cpad, void* stream);

/* gf_vae_im2col — patch gather for CausalConv3d (VAE:33-52; kt in {1,3}, ks in {1,3},
 * causal temporal padding kt-1 taken from `cache` [2,H,W,C] = the reference's
 * feat_cache entry, all-zero when there is no history).  Output frame j is the conv at
 * input frame t_off + j*t_stride (t_stride 2 = the encoder's strided time_conv, VAE:107-112, 160-170).
 *   mode 0: zero spatial padding ks/2, stride 1;
 *   mode 1: Upsample(nearest-exact 2x) folded into a 3x3 conv (decoder Resample, VAE:91-99);
 *   mode 2: ZeroPad2d((0,1,0,1)) + 3x3 conv stride 2 (encoder Resample, VAE:101-106).
 * src [T,H,W,C] -> out [T_out*Ho*Wo, kpad], column ((dt*ks+dy)*ks+dx)*C+c. */
GF_API int gf_vae_im2col(const void* src, const void* cache, void* out, int64_t T_out, int64_t H, int64_t W,
                         int64_t C, int64_t kt, int64_t ks, int mode, int64_t t_stride, int64_t t_off,
                         int64_t kpad, void* stream);

/* gf_conv3d_bf16 — the same convolution as ONE implicit GEMM: gf_vae_im2col's patch matrix is never written; the GEMM's
 * LDS-DMA fetches every 16-byte piece (8 channels of one tap of one output pixel) straight from src / cache (padding taps
 * and the K padding columns from a zero page).  Replaces CausalConv3d.forward (VAE:33-52) and the convs inside Resample
 * (VAE:82-174) end to end; arguments as gf_vae_im2col (T_in = frames in src) + gf_gemm_bf16 (Wm [N, ldw] with columns
 * ((dt*ks+dy)*ks+dx)*C+c, zero from kt*ks*ks*C up to K; K a multiple of 64; epilogue GF_EPI_BIAS or GF_EPI_BIAS_RESID).
 * out [T_out*Ho*Wo, ldc].  Bit-identical to gf_vae_im2col + gf_gemm_bf16.
 * kt == 3 with cache == NULL: the two history frames lie directly IN FRONT of src (src points two frames into one
 * [2 + T_in, H, W, C] buffer) — with mode 0 this selects the pointer-per-row gather (one 64-bit pointer + 4 border flags per
 * output pixel, one offset per tap), 20 % faster than the general one. */
GF_API int gf_conv3d_bf16(const void* src, const void* cache, const void* Wm, int64_t ldw, const void* bias,
                          void* out, int64_t ldc, int64_t T_in, int64_t T_out, int64_t H, int64_t W, int64_t C,
                          int kt, int ks, int mode, int t_stride, int t_off, int64_t N, int64_t K, int epilogue,
                          const void* resid, int64_t ldr, void* stream);

/* gf_vae_finish_latent — encoder tail: out[r,c] = bf16(bf16(x[r,c] - mean[c]) * inv_std[c]) for the first C
 * (mu) channels of the 1x1x1 conv1 output (VideoVAE_.encode, VAE:1002-1010). */
GF_API int gf_vae_finish_latent(const void* x, int64_t ldx, const void* mean, const void* inv_std, void* out,
                                int64_t rows, int64_t C, void* stream);

/* gf_vae_rmsnorm_silu — RMS_norm over channels (F.normalize * sqrt(C) * gamma, bf16
 * rounding after each eager op, VAE:55-70) optionally followed by SiLU
 * (ResidualBlock VAE:277-281, Decoder3d.head VAE:785).  x,out [rows, C], C <= 512.  */
GF_API int gf_vae_rmsnorm_silu(const void* x, const void* gamma, void* out, int64_t rows, int64_t C,
                               int silu, void* stream);

/* gf_softmax_rows — out[r,:ncols] = softmax(bf16(x[r,:]*scale + bias[r,:])) over the first nvalid columns
 * (columns >= nvalid: probability 0, = masked_fill(finfo.min)), out[r,ncols:ldo] = 0; bias may be NULL.
 * The single-head SDPA of the VAE AttentionBlock (VAE:326-333) and the umT5 attention with its relative
 * position bias and key mask (wan_video_text_encoder.py:72-84), between their two GEMMs.                  */
GF_API int gf_softmax_rows(const void* x, int64_t ldx, const void* bias, int64_t ldb, void* out, int64_t ldo,
                           int64_t rows, int64_t ncols, int64_t nvalid, float scale, void* stream);

/* gf_rowmax_neg_bf16 — out[r * ldo] = bf16(-max_c x[r, c]) (exact: a maximum of bf16 values).  The per-row offset of the
 * AttentionBlock's score GEMM (VAE:326-333): written into an extra K column of the Q operand whose counterpart in the K operand
 * is 1, it makes the GEMM accumulate q.k - rowmax in fp32, so that the bf16 rounding of the scores is small where the softmax
 * weight is large (F.scaled_dot_product_attention, which the reference calls, keeps its scores in fp32).  x [rows, ncols] bf16 (ldx). */
GF_API int gf_rowmax_neg_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int64_t ncols, void* stream);

/* gf_transpose_pad — dst[c, r] = src[r, c], r >= R zero-filled up to rpad (V^T operand
 * of the AttentionBlock's P·V GEMM).                                                   */
GF_API int gf_transpose_pad(const void* src, int64_t ld_src, void* dst, int64_t R, int64_t C, int64_t rpad,
                            void* stream);

/* gf_transpose_pad_batched — `batch` transposes in one launch: matrix b = src + b*stride_src ([R, C], row pitch ld_src; heads lying side by
 * side in one [R, H*C] tensor have stride_src = C) -> dst + b*stride_dst ([C, rpad], zero columns past R); strides in elements. */
GF_API int gf_transpose_pad_batched(const void* src, int64_t ld_src, int64_t stride_src, void* dst, int64_t stride_dst, int64_t R,
                                    int64_t C, int64_t rpad, int64_t batch, void* stream);

/* gf_vae_tile_blend / gf_vae_tile_finalize — WanVideoVAE.tiled_decode's weighted tile
 * accumulation (VAE:1128-1150, build_mask VAE:1081-1100) with bf16 accumulators:
 *   values[c,t,y0+y,x0+x] += tile[t,y,x,c]*mask(y,x); weight[y0+y,x0+x] += mask(y,x)
 * then values = clamp(values/weight, -1, 1).  tile is channels-last [T,th,tw,tc>=nch] (nch = 3 RGB / 16 latent);
 * top/bottom/left/right = the tile touches that frame border (no ramp on that side).   */
GF_API int gf_vae_tile_blend(void* values, void* weight, const void* tile, int64_t nch, int64_t T, int64_t th,
                             int64_t tw, int64_t tc, int64_t H, int64_t W, int64_t y0, int64_t x0,
                             int top, int bottom, int left, int right, int64_t border_h, int64_t border_w,
                             void* stream);
/* clamp1 != 0: clamp to [-1,1] (decode); 0: no clamp (tiled_encode, VAE:1155-1203). */
GF_API int gf_vae_tile_finalize(void* values, const void* weight, int64_t planes, int64_t hw, int clamp1, void* stream);

/* gf_modulate — the reference's module-level `modulate(x, shift, scale)` (DIT:64-65) in its eager bf16 rounding sequence:
 * out = bf16(bf16(x * bf16(1 + scale)) + shift); scale / shift [dim] bf16; x, out [rows, dim] bf16.  (On the hot path the
 * modulation is fused into gf_layernorm_modulate; this entry point backs the B3 drop-in name.) */
GF_API int gf_modulate(const void* x, void* out, const void* scale, const void* shift, int64_t rows, int64_t dim,
                       int64_t x_stride, int64_t out_stride, void* stream);

/* ---- Canny-edge control signal (ControlSignalDataset_CannyEdge._generate_control_video, src/goal_force/unified_dataset.py:559-578;
 * the reference calls cv2 / controlnet_aux per frame on the host — absent here: OpenCV's published algorithm restated, "parity unpinned").
 * All images are uint8 [frames, H, W, 3] (HWC), all tables device arrays built by goal_force_amd/canny.py.
 * gf_resize_lanczos4_u8 — cv2.resize(INTER_LANCZOS4) on 8-bit pixels: xofs/yofs = floor source coordinate per destination column / row,
 *   xcoef/ycoef [n][8] = the 8 tap weights in 11-bit fixed point; replicated border; one rounding shift by 22. */
GF_API int gf_resize_lanczos4_u8(const void* src, void* dst, const int* xofs, const short* xcoef, const int* yofs, const short* ycoef,
                                 int64_t frames, int64_t H, int64_t W, int64_t Hd, int64_t Wd, void* stream);
/* gf_resize_area_u8 — cv2.resize(INTER_AREA), shrinking by a non-integer ratio: CSR tables (start [n+1], source index, fp32 weight) per
 *   axis.  mode 0: uint8 [frames,H,W,3] -> uint8 [frames,Hd,Wd,3].  mode 1: src is gf_canny_u8's state map [frames,H,W] (2 = edge -> 255,
 *   else 0) and dst the dataset's control video: bf16 [frames,Hd,Wd,3], three equal channels of x / 127.5 - 1. */
GF_API int gf_resize_area_u8(const void* src, void* dst, const int* xstart, const int* xsrc, const float* xalpha, const int* ystart,
                             const int* ysrc, const float* yalpha, int64_t frames, int64_t H, int64_t W, int64_t Hd, int64_t Wd, int mode,
                             void* stream);
/* gf_canny_u8 — cv2.Canny(img, low, high), aperture 3, L1 gradient, on uint8 [frames,H,W,3]: Sobel (replicated border), strongest
 *   channel per pixel, non-maximum suppression, double threshold, hysteresis to the fixpoint -> state uint8 [frames,H,W]
 *   (2 = edge).  mag_ws / dxy_ws: int32 [frames*H*W] scratch each; changed: one device int.  Synchronises the stream once per
 *   hysteresis pass (a dataset loader, not the sampling loop); fails if max_passes are not enough. */
GF_API int gf_canny_u8(const void* img, void* state, void* mag_ws, void* dxy_ws, int* changed, int64_t frames, int64_t H, int64_t W,
                       int low, int high, int max_passes, void* stream);

/* gf_gate_residual — `GateModule.forward(x, gate, residual)` (DIT:189-194): out = bf16(x + bf16(gate * residual)), gate [dim]
 * bf16 (batch 1), x / residual / out [rows, dim] bf16 with row strides.  (Hot path: the GEMM epilogue GF_EPI_BIAS_GATE_RESID.) */
GF_API int gf_gate_residual(const void* x, const void* gate, const void* residual, void* out, int64_t rows, int64_t dim,
                            int64_t x_stride, int64_t r_stride, int64_t out_stride, void* stream);

/* gf_rope_apply — the reference's module-level `rope_apply(x, freqs, num_heads)` (DIT:92-97) alone: adjacent pairs of every
 * head rotated by cos / sin [rows, head_dim/2] fp32.  (Hot path: fused into gf_rmsnorm_rope.) */
GF_API int gf_rope_apply(const void* x, void* out, const float* cos_tab, const float* sin_tab, int64_t rows, int64_t dim,
                         int64_t head_dim, int64_t x_stride, int64_t out_stride, void* stream);

/* ========================================================================
 * fp8 Linear — the contract of AutoWrappedLinear.fp8_linear (VRAM:115-151):
 *   scale_a = clamp(rowmax|x| / 448, min=1);  x8 = e4m3(x / (scale_a + 1e-8));  W8 = e4m3(W) (unit scale)
 *   out = bf16((x8 · W8^T) * scale_a + bias)          [torch._scaled_mm, out_dtype bf16]
 * e4m3 is OCP e4m3fn (gfx950 native; the reference halves fp8_max only for MI300's fnuz).
 * ======================================================================== */

/* gf_quant_fp8_rowscale — per-row dynamic activation quantisation (VRAM:124-137).
 * x [rows, dim] bf16 -> out8 [rows, dim] e4m3 bytes, scale [rows] fp32; dim % 8 == 0, dim <= 14336. */
GF_API int gf_quant_fp8_rowscale(const void* x, void* out8, float* scale, int64_t rows, int64_t dim,
                                 int64_t x_stride, int64_t out_stride, void* stream);

/* gf_layernorm_modulate_fp8 — gf_layernorm_modulate whose consumer is an fp8 Linear: the normalised (+affine, +modulate) bf16
 * row stays in registers, its row maximum gives scale_a and only the e4m3 bytes + scale are written — bit-identical to
 * gf_layernorm_modulate followed by gf_quant_fp8_rowscale (DIT:206-208, 225-228 feeding VRAM:124-137).
 * dim must be one of the wave-per-row widths 5120 / 4096 / 1536; other widths: call the two functions. */
GF_API int gf_layernorm_modulate_fp8(const void* x, void* out8, float* scale, const void* weight, const void* bias,
                                     const void* scale1p, const void* shift, int64_t rows, int64_t dim,
                                     int64_t x_stride, int64_t out_stride, float eps, void* stream);

/* gf_linear_vt32_fp8 — gf_linear_vt32 on the fp8_linear contract: the V projection of a self-attention under config 5 written
 * directly as attention kernel 3's V^T operand.  x8 [kv_len, K] / w8 [N, K] e4m3 bytes, x_scale [kv_len] fp32 (the tokens'
 * activation scales); bit-identical to gf_gemm_fp8 + gf_transpose_v32.  N >= 512, N % 128 == 0, K % 128 == 0. */
GF_API int gf_linear_vt32_fp8(const void* x8, int64_t ldx, const float* x_scale, const void* w8, int64_t ldw, const void* bias,
                              void* vt, int64_t kv_len, int64_t kv_pad, int64_t N, int64_t K, void* stream);

/* gf_cast_fp8 — bf16 -> e4m3 elementwise (weight.to(float8_e4m3fn), VRAM:138); n % 8 == 0. */
GF_API int gf_cast_fp8(const void* x, void* out8, int64_t n, void* stream);

/* gf_gemm_fp8 — C = epilogue((A8 · W8^T) * row_scale[m] + bias); A8 [M,K], W8 [N,K] e4m3 bytes (K contiguous),
 * fp32 accumulate on v_mfma_f32_16x16x128_f8f6f4 (2x the bf16 MFMA rate; M >= 512: the 4-wave asm K loop);
 * same epilogues / residual / gate semantics as gf_gemm_bf16 (torch._scaled_mm, VRAM:141-148).
 * K % 128 == 0, lda/ldw % 16 == 0. */
GF_API int gf_gemm_fp8(const void* A8, int64_t lda, const void* W8, int64_t ldw, const float* row_scale,
                       const void* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                       int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream);

/* ------------------------------------------------------------------------
 * gf_conv3d_padded_bf16 — CausalConv3d 3x3x3, stride 1 (VAE:33-52) of the 192- and 384-channel levels of Decoder3d / Encoder3d
 * (VAE:736-838, 517-617) as a direct convolution on the 4-wave GEMM loop (gf_conv_a4.hip).
 *   xp   : zero-bordered activation [kt - 1 + T][H + 2][W + 2][C] bf16 — kt = 3: frames 0, 1 = the two cached history frames (zeros =
 *          no history), real pixel (t, y, x) at [2 + t][1 + y][1 + x]; kt = 1 (a 3x3 convolution per frame, e.g. the resample
 *          convolution behind the nearest-exact 2x upsample, VAE:82-96): no history frames; one pixel of zeros around every frame
 *   Wm   : [N, ldw] bf16, K order (dt, dy, dx, cin), ldw >= 9 kt C
 *   out  : [T H W, ldo] bf16 = conv + bias (GF_EPI_BIAS) or resid + bf16(conv + bias) (GF_EPI_BIAS_RESID, resid [T H W, ldr])
 * C = 192 or 384; results bit-identical to gf_conv3d_bf16 on the same values.  The caller owns every buffer; no workspace.
 * gf_vae_rmsnorm_silu_padded — gf_vae_rmsnorm_silu writing into the interior of such a buffer: out_interior = &xp[f][1][1][0].
 * gf_vae_upsample2x_padded — nearest-exact 2x upsample of [T, H, W, C] into the interior of a [T][2 H + 2][2 W + 2][C] buffer. */
GF_API int gf_conv3d_padded_bf16(const void* xp, const void* Wm, int64_t ldw, const void* bias, void* out, int64_t ldo, int64_t T,
                                 int64_t H, int64_t W, int64_t C, int64_t N, int64_t kt, int epilogue, const void* resid, int64_t ldr,
                                 void* stream);
GF_API int gf_vae_rmsnorm_silu_padded(const void* x, const void* gamma, void* out_interior, int64_t T, int64_t H, int64_t W, int64_t C,
                                      int silu, void* stream);
GF_API int gf_vae_upsample2x_padded(const void* x, void* out_interior, int64_t T, int64_t H, int64_t W, int64_t C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GOALFORCE_H */
