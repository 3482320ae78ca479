/* visitron_hip.h -- C ABI of the MI355X (gfx950) encoder hot path.
 *
 * The reference (alexa/visitron) has no FFI: its hot path is a stack of Python nn.Modules calling
 * stock torch ops.  This library is the drop-in boundary BELOW those modules: every entry point
 * replaces the torch-op sequence of the cited reference lines, takes plain device pointers, sizes
 * and a hipStream_t, and returns an int status.  Contract for all entry points:
 *   - pointers are DEVICE pointers (HBM) unless named host_*; 16-byte aligned; row strides (ld*)
 *     are in ELEMENTS and must keep rows 16-byte aligned;
 *   - "bf16" buffers hold raw bfloat16 bits (uint16_t); fp32 accumulation everywhere;
 *   - nothing is allocated, freed or synchronised here: the caller owns every buffer (workspace
 *     included) and the work is enqueued on `stream` (0 = the null stream);
 *   - re-entrant: callable from several host threads, each on its own current device and stream (the reference's
 *     multi-gpu-dp mode, torch.nn.DataParallel, pretrain.py:93-94).  What the host side remembers between calls --
 *     the autotuner's shape table (vt_gemm_tune), the compute-unit count, one-time kernel attribute setup -- is kept
 *     per device behind a mutex / atomics.  The vt_debug_* hooks (forced kernel variants, trace buffer) are
 *     process-global switches for benchmarks and tests, not for concurrent production use;
 *   - return 0 (VT_OK) or a negative VT_ERR_* code; vt_error_string() names it.
 */
#ifndef VISITRON_HIP_H
#define VISITRON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vt_stream_t; /* hipStream_t */

#define VT_OK 0
#define VT_ERR_BAD_SHAPE (-1)
#define VT_ERR_BAD_ALIGN (-2)
#define VT_ERR_NULL (-3)
#define VT_ERR_UNSUPPORTED (-4)
#define VT_ERR_HIP (-5)

#define VT_ACT_NONE 0
#define VT_ACT_GELU 1 /* erf-GELU, hidden_act == "gelu" */
#define VT_ACT_TANH 2
#define VT_ACT_MUL 3 /* out = acc * R: dgrad through the erf-GELU, R = gelu'(pre-activation) saved by the forward */

const char* vt_error_string(int code);
/* ABI version of this header; bumped on any signature change. */
int vt_abi_version(void);
/* Tuning hook (benchmarks only): force the GEMM kernel variant, -1 = built-in choice, -2 = built-in choice plus the
 * tail launch (the persistent kernel's half-empty last round of tiles handed to the 128x128-tile kernel).  Process-global. */
void vt_debug_set_gemm_variant(int variant);
/* Debug: device buffer (>= 64 * 8 B per workgroup) that the persistent GEMM fills with phase timestamps; NULL = off. */
void vt_debug_set_gemm_trace(void* buf);
/* Tuning/test hook for vt_wgrad_bf16: 0 automatic (the persistent kernel where N, K are multiples of 256 and the group has
   at least 12 288 token rows -- below that the one-tile-per-workgroup kernel is 7 ... 38 % faster, round 6;
   VT_WGRAD_PERSISTENT_MIN_ROWS), 128 / 256 force the one-tile-per-workgroup kernel with that n-tile width, -8 never the
   persistent kernel, 8 the persistent kernel wherever it is eligible. */
void vt_debug_set_wgrad_kernel(int mode);
/* Autotuner result: on the calling thread's CURRENT DEVICE use kernel `variant` for linear layers of exactly this
 * shape and epilogue (filled by the host before the shape is used; one table per device, mutex-guarded).
 * kind = act | 16 (a residual / factor operand R) | 32 (the second output C2) | 64 (fp32 output) | ln_mode << 8
 * (vt_linear_ln_bf16): the out-proj with its residual and a plain dgrad of the same (M, N, K) are different entries. */
void vt_gemm_tune(int M, int N, int K, int kind, int variant);
/* The persistent GEMM kernels launch one workgroup per compute unit; with k > 0 they leave k compute units free (for the
 * collective kernels of a data-parallel step that run beside the backward).  Process-global, 0 by default. */
void vt_gemm_reserve_cus(int k);
/* Workspace of the persistent GEMM's STREAM-K REGION (kernel variants 31 / 32; 28 .. 30 run as their plain twins; round 6):
 * the output tiles a launch's workgroups cannot take in whole rounds are worked as one sequence of K-steps in equal shares;
 * a tile's parts exchange fp32 accumulators through this memory (the sum is taken in workgroup order: deterministic).
 * `base`: device memory of the calling thread's CURRENT DEVICE, 256-byte aligned, ZEROED by the caller, `bytes` >= one
 * region (vt_gemm_workspace_region_bytes()); every whole region in it serves one launch at a time, launches take the
 * regions round-robin -- so as many launches may be in flight on DIFFERENT streams as there are regions (one stream never
 * overlaps its own launches).  The library neither allocates nor frees it; base == NULL unregisters.  Without a workspace
 * variants 28 .. 32 return VT_ERR_UNSUPPORTED (the autotuner then leaves them out). */
int vt_gemm_set_workspace(void* base, int64_t bytes);
int64_t vt_gemm_workspace_region_bytes(void);
/* The workgroup finishing a tile of the stream-K region waits for the tile's other parts behind a BOUNDED wait (a grid must always drain).
 * *host_count = how many ran out of it on the current device since the last call (then cleared): non-zero = some GEMM
 * output of an earlier launch is unreliable.  Blocking (4-byte copies): call it where the host synchronises anyway. */
int vt_gemm_shared_tile_timeouts(unsigned* host_count);
/* Both counters above (vt_wgrad_turn_timeouts first, vt_gemm_shared_tile_timeouts second) read and cleared by a one-thread
 * kernel on `stream` into the DEVICE int64 pair out2 -- for callers that read something back at that point anyway (the
 * training step's row counts): no host round trip of its own (ABI 12). */
int vt_step_counters(int64_t* out2, vt_stream_t stream);
/* Attention-probability dropout (oscar/modeling_bert.py:62, nn.Dropout(attention_probs_dropout_prob)): how finely the drop
 * probability is resolved.  16 (default since ABI 12): steps of 1/65536 -- two neighbouring keys share one hash word, a 16-bit
 * field each -- so the reference's 0.1 runs as 6554/65536 = 0.100006.  8 (the form of ABI 8 .. 11): steps of 1/256 -- four
 * keys per hash word, a byte each -- so 0.1 runs as 26/256 = 0.1016 (kept probabilities scaled by 1 / (1 - 0.1016):
 * unbiased either way); the forward's dropout arithmetic halves (that kernel ~5 % faster, the B = 256 step 0.15 %).
 * Process-wide, read when a call builds its dropout state: set it before a step, never between a forward and its
 * backward.  Also VT_ATTN_DROPOUT_BITS=8 in the environment.  vt_attn_dropout_effective(p) = the probability p runs as under
 * the current setting (-1 where it is refused: p < 0 or rounding to 1). */
int vt_set_attn_dropout_bits(int bits);
int vt_get_attn_dropout_bits(void);
float vt_attn_dropout_effective(float p);
/* Weight prefetch in the encoder layer loops (ABI 12).  At a small batch a layer's GEMMs are one round of tiles whose K loop
 * fetches two K-steps ahead, and a weight matrix last read a step (or a forward) ago comes from HBM rather than the Infinity
 * Cache: FFN-down at 7 091 rows runs 45 us on warm weights and 67 us in the training step.  The loops therefore read the
 * next GEMMs' weights ahead of time and drop them -- in spare workgroups of kernels that are launched anyway (training: the
 * LayerNorm forward / the LayerNorm backward's reduction; inference: the attention kernel), so without a launch of their own.
 * training_mode: 0 off, 1 one launch per layer, 2 two launches per layer, 3 on a side stream beside the attention kernels
 * (measured slower), 4 (default) riding in the LayerNorm kernels, applied below 16 384 token rows (VT_PREFETCH_MAX_ROWS);
 * inference_mode: 0 off, 1 / 2 launches, 3 (default) riding in the attention kernel.  -1 keeps a setting.  Also
 * VT_PREFETCH_WEIGHTS / VT_PREFETCH_INFER in the environment.  Results do not depend on it (reads only). */
int vt_set_weight_prefetch(int training_mode, int inference_mode);
int vt_get_weight_prefetch(int inference);
/* Tuning hook: 4 or 8 waves per workgroup in the attention backward (default 8). Process-global. */
void vt_debug_set_attn_bwd_waves(int waves);

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ R[M,N]);  A, W, R bf16; C bf16 (out_f32 == 0) or
 * fp32.  Replaces every nn.Linear call on the path -- query/key/value oscar/modeling_bert.py:43-45
 * (packed into one [3H,H] weight), BertSelfOutput.dense :94, BertIntermediate.dense + gelu :119,
 * BertOutput.dense :120, img_embedding + location_embeds tasks/viewpoint_select/encoder.py:277-279,
 * pooler :296, mlmhead :377, token_head :381, next_action :391 -- with bias, activation and the
 * residual add fused.  K % 64 == 0 (pad K with zeros); M, N arbitrary.  If grp_rows > 0 the
 * output (and residual) row of GEMM row m is (m / grp_rows) * grp_stride + m % grp_rows, which
 * lets the region projection write rows T..T+R-1 of every sequence of the [B,S,H] buffer in
 * place of torch.cat (encoder.py:287). */
int vt_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                   const void* R, int64_t ldr, void* C, int64_t ldc, int M, int N, int K, int act,
                   int out_f32, int grp_rows, int grp_stride, vt_stream_t stream);

/* Same (same reference lines; in training also the dgrad GEMMs of loss.backward(), tasks/viewpoint_select/pretrain.py:191),
 * plus C2: optional second bf16 output saved for backward, row stride ldc2 -- gelu'(acc + bias)
 * when act == VT_ACT_GELU (what the backward of BertIntermediate multiplies by), else acc + bias --
 * and act == VT_ACT_MUL (out = acc * R).  out_f32 is a bit set (ABI 8): 1 = fp32 output; 2 = C written as FP16
 * (saturating at +-65504) instead of bf16; 4 = the residual R holds FP16 -- the pre-LayerNorm sum and the residual operand
 * of a layer that keeps fp16 copies of its residual stream (vt_layer_acts::ln1_h). */
int vt_linear_bf16_ex(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                      const void* R, int64_t ldr, void* C, int64_t ldc, void* C2, int64_t ldc2, int M, int N,
                      int K, int act, int out_f32, int grp_rows, int grp_stride, float drop_p, uint64_t drop_seed,
                      uint32_t drop_site, vt_stream_t stream);

/* The dense + dropout + residual of BertSelfOutput / BertOutput (oscar/modeling_bert.py:94,120) with the residual given as
 * a LayerNorm that was never written out (ABI 10; vt_layer_acts::ln_residual_mode):
 *   C = dropout(A W^T + bias) + ((Rv - mean[row]) * rstd[row] * gamma[col] + beta[col])
 * Rv: FP16 [M,N] (row stride ldr) = the previous sub-layer's pre-LayerNorm sum; mean / rstd fp32 [M] = what the LayerNorm
 * kernel wrote for those rows (vt_layernorm_h_bf16); gamma / beta fp32 [N] = that LayerNorm's weight and bias (16-byte
 * aligned).  C: bf16, or FP16 with out_f16 != 0.  N a multiple of 16; no row remap, no activation. */
int vt_linear_lnres_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* Rv,
                         int64_t ldr, const float* mean, const float* rstd, const float* gamma, const float* beta, void* C,
                         int64_t ldc, int M, int N, int K, int out_f16, float drop_p, uint64_t drop_seed,
                         uint32_t drop_site, vt_stream_t stream);

/* ---- dropout (nn.Dropout of BertEmbeddings / BertSelfOutput / BertOutput / the image embedding,
 * tasks/viewpoint_select/encoder.py:284, and the attention-probability dropout oscar/modeling_bert.py:62).
 * Every kernel that applies dropout takes (p, step seed, site): keep(element) is a counter-based hash of
 * (seed, site, element index), kept values are scaled by 1/(1-p), and the backward kernels recompute the
 * mask instead of storing it.  p = 0 disables it.  Sites: 8*layer + {0 attention probabilities,
 * 1 attention.output, 2 output}; 0xE0 embeddings; 0xE1 image embedding.  In vt_linear_bf16_ex the
 * dropout is applied to act(acc + bias) BEFORE the residual add, element index m * N + n. */
int vt_apply_dropout_bf16(void* x, int64_t ld, int64_t rows, int cols, float drop_p, uint64_t drop_seed,
                          uint32_t drop_site, vt_stream_t stream);
/* Test hook: out[i] = 1 if element i of the site is kept (head_index = b*nh + h for attention sites where
 * element i = q * S' + key, S' = the sequence's length rounded up to a multiple of 4, and the keep decision is byte
 * (i & 3) of the hash word of i >> 2 against p quantised to 1/256 -- the attention sites' form, ABI 8; in exact-p mode
 * (vt_set_attn_dropout_bits(16)) field (i & 1) of the hash word of i >> 1 against p quantised to 1/65536 --, else -1). */
int vt_debug_dropout_mask(uint8_t* out, int64_t n, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                          int head_index, vt_stream_t stream);

/* Fused scaled-dot-product attention, head size 64 (oscar/modeling_bert.py:47-72):
 * ctx[b,s,h*64:(h+1)*64] = softmax_k(q.k / 8 + (1 - mask[b,k]) * -10000) . v  [* head_scale[h]].
 * qkv is the packed projection output [B*S, ld_qkv] = q | k | v (each nh*64 wide).  mask is the
 * caller's raw 2-D mask as fp32 [B,S] (null = all ones) when mask_additive == 0 -- the -10000
 * arithmetic of tasks/viewpoint_select/encoder.py:238-241 then happens in the kernel -- or the
 * already-additive [B,S] bias when mask_additive == 1 (what CaptionBertEncoder.forward receives), or an additive
 * per-query bias [B,S,S] when mask_additive == 2 (the reference's 3-D attention_mask, encoder.py:226-229; forward and
 * vt_attention_probs_f32 only, the backward returns VT_ERR_UNSUPPORTED).  lse (optional, [B,nh,S]) gets
 * the natural-log log-sum-exp of the masked scores for the backward pass.
 * keep_bits (optional; training with drop_p > 0): the kernel also writes its keep decisions, one word per (32-query block,
 * key): keep_bits[((b*nh + h) * nqb + qb) * kpitch + key], bit j = keep(query 32 qb + j, key), nqb = ceil(S/32),
 * kpitch = 32 nqb (VT_KEEP_WORDS(B, nh, S) words).  The backward kernels read them back instead of re-deriving the mask
 * from the hash (a third of their vector instructions). */
#define VT_KEEP_WORDS(B, nh, S) ((int64_t)(B) * (nh) * (((S) + 31) / 32) * (((S) + 31) / 32 * 32))
int vt_attention_fwd_bf16(const void* qkv, int64_t ld_qkv, const float* mask, int mask_additive,
                          const float* head_scale, void* ctx, int64_t ld_ctx, float* lse, int B, int S,
                          int nh, int head_size, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                          uint32_t* keep_bits, vt_stream_t stream);

/* The attention probabilities the reference returns under config.output_attentions (oscar/modeling_bert.py:58-66,
 * 74-79; eval mode): probs[b,h,q,k] = exp(q.k/8 + bias[k] - lse[b,h,q]) [* head_scale[h]], fp32 [B,nh,S,S], from qkv
 * and the lse written by vt_attention_fwd_bf16 (same mask arguments). */
int vt_attention_probs_f32(const void* qkv, int64_t ld_qkv, const float* mask, int mask_additive,
                           const float* head_scale, const float* lse, float* probs, int B, int S, int nh,
                           int head_size, vt_stream_t stream);

/* Backward of vt_attention_fwd_bf16: dqkv = dq | dk | dv packed like qkv.  ctx is the forward output,
 * lse its saved log-sum-exp, delta_ws a [B,nh,S] fp32 scratch (the 4-wave kernel and the debug form 10 compute
 * rowsum(dctx*ctx) into it with a separate pass; the default 8-wave kernel forms that row constant itself from ctx).
 * Autograd of oscar/modeling_bert.py:47-72 inside loss.backward() (tasks/viewpoint_select/
 * pretrain.py:191).  S <= 256: no atomics, bitwise reproducible, dq32_ws may be NULL.  S > 256: the
 * keys are processed in blocks of 256 and dq is accumulated with fp32 atomics in dq32_ws, an fp32
 * [B*S, nh*64] scratch slab (zeroed here), then rounded to bf16.  keep_bits: the words the forward wrote (same
 * drop_p / seed / site), or NULL = recompute the mask from the hash. */
int vt_attention_bwd_bf16(const void* qkv, int64_t ld_qkv, const void* dctx, int64_t ld_d, const void* ctx,
                          int64_t ld_ctx, const float* mask, int mask_additive, const float* lse,
                          float* delta_ws, void* dqkv, int64_t ld_dqkv, float* dq32_ws, int B, int S, int nh,
                          int head_size, float drop_p, uint64_t drop_seed, uint32_t drop_site, const uint32_t* keep_bits,
                          vt_stream_t stream);

/* y = BertLayerNorm(x) over rows of H (biased variance, eps inside the sqrt); x already holds
 * dense(h) + bias + residual.  BertSelfOutput / BertOutput LayerNorm (called at
 * oscar/modeling_bert.py:94,120) and the optional image LayerNorm (encoder.py:280-281; use
 * grp_rows/grp_stride as in vt_linear_bf16).  mean / rstd: optional [M] outputs. */
int vt_layernorm_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma,
                      const float* beta, float* mean, float* rstd, int M, int H, float eps,
                      int grp_rows, int grp_stride, vt_stream_t stream);

/* The same LayerNorm over FP16 input rows (the pre-LayerNorm sums of a layer that keeps fp16 copies of its residual
 * stream, vt_layer_acts::ln1_h), written as bf16 (y) and, when y_f16 is not NULL, as fp16 too. */
int vt_layernorm_h_bf16(const void* x_f16, int64_t ldx, void* y, int64_t ldy, void* y_f16, int64_t ldyh, const float* gamma,
                        const float* beta, float* mean, float* rstd, int M, int H, float eps, vt_stream_t stream);

/* Backward of vt_layernorm_bf16 (autograd of BertLayerNorm in BertSelfOutput / BertOutput, oscar/modeling_bert.py:94,120, and
 * of the image LayerNorm, encoder.py:280-281, inside loss.backward(), pretrain.py:191): dx, and dgamma / dbeta (fp32,
 * overwritten or accumulated).  x is the
 * pre-LayerNorm input (statistics are recomputed).  partial_ws: fp32 scratch of 1024 * 2 * H floats. */
int vt_layernorm_bwd_bf16(const void* x, int64_t ldx, const void* dy, int64_t ldy, const float* gamma,
                          void* dx, int64_t lddx, float* dgamma, float* dbeta, float* partial_ws, int M, int H,
                          float eps, int accumulate, void* dx_dropped, int64_t lddxd, float drop_p,
                          uint64_t drop_seed, uint32_t drop_site, vt_stream_t stream);
/* (dx_dropped, optional: dx * mask / (1-p) of the given site = the gradient of the dense output that was
 * dropped out before the residual add.) */
/* ... with the pre-LayerNorm input x held as FP16 (backward of vt_layernorm_h_bf16). */
int vt_layernorm_bwd_h_bf16(const void* x_f16, int64_t ldx, const void* dy, int64_t ldy, const float* gamma,
                            void* dx, int64_t lddx, float* dgamma, float* dbeta, float* partial_ws, int M, int H,
                            float eps, int accumulate, void* dx_dropped, int64_t lddxd, float drop_p,
                            uint64_t drop_seed, uint32_t drop_site, vt_stream_t stream);

/* out = g * d, bf16, n elements (n % 8 == 0), d = saved gelu' values: the dGELU of the MLM-head transform
 * (BertOnlyMLMHead's dense + gelu, constructed at encoder.py:322, in loss.backward(), pretrain.py:191). */
int vt_dgelu_mul_bf16(const void* g, const void* h, void* out, int64_t n, vt_stream_t stream);

/* BertEmbeddings (called at tasks/viewpoint_select/encoder.py:267-269): y[b*S + t, :] =
 * LayerNorm(word[ids[b,t]] + pos[pos_ids[b,t] or t] + type[type_ids[b,t] or 0]); tables fp32,
 * y bf16 rows of the [B,S,H] buffer (S >= T).  err_flag (device int, optional) is set to 1 if an
 * index is out of range. */
int vt_embed_layernorm(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids,
                       const float* word, const float* pos, const float* type, const float* gamma,
                       const float* beta, void* y, int64_t ldy, int B, int T, int S, int H,
                       int n_word, int n_pos, int n_type, float eps, int* err_flag, float drop_p,
                       uint64_t drop_seed, vt_stream_t stream);

/* Backward of vt_embed_layernorm (autograd of the BertEmbeddings call, encoder.py:267-269, in loss.backward(),
 * pretrain.py:191): de[B*T,H] fp32 = gradient w.r.t. (word + pos + type) per token (for
 * the three table scatter-adds), dgamma / dbeta of the embedding LayerNorm.  g: gradient rows b*S+t
 * of the [B,S,H] bf16 buffer.  partial_ws: 1024 * 2 * H floats. */
int vt_embed_layernorm_bwd(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids,
                           const float* word, const float* pos, const float* type, const float* gamma,
                           const void* g, int64_t ldg, float* de, float* dgamma, float* dbeta, float* partial_ws,
                           int B, int T, int S, int H, int n_word, int n_pos, int n_type, float eps,
                           int accumulate, float drop_p, uint64_t drop_seed, vt_stream_t stream);

/* What the host must know about a pretrain batch before it can size the step (pretrain.py:157-193 runs every position; this
 * path runs the heads on the supervised rows and the encoder on the rows with a non-zero mask).  counts[5] = {err_flag[0],
 * #(labels != -1), #(token_labels != -1), #(mask != 0), bad}, bad = the batch does not qualify for the compacted layout (a
 * mask value other than 0 / 1, a [CLS] or a supervised position with mask 0).  labels / token_labels / mask / err_flag may
 * each be NULL.  counts must be zero on entry; tile_counts = 3 * ceil(B*S / 1024) ints of scratch that vt_batch_row_lists
 * reads back (one workgroup per tile of 1024 positions).  One launch, one 40-byte read-back. */
int vt_batch_row_counts(const int64_t* labels, const int64_t* token_labels, const float* mask, const int32_t* err_flag, int B,
                        int S, int64_t* counts, int32_t* tile_counts, vt_stream_t stream);
/* The row lists to the sizes vt_batch_row_counts reported (n_w, n_t, n_keep: nothing is written past them; ascending): idx_w / idx_t = positions with a label / token label;
 * index = positions with a non-zero mask, inverse[position] = its rank among them or -1, start / length [B] = each sequence's
 * first compact row and number of kept rows (needs every position b*S kept); tile_counts as vt_batch_row_counts left them
 * (same labels / token_labels / mask).  Two launches. */
int vt_batch_row_lists(const int64_t* labels, const int64_t* token_labels, const float* mask, int B, int S, int64_t n_w,
                       int64_t n_t, int64_t n_keep, const int32_t* tile_counts, int64_t* idx_w, int64_t* idx_t, int64_t* index,
                       int64_t* inverse, int32_t* start, int32_t* length, vt_stream_t stream);

/* The action head's loss, accuracy and gradient (NextActionPrediction = Linear + LogSoftmax, encoder.py:142-151, under
 * CrossEntropyLoss(ignore_index=-1), :387-391, which applies log_softmax again): logits fp32 [B, >= A] (A <= 64),
 * next_action int64 [B] (-1 = ignored) -> loss_acc[0] = -mean over valid rows of log_softmax(log_softmax(z))[y],
 * loss_acc[1] = #(argmax == y) / B, dlogits bf16 [B, Ap] = d(grad_scale * loss)/dz (columns A..Ap-1 zero).  One launch. */
int vt_action_head_f32(const float* logits, int64_t ld, const int64_t* next_action, int B, int A, float grad_scale, void* dlogits,
                       int64_t ldd, int Ap, float* loss_acc, vt_stream_t stream);

/* y[M,N] (bf16) = x[M,K] w[N,K]^T for a LONG reduction and few output tiles (the MLM decoder's dgrad inside loss.backward(),
 * pretrain.py:191: d(transform output) = d(logits)[Ml, 30528] . W_dec[30528, 768] -- 51 tiles of 256 x 256): the K-steps are
 * split over ksplit copies of the tile list, each writing an fp32 plane of ws (ksplit * M * N floats), and summed.  No
 * bias / activation / residual.  K % 64 == 0, N % 4 == 0, 2 <= ksplit <= K / 64. */
int vt_linear_splitk_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, float* ws, int M, int N,
                          int K, int ksplit, vt_stream_t stream);

/* The table gradient of an embedding lookup (BertEmbeddings' three nn.Embedding backward passes inside loss.backward(),
 * tasks/viewpoint_select/pretrain.py:191; torch: one float atomic per element and row): grad[id, :] += sum of the rows of
 * `de` that looked `id` up.  The caller passes the ids stably sorted (sorted_ids, int32: a radix sort over half the key bytes)
 * with the sort's permutation (perm: sorted
 * position -> row of de); every run of equal ids is added by one workgroup in the rows' original order -- no atomics,
 * bitwise reproducible.  Rows whose id equals skip_id (nn.Embedding's padding_idx; -1 = none) contribute nothing.  A run is
 * cut at the multiples of 32 of the sorted order: a token repeated thousands of times ([MASK]) is added by many waves into
 * `scratch` (n * H floats, only the rows of such segments are touched) and joined in segment order by a second launch;
 * flag: one int of scratch. */
int vt_embed_table_grad(const int32_t* sorted_ids, const int64_t* perm, const float* de, int64_t ld_de, float* grad,
                        int64_t ld_grad, int64_t n, int H, int64_t n_rows_table, int64_t skip_id, float* scratch, int32_t* flag,
                        vt_stream_t stream);

/* Fused AdamW over a flat fp32 slab of n parameters (n % 4 == 0), the pytorch-transformers rule of
 * tasks/viewpoint_select/pretrain.py:128-130: m,v moments; p -= step_size * m / (sqrt(v) + eps) with
 * step_size = lr * sqrt(1 - b2^t) / (1 - b1^t) supplied by the host; then p -= lr * wd * p.  g is
 * multiplied by grad_scale first.  p_bf16 (optional) receives the refreshed bf16 working copy. */
int vt_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                  float step_size, float b1, float b2, float eps, float wd, float grad_scale,
                  vt_stream_t stream);
/* The same with the gradients given as bf16: the data-parallel step all-reduces a bf16 copy of the gradient slab (half
 * the bytes of the reference's fp32 DDP buckets, pretrain.py:96-102,191, over xGMI); moments and master weights stay fp32. */
int vt_adamw_flat_g16(float* p, const void* g_bf16, float* m, float* v, void* p_bf16, int64_t n, float lr, float step_size,
                      float b1, float b2, float eps, float wd, float grad_scale, vt_stream_t stream);
/* out[r, 64 h + d] = x[r, 64 h + d] * head_scale[h] (bf16 rows of nh * 64 columns): the reference's head_mask
 * (oscar/modeling_bert.py:65-66) on a context tensor or, on the way back, on its gradient (training path). */
int vt_scale_heads_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int nh, const float* head_scale,
                        vt_stream_t stream);
/* dst_bf16[i] = bf16(src[i] * scale), n % 8 == 0: the communication copy of a gradient-slab range. */
int vt_cast_f32_to_bf16(const float* src, void* dst_bf16, int64_t n, float scale, vt_stream_t stream);

/* Fused softmax cross-entropy rows for the MLM head (tasks/viewpoint_select/encoder.py:387-389, argmax
 * :399, and the criterion's backward): loss_row[r] = logsumexp(z[r, :V]) - z[r, y[r]], amax[r] =
 * argmax(z[r, :V]) (first index on ties), dz[r, :Vpad] = bf16((softmax(z[r]) - onehot(y[r])) * scale),
 * columns V..Vpad-1 zero.  z fp32 [rows, ldz], labels must be valid (0 <= y < V).  dz == NULL: loss and argmax only
 * (the inference path: the reference's 7-tuple needs nothing else of the 30522-wide logits). */
int vt_ce_softmax_rows(const float* z, int64_t ldz, const int64_t* y, float* loss_row, int64_t* amax, void* dz,
                       int64_t lddz, int64_t rows, int V, int Vpad, float scale, vt_stream_t stream);

/* The masked-region-token head's loss (tasks/viewpoint_select/encoder.py:323-326, 380-385): token_head ends in a
 * Softmax and the criterion applies log-softmax again.  Per supervised row of logits z [rows, V] (V <= 2048):
 * loss_row = logsumexp(softmax(z)) - softmax(z)[y], amax = argmax, dz (bf16 [rows, Vpad], zero past V) =
 * d(scale * loss_row)/dz (dz == NULL: loss and argmax only). */
int vt_ce_double_softmax_rows(const float* z, int64_t ldz, const int64_t* y, float* loss_row, int64_t* amax,
                              void* dz, int64_t lddz, int64_t rows, int V, int Vpad, float scale,
                              vt_stream_t stream);

/* ---- rollout caller around the trunk: tasks/viewpoint_select/agent_models.py ---- */

/* One LSTM time step with torch.nn.LSTM / nn.LSTMCell arithmetic (gate order i, f, g, o), replacing the recurrent
 * half of OscarEncoder.forward's nn.LSTM over the packed trunk output (agent_models.py:285-301) and the
 * nn.LSTMCell of AttnDecoderLSTM.forward (:416):  gates = xproj + h_prev . w_hh^T, c <- sig(f) c + sig(i) tanh(g)
 * (in place), h_out = sig(o) tanh(c).  xproj fp32 rows (row b at xproj + b*ldx, 4*hs wide) = x . W_ih^T + b_ih +
 * b_hh from vt_linear_bf16_ex; w_hh bf16 [4*hs, hs]; hs a multiple of 128; h_out must not alias h_prev.
 * lengths (optional int32 [B]) gives pack_padded_sequence semantics: a row with t >= lengths[b] keeps its state
 * and writes zeros to its seq_out position.  seq_out (optional): row b at seq_out + b*ld_seq, hs wide. */
int vt_lstm_step_f32(const float* xproj, int64_t ldx, const float* h_prev, float* h_out, float* c, const void* w_hh,
                     const int32_t* lengths, float* seq_out, int64_t ld_seq, int B, int hs, int t,
                     vt_stream_t stream);

/* The whole recurrence of one nn.LSTM direction over a padded batch (agent_models.py:285-301): T launches of the
 * step above issued from one call.  xproj fp32 [B, S, 4*hs] (element strides ldx_b / ldx_t); h2 = two fp32 [B, hs]
 * buffers (h2[0] holds the initial state, the final state is left in h2[0]); c fp32 [B, hs] in place; seq_out fp32
 * (strides lds_b / lds_t, hs wide per position; optional).  reverse != 0 walks t = T-1 .. 0 (the second direction of
 * a bidirectional LSTM: with the packed rule each row starts at its own last position). */
int vt_lstm_sequence_f32(const float* xproj, int64_t ldx_b, int64_t ldx_t, float* h2_0, float* h2_1, float* c,
                         const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b, int64_t lds_t,
                         int B, int hs, int T, int reverse, vt_stream_t stream);

/* The same recurrence (padded layout: row_start == NULL, xproj row (b, t) at b * ldx_b + t * ldx_t; compacted layout:
 * row (b, t) at (row_start[b] + t) * ldx_t) as ONE persistent launch: hs / 16 workgroups stay resident for all T steps
 * with their rows of W_hh and their cell / hidden state in registers and exchange the hidden state per step through `ws`
 * (write-through stores, one agent-scope arrival counter, bounded polls).  h / c: [B, hs] fp32, initial state in, final
 * state out.  ws: device scratch of vt_lstm_sequence_persistent_ws_bytes(B, hs) bytes, 16-byte aligned; its first 256
 * bytes are zeroed by this call; after completion the unsigned word at ws + 4 is non-zero if a workgroup ran out of
 * its bounded wait (not all workgroups were resident together): h / c are then untouched and the caller falls back to
 * vt_lstm_sequence_f32.  Returns VT_ERR_UNSUPPORTED outside B <= 64, hs in {128, 256, 512, 1024}: use the step form. */
int64_t vt_lstm_sequence_persistent_ws_bytes(int B, int hs);
int vt_lstm_sequence_persistent_f32(const float* xproj, int64_t ldx_b, int64_t ldx_t, const int32_t* row_start, float* h,
                                    float* c, const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b,
                                    int64_t lds_t, int B, int hs, int T, int reverse, void* ws, int64_t ws_bytes,
                                    vt_stream_t stream);
/* vt_lstm_sequence_f32 over COMPACTED input projections: xproj holds only the rows below each sequence's length, row of
 * (b, t) = row_start[b] + t (row stride ldx_row) -- what pack_padded_sequence feeds the LSTM (agent_models.py:285-287). */
int vt_lstm_sequence_rows_f32(const float* xproj, int64_t ldx_row, const int32_t* row_start, float* h2_0, float* h2_1,
                              float* c, const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b,
                              int64_t lds_t, int B, int hs, int T, int reverse, vt_stream_t stream);

/* out[M,N] (fp32) = act([x0 | x1] . W^T + bias) for a handful of rows: the dense layers of the decoder step
 * (AttnDecoderLSTM.forward, agent_models.py:406-425; SoftDotAttention.linear_in / linear_out :336, :354-355 with its
 * torch.cat folded in).  x0 [M,K0], x1 [M,K1] (optional) fp32 with K0, K1 multiples of 4; W bf16 [N, Kpad] zero-padded
 * past K0 + K1, Kpad a multiple of 32; act 0 = none, 2 = tanh. */
int vt_skinny_linear_f32(const float* x0, int64_t ld0, int K0, const float* x1, int64_t ld1, int K1, const void* w,
                         int64_t ldw, const float* bias, float* out, int64_t ldo, int M, int N, int Kpad, int act,
                         vt_stream_t stream);

/* SoftDotAttention.forward after linear_in (agent_models.py:336-349): attn[b,l] = context[b,l,:] . target[b,:];
 * mask (uint8 [B,L], nonzero = masked) -> -inf; softmax over l; weighted[b,:] = sum_l p[l] context[b,l,:].
 * context fp32 with element strides ld_batch / ld_row.  weighted (optional) fp32 [B,D]; attn (optional) fp32 [B,L]
 * receives the probabilities (output_prob != 0) or the masked logits (the reference's `logit` aliases the masked
 * tensor, :339-343). */
int vt_softdot_attention_f32(const float* target, const float* context, int64_t ld_batch, int64_t ld_row,
                             const uint8_t* mask, float* weighted, float* attn, int B, int L, int D, int output_prob,
                             vt_stream_t stream);

/* ---- rollout training: the gradients the reference gets from autograd through nn.LSTM / nn.LSTMCell / torch.bmm +
 * Softmax when agent.py:493-518 back-propagates the rollout loss through OscarEncoder and AttnDecoderLSTM ---- */

/* vt_lstm_step_f32 that also saves what the backward step needs: sv_gates fp32 [B, 4*hs] (the activated gates i, f, g,
 * o), sv_c fp32 [B, hs] (the cell state before the step), sv_h bf16 [B, hs] (the hidden state before the step); rows of
 * an inactive position (t >= lengths[b]) are left untouched.  Any of the three may be NULL. */
int vt_lstm_step_train_f32(const float* xproj, int64_t ldx, const float* h_prev, float* h_out, float* c, const void* w_hh,
                           const int32_t* lengths, float* seq_out, int64_t ld_seq, int B, int hs, int t, float* sv_gates,
                           float* sv_c, void* sv_h, vt_stream_t stream);

/* vt_lstm_sequence_f32 with the saves laid out like the padded sequence: sv_gates [B, S_sv, 4*hs] fp32, sv_c [B, S_sv, hs]
 * fp32, sv_h [B, S_sv, hs] bf16 (S_sv >= T; the caller zeroes sv_h: it is the X operand of the W_hh gradient GEMM). */
int vt_lstm_sequence_train_f32(const float* xproj, int64_t ldx_b, int64_t ldx_t, float* h2_0, float* h2_1, float* c,
                               const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b, int64_t lds_t,
                               int B, int hs, int T, int reverse, float* sv_gates, float* sv_c, void* sv_h, int64_t S_sv,
                               vt_stream_t stream);

/* One step of back-propagation through time at position t.  dh = d_out[b] (optional, row stride ld_dout) + the gradient
 * arriving through h: dg_next[b] . W_hh (dg_next = the bf16 gate gradients of position t_next, the step that consumed
 * h_t; w_hh_t = W_hh transposed, bf16 [hs, 4*hs]) for rows active at t_next, else dh_final[b] (NULL = zeros; also taken
 * by every row when dg_next is NULL).  dc fp32 [B, hs] in place: in = gradient of the cell state after this step,
 * out = before it.  dg_out bf16 (row stride ld_dg, 4*hs wide) receives the pre-activation gate gradients, zeros for
 * rows with t >= lengths[b]; dg_out_f32: optional fp32 copy.  dg_out must not alias dg_next. */
int vt_lstm_step_bwd_f32(const void* dg_next, int64_t ld_dgn, const void* w_hh_t, const float* dh_final, const float* d_out,
                         int64_t ld_dout, float* dc, const float* sv_gates, int64_t ld_svg, const float* sv_c,
                         int64_t ld_svc, void* dg_out, int64_t ld_dg, float* dg_out_f32, int64_t ld_dgf,
                         const int32_t* lengths, int B, int hs, int t, int t_next, vt_stream_t stream);

/* The T steps of vt_lstm_sequence_train_f32 backwards, T launches from one call.  d_seq_out: gradient of the padded
 * output (strides ldd_b / ldd_t, optional); dh_final / dc: gradients of the final hidden / cell state ([B, hs]; dc is
 * the running value afterwards = gradient of the initial cell state); dgates bf16 [B, S_sv, 4*hs], zeroed by the caller
 * at positions >= T: afterwards dW_hh = dgates^T . sv_h, dW_ih = dgates^T . x and db via vt_wgrad_bf16, dx = dgates . W_ih
 * via vt_linear_bf16_ex. */
int vt_lstm_sequence_bwd_f32(const float* d_seq_out, int64_t ldd_b, int64_t ldd_t, const float* dh_final, float* dc,
                             const void* w_hh_t, const int32_t* lengths, const float* sv_gates, const float* sv_c,
                             void* dgates, int64_t S_sv, int B, int hs, int T, int reverse, vt_stream_t stream);

/* Gradient of vt_softdot_attention_f32 (autograd through agent_models.py:336-352): d_weighted fp32 [B,D] and / or d_attn
 * fp32 [B,L] (gradient of the returned probabilities when output_prob != 0, of the returned masked logits otherwise; a
 * masked key's logit gradient is dropped like masked_fill_ does) -> d_target fp32 [B,D], d_context fp32 [B,L,D]
 * contiguous (optional).  The probabilities are recomputed from target and context. */
int vt_softdot_attention_bwd_f32(const float* target, const float* context, int64_t ld_batch, int64_t ld_row,
                                 const uint8_t* mask, const float* d_weighted, const float* d_attn, float* d_target,
                                 float* d_context, int B, int L, int D, int output_prob, vt_stream_t stream);

/* The same gradient for a long context (the decoder's attention over the instruction, L = 511) spread over keys in two
 * launches.  ws: vt_softdot_attention_bwd_split_ws_floats(B, L, D) floats of device scratch; afterwards its tail
 * [B, chunks, D] (chunks = ceil(L / 32), starting 2*B*L floats in) holds partial d_target sums that the caller adds up over
 * the chunk axis (no atomics).  d_context as above (optional). */
int64_t vt_softdot_attention_bwd_split_ws_floats(int B, int L, int D);
int vt_softdot_attention_bwd_split_f32(const float* target, const float* context, int64_t ld_batch, int64_t ld_row,
                                       const uint8_t* mask, const float* d_weighted, const float* d_attn, float* d_context,
                                       float* ws, int B, int L, int D, int output_prob, vt_stream_t stream);

/* out[c, r] = in[r, c] (bf16; R, C multiples of 8): refreshes the transposed weight copies (W^T of every nn.Linear on the
 * path, oscar/modeling_bert.py:43-45,94,119,120) that the
 * dgrad GEMMs consume (vt_layer_weights_t). */
int vt_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, vt_stream_t stream);

/* vt_transpose_bf16 for n matrices in one launch (arrays of n entries each): the per-step refresh of all encoder
 * layers' and heads' transposed weight copies after optimizer.step() (pretrain.py:192). */
int vt_transpose_batch_bf16(const void* const* in, const int64_t* ldi, void* const* out, const int64_t* ldo, const int* R,
                            const int* C, int n, vt_stream_t stream);

/* out[row, :] = bf16([s0[row, 0:d0] | s1[row, 0:d1] | zeros to kpad]) -- builds the K-concatenated
 * operand that turns img_embedding(img_feats) + location_embeds(loc) (encoder.py:277-279) into
 * one GEMM. */
int vt_pack_concat_bf16(const float* s0, int d0, const float* s1, int d1, void* out, int kpad,
                        int64_t rows, vt_stream_t stream);

/* The per-key attention mask [B, S] as the attention kernels take it: out = float(mask) - rowmax(float(mask)) + 1, fp32
 * contiguous.  (1 - m) * -10000 (encoder.py:238-241) then differs by one constant per sequence, which the softmax over the
 * keys does not see (oscar/modeling_bert.py:55-58); a 0/1 mask with a kept key is unchanged bit for bit, the rollout
 * caller's inverted uint8 mask (254 / 255, agent_models.py:267) becomes 0 / 1.  kind: 0 float32, 1 int64, 2 int32,
 * 3 one byte (bool / uint8); ldm = row pitch of mask in elements. */
int vt_center_mask(const void* mask, int kind, int64_t ldm, float* out, int B, int S, vt_stream_t stream);

/* Weight (and bias) gradients of nn.Linear layers -- the wgrad half of loss.backward()
 * (tasks/viewpoint_select/pretrain.py:191):  dW[N,K] (+)= dY[M,N]^T . X[M,K],  db[N] (+)= colsum(dY).
 * dY, X bf16 row-major over the same M token rows; dW, db fp32.  Up to 8 problems sharing M are
 * computed by ONE grouped launch (e.g. the four matrices of an encoder layer).  K % 4 == 0,
 * M * ld * 2 < 2^31 bytes per operand. */
typedef struct vt_wgrad_problem {
  const void* dY; int64_t ldy;
  const void* X;  int64_t ldx;
  float* dW;      int64_t ldw;
  float* db;      /* or NULL */
  int N, K;
  int accumulate; /* 0: overwrite, 1: add into dW / db */
} vt_wgrad_problem;
int vt_wgrad_bf16(const vt_wgrad_problem* problems, int nprob, int M, vt_stream_t stream);
/* The persistent weight-gradient kernel adds a tile's partial sums in ticket order behind a BOUNDED wait (a grid must
 * always drain).  *host_count = how many workgroups of the current device ran out of that wait since the last call
 * (then cleared); non-zero means some dW of an earlier launch is unreliable.  Blocking (copies 4 bytes device -> host):
 * call it where the host synchronises anyway, e.g. once per training step. */
int vt_wgrad_turn_timeouts(unsigned* host_count);

/* ---- fp32 parity path -------------------------------------------------------------------------------------------
 * The reference computes in fp32 (tasks/viewpoint_select/encoder.py:238-240; no AMP anywhere) and north_star asks for
 * logits within 1e-3 of it.  These entry points keep every operand, activation and accumulation in fp32 (matrix
 * products on the exact-fp32 MFMA, v_mfma_f32_32x32x2_f32); visitron_amd.set_precision(model, "fp32") routes a
 * model's forward through them.  A correctness mode: plainly tiled, not tuned.
 *
 * vt_linear_f32: out = act(alpha * a . op(w) + bias) (+ residual), fp32 everywhere.  a [M, K] (row stride lda);
 *   w_is_kn == 0: w is nn.Linear.weight [N, K] (oscar/modeling_bert.py:43-45, :94, :119, :120; encoder.py:277-279, :296,
 *   :377-391); w_is_kn == 1: w is [K, N].  act: VT_ACT_NONE / VT_ACT_GELU (erf) / VT_ACT_TANH.  grp_rows / grp_stride: output
 *   row remap as in vt_linear_bf16_ex.  Any K (tails are zero-filled); 16-byte loads when bases / strides allow. */
int vt_linear_f32(const float* a, int64_t lda, const float* w, int64_t ldw, int w_is_kn, const float* bias,
                  const float* residual, int64_t ldr, float* out, int64_t ldc, int M, int N, int K, int act, float alpha,
                  int grp_rows, int grp_stride, vt_stream_t stream);
/* The same product batched over (batch, head) with element strides per operand: torch.matmul(q, k^T) and
 * torch.matmul(probs, v) of oscar/modeling_bert.py:52,68 straight on the packed [B*S, 3H] projection buffer. */
int vt_bmm_f32(const float* a, int64_t lda, int64_t a_stride_b, int64_t a_stride_h, const float* w, int64_t ldw,
               int64_t w_stride_b, int64_t w_stride_h, int w_is_kn, float* out, int64_t ldc, int64_t c_stride_b,
               int64_t c_stride_h, int M, int N, int K, float alpha, int batch, int heads, vt_stream_t stream);
/* In place on x [rows, cols] fp32: x * scale + mask -> Softmax(dim=-1) -> * head_scale (oscar/modeling_bert.py:53-66).
 * Row r = (b * nh + h) * S + q.  mask_mode: -1 none; 0 raw mask [B, cols] -> (1 - m) * -10000 (encoder.py:238-241);
 * 1 additive [B, cols]; 2 additive per query [B, S, cols].  With nh = S = 1 and no mask: a plain row softmax (the token
 * head's nn.Softmax, encoder.py:323-326). */
int vt_softmax_rows_f32(float* x, int64_t ld, int64_t rows, int cols, float scale, const float* mask, int mask_mode,
                        const float* head_scale, int nh, int S, vt_stream_t stream);
/* BertLayerNorm over fp32 or bf16 rows into fp32 or bf16 rows (statistics in fp32).  H % 4 == 0, H <= 4096. */
int vt_layernorm_rows(const void* x, int64_t ldx, int x_is_f32, void* y, int64_t ldy, int y_is_f32, const float* gamma,
                      const float* beta, int64_t M, int H, float eps, int grp_rows, int grp_stride, vt_stream_t stream);
/* BertEmbeddings (encoder.py:267-269) with fp32 output rows b*S + t; arguments as vt_embed_layernorm. */
int vt_embed_layernorm_f32(const int64_t* input_ids, const int64_t* token_type_ids, const int64_t* position_ids,
                           const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                           float* out, int64_t ld_out, int B, int T, int S, int H, int n_word, int n_pos, int n_type,
                           float eps, int* err_flag, vt_stream_t stream);

/* ---- pretrain input preparation on the device ---------------------------------------------------
 * PretrainDataset._mask_tokens (tasks/viewpoint_select/data_loader_pretrain.py:549-613) over n = B*T tokens: BERT's
 * 15 % / 80-10-10 rule with the random draws handed in (u_* uniform [0,1) fp32, random_words int64), forced masking of
 * the token-class positions (token_classes != -1; NULL when masked_token_prediction is off), special tokens never
 * masked (special_mask uint8), attention_mask = id != pad_id.  Outputs int64 [n]. */
int vt_mask_tokens(const int64_t* input_ids, const uint8_t* special_mask, const int64_t* token_classes, const float* u_mask,
                   const float* u_replace, const float* u_random, const int64_t* random_words, int64_t* out_ids,
                   int64_t* labels, int64_t* attention_mask, int64_t n, int64_t pad_id, int64_t mask_id,
                   float mlm_probability, vt_stream_t stream);
/* The tail of PretrainDataset._preprocess_item (data_loader_pretrain.py:627-633, 654-712) for a batch: items keep their
 * LAST R region rows of region_counts[b] (<= R_in) or are zero-padded to R with attention mask 0; loc_out rows are
 * _static_loc_embeddings[current_view[b]][region_view_ids[b, row]] (loc_table fp32 [36, 36, 128], :25-49); labels /
 * token labels get -1 on every region position; the text parts [B, T] are copied in front.  Outputs: feats_out
 * [B, R, D], loc_out [B, R, 128] fp32; labels_out / mask_out / token_labels_out int64 [B, T + R] (token labels NULL
 * together with text_token_classes). */
int vt_assemble_regions(const float* img_feats, const int64_t* region_counts, const int64_t* region_view_ids,
                        const int64_t* current_view, const float* loc_table, const int64_t* text_labels,
                        const int64_t* text_mask, const int64_t* text_token_classes, float* feats_out, float* loc_out,
                        int64_t* labels_out, int64_t* mask_out, int64_t* token_labels_out, int B, int T, int R, int R_in,
                        int D, vt_stream_t stream);

/* ---- whole encoder stack: CaptionBertEncoder.forward, oscar/modeling_bert.py:140-169 ---------- */
typedef struct vt_layer_weights {
  const void* w_qkv;  const float* b_qkv;   /* [3H,H] bf16 = query|key|value weights, [3H] */
  const void* w_ao;   const float* b_ao;    /* attention.output.dense [H,H] */
  const float* ln1_g; const float* ln1_b;   /* attention.output.LayerNorm */
  const void* w_in;   const float* b_in;    /* intermediate.dense [I,H] */
  const void* w_out;  const float* b_out;   /* output.dense [H,I] */
  const float* ln2_g; const float* ln2_b;   /* output.LayerNorm */
} vt_layer_weights;

/* Per-layer activation buffers.  Inference: every layer may point at the same scratch (and `out`
 * may alias the layer input).  Training: distinct buffers per layer are what backward reads. */
typedef struct vt_layer_acts {
  void* qkv;       /* [M,3H] bf16 */
  void* ctx;       /* [M,H]  bf16 attention context */
  void* attn_pre;  /* [M,H]  bf16 dense(ctx)+bias+x (pre-LayerNorm) */
  void* attn_out;  /* [M,H]  bf16 LayerNorm output */
  void* mid_pre;   /* [M,I]  bf16 gelu'(intermediate pre-activation), saved for backward (training) or NULL */
  void* mid;       /* [M,I]  bf16 gelu(intermediate) */
  void* out_pre;   /* [M,H]  bf16 dense(mid)+bias+attn_out */
  void* out;       /* [M,H]  bf16 layer output */
  float* lse;      /* [B,nh,S] or null */
  float* ln1_mean; float* ln1_rstd; float* ln2_mean; float* ln2_rstd; /* [M] or null */
  uint32_t* keep_bits; /* VT_KEEP_WORDS(B, nh, S) words or null: the attention dropout's keep decisions (see vt_attention_fwd_bf16) */
  /* fp16 copies of the two LayerNorm outputs, [M,H] each, or both null (ABI 8).  Present: the layer keeps its residual
   * stream at 11 significant bits -- attn_pre / out_pre then hold FP16 (same bytes), each LayerNorm writes its output as
   * bf16 (attn_out / out: the next GEMM's operand, the backward's) and as fp16 (ln1_h / ln2_h: the next sub-layer's residual
   * add reads this copy).  Transient: every layer may point at the same two buffers (ln2_h of layer l is read by layer
   * l + 1's first residual add and overwritten after it).  The last layer's ln2_h is the encoder output at fp16 precision. */
  void* ln1_h; void* ln2_h;
  /* ABI 10.  1: attn_pre / out_pre hold FP16 as above, but the LayerNorm outputs' fp16 copies are never written (ln1_h /
   * ln2_h are ignored): each residual add reconstructs LayerNorm(v) from the fp16 sum v, the row statistics the LayerNorm
   * kernel wrote (ln1_mean / ln1_rstd / ln2_mean / ln2_rstd: REQUIRED in this mode) and that LayerNorm's weight and bias
   * -- (v - mean) * rstd * gamma + beta in the GEMM epilogue (BertSelfOutput / BertOutput.forward, oscar/modeling_bert.py:
   * 94,120: LayerNorm(dense(h) + input), `input` being the previous LayerNorm's output).  The LayerNorm kernel then writes 4
   * bytes per element instead of 6 and the residual branch is not rounded to fp16 a second time.  0: as described above. */
  int32_t ln_residual_mode; int32_t reserved0;
} vt_layer_acts;

/* x: [B*S, H] bf16 embedding output (layer-0 input).  head_scale: [L, nh] fp32 or null.
 * The output of the last layer is acts[L-1].out. */
int vt_encoder_forward_bf16(const vt_layer_weights* layers, const vt_layer_acts* acts, int num_layers,
                            const void* x, const float* mask, int mask_additive, const float* head_scale,
                            int B, int S, int H, int nh, int I, float ln_eps, float p_hidden, float p_attn,
                            uint64_t drop_seed, vt_stream_t stream);

/* ---- the encoder stack in eval mode with its LayerNorms DEFERRED (the inference path) -----------------------------
 * Same arithmetic as vt_encoder_forward_bf16 (CaptionBertEncoder.forward, oscar/modeling_bert.py:140-169; BertSelfOutput /
 * BertOutput: LayerNorm(dense(h) + input), :94,120) with no LayerNorm pass and a residual stream that is not rounded to
 * bf16 twice per sub-layer: between two sub-layers the stream is the PRE-LayerNorm sum v, held as fp16 [M,H] (11
 * significant bits; saturating at +-65504) + a bf16 copy (the next GEMM's A operand) + partial row statistics of the
 * unrounded fp32 sums, stats[p][row] = (sum, sum of squares) over columns 128 p .. 128 p + 127 (p < H / 128, rows per slice =
 * stat_rows >= M, even).  LayerNorm(v) is applied where v is consumed:
 *   ln_mode 1 (projection of LN(v): query|key|value :43-45, BertIntermediate.dense :119)
 *       C = act(rstd_r * (A W'^T - mean_r * colv) + bias),  A = bf16 copy of v, W' = W * gamma (per input column),
 *       colv[n] = sum_k W'[n,k], bias[n] = sum_k W[n,k] beta[k] + b[n]   -- LN(v) W^T + b, normalisation on the accumulator;
 *   ln_mode 2 (dense + residual whose residual is LN(v): BertSelfOutput.dense :94, BertOutput.dense :120)
 *       v' = A W^T + bias + colv * (R_f16 - mean_r) * rstd_r,  colv = gamma, bias = b + beta, R_f16 = v (the fp16 stream);
 *       v' -> C_f16 (the new stream), C (its bf16 copy) and stats_out (the slices of this call's columns).
 * mean_r / rstd_r come from stats_in (row length K in mode 1, N in mode 2; np = that / 128 <= 8).  bf16 output only,
 * N % 128 == 0, K % 64 == 0, K >= 128; no dropout (eval mode).  vt_ln_apply materialises LN(v) where a caller needs the
 * normalised tensor itself (the encoder's output); vt_ln_stream_init turns a plain fp32 tensor into a stream whose
 * statistics say "already normalised" (the embedding output entering layer 0). */
int vt_linear_ln_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* colv,
                      const float* stats_in, int np, int64_t stat_rows, float ln_eps, int ln_mode, const void* R_f16,
                      int64_t ldrs, void* C, int64_t ldc, void* C_f16, int64_t ldcs, float* stats_out, int M, int N, int K,
                      int act, vt_stream_t stream);
/* y = (v - mean) * rstd * gamma + beta per row of the fp16 stream v [M,H]; y_bf16 and / or y_f32 (either may be NULL). */
int vt_ln_apply(const void* v, int64_t ldv, const float* stats, int np, int64_t stat_rows, const float* gamma,
                const float* beta, float ln_eps, void* y_bf16, int64_t ldy16, float* y_f32, int64_t ldy32, int64_t M, int H,
                vt_stream_t stream);
/* x fp32 [M,H] -> the stream (x_f16), its bf16 copy and identity statistics (mean 0, rstd 1): x enters the
 * deferred-LayerNorm stack as it is. */
int vt_ln_stream_init(const float* x, int64_t ldx, void* x_f16, int64_t lds, void* x_bf16, int64_t ldy, float* stats, int np,
                      int64_t stat_rows, int64_t M, int H, float ln_eps, vt_stream_t stream);

typedef struct vt_layer_weights_ln {
  const void* w_qkv;  const float* g_qkv; const float* h_qkv; /* [3H,H] bf16 (q|k|v weights) * gamma_in; its row sums; W beta_in + b */
  const void* w_ao;   const float* cb_ao;                     /* attention.output.dense [H,H]; its bias + beta_in */
  const float* gamma_in;                                      /* gamma of the LayerNorm that normalises this layer's INPUT stream */
  const void* w_in;   const float* g_in;  const float* h_in;  /* [I,H] bf16 intermediate.dense * ln1 gamma; row sums; W ln1_beta + b */
  const void* w_out;  const float* cb_out;                    /* output.dense [H,I]; its bias + ln1 beta */
  const float* ln1_g;                                         /* attention.output.LayerNorm gamma */
} vt_layer_weights_ln;
/* Stream A (s16_a bf16 copy, sf_a fp16, stats_a) holds the layer-0 input on entry (vt_ln_stream_init) and the LAST layer's
 * pre-LayerNorm sum on return (vt_ln_apply with that layer's output.LayerNorm gives the encoder output); stream B and
 * qkv [M,3H], ctx [M,H], mid [M,I] (bf16) are scratch.  Five launches per layer. */
int vt_encoder_forward_ln_bf16(const vt_layer_weights_ln* layers, int num_layers, void* s16_a, void* sf_a, float* stats_a,
                               void* s16_b, void* sf_b, float* stats_b, void* qkv, void* ctx, void* mid, const float* mask,
                               int mask_additive, const float* head_scale, int B, int S, int H, int nh, int I, float ln_eps,
                               int64_t stat_rows, vt_stream_t stream);
/* The same loop over COMPACTED rows (the layout of vt_encoder_forward_seq_bf16: `rows` token rows without the masked
 * positions, sequence b = rows seq_start[b] .. seq_start[b] + seq_len[b], every key of a sequence attended, no mask): the
 * rollout's eval forward (agent_models.py:256-277 calling oscar/modeling_bert.py:140-169) on the rows that exist.  The
 * streams / scratch hold `rows` rows, stat_rows >= rows. */
int vt_encoder_forward_ln_seq_bf16(const vt_layer_weights_ln* layers, int num_layers, void* s16_a, void* sf_a, float* stats_a,
                                   void* s16_b, void* sf_b, float* stats_b, void* qkv, void* ctx, void* mid,
                                   const float* head_scale, int B, int S, int H, int nh, int I, float ln_eps, int64_t stat_rows,
                                   int64_t rows, const int32_t* seq_start, const int32_t* seq_len, vt_stream_t stream);

/* ---- backward of the encoder stack (the encoder part of loss.backward(), pretrain.py:191) ------ */
typedef struct vt_layer_weights_t { /* transposed bf16 copies consumed by the dgrad GEMMs */
  const void* wt_qkv; /* [H,3H] = w_qkv^T */
  const void* wt_ao;  /* [H,H]  */
  const void* wt_in;  /* [H,I]  = w_in^T  */
  const void* wt_out; /* [I,H]  = w_out^T */
} vt_layer_weights_t;

typedef struct vt_layer_grads { /* fp32 gradient destinations, same shapes as the parameters */
  float* d_w_qkv; float* d_b_qkv; float* d_w_ao; float* d_b_ao; float* d_ln1_g; float* d_ln1_b;
  float* d_w_in;  float* d_b_in;  float* d_w_out; float* d_b_out; float* d_ln2_g; float* d_ln2_b;
} vt_layer_grads;

typedef struct vt_bwd_workspace {
  void* g_pre;   /* [M,H]  bf16 */
  void* g_pre2;  /* [M,H]  bf16 */
  void* g_mid;   /* [M,I]  bf16 */
  void* g_ctx;   /* [M,H]  bf16 */
  void* g_qkv;   /* [M,3H] bf16 */
  float* delta;      /* [B,nh,S] */
  float* ln_partial; /* [1024*2*H] */
  float* dq32;       /* [M,H] fp32, required when S > 256 (else may be NULL) */
  void* g_pre_d;     /* [M,H] bf16, required when p_hidden > 0 */
  void* g_pre2_d;    /* [M,H] bf16, required when p_hidden > 0 */
} vt_bwd_workspace;

/* g: [M,H] bf16, IN dL/d(last layer output), OUT dL/d(x) (layer-0 input).  acts must come from a
 * forward run with per-layer buffers, lse and mid_pre set.  accumulate != 0 adds into the gradient
 * destinations instead of overwriting them. */
int vt_encoder_backward_bf16(const vt_layer_weights* layers, const vt_layer_weights_t* layers_t,
                             const vt_layer_acts* acts, const vt_layer_grads* grads, int num_layers,
                             const void* x, const float* mask, int mask_additive, void* g,
                             const vt_bwd_workspace* ws, int B, int S, int H, int nh, int I, float ln_eps,
                             int accumulate, float p_hidden, float p_attn, uint64_t drop_seed, int layer0,
                             vt_stream_t stream);

/* The same loop with the weight gradients of layer l on side_stream while the main stream goes on with layer l-1
 * (the grouped wgrad launch occupies 216 of 256 CUs; LayerNorm backward and the other non-persistent kernels of the next
 * layer fill the rest).  ws_b: a second workspace whose g_pre, g_pre_d, g_pre2, g_pre2_d, g_mid and g_qkv are buffers of
 * their own (the other members may alias ws's); layer l works in set (layer0 + l) & 1.  Ordering is by events created
 * once per device inside the library; on return the main stream has been made to wait for every wgrad of the call.
 * ws_b == NULL or side_stream == NULL / == stream: identical to vt_encoder_backward_bf16. */
int vt_encoder_backward_overlap_bf16(const vt_layer_weights* layers, const vt_layer_weights_t* layers_t,
                                     const vt_layer_acts* acts, const vt_layer_grads* grads, int num_layers,
                                     const void* x, const float* mask, int mask_additive, void* g,
                                     const vt_bwd_workspace* ws, const vt_bwd_workspace* ws_b, int B, int S, int H, int nh,
                                     int I, float ln_eps, int accumulate, float p_hidden, float p_attn,
                                     uint64_t drop_seed, int layer0, vt_stream_t stream, vt_stream_t side_stream);

/* ---- compacted rows: the training step without its padding rows ------------------------------------------------
 * The reference computes every position of the padded [B, S] batch (text tails and missing regions carry attention
 * mask 0, data_loader_pretrain.py:666-690).  Nothing in a training step reads those rows: as keys they get probability
 * exactly 0 (encoder.py:238-241: -10000 underflows in the softmax), their labels are -1, their gradient is exactly 0.
 * The *_seq_* entry points run the same kernels on the `rows` real rows only: sequence b occupies rows seq_start[b] ..
 * seq_start[b] + seq_len[b] of every activation, all of its keys are attended (no mask argument).  CONTRACT:
 * 1 <= seq_len[b] <= S (position 0, [CLS], is always kept) -- device data the host side cannot check without a
 * synchronisation: the kernels clamp a length outside that range into it (meaningless rows, never an out-of-bounds access);
 * lse / delta keep their [B, nh, S] layout.  Losses and gradients equal the padded run's; outputs AT padding positions
 * do not exist, so inference keeps the padded entry points. */
int vt_attention_fwd_seq_bf16(const void* qkv, int64_t ld_qkv, const float* head_scale, void* ctx, int64_t ld_ctx,
                              float* lse, int B, int S, int nh, int head_size, float drop_p, uint64_t drop_seed,
                              uint32_t drop_site, const int32_t* seq_start, const int32_t* seq_len, uint32_t* keep_bits,
                              vt_stream_t stream);
int vt_attention_bwd_seq_bf16(const void* qkv, int64_t ld_qkv, const void* dctx, int64_t ld_d, const void* ctx,
                              int64_t ld_ctx, const float* lse, float* delta_ws, void* dqkv, int64_t ld_dqkv,
                              float* dq32_ws, int B, int S, int nh, int head_size, float drop_p, uint64_t drop_seed,
                              uint32_t drop_site, const int32_t* seq_start, const int32_t* seq_len, int64_t rows,
                              const uint32_t* keep_bits, vt_stream_t stream);
int vt_encoder_forward_seq_bf16(const vt_layer_weights* layers, const vt_layer_acts* acts, int num_layers, const void* x,
                                const float* head_scale, int B, int S, int H, int nh, int I, float ln_eps, float p_hidden,
                                float p_attn, uint64_t drop_seed, int64_t rows, const int32_t* seq_start,
                                const int32_t* seq_len, vt_stream_t stream);
/* (ws_b / side_stream as in vt_encoder_backward_overlap_bf16; both may be NULL) */
int vt_encoder_backward_seq_bf16(const vt_layer_weights* layers, const vt_layer_weights_t* layers_t,
                                 const vt_layer_acts* acts, const vt_layer_grads* grads, int num_layers, const void* x,
                                 void* g, const vt_bwd_workspace* ws, const vt_bwd_workspace* ws_b, int B, int S, int H,
                                 int nh, int I, float ln_eps, int accumulate, float p_hidden, float p_attn,
                                 uint64_t drop_seed, int layer0, int64_t rows, const int32_t* seq_start,
                                 const int32_t* seq_len, vt_stream_t stream, vt_stream_t side_stream);
/* (p_hidden, p_attn, drop_seed: the values the forward used; layer0 = index of layers[0] in the full
 * stack when the backward is run over a sub-range of layers.) */

#ifdef __cplusplus
}
#endif
#endif /* VISITRON_HIP_H */
