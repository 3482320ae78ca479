// extern "C" entry points declared in include/visitron_hip.h.  Thin: argument checks live in the
// *_dispatch functions next to each kernel; this file adds the layer loop of the encoder stack.
#include "common.hpp"
#include "../../include/visitron_hip.h"

int vt_gemm_dispatch(const void* A, long lda, const void* W, long ldw, const float* bias, const void* R, long ldr,
                     void* C, long ldc, int M, int N, int K, int act, int out_f32, int grp_rows, int grp_stride,
                     hipStream_t stream, void* C2 = nullptr, long ldc2 = 0, const DropCfg* drop = nullptr,
                     const VtLnResidual* rln = nullptr);
int vt_attention_bwd_dispatch(const void* qkv, long ld_qkv, const void* dctx, long ld_d, const void* ctx, long ld_ctx,
                              const float* mask, int mask_additive, const float* lse, float* delta_ws, void* dqkv,
                              long ld_dqkv, float* dq32_ws, int B, int S, int nh, int head_size, hipStream_t stream,
                              const DropCfg* drop = nullptr, const int* seq_start = nullptr, const int* seq_len = nullptr,
                              long rows_total = 0, const uint32_t* keep_bits = nullptr);
int vt_layernorm_bwd_dispatch(const void* x, long ldx, const void* dy, long ldy, const float* gamma, void* dx, long lddx,
                              float* dgamma, float* dbeta, float* partial_ws, int M, int H, float eps, int accumulate,
                              hipStream_t stream, void* dx2 = nullptr, long lddx2 = 0, const DropCfg* drop = nullptr,
                              int x_f16 = 0, const PrefetchArgs* pf = nullptr);
int vt_apply_dropout_dispatch(void* x, long ld, long rows, int cols, const DropCfg& d, hipStream_t stream);
int vt_dropout_mask_dispatch(uint8_t* out, long n, const DropCfg& d, hipStream_t stream, int attn);
int vt_dgelu_mul_dispatch(const void* g, const void* h, void* out, long n, hipStream_t stream);
int vt_embed_layernorm_bwd_dispatch(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                                    const float* pos, const float* type, const float* gamma, const void* g, long ldg,
                                    float* de, float* dgamma, float* dbeta, float* partial_ws, int B, int T, int S, int H,
                                    int n_word, int n_pos, int n_type, float eps, int accumulate, hipStream_t stream,
                                    const DropCfg* drop = nullptr);
int vt_ce_double_softmax_dispatch(const float* z, long ldz, const int64_t* y, float* loss_row, int64_t* amax, void* dz, long lddz,
                                  long rows, int V, int Vpad, float scale, hipStream_t stream);
int vt_ce_softmax_dispatch(const float* z, long ldz, const int64_t* y, float* loss_row, int64_t* amax, void* dz, long lddz,
                           long rows, int V, int Vpad, float scale, hipStream_t stream);
int vt_transpose_dispatch(const void* in, long ldi, void* out, long ldo, int R, int C, hipStream_t stream);
int vt_adamw_dispatch(float* p, const void* g, int g_is_bf16, float* m, float* v, void* p_bf16, long n, float lr,
                      float step_size, float b1, float b2, float eps, float wd, float grad_scale, hipStream_t stream);
int vt_cast_scale_dispatch(const float* x, void* y, long n, float scale, hipStream_t stream);
struct LstmPersistArgs {   // lstm_persistent.hip
  const float* xproj; long ldx_b, ldx_t; const int* xrow_start; float* h; float* c; const bf16_t* w_hh; const int* lengths;
  float* seq_out; long lds_b, lds_t; bf16_t* xchg; unsigned* sync; int B, hs, T, reverse;
};
int vt_lstm_persistent_dispatch(LstmPersistArgs a, void* ws, long ws_bytes, hipStream_t stream);
long vt_lstm_persistent_ws_bytes(int B, int hs);
int vt_scale_heads_dispatch(const void* x, long ldx, void* out, long ldo, long rows, int nh, const float* scale, hipStream_t stream);
int vt_mask_tokens_dispatch(const int64_t* ids, const uint8_t* special, const int64_t* token_classes, const float* u_mask,
                            const float* u_replace, const float* u_random, const int64_t* random_words, int64_t* out_ids,
                            int64_t* labels, int64_t* attention_mask, long n, int64_t pad_id, int64_t mask_id,
                            float mlm_probability, hipStream_t stream);
int vt_assemble_regions_dispatch(const float* img_feats, const int64_t* region_counts, const int64_t* region_view_ids,
                                 const int64_t* current_view, const float* loc_table, const int64_t* text_labels,
                                 const int64_t* text_mask, const int64_t* text_token_classes, float* feats_out, float* loc_out,
                                 int64_t* labels_out, int64_t* mask_out, int64_t* token_labels_out, int B, int T, int R,
                                 int R_in, int D, hipStream_t stream);
int vt_center_mask_dispatch(const void* mask, int kind, long ldm, float* out, int B, int S, hipStream_t stream);
int vt_embed_table_grad_dispatch(const int* sorted_ids, const long* perm, const float* de, long ld_de, float* grad, long ld_grad,
                                 long n, int H, long n_rows_table, long skip_id, float* scratch, int* flag, hipStream_t stream);
struct BatchRowsArgs {
  const long* lab; const long* tl; const float* mask; const int* err; long M; int S; int B; long* counts; int* tile_counts;
  long* idx_w; long* idx_t; long* index; long* inverse; int* start; int* length; long n_w, n_t, n_keep;
};
int vt_batch_rows_dispatch(const BatchRowsArgs& a, int lists, hipStream_t stream);
int vt_action_head_dispatch(const float* z, long ldz, const long* y, int B, int A, float grad_scale, void* dz, long lddz, int Ap,
                            float* out, hipStream_t stream);
int vt_gemm_splitk_dispatch(const void* A, long lda, const void* W, long ldw, void* C, long ldc, float* ws, int M, int N, int K,
                            int ksplit, hipStream_t stream);
int vt_attention_probs_dispatch(const void* qkv, long ld_qkv, const float* mask, int mask_additive, const float* head_scale,
                                const float* lse, float* probs, int B, int S, int nh, int head_size, hipStream_t stream);
int vt_attention_fwd_dispatch(const void* qkv, long ld_qkv, const float* mask, int mask_additive, const float* head_scale, void* ctx,
                              long ld_ctx, float* lse, int B, int S, int nh, int head_size, hipStream_t stream,
                              const DropCfg* drop = nullptr, const int* seq_start = nullptr, const int* seq_len = nullptr,
                              uint32_t* keep_bits = nullptr, const PrefetchArgs* pf = nullptr);
int vt_layernorm_dispatch(const void* x, long ldx, void* y, long ldy, const float* gamma, const float* beta,
                          float* mean, float* rstd, int M, int H, float eps, int grp_rows, int grp_stride,
                          hipStream_t stream, int x_f16 = 0, void* y_f16 = nullptr, long ldyh = 0,
                          const PrefetchArgs* pf = nullptr);
int vt_embed_layernorm_dispatch(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                                const float* pos, const float* type, const float* gamma, const float* beta, void* y,
                                long ldy, int B, int T, int S, int H, int n_word, int n_pos, int n_type, float eps,
                                int* err_flag, hipStream_t stream, const DropCfg* drop = nullptr);
int vt_pack_concat_dispatch(const float* s0, int d0, const float* s1, int d1, void* out, int kpad, long rows,
                            hipStream_t stream);

int vt_transpose_batch_dispatch(const void* const* in, const long* ldi, void* const* out, const long* ldo, const int* R,
                                const int* C, int n, hipStream_t stream);
void vt_gemm_set_variant(int v);
void vt_gemm_set_trace(void* p);
void vt_wgrad_set_tile(int tn);
void vt_wgrad_v8_enable(int on);
int vt_wgrad_v8_timeouts(unsigned* out);
unsigned* vt_wgrad_timeouts_devptr();                         // gemm_wgrad_v8.hip
int vt_gemm_sk_counter_ptrs(unsigned** ptrs, int max);          // gemm_v7.hip
int vt_prefetch_dispatch(const void* const* ptrs, const long* bytes, int n, hipStream_t stream);   // rowops.hip
int vt_gemm_set_workspace_impl(void* base, long bytes);        // gemm_v7.hip
int vt_gemm_shared_tile_timeouts_impl(unsigned* out);
long vt_gemm_workspace_region_bytes_impl();
int vt_gemm_f32_dispatch(const float* A, long lda, long sA_b, long sA_h, const float* W, long ldw, long sW_b, long sW_h,
                         int w_is_kn, const float* bias, const float* R, long ldr, float* C, long ldc, long sC_b, long sC_h,
                         int M, int N, int K, int act, float alpha, int batch, int heads, int grp_rows, int grp_stride,
                         hipStream_t stream);
int vt_softmax_rows_f32_dispatch(float* x, long ld, long rows, int cols, float scale, const float* mask, int mask_mode,
                                 const float* head_scale, int nh, int S, hipStream_t stream);
int vt_layernorm_f32_dispatch(const void* x, long ldx, int x_is_f32, void* y, long ldy, int y_is_f32, const float* gamma,
                              const float* beta, long M, int H, float eps, int grp_rows, int grp_stride, hipStream_t stream);
int vt_embed_layernorm_f32_dispatch(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                                    const float* pos, const float* type, const float* gamma, const float* beta, float* y,
                                    long ldy, int B, int T, int S, int H, int n_word, int n_pos, int n_type, float eps,
                                    int* err_flag, hipStream_t stream);

void vt_gemm_tune_set(int M, int N, int K, int kind, int variant);
void vt_gemm_set_reserved_cus(int k);
int vt_gemm_ln_dispatch(const void* A, long lda, const void* W, long ldw, const float* bias, const float* colv,
                        const float* stats_in, int np, long stat_rows, float eps, int ln_mode, const void* Rs, long ldrs,
                        void* C, long ldc, void* Cs, long ldcs, float* stats_out, int M, int N, int K, int act,
                        hipStream_t stream);
int vt_ln_apply_dispatch(const void* v, long ldv, const float* stats, int np, long stat_rows, const float* gamma,
                         const float* beta, float eps, void* y16, long ldy16, float* y32, long ldy32, long M, int H,
                         hipStream_t stream);
int vt_ln_stream_init_dispatch(const float* x, long ldx, void* s16, long lds, void* y16, long ldy, float* stats, int np,
                               long stat_rows, long M, int H, float eps, hipStream_t stream);
void vt_attn_bwd_set_waves(int w);

#include "wgrad_common.hpp"
int vt_wgrad_dispatch(WgradArgs& a, hipStream_t stream);
#include "rollout_args.hpp"
int vt_lstm_step_dispatch(const LstmStepArgs& a, hipStream_t stream);
int vt_lstm_step_bwd_dispatch(const LstmBwdArgs& a, hipStream_t stream);
int vt_softdot_bwd_dispatch(const SoftDotBwdArgs& a, hipStream_t stream);
int vt_softdot_bwd_split_dispatch(const SoftDotBwdArgs& a, float* ws, hipStream_t stream);
long vt_softdot_bwd_split_ws_floats(int B, int L, int D);
int vt_softdot_dispatch(const SoftDotArgs& a, hipStream_t stream);
int vt_skinny_linear_dispatch(const SkinnyArgs& a, hipStream_t stream);

extern "C" {

const char* vt_error_string(int code) {
  switch (code) {
    case VT_OK: return "ok";
    case VT_ERR_BAD_SHAPE: return "bad shape";
    case VT_ERR_BAD_ALIGN: return "bad alignment or leading dimension";
    case VT_ERR_NULL: return "null pointer";
    case VT_ERR_UNSUPPORTED: return "unsupported configuration";
    case VT_ERR_HIP: return "HIP launch error";
    default: return "unknown error";
  }
}

int vt_abi_version(void) { return 12; }

// attention-probability dropout: 16-bit fields (default since ABI 12: p in steps of 1/65536, two keys per hash word -- the
// reference's nn.Dropout(0.1) runs as 0.100006) or 8-bit fields (rounds 4-5's form: steps of 1/256, four keys per hash word,
// 0.1 runs as 0.1016; the attention forward is ~5 % faster, the B = 256 step 0.15 %; common.hpp).  Process-wide; read when a
// call builds its DropCfg, so forward and backward of one step must run under the same setting (set it before building the
// engine; VT_ATTN_DROPOUT_BITS=8 in the environment selects the old form from the start).
static int g_attn_drop_bits = 0;
static int attn_drop_bits() {
  if (g_attn_drop_bits == 0) {
    const char* e = getenv("VT_ATTN_DROPOUT_BITS");
    g_attn_drop_bits = (e && atoi(e) == 8) ? 8 : 16;
  }
  return g_attn_drop_bits;
}
int vt_set_attn_dropout_bits(int bits) {
  if (bits != 8 && bits != 16) return VT_ERR_UNSUPPORTED;
  g_attn_drop_bits = bits;
  return VT_OK;
}
int vt_get_attn_dropout_bits(void) { return attn_drop_bits(); }
float vt_attn_dropout_effective(float p) { return vt_attn_drop_ok(p, attn_drop_bits()) ? vt_attn_drop_p(p, attn_drop_bits()) : -1.0f; }

int vt_batch_row_counts(const int64_t* labels, const int64_t* token_labels, const float* mask, const int32_t* err_flag, int B,
                        int S, int64_t* counts, int32_t* tile_counts, vt_stream_t stream) {
  BatchRowsArgs a = {(const long*)labels, (const long*)token_labels, mask, (const int*)err_flag, (long)B * S, S, B, (long*)counts,
                     (int*)tile_counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  return vt_batch_rows_dispatch(a, 0, (hipStream_t)stream);
}

int vt_batch_row_lists(const int64_t* labels, const int64_t* token_labels, const float* mask, int B, int S, int64_t n_w,
                       int64_t n_t, int64_t n_keep, const int32_t* tile_counts, int64_t* idx_w, int64_t* idx_t, int64_t* index,
                       int64_t* inverse, int32_t* start, int32_t* length, vt_stream_t stream) {
  BatchRowsArgs a = {(const long*)labels, (const long*)token_labels, mask, nullptr, (long)B * S, S, B, nullptr,
                     (int*)tile_counts, (long*)idx_w, (long*)idx_t, (long*)index, (long*)inverse, (int*)start, (int*)length, n_w,
                     n_t, n_keep};
  return vt_batch_rows_dispatch(a, 1, (hipStream_t)stream);
}

int vt_action_head_f32(const float* logits, int64_t ld, const int64_t* next_action, int B, int A, float grad_scale, void* dlogits,
                       int64_t ldd, int Ap, float* loss_acc, vt_stream_t stream) {
  return vt_action_head_dispatch(logits, ld, (const long*)next_action, B, A, grad_scale, dlogits, ldd, Ap, loss_acc,
                                 (hipStream_t)stream);
}

int vt_linear_splitk_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, float* ws, int M, int N,
                          int K, int ksplit, vt_stream_t stream) {
  return vt_gemm_splitk_dispatch(x, ldx, w, ldw, y, ldy, ws, M, N, K, ksplit, (hipStream_t)stream);
}

int vt_embed_table_grad(const int32_t* sorted_ids, const int64_t* perm, const float* de, int64_t ld_de, float* grad,
                        int64_t ld_grad, int64_t n, int H, int64_t n_rows_table, int64_t skip_id, float* scratch, int32_t* flag,
                        vt_stream_t stream) {
  return vt_embed_table_grad_dispatch((const int*)sorted_ids, (const long*)perm, de, ld_de, grad, ld_grad, n, H, n_rows_table,
                                      skip_id, scratch, (int*)flag, (hipStream_t)stream);
}

void vt_debug_set_gemm_variant(int variant) { vt_gemm_set_variant(variant); }
void vt_debug_set_gemm_trace(void* buf) { vt_gemm_set_trace(buf); }
void vt_debug_set_wgrad_kernel(int mode) {
  // 0: automatic; 128 / 256: the 128(k)-tile kernel with that n-tile width; 8: the persistent kernel wherever it is eligible
  // (also below its row threshold); -8: never the persistent kernel
  vt_wgrad_set_tile(mode == 128 || mode == 256 || mode == 8 ? mode : 0);
  vt_wgrad_v8_enable(mode == -8 ? 0 : 1);
}
void vt_gemm_tune(int M, int N, int K, int kind, int variant) { vt_gemm_tune_set(M, N, K, kind, variant); }
void vt_debug_set_attn_bwd_waves(int waves) { vt_attn_bwd_set_waves(waves); }
void vt_gemm_reserve_cus(int k) { vt_gemm_set_reserved_cus(k); }
int vt_gemm_set_workspace(void* base, int64_t bytes) { return vt_gemm_set_workspace_impl(base, (long)bytes); }
int64_t vt_gemm_workspace_region_bytes(void) { return (int64_t)vt_gemm_workspace_region_bytes_impl(); }
int vt_gemm_shared_tile_timeouts(unsigned* host_count) {
  if (!host_count) return VT_ERR_NULL;
  return vt_gemm_shared_tile_timeouts_impl(host_count);
}

int vt_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* R,
                   int64_t ldr, void* C, int64_t ldc, int M, int N, int K, int act, int out_f32, int grp_rows,
                   int grp_stride, vt_stream_t stream) {
  return vt_gemm_dispatch(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, act, out_f32, grp_rows, grp_stride,
                          (hipStream_t)stream);
}

int vt_linear_bf16_ex(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* R,
                      int64_t ldr, void* C, int64_t ldc, void* C2, int64_t ldc2, int M, int N, int K, int act,
                      int out_f32, int grp_rows, int grp_stride, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                      vt_stream_t stream) {
  const DropCfg d = vt_make_drop(drop_p, drop_seed, drop_site);
  return vt_gemm_dispatch(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, act, out_f32, grp_rows, grp_stride,
                          (hipStream_t)stream, C2, ldc2, &d);
}

int vt_linear_lnres_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* Rv,
                         int64_t ldr, const float* mean, const float* rstd, const float* gamma, const float* beta, void* C,
                         int64_t ldc, int M, int N, int K, int out_f16, float drop_p, uint64_t drop_seed,
                         uint32_t drop_site, vt_stream_t stream) {
  if (!Rv) return VT_ERR_NULL;
  const DropCfg d = vt_make_drop(drop_p, drop_seed, drop_site);
  const VtLnResidual rln = {mean, rstd, gamma, beta};
  return vt_gemm_dispatch(A, lda, W, ldw, bias, Rv, ldr, C, ldc, M, N, K, VT_ACT_NONE, (out_f16 ? 2 : 0) | 4, 0, 0,
                          (hipStream_t)stream, nullptr, 0, &d, &rln);
}

int vt_apply_dropout_bf16(void* x, int64_t ld, int64_t rows, int cols, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                          vt_stream_t stream) {
  return vt_apply_dropout_dispatch(x, ld, rows, cols, vt_make_drop(drop_p, drop_seed, drop_site), (hipStream_t)stream);
}

int vt_debug_dropout_mask(uint8_t* out, int64_t n, float drop_p, uint64_t drop_seed, uint32_t drop_site, int head_index,
                          vt_stream_t stream) {
  if (head_index >= 0) {   // attention sites: one stream per (b, h), a hash word per four keys, 8-bit thresholds (common.hpp)
    if (!vt_attn_drop_ok(drop_p, attn_drop_bits())) return VT_ERR_UNSUPPORTED;
    DropCfg d = vt_make_drop_attn(drop_p, drop_seed, drop_site, attn_drop_bits());
    d.seed = vt_hash32(d.seed, (uint32_t)head_index);
    return vt_dropout_mask_dispatch(out, n, d, (hipStream_t)stream, 1);
  }
  return vt_dropout_mask_dispatch(out, n, vt_make_drop(drop_p, drop_seed, drop_site), (hipStream_t)stream, 0);
}

int vt_attention_bwd_bf16(const void* qkv, int64_t ld_qkv, const void* dctx, int64_t ld_d, const void* ctx,
                          int64_t ld_ctx, const float* mask, int mask_additive, const float* lse, float* delta_ws,
                          void* dqkv, int64_t ld_dqkv, float* dq32_ws, int B, int S, int nh, int head_size,
                          float drop_p, uint64_t drop_seed, uint32_t drop_site, const uint32_t* keep_bits,
                          vt_stream_t stream) {
  if (!vt_attn_drop_ok(drop_p, attn_drop_bits())) return VT_ERR_UNSUPPORTED;
  const DropCfg d = vt_make_drop_attn(drop_p, drop_seed, drop_site, attn_drop_bits());
  return vt_attention_bwd_dispatch(qkv, ld_qkv, dctx, ld_d, ctx, ld_ctx, mask, mask_additive, lse, delta_ws, dqkv,
                                   ld_dqkv, dq32_ws, B, S, nh, head_size, (hipStream_t)stream, &d, nullptr, nullptr, 0,
                                   keep_bits);
}

int vt_attention_bwd_seq_bf16(const void* qkv, int64_t ld_qkv, const void* dctx, int64_t ld_d, const void* ctx,
                              int64_t ld_ctx, const float* lse, float* delta_ws, void* dqkv, int64_t ld_dqkv,
                              float* dq32_ws, int B, int S, int nh, int head_size, float drop_p, uint64_t drop_seed,
                              uint32_t drop_site, const int32_t* seq_start, const int32_t* seq_len, int64_t rows,
                              const uint32_t* keep_bits, vt_stream_t stream) {
  if (!seq_start || !seq_len) return VT_ERR_NULL;
  if (!vt_attn_drop_ok(drop_p, attn_drop_bits())) return VT_ERR_UNSUPPORTED;
  const DropCfg d = vt_make_drop_attn(drop_p, drop_seed, drop_site, attn_drop_bits());
  return vt_attention_bwd_dispatch(qkv, ld_qkv, dctx, ld_d, ctx, ld_ctx, nullptr, 0, lse, delta_ws, dqkv, ld_dqkv, dq32_ws,
                                   B, S, nh, head_size, (hipStream_t)stream, &d, seq_start, seq_len, rows, keep_bits);
}

int vt_layernorm_bwd_bf16(const void* x, int64_t ldx, const void* dy, int64_t ldy, const float* gamma, void* dx,
                          int64_t lddx, float* dgamma, float* dbeta, float* partial_ws, int M, int H, float eps,
                          int accumulate, void* dx_dropped, int64_t lddxd, float drop_p, uint64_t drop_seed,
                          uint32_t drop_site, vt_stream_t stream) {
  const DropCfg d = vt_make_drop(drop_p, drop_seed, drop_site);
  return vt_layernorm_bwd_dispatch(x, ldx, dy, ldy, gamma, dx, lddx, dgamma, dbeta, partial_ws, M, H, eps, accumulate,
                                   (hipStream_t)stream, dx_dropped, lddxd, &d);
}

int vt_layernorm_bwd_h_bf16(const void* x_f16, int64_t ldx, const void* dy, int64_t ldy, const float* gamma, void* dx,
                            int64_t lddx, float* dgamma, float* dbeta, float* partial_ws, int M, int H, float eps,
                            int accumulate, void* dx_dropped, int64_t lddxd, float drop_p, uint64_t drop_seed,
                            uint32_t drop_site, vt_stream_t stream) {
  const DropCfg d = vt_make_drop(drop_p, drop_seed, drop_site);
  return vt_layernorm_bwd_dispatch(x_f16, ldx, dy, ldy, gamma, dx, lddx, dgamma, dbeta, partial_ws, M, H, eps, accumulate,
                                   (hipStream_t)stream, dx_dropped, lddxd, &d, 1);
}

int vt_embed_layernorm_bwd(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                           const float* pos, const float* type, const float* gamma, const void* g, int64_t ldg, float* de,
                           float* dgamma, float* dbeta, float* partial_ws, int B, int T, int S, int H, int n_word, int n_pos,
                           int n_type, float eps, int accumulate, float drop_p, uint64_t drop_seed, vt_stream_t stream) {
  const DropCfg d = vt_make_drop(drop_p, drop_seed, VT_SITE_EMB);
  return vt_embed_layernorm_bwd_dispatch(ids, type_ids, pos_ids, word, pos, type, gamma, g, ldg, de, dgamma, dbeta, partial_ws,
                                         B, T, S, H, n_word, n_pos, n_type, eps, accumulate, (hipStream_t)stream, &d);
}

int vt_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float step_size, float b1,
                  float b2, float eps, float wd, float grad_scale, vt_stream_t stream) {
  return vt_adamw_dispatch(p, g, 0, m, v, p_bf16, n, lr, step_size, b1, b2, eps, wd, grad_scale, (hipStream_t)stream);
}

int vt_adamw_flat_g16(float* p, const void* g_bf16, float* m, float* v, void* p_bf16, int64_t n, float lr, float step_size,
                      float b1, float b2, float eps, float wd, float grad_scale, vt_stream_t stream) {
  return vt_adamw_dispatch(p, g_bf16, 1, m, v, p_bf16, n, lr, step_size, b1, b2, eps, wd, grad_scale, (hipStream_t)stream);
}

int vt_mask_tokens(const int64_t* input_ids, const uint8_t* special_mask, const int64_t* token_classes, const float* u_mask,
                   const float* u_replace, const float* u_random, const int64_t* random_words, int64_t* out_ids,
                   int64_t* labels, int64_t* attention_mask, int64_t n, int64_t pad_id, int64_t mask_id,
                   float mlm_probability, vt_stream_t stream) {
  return vt_mask_tokens_dispatch(input_ids, special_mask, token_classes, u_mask, u_replace, u_random, random_words, out_ids,
                                 labels, attention_mask, n, pad_id, mask_id, mlm_probability, (hipStream_t)stream);
}

int vt_assemble_regions(const float* img_feats, const int64_t* region_counts, const int64_t* region_view_ids,
                        const int64_t* current_view, const float* loc_table, const int64_t* text_labels,
                        const int64_t* text_mask, const int64_t* text_token_classes, float* feats_out, float* loc_out,
                        int64_t* labels_out, int64_t* mask_out, int64_t* token_labels_out, int B, int T, int R, int R_in,
                        int D, vt_stream_t stream) {
  return vt_assemble_regions_dispatch(img_feats, region_counts, region_view_ids, current_view, loc_table, text_labels,
                                      text_mask, text_token_classes, feats_out, loc_out, labels_out, mask_out,
                                      token_labels_out, B, T, R, R_in, D, (hipStream_t)stream);
}

int vt_scale_heads_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int nh, const float* head_scale,
                        vt_stream_t stream) {
  return vt_scale_heads_dispatch(x, ldx, out, ldo, rows, nh, head_scale, (hipStream_t)stream);
}

int vt_cast_f32_to_bf16(const float* src, void* dst_bf16, int64_t n, float scale, vt_stream_t stream) {
  return vt_cast_scale_dispatch(src, dst_bf16, n, scale, (hipStream_t)stream);
}

int vt_ce_softmax_rows(const float* z, int64_t ldz, const int64_t* y, float* loss_row, int64_t* amax, void* dz, int64_t lddz,
                       int64_t rows, int V, int Vpad, float scale, vt_stream_t stream) {
  return vt_ce_softmax_dispatch(z, ldz, y, loss_row, amax, dz, lddz, rows, V, Vpad, scale, (hipStream_t)stream);
}

int vt_ce_double_softmax_rows(const float* z, int64_t ldz, const int64_t* y, float* loss_row, int64_t* amax, void* dz,
                              int64_t lddz, int64_t rows, int V, int Vpad, float scale, vt_stream_t stream) {
  return vt_ce_double_softmax_dispatch(z, ldz, y, loss_row, amax, dz, lddz, rows, V, Vpad, scale, (hipStream_t)stream);
}

int vt_lstm_step_train_f32(const float* xproj, int64_t ldx, const float* h_prev, float* h_out, float* c, const void* w_hh,
                           const int32_t* lengths, float* seq_out, int64_t ld_seq, int B, int hs, int t, float* sv_gates,
                           float* sv_c, void* sv_h, vt_stream_t stream) {
  LstmStepArgs a;
  a.xproj = xproj; a.ldx = ldx; a.h_prev = h_prev; a.h_out = h_out; a.c = c; a.w_hh = (const bf16_t*)w_hh;
  a.lengths = lengths; a.seq_out = seq_out; a.ld_seq = ld_seq; a.B = B; a.hs = hs; a.t = t;
  a.xrow_start = nullptr; a.ldx_row = 0;
  a.sv_gates = sv_gates; a.ld_svg = 4L * hs; a.sv_c = sv_c; a.ld_svc = hs; a.sv_h = (bf16_t*)sv_h; a.ld_svh = hs;
  return vt_lstm_step_dispatch(a, (hipStream_t)stream);
}

int vt_lstm_step_f32(const float* xproj, int64_t ldx, const float* h_prev, float* h_out, float* c, const void* w_hh,
                     const int32_t* lengths, float* seq_out, int64_t ld_seq, int B, int hs, int t, vt_stream_t stream) {
  return vt_lstm_step_train_f32(xproj, ldx, h_prev, h_out, c, w_hh, lengths, seq_out, ld_seq, B, hs, t, nullptr, nullptr,
                                nullptr, stream);
}

// sv_*: optional training saves laid out like the padded sequence, [B, S_sv, .] (position t of row b at (b * S_sv + t))
static int lstm_sequence_impl(const float* xproj, int64_t ldx_b, int64_t ldx_t, float* h2_0, float* h2_1, float* c,
                              const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b, int64_t lds_t,
                              int B, int hs, int T, int reverse, vt_stream_t stream, const int32_t* xrow_start,
                              float* sv_gates = nullptr, float* sv_c = nullptr, void* sv_h = nullptr, int64_t S_sv = 0) {
  if (!xproj || !h2_0 || !h2_1 || !c || !w_hh) return VT_ERR_NULL;
  if (T <= 0) return VT_ERR_BAD_SHAPE;
  if (xrow_start && !lengths) return VT_ERR_NULL;   // compacted rows exist only below a sequence's length
  if (sv_gates && (!sv_c || !sv_h || S_sv < T)) return VT_ERR_BAD_SHAPE;
  float* hb[2] = {h2_0, h2_1};
  for (int i = 0; i < T; ++i) {
    const int t = reverse ? T - 1 - i : i;
    LstmStepArgs a;
    a.xrow_start = xrow_start; a.ldx_row = ldx_t;
    a.xproj = xrow_start ? xproj : xproj + (int64_t)t * ldx_t; a.ldx = ldx_b; a.h_prev = hb[i & 1]; a.h_out = hb[(i + 1) & 1]; a.c = c;
    a.w_hh = (const bf16_t*)w_hh; a.lengths = lengths; a.seq_out = seq_out ? seq_out + (int64_t)t * lds_t : nullptr;
    a.ld_seq = lds_b; a.B = B; a.hs = hs; a.t = t;
    a.sv_gates = sv_gates ? sv_gates + (int64_t)t * 4 * hs : nullptr; a.ld_svg = S_sv * 4 * hs;
    a.sv_c = sv_gates ? sv_c + (int64_t)t * hs : nullptr; a.ld_svc = S_sv * hs;
    a.sv_h = sv_gates ? (bf16_t*)sv_h + (int64_t)t * hs : nullptr; a.ld_svh = S_sv * hs;
    const int rc = vt_lstm_step_dispatch(a, (hipStream_t)stream);
    if (rc != VT_OK) return rc;
  }
  if (T & 1) {
    if (hipMemcpyAsync(h2_0, h2_1, (size_t)B * hs * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) !=
        hipSuccess)
      return VT_ERR_HIP;
  }
  return VT_OK;
}

int vt_lstm_sequence_f32(const float* xproj, int64_t ldx_b, int64_t ldx_t, float* h2_0, float* h2_1, float* c,
                         const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b, int64_t lds_t,
                         int B, int hs, int T, int reverse, vt_stream_t stream) {
  return lstm_sequence_impl(xproj, ldx_b, ldx_t, h2_0, h2_1, c, w_hh, lengths, seq_out, lds_b, lds_t, B, hs, T, reverse,
                            stream, nullptr);
}

int vt_lstm_sequence_train_f32(const float* xproj, int64_t ldx_b, int64_t ldx_t, float* h2_0, float* h2_1, float* c,
                               const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b, int64_t lds_t,
                               int B, int hs, int T, int reverse, float* sv_gates, float* sv_c, void* sv_h, int64_t S_sv,
                               vt_stream_t stream) {
  if (!sv_gates || !sv_c || !sv_h) return VT_ERR_NULL;
  return lstm_sequence_impl(xproj, ldx_b, ldx_t, h2_0, h2_1, c, w_hh, lengths, seq_out, lds_b, lds_t, B, hs, T, reverse,
                            stream, nullptr, sv_gates, sv_c, sv_h, S_sv);
}

int vt_lstm_step_bwd_f32(const void* dg_next, int64_t ld_dgn, const void* w_hh_t, const float* dh_final, const float* d_out,
                         int64_t ld_dout, float* dc, const float* sv_gates, int64_t ld_svg, const float* sv_c,
                         int64_t ld_svc, void* dg_out, int64_t ld_dg, float* dg_out_f32, int64_t ld_dgf,
                         const int32_t* lengths, int B, int hs, int t, int t_next, vt_stream_t stream) {
  LstmBwdArgs a;
  a.dg_next = (const bf16_t*)dg_next; a.ld_dgn = ld_dgn; a.w_hh_t = (const bf16_t*)w_hh_t; a.dh_final = dh_final;
  a.d_out = d_out; a.ld_dout = ld_dout; a.dc = dc; a.sv_gates = sv_gates; a.ld_svg = ld_svg; a.sv_c = sv_c;
  a.ld_svc = ld_svc; a.dg_out = (bf16_t*)dg_out; a.ld_dg = ld_dg; a.dg_out_f32 = dg_out_f32; a.ld_dgf = ld_dgf;
  a.lengths = lengths; a.B = B; a.hs = hs; a.t = t; a.t_next = t_next;
  return vt_lstm_step_bwd_dispatch(a, (hipStream_t)stream);
}

// Back-propagation through the T steps of vt_lstm_sequence_train_f32, in the reverse of the order they ran.  d_seq_out
// [B, ., hs] (strides ldd_b, ldd_t; may be null), dh_final / dc [B, hs] (dc: in = gradient of the final cell state, it is
// the running value afterwards), saves and dgates [B, S_sv, .]; dgates must arrive zeroed at positions >= T.
int vt_lstm_sequence_bwd_f32(const float* d_seq_out, int64_t ldd_b, int64_t ldd_t, const float* dh_final, float* dc,
                             const void* w_hh_t, const int32_t* lengths, const float* sv_gates, const float* sv_c,
                             void* dgates, int64_t S_sv, int B, int hs, int T, int reverse, vt_stream_t stream) {
  if (!dc || !w_hh_t || !sv_gates || !sv_c || !dgates) return VT_ERR_NULL;
  if (T <= 0 || S_sv < T) return VT_ERR_BAD_SHAPE;
  bf16_t* dg = (bf16_t*)dgates;
  for (int i = T - 1; i >= 0; --i) {
    const int t = reverse ? T - 1 - i : i;                     // the forward's i-th step ran position t
    const int t_next = (i == T - 1) ? -1 : (reverse ? t - 1 : t + 1);
    LstmBwdArgs a;
    a.dg_next = t_next < 0 ? nullptr : dg + (int64_t)t_next * 4 * hs; a.ld_dgn = S_sv * 4 * hs;
    a.w_hh_t = (const bf16_t*)w_hh_t; a.dh_final = dh_final;
    a.d_out = d_seq_out ? d_seq_out + (int64_t)t * ldd_t : nullptr; a.ld_dout = ldd_b; a.dc = dc;
    a.sv_gates = sv_gates + (int64_t)t * 4 * hs; a.ld_svg = S_sv * 4 * hs; a.sv_c = sv_c + (int64_t)t * hs; a.ld_svc = S_sv * hs;
    a.dg_out = dg + (int64_t)t * 4 * hs; a.ld_dg = S_sv * 4 * hs; a.dg_out_f32 = nullptr; a.ld_dgf = 0;
    a.lengths = lengths; a.B = B; a.hs = hs; a.t = t; a.t_next = t_next;
    const int rc = vt_lstm_step_bwd_dispatch(a, (hipStream_t)stream);
    if (rc != VT_OK) return rc;
  }
  return VT_OK;
}

int vt_lstm_sequence_rows_f32(const float* xproj, int64_t ldx_row, const int32_t* row_start, float* h2_0, float* h2_1,
                              float* c, const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b,
                              int64_t lds_t, int B, int hs, int T, int reverse, vt_stream_t stream) {
  if (!row_start) return VT_ERR_NULL;
  return lstm_sequence_impl(xproj, 0, ldx_row, h2_0, h2_1, c, w_hh, lengths, seq_out, lds_b, lds_t, B, hs, T, reverse,
                            stream, row_start);
}

int64_t vt_lstm_sequence_persistent_ws_bytes(int B, int hs) { return vt_lstm_persistent_ws_bytes(B, hs); }

int vt_lstm_sequence_persistent_f32(const float* xproj, int64_t ldx_b, int64_t ldx_t, const int32_t* row_start, float* h,
                                    float* c, const void* w_hh, const int32_t* lengths, float* seq_out, int64_t lds_b,
                                    int64_t lds_t, int B, int hs, int T, int reverse, void* ws, int64_t ws_bytes,
                                    vt_stream_t stream) {
  LstmPersistArgs a;
  a.xproj = xproj; a.ldx_b = ldx_b; a.ldx_t = ldx_t; a.xrow_start = row_start; a.h = h; a.c = c;
  a.w_hh = (const bf16_t*)w_hh; a.lengths = lengths; a.seq_out = seq_out; a.lds_b = lds_b; a.lds_t = lds_t;
  a.xchg = nullptr; a.sync = nullptr; a.B = B; a.hs = hs; a.T = T; a.reverse = reverse;
  return vt_lstm_persistent_dispatch(a, ws, ws_bytes, (hipStream_t)stream);
}

int vt_skinny_linear_f32(const float* x0, int64_t ld0, int K0, const float* x1, int64_t ld1, int K1, const void* w,
                         int64_t ldw, const float* bias, float* out, int64_t ldo, int M, int N, int Kpad, int act,
                         vt_stream_t stream) {
  SkinnyArgs a;
  a.x0 = x0; a.ld0 = ld0; a.K0 = K0; a.x1 = x1; a.ld1 = ld1; a.K1 = K1; a.w = (const bf16_t*)w; a.ldw = ldw;
  a.bias = bias; a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.Kpad = Kpad; a.act = act;
  return vt_skinny_linear_dispatch(a, (hipStream_t)stream);
}

int vt_softdot_attention_bwd_f32(const float* target, const float* context, int64_t ld_batch, int64_t ld_row,
                                 const uint8_t* mask, const float* d_weighted, const float* d_attn, float* d_target,
                                 float* d_context, int B, int L, int D, int output_prob, vt_stream_t stream) {
  SoftDotBwdArgs a;
  a.target = target; a.context = context; a.ld_batch = ld_batch; a.ld_row = ld_row; a.mask = mask;
  a.d_weighted = d_weighted; a.d_attn = d_attn; a.d_target = d_target; a.d_context = d_context;
  a.B = B; a.L = L; a.D = D; a.output_prob = output_prob;
  return vt_softdot_bwd_dispatch(a, (hipStream_t)stream);
}

int64_t vt_softdot_attention_bwd_split_ws_floats(int B, int L, int D) { return vt_softdot_bwd_split_ws_floats(B, L, D); }

int vt_softdot_attention_bwd_split_f32(const float* target, const float* context, int64_t ld_batch, int64_t ld_row,
                                       const uint8_t* mask, const float* d_weighted, const float* d_attn, float* d_context,
                                       float* ws, int B, int L, int D, int output_prob, vt_stream_t stream) {
  SoftDotBwdArgs a;
  a.target = target; a.context = context; a.ld_batch = ld_batch; a.ld_row = ld_row; a.mask = mask;
  a.d_weighted = d_weighted; a.d_attn = d_attn; a.d_target = nullptr; a.d_context = d_context;
  a.B = B; a.L = L; a.D = D; a.output_prob = output_prob;
  return vt_softdot_bwd_split_dispatch(a, ws, (hipStream_t)stream);
}

int vt_softdot_attention_f32(const float* target, const float* context, int64_t ld_batch, int64_t ld_row,
                             const uint8_t* mask, float* weighted, float* attn, int B, int L, int D, int output_prob,
                             vt_stream_t stream) {
  SoftDotArgs a;
  a.target = target; a.context = context; a.ld_batch = ld_batch; a.ld_row = ld_row; a.mask = mask;
  a.weighted = weighted; a.attn = attn; a.B = B; a.L = L; a.D = D; a.output_prob = output_prob;
  return vt_softdot_dispatch(a, (hipStream_t)stream);
}

int vt_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, vt_stream_t stream) {
  return vt_transpose_dispatch(in, ldi, out, ldo, R, C, (hipStream_t)stream);
}

int vt_transpose_batch_bf16(const void* const* in, const int64_t* ldi, void* const* out, const int64_t* ldo, const int* R,
                            const int* C, int n, vt_stream_t stream) {
  return vt_transpose_batch_dispatch(in, (const long*)ldi, out, (const long*)ldo, R, C, n, (hipStream_t)stream);
}

int vt_dgelu_mul_bf16(const void* g, const void* h, void* out, int64_t n, vt_stream_t stream) {
  return vt_dgelu_mul_dispatch(g, h, out, n, (hipStream_t)stream);
}

int vt_attention_fwd_bf16(const void* qkv, int64_t ld_qkv, const float* mask, int mask_additive, const float* head_scale, void* ctx,
                          int64_t ld_ctx, float* lse, int B, int S, int nh, int head_size, float drop_p, uint64_t drop_seed,
                          uint32_t drop_site, uint32_t* keep_bits, vt_stream_t stream) {
  if (!vt_attn_drop_ok(drop_p, attn_drop_bits())) return VT_ERR_UNSUPPORTED;
  const DropCfg d = vt_make_drop_attn(drop_p, drop_seed, drop_site, attn_drop_bits());
  return vt_attention_fwd_dispatch(qkv, ld_qkv, mask, mask_additive, head_scale, ctx, ld_ctx, lse, B, S, nh, head_size,
                                   (hipStream_t)stream, &d, nullptr, nullptr, keep_bits);
}

int vt_attention_fwd_seq_bf16(const void* qkv, int64_t ld_qkv, const float* head_scale, void* ctx, int64_t ld_ctx,
                              float* lse, int B, int S, int nh, int head_size, float drop_p, uint64_t drop_seed,
                              uint32_t drop_site, const int32_t* seq_start, const int32_t* seq_len, uint32_t* keep_bits,
                              vt_stream_t stream) {
  if (!seq_start || !seq_len) return VT_ERR_NULL;
  if (!vt_attn_drop_ok(drop_p, attn_drop_bits())) return VT_ERR_UNSUPPORTED;
  const DropCfg d = vt_make_drop_attn(drop_p, drop_seed, drop_site, attn_drop_bits());
  return vt_attention_fwd_dispatch(qkv, ld_qkv, nullptr, 0, head_scale, ctx, ld_ctx, lse, B, S, nh, head_size,
                                   (hipStream_t)stream, &d, seq_start, seq_len, keep_bits);
}

int vt_attention_probs_f32(const void* qkv, int64_t ld_qkv, const float* mask, int mask_additive, const float* head_scale,
                           const float* lse, float* probs, int B, int S, int nh, int head_size, vt_stream_t stream) {
  return vt_attention_probs_dispatch(qkv, ld_qkv, mask, mask_additive, head_scale, lse, probs, B, S, nh, head_size,
                                     (hipStream_t)stream);
}

int vt_layernorm_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                      float* mean, float* rstd, int M, int H, float eps, int grp_rows, int grp_stride,
                      vt_stream_t stream) {
  return vt_layernorm_dispatch(x, ldx, y, ldy, gamma, beta, mean, rstd, M, H, eps, grp_rows, grp_stride,
                               (hipStream_t)stream);
}

int vt_layernorm_h_bf16(const void* x_f16, int64_t ldx, void* y, int64_t ldy, void* y_f16, int64_t ldyh, const float* gamma,
                        const float* beta, float* mean, float* rstd, int M, int H, float eps, vt_stream_t stream) {
  return vt_layernorm_dispatch(x_f16, ldx, y, ldy, gamma, beta, mean, rstd, M, H, eps, 0, 0, (hipStream_t)stream, 1, y_f16, ldyh);
}

int vt_embed_layernorm(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                       const float* pos, const float* type, const float* gamma, const float* beta, void* y,
                       int64_t ldy, int B, int T, int S, int H, int n_word, int n_pos, int n_type, float eps,
                       int* err_flag, float drop_p, uint64_t drop_seed, vt_stream_t stream) {
  const DropCfg d = vt_make_drop(drop_p, drop_seed, VT_SITE_EMB);
  return vt_embed_layernorm_dispatch(ids, type_ids, pos_ids, word, pos, type, gamma, beta, y, ldy, B, T, S, H, n_word,
                                     n_pos, n_type, eps, err_flag, (hipStream_t)stream, &d);
}

int vt_pack_concat_bf16(const float* s0, int d0, const float* s1, int d1, void* out, int kpad, int64_t rows,
                        vt_stream_t stream) {
  return vt_pack_concat_dispatch(s0, d0, s1, d1, out, kpad, rows, (hipStream_t)stream);
}

int vt_center_mask(const void* mask, int kind, int64_t ldm, float* out, int B, int S, vt_stream_t stream) {
  return vt_center_mask_dispatch(mask, kind, ldm, out, B, S, (hipStream_t)stream);
}

// ---- fp32 parity path (fp32_path.hip) ----------------------------------------------------------------------------
int vt_linear_f32(const float* a, int64_t lda, const float* w, int64_t ldw, int w_is_kn, const float* bias,
                  const float* residual, int64_t ldr, float* out, int64_t ldc, int M, int N, int K, int act, float alpha,
                  int grp_rows, int grp_stride, vt_stream_t stream) {
  return vt_gemm_f32_dispatch(a, lda, 0, 0, w, ldw, 0, 0, w_is_kn, bias, residual, ldr, out, ldc, 0, 0, M, N, K, act, alpha,
                              1, 1, grp_rows, grp_stride, (hipStream_t)stream);
}

int vt_bmm_f32(const float* a, int64_t lda, int64_t a_stride_b, int64_t a_stride_h, const float* w, int64_t ldw,
               int64_t w_stride_b, int64_t w_stride_h, int w_is_kn, float* out, int64_t ldc, int64_t c_stride_b,
               int64_t c_stride_h, int M, int N, int K, float alpha, int batch, int heads, vt_stream_t stream) {
  return vt_gemm_f32_dispatch(a, lda, a_stride_b, a_stride_h, w, ldw, w_stride_b, w_stride_h, w_is_kn, nullptr, nullptr, 0,
                              out, ldc, c_stride_b, c_stride_h, M, N, K, VT_ACT_NONE, alpha, batch, heads, 0, 0,
                              (hipStream_t)stream);
}

int vt_softmax_rows_f32(float* x, int64_t ld, int64_t rows, int cols, float scale, const float* mask, int mask_mode,
                        const float* head_scale, int nh, int S, vt_stream_t stream) {
  return vt_softmax_rows_f32_dispatch(x, ld, rows, cols, scale, mask, mask_mode, head_scale, nh, S, (hipStream_t)stream);
}

int vt_layernorm_rows(const void* x, int64_t ldx, int x_is_f32, void* y, int64_t ldy, int y_is_f32, const float* gamma,
                      const float* beta, int64_t M, int H, float eps, int grp_rows, int grp_stride, vt_stream_t stream) {
  return vt_layernorm_f32_dispatch(x, ldx, x_is_f32, y, ldy, y_is_f32, gamma, beta, M, H, eps, grp_rows, grp_stride,
                                   (hipStream_t)stream);
}

int vt_embed_layernorm_f32(const int64_t* input_ids, const int64_t* token_type_ids, const int64_t* position_ids,
                           const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                           float* out, int64_t ld_out, int B, int T, int S, int H, int n_word, int n_pos, int n_type,
                           float eps, int* err_flag, vt_stream_t stream) {
  return vt_embed_layernorm_f32_dispatch(input_ids, token_type_ids, position_ids, word, pos, type, gamma, beta, out, ld_out,
                                         B, T, S, H, n_word, n_pos, n_type, eps, err_flag, (hipStream_t)stream);
}

}  // extern "C"
struct StepCounterArgs {
  unsigned* wg;
  unsigned* sk[8];
  int nsk;
  long long* out;
};
__global__ void step_counters(StepCounterArgs a) {
  if (threadIdx.x != 0) return;
  const unsigned w = a.wg ? atomicExch(a.wg, 0u) : 0u;
  unsigned s = 0;
  for (int i = 0; i < a.nsk; ++i) s += atomicExch(a.sk[i], 0u);
  a.out[0] = (long long)w;
  a.out[1] = (long long)s;
}
extern "C" {

// The two "ran out of a bounded wait" counters of the current device (vt_wgrad_turn_timeouts, vt_gemm_shared_tile_timeouts),
// read and cleared by a one-thread kernel on `stream` into out[0] / out[1] (device int64): a training step reads them with its
// one host synchronisation instead of three blocking 4-byte copies (25-30 us of idle GPU each at the start of every step).
int vt_step_counters(int64_t* out2, vt_stream_t stream) {
  if (!out2) return VT_ERR_NULL;
  StepCounterArgs a;
  a.wg = vt_wgrad_timeouts_devptr();
  a.nsk = vt_gemm_sk_counter_ptrs(a.sk, 8);
  for (int i = a.nsk; i < 8; ++i) a.sk[i] = nullptr;
  a.out = (long long*)out2;
  hipLaunchKernelGGL(step_counters, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_wgrad_turn_timeouts(unsigned* host_count) {
  if (!host_count) return VT_ERR_NULL;
  return vt_wgrad_v8_timeouts(host_count);
}

int vt_wgrad_bf16(const vt_wgrad_problem* problems, int nprob, int M, vt_stream_t stream) {
  if (!problems) return VT_ERR_NULL;
  if (nprob <= 0 || nprob > WG_MAX_PROBLEMS) return VT_ERR_BAD_SHAPE;
  WgradArgs a;
  a.nprob = nprob;
  a.M = M;
  for (int i = 0; i < nprob; ++i) {
    const vt_wgrad_problem& q = problems[i];
    WgradProblem& P = a.p[i];
    P.dY = (const bf16_t*)q.dY; P.ldy = q.ldy; P.X = (const bf16_t*)q.X; P.ldx = q.ldx;
    P.dW = q.dW; P.ldw = q.ldw; P.db = q.db; P.N = q.N; P.K = q.K; P.accumulate = q.accumulate;
    P.tiles_k = 0; P.tile_begin = 0;
  }
  for (int i = nprob; i < WG_MAX_PROBLEMS; ++i) a.p[i] = a.p[0];
  return vt_wgrad_dispatch(a, (hipStream_t)stream);
}

// ---- deferred-LayerNorm inference path ------------------------------------------------------------------------------
int vt_linear_ln_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* colv,
                      const float* stats_in, int np, int64_t stat_rows, float ln_eps, int ln_mode, const void* R_f16,
                      int64_t ldrs, void* C, int64_t ldc, void* C_f16, int64_t ldcs, float* stats_out, int M, int N, int K,
                      int act, vt_stream_t stream) {
  return vt_gemm_ln_dispatch(A, lda, W, ldw, bias, colv, stats_in, np, stat_rows, ln_eps, ln_mode, R_f16, ldrs, C, ldc, C_f16,
                             ldcs, stats_out, M, N, K, act, (hipStream_t)stream);
}

int vt_ln_apply(const void* v, int64_t ldv, const float* stats, int np, int64_t stat_rows, const float* gamma,
                const float* beta, float ln_eps, void* y_bf16, int64_t ldy16, float* y_f32, int64_t ldy32, int64_t M, int H,
                vt_stream_t stream) {
  return vt_ln_apply_dispatch(v, ldv, stats, np, stat_rows, gamma, beta, ln_eps, y_bf16, ldy16, y_f32, ldy32, M, H,
                              (hipStream_t)stream);
}

int vt_ln_stream_init(const float* x, int64_t ldx, void* x_f16, int64_t lds, void* x_bf16, int64_t ldy, float* stats, int np,
                      int64_t stat_rows, int64_t M, int H, float ln_eps, vt_stream_t stream) {
  return vt_ln_stream_init_dispatch(x, ldx, x_f16, lds, x_bf16, ldy, stats, np, stat_rows, M, H, ln_eps, (hipStream_t)stream);
}

// CaptionBertEncoder.forward (oscar/modeling_bert.py:140-169) in eval mode with the LayerNorms deferred: five launches per
// layer (no LayerNorm pass; the residual stream stays fp16):
//   qkv GEMM (LN of the incoming stream folded in) -> fused attention -> out-proj GEMM (+ LN(stream) as residual; new
//   stream + statistics) -> FFN-up GEMM (LN folded in, GELU) -> FFN-down GEMM (+ LN(stream); new stream + statistics)
// rows != 0: the streams hold `rows` compacted token rows, sequence b = rows seq_start[b] .. + seq_len[b], every key of a
// sequence attended (no mask) -- the layout of vt_encoder_forward_seq_bf16.
}  // extern "C"

static int prefetch4(const void* p0, long b0, const void* p1, long b1, const void* p2, long b2, const void* p3, long b3,
                     hipStream_t stream);
static PrefetchArgs prefetch_args(const void* p0, long b0, const void* p1, long b1, const void* p2 = nullptr, long b2 = 0,
                                  const void* p3 = nullptr, long b3 = 0) {
  PrefetchArgs a;
  a.n = 0;
  const void* p[4] = {p0, p1, p2, p3};
  const long b[4] = {b0, b1, b2, b3};
  for (int i = 0; i < 4; ++i)
    if (p[i] && b[i] > 0 && !((uintptr_t)p[i] & 15)) { a.p[a.n] = p[i]; a.bytes[a.n] = b[i]; ++a.n; }
  for (int i = a.n; i < 4; ++i) { a.p[i] = nullptr; a.bytes[i] = 0; }
  return a;
}
// The inference layer loop (buffers shared by all layers; the twelve layers' weights do not survive a forward's traffic in the
// Infinity Cache at small batch either): VT_PREFETCH_INFER = 0 off / 1 one launch per layer / 2 two launches per layer /
// 3 (default) riding in the attention kernel's spare workgroups: no launch, measured -6 ... -9 % at B <= 16
static int g_prefetch_infer = -2;
static int prefetch_infer_mode() {
  if (g_prefetch_infer == -2) {
    const char* e = getenv("VT_PREFETCH_INFER");
    g_prefetch_infer = e ? atoi(e) : 3;
  }
  return g_prefetch_infer;
}

static int encoder_forward_ln_impl(const vt_layer_weights_ln* layers, int num_layers, void* s16_a, void* sf_a, float* stats_a,
                                   void* s16_b, void* sf_b, float* stats_b, void* qkv, void* ctx, void* mid, const float* mask,
                                   int mask_additive, const float* head_scale, int B, int S, int H, int nh, int I, float ln_eps,
                                   int64_t stat_rows, hipStream_t stream, long rows, const int* seq_start, const int* seq_len) {
  if (!layers || !s16_a || !sf_a || !stats_a || !s16_b || !sf_b || !stats_b || !qkv || !ctx || !mid) return VT_ERR_NULL;
  if (num_layers <= 0 || B <= 0 || S <= 0 || nh <= 0 || H != nh * 64 || (H % 128) || (I % 128) || H > 1024) return VT_ERR_BAD_SHAPE;
  if (rows && (!seq_start || !seq_len || mask || rows < 0 || rows > (long)B * S)) return VT_ERR_BAD_SHAPE;
  const int M = rows ? (int)rows : B * S, np = H / 128;
  if (stat_rows < M) return VT_ERR_BAD_SHAPE;
  for (int l = 0; l < num_layers; ++l) {
    const vt_layer_weights_ln& w = layers[l];
    int rc;
    const int pfi = prefetch_infer_mode();
    const long b_qkv = 6L * H * H, b_ao = 2L * H * H, b_ffn = 2L * H * I;
    if (pfi == 1) {
      rc = prefetch4(w.w_ao, b_ao, w.w_in, b_ffn, w.w_out, b_ffn, l + 1 < num_layers ? layers[l + 1].w_qkv : nullptr, b_qkv, stream);
      if (rc) return rc;
    } else if (pfi == 2) {
      rc = prefetch4(w.w_ao, b_ao, w.w_in, b_ffn, nullptr, 0, nullptr, 0, stream);
      if (rc) return rc;
    }
    rc = vt_gemm_ln_dispatch(s16_a, H, w.w_qkv, H, w.h_qkv, w.g_qkv, stats_a, np, stat_rows, ln_eps, 1, nullptr, 0, qkv, 3L * H,
                             nullptr, 0, nullptr, M, 3 * H, H, VT_ACT_NONE, stream);
    if (rc) return rc;
    const DropCfg nodrop = vt_make_drop(0.f, 0, 0);
    const PrefetchArgs pf_att = prefetch_args(w.w_ao, b_ao, w.w_in, b_ffn, w.w_out, b_ffn,
                                              l + 1 < num_layers ? layers[l + 1].w_qkv : nullptr, b_qkv);
    rc = vt_attention_fwd_dispatch(qkv, 3L * H, mask, mask_additive, head_scale ? head_scale + (long)l * nh : nullptr, ctx, H,
                                   nullptr, B, S, nh, 64, stream, &nodrop, rows ? seq_start : nullptr, rows ? seq_len : nullptr,
                                   nullptr, pfi == 3 ? &pf_att : nullptr);
    if (rc) return rc;
    if (pfi == 2) {
      rc = prefetch4(w.w_out, b_ffn, l + 1 < num_layers ? layers[l + 1].w_qkv : nullptr, b_qkv, nullptr, 0, nullptr, 0, stream);
      if (rc) return rc;
    }
    rc = vt_gemm_ln_dispatch(ctx, H, w.w_ao, H, w.cb_ao, w.gamma_in, stats_a, np, stat_rows, ln_eps, 2, sf_a, H, s16_b, H, sf_b,
                             H, stats_b, M, H, H, VT_ACT_NONE, stream);
    if (rc) return rc;
    rc = vt_gemm_ln_dispatch(s16_b, H, w.w_in, H, w.h_in, w.g_in, stats_b, np, stat_rows, ln_eps, 1, nullptr, 0, mid, I, nullptr, 0,
                             nullptr, M, I, H, VT_ACT_GELU, stream);
    if (rc) return rc;
    rc = vt_gemm_ln_dispatch(mid, I, w.w_out, I, w.cb_out, w.ln1_g, stats_b, np, stat_rows, ln_eps, 2, sf_b, H, s16_a, H, sf_a, H,
                             stats_a, M, H, I, VT_ACT_NONE, stream);
    if (rc) return rc;
  }
  return VT_OK;
}

extern "C" {

int vt_encoder_forward_ln_bf16(const vt_layer_weights_ln* layers, int num_layers, void* s16_a, void* sf_a, float* stats_a,
                               void* s16_b, void* sf_b, float* stats_b, void* qkv, void* ctx, void* mid, const float* mask,
                               int mask_additive, const float* head_scale, int B, int S, int H, int nh, int I, float ln_eps,
                               int64_t stat_rows, vt_stream_t stream) {
  return encoder_forward_ln_impl(layers, num_layers, s16_a, sf_a, stats_a, s16_b, sf_b, stats_b, qkv, ctx, mid, mask, mask_additive,
                                 head_scale, B, S, H, nh, I, ln_eps, stat_rows, (hipStream_t)stream, 0, nullptr, nullptr);
}

int vt_encoder_forward_ln_seq_bf16(const vt_layer_weights_ln* layers, int num_layers, void* s16_a, void* sf_a, float* stats_a,
                                   void* s16_b, void* sf_b, float* stats_b, void* qkv, void* ctx, void* mid,
                                   const float* head_scale, int B, int S, int H, int nh, int I, float ln_eps, int64_t stat_rows,
                                   int64_t rows, const int32_t* seq_start, const int32_t* seq_len, vt_stream_t stream) {
  if (rows <= 0) return VT_ERR_BAD_SHAPE;
  return encoder_forward_ln_impl(layers, num_layers, s16_a, sf_a, stats_a, s16_b, sf_b, stats_b, qkv, ctx, mid, nullptr, 0,
                                 head_scale, B, S, H, nh, I, ln_eps, stat_rows, (hipStream_t)stream, (long)rows, seq_start, seq_len);
}

// CaptionBertEncoder.forward (oscar/modeling_bert.py:140-169): the Python loop over layers, each
// layer = CaptionBertLayer.forward (:112-124) as 7 launches on one stream:
//   qkv GEMM -> fused attention -> out-proj GEMM(+bias+residual) -> LayerNorm
//   -> FFN-up GEMM(+bias+GELU) -> FFN-down GEMM(+bias+residual) -> LayerNorm
}  // extern "C"

// rows != 0: the activations hold `rows` compacted token rows (no padding rows), sequence b = rows seq_start[b] ..
// seq_start[b] + seq_len[b]; every key of a sequence is attended (no mask).  rows == 0: B * S rows, sequence b at b * S.
// Weight prefetch in the training layer loops (rowops.hip, prefetch_ranges): 0 off, 1 one launch per layer, 2 one launch in
// front of every kernel that precedes a GEMM (the weights of that GEMM), -1 automatic = mode VT_PREFETCH_MODE (default 1)
// below VT_PREFETCH_MAX_ROWS rows (default 16 384: at B = 256 the K loops run three rounds per CU at the chip's power-limited
// rate and hide the first touch; measured level there).
static int g_prefetch_mode = -2;
static long g_prefetch_max_rows = 16384;
static int prefetch_mode(long rows) {
  if (g_prefetch_mode == -2) {
    const char* e = getenv("VT_PREFETCH_WEIGHTS");
    g_prefetch_mode = e ? atoi(e) : 4;
    const char* r = getenv("VT_PREFETCH_MAX_ROWS");
    if (r) g_prefetch_max_rows = atol(r);
  }
  return rows <= g_prefetch_max_rows ? g_prefetch_mode : 0;
}
static int prefetch4(const void* p0, long b0, const void* p1, long b1, const void* p2, long b2, const void* p3, long b3,
                     hipStream_t stream) {
  const void* p[4] = {p0, p1, p2, p3};
  const long b[4] = {b0, b1, b2, b3};
  return vt_prefetch_dispatch(p, b, 4, stream);
}
// Mode 3: the same reads on a SIDE stream of the library's own, started behind an event on the caller's stream and never
// waited for (nothing depends on them): issued in front of the attention kernels, whose waves leave registers and memory
// bandwidth free -- the persistent GEMMs hold every register of a CU they run on, so beside them a prefetch would only queue.
struct PrefetchSide {
  hipStream_t stream;
  hipEvent_t ev[64];
  unsigned next;
  bool ok;
};
static PrefetchSide* prefetch_side() {
  static PrefetchSide side[16];
  static bool made[16] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!made[dev]) {
    made[dev] = true;
    PrefetchSide& s = side[dev];
    s.ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; s.ok && i < 64; ++i) s.ok = hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) == hipSuccess;
    s.next = 0;
  }
  return side[dev].ok ? &side[dev] : nullptr;
}
static int prefetch4_side(const void* p0, long b0, const void* p1, long b1, const void* p2, long b2, const void* p3, long b3,
                          hipStream_t stream) {
  PrefetchSide* s = prefetch_side();
  if (!s) return prefetch4(p0, b0, p1, b1, p2, b2, p3, b3, stream);
  hipEvent_t e = s->ev[s->next++ & 63];
  if (hipEventRecord(e, stream) != hipSuccess || hipStreamWaitEvent(s->stream, e, 0) != hipSuccess) return VT_ERR_HIP;
  return prefetch4(p0, b0, p1, b1, p2, b2, p3, b3, s->stream);
}

static int encoder_forward_impl(const vt_layer_weights* layers, const vt_layer_acts* acts, int num_layers, const void* x,
                                const float* mask, int mask_additive, const float* head_scale, int B, int S, int H, int nh,
                                int I, float ln_eps, float p_hidden, float p_attn, uint64_t drop_seed, hipStream_t stream,
                                long rows, const int* seq_start, const int* seq_len) {
  if (!layers || !acts || !x) return VT_ERR_NULL;
  if (num_layers <= 0 || B <= 0 || S <= 0 || nh <= 0 || H != nh * 64 || (H % 64) || (I % 64)) return VT_ERR_BAD_SHAPE;
  if (rows && (!seq_start || !seq_len || mask || rows < 0 || rows > (long)B * S)) return VT_ERR_BAD_SHAPE;
  const int M = rows ? (int)rows : B * S;
  const void* cur = x;
  const void* cur_h = nullptr;   // fp16 copy of `cur` (the previous layer's output), when that layer kept one -- or, with
                                 // cur_ln set, the previous layer's fp16 pre-LayerNorm sum whose LayerNorm `cur` is
  VtLnResidual cur_ln = {nullptr, nullptr, nullptr, nullptr};
  for (int l = 0; l < num_layers; ++l) {
    const vt_layer_weights& w = layers[l];
    const vt_layer_acts& a = acts[l];
    if (!a.qkv || !a.ctx || !a.attn_pre || !a.attn_out || !a.mid || !a.out_pre || !a.out) return VT_ERR_NULL;
    int rc;
    const int pf = prefetch_mode(M);
    const long b_qkv = 6L * H * H, b_ao = 2L * H * H, b_ffn = 2L * H * I;   // bytes of the layer's four bf16 weight matrices
    if (pf == 1) {   // this layer's later weights and the next layer's first one, in one launch
      rc = prefetch4(w.w_ao, b_ao, w.w_in, b_ffn, w.w_out, b_ffn, l + 1 < num_layers ? layers[l + 1].w_qkv : nullptr, b_qkv, stream);
      if (rc) return rc;
    } else if (pf == 2) {
      rc = prefetch4(w.w_ao, b_ao, nullptr, 0, nullptr, 0, nullptr, 0, stream);
      if (rc) return rc;
    }
    rc = vt_gemm_dispatch(cur, H, w.w_qkv, H, w.b_qkv, nullptr, 0, a.qkv, 3L * H, M, 3 * H, H, VT_ACT_NONE, 0, 0, 0, stream);
    if (rc) return rc;
    if (!vt_attn_drop_ok(p_attn, attn_drop_bits())) return VT_ERR_UNSUPPORTED;
    const DropCfg d_att = vt_make_drop_attn(p_attn, drop_seed, VT_SITE_ATTN(l), attn_drop_bits());
    const DropCfg d_so = vt_make_drop(p_hidden, drop_seed, VT_SITE_SELFOUT(l));
    const DropCfg d_out = vt_make_drop(p_hidden, drop_seed, VT_SITE_OUT(l));
    if (pf == 3) {   // beside the attention kernel: the three weights the rest of this layer reads, and the next layer's first
      rc = prefetch4_side(w.w_ao, b_ao, w.w_in, b_ffn, w.w_out, b_ffn, l + 1 < num_layers ? layers[l + 1].w_qkv : nullptr, b_qkv, stream);
      if (rc) return rc;
    }
    rc = vt_attention_fwd_dispatch(a.qkv, 3L * H, mask, mask_additive, head_scale ? head_scale + (long)l * nh : nullptr, a.ctx, H,
                                   a.lse, B, S, nh, 64, stream, &d_att, rows ? seq_start : nullptr, rows ? seq_len : nullptr,
                                   a.keep_bits);
    if (rc) return rc;
    // The residual stream.  Plain form: every tensor bf16.  With the layer's fp16 copies present (ln1_h / ln2_h non-null,
    // vt_layer_acts): the pre-LayerNorm sums attn_pre / out_pre are written and read as fp16, each LayerNorm writes its
    // output twice -- bf16 for the next GEMM's A operand (and the backward), fp16 for the next sub-layer's residual add --
    // so the stream itself is rounded to 11 significant bits instead of 8 (north_star's 5e-2 on the hidden states of the
    // path training runs: 5.8e-2 with the bf16 stream on the stress weights).
    // ln_residual_mode 1: the fp16 copies are never written -- a residual add reads the previous sub-layer's fp16 SUM and
    // reconstructs its LayerNorm from the row statistics that LayerNorm's kernel wrote (GemmArgs::r_mean); the LayerNorm
    // kernel then has one output instead of two.
    const bool rln = a.ln_residual_mode == 1;
    if (rln && (!a.ln1_mean || !a.ln1_rstd || !a.ln2_mean || !a.ln2_rstd)) return VT_ERR_NULL;
    if (a.ln_residual_mode != 0 && !rln) return VT_ERR_UNSUPPORTED;
    const bool h16 = rln || (a.ln1_h && a.ln2_h);
    const void* res = cur_h ? cur_h : cur;
    if (pf == 2) {
      rc = prefetch4(w.w_in, b_ffn, w.w_out, b_ffn, nullptr, 0, nullptr, 0, stream);
      if (rc) return rc;
    }
    rc = vt_gemm_dispatch(a.ctx, H, w.w_ao, H, w.b_ao, res, H, a.attn_pre, H, M, H, H, VT_ACT_NONE,
                          (h16 ? 2 : 0) | (cur_h ? 4 : 0), 0, 0, stream, nullptr, 0, &d_so, cur_ln.mean ? &cur_ln : nullptr);
    if (rc) return rc;
    // mode 4: the LayerNorm kernels carry the prefetch in spare workgroups -- LayerNorm 1 the two FFN weights, LayerNorm 2 the
    // next layer's attention weights (no launch of its own)
    const PrefetchArgs pf_ln1 = prefetch_args(w.w_in, b_ffn, w.w_out, b_ffn);
    const PrefetchArgs pf_ln2 = l + 1 < num_layers ? prefetch_args(layers[l + 1].w_qkv, b_qkv, layers[l + 1].w_ao, b_ao)
                                                   : prefetch_args(nullptr, 0, nullptr, 0);
    rc = vt_layernorm_dispatch(a.attn_pre, H, a.attn_out, H, w.ln1_g, w.ln1_b, a.ln1_mean, a.ln1_rstd, M, H, ln_eps, 0, 0, stream,
                               h16 ? 1 : 0, (h16 && !rln) ? a.ln1_h : nullptr, H, pf == 4 ? &pf_ln1 : nullptr);
    if (rc) return rc;
    rc = vt_gemm_dispatch(a.attn_out, H, w.w_in, H, w.b_in, nullptr, 0, a.mid, I, M, I, H, VT_ACT_GELU, 0, 0, 0, stream,
                          a.mid_pre, I);
    if (rc) return rc;
    const VtLnResidual ln1 = {a.ln1_mean, a.ln1_rstd, w.ln1_g, w.ln1_b};
    if (pf == 2 && l + 1 < num_layers) {
      rc = prefetch4(layers[l + 1].w_qkv, b_qkv, nullptr, 0, nullptr, 0, nullptr, 0, stream);
      if (rc) return rc;
    }
    rc = vt_gemm_dispatch(a.mid, I, w.w_out, I, w.b_out, rln ? a.attn_pre : (h16 ? a.ln1_h : a.attn_out), H, a.out_pre, H, M, H, I,
                          VT_ACT_NONE, h16 ? 6 : 0, 0, 0, stream, nullptr, 0, &d_out, rln ? &ln1 : nullptr);
    if (rc) return rc;
    rc = vt_layernorm_dispatch(a.out_pre, H, a.out, H, w.ln2_g, w.ln2_b, a.ln2_mean, a.ln2_rstd, M, H, ln_eps, 0, 0, stream,
                               h16 ? 1 : 0, (h16 && !rln) ? a.ln2_h : nullptr, H, pf == 4 ? &pf_ln2 : nullptr);
    if (rc) return rc;
    cur = a.out;
    cur_h = rln ? a.out_pre : (h16 ? a.ln2_h : nullptr);
    cur_ln = rln ? VtLnResidual{a.ln2_mean, a.ln2_rstd, w.ln2_g, w.ln2_b} : VtLnResidual{nullptr, nullptr, nullptr, nullptr};
  }
  return VT_OK;
}

extern "C" {

int vt_encoder_forward_bf16(const vt_layer_weights* layers, const vt_layer_acts* acts, int num_layers, const void* x,
                            const float* mask, int mask_additive, const float* head_scale, int B, int S, int H, int nh,
                            int I, float ln_eps, float p_hidden, float p_attn, uint64_t drop_seed, vt_stream_t stream) {
  return encoder_forward_impl(layers, acts, num_layers, x, mask, mask_additive, head_scale, B, S, H, nh, I, ln_eps, p_hidden,
                              p_attn, drop_seed, (hipStream_t)stream, 0, nullptr, nullptr);
}

int vt_encoder_forward_seq_bf16(const vt_layer_weights* layers, const vt_layer_acts* acts, int num_layers, const void* x,
                                const float* head_scale, int B, int S, int H, int nh, int I, float ln_eps, float p_hidden,
                                float p_attn, uint64_t drop_seed, int64_t rows, const int32_t* seq_start,
                                const int32_t* seq_len, vt_stream_t stream) {
  if (rows <= 0) return VT_ERR_BAD_SHAPE;
  return encoder_forward_impl(layers, acts, num_layers, x, nullptr, 0, head_scale, B, S, H, nh, I, ln_eps, p_hidden, p_attn,
                              drop_seed, (hipStream_t)stream, (long)rows, seq_start, seq_len);
}

// Weight prefetch of the layer loops (see prefetch_mode / prefetch_infer_mode above): training 0 off, 1 one launch per layer,
// 2 a launch per GEMM pair, 3 side stream, 4 (default) in the LayerNorm kernels' spare workgroups; inference 0 off, 1 / 2
// launches, 3 (default) in the attention kernel's spare workgroups.  -1 keeps a setting.  Values are never changed by it.
int vt_set_weight_prefetch(int training_mode, int inference_mode) {
  if (training_mode < -1 || training_mode > 4 || inference_mode < -1 || inference_mode > 3) return VT_ERR_UNSUPPORTED;
  (void)prefetch_mode(0);
  (void)prefetch_infer_mode();
  if (training_mode >= 0) g_prefetch_mode = training_mode;
  if (inference_mode >= 0) g_prefetch_infer = inference_mode;
  return VT_OK;
}
int vt_get_weight_prefetch(int inference) {
  (void)prefetch_mode(0);
  return inference ? prefetch_infer_mode() : g_prefetch_mode;
}

// Backward of CaptionBertEncoder (oscar/modeling_bert.py:140-169) = the reverse layer loop; per layer
// 4 dgrad GEMMs (residual adds and the dGELU fused in their epilogues), 2 LayerNorm backwards, the
// fused attention backward and ONE grouped weight-gradient launch for the layer's four matrices
// (bias gradients ride along in it).
}  // extern "C"

// Events that order the weight-gradient launches on the side stream against the dgrad chain (one pair per layer of
// a call; created once per device, never destroyed: a few dozen host-side handles).
#define VT_BWD_MAX_LAYERS 64
static hipEvent_t* bwd_events(int which) {
  static hipEvent_t ev[16][2][VT_BWD_MAX_LAYERS];
  static bool made[16] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!made[dev]) {
    for (int k = 0; k < 2; ++k)
      for (int i = 0; i < VT_BWD_MAX_LAYERS; ++i)
        if (hipEventCreateWithFlags(&ev[dev][k][i], hipEventDisableTiming) != hipSuccess) return nullptr;
    made[dev] = true;
  }
  return ev[dev][which];
}

// The reverse layer loop.  With a second workspace set (ws_b) and a side stream the four weight gradients of layer l
// run on the side stream while the main stream goes on with layer l-1: the grouped wgrad launch keeps 216 of the 256
// CUs busy (108 tiles x 2 row ranges), the next layer's LayerNorm backward and whatever else is not a persistent
// kernel fills the rest.  Layer l works in workspace set (layer0 + l) & 1, so the buffers a wgrad still reads are
// rewritten two layers later, behind a wait on that wgrad's completion event; before returning the main stream waits
// for every wgrad of the call.
static int encoder_backward_impl(const vt_layer_weights* layers, const vt_layer_weights_t* layers_t,
                                 const vt_layer_acts* acts, const vt_layer_grads* grads, int num_layers, const void* x,
                                 const float* mask, int mask_additive, void* g, const vt_bwd_workspace* ws_a,
                                 const vt_bwd_workspace* ws_b, int B, int S, int H, int nh, int I, float ln_eps,
                                 int accumulate, float p_hidden, float p_attn, uint64_t drop_seed, int layer0,
                                 hipStream_t stream, hipStream_t side, long rows = 0, const int* seq_start = nullptr,
                                 const int* seq_len = nullptr) {
  if (!layers || !layers_t || !acts || !grads || !x || !g || !ws_a) return VT_ERR_NULL;
  if (rows && (!seq_start || !seq_len || mask || rows < 0 || rows > (long)B * S)) return VT_ERR_BAD_SHAPE;
  const bool overlap = ws_b != nullptr && side != nullptr && side != stream;
  for (int k = 0; k < (overlap ? 2 : 1); ++k) {
    const vt_bwd_workspace* ws = k ? ws_b : ws_a;
    if (p_hidden > 0.f && (!ws->g_pre_d || !ws->g_pre2_d)) return VT_ERR_NULL;
    if (!ws->g_pre || !ws->g_pre2 || !ws->g_mid || !ws->g_ctx || !ws->g_qkv || !ws->delta || !ws->ln_partial) return VT_ERR_NULL;
  }
  if (num_layers <= 0 || B <= 0 || S <= 0 || nh <= 0 || H != nh * 64 || (I % 64)) return VT_ERR_BAD_SHAPE;
  if (overlap && num_layers > VT_BWD_MAX_LAYERS) return VT_ERR_BAD_SHAPE;
  hipEvent_t* ev_in = overlap ? bwd_events(0) : nullptr;    // E[l]: layer l's wgrad operands are complete (main)
  hipEvent_t* ev_done = overlap ? bwd_events(1) : nullptr;  // F[l]: layer l's wgrad has finished (side)
  if (overlap && (!ev_in || !ev_done)) return VT_ERR_HIP;
  const int M = rows ? (int)rows : B * S;
  for (int l = num_layers - 1; l >= 0; --l) {
    const vt_layer_weights& w = layers[l];
    const vt_layer_weights_t& wt = layers_t[l];
    const vt_layer_acts& a = acts[l];
    const vt_layer_grads& d = grads[l];
    if (!a.mid_pre || !a.lse) return VT_ERR_NULL;
    const void* x_in = l == 0 ? x : acts[l - 1].out;
    const vt_bwd_workspace* ws = (overlap && ((layer0 + l) & 1)) ? ws_b : ws_a;
    // this layer rewrites the set that the wgrad of layer l + 2 reads
    if (overlap && l + 2 < num_layers && hipStreamWaitEvent(stream, ev_done[l + 2], 0) != hipSuccess) return VT_ERR_HIP;
    int rc;
    // dropout sites of this layer (the forward used layer index layer0 + l)
    if (!vt_attn_drop_ok(p_attn, attn_drop_bits())) return VT_ERR_UNSUPPORTED;
    const DropCfg d_att = vt_make_drop_attn(p_attn, drop_seed, VT_SITE_ATTN(layer0 + l), attn_drop_bits());
    const DropCfg d_so = vt_make_drop(p_hidden, drop_seed, VT_SITE_SELFOUT(layer0 + l));
    const DropCfg d_out = vt_make_drop(p_hidden, drop_seed, VT_SITE_OUT(layer0 + l));
    // with hidden dropout the gradient of a dense output is the pre-LayerNorm gradient times the mask
    void* g_pre_dn = p_hidden > 0.f ? ws->g_pre_d : ws->g_pre;
    void* g_pre2_dn = p_hidden > 0.f ? ws->g_pre2_d : ws->g_pre2;
    const int pf = prefetch_mode(M);
    const long b_qkv = 6L * H * H, b_ao = 2L * H * H, b_ffn = 2L * H * I;   // the transposed copies have the same sizes
    if (pf == 1) {   // the four transposed weight copies this layer's dgrad GEMMs read
      rc = prefetch4(wt.wt_out, b_ffn, wt.wt_in, b_ffn, wt.wt_ao, b_ao, wt.wt_qkv, b_qkv, stream);
      if (rc) return rc;
    } else if (pf == 2) {
      rc = prefetch4(wt.wt_out, b_ffn, wt.wt_in, b_ffn, nullptr, 0, nullptr, 0, stream);
      if (rc) return rc;
    } else if (pf == 3 && l == num_layers - 1) {   // the call's first layer: beside its LayerNorm backward
      rc = prefetch4_side(wt.wt_out, b_ffn, wt.wt_in, b_ffn, wt.wt_ao, b_ao, nullptr, 0, stream);
      if (rc) return rc;
    }
    // LayerNorm 2 backward: dL/d(out_pre)
    const int h16 = ((a.ln1_h && a.ln2_h) || a.ln_residual_mode == 1) ? 1 : 0;   // the forward kept the pre-LayerNorm sums as fp16 (see encoder_forward_impl)
    const PrefetchArgs pf_ln2 = prefetch_args(wt.wt_out, b_ffn, wt.wt_in, b_ffn);   // mode 4: riding in the reduce kernels
    const PrefetchArgs pf_ln1 = prefetch_args(wt.wt_ao, b_ao, wt.wt_qkv, b_qkv);
    rc = vt_layernorm_bwd_dispatch(a.out_pre, H, g, H, w.ln2_g, ws->g_pre, H, d.d_ln2_g, d.d_ln2_b, ws->ln_partial, M, H,
                                   ln_eps, accumulate, stream, p_hidden > 0.f ? ws->g_pre_d : nullptr, H, &d_out, h16,
                                   pf == 4 ? &pf_ln2 : nullptr);
    if (rc) return rc;
    // through output.dense and the GELU: g_mid = (g_pre . W_out) * gelu'(pre-activation) (saved in mid_pre)
    rc = vt_gemm_dispatch(g_pre_dn, H, wt.wt_out, H, nullptr, a.mid_pre, I, ws->g_mid, I, M, I, H, VT_ACT_MUL, 0, 0, 0, stream);
    if (rc) return rc;
    // through intermediate.dense, plus the residual branch: dL/d(attn_out) -> g
    rc = vt_gemm_dispatch(ws->g_mid, I, wt.wt_in, I, nullptr, ws->g_pre, H, g, H, M, H, I, VT_ACT_NONE, 0, 0, 0, stream);
    if (rc) return rc;
    if (pf == 2) {
      rc = prefetch4(wt.wt_ao, b_ao, wt.wt_qkv, b_qkv, nullptr, 0, nullptr, 0, stream);
      if (rc) return rc;
    }
    // LayerNorm 1 backward: dL/d(attn_pre)
    rc = vt_layernorm_bwd_dispatch(a.attn_pre, H, g, H, w.ln1_g, ws->g_pre2, H, d.d_ln1_g, d.d_ln1_b, ws->ln_partial, M, H,
                                   ln_eps, accumulate, stream, p_hidden > 0.f ? ws->g_pre2_d : nullptr, H, &d_so, h16,
                                   pf == 4 ? &pf_ln1 : nullptr);
    if (rc) return rc;
    // through attention.output.dense: dL/d(ctx)
    rc = vt_gemm_dispatch(g_pre2_dn, H, wt.wt_ao, H, nullptr, nullptr, 0, ws->g_ctx, H, M, H, H, VT_ACT_NONE, 0, 0, 0, stream);
    if (rc) return rc;
    if (pf == 3) {   // beside the attention backward: this layer's last dgrad weight and the three of the layer below
      const vt_layer_weights_t* nx = l > 0 ? &layers_t[l - 1] : nullptr;
      rc = prefetch4_side(wt.wt_qkv, b_qkv, nx ? nx->wt_out : nullptr, b_ffn, nx ? nx->wt_in : nullptr, b_ffn,
                          nx ? nx->wt_ao : nullptr, b_ao, stream);
      if (rc) return rc;
    }
    rc = vt_attention_bwd_dispatch(a.qkv, 3L * H, ws->g_ctx, H, a.ctx, H, mask, mask_additive, a.lse, ws->delta, ws->g_qkv,
                                   3L * H, ws->dq32, B, S, nh, 64, stream, &d_att, rows ? seq_start : nullptr,
                                   rows ? seq_len : nullptr, rows, a.keep_bits);
    if (rc) return rc;
    // through the packed q|k|v projection, plus the residual branch: dL/d(layer input) -> g
    rc = vt_gemm_dispatch(ws->g_qkv, 3L * H, wt.wt_qkv, 3L * H, nullptr, ws->g_pre2, H, g, H, M, H, 3 * H, VT_ACT_NONE, 0, 0, 0, stream);
    if (rc) return rc;
    // the four weight (+bias) gradients of this layer in one grouped launch
    WgradArgs wa;
    wa.nprob = 4;
    wa.M = M;
    auto set = [&](int i, const void* dY, long ldy, const void* X, long ldx, float* dW, float* db, int N, int K) {
      WgradProblem& P = wa.p[i];
      P.dY = (const bf16_t*)dY; P.ldy = ldy; P.X = (const bf16_t*)X; P.ldx = ldx; P.dW = dW; P.ldw = K; P.db = db;
      P.N = N; P.K = K; P.accumulate = accumulate; P.tiles_k = 0; P.tile_begin = 0;
    };
    set(0, ws->g_mid, I, a.attn_out, H, d.d_w_in, d.d_b_in, I, H);
    set(1, g_pre_dn, H, a.mid, I, d.d_w_out, d.d_b_out, H, I);
    set(2, ws->g_qkv, 3L * H, x_in, H, d.d_w_qkv, d.d_b_qkv, 3 * H, H);
    set(3, g_pre2_dn, H, a.ctx, H, d.d_w_ao, d.d_b_ao, H, H);
    for (int i = 4; i < WG_MAX_PROBLEMS; ++i) wa.p[i] = wa.p[0];
    if (overlap) {
      if (hipEventRecord(ev_in[l], stream) != hipSuccess || hipStreamWaitEvent(side, ev_in[l], 0) != hipSuccess) return VT_ERR_HIP;
      rc = vt_wgrad_dispatch(wa, side);
      if (rc) return rc;
      if (hipEventRecord(ev_done[l], side) != hipSuccess) return VT_ERR_HIP;
    } else {
      rc = vt_wgrad_dispatch(wa, stream);
      if (rc) return rc;
    }
  }
  if (overlap)   // the caller's next work on the main stream (all-reduce, optimizer) sees every weight gradient
    for (int l = (num_layers < 2 ? num_layers : 2) - 1; l >= 0; --l)
      if (hipStreamWaitEvent(stream, ev_done[l], 0) != hipSuccess) return VT_ERR_HIP;
  return VT_OK;
}

extern "C" {

int vt_encoder_backward_bf16(const vt_layer_weights* layers, const vt_layer_weights_t* layers_t,
                             const vt_layer_acts* acts, const vt_layer_grads* grads, int num_layers, const void* x,
                             const float* mask, int mask_additive, void* g, const vt_bwd_workspace* ws, int B, int S,
                             int H, int nh, int I, float ln_eps, int accumulate, float p_hidden, float p_attn,
                             uint64_t drop_seed, int layer0, vt_stream_t stream) {
  return encoder_backward_impl(layers, layers_t, acts, grads, num_layers, x, mask, mask_additive, g, ws, nullptr, B, S, H,
                               nh, I, ln_eps, accumulate, p_hidden, p_attn, drop_seed, layer0, (hipStream_t)stream, nullptr);
}

int vt_encoder_backward_overlap_bf16(const vt_layer_weights* layers, const vt_layer_weights_t* layers_t,
                                     const vt_layer_acts* acts, const vt_layer_grads* grads, int num_layers,
                                     const void* x, const float* mask, int mask_additive, void* g,
                                     const vt_bwd_workspace* ws, const vt_bwd_workspace* ws_b, int B, int S, int H, int nh,
                                     int I, float ln_eps, int accumulate, float p_hidden, float p_attn,
                                     uint64_t drop_seed, int layer0, vt_stream_t stream, vt_stream_t side_stream) {
  return encoder_backward_impl(layers, layers_t, acts, grads, num_layers, x, mask, mask_additive, g, ws, ws_b, B, S, H, nh,
                               I, ln_eps, accumulate, p_hidden, p_attn, drop_seed, layer0, (hipStream_t)stream,
                               (hipStream_t)side_stream);
}

int vt_encoder_backward_seq_bf16(const vt_layer_weights* layers, const vt_layer_weights_t* layers_t,
                                 const vt_layer_acts* acts, const vt_layer_grads* grads, int num_layers, const void* x,
                                 void* g, const vt_bwd_workspace* ws, const vt_bwd_workspace* ws_b, int B, int S, int H,
                                 int nh, int I, float ln_eps, int accumulate, float p_hidden, float p_attn,
                                 uint64_t drop_seed, int layer0, int64_t rows, const int32_t* seq_start,
                                 const int32_t* seq_len, vt_stream_t stream, vt_stream_t side_stream) {
  if (rows <= 0) return VT_ERR_BAD_SHAPE;
  return encoder_backward_impl(layers, layers_t, acts, grads, num_layers, x, nullptr, 0, g, ws, ws_b, B, S, H, nh, I, ln_eps,
                               accumulate, p_hidden, p_attn, drop_seed, layer0, (hipStream_t)stream,
                               (hipStream_t)side_stream, (long)rows, seq_start, seq_len);
}

}  // extern "C"
