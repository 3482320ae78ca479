// extern "C" entry points declared in include/visitron_hip.h.  Thin: argument checks live in the
// *_dispatch functions next to each kernel; this file adds the layer loop of the encoder stack.
#include "common.hpp"
#include "../../include/visitron_hip.h"

int vt_gemm_dispatch(const void* A, long lda, const void* W, long ldw, const float* bias, const void* R, long ldr,
                     void* C, long ldc, int M, int N, int K, int act, int out_f32, int grp_rows, int grp_stride,
                     hipStream_t stream);
int vt_attention_fwd_dispatch(const void* qkv, long ld_qkv, const float* mask, int mask_additive, const float* head_scale, void* ctx,
                              long ld_ctx, float* lse, int B, int S, int nh, int head_size, hipStream_t stream);
int vt_layernorm_dispatch(const void* x, long ldx, void* y, long ldy, const float* gamma, const float* beta,
                          float* mean, float* rstd, int M, int H, float eps, int grp_rows, int grp_stride,
                          hipStream_t stream);
int vt_embed_layernorm_dispatch(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                                const float* pos, const float* type, const float* gamma, const float* beta, void* y,
                                long ldy, int B, int T, int S, int H, int n_word, int n_pos, int n_type, float eps,
                                int* err_flag, hipStream_t stream);
int vt_pack_concat_dispatch(const float* s0, int d0, const float* s1, int d1, void* out, int kpad, long rows,
                            hipStream_t stream);

void vt_gemm_set_variant(int v);

#define WG_MAX_PROBLEMS 8
struct WgradProblem {
  const bf16_t* dY; long ldy;
  const bf16_t* X; long ldx;
  float* dW; long ldw;
  float* db;
  int N, K;
  int tiles_k;
  int tile_begin;
  int accumulate;
};
struct WgradArgs {
  WgradProblem p[WG_MAX_PROBLEMS];
  int nprob;
  int M;
};
int vt_wgrad_dispatch(WgradArgs& a, hipStream_t stream);

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

int vt_abi_version(void) { return 1; }

void vt_debug_set_gemm_variant(int variant) { vt_gemm_set_variant(variant); }

int vt_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* R,
                   int64_t ldr, void* C, int64_t ldc, int M, int N, int K, int act, int out_f32, int grp_rows,
                   int grp_stride, vt_stream_t stream) {
  return vt_gemm_dispatch(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, act, out_f32, grp_rows, grp_stride,
                          (hipStream_t)stream);
}

int vt_attention_fwd_bf16(const void* qkv, int64_t ld_qkv, const float* mask, int mask_additive, const float* head_scale, void* ctx,
                          int64_t ld_ctx, float* lse, int B, int S, int nh, int head_size, vt_stream_t stream) {
  return vt_attention_fwd_dispatch(qkv, ld_qkv, mask, mask_additive, head_scale, ctx, ld_ctx, lse, B, S, nh, head_size,
                                   (hipStream_t)stream);
}

int vt_layernorm_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                      float* mean, float* rstd, int M, int H, float eps, int grp_rows, int grp_stride,
                      vt_stream_t stream) {
  return vt_layernorm_dispatch(x, ldx, y, ldy, gamma, beta, mean, rstd, M, H, eps, grp_rows, grp_stride,
                               (hipStream_t)stream);
}

int vt_embed_layernorm(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                       const float* pos, const float* type, const float* gamma, const float* beta, void* y,
                       int64_t ldy, int B, int T, int S, int H, int n_word, int n_pos, int n_type, float eps,
                       int* err_flag, vt_stream_t stream) {
  return vt_embed_layernorm_dispatch(ids, type_ids, pos_ids, word, pos, type, gamma, beta, y, ldy, B, T, S, H, n_word,
                                     n_pos, n_type, eps, err_flag, (hipStream_t)stream);
}

int vt_pack_concat_bf16(const float* s0, int d0, const float* s1, int d1, void* out, int kpad, int64_t rows,
                        vt_stream_t stream) {
  return vt_pack_concat_dispatch(s0, d0, s1, d1, out, kpad, rows, (hipStream_t)stream);
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

// CaptionBertEncoder.forward (oscar/modeling_bert.py:140-169): the Python loop over layers, each
// layer = CaptionBertLayer.forward (:112-124) as 7 launches on one stream:
//   qkv GEMM -> fused attention -> out-proj GEMM(+bias+residual) -> LayerNorm
//   -> FFN-up GEMM(+bias+GELU) -> FFN-down GEMM(+bias+residual) -> LayerNorm
int vt_encoder_forward_bf16(const vt_layer_weights* layers, const vt_layer_acts* acts, int num_layers, const void* x,
                            const float* mask, int mask_additive, const float* head_scale, int B, int S, int H, int nh,
                            int I, float ln_eps, vt_stream_t stream_) {
  if (!layers || !acts || !x) return VT_ERR_NULL;
  if (num_layers <= 0 || B <= 0 || S <= 0 || nh <= 0 || H != nh * 64 || (H % 64) || (I % 64)) return VT_ERR_BAD_SHAPE;
  hipStream_t stream = (hipStream_t)stream_;
  const int M = B * S;
  const void* cur = x;
  for (int l = 0; l < num_layers; ++l) {
    const vt_layer_weights& w = layers[l];
    const vt_layer_acts& a = acts[l];
    if (!a.qkv || !a.ctx || !a.attn_pre || !a.attn_out || !a.mid || !a.out_pre || !a.out) return VT_ERR_NULL;
    int rc;
    rc = vt_gemm_dispatch(cur, H, w.w_qkv, H, w.b_qkv, nullptr, 0, a.qkv, 3L * H, M, 3 * H, H, VT_ACT_NONE, 0, 0, 0, stream);
    if (rc) return rc;
    rc = vt_attention_fwd_dispatch(a.qkv, 3L * H, mask, mask_additive, head_scale ? head_scale + (long)l * nh : nullptr, a.ctx, H,
                                   a.lse, B, S, nh, 64, stream);
    if (rc) return rc;
    rc = vt_gemm_dispatch(a.ctx, H, w.w_ao, H, w.b_ao, cur, H, a.attn_pre, H, M, H, H, VT_ACT_NONE, 0, 0, 0, stream);
    if (rc) return rc;
    rc = vt_layernorm_dispatch(a.attn_pre, H, a.attn_out, H, w.ln1_g, w.ln1_b, a.ln1_mean, a.ln1_rstd, M, H, ln_eps, 0, 0, stream);
    if (rc) return rc;
    rc = vt_gemm_dispatch(a.attn_out, H, w.w_in, H, w.b_in, nullptr, 0, a.mid, I, M, I, H, VT_ACT_GELU, 0, 0, 0, stream);
    if (rc) return rc;
    rc = vt_gemm_dispatch(a.mid, I, w.w_out, I, w.b_out, a.attn_out, H, a.out_pre, H, M, H, I, VT_ACT_NONE, 0, 0, 0, stream);
    if (rc) return rc;
    rc = vt_layernorm_dispatch(a.out_pre, H, a.out, H, w.ln2_g, w.ln2_b, a.ln2_mean, a.ln2_rstd, M, H, ln_eps, 0, 0, stream);
    if (rc) return rc;
    cur = a.out;
  }
  return VT_OK;
}

}  // extern "C"
