// Pretrain input preparation on the device: what PretrainDataset does per item in Python before the
// hot path -- tasks/viewpoint_select/data_loader_pretrain.py:549-613 (_mask_tokens) and :615-712 (region features,
// location embeddings, padding / truncation, label and attention-mask assembly) -- as two HBM-bound kernels over a whole
// batch.  Integer work: outputs are bit-exact against the per-item restatement (oracle/data.py) on shared random draws.
#include "common.hpp"

// ---- _mask_tokens (data_loader_pretrain.py:549-613), one thread per token ------------------------------------------
//   masked   = bernoulli(mlm_probability, 0 on special tokens)  [u_mask < p]   | token-class positions (forced)
//   labels   = masked ? id : -1 ; token-class positions -> -1 (they are supervised by the token head instead)
//   inputs   = [MASK] where (u_replace < 0.8 & masked) or a token class is set;
//              a random word where (u_random < 0.5 & masked & not replaced); else unchanged
//   attention_mask = id != pad_id
struct MaskTokensArgs {
  const int64_t* ids; const uint8_t* special; const int64_t* token_classes;   // token_classes may be null
  const float* u_mask; const float* u_replace; const float* u_random; const int64_t* random_words;
  int64_t* out_ids; int64_t* labels; int64_t* attention_mask;
  long n; int64_t pad_id, mask_id; float mlm_probability;
};

__global__ __launch_bounds__(256) void mask_tokens_kernel(MaskTokensArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int64_t id = a.ids[i];
  const float prob = a.special[i] ? 0.0f : a.mlm_probability;
  bool masked = a.u_mask[i] < prob;
  const bool tcm = a.token_classes && a.token_classes[i] != -1;
  masked = masked || tcm;
  int64_t lab = masked ? id : (int64_t)-1;
  if (tcm) lab = -1;
  bool replaced = (a.u_replace[i] < 0.8f) && masked;
  int64_t out = replaced ? a.mask_id : id;
  if (tcm) { replaced = true; out = a.mask_id; }
  const bool rnd = (a.u_random[i] < 0.5f) && masked && !replaced;
  if (rnd) out = a.random_words[i];
  a.out_ids[i] = out;
  a.labels[i] = lab;
  a.attention_mask[i] = id != a.pad_id ? 1 : 0;
}

int vt_mask_tokens_dispatch(const int64_t* ids, const uint8_t* special, const int64_t* token_classes, const float* u_mask,
                            const float* u_replace, const float* u_random, const int64_t* random_words, int64_t* out_ids,
                            int64_t* labels, int64_t* attention_mask, long n, int64_t pad_id, int64_t mask_id,
                            float mlm_probability, hipStream_t stream) {
  if (!ids || !special || !u_mask || !u_replace || !u_random || !random_words || !out_ids || !labels || !attention_mask) return VT_ERR_NULL;
  if (n <= 0) return VT_ERR_BAD_SHAPE;
  MaskTokensArgs a;
  a.ids = ids; a.special = special; a.token_classes = token_classes; a.u_mask = u_mask; a.u_replace = u_replace;
  a.u_random = u_random; a.random_words = random_words; a.out_ids = out_ids; a.labels = labels; a.attention_mask = attention_mask;
  a.n = n; a.pad_id = pad_id; a.mask_id = mask_id; a.mlm_probability = mlm_probability;
  hipLaunchKernelGGL(mask_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---- tail of _preprocess_item (data_loader_pretrain.py:627-633, 654-712), one workgroup per output region row --------
// Item b holds n_b = min(region_counts[b], R_in) region rows; more than R keep their LAST R rows (:660-665), fewer are
// zero-padded with attention mask 0 (:669-689).  Output row (b, r) <- input row start_b + r, start_b = max(n_b - R, 0):
//   img_feats_out [B, R, D]   = the feature row (or zeros)
//   loc_out       [B, R, 128] = _static_loc_embeddings[current_view[b]][region_view_ids[b, row]] (:25-49, :627-633) (or zeros)
//   attention_mask[b, T + r]  = row valid;  labels / token_labels [b, T + r] = -1 (:691-700)
// and, by the first R workgroups' spare lanes, nothing else: the text part of the [B, T + R] tensors is copied by rows 0.
struct AssembleArgs {
  const float* img_feats; const int64_t* region_counts; const int64_t* region_view_ids; const int64_t* current_view;
  const float* loc_table;   // [36, 36, 128]
  const int64_t* text_labels; const int64_t* text_mask; const int64_t* text_token_classes;   // [B, T] (token classes may be null)
  float* feats_out; float* loc_out; int64_t* labels_out; int64_t* mask_out; int64_t* token_labels_out;   // last may be null
  int B, T, R, R_in, D;
};

__global__ __launch_bounds__(256) void assemble_regions_kernel(AssembleArgs a) {
  const int b = blockIdx.y, r = blockIdx.x, tid = threadIdx.x;
  const int S = a.T + a.R;
  if (r == a.R) {   // one extra workgroup per item copies the text part of the label / mask rows
    for (int t = tid; t < a.T; t += 256) {
      a.labels_out[(long)b * S + t] = a.text_labels[(long)b * a.T + t];
      a.mask_out[(long)b * S + t] = a.text_mask[(long)b * a.T + t] ? 1 : 0;
      if (a.token_labels_out) a.token_labels_out[(long)b * S + t] = a.text_token_classes[(long)b * a.T + t];
    }
    return;
  }
  int64_t n = a.region_counts[b];
  n = n < 0 ? 0 : (n > a.R_in ? a.R_in : n);
  const int64_t start = n > a.R ? n - a.R : 0;
  const int64_t src = start + r;
  const bool valid = src < n;
  float* fo = a.feats_out + ((long)b * a.R + r) * a.D;
  const float* fi = a.img_feats + ((long)b * a.R_in + (valid ? src : 0)) * a.D;
  for (int c = tid; c < a.D; c += 256) fo[c] = valid ? fi[c] : 0.0f;
  if (tid < 128) {
    float v = 0.0f;
    if (valid) {
      int64_t cur = a.current_view[b], vid = a.region_view_ids[(long)b * a.R_in + src];
      cur = cur < 0 ? 0 : (cur > 35 ? 35 : cur);
      vid = vid < 0 ? 0 : (vid > 35 ? 35 : vid);
      v = a.loc_table[(cur * 36 + vid) * 128 + tid];
    }
    a.loc_out[((long)b * a.R + r) * 128 + tid] = v;
  }
  if (tid == 0) {
    a.mask_out[(long)b * S + a.T + r] = valid ? 1 : 0;
    a.labels_out[(long)b * S + a.T + r] = -1;
    if (a.token_labels_out) a.token_labels_out[(long)b * S + a.T + r] = -1;
  }
}

int vt_assemble_regions_dispatch(const float* img_feats, const int64_t* region_counts, const int64_t* region_view_ids,
                                 const int64_t* current_view, const float* loc_table, const int64_t* text_labels,
                                 const int64_t* text_mask, const int64_t* text_token_classes, float* feats_out, float* loc_out,
                                 int64_t* labels_out, int64_t* mask_out, int64_t* token_labels_out, int B, int T, int R,
                                 int R_in, int D, hipStream_t stream) {
  if (!region_counts || !current_view || !loc_table || !text_labels || !text_mask || !labels_out || !mask_out) return VT_ERR_NULL;
  if (R > 0 && (!feats_out || !loc_out || !region_view_ids || !img_feats)) return VT_ERR_NULL;
  if ((token_labels_out != nullptr) != (text_token_classes != nullptr)) return VT_ERR_NULL;
  if (B <= 0 || T <= 0 || R < 0 || R_in < 0 || D <= 0 || B > 65535) return VT_ERR_BAD_SHAPE;
  AssembleArgs a;
  a.img_feats = img_feats; a.region_counts = region_counts; a.region_view_ids = region_view_ids; a.current_view = current_view;
  a.loc_table = loc_table; a.text_labels = text_labels; a.text_mask = text_mask; a.text_token_classes = text_token_classes;
  a.feats_out = feats_out; a.loc_out = loc_out; a.labels_out = labels_out; a.mask_out = mask_out; a.token_labels_out = token_labels_out;
  a.B = B; a.T = T; a.R = R; a.R_in = R_in; a.D = D;
  hipLaunchKernelGGL(assemble_regions_kernel, dim3(R + 1, B), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
