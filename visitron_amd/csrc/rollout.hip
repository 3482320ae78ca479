// Kernels of the rollout caller around the trunk: tasks/viewpoint_select/agent_models.py
//   * OscarEncoder.forward :256-310  -- nn.LSTM over the packed trunk output (one lstm_step launch per position)
//   * SoftDotAttention.forward :328-357  -- softdot_attention
//   * AttnDecoderLSTM.forward :384-428  -- nn.LSTMCell = one lstm_step, three softdot_attention
// The dense projections of these modules go through the NT GEMM (gemm_bf16.hip); what is here is the part with no
// GEMM shape: the recurrent product h . W_hh^T fused with the gate arithmetic, and the batched dot / softmax / weighted
// sum over a short context.
#include "common.hpp"
#include "rollout_args.hpp"

// ---------------------------------------------------------------------------------------------
// One LSTM time step (torch.nn.LSTM / LSTMCell semantics, gate order i, f, g, o):
//   gates = xproj + h_prev . W_hh^T          xproj = x . W_ih^T + b_ih + b_hh (the caller's GEMM)
//   c' = sigmoid(f) * c + sigmoid(i) * tanh(g);  h' = sigmoid(o) * tanh(c')
// Packed-sequence rule (pack_padded_sequence / pad_packed_sequence, agent_models.py:286-301): a row with
// t >= lengths[b] keeps its state and its output position is zero.
//
// One workgroup = 16 hidden units x 16 batch rows; its 4 waves split K = hs four ways and each runs the 4 gates'
// 16x16x32 MFMAs with W_hh rows on the A port and the h rows on the B port (both straight from global / L2: W_hh is
// 2 MiB at hs = 512 and is re-read every step), so a lane ends with D[hidden = 4*(lane/16)+r][batch = lane%16].
// The partial sums meet in LDS and each of the 256 threads finishes one (hidden, batch) element.

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs a) {
  __shared__ f32x4 part[4][4][64];   // [wave][gate][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int hs = a.hs;
  const int kq = hs >> 2;                       // this wave's K range: [wave*kq, (wave+1)*kq), kq % 32 == 0
  const int kl = (lane >> 4) * 8;               // the lane's 8 consecutive k inside a 32-wide MFMA step
  const int brow = b0 + (lane & 15);
  const bool bvalid = brow < a.B;
  const float* hp = a.h_prev + (long)(bvalid ? brow : 0) * hs + wave * kq + kl;
  const bf16_t* wp = a.w_hh + (long)(h0 + (lane & 15)) * hs + wave * kq + kl;
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < kq; k += 32) {
    const f32x4 x0 = *(const f32x4*)(hp + k), x1 = *(const f32x4*)(hp + k + 4);
    u32x4 hb;
    hb[0] = pack_bf16x2(x0[0], x0[1]); hb[1] = pack_bf16x2(x0[2], x0[3]);
    hb[2] = pack_bf16x2(x1[0], x1[1]); hb[3] = pack_bf16x2(x1[2], x1[3]);
    if (!bvalid) hb = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 wv = *(const u32x4*)(wp + (long)g * hs * hs + k);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, hb),
                                                       acc[g], 0, 0, 0);
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) part[wave][g][lane] = acc[g];
  __syncthreads();
  // thread (lane, r = wave): element hidden = h0 + 4*(lane/16) + r, batch = b0 + lane%16
  const int r = wave;
  const int hid = h0 + 4 * (lane >> 4) + r;
  if (!bvalid) return;
  const long e = (long)brow * hs + hid;
  const bool active = a.lengths == nullptr || a.t < a.lengths[brow];
  float gate[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    // (a row past its length has no input projection in the compacted layout: nothing is read for it)
    const long xr = a.xrow_start ? ((long)a.xrow_start[brow] + a.t) * a.ldx_row : (long)brow * a.ldx;
    float s = active ? a.xproj[xr + (long)g * hs + hid] : 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += ((const float*)&part[w][g][lane])[r];
    gate[g] = s;
  }
  if (active) {
    const float ig = sigmoidf_(gate[0]), fg = sigmoidf_(gate[1]), gg = tanhf_(gate[2]), og = sigmoidf_(gate[3]);
    const float cp = a.c[e];
    const float cn = fg * cp + ig * gg;
    const float hn = og * tanhf_(cn);
    if (a.sv_gates) {   // training: what the backward step needs (lstm_step_bwd_kernel)
      float* sg = a.sv_gates + (long)brow * a.ld_svg + hid;
      sg[0] = ig; sg[hs] = fg; sg[2 * hs] = gg; sg[3 * hs] = og;
    }
    if (a.sv_c) a.sv_c[(long)brow * a.ld_svc + hid] = cp;
    if (a.sv_h) a.sv_h[(long)brow * a.ld_svh + hid] = f32_to_bf16(a.h_prev[e]);
    a.c[e] = cn;
    a.h_out[e] = hn;
    if (a.seq_out) a.seq_out[(long)brow * a.ld_seq + hid] = hn;
  } else {
    a.h_out[e] = a.h_prev[e];
    if (a.seq_out) a.seq_out[(long)brow * a.ld_seq + hid] = 0.f;
  }
}

int vt_lstm_step_dispatch(const LstmStepArgs& a, hipStream_t stream) {
  if (!a.xproj || !a.h_prev || !a.h_out || !a.c || !a.w_hh) return VT_ERR_NULL;
  if (a.B <= 0 || a.hs <= 0 || (a.hs % 128) != 0 || a.t < 0) return VT_ERR_BAD_SHAPE;   // 4 waves x 32-wide MFMA steps
  if (a.h_prev == a.h_out) return VT_ERR_UNSUPPORTED;
  if ((((uintptr_t)a.h_prev) | ((uintptr_t)a.w_hh)) & 15) return VT_ERR_BAD_ALIGN;
  hipLaunchKernelGGL(lstm_step_kernel, dim3(a.hs / 16, (a.B + 15) / 16), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// One step of back-propagation through time for that recurrence (the reference gets it from autograd through nn.LSTM /
// nn.LSTMCell: agent.py:493-518 back-propagates the rollout loss through OscarEncoder and AttnDecoderLSTM).
// Position t, with the gradient flowing in from the step that consumed h_t (position t_next):
//   dh  = d_out[t] + (row active at t_next ? dgates[t_next] . W_hh : dh_final)
//   dc += dh * o * (1 - tanh(c')^2);  do = dh * tanh(c');  di = dc * g;  dg = dc * i;  df = dc * c_prev;  dc <- dc * f
//   dgates[t] = (di i(1-i), df f(1-f), dg (1-g^2), do o(1-o))     (pre-activation; bf16: it is the MFMA operand of the
//   next step here and of the weight-gradient / input-gradient GEMMs afterwards)
// A row with t >= lengths[b] passed its state through: its dgates are zero and dc stays.
// Same decomposition as the forward step: workgroup = 16 hidden units x 16 batch rows, the 4 waves split K = 4*hs,
// W_hh^T rows on the MFMA A port, the bf16 gate gradients of t_next on the B port.
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(LstmBwdArgs a) {
  __shared__ f32x4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int hs = a.hs;
  const int kl = (lane >> 4) * 8;
  const int brow = b0 + (lane & 15);
  const bool bvalid = brow < a.B;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (a.dg_next) {
    const bf16_t* gp = a.dg_next + (long)(bvalid ? brow : 0) * a.ld_dgn + wave * hs + kl;
    const bf16_t* wp = a.w_hh_t + (long)(h0 + (lane & 15)) * (4L * hs) + wave * hs + kl;
    for (int k = 0; k < hs; k += 32) {
      u32x4 gv = *(const u32x4*)(gp + k);
      if (!bvalid) gv = (u32x4){0u, 0u, 0u, 0u};
      const u32x4 wv = *(const u32x4*)(wp + k);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, gv), acc, 0, 0, 0);
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  const int r = wave;
  const int hid = h0 + 4 * (lane >> 4) + r;
  if (!bvalid) return;
  const long e = (long)brow * hs + hid;
  const int len = a.lengths ? a.lengths[brow] : 0x7fffffff;
  bf16_t* dg = a.dg_out + (long)brow * a.ld_dg + hid;
  float* dgf = a.dg_out_f32 ? a.dg_out_f32 + (long)brow * a.ld_dgf + hid : nullptr;
  float d[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.t < len) {
    float dh;
    if (a.dg_next && a.t_next >= 0 && a.t_next < len) {
      dh = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) dh += ((const float*)&part[w][lane])[r];
    } else {
      dh = a.dh_final ? a.dh_final[e] : 0.f;
    }
    if (a.d_out) dh += a.d_out[(long)brow * a.ld_dout + hid];
    const float* sg = a.sv_gates + (long)brow * a.ld_svg + hid;
    const float ig = sg[0], fg = sg[hs], gg = sg[2 * hs], og = sg[3 * hs];
    const float cp = a.sv_c[(long)brow * a.ld_svc + hid];
    const float tc = tanhf_(fg * cp + ig * gg);
    const float dc = a.dc[e] + dh * og * (1.0f - tc * tc);
    d[0] = dc * gg * ig * (1.0f - ig);
    d[1] = dc * cp * fg * (1.0f - fg);
    d[2] = dc * ig * (1.0f - gg * gg);
    d[3] = dh * tc * og * (1.0f - og);
    a.dc[e] = dc * fg;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    dg[(long)g * hs] = f32_to_bf16(d[g]);
    if (dgf) dgf[(long)g * hs] = d[g];
  }
}

int vt_lstm_step_bwd_dispatch(const LstmBwdArgs& a, hipStream_t stream) {
  if (!a.w_hh_t || !a.dc || !a.sv_gates || !a.sv_c || !a.dg_out) return VT_ERR_NULL;
  if (a.B <= 0 || a.hs <= 0 || (a.hs % 128) != 0 || a.t < 0) return VT_ERR_BAD_SHAPE;
  if ((const void*)a.dg_next == (const void*)a.dg_out) return VT_ERR_UNSUPPORTED;   // other workgroups still read it
  if (a.dg_next && ((a.ld_dgn & 7) || (((uintptr_t)a.dg_next) & 15))) return VT_ERR_BAD_ALIGN;
  if (((uintptr_t)a.w_hh_t) & 15) return VT_ERR_BAD_ALIGN;
  hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(a.hs / 16, (a.B + 15) / 16), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// SoftDotAttention.forward (agent_models.py:328-357) after linear_in:
//   attn[b, l] = context[b, l, :] . target[b, :]                     (:338, torch.bmm)
//   mask != 0 -> -inf  (in place, so the returned "logit" alias is masked too, :339-343)
//   p = softmax(attn) over l (:344)                                   weighted[b, :] = sum_l p[l] context[b, l, :] (:349)
// One workgroup per batch row; logits / probabilities live in LDS.  fp32 throughout (the context is the caller's
// fp32 feature tensor: the kernel is a pure read of it, twice, the second time from L2).

#define SD_WAVES 16
__global__ __launch_bounds__(64 * SD_WAVES) void softdot_kernel(SoftDotArgs a, int parts) {
  extern __shared__ float sl[];          // [L] logits -> probabilities | [2 * SD_WAVES] reduction scratch | [parts][D] partial sums
  float* red = sl + a.L;
  float* psum = red + 2 * SD_WAVES;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* ctx = a.context + (long)b * a.ld_batch;
  const float* tg = a.target + (long)b * a.D;
  const bool vec = ((a.D & 3) == 0) && ((a.ld_row & 3) == 0) && ((a.ld_batch & 3) == 0) &&
                   ((((uintptr_t)a.context) | ((uintptr_t)a.target)) & 15) == 0;
  for (int l = wave; l < a.L; l += SD_WAVES) {
    const float* row = ctx + (long)l * a.ld_row;
    float s = 0.f;
    if (vec) {
      for (int d = lane * 4; d < a.D; d += 256) {
        const f32x4 c = *(const f32x4*)(row + d), t = *(const f32x4*)(tg + d);
        s += c[0] * t[0] + c[1] * t[1] + c[2] * t[2] + c[3] * t[3];
      }
    } else {
      for (int d = lane; d < a.D; d += 64) s += row[d] * tg[d];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
      if (a.mask && a.mask[(long)b * a.L + l]) s = -INFINITY;
      sl[l] = s;
    }
  }
  __syncthreads();
  if (a.attn && !a.output_prob)
    for (int l = tid; l < a.L; l += 64 * SD_WAVES) a.attn[(long)b * a.L + l] = sl[l];
  if (!a.weighted && !(a.attn && a.output_prob)) return;
  // softmax over L
  float m = -INFINITY;
  for (int l = tid; l < a.L; l += 64 * SD_WAVES) m = fmaxf(m, sl[l]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < SD_WAVES; ++w) m = fmaxf(m, red[w]);
  float z = 0.f;
  for (int l = tid; l < a.L; l += 64 * SD_WAVES) {
    const float e = expf(sl[l] - m);      // every key masked: -inf - -inf = NaN, as torch's softmax gives
    sl[l] = e;
    z += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
  if (lane == 0) red[SD_WAVES + wave] = z;
  __syncthreads();
  z = 0.f;
#pragma unroll
  for (int w = 0; w < SD_WAVES; ++w) z += red[SD_WAVES + w];
  const float inv = 1.0f / z;
  for (int l = tid; l < a.L; l += 64 * SD_WAVES) {
    const float p = sl[l] * inv;
    sl[l] = p;
    if (a.attn && a.output_prob) a.attn[(long)b * a.L + l] = p;
  }
  __syncthreads();
  if (!a.weighted) return;
  // weighted[d] = sum_l p[l] ctx[l][d]: thread = (column group g, key part): part sums every parts-th key, the parts
  // meet in LDS.  A masked key carries probability exactly 0 and is skipped (0 * inf must not appear).
  const int gw = vec ? 4 : 1;
  const int G = (a.D + gw - 1) / gw;
  const int g = tid % G, part = tid / G;
  if (tid < G * parts) {
    if (vec) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int l = part; l < a.L; l += parts) {
        const float p = sl[l];
        if (p != 0.f) {
          const f32x4 c = *(const f32x4*)(ctx + (long)l * a.ld_row + 4 * g);
          acc[0] += p * c[0]; acc[1] += p * c[1]; acc[2] += p * c[2]; acc[3] += p * c[3];
        }
      }
      *(f32x4*)(psum + (long)part * a.D + 4 * g) = acc;
    } else {
      float acc = 0.f;
      for (int l = part; l < a.L; l += parts) {
        const float p = sl[l];
        if (p != 0.f) acc += p * ctx[(long)l * a.ld_row + g];
      }
      psum[(long)part * a.D + g] = acc;
    }
  }
  __syncthreads();
  for (int d = tid; d < a.D; d += 64 * SD_WAVES) {
    float acc = 0.f;
    for (int q = 0; q < parts; ++q) acc += psum[(long)q * a.D + d];
    a.weighted[(long)b * a.D + d] = acc;
  }
}

int vt_softdot_dispatch(const SoftDotArgs& a, hipStream_t stream) {
  if (!a.target || !a.context) return VT_ERR_NULL;
  if (!a.weighted && !a.attn) return VT_ERR_NULL;
  if (a.B <= 0 || a.L <= 0 || a.D <= 0 || a.L > 8192) return VT_ERR_BAD_SHAPE;
  if (a.weighted && (((uintptr_t)a.weighted) & 15)) return VT_ERR_BAD_ALIGN;
  const bool vec = ((a.D & 3) == 0) && ((a.ld_row & 3) == 0) && ((a.ld_batch & 3) == 0) &&
                   ((((uintptr_t)a.context) | ((uintptr_t)a.target)) & 15) == 0;
  const int G = vec ? a.D / 4 : a.D;
  if (G > 64 * SD_WAVES) return VT_ERR_BAD_SHAPE;   // D <= 4096 (1024 unaligned)
  int parts = (64 * SD_WAVES) / G;
  if (parts > 8) parts = 8;                          // 8 x D floats of LDS
  if (parts > a.L) parts = a.L;
  const size_t lds = ((size_t)a.L + 2 * SD_WAVES + (size_t)parts * a.D) * sizeof(float);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)softdot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VT_ERR_HIP;
  hipLaunchKernelGGL(softdot_kernel, dim3(a.B), dim3(64 * SD_WAVES), lds, stream, a, parts);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Gradient of the same block (autograd through agent_models.py:336-352 in the reference): with p = softmax(masked logits),
//   dp[l]   = context[l] . d_weighted  (+ d_attn[l] when the probabilities were returned)
//   dlog[l] = p[l] (dp[l] - sum_j p[j] dp[j])  (+ d_attn[l] when the masked logits were returned; a masked key gets 0:
//             masked_fill_ cut it out of the graph)
//   d_target = sum_l dlog[l] context[l];   d_context[l] = p[l] d_weighted + dlog[l] target
// One workgroup per batch row; logits and dp in LDS from one pass over the context, d_target / d_context from a second.
__global__ __launch_bounds__(64 * SD_WAVES) void softdot_bwd_kernel(SoftDotBwdArgs a, int parts) {
  extern __shared__ float sl[];          // [L] logits -> p | [L] dp -> dlog | [2 * SD_WAVES] scratch | [parts][D] partial sums
  float* ql = sl + a.L;
  float* red = ql + a.L;
  float* psum = red + 2 * SD_WAVES;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* ctx = a.context + (long)b * a.ld_batch;
  const float* tg = a.target + (long)b * a.D;
  const float* dw = a.d_weighted ? a.d_weighted + (long)b * a.D : nullptr;
  const float* da = a.d_attn ? a.d_attn + (long)b * a.L : nullptr;
  const unsigned char* mk = a.mask ? a.mask + (long)b * a.L : nullptr;
  for (int l = wave; l < a.L; l += SD_WAVES) {
    const float* row = ctx + (long)l * a.ld_row;
    float s = 0.f, q = 0.f;
    for (int d = lane; d < a.D; d += 64) {
      const float c = row[d];
      s += c * tg[d];
      if (dw) q += c * dw[d];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane == 0) {
      if (mk && mk[l]) s = -INFINITY;
      sl[l] = s;
      ql[l] = q + ((da && a.output_prob) ? da[l] : 0.f);
    }
  }
  __syncthreads();
  float m = -INFINITY;
  for (int l = tid; l < a.L; l += 64 * SD_WAVES) m = fmaxf(m, sl[l]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < SD_WAVES; ++w) m = fmaxf(m, red[w]);
  float z = 0.f;
  for (int l = tid; l < a.L; l += 64 * SD_WAVES) {
    const float e = expf(sl[l] - m);
    sl[l] = e;
    z += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
  if (lane == 0) red[SD_WAVES + wave] = z;
  __syncthreads();
  z = 0.f;
#pragma unroll
  for (int w = 0; w < SD_WAVES; ++w) z += red[SD_WAVES + w];
  const float inv = 1.0f / z;
  __syncthreads();                       // red is reused below
  float pd = 0.f;
  for (int l = tid; l < a.L; l += 64 * SD_WAVES) {
    const float p = sl[l] * inv;
    sl[l] = p;
    pd += p * ql[l];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pd += __shfl_xor(pd, o, 64);
  if (lane == 0) red[wave] = pd;
  __syncthreads();
  pd = 0.f;
#pragma unroll
  for (int w = 0; w < SD_WAVES; ++w) pd += red[w];
  for (int l = tid; l < a.L; l += 64 * SD_WAVES) {
    float g = sl[l] * (ql[l] - pd);
    if (da && !a.output_prob && !(mk && mk[l])) g += da[l];
    ql[l] = g;
  }
  __syncthreads();
  // thread = (key part, column): d_target partial sums over every parts-th key, d_context written on the way
  const int cols = (64 * SD_WAVES) / parts;
  const int part = tid / cols;
  if (part < parts) {
    for (int d = tid % cols; d < a.D; d += cols) {
      float acc = 0.f;
      const float t = tg[d], w = dw ? dw[d] : 0.f;
      float* dcx = a.d_context ? a.d_context + ((long)b * a.L) * a.D + d : nullptr;
      for (int l = part; l < a.L; l += parts) {
        const float g = ql[l];
        if (g != 0.f) acc += g * ctx[(long)l * a.ld_row + d];    // a masked key has g == 0 exactly (and may hold anything)
        if (dcx) dcx[(long)l * a.D] = sl[l] * w + g * t;
      }
      psum[(long)part * a.D + d] = acc;
    }
  }
  __syncthreads();
  for (int c = tid; c < a.D; c += 64 * SD_WAVES) {
    float acc = 0.f;
    for (int q = 0; q < parts; ++q) acc += psum[(long)q * a.D + c];
    a.d_target[(long)b * a.D + c] = acc;
  }
}

// The same gradient for a long context (the decoder's attention over the 511 instruction positions: one workgroup per
// batch row leaves 3/4 of the chip idle and walks 1 MB three times) as two launches spread over keys:
//   softdot_bwd_dots  : one wave per key -> logit[b,l] (masked: -inf) and dp[b,l]            grid (L / 4 keys, B)
//   softdot_bwd_apply : every workgroup redoes the row's softmax statistics from those two [L] vectors (cheap), then its
//                       SDB_KEYS keys: d_context rows and a partial d_target                    grid (L / SDB_KEYS, B)
// the partial d_target sums [B, chunks, D] are added up by the caller (no atomics: the result is reproducible).
#define SDB_KEYS 32
__global__ __launch_bounds__(256) void softdot_bwd_dots(SoftDotBwdArgs a, float* sl_g, float* ql_g) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = blockIdx.x * 4 + wave;
  if (l >= a.L) return;
  const float* row = a.context + (long)b * a.ld_batch + (long)l * a.ld_row;
  const float* tg = a.target + (long)b * a.D;
  const float* dw = a.d_weighted ? a.d_weighted + (long)b * a.D : nullptr;
  float s = 0.f, q = 0.f;
  for (int d = lane; d < a.D; d += 64) {
    const float c = row[d];
    s += c * tg[d];
    if (dw) q += c * dw[d];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if (lane == 0) {
    if (a.mask && a.mask[(long)b * a.L + l]) s = -INFINITY;
    if (a.d_attn && a.output_prob) q += a.d_attn[(long)b * a.L + l];
    sl_g[(long)b * a.L + l] = s;
    ql_g[(long)b * a.L + l] = q;
  }
}

__global__ __launch_bounds__(256) void softdot_bwd_apply(SoftDotBwdArgs a, const float* sl_g, const float* ql_g,
                                                         float* dt_part, int chunks) {
  __shared__ float red[8];
  __shared__ float gk[SDB_KEYS], pk[SDB_KEYS];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sl = sl_g + (long)b * a.L;
  const float* ql = ql_g + (long)b * a.L;
  float m = -INFINITY;
  for (int l = tid; l < a.L; l += 256) m = fmaxf(m, sl[l]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float z = 0.f, pd = 0.f;
  for (int l = tid; l < a.L; l += 256) {
    const float e = expf(sl[l] - m);
    z += e;
    pd += e * ql[l];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { z += __shfl_xor(z, o, 64); pd += __shfl_xor(pd, o, 64); }
  __syncthreads();
  if (lane == 0) { red[wave] = z; red[4 + wave] = pd; }
  __syncthreads();
  z = red[0] + red[1] + red[2] + red[3];
  pd = (red[4] + red[5] + red[6] + red[7]) / z;
  const float inv = 1.0f / z;
  const int l0 = chunk * SDB_KEYS;
  const int nk = min(SDB_KEYS, a.L - l0);
  if (tid < nk) {
    const int l = l0 + tid;
    const float p = expf(sl[l] - m) * inv;
    float g = p * (ql[l] - pd);
    if (a.d_attn && !a.output_prob && !(a.mask && a.mask[(long)b * a.L + l])) g += a.d_attn[(long)b * a.L + l];
    pk[tid] = p;
    gk[tid] = g;
  }
  __syncthreads();
  const float* ctx = a.context + (long)b * a.ld_batch + (long)l0 * a.ld_row;
  const float* tg = a.target + (long)b * a.D;
  const float* dw = a.d_weighted ? a.d_weighted + (long)b * a.D : nullptr;
  float* dcx = a.d_context ? a.d_context + ((long)b * a.L + l0) * a.D : nullptr;
  for (int d = tid; d < a.D; d += 256) {
    const float t = tg[d], w = dw ? dw[d] : 0.f;
    float acc = 0.f;
    for (int k = 0; k < nk; ++k) {
      const float g = gk[k];
      if (g != 0.f) acc += g * ctx[(long)k * a.ld_row + d];
      if (dcx) dcx[(long)k * a.D + d] = pk[k] * w + g * t;
    }
    dt_part[((long)b * chunks + chunk) * a.D + d] = acc;
  }
}

// split form: scratch = 2 * B * L floats (logits, dp) + B * chunks * D floats (partial d_target), chunks = ceil(L / SDB_KEYS)
long vt_softdot_bwd_split_ws_floats(int B, int L, int D) {
  return 2L * B * L + (long)B * ((L + SDB_KEYS - 1) / SDB_KEYS) * D;
}

int vt_softdot_bwd_split_dispatch(const SoftDotBwdArgs& a, float* ws, hipStream_t stream) {
  if (!a.target || !a.context || !ws) return VT_ERR_NULL;
  if (!a.d_weighted && !a.d_attn) return VT_ERR_NULL;
  if (a.B <= 0 || a.L <= 0 || a.D <= 0 || a.B > 65535) return VT_ERR_BAD_SHAPE;
  const int chunks = (a.L + SDB_KEYS - 1) / SDB_KEYS;
  float* sl = ws;
  float* ql = ws + (long)a.B * a.L;
  float* part = ql + (long)a.B * a.L;
  hipLaunchKernelGGL(softdot_bwd_dots, dim3((a.L + 3) / 4, a.B), dim3(256), 0, stream, a, sl, ql);
  hipLaunchKernelGGL(softdot_bwd_apply, dim3(chunks, a.B), dim3(256), 0, stream, a, (const float*)sl, (const float*)ql, part,
                     chunks);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_softdot_bwd_dispatch(const SoftDotBwdArgs& a, hipStream_t stream) {
  if (!a.target || !a.context || !a.d_target) return VT_ERR_NULL;
  if (!a.d_weighted && !a.d_attn) return VT_ERR_NULL;
  if (a.B <= 0 || a.L <= 0 || a.D <= 0 || a.L > 8192) return VT_ERR_BAD_SHAPE;
  int parts = (64 * SD_WAVES) / a.D;                 // key parts when a row has fewer columns than the workgroup threads
  if (parts > 8) parts = 8;
  if (parts > a.L) parts = a.L;
  if (parts < 1) parts = 1;
  const size_t lds = (2 * (size_t)a.L + 2 * SD_WAVES + (size_t)parts * a.D) * sizeof(float);
  if (lds > 160 * 1024) return VT_ERR_BAD_SHAPE;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)softdot_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VT_ERR_HIP;
  hipLaunchKernelGGL(softdot_bwd_kernel, dim3(a.B), dim3(64 * SD_WAVES), lds, stream, a, parts);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Dense layer on a handful of rows (the decoder step's projections, agent_models.py:406-425: 64 or fewer rows against
// weight matrices of 0.1-9 MB): out = act([x0 | x1] . W^T + bias) in fp32 from fp32 activations and bf16 weights.
// The 256x256-tile GEMM would run such a shape on N/256 workgroups; here one workgroup = 16 output columns x 16 rows,
// its 4 waves take every fourth 32-wide K-step (W rows on the MFMA A port, the activations converted to bf16 on the B
// port, both straight from L2), partial sums meet in LDS.  Also folds the K-concatenation (torch.cat of the weighted
// context and the query, :354) and the bf16 packing that the big kernel needs as separate launches.
__global__ __launch_bounds__(256) void skinny_linear_kernel(SkinnyArgs a) {
  __shared__ f32x4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int kl = (lane >> 4) * 8;
  int wrow = n0 + (lane & 15);
  wrow = wrow < a.N ? wrow : a.N - 1;
  const int xrow = m0 + (lane & 15);
  const bool xok = xrow < a.M;
  const bf16_t* wp = a.w + (long)wrow * a.ldw + kl;
  const float* p0 = a.x0 + (long)(xok ? xrow : 0) * a.ld0;
  const float* p1 = a.x1 ? a.x1 + (long)(xok ? xrow : 0) * a.ld1 : nullptr;
  const int K = a.K0 + a.K1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = wave * 32; k < a.Kpad; k += 128) {
    const int kk = k + kl;
    f32x4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = kk + 4 * h;
      v[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (xok && c < K) v[h] = c < a.K0 ? *(const f32x4*)(p0 + c) : *(const f32x4*)(p1 + (c - a.K0));   // K0, K % 4 == 0
    }
    u32x4 xb;
    xb[0] = pack_bf16x2(v[0][0], v[0][1]); xb[1] = pack_bf16x2(v[0][2], v[0][3]);
    xb[2] = pack_bf16x2(v[1][0], v[1][1]); xb[3] = pack_bf16x2(v[1][2], v[1][3]);
    const u32x4 wv = *(const u32x4*)(wp + k);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
  }
  part[wave][lane] = acc;
  __syncthreads();
  // thread (lane, r = wave): column n0 + 4*(lane/16) + r of row m0 + lane%16
  const int n = n0 + 4 * (lane >> 4) + wave;
  if (!xok || n >= a.N) return;
  float s = a.bias ? a.bias[n] : 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) s += ((const float*)&part[w][lane])[wave];
  if (a.act == 2) s = tanhf_(s);
  a.out[(long)xrow * a.ldo + n] = s;
}

int vt_skinny_linear_dispatch(const SkinnyArgs& a, hipStream_t stream) {
  if (!a.x0 || !a.w || !a.out) return VT_ERR_NULL;
  if (a.M <= 0 || a.N <= 0 || a.K0 <= 0 || a.K1 < 0 || (a.K1 > 0 && !a.x1)) return VT_ERR_BAD_SHAPE;
  if ((a.K0 & 3) || (a.K1 & 3) || (a.Kpad & 31) || a.Kpad < a.K0 + a.K1 || a.ldw < a.Kpad) return VT_ERR_BAD_SHAPE;
  if ((a.ld0 & 3) || (a.K1 && (a.ld1 & 3)) || (a.ldw & 7)) return VT_ERR_BAD_ALIGN;
  if ((((uintptr_t)a.x0) | ((uintptr_t)a.x1) | ((uintptr_t)a.w)) & 15) return VT_ERR_BAD_ALIGN;
  if (a.act != 0 && a.act != 2) return VT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(skinny_linear_kernel, dim3((a.N + 15) / 16, (a.M + 15) / 16), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
