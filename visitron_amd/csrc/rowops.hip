// HBM-bound row kernels of the hot path (one wave64 per row, 16-byte vector accesses, wave-shuffle
// reductions, fp32 statistics):
//   layernorm_rows      BertLayerNorm over a row that already holds dense(h)+bias+residual (the GEMM
//                       epilogue added them): (x-u)/sqrt(var+eps)*w+b, biased variance, eps inside the
//                       sqrt -- BertSelfOutput / BertOutput, called at oscar/modeling_bert.py:94,120;
//                       also the optional image LayerNorm, tasks/viewpoint_select/encoder.py:280-281.
//   embed_layernorm     BertEmbeddings: word + position + token_type gather-sum -> LayerNorm, called at
//                       tasks/viewpoint_select/encoder.py:267-269; writes rows b*S+t of the [B,S,H]
//                       sequence buffer directly (no torch.cat, :287).
//   pack_region_inputs  [img_feats | img_location_embeddings | 0-pad] fp32 -> bf16 K-concatenated GEMM
//                       operand, so img_embedding + location_embeds (encoder.py:277-279) is ONE GEMM.
#include "common.hpp"

__global__ void prefetch_ranges(PrefetchArgs a);   // (defined at the end of this file)

struct LnArgs {
  const bf16_t* x; long ldx;   // bf16, or fp16 (template parameter XF16: the training layer's pre-LayerNorm sums)
  bf16_t* y; long ldy;
  uint16_t* yh; long ldyh;     // optional second output: the same rows as fp16 (the next sub-layer's residual operand)
  const float* gamma; const float* beta;
  float* mean; float* rstd;  // optional [M] outputs for backward
  int M, H;
  int grp_rows, grp_stride;  // row remap as in the GEMM (0: identity); applies to x and y
  float eps;
  // layernorm_rows_full only: workgroups n_main .. gridDim.x - 1 do not normalise rows but read `pf`'s ranges (the weights of
  // the GEMMs that follow: vt_prefetch_role); n_main == gridDim.x and pf.n == 0 without a prefetch
  int n_main;
  PrefetchArgs pf;
};

template <int CH, bool XF16>
__global__ __launch_bounds__(256) void layernorm_rows(LnArgs a) {
  auto lo = [](uint32_t w) { return XF16 ? f16lo(w) : bf16lo(w); };
  auto hi = [](uint32_t w) { return XF16 ? f16hi(w) : bf16hi(w); };
  // One wave = LN_ROWS consecutive rows at once: their loads are all in flight together and the scale / shift vectors
  // (6 KiB of fp32 against 1.5 KiB per row) are fetched once per wave instead of once per row.
  constexpr int LN_ROWS = 4;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_ROWS;
  if (row0 >= a.M) return;
  u32x4 w[LN_ROWS][CH];
  long prow[LN_ROWS];
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    const int row = row0 + r < a.M ? row0 + r : a.M - 1;   // the tail repeats the last row (its store is skipped)
    prow[r] = a.grp_rows ? (long)(row / a.grp_rows) * a.grp_stride + (row % a.grp_rows) : (long)row;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      w[r][c] = col < a.H ? *(const u32x4*)(a.x + prow[r] * a.ldx + col) : (u32x4){0u, 0u, 0u, 0u};
    }
  }
  float gam[CH][8], bet[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (lane + 64 * c) * 8;
    if (col < a.H) {
      const f32x4 g0 = *(const f32x4*)(a.gamma + col), g1 = *(const f32x4*)(a.gamma + col + 4);
      const f32x4 b0 = *(const f32x4*)(a.beta + col), b1 = *(const f32x4*)(a.beta + col + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { gam[c][i] = g0[i]; gam[c][4 + i] = g1[i]; bet[c][i] = b0[i]; bet[c][4 + i] = b1[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { gam[c][i] = 0.f; bet[c][i] = 0.f; }
    }
  }
  const float invH = 1.0f / (float)a.H;
  float u[LN_ROWS], rs[LN_ROWS];
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) s += lo(w[r][c][i]) + hi(w[r][c][i]);   // columns past H hold zeros
    u[r] = s;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) u[r] += __shfl_xor(u[r], o, 64);
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    u[r] *= invH;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d0 = lo(w[r][c][i]) - u[r], d1 = hi(w[r][c][i]) - u[r];
          ss += d0 * d0 + d1 * d1;
        }
      }
    }
    rs[r] = ss;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) rs[r] += __shfl_xor(rs[r], o, 64);
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    const int row = row0 + r;
    if (row >= a.M) break;
    const float rstd = 1.0f / sqrtf(rs[r] * invH + a.eps);
    if (lane == 0) {
      if (a.mean) a.mean[row] = u[r];
      if (a.rstd) a.rstd[row] = rstd;
    }
    bf16_t* yp = a.y + prow[r] * a.ldy;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
        u32x4 o4, h4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float o0 = (lo(w[r][c][i]) - u[r]) * rstd * gam[c][2 * i] + bet[c][2 * i];
          const float o1 = (hi(w[r][c][i]) - u[r]) * rstd * gam[c][2 * i + 1] + bet[c][2 * i + 1];
          o4[i] = pack_bf16x2(o0, o1);
          h4[i] = pack_f16x2(o0, o1);
        }
        *(u32x4*)(yp + col) = o4;
        if (a.yh) *(u32x4*)(a.yh + prow[r] * a.ldyh + col) = h4;
      }
    }
  }
}

// The same forward for H == 512 * C8 + 256 * C4 exactly and no row remap, laid out like layernorm_bwd_rows_full below: every
// lane owns 8 * C8 + 4 * C4 columns (H = 768: 12 columns on all 64 lanes instead of 8 + 8 with half the wave idle in the second
// chunk), a wave walks rows R at a time over a fixed grid with the next R rows' loads issued before the current rows'
// arithmetic, and the two reductions per row run on the vector ALU (wave_sum_valu).
template <int C8, int C4, int R, bool XF16>
__global__ __launch_bounds__(256) void layernorm_rows_full(LnArgs a) {
  constexpr int E = 8 * C8 + 4 * C4, H = 512 * C8 + 256 * C4, W = E / 2;
  if ((int)blockIdx.x >= a.n_main) {   // a spare workgroup: the weight prefetch riding along (uniform per workgroup)
    vt_prefetch_role(a.pf, (int)blockIdx.x - a.n_main, (int)gridDim.x - a.n_main);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col4 = 512 * C8 + lane * 4;
  float gam[E], bet[E];
#pragma unroll
  for (int c = 0; c < C8; ++c) {
    const int col = (lane + 64 * c) * 8;
    const f32x4 g0 = *(const f32x4*)(a.gamma + col), g1 = *(const f32x4*)(a.gamma + col + 4);
    const f32x4 b0 = *(const f32x4*)(a.beta + col), b1 = *(const f32x4*)(a.beta + col + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { gam[8 * c + i] = g0[i]; gam[8 * c + 4 + i] = g1[i]; bet[8 * c + i] = b0[i]; bet[8 * c + 4 + i] = b1[i]; }
  }
  if (C4) {
    const f32x4 g0 = *(const f32x4*)(a.gamma + col4), b0 = *(const f32x4*)(a.beta + col4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { gam[8 * C8 + i] = g0[i]; bet[8 * C8 + i] = b0[i]; }
  }
  const float invH = 1.0f / (float)H;
  const long ngroups = ((long)a.M + R - 1) / R, stride = (long)a.n_main * 4;
  uint32_t xw[R][W];
  auto load_rows = [&](long g) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      long row = g * R + r;
      row = row < a.M ? row : (long)a.M - 1;   // the tail repeats the last row (its stores are skipped)
      const bf16_t* px = a.x + row * a.ldx;
#pragma unroll
      for (int c = 0; c < C8; ++c) {
        const u32x4 vx = *(const u32x4*)(px + (lane + 64 * c) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) xw[r][4 * c + i] = vx[i];
      }
      if (C4) {
        const u32x2 vx = *(const u32x2*)(px + col4);
        xw[r][4 * C8] = vx[0]; xw[r][4 * C8 + 1] = vx[1];
      }
    }
  };
  long g = (long)blockIdx.x * 4 + wave;
  if (g < ngroups) load_rows(g);
  for (; g < ngroups; g += stride) {
    float xv[R][E];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < W; ++k) {
        xv[r][2 * k] = XF16 ? f16lo(xw[r][k]) : bf16lo(xw[r][k]);
        xv[r][2 * k + 1] = XF16 ? f16hi(xw[r][k]) : bf16hi(xw[r][k]);
      }
    if (g + stride < ngroups) load_rows(g + stride);
    float u[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s += xv[r][e];
      u[r] = s;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = wave_sum_valu(u[r]) * invH;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) { const float d = xv[r][e] - u[r]; ss += d * d; }
      rs[r] = ss;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rs[r] = 1.0f / sqrtf(wave_sum_valu(rs[r]) * invH + a.eps);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long row = g * R + r;
      if (row < a.M) {
        if (lane == 0) {
          if (a.mean) a.mean[row] = u[r];
          if (a.rstd) a.rstd[row] = rs[r];
        }
        float o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = (xv[r][e] - u[r]) * rs[r] * gam[e] + bet[e];
        bf16_t* yp = a.y + row * a.ldy;
        uint16_t* hp = a.yh ? a.yh + row * a.ldyh : nullptr;
#pragma unroll
        for (int c = 0; c < C8; ++c) {
          const int col = (lane + 64 * c) * 8;
          u32x4 o4, h4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o4[i] = pack_bf16x2(o[8 * c + 2 * i], o[8 * c + 2 * i + 1]);
            h4[i] = pack_f16x2(o[8 * c + 2 * i], o[8 * c + 2 * i + 1]);
          }
          *(u32x4*)(yp + col) = o4;
          if (hp) *(u32x4*)(hp + col) = h4;
        }
        if (C4) {
          *(u32x2*)(yp + col4) = (u32x2){pack_bf16x2(o[8 * C8], o[8 * C8 + 1]), pack_bf16x2(o[8 * C8 + 2], o[8 * C8 + 3])};
          if (hp) *(u32x2*)(hp + col4) = (u32x2){pack_f16x2(o[8 * C8], o[8 * C8 + 1]), pack_f16x2(o[8 * C8 + 2], o[8 * C8 + 3])};
        }
      }
    }
  }
}

int vt_layernorm_dispatch(const void* x, long ldx, void* y, long ldy, const float* gamma, const float* beta,
                          float* mean, float* rstd, int M, int H, float eps, int grp_rows, int grp_stride,
                          hipStream_t stream, int x_f16 = 0, void* y_f16 = nullptr, long ldyh = 0,
                          const PrefetchArgs* pf = nullptr) {
  if (!x || !y || !gamma || !beta) return VT_ERR_NULL;
  if (M <= 0 || H <= 0 || (H % 8) || H > 64 * 8 * 4) return VT_ERR_BAD_SHAPE;
  if ((ldx % 8) || (ldy % 8) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15)) return VT_ERR_BAD_ALIGN;
  if (y_f16 && ((ldyh % 8) || ((uintptr_t)y_f16 & 15))) return VT_ERR_BAD_ALIGN;
  LnArgs a;
  a.x = (const bf16_t*)x; a.ldx = ldx; a.y = (bf16_t*)y; a.ldy = ldy; a.gamma = gamma; a.beta = beta;
  a.yh = (uint16_t*)y_f16; a.ldyh = ldyh;
  a.mean = mean; a.rstd = rstd; a.M = M; a.H = H; a.grp_rows = grp_rows; a.grp_stride = grp_stride; a.eps = eps;
  // VT_LN_FWD_ROWS: rows a wave holds at once (1 or 2; 0 = the chunked kernel above).  Measured at M = 50 820, H = 768, two
  // outputs: cold operands (tools/ln_bench.py) chunked 49.0 us, 1 row 44.2, 2 rows 44.7 (1 024 / 2 048 / 4 096 workgroups
  // within 2 us of each other); inside the pretrain step on one box (tools/experiments/ln_fwd_ab.sh) 41.8-42.1 / 39.4 / 38.1.
  static const int rows_per_wave = [] { const char* e = getenv("VT_LN_FWD_ROWS"); return e ? atoi(e) : 2; }();
  if (rows_per_wave > 0 && grp_rows == 0 && (H == 768 || H == 512 || H == 1024 || H == 256)) {
    const int R = rows_per_wave >= 2 ? 2 : 1;
    long nb = (((long)M + R - 1) / R + 3) / 4;
    static const int max_blocks = [] { const char* e = getenv("VT_LN_FWD_BLOCKS"); return e ? atoi(e) : 1024; }();
    if (nb > max_blocks) nb = max_blocks;
    a.n_main = (int)nb;
    a.pf.n = 0;
    if (pf && pf->n > 0) {   // spare workgroups behind the row workgroups read the ranges (16 KiB per workgroup and pass)
      long extra = 0;
      for (int i = 0; i < pf->n; ++i) extra += (pf->bytes[i] + 16383) >> 14;
      a.pf = *pf;
      nb += extra < 1 ? 1 : (extra > 640 ? 640 : extra);
    }
#define VT_LNF_LAUNCH(C8, C4, RR)                                                                                          \
    do {                                                                                                                   \
      if (x_f16) hipLaunchKernelGGL((layernorm_rows_full<C8, C4, RR, true>), dim3((unsigned)nb), dim3(256), 0, stream, a);  \
      else hipLaunchKernelGGL((layernorm_rows_full<C8, C4, RR, false>), dim3((unsigned)nb), dim3(256), 0, stream, a);      \
    } while (0)
#define VT_LNF_SHAPE(RR)                                                                                                   \
    do {                                                                                                                   \
      if (H == 768) VT_LNF_LAUNCH(1, 1, RR); else if (H == 512) VT_LNF_LAUNCH(1, 0, RR);                                   \
      else if (H == 1024) VT_LNF_LAUNCH(2, 0, RR); else VT_LNF_LAUNCH(0, 1, RR);                                           \
    } while (0)
    if (R == 2) VT_LNF_SHAPE(2); else VT_LNF_SHAPE(1);
#undef VT_LNF_SHAPE
#undef VT_LNF_LAUNCH
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
  }
  if (pf && pf->n > 0) hipLaunchKernelGGL(prefetch_ranges, dim3(256), dim3(256), 0, stream, *pf);   // (no spare workgroups in the chunked kernel)
  const dim3 grid((M + 15) / 16), block(256);   // 4 waves x 4 rows per workgroup
  const int ch = (H + 511) / 512;
#define VT_LN_LAUNCH(CH_)                                                                     \
  if (x_f16) hipLaunchKernelGGL((layernorm_rows<CH_, true>), grid, block, 0, stream, a);      \
  else hipLaunchKernelGGL((layernorm_rows<CH_, false>), grid, block, 0, stream, a)
  if (ch == 1) { VT_LN_LAUNCH(1); }
  else if (ch == 2) { VT_LN_LAUNCH(2); }
  else if (ch == 3) { VT_LN_LAUNCH(3); }
  else { VT_LN_LAUNCH(4); }
#undef VT_LN_LAUNCH
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
struct EmbArgs {
  const int64_t* ids; const int64_t* type_ids; const int64_t* pos_ids;  // [B,T]; type/pos may be null
  const float* word; const float* pos; const float* type;               // fp32 tables [*, H]
  const float* gamma; const float* beta;
  bf16_t* y; long ldy;                                                  // row b*S + t
  int B, T, S, H;
  int n_word, n_pos, n_type;
  float eps;
  int* err;  // set to 1 when an index is out of range (the torch reference would raise)
  DropCfg drop;  // BertEmbeddings.dropout after the LayerNorm; element index = token * H + col
};

template <int CH>
__global__ __launch_bounds__(256) void embed_layernorm(EmbArgs a) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= a.B * a.T) return;
  const int b = tok / a.T, t = tok - b * a.T;
  long wi = a.ids[tok];
  long pi = a.pos_ids ? a.pos_ids[tok] : (long)t;
  long ti = a.type_ids ? a.type_ids[tok] : 0L;
  if (wi < 0 || wi >= a.n_word || pi < 0 || pi >= a.n_pos || ti < 0 || ti >= a.n_type) {
    if (lane == 0 && a.err) *a.err = 1;
    wi = wi < 0 ? 0 : (wi >= a.n_word ? a.n_word - 1 : wi);
    pi = pi < 0 ? 0 : (pi >= a.n_pos ? a.n_pos - 1 : pi);
    ti = ti < 0 ? 0 : (ti >= a.n_type ? a.n_type - 1 : ti);
  }
  const float* wp = a.word + wi * a.H;
  const float* pp = a.pos + pi * a.H;
  const float* tp = a.type + ti * a.H;
  float v[CH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (lane + 64 * c) * 8;
    if (col < a.H) {
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const f32x4 w4 = *(const f32x4*)(wp + col + 4 * hlf);
        const f32x4 p4 = *(const f32x4*)(pp + col + 4 * hlf);
        const f32x4 t4 = *(const f32x4*)(tp + col + 4 * hlf);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[c][4 * hlf + i] = (w4[i] + p4[i]) + t4[i]; s += v[c][4 * hlf + i]; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
    }
  }
  const float invH = 1.0f / (float)a.H;
  const float u = wave_sum(s) * invH;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (lane + 64 * c) * 8;
    if (col < a.H) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[c][i] - u; ss += d * d; }
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(ss) * invH + a.eps);
  bf16_t* yp = a.y + ((long)b * a.S + t) * a.ldy;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (lane + 64 * c) * 8;
    if (col < a.H) {
      const f32x4 g0 = *(const f32x4*)(a.gamma + col), g1 = *(const f32x4*)(a.gamma + col + 4);
      const f32x4 b0 = *(const f32x4*)(a.beta + col), b1 = *(const f32x4*)(a.beta + col + 4);
      float o[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = (v[c][i] - u) * rs * g0[i] + b0[i];
        o[4 + i] = (v[c][4 + i] - u) * rs * g1[i] + b1[i];
      }
      if (a.drop.thresh) {
        const uint32_t e0 = (uint32_t)tok * (uint32_t)a.H + (uint32_t)col;
        vt_drop_run<8>(a.drop, e0, o);
      }
      u32x4 w;
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
      *(u32x4*)(yp + col) = w;
    }
  }
}

int vt_embed_layernorm_dispatch(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                                const float* pos, const float* type, const float* gamma, const float* beta, void* y,
                                long ldy, int B, int T, int S, int H, int n_word, int n_pos, int n_type, float eps,
                                int* err_flag, hipStream_t stream, const DropCfg* drop = nullptr) {
  if (!ids || !word || !pos || !type || !gamma || !beta || !y) return VT_ERR_NULL;
  if (B <= 0 || T <= 0 || S < T || H <= 0 || (H % 8) || H > 2048) return VT_ERR_BAD_SHAPE;
  if ((ldy % 8) || (((uintptr_t)word | (uintptr_t)pos | (uintptr_t)type | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15))
    return VT_ERR_BAD_ALIGN;
  EmbArgs a;
  a.ids = ids; a.type_ids = type_ids; a.pos_ids = pos_ids; a.word = word; a.pos = pos; a.type = type;
  a.gamma = gamma; a.beta = beta; a.y = (bf16_t*)y; a.ldy = ldy; a.B = B; a.T = T; a.S = S; a.H = H;
  a.n_word = n_word; a.n_pos = n_pos; a.n_type = n_type; a.eps = eps; a.err = err_flag;
  if (drop) a.drop = *drop; else { a.drop.thresh = 0; a.drop.seed = 0; a.drop.scale = 1.0f; }
  const dim3 grid((B * T + 3) / 4), block(256);
  const int ch = (H + 511) / 512;
  if (ch == 1) hipLaunchKernelGGL(embed_layernorm<1>, grid, block, 0, stream, a);
  else if (ch == 2) hipLaunchKernelGGL(embed_layernorm<2>, grid, block, 0, stream, a);
  else if (ch == 3) hipLaunchKernelGGL(embed_layernorm<3>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(embed_layernorm<4>, grid, block, 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// out[row, :] = bf16([src0[row, 0:d0] | src1[row, 0:d1] | zeros up to kpad]); 8 columns per thread.  A group that lies
// inside one source reads it as four float2 (d0 = 2054: the rows of the region features are 8-byte, not 16-byte, aligned);
// only the group that straddles the seam (and any source with an odd width) takes the element-wise path.
template <bool PAIRS>
__global__ __launch_bounds__(256) void pack_concat_bf16(const float* __restrict__ s0, int d0, const float* __restrict__ s1,
                                                        int d1, bf16_t* __restrict__ out, int kpad, long rows) {
  const int cpr = kpad >> 3;  // 8-column groups per row
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows * cpr) return;
  const long row = gid / cpr;
  const int col = (int)(gid - row * cpr) * 8;
  float v[8];
  const float* src = nullptr;
  if (PAIRS) {
    if (col + 8 <= d0) src = s0 + row * d0 + col;
    else if (col >= d0 && col + 8 <= d0 + d1) src = s1 + row * d1 + (col - d0);
  }
  if (src) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 t = *(const float2*)(src + 2 * i);
      v[2 * i] = t.x;
      v[2 * i + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = col + i;
      float x = 0.f;
      if (c < d0) x = s0[row * d0 + c];
      else if (c < d0 + d1) x = s1[row * d1 + (c - d0)];
      v[i] = x;
    }
  }
  u32x4 w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
  *(u32x4*)(out + row * kpad + col) = w;
}

int vt_pack_concat_dispatch(const float* s0, int d0, const float* s1, int d1, void* out, int kpad, long rows,
                            hipStream_t stream) {
  if (!s0 || !out || (d1 > 0 && !s1)) return VT_ERR_NULL;
  if (rows <= 0 || d0 <= 0 || d1 < 0 || kpad < d0 + d1 || (kpad % 8)) return VT_ERR_BAD_SHAPE;
  if (((uintptr_t)out) & 15) return VT_ERR_BAD_ALIGN;
  const long n = rows * (kpad >> 3);
  // float2 reads need even widths (every row and the seam then start on 8 bytes) and 8-byte aligned bases
  const bool pairs = !(d0 & 1) && !(d1 & 1) && !(((uintptr_t)s0 | (uintptr_t)s1) & 7);
  if (pairs) hipLaunchKernelGGL(pack_concat_bf16<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, s0, d0, s1, d1,
                                (bf16_t*)out, kpad, rows);
  else hipLaunchKernelGGL(pack_concat_bf16<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, s0, d0, s1, d1,
                          (bf16_t*)out, kpad, rows);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward (autograd of BertLayerNorm inside loss.backward(), pretrain.py:191):
//   xhat = (x - u) * rstd;  g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
//   dgamma = sum_rows dy * xhat;  dbeta = sum_rows dy
// One wave per row (statistics recomputed from x: no saved mean/rstd needed), rows strided over a
// fixed grid; each wave keeps its dgamma/dbeta partial sums in registers, the block combines them
// through LDS and writes ONE partial row per block (no atomics: reproducible); ln_bwd_reduce sums
// the partial rows.
struct LnBwdArgs {
  const bf16_t* x; long ldx;
  const bf16_t* dy; long ldy;
  const float* gamma;
  bf16_t* dx; long lddx;
  bf16_t* dx2; long lddx2;   // optional: dx * dropout mask * scale (gradient of the dense output that was dropped
  DropCfg drop;              // out before the residual add); element index = row * H + col
  float* partial;  // [gridDim.x][2][H]
  int M, H;
  float eps;
};

template <int CH, bool XF16>
__global__ __launch_bounds__(256) void layernorm_bwd_rows(LnBwdArgs a) {
  __shared__ float red[4][2][CH * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gam[CH][8], dg[CH][8], db[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (lane + 64 * c) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dg[c][i] = 0.f; db[c][i] = 0.f;
      gam[c][i] = col < a.H ? a.gamma[col + i] : 0.f;
    }
  }
  const float invH = 1.0f / (float)a.H;
  for (long row = (long)blockIdx.x * 4 + wave; row < a.M; row += (long)gridDim.x * 4) {
    float xv[CH][8], gv[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
        const u32x4 w = *(const u32x4*)(a.x + row * a.ldx + col);
        const u32x4 d = *(const u32x4*)(a.dy + row * a.ldy + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xv[c][2 * i] = XF16 ? f16lo(w[i]) : bf16lo(w[i]); xv[c][2 * i + 1] = XF16 ? f16hi(w[i]) : bf16hi(w[i]);
          gv[c][2 * i] = bf16lo(d[i]); gv[c][2 * i + 1] = bf16hi(d[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += xv[c][i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xv[c][i] = 0.f; gv[c][i] = 0.f; }
      }
    }
    const float u = wave_sum(s) * invH;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = xv[c][i] - u; ss += d * d; }
      }
    }
    const float rs = 1.0f / sqrtf(wave_sum(ss) * invH + a.eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[c][i] - u) * rs;
          const float dyv = gv[c][i];
          dg[c][i] += dyv * xh;
          db[c][i] += dyv;
          const float gg = dyv * gam[c][i];
          xv[c][i] = xh;
          gv[c][i] = gg;
          s1 += gg;
          s2 += gg * xh;
        }
      }
    }
    const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rs * (gv[c][i] - m1 - xv[c][i] * m2);
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
        *(u32x4*)(a.dx + row * a.lddx + col) = w;
        if (a.dx2) {
          const uint32_t e0 = (uint32_t)row * (uint32_t)a.H + (uint32_t)col;
          vt_drop_run<8>(a.drop, e0, o);   // thresh 0: everything kept, scale 1
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
          *(u32x4*)(a.dx2 + row * a.lddx2 + col) = w;
        }
      }
    }
  }
  // block combine: 4 waves -> one partial row
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[wave][0][(lane + 64 * c) * 8 + i] = dg[c][i];
      red[wave][1][(lane + 64 * c) * 8 + i] = db[c][i];
    }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * a.H; idx += 256) {
    const int which = idx / a.H, col = idx - which * a.H;
    const float v = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
    a.partial[((long)blockIdx.x * 2 + which) * a.H + col] = v;
  }
}

// The same backward for H == 512 * C8 + 256 * C4 exactly (768 = 512 + 256, 512, 1024, 256), built for the memory system
// instead of around it:
//  * every lane owns E = 8 * C8 + 4 * C4 columns of a row -- chunk c < C8: columns (lane + 64 c) * 8 .. + 8 (16-byte
//    accesses), then 4 columns at 512 * C8 + lane * 4 (8-byte accesses): at H = 768 all 64 lanes carry 12 columns (the
//    chunked form leaves half the wave idle in its second chunk and pays the full vector-issue time for it);
//  * a wave works on R consecutive rows at once and loads the NEXT R rows (raw words) before it starts on the current ones:
//    with one row per wave and four dependent reductions between its loads and the next row's, the kernel kept ~ 30 KB per CU
//    in flight where 8 TB/s needs ~ 64 KB;
//  * the four reductions per row run on the vector ALU (wave_sum_valu) instead of 24 LDS-crossbar shuffles.
// Same arithmetic per element as layernorm_bwd_rows; the in-lane summation order differs (last-bit differences in dx).
template <int C8, int C4, int R, bool XF16>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_full(LnBwdArgs a) {
  constexpr int E = 8 * C8 + 4 * C4, H = 512 * C8 + 256 * C4, W = E / 2;
  __shared__ float red[4][2][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col4 = 512 * C8 + lane * 4;
  float gam[E], dg[E], db[E];
#pragma unroll
  for (int c = 0; c < C8; ++c) {
    const int col = (lane + 64 * c) * 8;
    const f32x4 g0 = *(const f32x4*)(a.gamma + col), g1 = *(const f32x4*)(a.gamma + col + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { gam[8 * c + i] = g0[i]; gam[8 * c + 4 + i] = g1[i]; }
  }
  if (C4) {
    const f32x4 g0 = *(const f32x4*)(a.gamma + col4);
#pragma unroll
    for (int i = 0; i < 4; ++i) gam[8 * C8 + i] = g0[i];
  }
#pragma unroll
  for (int e = 0; e < E; ++e) { dg[e] = 0.f; db[e] = 0.f; }
  const float invH = 1.0f / (float)H;
  const long ngroups = ((long)a.M + R - 1) / R, stride = (long)gridDim.x * 4;
  uint32_t xw[R][W], dw[R][W];
  auto load_rows = [&](long g) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      long row = g * R + r;
      row = row < a.M ? row : (long)a.M - 1;   // the tail repeats the last row: no contribution, no store (below)
      const bf16_t* px = a.x + row * a.ldx;
      const bf16_t* pd = a.dy + row * a.ldy;
#pragma unroll
      for (int c = 0; c < C8; ++c) {
        const int col = (lane + 64 * c) * 8;
        const u32x4 vx = *(const u32x4*)(px + col), vd = *(const u32x4*)(pd + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xw[r][4 * c + i] = vx[i]; dw[r][4 * c + i] = vd[i]; }
      }
      if (C4) {
        const u32x2 vx = *(const u32x2*)(px + col4), vd = *(const u32x2*)(pd + col4);
        xw[r][4 * C8] = vx[0]; xw[r][4 * C8 + 1] = vx[1]; dw[r][4 * C8] = vd[0]; dw[r][4 * C8 + 1] = vd[1];
      }
    }
  };
  long g = (long)blockIdx.x * 4 + wave;
  if (g < ngroups) load_rows(g);
  for (; g < ngroups; g += stride) {
    float xv[R][E], gv[R][E];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool valid = g * R + r < a.M;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        xv[r][2 * k] = XF16 ? f16lo(xw[r][k]) : bf16lo(xw[r][k]);
        xv[r][2 * k + 1] = XF16 ? f16hi(xw[r][k]) : bf16hi(xw[r][k]);
        gv[r][2 * k] = valid ? bf16lo(dw[r][k]) : 0.f;
        gv[r][2 * k + 1] = valid ? bf16hi(dw[r][k]) : 0.f;
      }
    }
    if (g + stride < ngroups) load_rows(g + stride);
    float u[R], rs[R], m1[R], m2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s += xv[r][e];
      u[r] = s;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = wave_sum_valu(u[r]) * invH;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) { const float d = xv[r][e] - u[r]; ss += d * d; }
      rs[r] = ss;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rs[r] = 1.0f / sqrtf(wave_sum_valu(rs[r]) * invH + a.eps);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float xh = (xv[r][e] - u[r]) * rs[r];
        const float dyv = gv[r][e];
        dg[e] += dyv * xh;
        db[e] += dyv;
        const float gg = dyv * gam[e];
        xv[r][e] = xh;
        gv[r][e] = gg;
        s1 += gg;
        s2 += gg * xh;
      }
      m1[r] = s1; m2[r] = s2;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { m1[r] = wave_sum_valu(m1[r]) * invH; m2[r] = wave_sum_valu(m2[r]) * invH; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long row = g * R + r;
      if (row < a.M) {
        bf16_t* po = a.dx + row * a.lddx;
        bf16_t* po2 = a.dx2 ? a.dx2 + row * a.lddx2 : nullptr;
        const uint32_t erow = (uint32_t)row * (uint32_t)H;
#pragma unroll
        for (int c = 0; c < C8; ++c) {
          const int col = (lane + 64 * c) * 8;
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = rs[r] * (gv[r][8 * c + i] - m1[r] - xv[r][8 * c + i] * m2[r]);
          u32x4 w;
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
          *(u32x4*)(po + col) = w;
          if (po2) {
            vt_drop_run<8>(a.drop, erow + (uint32_t)col, o);   // thresh 0: everything kept, scale 1
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
            *(u32x4*)(po2 + col) = w;
          }
        }
        if (C4) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = rs[r] * (gv[r][8 * C8 + i] - m1[r] - xv[r][8 * C8 + i] * m2[r]);
          *(u32x2*)(po + col4) = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          if (po2) {
            vt_drop_run<4>(a.drop, erow + (uint32_t)col4, o);
            *(u32x2*)(po2 + col4) = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          }
        }
      }
    }
  }
  // block combine: 4 waves -> one partial row
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int col = e < 8 * C8 ? (lane + 64 * (e >> 3)) * 8 + (e & 7) : col4 + (e - 8 * C8);
    red[wave][0][col] = dg[e];
    red[wave][1][col] = db[e];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * H; idx += 256) {
    const int which = idx >= H ? 1 : 0, col = idx - which * H;
    const float v = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
    a.partial[((long)blockIdx.x * 2 + which) * H + col] = v;
  }
}

// out[j] (+)= sum_b partial[b][j], j < n  (n = 2H: dgamma | dbeta; n % 4 == 0).  A block owns 16 columns (4 lanes x
// 16 bytes); its 64 row groups each sum every 64th partial row with independent 16-byte loads (1024 partial rows = 16
// loads per thread, all in flight), then combine through LDS.  (The first form -- 32 columns x 8 row groups, 128
// dependent-latency loads per thread on 48 workgroups -- took 12 us per call, as much as the LayerNorm backward itself
// at a small batch.)
__global__ __launch_bounds__(256) void ln_bwd_reduce(const float* __restrict__ partial, int nblocks, int n,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int H,
                                                     int accumulate, int n_main, PrefetchArgs pf) {
  __shared__ f32x4 red[64][4];
  if ((int)blockIdx.x >= n_main) {   // a spare workgroup: the weight prefetch riding along (see LnArgs::pf)
    vt_prefetch_role(pf, (int)blockIdx.x - n_main, (int)gridDim.x - n_main);
    return;
  }
  const int cg = threadIdx.x & 3, part = threadIdx.x >> 2;
  const int j = blockIdx.x * 16 + cg * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (j < n) {
    int b = part;
    for (; b + 192 < nblocks; b += 256) {
      const f32x4 v0 = *(const f32x4*)(partial + (long)b * n + j), v1 = *(const f32x4*)(partial + (long)(b + 64) * n + j);
      const f32x4 v2 = *(const f32x4*)(partial + (long)(b + 128) * n + j), v3 = *(const f32x4*)(partial + (long)(b + 192) * n + j);
      s += (v0 + v1) + (v2 + v3);
    }
    for (; b < nblocks; b += 64) s += *(const f32x4*)(partial + (long)b * n + j);
  }
  red[part][cg] = s;
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {   // tree over the 64 row groups: a fixed order, reproducible
    if (part < h) red[part][cg] += red[part + h][cg];
    __syncthreads();
  }
  if (part == 0 && j < n) {
    const f32x4 t = red[0][cg];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int jj = j + k;
      float* dst = jj < H ? dgamma + jj : dbeta + (jj - H);
      *dst = accumulate ? *dst + t[k] : t[k];
    }
  }
}

#define LN_BWD_MAX_BLOCKS 1024
int vt_layernorm_bwd_dispatch(const void* x, long ldx, const void* dy, long ldy, const float* gamma, void* dx, long lddx,
                              float* dgamma, float* dbeta, float* partial_ws, int M, int H, float eps, int accumulate,
                              hipStream_t stream, void* dx2 = nullptr, long lddx2 = 0, const DropCfg* drop = nullptr,
                              int x_f16 = 0, const PrefetchArgs* pf = nullptr) {
  if (!x || !dy || !gamma || !dx || !dgamma || !dbeta || !partial_ws) return VT_ERR_NULL;
  if (M <= 0 || H <= 0 || (H % 8) || H > 1024) return VT_ERR_BAD_SHAPE;
  if ((ldx % 8) || (ldy % 8) || (lddx % 8) || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15)) return VT_ERR_BAD_ALIGN;
  LnBwdArgs a;
  a.x = (const bf16_t*)x; a.ldx = ldx; a.dy = (const bf16_t*)dy; a.ldy = ldy; a.gamma = gamma;
  a.dx = (bf16_t*)dx; a.lddx = lddx; a.partial = partial_ws; a.M = M; a.H = H; a.eps = eps;
  a.dx2 = (bf16_t*)dx2; a.lddx2 = lddx2;
  if (drop) a.drop = *drop; else { a.drop.thresh = 0; a.drop.seed = 0; a.drop.scale = 1.0f; }
  if (dx2 && ((lddx2 % 8) || ((uintptr_t)dx2 & 15))) return VT_ERR_BAD_ALIGN;
  int nblocks = (M + 3) / 4;
  if (nblocks > LN_BWD_MAX_BLOCKS) nblocks = LN_BWD_MAX_BLOCKS;
  // VT_LN_BWD_ROWS: rows a wave holds at once (1, 2 or 4; 0 = the chunked kernel).  Measured at M = 50 820, H = 768 with the
  // masked second copy, cold operands, reduce kernel included (tools/ln_bench.py): chunked 74.9 us, 1 row 66.6 (122 registers,
  // four waves per SIMD), 2 rows 70.3 (182), 4 rows 78.3 (302); in the pretrain step 65.3 -> 51.5 us per launch.
  static const int rows_per_wave = [] { const char* e = getenv("VT_LN_BWD_ROWS"); return e ? atoi(e) : 1; }();
  if (rows_per_wave > 0 && (H == 768 || H == 512 || H == 1024 || H == 256) && (((uintptr_t)gamma) & 15) == 0) {
    const int R = rows_per_wave >= 4 ? 4 : rows_per_wave >= 2 ? 2 : 1;
    nblocks = (int)((((long)M + R - 1) / R + 3) / 4);
    if (nblocks > LN_BWD_MAX_BLOCKS) nblocks = LN_BWD_MAX_BLOCKS;
#define VT_LNB_LAUNCH(C8, C4, RR)                                                                                        \
    do {                                                                                                                 \
      if (x_f16) hipLaunchKernelGGL((layernorm_bwd_rows_full<C8, C4, RR, true>), dim3(nblocks), dim3(256), 0, stream, a);  \
      else hipLaunchKernelGGL((layernorm_bwd_rows_full<C8, C4, RR, false>), dim3(nblocks), dim3(256), 0, stream, a);      \
    } while (0)
#define VT_LNB_SHAPE(RR)                                                                                                 \
    do {                                                                                                                 \
      if (H == 768) VT_LNB_LAUNCH(1, 1, RR); else if (H == 512) VT_LNB_LAUNCH(1, 0, RR);                                 \
      else if (H == 1024) VT_LNB_LAUNCH(2, 0, RR); else VT_LNB_LAUNCH(0, 1, RR);                                         \
    } while (0)
    if (R == 4) VT_LNB_SHAPE(4); else if (R == 2) VT_LNB_SHAPE(2); else VT_LNB_SHAPE(1);
#undef VT_LNB_SHAPE
#undef VT_LNB_LAUNCH
  } else if (H <= 512) {
    if (x_f16) hipLaunchKernelGGL((layernorm_bwd_rows<1, true>), dim3(nblocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((layernorm_bwd_rows<1, false>), dim3(nblocks), dim3(256), 0, stream, a);
  } else {
    if (x_f16) hipLaunchKernelGGL((layernorm_bwd_rows<2, true>), dim3(nblocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((layernorm_bwd_rows<2, false>), dim3(nblocks), dim3(256), 0, stream, a);
  }
  {
    const int n_main = (2 * H + 15) / 16;
    PrefetchArgs none;
    none.n = 0;
    for (int i = 0; i < 4; ++i) { none.p[i] = nullptr; none.bytes[i] = 0; }
    long extra = 0;
    if (pf && pf->n > 0) {
      for (int i = 0; i < pf->n; ++i) extra += (pf->bytes[i] + 16383) >> 14;
      extra = extra < 1 ? 1 : (extra > 640 ? 640 : extra);
    }
    hipLaunchKernelGGL(ln_bwd_reduce, dim3((unsigned)(n_main + extra)), dim3(256), 0, stream, partial_ws, nblocks, 2 * H, dgamma,
                       dbeta, H, accumulate, n_main, extra ? *pf : none);
  }
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
// workspace floats needed by vt_layernorm_bwd: LN_BWD_MAX_BLOCKS * 2 * H

// out = g * d elementwise (bf16, 8 per thread), d = saved gelu'(pre-activation): backward through the
// MLM-head transform's GELU.
__global__ __launch_bounds__(256) void dgelu_mul_bf16(const bf16_t* __restrict__ g, const bf16_t* __restrict__ h,
                                                      bf16_t* __restrict__ out, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const u32x4 gv = ((const u32x4*)g)[i], hv = ((const u32x4*)h)[i];
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    o[k] = pack_bf16x2(bf16lo(gv[k]) * bf16lo(hv[k]), bf16hi(gv[k]) * bf16hi(hv[k]));
  ((u32x4*)out)[i] = o;
}

int vt_dgelu_mul_dispatch(const void* g, const void* h, void* out, long n, hipStream_t stream) {
  if (!g || !h || !out) return VT_ERR_NULL;
  if (n <= 0 || (n % 8)) return VT_ERR_BAD_SHAPE;
  if (((uintptr_t)g | (uintptr_t)h | (uintptr_t)out) & 15) return VT_ERR_BAD_ALIGN;
  const long n8 = n / 8;
  hipLaunchKernelGGL(dgelu_mul_bf16, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)g,
                     (const bf16_t*)h, (bf16_t*)out, n8);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Backward of embed_layernorm: recompute e = word + pos + type (fp32, as the forward did), run the
// LayerNorm backward against the incoming gradient rows b*S+t of g, and emit d_e (fp32 [B*T,H]) for
// the three table scatter-adds, plus dgamma/dbeta partial rows (same scheme as layernorm_bwd_rows).
struct EmbBwdArgs {
  const int64_t* ids; const int64_t* type_ids; const int64_t* pos_ids;
  const float* word; const float* pos; const float* type;
  const float* gamma;
  const bf16_t* g; long ldg;   // gradient rows b*S + t
  float* de;                   // [B*T, H]
  float* partial;              // [gridDim.x][2][H]
  int B, T, S, H;
  int n_word, n_pos, n_type;
  float eps;
  DropCfg drop;  // same mask as the forward: the incoming gradient is masked and scaled first
};

template <int CH>
__global__ __launch_bounds__(256) void embed_layernorm_bwd(EmbBwdArgs a) {
  __shared__ float red[4][2][CH * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gam[CH][8], dg[CH][8], db[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (lane + 64 * c) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dg[c][i] = 0.f; db[c][i] = 0.f;
      gam[c][i] = col < a.H ? a.gamma[col + i] : 0.f;
    }
  }
  const float invH = 1.0f / (float)a.H;
  const long ntok = (long)a.B * a.T;
  for (long tok = (long)blockIdx.x * 4 + wave; tok < ntok; tok += (long)gridDim.x * 4) {
    const int b = (int)(tok / a.T), t = (int)(tok - (long)b * a.T);
    long wi = a.ids[tok];
    long pi = a.pos_ids ? a.pos_ids[tok] : (long)t;
    long ti = a.type_ids ? a.type_ids[tok] : 0L;
    wi = wi < 0 ? 0 : (wi >= a.n_word ? a.n_word - 1 : wi);
    pi = pi < 0 ? 0 : (pi >= a.n_pos ? a.n_pos - 1 : pi);
    ti = ti < 0 ? 0 : (ti >= a.n_type ? a.n_type - 1 : ti);
    const float* wp = a.word + wi * a.H;
    const float* pp = a.pos + pi * a.H;
    const float* tp = a.type + ti * a.H;
    const bf16_t* gp = a.g + ((long)b * a.S + t) * a.ldg;
    float xv[CH][8], gv[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
          const f32x4 w4 = *(const f32x4*)(wp + col + 4 * hlf);
          const f32x4 p4 = *(const f32x4*)(pp + col + 4 * hlf);
          const f32x4 t4 = *(const f32x4*)(tp + col + 4 * hlf);
#pragma unroll
          for (int i = 0; i < 4; ++i) { xv[c][4 * hlf + i] = (w4[i] + p4[i]) + t4[i]; s += xv[c][4 * hlf + i]; }
        }
        const u32x4 d = *(const u32x4*)(gp + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gv[c][2 * i] = bf16lo(d[i]); gv[c][2 * i + 1] = bf16hi(d[i]); }
        if (a.drop.thresh) {
          const uint32_t e0 = (uint32_t)tok * (uint32_t)a.H + (uint32_t)col;
          vt_drop_run<8>(a.drop, e0, gv[c]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xv[c][i] = 0.f; gv[c][i] = 0.f; }
      }
    }
    const float u = wave_sum(s) * invH;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = xv[c][i] - u; ss += d * d; }
      }
    }
    const float rs = 1.0f / sqrtf(wave_sum(ss) * invH + a.eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[c][i] - u) * rs;
          const float dyv = gv[c][i];
          dg[c][i] += dyv * xh;
          db[c][i] += dyv;
          const float gg = dyv * gam[c][i];
          xv[c][i] = xh;
          gv[c][i] = gg;
          s1 += gg;
          s2 += gg * xh;
        }
      }
    }
    const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = (lane + 64 * c) * 8;
      if (col < a.H) {
        float* op = a.de + tok * a.H + col;
        f32x4 o0, o1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o0[i] = rs * (gv[c][i] - m1 - xv[c][i] * m2);
          o1[i] = rs * (gv[c][4 + i] - m1 - xv[c][4 + i] * m2);
        }
        *(f32x4*)op = o0;
        *(f32x4*)(op + 4) = o1;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[wave][0][(lane + 64 * c) * 8 + i] = dg[c][i];
      red[wave][1][(lane + 64 * c) * 8 + i] = db[c][i];
    }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * a.H; idx += 256) {
    const int which = idx / a.H, col = idx - which * a.H;
    a.partial[((long)blockIdx.x * 2 + which) * a.H + col] =
        red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
  }
}

int vt_embed_layernorm_bwd_dispatch(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                                    const float* pos, const float* type, const float* gamma, const void* g, long ldg,
                                    float* de, float* dgamma, float* dbeta, float* partial_ws, int B, int T, int S, int H,
                                    int n_word, int n_pos, int n_type, float eps, int accumulate, hipStream_t stream,
                                    const DropCfg* drop = nullptr) {
  if (!ids || !word || !pos || !type || !gamma || !g || !de || !dgamma || !dbeta || !partial_ws) return VT_ERR_NULL;
  if (B <= 0 || T <= 0 || S < T || H <= 0 || (H % 8) || H > 1024) return VT_ERR_BAD_SHAPE;
  if ((ldg % 8) || (((uintptr_t)word | (uintptr_t)pos | (uintptr_t)type | (uintptr_t)g | (uintptr_t)de) & 15)) return VT_ERR_BAD_ALIGN;
  EmbBwdArgs a;
  a.ids = ids; a.type_ids = type_ids; a.pos_ids = pos_ids; a.word = word; a.pos = pos; a.type = type; a.gamma = gamma;
  a.g = (const bf16_t*)g; a.ldg = ldg; a.de = de; a.partial = partial_ws; a.B = B; a.T = T; a.S = S; a.H = H;
  a.n_word = n_word; a.n_pos = n_pos; a.n_type = n_type; a.eps = eps;
  if (drop) a.drop = *drop; else { a.drop.thresh = 0; a.drop.seed = 0; a.drop.scale = 1.0f; }
  long nb = ((long)B * T + 3) / 4;
  const int nblocks = (int)(nb > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : nb);
  if (H <= 512) hipLaunchKernelGGL(embed_layernorm_bwd<1>, dim3(nblocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(embed_layernorm_bwd<2>, dim3(nblocks), dim3(256), 0, stream, a);
  PrefetchArgs none;
  none.n = 0;
  for (int i = 0; i < 4; ++i) { none.p[i] = nullptr; none.bytes[i] = 0; }
  hipLaunchKernelGGL(ln_bwd_reduce, dim3((2 * H + 15) / 16), dim3(256), 0, stream, partial_ws, nblocks, 2 * H, dgamma,
                     dbeta, H, accumulate, (2 * H + 15) / 16, none);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Fused AdamW step over a flat fp32 parameter slab, the pytorch-transformers rule used by
// tasks/viewpoint_select/pretrain.py:128-130,192:
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= step_size * m / (sqrt(v) + eps);  p -= lr*wd*p
// with step_size = lr * sqrt(1-b2^t) / (1-b1^t) computed on the host (eps is added to the UN-corrected
// sqrt(v); the decoupled decay uses the already-moved p).  Also refreshes the bf16 working copy the
// GEMMs read.  grad_scale multiplies g first (1/world_size for the data-parallel mean).
// G16: the gradients arrive as bf16 (the data-parallel all-reduce ran on a bf16 copy of the slab: half the bytes over
// xGMI); moments and master weights stay fp32.
template <bool G16>
__global__ __launch_bounds__(256) void adamw_flat(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, bf16_t* __restrict__ p_bf16, long n4, float lr,
                                                  float step_size, float b1, float b2, float eps, float wd,
                                                  float grad_scale) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
    f32x4 gv;
    if (G16) {
      const u32x2 w = ((const u32x2*)g)[i];
      gv = (f32x4){bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
    } else {
      gv = ((const f32x4*)g)[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gv[k] * grad_scale;
      mv[k] = b1 * mv[k] + (1.0f - b1) * gg;
      vv[k] = b2 * vv[k] + (1.0f - b2) * gg * gg;
      float x = pv[k] - step_size * (mv[k] / (sqrtf(vv[k]) + eps));
      if (wd > 0.f) x = x - lr * wd * x;
      pv[k] = x;
    }
    ((f32x4*)p)[i] = pv;
    ((f32x4*)m)[i] = mv;
    ((f32x4*)v)[i] = vv;
    if (p_bf16) {
      u32x2 o;
      o[0] = pack_bf16x2(pv[0], pv[1]);
      o[1] = pack_bf16x2(pv[2], pv[3]);
      ((u32x2*)p_bf16)[i] = o;
    }
  }
}

int vt_adamw_dispatch(float* p, const void* g, int g_is_bf16, float* m, float* v, void* p_bf16, long n, float lr,
                      float step_size, float b1, float b2, float eps, float wd, float grad_scale, hipStream_t stream) {
  if (!p || !g || !m || !v) return VT_ERR_NULL;
  if (n <= 0 || (n % 4)) return VT_ERR_BAD_SHAPE;
  if (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15 || ((uintptr_t)g & (g_is_bf16 ? 7 : 15)) || ((uintptr_t)p_bf16 & 7)) return VT_ERR_BAD_ALIGN;
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (g_is_bf16)
    hipLaunchKernelGGL(adamw_flat<true>, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n4, lr,
                       step_size, b1, b2, eps, wd, grad_scale);
  else
    hipLaunchKernelGGL(adamw_flat<false>, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n4, lr,
                       step_size, b1, b2, eps, wd, grad_scale);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// out[r, 64 h + d] = x[r, 64 h + d] * scale[h]: head_mask applied to a context tensor (oscar/modeling_bert.py:65-66 scales
// the probabilities of head h, i.e. that head's 64 context columns) -- the training path's forward copy and its
// gradient's way back.  One thread = 8 columns.
__global__ __launch_bounds__(256) void scale_heads_bf16(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo,
                                                        long rows, int nh, const float* __restrict__ scale) {
  const int cpr = nh * 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cpr) return;
  const long row = i / cpr;
  const int c = (int)(i - row * cpr);
  const float sc = scale[c >> 3];
  const u32x4 v = *(const u32x4*)(x + row * ldx + c * 8);
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(bf16lo(v[k]) * sc, bf16hi(v[k]) * sc);
  *(u32x4*)(out + row * ldo + c * 8) = o;
}

int vt_scale_heads_dispatch(const void* x, long ldx, void* out, long ldo, long rows, int nh, const float* scale, hipStream_t stream) {
  if (!x || !out || !scale) return VT_ERR_NULL;
  if (rows <= 0 || nh <= 0) return VT_ERR_BAD_SHAPE;
  if ((ldx % 8) || (ldo % 8) || (((uintptr_t)x | (uintptr_t)out) & 15)) return VT_ERR_BAD_ALIGN;
  const long n = rows * nh * 8;
  hipLaunchKernelGGL(scale_heads_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx,
                     (bf16_t*)out, ldo, rows, nh, scale);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// y = bf16(x * scale), flat: the communication copy of a gradient-slab range (half the all-reduce bytes)
__global__ __launch_bounds__(256) void cast_scale_f32_bf16(const float* __restrict__ x, bf16_t* __restrict__ y, long n8, float scale) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const f32x4 a = ((const f32x4*)x)[2 * i], b = ((const f32x4*)x)[2 * i + 1];
    u32x4 o;
    o[0] = pack_bf16x2(a[0] * scale, a[1] * scale); o[1] = pack_bf16x2(a[2] * scale, a[3] * scale);
    o[2] = pack_bf16x2(b[0] * scale, b[1] * scale); o[3] = pack_bf16x2(b[2] * scale, b[3] * scale);
    ((u32x4*)y)[i] = o;
  }
}

int vt_cast_scale_dispatch(const float* x, void* y, long n, float scale, hipStream_t stream) {
  if (!x || !y) return VT_ERR_NULL;
  if (n <= 0 || (n % 8)) return VT_ERR_BAD_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return VT_ERR_BAD_ALIGN;
  const long n8 = n / 8;
  long blocks = (n8 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(cast_scale_f32_bf16, dim3((unsigned)blocks), dim3(256), 0, stream, x, (bf16_t*)y, n8, scale);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// out[c][r] = in[r][c], bf16, 64x64 tiles through LDS (both sides 16-byte coalesced).  Refreshes the
// transposed weight copies the dgrad GEMMs read after every optimizer step.
__global__ __launch_bounds__(256) void transpose_bf16(const bf16_t* __restrict__ in, long ldi, bf16_t* __restrict__ out,
                                                      long ldo, int R, int C) {
  __shared__ bf16_t tile[64][72];  // [c][r], pitch 72 elements = 144 B keeps 16-B row reads aligned
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int chunk = t + 256 * it;          // 512 chunks of 8 elements
    const int r = chunk >> 3, cc = (chunk & 7) * 8;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (r0 + r < R && c0 + cc < C) v = *(const u32x4*)(in + (long)(r0 + r) * ldi + c0 + cc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tile[cc + 2 * i][r] = (bf16_t)(v[i] & 0xffffu);
      tile[cc + 2 * i + 1][r] = (bf16_t)(v[i] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int chunk = t + 256 * it;
    const int c = chunk >> 3, rr = (chunk & 7) * 8;
    if (c0 + c < C && r0 + rr < R) *(u32x4*)(out + (long)(c0 + c) * ldo + r0 + rr) = *(const u32x4*)(&tile[c][rr]);
  }
}

int vt_transpose_dispatch(const void* in, long ldi, void* out, long ldo, int R, int C, hipStream_t stream) {
  if (!in || !out) return VT_ERR_NULL;
  if (R <= 0 || C <= 0 || (R % 8) || (C % 8)) return VT_ERR_BAD_SHAPE;
  if ((ldi % 8) || (ldo % 8) || (((uintptr_t)in | (uintptr_t)out) & 15)) return VT_ERR_BAD_ALIGN;
  hipLaunchKernelGGL(transpose_bf16, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, (const bf16_t*)in, ldi,
                     (bf16_t*)out, ldo, R, C);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// The same for a batch of matrices in ONE launch (blockIdx.z = matrix): after an optimizer step the 48 layer weights
// are re-transposed, and 48 five-microsecond launches cost more in gaps than in work.
struct TransposeBatch {
  const bf16_t* in[64];
  bf16_t* out[64];
  int ldi[64], ldo[64], R[64], C[64];
  int blk_begin[65];   // first workgroup of each matrix in the 1-D grid (64x64-element tiles, column-tile fastest)
};
__global__ __launch_bounds__(256) void transpose_batch_bf16(TransposeBatch b, int n) {
  __shared__ bf16_t tile[64][72];
  int z = 0;
  for (int i = 1; i < n; ++i)   // wave-uniform scan
    if ((int)blockIdx.x >= b.blk_begin[i]) z = i;
  const int R = b.R[z], C = b.C[z];
  const int local = blockIdx.x - b.blk_begin[z];
  const int bx = (C + 63) >> 6;
  const int r0 = (local / bx) * 64, c0 = (local % bx) * 64;
  const bf16_t* __restrict__ in = b.in[z];
  bf16_t* __restrict__ out = b.out[z];
  const long ldi = b.ldi[z], ldo = b.ldo[z];
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int chunk = t + 256 * it;
    const int r = chunk >> 3, cc = (chunk & 7) * 8;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (r0 + r < R && c0 + cc < C) v = *(const u32x4*)(in + (long)(r0 + r) * ldi + c0 + cc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tile[cc + 2 * i][r] = (bf16_t)(v[i] & 0xffffu);
      tile[cc + 2 * i + 1][r] = (bf16_t)(v[i] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int chunk = t + 256 * it;
    const int c = chunk >> 3, rr = (chunk & 7) * 8;
    if (c0 + c < C && r0 + rr < R) *(u32x4*)(out + (long)(c0 + c) * ldo + r0 + rr) = *(const u32x4*)(&tile[c][rr]);
  }
}

int vt_transpose_batch_dispatch(const void* const* in, const long* ldi, void* const* out, const long* ldo, const int* R,
                                const int* C, int n, hipStream_t stream) {
  if (!in || !out || !ldi || !ldo || !R || !C) return VT_ERR_NULL;
  if (n <= 0) return VT_ERR_BAD_SHAPE;
  for (int base = 0; base < n; base += 64) {
    const int cnt = n - base < 64 ? n - base : 64;
    TransposeBatch b;
    int blocks = 0;
    for (int i = 0; i < 64; ++i) {
      const int j = base + (i < cnt ? i : 0);
      if (!in[j] || !out[j]) return VT_ERR_NULL;
      // a row count that is not a multiple of 8 is served when the output rows have room for the rounded-up count
      // (the tail of the last 8-element group is written as zeros: padding columns of the transposed copy)
      if (R[j] <= 0 || C[j] <= 0 || (C[j] % 8) || ldo[j] < ((R[j] + 7) & ~7)) return VT_ERR_BAD_SHAPE;
      if ((ldi[j] % 8) || (ldo[j] % 8) || (((uintptr_t)in[j] | (uintptr_t)out[j]) & 15)) return VT_ERR_BAD_ALIGN;
      b.in[i] = (const bf16_t*)in[j]; b.out[i] = (bf16_t*)out[j]; b.ldi[i] = (int)ldi[j]; b.ldo[i] = (int)ldo[j];
      b.R[i] = R[j]; b.C[i] = C[j];
      b.blk_begin[i] = blocks;
      if (i < cnt) blocks += ((R[j] + 63) / 64) * ((C[j] + 63) / 64);
    }
    b.blk_begin[64] = blocks;
    hipLaunchKernelGGL(transpose_batch_bf16, dim3(blocks), dim3(256), 0, stream, b, cnt);
    if (hipGetLastError() != hipSuccess) return VT_ERR_HIP;
  }
  return VT_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused softmax cross-entropy over the MLM logits (tasks/viewpoint_select/encoder.py:387-389 + the
// argmax of :399 + the backward of the criterion): per supervised row, ONE kernel produces
//   loss_row = logsumexp(z) - z[y],  argmax(z),  dz = (softmax(z) - onehot(y)) * scale   (bf16, zero-padded)
// instead of torch's log_softmax / exp / scatter / mul / cast passes over a [rows, 30522] fp32 tensor.
// One 256-thread workgroup per row; the row (<= 122 KB) is read twice (second pass from L2).
__global__ __launch_bounds__(256) void ce_softmax_rows(const float* __restrict__ z, long ldz, const int64_t* __restrict__ y,
                                                       float* __restrict__ loss_row, int64_t* __restrict__ amax,
                                                       bf16_t* __restrict__ dz, long lddz, int V, int Vpad, float scale) {
  __shared__ float red_m[4], red_s[4], red_bv[4];
  __shared__ int red_bi[4];
  const long row = blockIdx.x;
  const float* zp = z + row * ldz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // pass 1: online max / sum-exp, and argmax (first index on ties, as torch.argmax)
  float m = -INFINITY, s = 0.f, bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = tid * 4; c < V; c += 1024) {
    float v[4];
    if (c + 4 <= V) {
      const f32x4 t = *(const f32x4*)(zp + c);
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (c + i < V) ? zp[c + i] : -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (v[i] > bv) { bv = v[i]; bi = c + i; }
      const float mn = fmaxf(m, v[i]);
      s = s * __expf(m - mn) + __expf(v[i] - mn);
      m = mn;
    }
  }
  // wave reduce (m, s) and (bv, bi)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (mn == -INFINITY) ? 0.f : s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
    const float bv2 = __shfl_xor(bv, o, 64);
    const int bi2 = __shfl_xor(bi, o, 64);
    if (bv2 > bv || (bv2 == bv && bi2 < bi)) { bv = bv2; bi = bi2; }
  }
  if (lane == 0) { red_m[wave] = m; red_s[wave] = s; red_bv[wave] = bv; red_bi[wave] = bi; }
  __syncthreads();
  m = red_m[0]; s = red_s[0]; bv = red_bv[0]; bi = red_bi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float mn = fmaxf(m, red_m[w]);
    s = s * __expf(m - mn) + red_s[w] * __expf(red_m[w] - mn);
    m = mn;
    if (red_bv[w] > bv || (red_bv[w] == bv && red_bi[w] < bi)) { bv = red_bv[w]; bi = red_bi[w]; }
  }
  const float lse = m + __logf(s);
  const int64_t label = y[row];
  if (tid == 0) {
    loss_row[row] = lse - zp[label];
    amax[row] = bi;
  }
  // pass 2: gradient row (skipped when the caller wants the loss and the argmax only: dz == nullptr)
  if (!dz) return;
  bf16_t* dp = dz + row * lddz;
  for (int c = tid * 8; c < Vpad; c += 2048) {
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = c + i;
      float p = 0.f;
      if (col < V) {
        p = __expf(zp[col] - lse);
        if (col == (int)label) p -= 1.0f;
        p *= scale;
      }
      g[i] = p;
    }
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(g[2 * i], g[2 * i + 1]);
    *(u32x4*)(dp + c) = o;
  }
}

// The same with the row held in registers (V <= 32 768: 32 x f32x4 per thread of a 256-thread workgroup): the logits are
// read from HBM ONCE, the maximum is taken without exponentials, exp(z - max) is computed once per element and kept, the
// gradient is that value times scale / sum (no second exponential, no second read).  Against the two-pass kernel above
// (one exponential pair per element in the online pass, a third in the gradient pass, the row read twice): 440 -> ~170 us
// for [4 272, 30 522] at B = 256.
template <int NV>
__global__ __launch_bounds__(256) void ce_softmax_rows_reg(const float* __restrict__ z, long ldz, const int64_t* __restrict__ y,
                                                           float* __restrict__ loss_row, int64_t* __restrict__ amax,
                                                           bf16_t* __restrict__ dz, long lddz, int V, int Vpad, float scale) {
  __shared__ float red_a[4], red_b[4];
  __shared__ int red_i[4];
  const long row = blockIdx.x;
  const float* zp = z + row * ldz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 v[NV];
  float bv = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = tid * 4 + k * 1024;
    if (c + 4 <= V) v[k] = *(const f32x4*)(zp + c);
    else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[k][i] = (c + i < V) ? zp[c + i] : -INFINITY;
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (v[k][i] > bv) { bv = v[k][i]; bi = tid * 4 + k * 1024 + i; }   // (ascending columns per thread: first maximum)
  // block argmax (value, then smaller index: torch.argmax's first maximum)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float bv2 = __shfl_xor(bv, o, 64);
    const int bi2 = __shfl_xor(bi, o, 64);
    if (bv2 > bv || (bv2 == bv && bi2 < bi)) { bv = bv2; bi = bi2; }
  }
  if (lane == 0) { red_a[wave] = bv; red_i[wave] = bi; }
  __syncthreads();
  bv = red_a[0]; bi = red_i[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (red_a[w] > bv || (red_a[w] == bv && red_i[w] < bi)) { bv = red_a[w]; bi = red_i[w]; }
  const float m = bv;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[k][i] = __expf(v[k][i] - m); s += v[k][i]; }   // (-inf columns: 0)
  s = wave_sum(s);
  if (lane == 0) red_b[wave] = s;
  __syncthreads();
  s = red_b[0] + red_b[1] + red_b[2] + red_b[3];
  const float lse = m + __logf(s);
  const int64_t label = y[row];
  if (tid == 0) {
    loss_row[row] = lse - zp[label];
    amax[row] = bi;
  }
  if (!dz) return;
  bf16_t* dp = dz + row * lddz;
  const float f = scale / s;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = tid * 4 + k * 1024;
    if (c < Vpad) {
      float g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = (c + i < V) ? v[k][i] * f - ((c + i == (int)label) ? scale : 0.f) : 0.f;
      u32x2 o;
      o[0] = pack_bf16x2(g[0], g[1]);
      o[1] = pack_bf16x2(g[2], g[3]);
      *(u32x2*)(dp + c) = o;
    }
  }
}

int vt_ce_softmax_dispatch(const float* z, long ldz, const int64_t* y, float* loss_row, int64_t* amax, void* dz, long lddz,
                           long rows, int V, int Vpad, float scale, hipStream_t stream) {
  if (!z || !y || !loss_row || !amax) return VT_ERR_NULL;
  if (!dz) { Vpad = (V + 7) / 8 * 8; lddz = Vpad; }   // no gradient row wanted
  if (rows <= 0 || V <= 0 || Vpad < V || (Vpad % 8) || lddz < Vpad) return VT_ERR_BAD_SHAPE;
  if ((ldz % 4) || (lddz % 8) || (((uintptr_t)z | (uintptr_t)dz) & 15)) return VT_ERR_BAD_ALIGN;
  const int need = (Vpad + 1023) / 1024;   // f32x4 per thread
  if (need <= 8)
    hipLaunchKernelGGL(ce_softmax_rows_reg<8>, dim3((unsigned)rows), dim3(256), 0, stream, z, ldz, y, loss_row, amax, (bf16_t*)dz,
                       lddz, V, Vpad, scale);
  else if (need <= 32)
    hipLaunchKernelGGL(ce_softmax_rows_reg<32>, dim3((unsigned)rows), dim3(256), 0, stream, z, ldz, y, loss_row, amax, (bf16_t*)dz,
                       lddz, V, Vpad, scale);
  else
    hipLaunchKernelGGL(ce_softmax_rows, dim3((unsigned)rows), dim3(256), 0, stream, z, ldz, y, loss_row, amax, (bf16_t*)dz, lddz,
                       V, Vpad, scale);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// The masked-region-token head's loss (tasks/viewpoint_select/encoder.py:323-326, 380-385): token_head = Linear +
// Softmax, and the criterion applies log-softmax AGAIN, so per supervised row
//     p = softmax(z),   loss = logsumexp(p) - p[y],   argmax = argmax(p) = argmax(z),
//     dL/dp_i = softmax(p)_i - onehot_i,   dL/dz_j = p_j (dL/dp_j - sum_i dL/dp_i p_i).
// One workgroup per row, the row (V <= 2048 classes) held in registers, three block reductions.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ce_double_softmax_rows(const float* __restrict__ z, long ldz, const int64_t* __restrict__ y,
                                                              float* __restrict__ loss_row, int64_t* __restrict__ amax,
                                                              bf16_t* __restrict__ dz, long lddz, int V, int Vpad, float scale) {
  __shared__ float red[4];
  __shared__ float red_bv[4];
  __shared__ int red_bi[4];
  const long row = blockIdx.x;
  const float* zp = z + row * ldz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = tid * 8;
  float v[8];
  float bv = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = (c0 + i < V) ? zp[c0 + i] : -INFINITY;
    if (v[i] > bv) { bv = v[i]; bi = c0 + i; }
  }
  // max and argmax (first index on ties, as torch.argmax)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float bv2 = __shfl_xor(bv, o, 64);
    const int bi2 = __shfl_xor(bi, o, 64);
    if (bv2 > bv || (bv2 == bv && bi2 < bi)) { bv = bv2; bi = bi2; }
  }
  if (lane == 0) { red_bv[wave] = bv; red_bi[wave] = bi; }
  __syncthreads();
  bv = red_bv[0]; bi = red_bi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (red_bv[w] > bv || (red_bv[w] == bv && red_bi[w] < bi)) { bv = red_bv[w]; bi = red_bi[w]; }
  // p = softmax(z)
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = (c0 + i < V) ? __expf(v[i] - bv) : 0.f; s1 += v[i]; }
  const float inv = 1.0f / block_sum_256(s1, red);
  // second softmax over p: s2 = sum exp(p), t = sum exp(p) p
  float s2 = 0.f, t = 0.f;
  float e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] *= inv;
    e[i] = (c0 + i < V) ? __expf(v[i]) : 0.f;
    s2 += e[i];
    t += e[i] * v[i];
  }
  s2 = block_sum_256(s2, red);
  t = block_sum_256(t, red);
  const int label = (int)y[row];
  const float lse2 = __logf(s2);
  // p[label] lives in thread label / 8
  __shared__ float p_label;
  if (label >= c0 && label < c0 + 8) {
    float pl = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) pl = (c0 + i == label) ? v[i] : pl;
    p_label = pl;
  }
  __syncthreads();
  const float py = p_label;
  if (tid == 0) {
    loss_row[row] = lse2 - py;
    amax[row] = bi;
  }
  const float dot = t / s2 - py;   // sum_i (softmax(p)_i - onehot_i) p_i
  if (!dz) return;                 // loss and argmax only
  bf16_t* dp = dz + row * lddz;
  if (c0 < Vpad) {
    float gz[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float dpi = e[i] / s2 - ((c0 + i == label) ? 1.0f : 0.f);
      gz[i] = (c0 + i < V) ? v[i] * (dpi - dot) * scale : 0.f;
    }
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(gz[2 * i], gz[2 * i + 1]);
    *(u32x4*)(dp + c0) = o;
  }
}

int vt_ce_double_softmax_dispatch(const float* z, long ldz, const int64_t* y, float* loss_row, int64_t* amax, void* dz, long lddz,
                                  long rows, int V, int Vpad, float scale, hipStream_t stream) {
  if (!z || !y || !loss_row || !amax) return VT_ERR_NULL;
  if (!dz) { Vpad = (V + 7) / 8 * 8; lddz = Vpad; }   // no gradient row wanted
  if (rows <= 0 || V <= 0 || Vpad < V || (Vpad % 8) || lddz < Vpad) return VT_ERR_BAD_SHAPE;
  if (Vpad > 2048) return VT_ERR_UNSUPPORTED;   // the row is held in registers: 256 threads x 8 classes
  if ((lddz % 8) || ((uintptr_t)dz & 15)) return VT_ERR_BAD_ALIGN;
  hipLaunchKernelGGL(ce_double_softmax_rows, dim3((unsigned)rows), dim3(256), 0, stream, z, ldz, y, loss_row, amax, (bf16_t*)dz,
                     lddz, V, Vpad, scale);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// x *= dropout mask * scale in place (bf16 [rows, cols], element index = row * cols + col): masks the
// compacted image-row gradient with the mask the region-projection epilogue used.
__global__ __launch_bounds__(256) void apply_dropout_bf16(bf16_t* __restrict__ x, long ld, long rows, int cols, DropCfg d) {
  const int cpr = cols >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cpr) return;
  const long row = i / cpr;
  const int col = (int)(i - row * cpr) * 8;
  u32x4 w = *(u32x4*)(x + row * ld + col);
  const uint32_t e0 = (uint32_t)row * (uint32_t)cols + (uint32_t)col;
  float v[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = bf16lo(w[k]); v[2 * k + 1] = bf16hi(w[k]); }
  vt_drop_run<8>(d, e0, v);
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
  *(u32x4*)(x + row * ld + col) = w;
}

int vt_apply_dropout_dispatch(void* x, long ld, long rows, int cols, const DropCfg& d, hipStream_t stream) {
  if (!x) return VT_ERR_NULL;
  if (rows <= 0 || cols <= 0 || (cols % 8) || (ld % 8) || rows * cols >= (1L << 32)) return VT_ERR_BAD_SHAPE;
  if (!d.thresh) return VT_OK;
  const long n = rows * (cols >> 3);
  hipLaunchKernelGGL(apply_dropout_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (bf16_t*)x, ld, rows, cols, d);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// out[i] = 1 if element i of the site is kept (tests: lets the CPU oracle run with the SAME masks)
// ATTN: the attention sites' function (a hash word per four keys, 8-bit thresholds: vt_keep_attn)
template <bool ATTN>
__global__ __launch_bounds__(256) void dropout_mask_dump(uint8_t* __restrict__ out, long n, DropCfg d) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (ATTN ? vt_keep_attn(d, (uint32_t)i) : vt_keep(d, (uint32_t)i)) ? 1 : 0;
}

int vt_dropout_mask_dispatch(uint8_t* out, long n, const DropCfg& d, hipStream_t stream, int attn) {
  if (!out) return VT_ERR_NULL;
  if (n <= 0 || n >= (1L << 32)) return VT_ERR_BAD_SHAPE;
  if (attn) hipLaunchKernelGGL(dropout_mask_dump<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n, d);
  else hipLaunchKernelGGL(dropout_mask_dump<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n, d);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---- table gradients of an embedding lookup without atomics ------------------------------------------------------
// grad[id[i], :] += sum of de[perm[j], :] over the rows j whose id equals id[i], for the sorted id list `sorted_ids`
// (perm = the stable sort's permutation): the scatter-add of BertEmbeddings' backward (torch: index_add_ = 25 M float
// atomics for the word table at B = 256, 179 us, and an order of additions that changes from run to run).  No atomics
// here: a table row is written by one wave, the rows of a run added in their original order (bitwise reproducible); the
// padding index `skip_id`, whose row torch.nn.Embedding leaves without gradient, is passed over.  Most runs are short (a
// token id seldom repeats in a batch), [CLS] repeats B times, and [MASK] may repeat thousands of times: hence the segments.
#define ETG_SEG 32   // rows one wave adds at most: a run of equal ids is cut at the multiples of ETG_SEG of the sorted order

// the rows [i, e) of the sorted order (one id), added in their original order into acc[3] (columns c0 + 256 k + 4 lane):
// 64 permutation entries per vector load, four rows in flight at a time
__device__ __forceinline__ void etg_sum_rows(const long* __restrict__ perm, const float* __restrict__ de, long ld_de, long i, long len,
                                             int c0, int H, int lane, f32x4 (&acc)[3]) {
  for (long j0 = 0; j0 < len; j0 += 64) {
    const long left = len - j0 < 64 ? len - j0 : 64;
    const long pv = lane < left ? perm[i + j0 + lane] : 0;
    for (int j = 0; j < (int)left; j += 4) {
      f32x4 v[4][3];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = j + r < (int)left;
        const long pr = ((long)__builtin_amdgcn_readlane((int)(pv >> 32), on ? j + r : 0) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)pv, on ? j + r : 0);
        const float* row = de + pr * ld_de;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = c0 + 256 * k + 4 * lane;
          v[r][k] = (on && c < H) ? *(const f32x4*)(row + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[k][e] += v[r][k][e];
    }
  }
}

// Pass 1, one WAVE per sorted position (four per workgroup).  A position works if it starts a SEGMENT: a run of equal ids, cut
// at every multiple of ETG_SEG of the sorted order (so that a token repeated thousands of times in a batch -- [MASK] -- is
// added by many waves, not by one).  A run that is one segment is added straight into its table row (plain loads and
// stores: the wave owns the row); the segments of a longer run park their sums in `scratch` (row = the segment's first
// sorted position) for pass 2.
__global__ __launch_bounds__(256) void embed_table_grad_runs(const int* __restrict__ sorted_ids, const long* __restrict__ perm,
                                                             const float* __restrict__ de, long ld_de, float* __restrict__ grad,
                                                             long ld_grad, long n, int H, long n_rows_table, long skip_id,
                                                             float* __restrict__ scratch, int* __restrict__ any_long) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const long id = sorted_ids[i];
  if (id == skip_id || id < 0 || id >= n_rows_table) return;          // uniform per wave
  const bool run_head = i == 0 || sorted_ids[i - 1] != id;
  if (!run_head && (i % ETG_SEG) != 0) return;                        // not the first row of a segment
  const long bound = (i / ETG_SEG + 1) * ETG_SEG < n ? (i / ETG_SEG + 1) * ETG_SEG : n;
  long e = i + 1;
  while (e < bound && sorted_ids[e] == id) ++e;                       // (uniform scalar loop, <= ETG_SEG steps)
  const bool run_ends = e == n || sorted_ids[e] != id;
  const bool whole_run = run_head && run_ends;
  if (!whole_run && lane == 0) *any_long = 1;
  float* dst = whole_run ? grad + id * ld_grad : scratch + i * (long)H;
  for (int c0 = 0; c0 < H; c0 += 768) {
    f32x4 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = c0 + 256 * k + 4 * lane;
      acc[k] = (whole_run && c < H) ? *(const f32x4*)(dst + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    etg_sum_rows(perm, de, ld_de, i, e - i, c0, H, lane, acc);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = c0 + 256 * k + 4 * lane;
      if (c < H) *(f32x4*)(dst + c) = acc[k];
    }
  }
}

// Pass 2 (does nothing unless pass 1 met a run of several segments): the head of such a run adds its segments' sums, in
// the order of the segments, into the table row.
__global__ __launch_bounds__(256) void embed_table_grad_join(const int* __restrict__ sorted_ids, float* __restrict__ grad, long ld_grad,
                                                             long n, int H, long n_rows_table, long skip_id,
                                                             const float* __restrict__ scratch, const int* __restrict__ any_long) {
  if (*any_long == 0) return;
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const long id = sorted_ids[i];
  if (id == skip_id || id < 0 || id >= n_rows_table) return;
  if (i > 0 && sorted_ids[i - 1] == id) return;                       // not a run head
  const long b1 = (i / ETG_SEG + 1) * ETG_SEG;                        // the run's second segment would start here
  if (b1 >= n || sorted_ids[b1] != id) return;                        // one segment: pass 1 added it straight into the table
  float* gr = grad + id * ld_grad;
  for (int c0 = 0; c0 < H; c0 += 768) {
    f32x4 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = c0 + 256 * k + 4 * lane;
      acc[k] = c < H ? *(const f32x4*)(gr + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the first segment starts at i, the others at the multiples of ETG_SEG after it; four sums in flight at a time
    long sgm = i;
    while (sgm < n && sorted_ids[sgm] == id) {
      f32x4 v[4][3];
      bool on[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        on[r] = sgm < n && sorted_ids[sgm] == id;
        const float* src = scratch + (on[r] ? sgm : i) * (long)H;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = c0 + 256 * k + 4 * lane;
          v[r][k] = (on[r] && c < H) ? *(const f32x4*)(src + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (on[r]) sgm = (sgm / ETG_SEG + 1) * ETG_SEG;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (on[r])
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[k][e] += v[r][k][e];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = c0 + 256 * k + 4 * lane;
      if (c < H) *(f32x4*)(gr + c) = acc[k];
    }
  }
}

int vt_embed_table_grad_dispatch(const int* sorted_ids, const long* perm, const float* de, long ld_de, float* grad, long ld_grad,
                                 long n, int H, long n_rows_table, long skip_id, float* scratch, int* flag, hipStream_t stream) {
  if (!sorted_ids || !perm || !de || !grad || !scratch || !flag) return VT_ERR_NULL;
  if (n <= 0 || H <= 0 || (H & 3) || n_rows_table <= 0 || n > 0x7fffffffL) return VT_ERR_BAD_SHAPE;
  if ((ld_de & 3) || (ld_grad & 3) || (((uintptr_t)de | (uintptr_t)grad | (uintptr_t)scratch) & 15)) return VT_ERR_BAD_ALIGN;
  if (hipMemsetAsync(flag, 0, sizeof(int), stream) != hipSuccess) return VT_ERR_HIP;
  const unsigned nwg = (unsigned)((n + 3) / 4);
  hipLaunchKernelGGL(embed_table_grad_runs, dim3(nwg), dim3(256), 0, stream, sorted_ids, perm, de, ld_de, grad, ld_grad, n, H,
                     n_rows_table, skip_id, scratch, flag);
  hipLaunchKernelGGL(embed_table_grad_join, dim3(nwg), dim3(256), 0, stream, sorted_ids, grad, ld_grad, n, H, n_rows_table, skip_id,
                     scratch, flag);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---- the per-key attention mask as the kernels take it ---------------------------------------------------------------
// out[b, :] = float(mask[b, :]) - max_s float(mask[b, s]) + 1 (modeling._centered_mask: one constant per sequence off the
// additive bias (1 - m) * -10000 of encoder.py:238-241, a bitwise no-op for a 0/1 mask with a kept key), from the dtypes
// callers pass (float32, int64, int32, one-byte bool / uint8): one launch where torch took a cast, amax, sub and add.
template <typename T>
__global__ __launch_bounds__(256) void center_mask_rows(const T* __restrict__ m, long ldm, float* __restrict__ out, int S) {
  __shared__ float red[4];
  const T* row = m + (long)blockIdx.x * ldm;
  float mx = -INFINITY;
  for (int s = threadIdx.x; s < S; s += 256) mx = fmaxf(mx, (float)row[s]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int s = threadIdx.x; s < S; s += 256) out[(long)blockIdx.x * S + s] = ((float)row[s] - mx) + 1.0f;
}

// kind: 0 float32, 1 int64, 2 int32, 3 one byte (bool / uint8)
int vt_center_mask_dispatch(const void* mask, int kind, long ldm, float* out, int B, int S, hipStream_t stream) {
  if (!mask || !out) return VT_ERR_NULL;
  if (B <= 0 || S <= 0 || ldm < S) return VT_ERR_BAD_SHAPE;
  switch (kind) {
    case 0: hipLaunchKernelGGL(center_mask_rows<float>, dim3(B), dim3(256), 0, stream, (const float*)mask, ldm, out, S); break;
    case 1: hipLaunchKernelGGL(center_mask_rows<long>, dim3(B), dim3(256), 0, stream, (const long*)mask, ldm, out, S); break;
    case 2: hipLaunchKernelGGL(center_mask_rows<int>, dim3(B), dim3(256), 0, stream, (const int*)mask, ldm, out, S); break;
    case 3: hipLaunchKernelGGL(center_mask_rows<unsigned char>, dim3(B), dim3(256), 0, stream, (const unsigned char*)mask, ldm, out, S); break;
    default: return VT_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---- what the host must know about a training batch, and its row lists -------------------------------------------
// The pretrain step runs its heads on the supervised rows only (labels != -1, token_labels != -1: encoder.py:377-385,
// CrossEntropyLoss(ignore_index=-1)) and its encoder on the rows with a non-zero attention mask only; the host needs the
// three counts to size those launches.  batch_row_counts reduces them -- with the verdict on the compacted layout (a 0/1
// mask, every [CLS] and every supervised position kept) and the embedding kernel's out-of-range flag -- into five words the
// host reads back in one synchronisation; batch_row_lists then writes the row lists (ascending), the padded -> compact
// map and the per-sequence start / length.  Three short launches where torch needed ~25 (sum, any, nonzero, cumsum, where ...).
struct BatchRowsArgs {
  const long* lab; const long* tl;      // [M] or null
  const float* mask;                    // [B*S] fp32 or null (no compaction wanted)
  const int* err;                       // the embedding kernel's flag or null
  long M; int S; int B;
  long* counts;                         // [5]: err, n_w, n_t, n_keep, bad   (zeroed by the caller)
  int* tile_counts;                     // [ntiles][3]: per 1024-position tile (written by the counts kernel, read by the lists kernel)
  long* idx_w; long* idx_t;             // row lists
  long* index; long* inverse;           // kept rows; padded position -> compact row or -1
  int* start; int* length;              // [B]
  long n_w, n_t, n_keep;                // capacities of the three lists (the counts batch_row_counts reported)
};

// one workgroup per tile of 1024 positions (one position per thread: coalesced)
__global__ __launch_bounds__(1024) void batch_row_counts(BatchRowsArgs a) {
  __shared__ int red[16][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long i = (long)blockIdx.x * 1024 + tid;
  int w = 0, t = 0, k = 0, bad = 0;
  if (i < a.M) {
    w = a.lab && a.lab[i] != -1;
    t = a.tl && a.tl[i] != -1;
    k = 1;
    if (a.mask) {
      const float m = a.mask[i];
      k = m != 0.f;
      if ((k && m != 1.f) || ((w || t) && !k) || (!k && (i % a.S) == 0)) bad = 1;
    }
  }
  const int nw = __builtin_popcountll(__builtin_amdgcn_ballot_w64(w != 0)), nt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(t != 0));
  const int nk = __builtin_popcountll(__builtin_amdgcn_ballot_w64(k != 0)), nb = __builtin_amdgcn_ballot_w64(bad != 0) != 0;
  if (lane == 0) { red[wave][0] = nw; red[wave][1] = nt; red[wave][2] = nk; red[wave][3] = nb; }
  __syncthreads();
  if (tid == 0) {
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int v = 0; v < 16; ++v) { s0 += red[v][0]; s1 += red[v][1]; s2 += red[v][2]; s3 |= red[v][3]; }
    a.tile_counts[3 * blockIdx.x + 0] = s0; a.tile_counts[3 * blockIdx.x + 1] = s1; a.tile_counts[3 * blockIdx.x + 2] = s2;
    // (integer atomics: the totals do not depend on the order)
    if (s0) atomicAdd((unsigned long long*)&a.counts[1], (unsigned long long)s0);
    if (s1) atomicAdd((unsigned long long*)&a.counts[2], (unsigned long long)s1);
    if (s2) atomicAdd((unsigned long long*)&a.counts[3], (unsigned long long)s2);
    if (s3) atomicOr((unsigned long long*)&a.counts[4], 1ull);
    if (blockIdx.x == 0 && a.err) a.counts[0] = (long)a.err[0];
  }
}

// grid (ntiles, 3): blockIdx.y 0 = labels list, 1 = token-labels list, 2 = kept rows + inverse map.  A tile's first output
// slot is the sum of the earlier tiles' counts; inside the tile the slots follow the positions (ballot prefix per wave,
// the waves' totals through LDS).
__global__ __launch_bounds__(1024) void batch_row_lists(BatchRowsArgs a) {
  __shared__ int wsum[16];
  __shared__ long s_base;
  const int which = blockIdx.y;
  if ((which == 0 && !a.lab) || (which == 1 && !a.tl) || (which == 2 && !a.mask)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // base: counts of the tiles before this one (<= a few hundred values)
  long part = 0;
  for (int t = tid; t < (int)blockIdx.x; t += 1024) part += a.tile_counts[3 * t + which];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) wsum[wave] = (int)part;
  __syncthreads();
  if (tid == 0) { long b = 0; for (int v = 0; v < 16; ++v) b += wsum[v]; s_base = b; }
  __syncthreads();
  const long base = s_base;
  const long i = (long)blockIdx.x * 1024 + tid;
  bool f = false;
  if (i < a.M) f = which == 0 ? a.lab[i] != -1 : (which == 1 ? a.tl[i] != -1 : a.mask[i] != 0.f);
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(f);
  const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
  __syncthreads();                       // (wsum is reused)
  if (lane == 0) wsum[wave] = __builtin_popcountll(bal);
  __syncthreads();
  int wbase = 0;
  for (int v = 0; v < wave; ++v) wbase += wsum[v];
  const long pos = base + wbase + before;
  long* out = which == 0 ? a.idx_w : (which == 1 ? a.idx_t : a.index);
  const long cap = which == 0 ? a.n_w : (which == 1 ? a.n_t : a.n_keep);   // (a stale count must not write past a list)
  if (f && pos < cap) out[pos] = i;
  if (which == 2 && i < a.M) a.inverse[i] = f ? pos : -1;
}

// per-sequence first compact row and number of kept rows, from the finished inverse map ([CLS] of every sequence kept)
__global__ __launch_bounds__(256) void batch_seq_starts(BatchRowsArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const long s0 = a.inverse[(long)b * a.S];
  const long s1 = b + 1 < a.B ? a.inverse[(long)(b + 1) * a.S] : a.n_keep;
  a.start[b] = (int)s0;
  a.length[b] = (int)(s1 - s0);
}

int vt_batch_rows_dispatch(const BatchRowsArgs& a, int lists, hipStream_t stream) {
  if (a.M <= 0 || a.S <= 0 || a.B <= 0 || (long)a.B * a.S != a.M) return VT_ERR_BAD_SHAPE;
  if (!a.tile_counts) return VT_ERR_NULL;
  const unsigned ntiles = (unsigned)((a.M + 1023) / 1024);
  if (!lists) {
    if (!a.counts) return VT_ERR_NULL;
    hipLaunchKernelGGL(batch_row_counts, dim3(ntiles), dim3(1024), 0, stream, a);
  } else {
    if ((a.lab && a.n_w > 0 && !a.idx_w) || (a.tl && a.n_t > 0 && !a.idx_t) ||
        (a.mask && ((a.n_keep > 0 && !a.index) || !a.inverse || !a.start || !a.length))) return VT_ERR_NULL;
    if (a.n_w < 0 || a.n_t < 0 || a.n_keep < 0) return VT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(batch_row_lists, dim3(ntiles, 3), dim3(1024), 0, stream, a);
    if (a.mask) hipLaunchKernelGGL(batch_seq_starts, dim3((a.B + 255) / 256), dim3(256), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---- the 1-in-A action head's loss, accuracy and gradient in one launch ------------------------------------------
// NextActionPrediction = Linear + LogSoftmax (tasks/viewpoint_select/encoder.py:142-151) and the criterion
// CrossEntropyLoss(ignore_index=-1) applies log_softmax AGAIN (encoder.py:387-391): on logits z [B, A] with targets y,
//   l = log_softmax(z), m = log_softmax(l), loss = -sum_{valid b} m[b, y_b] / n_valid, accuracy = #(argmax l == y) / B,
//   dz = g - exp(l) * rowsum(g),  g = (exp(m) - onehot(y)) * valid * grad_scale / n_valid
// (torch: two log_softmax, gather, clamp, exp, scatter_add, several multiplies and reductions = ~20 launches on a [256, 36]
// tensor).  One workgroup, one wave per row in turn; A <= 64.  n_valid = 0 gives the NaN torch gives.
__global__ __launch_bounds__(1024) void action_head_rows(const float* __restrict__ z, long ldz, const long* __restrict__ y, int B,
                                                         int A, float grad_scale, bf16_t* __restrict__ dz, long lddz, int Ap,
                                                         float* __restrict__ out /* loss, accuracy */) {
  __shared__ float red[16][2];
  __shared__ int redn[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int nv = 0;
  for (int b = tid; b < B; b += 1024) nv += y[b] != -1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o, 64);
  if (lane == 0) redn[wave] = nv;
  __syncthreads();
  int n_valid = 0;
  for (int w = 0; w < 16; ++w) n_valid += redn[w];
  const float inv_n = 1.0f / (float)n_valid;          // inf when nothing is valid: loss 0 * inf = NaN, as torch's 0 / 0
  float loss = 0.f, hits = 0.f;
  for (int b = wave; b < B; b += 16) {
    const float v = lane < A ? z[b * ldz + lane] : -INFINITY;
    const float mx = wave_max(v);
    const float e = lane < A ? __expf(v - mx) : 0.f;
    const float l = v - mx - __logf(wave_sum(e));                     // log_softmax(z)
    const float lm = lane < A ? l : -INFINITY;
    const float mx2 = wave_max(lm);
    const float e2 = lane < A ? __expf(lm - mx2) : 0.f;
    const float m = lm - mx2 - __logf(wave_sum(e2));                  // log_softmax(log_softmax(z))
    const long yb = y[b];
    const bool valid = yb != -1;
    const int yc = yb < 0 ? 0 : (int)yb;                              // torch: clamp(min=0) for the gather
    // argmax of l (first maximum, as torch.argmax)
    const unsigned long long atmax = __builtin_amdgcn_ballot_w64(lane < A && lm == mx2);
    const int amax = __builtin_ctzll(atmax);
    const float m_y = __shfl(m, yc < A ? yc : 0, 64);
    if (lane == 0) {
      if (valid) loss -= m_y;
      hits += (long)amax == yb ? 1.f : 0.f;
    }
    // (exp(m) - onehot) * valid * (grad_scale / n_valid); an ignored row is 0 * (grad_scale / n_valid): NaN only when n_valid = 0
    const float g = lane < A ? (__expf(m) - (lane == yc ? 1.f : 0.f)) * ((valid ? 1.f : 0.f) * (grad_scale * inv_n)) : 0.f;
    const float gs = wave_sum(g);
    const float d = g - __expf(l) * gs;
    if (lane < Ap) dz[b * lddz + lane] = f32_to_bf16(lane < A ? d : 0.f);
  }
  if (lane == 0) { red[wave][0] = loss; red[wave][1] = hits; }
  __syncthreads();
  if (tid == 0) {
    float ls = 0.f, hs = 0.f;
    for (int w = 0; w < 16; ++w) { ls += red[w][0]; hs += red[w][1]; }
    out[0] = ls * inv_n;
    out[1] = hs / (float)B;
  }
}

int vt_action_head_dispatch(const float* z, long ldz, const long* y, int B, int A, float grad_scale, void* dz, long lddz, int Ap,
                            float* out, hipStream_t stream) {
  if (!z || !y || !dz || !out) return VT_ERR_NULL;
  if (B <= 0 || A <= 0 || A > 64 || Ap < A || Ap > 64 || ldz < A || lddz < Ap) return VT_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(action_head_rows, dim3(1), dim3(1024), 0, stream, z, ldz, y, B, A, grad_scale, (bf16_t*)dz, lddz, Ap, out);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---- weight prefetch (round 6) --------------------------------------------------------------------------------------------------
// At small batch the layer GEMMs are ONE round of tiles whose K loop prefetches two K-steps ahead; a weight matrix that was last
// touched a step ago comes from HBM, not from the Infinity Cache (the step streams ~1.5 GB of activations through its 256 MB
// between two uses of a weight), and a K = 3 072 tile then stalls on most of its 48 dependent K-steps: FFN-down 45 us on warm
// weights, 67 us in the step (tools/r6/pair_cold.py: distinct weights AND distinct activations per layer reproduce it, either
// alone does not; reading W2 once before FFN-up gives the 12 us back).  This kernel reads up to four byte ranges and drops
// the data: what it leaves behind is the lines in the Infinity Cache (and in the L2 of the XCD that happened to read them).
__global__ __launch_bounds__(256) void prefetch_ranges(PrefetchArgs a) { vt_prefetch_role(a, blockIdx.x, gridDim.x); }

// host side: the ranges a caller names (null / empty ones dropped), and how many workgroups they are worth
static bool prefetch_pack(const void* const* ptrs, const long* bytes, int n, PrefetchArgs& a, long& wgs) {
  long total = 0;
  a.n = 0;
  for (int i = 0; i < n && i < 4; ++i) {
    if (!ptrs[i] || bytes[i] <= 0 || ((uintptr_t)ptrs[i] & 15)) continue;
    a.p[a.n] = ptrs[i];
    a.bytes[a.n] = bytes[i];
    total += bytes[i];
    ++a.n;
  }
  for (int i = a.n; i < 4; ++i) { a.p[i] = nullptr; a.bytes[i] = 0; }
  wgs = (total + 16383) >> 14;
  return a.n > 0;
}

int vt_prefetch_dispatch(const void* const* ptrs, const long* bytes, int n, hipStream_t stream) {
  if (n <= 0) return VT_OK;
  if (n > 4 || !ptrs || !bytes) return VT_ERR_BAD_SHAPE;
  PrefetchArgs a;
  long grid;
  if (!prefetch_pack(ptrs, bytes, n, a, grid)) return VT_OK;
  grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
  hipLaunchKernelGGL(prefetch_ranges, dim3((unsigned)grid), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
