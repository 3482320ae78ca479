// Shared by the NT GEMM kernels (gemm_bf16.hip, gemm_v7.hip): argument block and the register epilogue.
#pragma once
#include "common.hpp"

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;
  const bf16_t* R;   // residual added after the activation; for ACT_MUL: the factor (saved gelu')
  void* C;
  bf16_t* C2;        // optional second output for backward: gelu'(acc + bias) if ACT_GELU, else acc + bias
  long lda, ldw, ldr, ldc, ldc2;
  int M, N, K;
  int grp_rows, grp_stride;  // output row = (m / grp_rows) * grp_stride + m % grp_rows  (0: identity)
  int tiles_m, tiles_n;
  unsigned long long* trace; // debug: per-workgroup phase timestamps of the persistent kernel (vt_debug_set_gemm_trace)
  DropCfg drop;              // dropout on act(acc + bias) BEFORE the residual add (BertSelfOutput / BertOutput /
                             // image embedding); element index = m * N + n
  // ---- deferred LayerNorm (the inference path's fused residual + LayerNorm, gemm_v7_ln.hip) ---------------------
  // The residual stream between two sub-layers is the PRE-LayerNorm sum v, kept as fp16 [M, H] (the stream itself: 11
  // significant bits against bf16's 8; values beyond +-65504 saturate) plus a bf16 copy (the next GEMM's A operand) plus
  // per-row partial statistics stats[p][row] = (sum, sum of squares) of the UNROUNDED fp32 sums over the 128-column slice p
  // (p < H / 128).  LayerNorm itself is never run as a pass of its own:
  //   ln_mode 1 (consumer: QKV, FFN-up)    C = act(rstd_r * (A W'^T - mean_r * g) + h)     A = bf16 copy of v, W' = W * gamma,
  //                                        g = rowsum(W') (colv), h = W beta + b (bias): LN(v) W^T + b with the
  //                                        normalisation applied to the accumulator;
  //   ln_mode 2 (producer: out-proj, FFN-down)  v' = A W^T + cb + LN(v)                    LN(v) = (Rs - mean_r) rstd_r gamma
  //                                        (colv) + beta, cb = b + beta (bias); writes v' as fp16 (Cs), bf16 (C) and its
  //                                        partial statistics (stats_out, slices of this tile's columns).
  int ln_mode;
  int ln_np;                 // partials per row of ln_stats (H / 128, <= 8)
  int ln_rows;               // rows per slice of ln_stats / stats_out (>= M, even: slices stay 16-byte aligned)
  float ln_inv_n, ln_eps;    // 1 / H, LayerNorm epsilon
  const float* ln_stats;     // [ln_np][ln_rows][2]
  const float* colv;         // per-column vector, see above
  const uint16_t* Rs;        // mode 2: the fp16 stream in (row stride ldrs)
  uint16_t* Cs;              // mode 2: the fp16 stream out (row stride ldcs)
  float* stats_out;          // mode 2: [N / 128][ln_rows][2]
  long ldrs, ldcs;
  // ---- split-K (one-tile-per-workgroup kernel, fp32 partial planes): workgroup (tile, s) reduces K-steps
  // [s * kq, min((s + 1) * kq, K / 64)), kq = ceil(K / 64 / ksplit), and writes plane s of C (c_plane elements apart);
  // a reduce pass sums the planes.  For the shapes whose tile count leaves most of the chip idle and whose K is long:
  // the MLM decoder's dgrad, [4 272, 30 528] x [30 528, 768] = 51 tiles of 256 x 256 with 477 K-steps each.
  int ksplit;                // 0 / 1: off
  int reverse;               // persistent kernel: walk the tile order backwards (experiment: consume a > 256 MB operand in the reverse
                             // of the order its producer wrote it, so that the rows still in the Infinity Cache are read first)
  long c_plane;
  // ---- fp16 copies of the residual stream in the seven-launch (training) layer ------------------------------------
  // r_f16: the residual operand R holds fp16 (the previous LayerNorm's output, kept beside its bf16 copy);
  // c_f16: C is written as fp16 (saturating): the pre-LayerNorm sum dense(h) + bias + residual.  Same bytes as bf16, three
  // more significant bits where the stream is rounded twice per sub-layer.
  int r_f16, c_f16;
  // ---- the residual as a LayerNorm that was never written out (training layer, vt_layer_acts::ln_residual_mode) ------
  // r_mean != null: R holds the fp16 PRE-LayerNorm sum v of the previous sub-layer (r_f16 is set) and the value added is
  // LayerNorm(v) = (v - r_mean[row]) * r_rstd[row] * r_gamma[col] + r_beta[col], the statistics being the ones the
  // LayerNorm kernel wrote beside its bf16 output -- so that kernel writes ONE output instead of two (the fp16 copy of its
  // output was 2 of its 6 bytes per element) and the stream is not rounded to fp16 a second time.
  const float* r_mean = nullptr;
  const float* r_rstd = nullptr;
  const float* r_gamma = nullptr;
  const float* r_beta = nullptr;
  // ---- stream-K region of the persistent kernel (gemm_nt_bf16_v8, round 6; kernel variants 28 .. 32) --------------------
  // sk_parts > 1: the tiles an XCD's workgroups cannot take in whole rounds are worked as ONE sequence of K-steps cut into
  // equal contiguous shares (see the kernel).  A share that starts inside a tile leaves that tile's partial accumulators
  // (fp32, raw register layout, 256 KiB) in slot [xcd][workgroup] of sk_ws; the workgroup holding the tile's first K-steps
  // adds the parts to its own in workgroup order (deterministic) and runs the ordinary epilogue.  sk_sem[xcd][workgroup]:
  // arrival counter of the tile that workgroup finishes, left at zero by it.  (vt_gemm_set_workspace)
  float* sk_ws = nullptr;
  int* sk_sem = nullptr;
  unsigned* sk_err = nullptr;   // bounded waits that ran out (host: vt_gemm_shared_tile_timeouts)
  int sk_parts = 0;
};
#define V8_SK_WGS_PER_XCD 32                   // workgroups per XCD the region is laid out for (a 256-CU grid)
#define V8_SK_PART_BYTES (256 * 64 * 16)       // one part's accumulators: 256 lanes x 64 registers of 16 B
#define V8_SK_REGION_BYTES ((long)8 * V8_SK_WGS_PER_XCD * V8_SK_PART_BYTES + 4096)   // + the counters and the error word

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_TANH = 2, ACT_MUL = 3 };  // MUL: out = acc * R (R = saved gelu'(pre-activation))

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_TILE_BYTES (128 * 64 * 2)
#define GEMM_LDS_BYTES (4 * GEMM_TILE_BYTES)
#define GEMM_DEFAULT_VARIANT 1

template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
  if (ACT == ACT_GELU) return gelu_erf(x);
  if (ACT == ACT_TANH) return tanh_fast(x);
  return x;  // ACT_NONE and ACT_MUL (the latter multiplies by R where R is read)
}

// ---- register epilogue pieces ----------------------------------------------------------------------------------
// Lane (j = lane&15, gq = lane>>4) owns output rows row0+16mt+j (mt = 0..3) and the 16 consecutive columns
// nb = col0+16gq .. +15 (accumulators a[t][e] -> column 4t+e); bias, activation, dropout and residual are applied
// in registers.  The pieces work on ONE 16-row block so that callers holding many accumulators (the 256x256-tile
// kernels) keep only 16 of them in VGPRs at a time.
__device__ __forceinline__ void epi_load_bias(const GemmArgs& g, int nb, bool full, float (&bv)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) bv[i] = 0.f;
  if (g.bias) {
    if (full) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 t4 = *(const f32x4*)(g.bias + nb + 4 * i);
        bv[4 * i + 0] = t4[0]; bv[4 * i + 1] = t4[1]; bv[4 * i + 2] = t4[2]; bv[4 * i + 3] = t4[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (nb + i < g.N) bv[i] = g.bias[nb + i];
    }
  }
}

__device__ __forceinline__ long epi_out_row(const GemmArgs& g, int m) {
  return g.grp_rows ? (long)(m / g.grp_rows) * g.grp_stride + (m % g.grp_rows) : (long)m;
}

// v = act(a + bias) [dropout] [+|* residual] for one row (all 16 columns valid); s2 = what C2 stores
template <int ACT, bool WANT_S2>
__device__ __forceinline__ void epi_row_values(const GemmArgs& g, const f32x4 (&a)[4], const float (&bv)[16], int m, long orow, int nb,
                                               float (&v)[16], float (&s2)[16]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * t + e] = a[t][e] + bv[4 * t + e];
  if (WANT_S2 && ACT == ACT_GELU) {   // value and derivative share the reciprocal / polynomial / exponential
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      f32x2 gg, dd;
      gelu_erf_both2((f32x2){v[i], v[i + 1]}, gg, dd);
      v[i] = gg[0]; v[i + 1] = gg[1]; s2[i] = dd[0]; s2[i + 1] = dd[1];
    }
  } else {
    if (WANT_S2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s2[i] = v[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = apply_act<ACT>(v[i]);
  }
  if (g.drop.thresh) {
    const uint32_t e0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)nb;
    vt_drop_run<16>(g.drop, e0, v);
  }
  if (g.R) {
    const u32x4* rp = (const u32x4*)(g.R + orow * g.ldr + nb);
    const u32x4 r0 = rp[0], r1 = rp[1];
    float rv[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rv[2 * i] = g.r_f16 ? f16lo(r0[i]) : bf16lo(r0[i]);
      rv[2 * i + 1] = g.r_f16 ? f16hi(r0[i]) : bf16hi(r0[i]);
      rv[8 + 2 * i] = g.r_f16 ? f16lo(r1[i]) : bf16lo(r1[i]);
      rv[8 + 2 * i + 1] = g.r_f16 ? f16hi(r1[i]) : bf16hi(r1[i]);
    }
    if (g.r_mean) {   // the residual is LayerNorm(v), v = the fp16 row just read (GemmArgs::r_mean)
      const float mu = g.r_mean[orow], rs = g.r_rstd[orow];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 gm = *(const f32x4*)(g.r_gamma + nb + 4 * i), bt = *(const f32x4*)(g.r_beta + nb + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[4 * i + e] = (rv[4 * i + e] - mu) * rs * gm[e] + bt[e];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (ACT == ACT_MUL) ? v[i] * rv[i] : v[i] + rv[i];
  }
}

// one 16-row block, stored straight from registers: 32 (bf16) or 64 (fp32) contiguous bytes per lane
template <int ACT, bool OUT_F32>
__device__ __forceinline__ void epi_row_direct(const GemmArgs& g, const f32x4 (&a)[4], const float (&bv)[16], int m, int nb, bool full) {
  if (m >= g.M) return;
  const long orow = epi_out_row(g, m);
  if (full) {
    float v[16], s2[16];
    if (g.C2) {
      epi_row_values<ACT, true>(g, a, bv, m, orow, nb, v, s2);
      u32x4* cp2 = (u32x4*)(g.C2 + orow * g.ldc2 + nb);
      u32x4 o0, o1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o0[i] = pack_bf16x2(s2[2 * i], s2[2 * i + 1]);
        o1[i] = pack_bf16x2(s2[8 + 2 * i], s2[8 + 2 * i + 1]);
      }
      cp2[0] = o0;
      cp2[1] = o1;
    } else {
      epi_row_values<ACT, false>(g, a, bv, m, orow, nb, v, s2);
    }
    if (OUT_F32) {
      f32x4* cp = (f32x4*)((float*)g.C + orow * g.ldc + nb);
#pragma unroll
      for (int i = 0; i < 4; ++i) cp[i] = (f32x4){v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    } else {
      u32x4* cp = (u32x4*)((bf16_t*)g.C + orow * g.ldc + nb);
      u32x4 o0, o1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o0[i] = g.c_f16 ? pack_f16x2(v[2 * i], v[2 * i + 1]) : pack_bf16x2(v[2 * i], v[2 * i + 1]);
        o1[i] = g.c_f16 ? pack_f16x2(v[8 + 2 * i], v[8 + 2 * i + 1]) : pack_bf16x2(v[8 + 2 * i], v[8 + 2 * i + 1]);
      }
      cp[0] = o0;
      cp[1] = o1;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (nb + i < g.N) {
        const float pre = a[i >> 2][i & 3] + bv[i];
        if (g.C2) g.C2[orow * g.ldc2 + nb + i] = f32_to_bf16((ACT == ACT_GELU) ? gelu_erf_grad(pre) : pre);
        float x = apply_act<ACT>(pre);
        if (g.drop.thresh) x = vt_keep(g.drop, (uint32_t)m * (uint32_t)g.N + (uint32_t)(nb + i)) ? x * g.drop.scale : 0.f;
        if (g.R) {
          const uint16_t rb = g.R[orow * g.ldr + nb + i];
          float rr = g.r_f16 ? f16bits_to_f32(rb) : bf16_to_f32(rb);
          if (g.r_mean) rr = (rr - g.r_mean[orow]) * g.r_rstd[orow] * g.r_gamma[nb + i] + g.r_beta[nb + i];
          x = (ACT == ACT_MUL) ? x * rr : x + rr;
        }
        if (OUT_F32) ((float*)g.C)[orow * g.ldc + nb + i] = x;
        else ((bf16_t*)g.C)[orow * g.ldc + nb + i] = g.c_f16 ? f32_to_f16bits(x) : f32_to_bf16(x);
      }
    }
  }
}

// Epilogue of the 128x128-tile kernels: a 64x64 wave tile held as acc[mt][t].
template <int ACT, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[4][4], int lane, int row0, int col0) {
  const int nb = col0 + 16 * (lane >> 4);
  if (nb >= g.N) return;
  const bool full = (nb + 16 <= g.N);
  float bv[16];
  epi_load_bias(g, nb, full, bv);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) epi_row_direct<ACT, OUT_F32>(g, acc[mt], bv, row0 + 16 * mt + (lane & 15), nb, full);
}

// ---- bf16 epilogue with an LDS transpose (256x256-tile kernels) -------------------------------------------------
// The register path stores 2 x 16 B per lane and row: one store instruction scatters 64 16-byte pieces over 16
// rows.  Here a wave parks a 64x64 quadrant in 8 KiB of LDS (same XOR swizzle as the operand images: conflict-free
// both ways) and stores it back row-major, 8 full 128-byte lines per store instruction.  LDS traffic is inline asm
// with hand-placed lgkmcnt waits: the compiler would drain the in-flight LDS-DMA of the next tile (vmcnt(0)) in
// front of compiler-visible LDS accesses.
__device__ __forceinline__ unsigned epi_lds_addr(unsigned base, int row, int chunk) {
  return base + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ void epi_lds_put(unsigned lds_wave, int lane, int mt, const float (&v)[16]) {
  u32x4 o0, o1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o0[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    o1[i] = pack_bf16x2(v[8 + 2 * i], v[8 + 2 * i + 1]);
  }
  const int row = 16 * mt + (lane & 15), gq = lane >> 4;
  asm volatile("ds_write_b128 %0, %1" ::"v"(epi_lds_addr(lds_wave, row, 2 * gq)), "v"(o0) : "memory");
  asm volatile("ds_write_b128 %0, %1" ::"v"(epi_lds_addr(lds_wave, row, 2 * gq + 1)), "v"(o1) : "memory");
}
// the parked 32x64 bf16 slab -> C[row0 .. row0+31, col0 .. col0+63], rows past M masked
__device__ __forceinline__ void epi_store_slab(const GemmArgs& g, bf16_t* C, long ldc, unsigned lds_wave, int lane, int row0, int col0) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  u32x4 rv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = epi_lds_addr(lds_wave, 8 * i + (lane >> 3), lane & 7);
    asm volatile("ds_read_b128 %0, %1" : "=v"(rv[i]) : "v"(a));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = row0 + 8 * i + (lane >> 3);
    asm volatile("" : "+v"(rv[i]));   // the value is final only after the wait above
    if (m < g.M) *(u32x4*)(C + epi_out_row(g, m) * ldc + col0 + 8 * (lane & 7)) = rv[i];
  }
}
