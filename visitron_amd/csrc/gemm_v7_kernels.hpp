// NT GEMM, 256x256 tile with 128x128 wave tiles and AGPR accumulators: the kernel templates shared by gemm_v7.hip (plain
// epilogues) and gemm_v7_ln.hip (the deferred-LayerNorm epilogues of the inference path).
#pragma once
#include "gemm_common.hpp"

// ================================================================================================
// v7: 256x256 tile, BK = 64, FOUR waves as 2(M) x 2(N), wave tile 128 x 128 = 8x8 MFMA tiles whose 256
// accumulator registers live in AGPRs; one wave per SIMD, 512 registers each.  A wave reads 16 fragments
// per 64 MFMAs (0.25 LDS fragment reads per MFMA, half of the 128x128 kernels above), which is what the
// measurements above said bounds them.  With a single wave per SIMD nothing hides a stall, so the K loop
// is one hand-ordered instruction stream (inline asm MFMAs / LDS reads, at most two other instructions
// between consecutive MFMAs):
//   tile kt sits in LDS stage s = kt & 1 (X image 32 KiB, then W image 32 KiB; same row images and XOR
//   swizzle as above), its k-substep-0 fragments are already in register set 0;
//   phase A (64 MFMAs, set 0): read X fragments of substep 1 -> set 1 | lgkmcnt(0) + barrier: every wave
//            is done with the X image of stage s | issue the 8 X DMA pieces of tile kt+2 into it,
//            alternating with the W fragment reads of substep 1 | lgkmcnt(0) + barrier: W image free |
//            first W DMA pieces of tile kt+2;
//   phase B (64 MFMAs, set 1): more W pieces | vmcnt(13): the 16 pieces of tile kt+1 (issued one
//            iteration ago) have landed | barrier | read substep-0 fragments of tile kt+1 from stage s^1
//            -> set 0, alternating with the last W pieces | lgkmcnt(0).
// Operands come through buffer_load ... lds with per-piece scalar offsets and two per-lane offsets (the
// swizzle depends on the piece parity only); rows past M / N and tiles past K read as zeros through
// num_records, so no clamping and no tail code.
#define V7_STAGE 65536
// two operand stages + 1 KiB per wave: the tile's bias values; + four more such slots for a residual that is a LayerNorm
// never written out (GemmArgs::r_mean): gamma and beta of the tile's 256 columns, mean and rstd of the wave's 16 MTN rows
#define V7_RLN_GAMMA (2 * V7_STAGE + 4096)
#define V7_RLN_BETA (2 * V7_STAGE + 8192)
#define V7_RLN_MEAN (2 * V7_STAGE + 12288)
#define V7_RLN_RSTD (2 * V7_STAGE + 16384)
#define V7_LDS_BYTES (2 * V7_STAGE + 4096 + 16384)
#define V7_WOFF 32768
// deferred-LayerNorm modes (GemmArgs::ln_mode): + 1 KiB per wave for the second per-column vector, + 8 KiB per wave-row
// half for the row statistics (8 slices x 128 rows x 8 B; shared by the two column waves of a row half)
#define V7_LN_COLV (2 * V7_STAGE + 4096)
#define V7_LN_STAT (2 * V7_STAGE + 8192)
#define V7_LN_ROWF (2 * V7_STAGE + 8192 + 16384)   // + 1 KiB per wave: the finished row factors of its 128 rows
#define V7_LDS_BYTES_LN (2 * V7_STAGE + 8192 + 16384 + 4096)
// MTN = row blocks (of 16) per wave: 8 (tile height 256) down to 4 (128).  The stream keeps its 64 slots per phase;
// with MTN < 8 the MFMAs of the missing row blocks (and their fragment reads) simply are not emitted.
#define V7_MFMA(S, i)                                                                                             \
  if (((i) & 7) < MTN)                                                                                            \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(i) & 7][(i) >> 3]) : "v"(wf[S][(i) >> 3]), \
               "v"(xf[S][(i) & 7]))
// first K-step of a tile: C = 0 (an inline constant), so the 256 accumulator registers need no zeroing pass
#define V7_MFMA0(S, i)                                                                                            \
  if (((i) & 7) < MTN)                                                                                            \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[(i) & 7][(i) >> 3]) : "v"(wf[S][(i) >> 3]), \
               "v"(xf[S][(i) & 7]))
#define V7_LDSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// The lane index computed on the spot (two VALU instructions): in the deferred-LayerNorm kernels every register counts, and
// a lane-derived loop invariant such as lane * 16 is one more value the allocator parks in scratch across the epilogue.
__device__ __forceinline__ int v7_lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// ---- epilogue of the 256x256-tile kernels (bf16 output, whole 64-column slabs: EPI_LDS) ---------------------------
// No compiler-visible VMEM loads and no conditional VMEM instructions: hipcc puts s_waitcnt vmcnt(0) at the joins
// behind conditional loads (bias / residual), and with one wave per SIMD every such wait drains the stores of the
// previous slab and the next tile's LDS-DMA (measured: 1.5 us per slab, 10-13 us per tile, against a 17 us K loop).
//   * bias: fetched with one LDS-DMA piece per wave into a private 1 KiB slot in the tile's first K-step (null
//     bias = zero-length descriptor = zeros), read back with ds_read in the epilogue;
//   * residual / GELU' factor R: buffer loads in inline asm, prefetched one slab ahead and retired by a counted
//     vmcnt (the stores of the current slab may stay in flight);
//   * C / C2: buffer stores from the LDS-transposed slab, unconditional (rows past M fall outside num_records).
__device__ __forceinline__ u32x4 v7_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void v7_buf_load16(u32x4& d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void v7_buf_load16_o16(u32x4& d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// V7_STORE_CC: cache-control suffix of the C stores (experiment builds: " nt", " sc1", ...; empty in the product)
#ifndef V7_STORE_CC
#define V7_STORE_CC ""
#endif
__device__ __forceinline__ void v7_buf_store16(u32x4 d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" V7_STORE_CC ::"v"(d), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void v7_buf_store16_o16(u32x4 d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:16" V7_STORE_CC ::"v"(d), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// slab = 16 rows x 64 columns of the 128x128 wave tile: accumulator row block MT, column half NH.  Spelled out 16
// times (about 50 instructions each): one wave per SIMD has nothing to hide a taken branch or an LDS round trip
// behind, so the epilogue is straight-line code with the stores going out directly from the accumulator layout
// (2 x 16 B per lane and row).
#define V7_SLAB(MT, NH)                                                                                   \
  {                                                                                                       \
    float v[16];                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                       \
      asm volatile("" : "+a"(acc[MT][4 * (NH) + t]));   /* stays in AGPRs until its slab's turn */         \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) v[4 * t + e] = acc[MT][4 * (NH) + t][e] + bv[4 * t + e]; \
    }                                                                                                     \
    const int so_row = 16 * MTN * wm + 16 * (MT);                                                         \
    if (has_c2) {   /* saved for the backward pass: the activation's derivative (GELU) or the pre-activation */ \
      float d2[16];                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 16; i += 4) {                                                 \
        if (ACT == ACT_GELU) {                                                                            \
          f32x4 gg, dd;                                                                                   \
          gelu_erf_both4((f32x4){v[i], v[i + 1], v[i + 2], v[i + 3]}, gg, dd);                            \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) { v[i + e] = gg[e]; d2[i + e] = dd[e]; }          \
        } else {                                                                                          \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) { d2[i + e] = v[i + e]; v[i + e] = apply_act<ACT>(v[i + e]); } \
        }                                                                                                 \
      }                                                                                                   \
      u32x4 p0, p1;                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        p0[i] = pack_bf16x2(d2[2 * i], d2[2 * i + 1]);                                                    \
        p1[i] = pack_bf16x2(d2[8 + 2 * i], d2[8 + 2 * i + 1]);                                            \
      }                                                                                                   \
      v7_buf_store16(p0, rs_c2, vo_c2, so_row * ldc2_b + ec * 2);                                         \
      v7_buf_store16_o16(p1, rs_c2, vo_c2, so_row * ldc2_b + ec * 2);                                     \
    } else if (ACT == ACT_GELU) {   /* inference (nothing saved for a backward): the transcendental-free form */ \
      _Pragma("unroll") for (int i = 0; i < 16; i += 4) {                                                 \
        const f32x4 gg = gelu_poly4((f32x4){v[i], v[i + 1], v[i + 2], v[i + 3]});                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) v[i + e] = gg[e];                                   \
      }                                                                                                   \
    } else {                                                                                              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) v[i] = apply_act<ACT>(v[i]);                         \
    }                                                                                                     \
    if (g.drop.thresh) {                                                                                  \
      const uint32_t e0 = (uint32_t)(m0 + so_row + j) * (uint32_t)g.N + (uint32_t)(ec + 16 * gq);         \
      vt_drop_run<16>(g.drop, e0, v);                                                                     \
    }                                                                                                     \
    if (HAS_R) {                                                                                          \
      /* residual = LayerNorm(R): this slab's gamma / beta (16 columns) and the row's mean / rstd come out of the wave's  \
         LDS slots (V7_DMA_BIAS parked them), issued here -- in front of the ring's wait, which hides their round trip --   \
         and awaited where they are used (held across slabs they cost 48 registers: the 256-row kernel spilled; issued  \
         at the slab's top they overlapped the dropout hash's temporaries: it spilled again) */                            \
      u32x4 lg_[4], lb_[4];                                                                               \
      u32x2 ls_;                                                                                          \
      if (ACT != ACT_MUL && r_ln) {                                                                       \
        const unsigned co_ = lds0 + wave * 1024 + 4 * (128 * wn + 64 * (NH) + 16 * gq);                   \
        const unsigned ro_ = lds0 + wave * 1024 + 4 * (16 * (MT) + j);                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
          asm volatile("ds_read_b128 %0, %1" : "=v"(lg_[i]) : "v"(co_ + V7_RLN_GAMMA + 16 * i));          \
          asm volatile("ds_read_b128 %0, %1" : "=v"(lb_[i]) : "v"(co_ + V7_RLN_BETA + 16 * i));           \
        }                                                                                                 \
        asm volatile("ds_read_b32 %0, %1" : "=v"(ls_[0]) : "v"(ro_ + V7_RLN_MEAN));                       \
        asm volatile("ds_read_b32 %0, %1" : "=v"(ls_[1]) : "v"(ro_ + V7_RLN_RSTD));                       \
      }                                                                                                   \
      /* slab s = 8*NH + MT; its residual was issued 8 slabs ago (the first eight before slab 0: a 4-deep ring left \
         each slab waiting ~0.45 us on HBM latency).  Behind it in the queue: the younger residual loads and the    \
         stores issued since -> counted wait (VMEM retires in order).  With C2 stores in the stream as well (not a  \
         combination the encoder uses) simply drain. */                                                            \
      if (has_c2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NH) == 0 ? 2 * (MTN - 1) + 2 * (MT) : 4 * MTN - 2 - 2 * (MT)) : "memory"); \
      asm volatile("" : "+v"(rq[MT][0]), "+v"(rq[MT][1]));                                                \
      if (ACT != ACT_MUL && r_ln) {   /* the residual is LayerNorm(row of fp16 sums): (x - mean) rstd gamma + beta */ \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        asm volatile("" : "+v"(lg_[0]), "+v"(lg_[1]), "+v"(lg_[2]), "+v"(lg_[3]));                        \
        asm volatile("" : "+v"(lb_[0]), "+v"(lb_[1]), "+v"(lb_[2]), "+v"(lb_[3]), "+v"(ls_));             \
        const float mu_ = __uint_as_float(ls_[0]), rs_ = __uint_as_float(ls_[1]);                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
          const u32x4 q0 = rq[MT][0], q1 = rq[MT][1];                                                     \
          /* v[k] <-> column k of the lane's 16: gamma / beta word k = lg_[k >> 2][k & 3] */               \
          v[2 * i] += fmaf((f16lo(q0[i]) - mu_) * rs_, __uint_as_float(lg_[i >> 1][(2 * i) & 3]), __uint_as_float(lb_[i >> 1][(2 * i) & 3])); \
          v[2 * i + 1] += fmaf((f16hi(q0[i]) - mu_) * rs_, __uint_as_float(lg_[i >> 1][(2 * i + 1) & 3]), __uint_as_float(lb_[i >> 1][(2 * i + 1) & 3])); \
          v[8 + 2 * i] += fmaf((f16lo(q1[i]) - mu_) * rs_, __uint_as_float(lg_[2 + (i >> 1)][(2 * i) & 3]), __uint_as_float(lb_[2 + (i >> 1)][(2 * i) & 3])); \
          v[8 + 2 * i + 1] += fmaf((f16hi(q1[i]) - mu_) * rs_, __uint_as_float(lg_[2 + (i >> 1)][(2 * i + 1) & 3]), __uint_as_float(lb_[2 + (i >> 1)][(2 * i + 1) & 3])); \
        }                                                                                                 \
      } else {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        const u32x4 q0 = rq[MT][0], q1 = rq[MT][1];                                                       \
        const float r0 = r_f16 ? f16lo(q0[i]) : bf16lo(q0[i]), r1 = r_f16 ? f16hi(q0[i]) : bf16hi(q0[i]); \
        const float r2 = r_f16 ? f16lo(q1[i]) : bf16lo(q1[i]), r3 = r_f16 ? f16hi(q1[i]) : bf16hi(q1[i]); \
        if (ACT == ACT_MUL) { v[2 * i] *= r0; v[2 * i + 1] *= r1; v[8 + 2 * i] *= r2; v[8 + 2 * i + 1] *= r3; } \
        else { v[2 * i] += r0; v[2 * i + 1] += r1; v[8 + 2 * i] += r2; v[8 + 2 * i + 1] += r3; }          \
      }                                                                                                   \
      }                                                                                                   \
      if ((NH) == 0) {   /* same rows of the other column half; issued unconditionally: the counts above rely on it \
                            (columns past N are never used, rows past M read zeros) */                              \
        v7_buf_load16(rq[MT][0], rs_r, vo_r, so_row * ldr_b + (ec + 64) * 2);                             \
        v7_buf_load16_o16(rq[MT][1], rs_r, vo_r, so_row * ldr_b + (ec + 64) * 2);                         \
      }                                                                                                   \
    }                                                                                                     \
    u32x4 o0, o1;                                                                                         \
    if (c_f16) {   /* the pre-LayerNorm sum of the training layer as fp16 (wave-uniform branch) */           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        o0[i] = pack_f16x2(v[2 * i], v[2 * i + 1]);                                                       \
        o1[i] = pack_f16x2(v[8 + 2 * i], v[8 + 2 * i + 1]);                                               \
      }                                                                                                   \
    } else {                                                                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        o0[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);                                                      \
        o1[i] = pack_bf16x2(v[8 + 2 * i], v[8 + 2 * i + 1]);                                              \
      }                                                                                                   \
    }                                                                                                     \
    v7_buf_store16(o0, rs_c, vo_c, so_row * ldc_b + ec * 2);                                              \
    v7_buf_store16_o16(o1, rs_c, vo_c, so_row * ldc_b + ec * 2);                                          \
  }

// RLN: the instantiation can rebuild a LayerNorm residual (GemmArgs::r_mean).  Not the persistent kernel's 256-row tile: it
// carries the next tile's 64 fragment registers through its epilogue and the 34 transient registers of the rebuilt residual
// spilled there (tests/test_build_isa.py) -- the host gives such a GEMM the 224-row tile instead (launch_v8).
template <int ACT, bool HAS_R, int MTN, bool RLN = true>
__device__ __forceinline__ void v7_epilogue_fast(const GemmArgs& g, f32x4 (&acc)[8][8], int lane, int wave, int m0, int n0, unsigned lds0) {
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned bias_slot = lds0 + 2 * V7_STAGE + wave * 1024;   // this wave's copy of the tile's 256 bias values
  const int rows = g.M - m0 < 32 * MTN ? g.M - m0 : 32 * MTN;
  const int ldc_b = (int)g.ldc * 2, ldc2_b = (int)g.ldc2 * 2, ldr_b = (int)g.ldr * 2;
  // rows past M fall outside num_records: their loads read zeros, their stores are dropped
#ifdef V7_LAB_DROP_STORES   // experiment builds only: zero-length descriptors, every C / C2 store is issued and dropped
  const u32x4 rs_c = v7_rsrc((const bf16_t*)g.C + (long)m0 * g.ldc, 0u);
  const u32x4 rs_c2 = v7_rsrc(g.C2 ? g.C2 + (long)m0 * g.ldc2 : nullptr, 0u);
#else
  const u32x4 rs_c = v7_rsrc((const bf16_t*)g.C + (long)m0 * g.ldc, (unsigned)rows * ldc_b);
  const u32x4 rs_c2 = v7_rsrc(g.C2 ? g.C2 + (long)m0 * g.ldc2 : nullptr, g.C2 ? (unsigned)rows * ldc2_b : 0u);
#endif
  const u32x4 rs_r = v7_rsrc(g.R ? g.R + (long)m0 * g.ldr : nullptr, g.R ? (unsigned)rows * ldr_b : 0u);
  const int gq = lane >> 4, j = lane & 15;
  const int vo_c = j * ldc_b + gq * 32, vo_c2 = j * ldc2_b + gq * 32, vo_r = j * ldr_b + gq * 32;
  const bool has_c2 = g.C2 != nullptr;
  const bool r_f16 = g.r_f16 != 0, c_f16 = g.c_f16 != 0;
  const bool r_ln = RLN && HAS_R && g.r_mean != nullptr;   // residual = LayerNorm(R) (GemmArgs::r_mean); vectors parked in LDS by V7_DMA_BIAS
  u32x4 rq[8][2];   // residual ring: the next eight slabs (one column half), two 8-column halves each
  // the two 64-column halves spelled out: a rolled (or not fully unrolled) loop would index the accumulators
  // dynamically and demote them to scratch
#define V7_HALF(NH)                                                                                       \
  {                                                                                                       \
    const int ecr = 128 * wn + 64 * (NH);   /* tile-relative first column */                              \
    const int ec = n0 + ecr;                                                                              \
    if (ec < g.N) {   /* N % 64 == 0 (host-checked): the 64 columns are all valid */                      \
      if (HAS_R && (NH) == 0) {                                                                           \
        _Pragma("unroll") for (int q = 0; q < MTN; ++q) {                                                 \
          v7_buf_load16(rq[q][0], rs_r, vo_r, (16 * MTN * wm + 16 * q) * ldr_b + ec * 2);                 \
          v7_buf_load16_o16(rq[q][1], rs_r, vo_r, (16 * MTN * wm + 16 * q) * ldr_b + ec * 2);             \
        }                                                                                                 \
      }                                                                                                   \
      u32x4 bq[4];                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        const unsigned ba = bias_slot + 4 * (ecr + 16 * gq) + 16 * i;                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(bq[i]) : "v"(ba));                                      \
      }                                                                                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
      float bv[16];                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        asm volatile("" : "+v"(bq[i]));                                                                   \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) bv[4 * i + e] = __uint_as_float(bq[i][e]);          \
      }                                                                                                   \
      V7_SLAB(0, NH) V7_SLAB(1, NH) V7_SLAB(2, NH) V7_SLAB(3, NH)                                         \
      if (MTN > 4) V7_SLAB(4, NH)                                                                         \
      if (MTN > 5) V7_SLAB(5, NH)                                                                         \
      if (MTN > 6) V7_SLAB(6, NH)                                                                         \
      if (MTN > 7) V7_SLAB(7, NH)                                                                         \
    } else if (HAS_R && (NH) == 1) {                                                                      \
      /* This half lies past N but half 0 issued its ring loads all the same (the counted waits rely on it): they must  \
         land before the ring's registers mean anything else.  Unwaited, the youngest one (rq[MTN-1][1] = v[62:65] in  \
         the 224-row kernel) was still in flight when the next tile's cursor_tile() used v62 as the scratch register of \
         its uniform division (v_cvt / v_rcp / v_readfirstlane): a garbage quotient = a garbage tile origin = an operand \
         descriptor based outside the allocation -> memory fault, process abort (round 5's test2.log; K = 128 puts the  \
         division right behind the epilogue).  The pins keep the registers allocated to the ring up to the wait. */     \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
      _Pragma("unroll") for (int q = 0; q < MTN; ++q) asm volatile("" : "+v"(rq[q][0]), "+v"(rq[q][1]));  \
    }                                                                                                     \
  }
  V7_HALF(0)
  V7_HALF(1)
}

#include "gemm_v7_ln_epilogue.hpp"

// slab h of the plain register epilogue (fp32 output, N not a multiple of 64, row remap): a switch moves the slab's
// 16 accumulator registers to VGPRs (AGPRs cannot be indexed dynamically) inside a rolled loop
#define V7_SLAB_CASE(h)                                                                                   \
  case h:                                                                                                 \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                       \
      asm volatile("" : "+a"(acc[(h) >> 1][4 * ((h) & 1) + t]));                                          \
      a[t] = acc[(h) >> 1][4 * ((h) & 1) + t];                                                            \
    }                                                                                                     \
    break;
#define V7_SLAB_SWITCH(h)                                                                                 \
  switch (h) {                                                                                            \
    V7_SLAB_CASE(0) V7_SLAB_CASE(1) V7_SLAB_CASE(2) V7_SLAB_CASE(3)                                       \
    V7_SLAB_CASE(4) V7_SLAB_CASE(5) V7_SLAB_CASE(6) V7_SLAB_CASE(7)                                       \
    V7_SLAB_CASE(8) V7_SLAB_CASE(9) V7_SLAB_CASE(10) V7_SLAB_CASE(11)                                     \
    V7_SLAB_CASE(12) V7_SLAB_CASE(13) V7_SLAB_CASE(14) V7_SLAB_CASE(15)                                   \
  }

// bias of the tile's 256 columns -> this wave's LDS slot, as one more LDS-DMA piece (null bias: zero-length)
#define V7_DMA_BIAS()                                                                                     \
  if (EPI_LDS) {                                                                                          \
    const int bn_ = g.N - n0 < 256 ? g.N - n0 : 256;                                                      \
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bias ? g.bias + n0 : nullptr), 0, g.bias ? bn_ * 4 : 0, 0x00020000); \
    const int l16_ = LNM ? v7_lane_now() * 16 : lane * 16;                                                \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(smem + 2 * V7_STAGE + wave * 1024), 16, l16_, 0, 0, 0); \
    if (LNM) {   /* the second per-column vector of the deferred-LayerNorm modes (g / gamma) */                 \
      __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(g.colv ? g.colv + n0 : nullptr), 0, g.colv ? bn_ * 4 : 0, 0x00020000); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, LDS_PTR(smem + V7_LN_COLV + wave * 1024), 16, l16_, 0, 0, 0); \
    }                                                                                                     \
    if (LNM == 0 && HAS_R && V7_RLN_OK && g.r_mean) {   /* residual = LayerNorm(R): its per-column and per-row vectors (wave-uniform) */ \
      const int r0_ = m0 + 16 * MTN * wm;                                                                 \
      const int nr_ = g.M - r0_ < 16 * MTN ? (g.M - r0_ < 0 ? 0 : g.M - r0_) : 16 * MTN;                  \
      __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc((void*)(g.r_gamma + n0), 0, bn_ * 4, 0x00020000);   \
      __amdgpu_buffer_rsrc_t rbt_ = __builtin_amdgcn_make_buffer_rsrc((void*)(g.r_beta + n0), 0, bn_ * 4, 0x00020000);   \
      __amdgpu_buffer_rsrc_t rm_ = __builtin_amdgcn_make_buffer_rsrc((void*)(g.r_mean + r0_), 0, nr_ * 4, 0x00020000);   \
      __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void*)(g.r_rstd + r0_), 0, nr_ * 4, 0x00020000);   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rg_, LDS_PTR(smem + V7_RLN_GAMMA + wave * 1024), 16, l16_, 0, 0, 0);      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rbt_, LDS_PTR(smem + V7_RLN_BETA + wave * 1024), 16, l16_, 0, 0, 0);      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rm_, LDS_PTR(smem + V7_RLN_MEAN + wave * 1024), 16, l16_, 0, 0, 0);       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr_, LDS_PTR(smem + V7_RLN_RSTD + wave * 1024), 16, l16_, 0, 0, 0);       \
    }                                                                                                     \
  }
// slice p = 2 i + (column wave) of the row statistics of this wave's row half -> the half's LDS slot (i = 0 .. 3; slices
// past ln_np and rows past M through zero-length descriptors: zeros)
#define V7_DMA_STAT(i)                                                                                    \
  {                                                                                                       \
    const int p_ = 2 * (i) + wn;                                                                          \
    const int r0_ = m0 + 16 * MTN * wm;                                                                   \
    const int nr_ = g.M - r0_ < 128 ? g.M - r0_ : 128;                                                    \
    const bool ok_ = p_ < g.ln_np && nr_ > 0;                                                             \
    __amdgpu_buffer_rsrc_t rt_ = __builtin_amdgcn_make_buffer_rsrc(                                       \
        (void*)(ok_ ? g.ln_stats + ((long)p_ * g.ln_rows + r0_) * 2 : nullptr), 0, ok_ ? nr_ * 8 : 0, 0x00020000); \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rt_, LDS_PTR(smem + V7_LN_STAT + wm * 8192 + p_ * 1024), 16, stat_l16, 0, 0, 0); \
  }

// One K-step: the current stage holds the K-tile whose substep-0 fragments are in set 0; (rx, rw) describe the
// K-tile two steps ahead.  VMW is the vmcnt that proves the NEXT K-tile has landed: 13 in steady state (the 13
// pieces issued so far in this step may stay in flight), 0 right after an epilogue (its stores share the counter).
#define V7_NOHOOK(i)
#define V7_STEP(VMW) V7_STEP_(VMW, V7_MFMA, V7_NOHOOK)
#define V7_STEP_FIRST(VMW) V7_STEP_(VMW, V7_MFMA0, V7_NOHOOK)   /* first K-step of a tile: phase A starts from C = 0 */
/* ... of a deferred-LayerNorm tile: four more LDS-DMA pieces (row statistics) behind the step's first barrier, where every
   wave has left the previous tile's epilogue (the statistics slot is shared by two waves); they are younger than the
   next K-tile's pieces, so the landing wait counts them in */
#define V7_STEP_FIRST_LN() const int stat_l16 = v7_lane_now() * 16;   /* no VALU inside the MFMA stream */ \
  V7_STEP_(17, V7_MFMA0, V7_STAT_HOOK)
#define V7_STAT_HOOK(i) if ((i) >= 40 && (i) < 48 && !((i) & 1)) V7_DMA_STAT(((i) - 40) >> 1);
/* (tools/experiments/kstep_lab.hip pre-defines V7_STEP_ with its timing-only ablations of this stream) */
#ifndef V7_STEP_
#define V7_STEP_(VMW, MFMA_A, HOOK)                                                                               \
  {                                                                                                        \
    const unsigned xn0 = xa0 ^ V7_STAGE, wn0 = wa0 ^ V7_STAGE; /* substep-0 addresses of the other stage */ \
    /* re-defined every step: as plain loop invariants the allocator parks them in scratch (reload + vmcnt(0)) */ \
    asm volatile("" : "+v"(vx[0]), "+v"(vx[1]), "+v"(vw[0]), "+v"(vw[1]));                                \
    _Pragma("unroll") for (int i = 0; i < 64; ++i) {                                                      \
      MFMA_A(0, i);                                                                                        \
      if (i < 16 && (i & 1) && (i >> 1) < MTN) V7_LDSR(xf[1][i >> 1], xa1, (i >> 1) * 2048);               \
      if (i == 20) {                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                      \
      }                                                                                                    \
      if (i >= 22 && i < 38 && !(i & 1)) V7_DMA_X(rx, dst, (i - 22) >> 1);                                 \
      if (i >= 22 && i < 38 && (i & 1)) V7_LDSR(wf[1][(i - 22) >> 1], wa1, ((i - 22) >> 1) * 2048);        \
      if (i == 50) {                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                      \
      }                                                                                                    \
      if (i >= 52 && !(i & 3)) V7_DMA_W(rw, dst, (i - 52) >> 2);                                           \
      HOOK(i)                                                                                              \
    }                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 64; ++i) {                                                      \
      V7_MFMA(1, i);                                                                                       \
      if (i == 4) V7_DMA_W(rw, dst, 3);                                                                    \
      if (i == 10) V7_DMA_W(rw, dst, 4);                                                                   \
      if (i == 24) {                                                                                       \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");                                         \
        __builtin_amdgcn_s_barrier();                                                                      \
      }                                                                                                    \
      if (i >= 26 && i < 58 && !(i & 1)) {                                                                 \
        const int j = (i - 26) >> 1;                                                                       \
        if (j < 8) { if (j < MTN) V7_LDSR(xf[0][j], xn0, j * 2048); }                                      \
        else V7_LDSR(wf[0][j - 8], wn0, (j - 8) * 2048);                                                   \
      }                                                                                                    \
      if (i == 31) V7_DMA_W(rw, dst, 5);                                                                   \
      if (i == 39) V7_DMA_W(rw, dst, 6);                                                                   \
      if (i == 47) V7_DMA_W(rw, dst, 7);                                                                   \
    }                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    xa0 ^= V7_STAGE; xa1 ^= V7_STAGE; wa0 ^= V7_STAGE; wa1 ^= V7_STAGE; dst ^= V7_STAGE;                   \
  }
#endif

#ifndef V7_LAB_DECLS
#define V7_LAB_DECLS
#define V7_LAB_EXIT
#endif
#define V7_TR nullptr
#define V7_RING_PRE (LNM == 2)
template <int ACT, bool OUT_F32, bool EPI_LDS, bool HAS_R, int MTN = 8, int LNM = 0>
__global__ __launch_bounds__(256, 1) void gemm_nt_bf16_v7(GemmArgs g) {
  constexpr bool V7_RLN_OK = true;
  constexpr int TH = 32 * MTN;   // tile height (see gemm_nt_bf16_v8): 256, or 224 / 192 to fill one round better
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-contiguous chunk of the grouped tile order (bands of 8 row-tiles, column-tile major inside)
  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  int t_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  // split-K (GemmArgs::ksplit): the grid holds ksplit copies of the tile list, copy s takes its share of the K-steps
  int ks_k0 = 0, ks_nk = g.K >> 6;
  if (g.ksplit > 1) {
    const int ntile = g.tiles_m * g.tiles_n;
    const int split = t_id / ntile;
    t_id -= split * ntile;
    const int kq = ((g.K >> 6) + g.ksplit - 1) / g.ksplit;
    ks_k0 = split * kq;
    ks_nk = (g.K >> 6) - ks_k0 < kq ? (g.K >> 6) - ks_k0 : kq;
    if (ks_nk < 0) ks_nk = 0;
    g.C = (float*)g.C + (long)split * g.c_plane;
  }
  const int band_tiles = 8 * g.tiles_n;
  const int band = t_id / band_tiles;
  const int within = t_id - band * band_tiles;
  const int rows_left = g.tiles_m - band * 8;
  const int band_h = rows_left < 8 ? rows_left : 8;
  const int bn = within / band_h;
  const int bm = band * 8 + (within - bn * band_h);
  const int m0 = bm * TH, n0 = bn * 256;

  // ---- DMA addressing.  Piece p = wave*8 + i covers LDS rows 8p .. 8p+7 of an operand image (128 B per row);
  // lane -> row 8p + (lane>>3), 16-B chunk (lane&7) ^ swz(row), swz(row) = (4*(i&1) + (lane>>4)) & 7.
  const int rows_x = g.M - m0 < TH ? g.M - m0 : TH;
  const int rows_w = g.N - n0 < 256 ? g.N - n0 : 256;
  const bf16_t* xbase = g.A + (long)m0 * g.lda + 64 * ks_k0;
  const bf16_t* wbase = g.W + (long)n0 * g.ldw + 64 * ks_k0;
  // bytes of the tile's row panel that may be read: full rows except the last one, which holds this workgroup's K elements
  const unsigned xbytes = (unsigned)(((long)(rows_x - 1) * g.lda + 64L * ks_nk) * 2);
  const unsigned wbytes = (unsigned)(((long)(rows_w - 1) * g.ldw + 64L * ks_nk) * 2);
  int vx[2], vw[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int c = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
    vx[par] = (lane >> 3) * (int)g.lda * 2 + c * 16;
    // W rows are permuted inside each 64-row block so that a lane's 4 N-subtiles interleave to 16
    // consecutive output columns: image row r <- W row 16*((r>>2)&3) + 4*((r>>4)&3) + (r&3)
    vw[par] = (16 * (lane >> 5) + ((lane >> 3) & 3)) * (int)g.ldw * 2 + c * 16;
  }
  int sx[8], sw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sx[i] = (wave * 8 + i) * 8 * (int)g.lda * 2;
    sw[i] = (64 * wave + 32 * (i & 1) + 4 * (i >> 1)) * (int)g.ldw * 2;
  }
  const int nk = ks_nk;

  // ---- fragment addresses: X image row 128*wm + 16*mt + (lane&15), W image row 128*wn + 16*nt + (lane&15);
  // chunk (lane>>4) ^ swz(row) for k-substep 0, the same ^ 4 (address ^ 64) for substep 1
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned fr = (lane & 15) * 128 + ((((unsigned)lane >> 4) ^ (((unsigned)lane & 15) >> 1)) << 4);
  // current-stage addresses of the two k-substeps; toggled (^ V7_STAGE) every K-step.  They are loop-carried on
  // purpose: as long-lived loop invariants the register allocator would park them in scratch.
  unsigned xa0 = lds0 + wm * (MTN * 2048) + fr, xa1 = xa0 ^ 64;
  unsigned wa0 = lds0 + V7_WOFF + wn * 16384 + fr, wa1 = wa0 ^ 64;
  unsigned dst = wave * 8192;   // byte offset of this wave's first DMA piece inside the current stage's X image

  f32x4 acc[8][8];   // defined by the first K-step (C = 0)
  u32x4 xf[2][8], wf[2][8];
  V7_LAB_DECLS

  auto rsrc_x = [&](int kt) {
    const unsigned kb = (unsigned)kt * 128u;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)xbase + kb), 0, kt < nk ? (int)(xbytes - kb) : 0, 0x00020000);
  };
  auto rsrc_w = [&](int kt) {
    const unsigned kb = (unsigned)kt * 128u;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)wbase + kb), 0, kt < nk ? (int)(wbytes - kb) : 0, 0x00020000);
  };
#define V7_DMA_X(rs, d, i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (d) + (i) * 1024), 16, vx[(i) & 1], sx[i], 0, 0)
#define V7_DMA_W(rs, d, i) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (d) + V7_WOFF + (i) * 1024), 16, vw[(i) & 1], sw[i], 0, 0)

  // deferred-LayerNorm mode 2: the first ring of residual-stream slabs goes out before anything else (see
  // v7_ln_ring_prefetch); the prologue's operand wait below covers them
  u32x4 ln_ring[V7_LN_RING][2];
  if constexpr (LNM == 2) v7_ln_ring_prefetch<LNM, MTN>(g, ln_ring, wave, m0, n0);

  // prologue: tiles 0 and 1 in flight, substep-0 fragments of tile 0 in set 0
  {
    __amdgpu_buffer_rsrc_t rx0 = rsrc_x(0), rw0 = rsrc_w(0), rx1 = rsrc_x(1), rw1 = rsrc_w(1);
    V7_DMA_BIAS()
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_X(rx0, dst, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_W(rw0, dst, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_X(rx1, dst + V7_STAGE, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_W(rw1, dst + V7_STAGE, i);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (LNM == 2) {   // the ring has landed (older than the 16 pieces still in flight): its registers are final from here
#pragma unroll
      for (int i = 0; i < V7_LN_RING; ++i) V7_LN_PIN("+v"(ln_ring[i][0]), "+v"(ln_ring[i][1]));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < MTN) V7_LDSR(xf[0][i], xa0, i * 2048);
      V7_LDSR(wf[0][i], wa0, i * 2048);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // one K-step per iteration: tile kt in the current stage; MFMA i = 8*nt + mt
  {
    __amdgpu_buffer_rsrc_t rx = rsrc_x(2), rw = rsrc_w(2);
    if constexpr (LNM != 0) { V7_STEP_FIRST_LN() } else { V7_STEP_FIRST(13) }
  }
  for (int kt = 1; kt < nk; ++kt) {
    __amdgpu_buffer_rsrc_t rx = rsrc_x(kt + 2), rw = rsrc_w(kt + 2);
    V7_STEP(13)
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // epilogue: EPI_LDS -> v7_epilogue_lds (above); otherwise (fp32 output, N not a multiple of 64, row remap) the
  // plain register epilogue, slab by slab
#define V7_EPILOGUE()                                                                                     \
  if constexpr (LNM != 0) {                                                                               \
    v7_epilogue_ln<ACT, LNM, MTN, V7_RING_PRE>(g, acc, lane, wave, m0, n0, lds0, ln_ring);                \
  } else if (EPI_LDS) {                                                                                   \
    v7_epilogue_fast<ACT, HAS_R, MTN, V7_RLN_OK>(g, acc, lane, wave, m0, n0, lds0);                       \
  } else {                                                                                                \
    _Pragma("unroll 1") for (int h = 0; h < 16; ++h) {                                                    \
      f32x4 a[4];                                                                                         \
      V7_SLAB_SWITCH(h)                                                                                   \
      const int er0 = m0 + 128 * wm + 16 * (h >> 1), ec0 = n0 + 128 * wn + 64 * (h & 1);                  \
      const int nb = ec0 + 16 * (lane >> 4);                                                              \
      if (nb < g.N) {                                                                                     \
        const bool full = nb + 16 <= g.N;                                                                 \
        float bv[16];                                                                                     \
        epi_load_bias(g, nb, full, bv);                                                                   \
        epi_row_direct<ACT, OUT_F32>(g, a, bv, er0 + (lane & 15), nb, full);                              \
      }                                                                                                   \
    }                                                                                                     \
  }
  // split-K with raw partial tiles (variant 33, GemmArgs::sk_ws set): copy `split` of tile (bm, bn) leaves its accumulators
  // as they sit in the registers -- AGPR-sourced 16-byte stores, 256 KiB per workgroup -- in slot split * tiles + bm * tiles_n +
  // bn of the workspace; splitk_tiles_epilogue (gemm_v7.hip) sums the slots in order and runs the register epilogue.  Only
  // the instantiation the launcher names carries the code.
  if constexpr (EPI_LDS && LNM == 0 && MTN == 8 && !HAS_R && ACT == ACT_NONE && !OUT_F32) {
    if (g.ksplit > 1 && g.sk_ws) {
      const int ntile = g.tiles_m * g.tiles_n;
      const int split = ks_k0 / (((g.K >> 6) + g.ksplit - 1) / g.ksplit);
      const u32x4 rs = v7_rsrc((const char*)g.sk_ws + ((long)split * ntile + (long)bm * g.tiles_n + bn) * V8_SK_PART_BYTES,
                               (unsigned)V8_SK_PART_BYTES);
      const int vo = tid * 16;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"a"(acc[mt][nt]), "v"(vo), "s"(rs), "s"((8 * mt + nt) * 4096) : "memory");
      return;
    }
  }
#ifdef V7_DIAG
  {
    float t = 0.f;
    _Pragma("unroll") for (int i = 0; i < 8; ++i) _Pragma("unroll") for (int j = 0; j < 8; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    ((float*)g.C)[blockIdx.x * 256 + tid] = t;
  }
#else
  V7_EPILOGUE()
#endif
}


// ================================================================================================
// v8: v7 made persistent.  One workgroup per CU walks its share of the output tiles and the K-step pipeline
// runs straight across tile boundaries: while tile t is in its last two K-steps the DMA already fetches the
// first two K-tiles of tile t+1, so the epilogue of t is the only time the MFMA pipes idle (v7 pays launch +
// first-fetch latency per tile on top: ~11 us of a ~28 us tile at K = 768).  Each XCD owns a contiguous
// chunk of the grouped tile order; its workgroups take tiles of the chunk round-robin, so the tiles in flight
// on one L2 are neighbours.
#undef V7_TR
#define V7_TR tr
#undef V7_RING_PRE
#define V7_RING_PRE false
// MTN (row blocks of 16 per wave, 8 .. 4 = tile height TH 256 .. 128): with ~200 row tiles of 256 the
// three column tiles of the N = 768 shapes make 600 tiles = 2.34 rounds on 256 CUs, paid as 3; 224-row tiles make 681 =
// 2.66 rounds of tiles that are 7/8 the work -- the same 3 rounds, 12.5 % fewer MFMAs.  The host picks per shape.
template <int ACT, bool OUT_F32, bool EPI_LDS, bool HAS_R, int MTN = 8, int LNM = 0>
__global__ __launch_bounds__(256, 1) void gemm_nt_bf16_v8(GemmArgs g) {
  constexpr bool V7_RLN_OK = MTN < 8;   // see v7_epilogue_fast
  constexpr int TH = 32 * MTN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int T = g.tiles_m * g.tiles_n;
  const int nwg = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7;
  const int nx = (nwg - xcd + 7) >> 3;                 // workgroups on this XCD
  // The XCD's chunk of the tile order is sized by its SHARE OF THE WORKGROUPS (w0 = workgroups on the XCDs before this
  // one), not as T / 8: with nwg = T = 195 (M = 8 208 on 128-row tiles) equal chunks of 24 / 25 tiles met 25 / 24
  // workgroups in the other order on three XCDs, and one workgroup there walked two tiles -- the whole launch took two
  // rounds (round 6: 79 -> 45 us at [8 208, 768] x 3 072).
  const int wq = nwg >> 3, wr = nwg & 7;
  const int w0 = xcd * wq + (xcd < wr ? xcd : wr);
  const int c0 = (int)((long)T * w0 / nwg), c1 = (int)((long)T * (w0 + nx) / nwg);
  const int iw = b >> 3;                               // this workgroup's index on its XCD
  const int band_tiles = 8 * g.tiles_n;
  const int nk = g.K >> 6;

  // ---- work units.  The XCD's chunk holds rounds * nx + Rx tiles.  Plain: unit j of workgroup iw is the whole tile
  // c0 + iw + j * nx (the Rx left-over tiles cost a last round with nx - Rx workgroups idle).  STREAM-K REGION
  // (GemmArgs::sk_parts > 1, round 6): with Rx > 0 the last rounds' tiles -- nx + Rx of them, or all Rx when the chunk is
  // shorter than one round -- are laid end to end as one sequence of Tsk * nk K-steps and cut into nx contiguous shares
  // [skb(i), skb(i + 1)); a workgroup's share starts inside a tile (a TAIL part: computed first, its fp32 accumulators go
  // to the workspace), runs over whole tiles, and ends inside a tile (a HEAD part, k = 0 ..: computed last; it adds the
  // tile's other parts -- stored long before, or at this very moment by workgroups whose whole share lies inside the tile
  // -- in workgroup order and runs the ordinary epilogue).  Every workgroup works skb(iw+1) - skb(iw) K-steps whatever
  // T mod grid is, stores at most one partial tile and finishes at most one.  Cut points closer than three K-steps to a
  // tile boundary snap to it.
  const int Tx = c1 - c0;
  const int rounds = Tx / nx, Rx = Tx - rounds * nx;
  // (not in the register-epilogue instantiations -- fp32 output, N % 64 != 0, row remap: the extra code sent their
  // accumulators to scratch; the host leaves sk_parts at 0 for them)
  // Nor on the 192- .. 256-row tiles: with 192 .. 256 accumulator registers and every VGPR spoken for, hipcc's allocator
  // gives the accumulators OTHER AGPR tuples inside the fix-up region than in the K loop (a live-range split around the
  // region) and moves all of them through VGPRs and scratch at its borders -- measured in the ISA, whatever the form of the
  // fix-up (tied operands, one statement per tile, do-while, break after the head).  160- and 128-row tiles compile clean,
  // and they are the tiles small batches use.
  constexpr bool SK_OK = (EPI_LDS || LNM != 0) && MTN <= 5;
  // (32-bit arithmetic throughout: Tsk <= 2 nx tiles; every share at least six K-steps, so that the snapped cut points
  // leave no share empty)
  const int rounds_sk = rounds > 0 ? rounds - 1 : 0;
  const int Wsk0 = (Tx - rounds_sk * nx) * nk;
  const bool sk = SK_OK && g.sk_parts > 1 && Rx > 0 && nx <= V8_SK_WGS_PER_XCD && Wsk0 >= 6 * nx && Wsk0 < (1 << 24);
  const int rounds_dp = sk ? rounds_sk : rounds;
  const int Tsk = sk ? Tx - rounds_dp * nx : 0;          // tiles of the stream-K region
  const int Wsk = Tsk * nk;                              // ... and its K-steps
  auto skb = [&](int i) {                                // first K-step of workgroup i's share (i = nx: the end)
    int p = (int)((unsigned)(Wsk * i) / (unsigned)nx);
    const int k = (int)((unsigned)p % (unsigned)nk);
    if (k < 3) p -= k;
    else if (nk - k < 3) p += nk - k;
    return p;
  };
  const int sk_lo = sk ? skb(iw) : 0, sk_hi = sk ? skb(iw + 1) : 0;
  const int sk_t0 = sk_lo / nk;                          // first region tile of the share
  const int n_sk = sk_hi > sk_lo ? (sk_hi - 1) / nk - sk_t0 + 1 : 0;
  const int n_units = rounds_dp + (sk ? n_sk : (iw < Rx ? 1 : 0));
  // unit j -> tile, first K-step, K-steps; role: 0 whole tile, 1 producer (the tile's head lies in another share), 2 head
  auto unit = [&](int j, int& t, int& k0, int& nku, int& role) {
    if (j < rounds_dp || !sk) { t = c0 + iw + j * nx; k0 = 0; nku = nk; role = 0; return; }
    const int ts = sk_t0 + (j - rounds_dp);
    k0 = j == rounds_dp ? sk_lo - sk_t0 * nk : 0;
    const int kend = sk_hi - ts * nk < nk ? sk_hi - ts * nk : nk;
    nku = kend - k0;
    t = c0 + rounds_dp * nx + ts;
    role = k0 > 0 ? 1 : (kend < nk ? 2 : 0);
  };

  auto tile_origin = [&](int t, int& m0, int& n0) {   // grouped order: bands of 8 row-tiles, column-tile major inside
    if (g.reverse) t = T - 1 - t;
    const int band = t / band_tiles;
    const int within = t - band * band_tiles;
    const int rows_left = g.tiles_m - band * 8;
    const int band_h = rows_left < 8 ? rows_left : 8;
    const int bn = within / band_h;
    m0 = (band * 8 + (within - bn * band_h)) * TH;
    n0 = bn * 256;
  };

  int vx[2], vw[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int c = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
    vx[par] = (lane >> 3) * (int)g.lda * 2 + c * 16;
    vw[par] = (16 * (lane >> 5) + ((lane >> 3) & 3)) * (int)g.ldw * 2 + c * 16;
  }
  int sx[8], sw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sx[i] = (wave * 8 + i) * 8 * (int)g.lda * 2;
    sw[i] = (64 * wave + 32 * (i & 1) + 4 * (i >> 1)) * (int)g.ldw * 2;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned fr = (lane & 15) * 128 + ((((unsigned)lane >> 4) ^ (((unsigned)lane & 15) >> 1)) << 4);
  unsigned xa0 = lds0 + wm * (MTN * 2048) + fr, xa1 = xa0 ^ 64;
  unsigned wa0 = lds0 + V7_WOFF + wn * 16384 + fr, wa1 = wa0 ^ 64;
  unsigned dst = wave * 8192;

  // ---- DMA cursor: the K-tile the next 16 pieces fetch (runs two K-tiles ahead of the MFMAs, across units)
  int cur_j = 0, cur_kt = 0, cur_nk = nk;
  const char* cur_x = nullptr;
  const char* cur_w = nullptr;
  unsigned cur_xb = 0, cur_wb = 0;
  const int dbg_same = g.trace ? (int)g.trace[256 * 64 + 1] : 0;   // experiment: every tile fetches tile (0,0)'s operands
  auto cursor_unit = [&]() {
    if (cur_j < n_units) {
      int t, k0, nku, role_, m0, n0;
      unit(cur_j, t, k0, nku, role_);
      tile_origin(t, m0, n0);
      if (dbg_same) { m0 = 0; n0 = 0; }
      const int rows_x = g.M - m0 < TH ? g.M - m0 : TH;
      const int rows_w = g.N - n0 < 256 ? g.N - n0 : 256;
      cur_x = (const char*)(g.A + (long)m0 * g.lda) + k0 * 128;
      cur_w = (const char*)(g.W + (long)n0 * g.ldw) + k0 * 128;
      cur_xb = (unsigned)(((long)(rows_x - 1) * g.lda + g.K - 64 * k0) * 2);
      cur_wb = (unsigned)(((long)(rows_w - 1) * g.ldw + g.K - 64 * k0) * 2);
      cur_nk = nku;
    } else {
      cur_xb = 0; cur_wb = 0;   // past the last unit: null descriptors, the pieces read nothing
      cur_nk = 1 << 30;
    }
  };
  auto cursor_next = [&]() {
    if (++cur_kt == cur_nk) { cur_kt = 0; ++cur_j; cursor_unit(); }
  };
#define V8_RSRC_X() __builtin_amdgcn_make_buffer_rsrc((void*)(cur_x + cur_kt * 128), 0, cur_xb ? (int)(cur_xb - cur_kt * 128) : 0, 0x00020000)
#define V8_RSRC_W() __builtin_amdgcn_make_buffer_rsrc((void*)(cur_w + cur_kt * 128), 0, cur_wb ? (int)(cur_wb - cur_kt * 128) : 0, 0x00020000)

  f32x4 acc[8][8];
  u32x4 xf[2][8], wf[2][8];
  u32x4 ln_ring[V7_LN_RING][2];   // deferred-LayerNorm mode 2: filled inside the epilogue (tiles of a persistent workgroup drift apart)
  V7_LAB_DECLS   // (empty in the product: tools/experiments/kstep_lab.hip declares its stamp accumulators here)

  if (n_units == 0) return;   // uniform: more workgroups than work on this XCD
  // debug trace: slot 0 = realtime (100 MHz) at entry, 1 = shader clock at entry, then per tile (realtime): K loop
  // start, K loop end, epilogue end; last two slots repeat (realtime, shader clock) at exit
  unsigned long long* tr = g.trace ? g.trace + (long)b * 64 : nullptr;
  int tri = 2;
#define V8_TRACE_RT() if constexpr (LNM == 0) { if (tr && tid == 0 && tri < 40) tr[tri++] = __builtin_amdgcn_s_memrealtime(); }
  if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_memrealtime(); tr[1] = __builtin_amdgcn_s_memtime(); }
  if (tr && g.trace[256 * 64]) {   // experiment: staggered start, delay = (b * 37 % 64) / 64 * trace[256*64] * 10 ns
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long d = (unsigned long long)((b * 37) & 63) * g.trace[256 * 64] / 64;
    while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
  cursor_unit();
  {
    __amdgpu_buffer_rsrc_t rx0 = V8_RSRC_X(), rw0 = V8_RSRC_W();
    cursor_next();
    __amdgpu_buffer_rsrc_t rx1 = V8_RSRC_X(), rw1 = V8_RSRC_W();
    cursor_next();
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_X(rx0, dst, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_W(rw0, dst, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_X(rx1, dst + V7_STAGE, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V7_DMA_W(rw1, dst + V7_STAGE, i);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < MTN) V7_LDSR(xf[0][i], xa0, i * 2048);
      V7_LDSR(wf[0][i], wa0, i * 2048);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  for (int j = 0; j < n_units; ++j) {
    int t, k0, nku, role, m0, n0;
    unit(j, t, k0, nku, role);
    tile_origin(t, m0, n0);
    V8_TRACE_RT();
    {  // first K-step of the unit; the bias piece goes first, so the step's landing wait covers it too
      __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
      cursor_next();
      V7_DMA_BIAS()
      if constexpr (LNM != 0) { V7_STEP_FIRST_LN() } else { V7_STEP_FIRST(13) }
    }
    for (int kt = 1; kt < nku; ++kt) {
      __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
      cursor_next();
      V7_STEP(13)
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    V8_TRACE_RT();
    if constexpr (SK_OK) if (role != 0) {
      // ---- a tile of the stream-K region that this workgroup holds a part of.  Partials travel as agent-scope accesses
      // (sc1 on the stores and loads themselves: written through / fetched past whatever the L2 of another XCD holds), so
      // no cache-wide write-back or invalidate is needed around the counter; the parts of a tile sit on one XCD anyway when
      // workgroups are dealt to XCDs round-robin, but nothing here relies on it.  The head's wait for the other parts is
      // bounded (a wave must reach the end of the grid); running out of it is recorded for the host
      // (vt_gemm_shared_tile_timeouts), as in the weight-gradient kernel.
      // Every touch of the accumulators below is an asm statement with the tuple as a TIED operand (or a read-only one): a
      // C++ expression that redefines acc[..][..] lets the register allocator give the new value another AGPR tuple, and
      // with all 256 in use that means shuffles through scratch (the first build: 120 .. 424 B of it, and scratch holding
      // accumulators of asynchronous MFMAs is wrong, not just slow).
      const int vo = tid * 16;
      const int ts = t - c0 - rounds_dp * nx;            // the tile's index in the region
      int head = iw;                                      // the workgroup whose share holds the tile's first K-step
      if (role == 1)
        for (head = iw - 1; head > 0 && skb(head) > ts * nk; --head) {}
      int* sem = g.sk_sem + xcd * V8_SK_WGS_PER_XCD + head;
      if (role == 1) {
        const u32x4 rs = v7_rsrc((const char*)g.sk_ws + (long)(xcd * V8_SK_WGS_PER_XCD + iw) * V8_SK_PART_BYTES, (unsigned)V8_SK_PART_BYTES);
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)   // (gfx90a and later: a VMEM store takes its data from AGPRs directly)
            // (the tuple as a TIED operand although it is only read: as a plain input the allocator may stage it through
            // another register, and there is none free)
            asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1" : "+a"(acc[mt][nt]) : "v"(vo), "s"(rs), "s"((8 * mt + nt) * 4096) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(sem, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SK_TRACE
        V8_TRACE_RT();
#endif
        continue;
      }
      // head: the other parts are the non-empty shares after this one up to the share holding the tile's last K-step
      int last = iw;                                      // ... and the one holding its last K-step
      while (last + 1 < nx && skb(last + 1) < (ts + 1) * nk) ++last;
      const int expect = last - iw;
      if (tid == 0) {
        int it = 0;
        for (; it < (1 << 22) && __hip_atomic_load(sem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != expect; ++it)
          __builtin_amdgcn_s_sleep(2);
        if (it == (1 << 22)) __hip_atomic_fetch_add(g.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sem, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch finds the counter at zero
      }
      __syncthreads();
#ifdef SK_TRACE   // (-DSK_TRACE: tools/sk_trace.py's stamps inside the fix-up -- a head: K loop end, parts arrived, parts added, epilogue end)
      V8_TRACE_RT();
#endif
      // acc += part, exactly and without the vector ALU (which cannot address AGPRs): the loaded tuple q of a lane is the
      // part's 16x16 tile in the accumulator layout (lane (g, j): rows 4g + r of column j), and
      //   v_mfma_f32_16x16x4_f32  D += A_r B_r,   B_r = q[r] (lane (k, j): P[4k + r][j]),   A_r[i][k] = (i == 4k + r)
      // adds row 4k + r of the part to row 4k + r of D and zero to the others: four of them (r = 0 .. 3) add the whole tile,
      // every element receiving ONE product 1.0 * p beside zeros -- the fp32 sum, rounded once.  Four tiles round-robin, so
      // that an MFMA never follows the one whose result it accumulates onto.
      float ar[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ar[r] = ((lane & 15) == 4 * (lane >> 4) + r) ? 1.0f : 0.0f;
      int pw = iw + 1;   // (a head's tile has at least one other part: do-while -- no zero-trip path around the tied operands)
      do {
        const u32x4 rs = v7_rsrc((const char*)g.sk_ws + (long)(xcd * V8_SK_WGS_PER_XCD + pw) * V8_SK_PART_BYTES, (unsigned)V8_SK_PART_BYTES);
        f32x4 q[8];
#define V8_SK_LOAD(i) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=v"(q[(i) & 7]) : "v"(vo), "s"(rs), "s"((i) * 4096) : "memory")
#pragma unroll
        for (int i = 0; i < 8; ++i) V8_SK_LOAD(i);
#pragma unroll
        for (int gi = 0; gi < 2 * MTN; ++gi) {   // groups of four tiles: loads 4 gi .. 4 gi + 3 have landed
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * MTN - 4 * gi - 4 < 4 ? 8 * MTN - 4 * gi - 4 : 4) : "memory");
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) asm volatile("" : "+v"(q[(4 * gi + tt) & 7]));
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            const int i = 4 * gi + tt;
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n\ts_nop 1\n\t"
                         "v_mfma_f32_16x16x4_f32 %0, %2, %6, %0\n\ts_nop 1\n\t"
                         "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\ts_nop 1\n\t"
                         "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0"
                         : "+a"(acc[i >> 3][i & 7])
                         : "v"(ar[0]), "v"(ar[1]), "v"(ar[2]), "v"(ar[3]), "v"(q[i & 7][0]), "v"(q[i & 7][1]), "v"(q[i & 7][2]), "v"(q[i & 7][3]));
          }
          // the next loads into these four slots: behind the MFMAs that read them (sources are read at issue; the data
          // returns hundreds of cycles later)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            if (4 * gi + tt + 8 < 8 * MTN) V8_SK_LOAD(4 * gi + tt + 8);
        }
#undef V8_SK_LOAD
      } while (++pw <= last);
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results, before the epilogue reads them
#ifdef SK_TRACE
      V8_TRACE_RT();
#endif
    }
    V7_EPILOGUE()
    V8_TRACE_RT();
    if constexpr (SK_OK) if (role == 2) break;   // a head is its share's last unit: said aloud, the next unit's fragment registers are free in its fix-up
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  V7_LAB_EXIT
  if (tr && tid == 0) { tr[62] = __builtin_amdgcn_s_memrealtime(); tr[63] = __builtin_amdgcn_s_memtime(); }
}

