// Epilogues of the deferred-LayerNorm GEMM modes (GemmArgs::ln_mode; the inference path's fused residual + LayerNorm).
// Included by gemm_v7_kernels.hpp behind v7_epilogue_fast, whose construction they share: straight-line code, the
// per-column vectors and the row statistics arrive through LDS-DMA pieces, every VMEM instruction is inline asm with
// counted waits (hipcc would put s_waitcnt vmcnt(0) around compiler-visible ones and drain the next tile's operand DMA).
//
// Reference arithmetic being restated (un-vendored pytorch-transformers blocks, restated in oracle/bert_blocks.py):
//   BertSelfOutput / BertOutput.forward   LayerNorm(dropout(dense(h)) + input_tensor)      (oscar/modeling_bert.py:94,120)
//   BertLayerNorm                         (x - mean) / sqrt(var + eps) * weight + bias
// Here the LayerNorm of sub-layer i is applied where sub-layer i's OUTPUT is consumed: inside the accumulator of the next
// projection (mode 1) and inside the residual add of the next dense + residual (mode 2).
//
// Slab s = 0 .. 2 MTN - 1 is row block s % MTN of column half s / MTN (16 rows x 64 columns of the 16 MTN x 128 wave tile);
// a lane (j = lane & 15, gq = lane >> 4) owns row j of the block and 16 consecutive columns 16 gq .. 16 gq + 15.
//   mode 1   v = act(ra_r * acc + (rb_r * g_c + h_c))            ra = rstd_r, rb = -mean_r * rstd_r of the A operand's row r
//   mode 2   v = acc + cb_c + gamma_c * (Rs_rc * ra_r + rb_r)    v -> Cs (fp16: the stream), C (bf16: the next GEMM's operand),
//            and its row sums / sums of squares over the wave's 128 columns -> one statistics slice
//            (stats_out[2 * column tile + column wave]).
#pragma once

// A register pin (the value is final / stays where it is from here on).  NOT an empty asm statement: hipcc's hazard
// recognizer counts an inline-asm statement as an instruction, so between a packed fp32 op and a dependent one (which need a
// wait state between them) an EMPTY statement makes it drop the s_nop it would otherwise insert -- seen as v_pk_fma_f32
// reading the previous v_pk_fma_f32's result one slot early: stale values in the last lanes of every row of 16.  With a real
// s_nop inside, what the recognizer assumes is true.
#define V7_LN_PIN(...) asm volatile("s_nop 0" : __VA_ARGS__)

// The fp16 stream of mode 2 comes through a ring of V7_LN_RING slabs (2 x 16 B per lane and slab = 8 registers), as the
// bf16 residual does in v7_epilogue_fast: eight slabs cover the ~3.5 us the stream takes to arrive under load.
// (The stream was fp32 at first: 10 bytes per element through this epilogue against 6 now, and these GEMMs are
// memory-bound in it -- out-proj at M = 14 592: 38.5 us against 23.6 us for the plain bf16 residual epilogue.  An
// emulation of the forward with an fp16 stream gives the same error against the fp32 reference as an fp32 stream:
// 2.2e-2 against 2.1e-2 on the base config, where the bf16 stream of the seven-launch layer gives 7.4e-2.)
#define V7_LN_RING 8
// VMEM operations younger than slab s's ring loads at the moment slab s waits for them (T slabs; LD loads and ST stores
// per slab; issue order per slab: wait, arithmetic, loads of slab s + RING, stores of slab s; prologue: slabs 0 .. RING-1)
constexpr int v7_ln_vmcnt(int s, int T) {
  const int D = V7_LN_RING, LD = 2, ST = 4;   // every second slab issues a fifth store (the row block's statistics):
                                                // counting 4 makes the wait ask for a little more than needed, never less
  int n = 0;
  if (s < D) {
    n += LD * ((T < D ? T : D) - 1 - s);
    for (int k = 0; k < s; ++k) n += (k + D < T ? LD : 0) + ST;
  } else {
    n += ST;
    for (int k = s - D + 1; k < s; ++k) n += (k + D < T ? LD : 0) + ST;
  }
  // The counter has six bits: a wave must never have more than 63 operations in flight.  After its wait a slab issues up to
  // LD + ST + 1 = 7 more, so the wait leaves at most 52 (a smaller count only waits for more of the OLDER operations: stores
  // issued a ring ago).
  return n > 52 ? 52 : n;
}

template <int OFF>
__device__ __forceinline__ void v7_buf_load16_at(u32x4& d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void v7_buf_store16_at(u32x4 d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4" ::"v"(d), "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
}
__device__ __forceinline__ void v7_buf_store8(u32x2 d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" ::"v"(d), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// 16 fp32 values of a per-column vector parked in a wave's 1 KiB LDS slot (256 floats = the tile's columns)
__device__ __forceinline__ void v7_ln_colvec(unsigned slot, int ecr, int gq, float (&out)[16]) {
  u32x4 q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = slot + 4 * (ecr + 16 * gq) + 16 * i;
    asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(a));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    V7_LN_PIN("+v"(q[i]));
#pragma unroll
    for (int e = 0; e < 4; ++e) out[4 * i + e] = __uint_as_float(q[i][e]);
  }
}

#define V7_LN_STAT_READ(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// Slab order.  Mode 1: column half outside, row block inside (the per-column vectors h, g of a half are read once).
// Mode 2: row block outside, column half inside: a row block's sums are finished (and stored) after its second slab, so two
// running sums are alive instead of 2 MTN, and the two 256-byte halves of a stream row are written back to back; the
// per-column vectors are read again for every slab (eight ds_read_b128, issued in front of the slab's VMEM wait).
// Register budget of mode 2 (the reason for this): the ring registers are written by loads hipcc does not see, so anything
// the allocator moves out of the way (a copy to an AGPR, a v_mov) BEFORE the data has landed carries the old bits -- which is
// what 16 row factors + 16 running sums in registers beside a 96-register fp32 ring caused on the 224- and 256-row tiles
// (wrong rows that came and went).  tests/test_build_isa.py checks that the epilogue makes no such copy.
template <int LNM, int MTN>
struct V7LnOrder {
  static constexpr int mt(int s) { return LNM == 2 ? s / 2 : s % MTN; }
  static constexpr int nh(int s) { return LNM == 2 ? s % 2 : s / MTN; }
};

// The ring's first V7_LN_RING slabs, issued by the one-tile-per-workgroup kernel at its very start (PRE): with a single round
// of tiles every workgroup reaches its epilogue at the same moment, and stream loads issued there arrive together with
// everybody else's at the memory's pace; issued before the K loop they have long landed (the prologue's operand wait covers
// them: they are older) and two thirds of a 192-row tile's stream never wait.
template <int LNM, int MTN>
__device__ __forceinline__ void v7_ln_ring_prefetch(const GemmArgs& g, u32x4 (&ring)[V7_LN_RING][2], int wave, int m0, int n0) {
  typedef V7LnOrder<LNM, MTN> Ord;
  const int lane = v7_lane_now();
  const int wm = wave >> 1, wn = wave & 1;
  const int gq = lane >> 4, j = lane & 15;
  const int rows = g.M - m0 < 32 * MTN ? g.M - m0 : 32 * MTN;
  const int ldr_b = (int)g.ldrs * 2;
  const u32x4 rs_r = v7_rsrc(g.Rs + (long)m0 * g.ldrs, (unsigned)rows * ldr_b);
  const int vo_r = j * ldr_b + gq * 32;
#pragma unroll
  for (int s_ = 0; s_ < V7_LN_RING; ++s_) {
    const int so_ = (16 * MTN * wm + 16 * Ord::mt(s_)) * ldr_b + (n0 + 128 * wn + 64 * Ord::nh(s_)) * 2;
    v7_buf_load16_at<0>(ring[s_][0], rs_r, vo_r, so_);
    v7_buf_load16_at<16>(ring[s_][1], rs_r, vo_r, so_);
  }
}

template <int ACT, int LNM, int MTN, bool PRE = false>
__device__ __forceinline__ void v7_epilogue_ln(const GemmArgs& g, f32x4 (&acc)[8][8], int lane_in, int wave, int m0, int n0, unsigned lds0,
                                               u32x4 (&ring)[V7_LN_RING][2]) {
  static_assert(LNM == 1 || LNM == 2, "deferred-LayerNorm mode");
  // every per-lane address below derives from a lane index computed here: as loop invariants of the tile loop the allocator
  // would compute them once at kernel entry and park them in scratch across the K loop
  const int lane = v7_lane_now();
  (void)lane_in;
  const int wm = wave >> 1, wn = wave & 1;
  if (n0 + 128 * wn >= g.N) return;   // N % 128 == 0 (host-checked): a wave's 128 columns are all inside N or all past it
  const unsigned bias_slot = lds0 + 2 * V7_STAGE + wave * 1024;   // mode 1: h, mode 2: cb
  const unsigned colv_slot = lds0 + V7_LN_COLV + wave * 1024;     // mode 1: g, mode 2: gamma
  const unsigned stat_slot = lds0 + V7_LN_STAT + wm * 8192;       // [8 slices][128 rows][sum, sum of squares]
  const unsigned rowf_slot = lds0 + V7_LN_ROWF + wave * 1024;     // this wave's [128 rows][ra, rb]
  const int gq = lane >> 4, j = lane & 15;
  const int rows = g.M - m0 < 32 * MTN ? g.M - m0 : 32 * MTN;
  const int ldc_b = (int)g.ldc * 2;
  const u32x4 rs_c = v7_rsrc((const bf16_t*)g.C + (long)m0 * g.ldc, (unsigned)rows * ldc_b);
  const int vo_c = j * ldc_b + gq * 32;

  // ---- row factors ra = rstd, rb = -mean * rstd of the wave's 16 MTN rows: lane l finishes rows l and l + 64 from the
  // statistics slices (slices past ln_np and rows past M were fetched through zero-length descriptors: zeros, giving
  // rstd = 1 / sqrt(eps) against zero operands: finite) and parks them in the wave's own slot; a slab reads its row's pair.
  {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (64 * h < 16 * MTN) {
        u32x2 sp[8];
        const unsigned sa = stat_slot + (lane + 64 * h) * 8;
#pragma unroll
        for (int p = 0; p < 8; ++p) V7_LN_STAT_READ(sp[p], sa, p * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          V7_LN_PIN("+v"(sp[p]));
          sum += __uint_as_float(sp[p][0]);
          sq += __uint_as_float(sp[p][1]);
        }
        const float mean = sum * g.ln_inv_n;
        const float var = fmaxf(sq * g.ln_inv_n - mean * mean, 0.f);
        const float rstd = rsqrtf(var + g.ln_eps);
        const u32x2 f = {__float_as_uint(rstd), __float_as_uint(-mean * rstd)};
        asm volatile("ds_write_b64 %0, %1" ::"v"(rowf_slot + (lane + 64 * h) * 8), "v"(f) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave's LDS operations complete in order: its own writes are visible to it
  }

  // mode 2 state: descriptors of the fp16 stream in / out, the ring
  const int ldr_b = (int)g.ldrs * 2, ldo_b = (int)g.ldcs * 2;
  const u32x4 rs_r = v7_rsrc(LNM == 2 ? g.Rs + (long)m0 * g.ldrs : nullptr, LNM == 2 ? (unsigned)rows * ldr_b : 0u);
  const u32x4 rs_o = v7_rsrc(LNM == 2 ? g.Cs + (long)m0 * g.ldcs : nullptr, LNM == 2 ? (unsigned)rows * ldo_b : 0u);
  const int vo_r = j * ldr_b + gq * 32, vo_o = j * ldo_b + gq * 32;
  const int part = (n0 >> 7) + wn;   // mode 2: the statistics slice this wave's 128 columns make
  const u32x4 rs_s = v7_rsrc(LNM == 2 ? g.stats_out + (long)part * g.ln_rows * 2 : nullptr, LNM == 2 ? (unsigned)g.M * 8u : 0u);
  float s1 = 0.f, s2 = 0.f;
  float c0[16], c1[16];   // mode 1: h, g; mode 2: cb, gamma -- of the slab's column half
  typedef V7LnOrder<LNM, MTN> Ord;

#define V7_LN_RING_LOAD(S)                                                                                  \
  {                                                                                                         \
    constexpr int mt_ = Ord::mt(S), nh_ = Ord::nh(S), sl_ = (S) % V7_LN_RING;                               \
    const int so_ = (16 * MTN * wm + 16 * mt_) * ldr_b + (n0 + 128 * wn + 64 * nh_) * 2;                    \
    v7_buf_load16_at<0>(ring[sl_][0], rs_r, vo_r, so_);                                                     \
    v7_buf_load16_at<16>(ring[sl_][1], rs_r, vo_r, so_);                                                    \
  }
#define V7_LN_SLAB(S)                                                                                       \
  if ((S) < 2 * MTN) {                                                                                      \
    constexpr int MT = Ord::mt(S), NH = Ord::nh(S), SL = (S) % V7_LN_RING;                                  \
    const int ecr = 128 * wn + 64 * NH;   /* tile-relative first column of the slab */                      \
    const int ec = n0 + ecr;                                                                                \
    const int so_row = 16 * MTN * wm + 16 * MT;                                                             \
    if (LNM == 2 || MT == 0) {                                                                              \
      v7_ln_colvec(bias_slot, ecr, gq, c0);                                                                 \
      v7_ln_colvec(colv_slot, ecr, gq, c1);                                                                 \
    }                                                                                                       \
    u32x2 rf;   /* this row's factors */                                                                    \
    asm volatile("ds_read_b64 %0, %1" : "=v"(rf) : "v"(rowf_slot + (16 * MT + j) * 8));                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    V7_LN_PIN("+v"(rf));                                                                                    \
    const float ra = __uint_as_float(rf[0]), rb = __uint_as_float(rf[1]);                                   \
    float v[16];                                                                                            \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                           \
      V7_LN_PIN("+a"(acc[MT][4 * NH + t]));   /* stays in AGPRs until its slab's turn */                     \
    if (LNM == 1) {                                                                                         \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                       \
        f32x4 z;                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                       \
          z[e] = fmaf(ra, acc[MT][4 * NH + t][e], fmaf(rb, c1[4 * t + e], c0[4 * t + e]));                  \
        if (ACT == ACT_GELU) z = gelu_poly4(z);                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) v[4 * t + e] = (ACT == ACT_GELU) ? z[e] : apply_act<ACT>(z[e]); \
      }                                                                                                     \
    } else {                                                                                                \
      if (!PRE || (S) >= V7_LN_RING)   /* PRE: the first ring of slabs landed before the K loop */              \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(v7_ln_vmcnt((S), 2 * MTN)) : "memory");                    \
      V7_LN_PIN("+v"(ring[SL][0]), "+v"(ring[SL][1]));                                                      \
      if (NH == 0) { s1 = 0.f; s2 = 0.f; }                                                                  \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                     \
          /* column 4 t + e of the lane's 16: dword (4 t + e) / 2 of the slab's 8, its low or high half */     \
          const uint32_t rw_ = ring[SL][t >> 1][2 * (t & 1) + (e >> 1)];                                    \
          const float xh = fmaf((e & 1) ? f16hi(rw_) : f16lo(rw_), ra, rb);                                 \
          const float w = fmaf(xh, c1[4 * t + e], acc[MT][4 * NH + t][e] + c0[4 * t + e]);                  \
          v[4 * t + e] = w;                                                                                 \
          s1 += w;                                                                                          \
          s2 = fmaf(w, w, s2);                                                                              \
        }                                                                                                   \
      if ((S) + V7_LN_RING < 2 * MTN) V7_LN_RING_LOAD((S) + V7_LN_RING)                                     \
      const int so_o = so_row * ldo_b + ec * 2;                                                             \
      u32x4 h0, h1;                                                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
        h0[i] = pack_f16x2(v[2 * i], v[2 * i + 1]);                                                         \
        h1[i] = pack_f16x2(v[8 + 2 * i], v[8 + 2 * i + 1]);                                                 \
      }                                                                                                     \
      v7_buf_store16_at<0>(h0, rs_o, vo_o, so_o);                                                           \
      v7_buf_store16_at<16>(h1, rs_o, vo_o, so_o);                                                          \
    }                                                                                                       \
    u32x4 o0, o1;                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      o0[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);                                                          \
      o1[i] = pack_bf16x2(v[8 + 2 * i], v[8 + 2 * i + 1]);                                                  \
    }                                                                                                       \
    v7_buf_store16(o0, rs_c, vo_c, so_row * ldc_b + ec * 2);                                                \
    v7_buf_store16_o16(o1, rs_c, vo_c, so_row * ldc_b + ec * 2);                                            \
    if (LNM == 2 && NH == 1) {                                                                              \
      /* the row block's sums over the wave's 128 columns: the four lanes of a row are l, l ^ 16 (ds_swizzle: a xor    \
         inside a half) and l ^ 32 (permlane32_swap) -- fixed patterns, no lane index to keep alive */                   \
      float a_ = s1, b_ = s2;                                                                               \
      a_ += __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(a_), 0x401f));                      \
      b_ += __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(b_), 0x401f));                      \
      const auto pa_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_), __float_as_uint(a_), false, false); \
      const auto pb_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(b_), __float_as_uint(b_), false, false); \
      a_ = __uint_as_float(pa_[0]) + __uint_as_float(pa_[1]);                                               \
      b_ = __uint_as_float(pb_[0]) + __uint_as_float(pb_[1]);                                               \
      if (gq == 0) v7_buf_store8((u32x2){__float_as_uint(a_), __float_as_uint(b_)}, rs_s, j * 8, (m0 + so_row) * 8); \
    }                                                                                                       \
  }

  if (LNM == 2 && !PRE) {   // ring prologue: the first V7_LN_RING slabs (every tile has at least eight: MTN >= 4)
    V7_LN_RING_LOAD(0) V7_LN_RING_LOAD(1) V7_LN_RING_LOAD(2) V7_LN_RING_LOAD(3)
    V7_LN_RING_LOAD(4) V7_LN_RING_LOAD(5) V7_LN_RING_LOAD(6) V7_LN_RING_LOAD(7)
  }
  V7_LN_SLAB(0) V7_LN_SLAB(1) V7_LN_SLAB(2) V7_LN_SLAB(3) V7_LN_SLAB(4) V7_LN_SLAB(5) V7_LN_SLAB(6) V7_LN_SLAB(7)
  V7_LN_SLAB(8) V7_LN_SLAB(9) V7_LN_SLAB(10) V7_LN_SLAB(11) V7_LN_SLAB(12) V7_LN_SLAB(13) V7_LN_SLAB(14) V7_LN_SLAB(15)
#undef V7_LN_SLAB
#undef V7_LN_RING_LOAD
}
