// NT GEMM, persistent 256x256-tile kernel on 32x32x16 MFMAs (variant 17 of vt_gemm_dispatch).
//
// Same design as gemm_nt_bf16_v8 (gemm_v7.hip): one workgroup per CU, four waves of 128x128, 256 accumulator
// registers in AGPRs, two 64 KiB LDS stages filled by buffer_load ... lds, a hand-ordered K-step that runs across
// tile boundaries, straight-line epilogue without compiler-visible VMEM.  What changes is the matrix instruction:
// v8's K-step is 128 v_mfma_f32_16x16x32_bf16, each of which holds the SIMD's vector issue for 8 of its 16 cycles, so
// the step's 16 LDS-DMA pieces (~48 issue cycles each, measured by removing them: 1.50 -> 1.12 us per step) cannot
// hide behind the matrix pipe.  64 v_mfma_f32_32x32x16_bf16 do the same work with 24 free issue cycles per MFMA.
//   * wave tile = 4 x 4 MFMA tiles of 32x32 (16 accumulator registers each); operands swapped (W rows feed the A
//     port) and W rows permuted inside each 32-row block at staging time (image row r <- W row 16*((r>>2)&1) +
//     4*(r>>3) + (r&3)) so that a lane ends up with 16 consecutive output columns of one output row;
//   * per K-step (BK = 64 = four k16 substeps of 16 MFMAs): substep-0 fragments are already in registers; during
//     substep 0 the W fragments of substep 1 and the X fragments of substeps 1..3 are read (X image done -> barrier),
//     then the 8 X pieces of the K-tile two steps ahead alternate with the W fragment reads of substeps 2, 3
//     (-> barrier), then the 8 W pieces every third MFMA, vmcnt(12) + barrier, the next stage's substep-0 fragments;
//   * serves bf16 output with N a multiple of 128 and no row remap (the epilogue of gemm_v7.hip's fast path, on
//     32-row slabs); everything else stays with the other variants.
#include "gemm_common.hpp"

#define V9_STAGE 65536
#define V9_WOFF 32768
#define V9_LDS_BYTES (2 * V9_STAGE + 4096)   // two operand stages + 1 KiB per wave: the tile's bias values

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define V9_MFMA(ks, nt, mt) \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wf[ks][nt]), "v"(xf[ks][mt]))
#define V9_MFMA0(ks, nt, mt) \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[mt][nt]) : "v"(wf[ks][nt]), "v"(xf[ks][mt]))
#define V9_LDSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

__device__ __forceinline__ u32x4 v9_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void v9_load16(u32x4& d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void v9_load16_o16(u32x4& d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void v9_store16(u32x4 d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(d), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void v9_store16_o16(u32x4 d, u32x4 rs, int voff, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:16" ::"v"(d), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// slab (MT, NT) = 32 rows x 32 columns: lane (j = lane&31, h = lane>>5) owns row 32*MT + j, columns 32*NT + 16h .. +15
// (accumulator element i -> column 16h + i after the W-row permutation).  Slabs run NT-major; the residual of slab
// s = 4*NT + MT was issued 4 slabs earlier (the first four before slab 0) -> counted wait, VMEM retires in order.
#define V9_SLAB(MT, NT)                                                                                   \
  {                                                                                                       \
    constexpr int S_ = 4 * (NT) + (MT);                                                                   \
    float v[16];                                                                                          \
    asm volatile("" : "+a"(acc[MT][NT]));   /* stays in AGPRs until its slab's turn */                     \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) v[i] = acc[MT][NT][i] + bv[i];                         \
    const int so_row = 128 * wm + 32 * (MT);                                                              \
    const int ecb = (n0 + 128 * wn + 32 * (NT)) * 2;   /* byte offset of the slab's first column */        \
    if (has_c2) {   /* saved for the backward pass: the activation's derivative (GELU) or the pre-activation */ \
      float d2[16];                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 16; i += 2) {                                                 \
        if (ACT == ACT_GELU) {                                                                            \
          f32x2 gg, dd;                                                                                   \
          gelu_erf_both2((f32x2){v[i], v[i + 1]}, gg, dd);                                                \
          v[i] = gg[0]; v[i + 1] = gg[1]; d2[i] = dd[0]; d2[i + 1] = dd[1];                               \
        } else {                                                                                          \
          d2[i] = v[i]; d2[i + 1] = v[i + 1];                                                             \
          v[i] = apply_act<ACT>(v[i]); v[i + 1] = apply_act<ACT>(v[i + 1]);                               \
        }                                                                                                 \
      }                                                                                                   \
      u32x4 p0, p1;                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        p0[i] = pack_bf16x2(d2[2 * i], d2[2 * i + 1]);                                                    \
        p1[i] = pack_bf16x2(d2[8 + 2 * i], d2[8 + 2 * i + 1]);                                            \
      }                                                                                                   \
      v9_store16(p0, rs_c2, vo_c2, so_row * ldc2_b + ecb);                                                \
      v9_store16_o16(p1, rs_c2, vo_c2, so_row * ldc2_b + ecb);                                            \
    } else {                                                                                              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) v[i] = apply_act<ACT>(v[i]);                         \
    }                                                                                                     \
    if (g.drop.thresh) {                                                                                  \
      const uint32_t e0 = (uint32_t)(m0 + so_row + j) * (uint32_t)g.N + (uint32_t)(ecb / 2 + 16 * h);     \
      vt_drop_run<16>(g.drop, e0, v);                                                                     \
    }                                                                                                     \
    if (HAS_R) {                                                                                          \
      if (has_c2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_ < 4 ? 6 + 2 * S_ : (S_ <= 12 ? 14 : 38 - 2 * S_)) : "memory"); \
      asm volatile("" : "+v"(rq[S_ & 3][0]), "+v"(rq[S_ & 3][1]));                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        const u32x4 q0 = rq[S_ & 3][0], q1 = rq[S_ & 3][1];                                               \
        const float r0 = bf16lo(q0[i]), r1 = bf16hi(q0[i]), r2 = bf16lo(q1[i]), r3 = bf16hi(q1[i]);       \
        if (ACT == ACT_MUL) { v[2 * i] *= r0; v[2 * i + 1] *= r1; v[8 + 2 * i] *= r2; v[8 + 2 * i + 1] *= r3; } \
        else { v[2 * i] += r0; v[2 * i + 1] += r1; v[8 + 2 * i] += r2; v[8 + 2 * i + 1] += r3; }          \
      }                                                                                                   \
      if (S_ < 12) {   /* slab S_ + 4 = same rows, next 32 columns */                                      \
        v9_load16(rq[S_ & 3][0], rs_r, vo_r, so_row * ldr_b + ecb + 64);                                  \
        v9_load16_o16(rq[S_ & 3][1], rs_r, vo_r, so_row * ldr_b + ecb + 64);                              \
      }                                                                                                   \
    }                                                                                                     \
    u32x4 o0, o1;                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
      o0[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);                                                        \
      o1[i] = pack_bf16x2(v[8 + 2 * i], v[8 + 2 * i + 1]);                                                \
    }                                                                                                     \
    v9_store16(o0, rs_c, vo_c, so_row * ldc_b + ecb);                                                     \
    v9_store16_o16(o1, rs_c, vo_c, so_row * ldc_b + ecb);                                                 \
  }

#define V9_COLS(NT)                                                                                       \
  {                                                                                                       \
    u32x4 bq[4];                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
      const unsigned ba = bias_slot + 4 * (128 * wn + 32 * (NT) + 16 * h) + 16 * i;                       \
      asm volatile("ds_read_b128 %0, %1" : "=v"(bq[i]) : "v"(ba));                                        \
    }                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    float bv[16];                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
      asm volatile("" : "+v"(bq[i]));                                                                     \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) bv[4 * i + e] = __uint_as_float(bq[i][e]);            \
    }                                                                                                     \
    V9_SLAB(0, NT) V9_SLAB(1, NT) V9_SLAB(2, NT) V9_SLAB(3, NT)                                           \
  }

template <int ACT, bool HAS_R>
__device__ __forceinline__ void v9_epilogue(const GemmArgs& g, f32x16_t (&acc)[4][4], int lane, int wave, int m0, int n0, unsigned lds0) {
  const int wm = wave >> 1, wn = wave & 1;
  if (n0 + 128 * wn >= g.N) return;   // N % 128 == 0 (host-checked): the wave's 128 columns are valid or absent
  const unsigned bias_slot = lds0 + 2 * V9_STAGE + wave * 1024;   // this wave's copy of the tile's 256 bias values
  const int rows = g.M - m0 < 256 ? g.M - m0 : 256;
  const int ldc_b = (int)g.ldc * 2, ldc2_b = (int)g.ldc2 * 2, ldr_b = (int)g.ldr * 2;
  // rows past M fall outside num_records: their loads read zeros, their stores are dropped
  const u32x4 rs_c = v9_rsrc((const bf16_t*)g.C + (long)m0 * g.ldc, (unsigned)rows * ldc_b);
  const u32x4 rs_c2 = v9_rsrc(g.C2 ? g.C2 + (long)m0 * g.ldc2 : nullptr, g.C2 ? (unsigned)rows * ldc2_b : 0u);
  const u32x4 rs_r = v9_rsrc(g.R ? g.R + (long)m0 * g.ldr : nullptr, g.R ? (unsigned)rows * ldr_b : 0u);
  const int h = lane >> 5, j = lane & 31;
  const int vo_c = j * ldc_b + h * 32, vo_c2 = j * ldc2_b + h * 32, vo_r = j * ldr_b + h * 32;
  const bool has_c2 = g.C2 != nullptr;
  u32x4 rq[4][2];   // residual ring: the next four slabs, two 8-column halves each
  if (HAS_R) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v9_load16(rq[q][0], rs_r, vo_r, (128 * wm + 32 * q) * ldr_b + (n0 + 128 * wn) * 2);
      v9_load16_o16(rq[q][1], rs_r, vo_r, (128 * wm + 32 * q) * ldr_b + (n0 + 128 * wn) * 2);
    }
  }
  V9_COLS(0)
  V9_COLS(1)
  V9_COLS(2)
  V9_COLS(3)
}

// K-step.  MFMA i = 16*ks + 4*nt + mt uses fragment sets xf[ks][mt], wf[ks][nt]; FIRST: substep 0 starts from C = 0.
#define V9_STEP(MFMA_K0)                                                                                  \
  {                                                                                                       \
    /* re-defined every step: as plain loop invariants the allocator parks them in scratch */             \
    asm volatile("" : "+v"(vx[0]), "+v"(vx[1]), "+v"(vw[0]), "+v"(vw[1]));                                \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {   /* substep 0: gaps 0..15 */                        \
      MFMA_K0(0, i >> 2, i & 3);                                                                          \
      if (i < 4) V9_LDSR(wf[1][i & 3], wa[1], (i & 3) * 4096);                                            \
      if (i >= 4 && i < 8) V9_LDSR(xf[1][i & 3], xa[1], (i & 3) * 4096);                                  \
      if (i >= 8 && i < 12) V9_LDSR(xf[2][i & 3], xa[2], (i & 3) * 4096);                                 \
      if (i >= 12) V9_LDSR(xf[3][i & 3], xa[3], (i & 3) * 4096);                                          \
    }                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   /* substep-1 fragments are in */                  \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {   /* substep 1: gaps 16..31 */                       \
      V9_MFMA(1, i >> 2, i & 3);                                                                          \
      if (i == 3) {   /* every wave is done with the X image of this stage */                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                     \
      }                                                                                                   \
      if (i >= 4 && !(i & 1)) V9_DMA_X(rx, dst, (i - 4) >> 1);             /* gaps 20,22,..,30: X pieces 0..5 */ \
      if (i >= 5 && i < 13 && (i & 1)) V9_LDSR(wf[2][((i - 5) >> 1) & 3], wa[2], (((i - 5) >> 1) & 3) * 4096); \
      if (i >= 13 && (i & 1)) V9_LDSR(wf[3][((i - 13) >> 1) & 3], wa[3], (((i - 13) >> 1) & 3) * 4096);   \
    }                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   /* substep-2 W fragments are in */                \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {   /* substep 2: gaps 32..47 */                       \
      V9_MFMA(2, i >> 2, i & 3);                                                                          \
      if (i == 0) V9_DMA_X(rx, dst, 6);                                                                   \
      if (i == 2) V9_DMA_X(rx, dst, 7);                                                                   \
      if (i == 1) V9_LDSR(wf[3][2], wa[3], 2 * 4096);                                                     \
      if (i == 3) V9_LDSR(wf[3][3], wa[3], 3 * 4096);                                                     \
      if (i == 5) {   /* ... and with its W image */                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                     \
      }                                                                                                   \
      if (i == 6) V9_DMA_W(rw, dst, 0);                                                                   \
      if (i == 9) V9_DMA_W(rw, dst, 1);                                                                   \
      if (i == 12) V9_DMA_W(rw, dst, 2);                                                                  \
      if (i == 15) V9_DMA_W(rw, dst, 3);                                                                  \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {   /* substep 3: gaps 48..63 */                       \
      V9_MFMA(3, i >> 2, i & 3);                                                                          \
      if (i == 1) {   /* 12 pieces of this step are out: the next K-tile (issued a step ago) is in */     \
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                     \
      }                                                                                                   \
      if (i == 2) V9_DMA_W(rw, dst, 4);                                                                   \
      if (i == 5) V9_DMA_W(rw, dst, 5);                                                                   \
      if (i == 8) V9_DMA_W(rw, dst, 6);                                                                   \
      if (i == 11) V9_DMA_W(rw, dst, 7);                                                                  \
      if (i == 3) V9_LDSR(wf[0][0], wa[0] ^ V9_STAGE, 0 * 4096);                                          \
      if (i == 4) V9_LDSR(wf[0][1], wa[0] ^ V9_STAGE, 1 * 4096);                                          \
      if (i == 6) V9_LDSR(wf[0][2], wa[0] ^ V9_STAGE, 2 * 4096);                                          \
      if (i == 7) V9_LDSR(wf[0][3], wa[0] ^ V9_STAGE, 3 * 4096);                                          \
      if (i == 9) V9_LDSR(xf[0][0], xa[0] ^ V9_STAGE, 0 * 4096);                                          \
      if (i == 10) V9_LDSR(xf[0][1], xa[0] ^ V9_STAGE, 1 * 4096);                                         \
      if (i == 12) V9_LDSR(xf[0][2], xa[0] ^ V9_STAGE, 2 * 4096);                                         \
      if (i == 13) V9_LDSR(xf[0][3], xa[0] ^ V9_STAGE, 3 * 4096);                                         \
    }                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) { xa[q_] ^= V9_STAGE; wa[q_] ^= V9_STAGE; }          \
    dst ^= V9_STAGE;                                                                                      \
  }

template <int ACT, bool HAS_R>
__global__ __launch_bounds__(256, 1) void gemm_nt_bf16_v9(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int T = g.tiles_m * g.tiles_n;
  const int nwg = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7;
  const int nx = (nwg - xcd + 7) >> 3;                 // workgroups on this XCD
  const int ng = nwg < 8 ? nwg : 8;                    // XCD groups that have a workgroup
  const int c0 = (int)((long)T * xcd / ng), c1 = (int)((long)T * (xcd + 1) / ng);
  const int first = c0 + (b >> 3);
  const int band_tiles = 8 * g.tiles_n;
  const int nk = g.K >> 6;

  auto tile_origin = [&](int t, int& m0, int& n0) {   // grouped order: bands of 8 row-tiles, column-tile major inside
    const int band = t / band_tiles;
    const int within = t - band * band_tiles;
    const int rows_left = g.tiles_m - band * 8;
    const int band_h = rows_left < 8 ? rows_left : 8;
    const int bn = within / band_h;
    m0 = (band * 8 + (within - bn * band_h)) * 256;
    n0 = bn * 256;
  };

  // ---- DMA addressing.  Piece p = 8*wave + i covers image rows 8p .. 8p+7 (128 B per row); lane -> row 8p + (lane>>3),
  // 16-B chunk (lane&7) ^ swz(row), swz(row) = (4*(i&1) + (lane>>4)) & 7.  X rows as they lie; W rows permuted inside
  // each 32-row block: image row r <- W row 16*((r>>2)&1) + 4*((r>>3)&3) + (r&3).
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
    sw[i] = (64 * wave + 32 * (i >> 2) + 4 * (i & 3)) * (int)g.ldw * 2;
  }
  // ---- fragment addresses: image row 128*w + 32*blk + (lane&31), 16-B chunk (2*ks + (lane>>5)) ^ swz(row)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned fr = (lane & 31) * 128 + ((((unsigned)lane >> 5) ^ ((((unsigned)lane & 31) >> 1) & 7)) << 4);
  unsigned xa[4], wa[4];   // per k16 substep; toggled (^ V9_STAGE) every K-step
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    xa[ks] = (lds0 + wm * 16384 + fr) ^ (32u * ks);
    wa[ks] = (lds0 + V9_WOFF + wn * 16384 + fr) ^ (32u * ks);
  }
  unsigned dst = wave * 8192;

  // ---- DMA cursor: the K-tile the next 16 pieces fetch (two K-tiles ahead of the MFMAs, across tiles)
  int cur_t = first, cur_kt = 0;
  const char* cur_x = nullptr;
  const char* cur_w = nullptr;
  unsigned cur_xb = 0, cur_wb = 0;
  auto cursor_tile = [&]() {
    if (cur_t < c1) {
      int m0, n0;
      tile_origin(cur_t, m0, n0);
      const int rows_x = g.M - m0 < 256 ? g.M - m0 : 256;
      const int rows_w = g.N - n0 < 256 ? g.N - n0 : 256;
      cur_x = (const char*)(g.A + (long)m0 * g.lda);
      cur_w = (const char*)(g.W + (long)n0 * g.ldw);
      cur_xb = (unsigned)(((long)(rows_x - 1) * g.lda + g.K) * 2);
      cur_wb = (unsigned)(((long)(rows_w - 1) * g.ldw + g.K) * 2);
    } else {
      cur_xb = 0; cur_wb = 0;   // past the last tile: null descriptors, the pieces read nothing
    }
  };
  auto cursor_next = [&]() {
    if (++cur_kt == nk) { cur_kt = 0; cur_t += nx; cursor_tile(); }
  };
#define V9_RSRC_X() __builtin_amdgcn_make_buffer_rsrc((void*)(cur_x + cur_kt * 128), 0, cur_xb ? (int)(cur_xb - cur_kt * 128) : 0, 0x00020000)
#define V9_RSRC_W() __builtin_amdgcn_make_buffer_rsrc((void*)(cur_w + cur_kt * 128), 0, cur_wb ? (int)(cur_wb - cur_kt * 128) : 0, 0x00020000)
#define V9_DMA_X(rs, d, i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (d) + (i) * 1024), 16, vx[(i) & 1], sx[i], 0, 0)
#define V9_DMA_W(rs, d, i) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (d) + V9_WOFF + (i) * 1024), 16, vw[(i) & 1], sw[i], 0, 0)

  f32x16_t acc[4][4];     // defined by the first K-step of every tile (C = 0)
  u32x4 xf[4][4], wf[4][4];

  if (first >= c1) return;   // uniform: more workgroups than tiles on this XCD
  cursor_tile();
  {
    __amdgpu_buffer_rsrc_t rx0 = V9_RSRC_X(), rw0 = V9_RSRC_W();
    cursor_next();
    __amdgpu_buffer_rsrc_t rx1 = V9_RSRC_X(), rw1 = V9_RSRC_W();
    cursor_next();
#pragma unroll
    for (int i = 0; i < 8; ++i) V9_DMA_X(rx0, dst, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V9_DMA_W(rw0, dst, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V9_DMA_X(rx1, dst + V9_STAGE, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) V9_DMA_W(rw1, dst + V9_STAGE, i);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V9_LDSR(wf[0][i], wa[0], i * 4096);
      V9_LDSR(xf[0][i], xa[0], i * 4096);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  for (int t = first; t < c1; t += nx) {
    int m0, n0;
    tile_origin(t, m0, n0);
    {  // first K-step of the tile; the bias piece goes first, so the step's landing wait covers it too
      __amdgpu_buffer_rsrc_t rx = V9_RSRC_X(), rw = V9_RSRC_W();
      cursor_next();
      {
        const int bn_ = g.N - n0 < 256 ? g.N - n0 : 256;
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bias ? g.bias + n0 : nullptr), 0, g.bias ? bn_ * 4 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(smem + 2 * V9_STAGE + wave * 1024), 16, lane * 16, 0, 0, 0);
      }
      V9_STEP(V9_MFMA0)
    }
    for (int kt = 1; kt < nk; ++kt) {
      __amdgpu_buffer_rsrc_t rx = V9_RSRC_X(), rw = V9_RSRC_W();
      cursor_next();
      V9_STEP(V9_MFMA)
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    v9_epilogue<ACT, HAS_R>(g, acc, lane, wave, m0, n0, lds0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int v9_grid(int tiles) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    cus = p.multiProcessorCount;
  }
  return tiles < cus ? tiles : cus;
}

template <int ACT>
static int launch_v9(const GemmArgs& g, hipStream_t stream) {
  GemmArgs g9 = g;
  g9.tiles_m = (g.M + 255) / 256;
  g9.tiles_n = (g.N + 255) / 256;
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  if ((g.N & 127) || g.grp_rows != 0) return VT_ERR_UNSUPPORTED;
  if (ACT == ACT_MUL && !g.R) return VT_ERR_NULL;
  const int grid = v9_grid(g9.tiles_m * g9.tiles_n);
  if (grid <= 0) return VT_ERR_HIP;
  auto kern = (g.R || ACT == ACT_MUL) ? gemm_nt_bf16_v9<ACT, true> : gemm_nt_bf16_v9<ACT, ACT == ACT_MUL>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V9_LDS_BYTES) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), V9_LDS_BYTES, stream, g9);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_gemm_v9_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream) {
  if (out_f32) return VT_ERR_UNSUPPORTED;
  switch (act) {
    case ACT_NONE: return launch_v9<ACT_NONE>(g, stream);
    case ACT_GELU: return launch_v9<ACT_GELU>(g, stream);
    case ACT_TANH: return launch_v9<ACT_TANH>(g, stream);
    case ACT_MUL: return launch_v9<ACT_MUL>(g, stream);
    default: return VT_ERR_UNSUPPORTED;
  }
}
