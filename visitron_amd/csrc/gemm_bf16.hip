// C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ R[M,N])   bf16 in, fp32 accumulate, bf16|fp32 out.
//
// The "NT" contraction of every nn.Linear on the hot path (weights are [out,in], K-contiguous):
//   K1  query/key/value        oscar/modeling_bert.py:43-45     (one packed [3H,H] weight)
//   K5  BertSelfOutput.dense   (+bias +residual)                called at oscar/modeling_bert.py:94
//   K6  BertIntermediate.dense (+bias +erf-GELU)                called at oscar/modeling_bert.py:119
//   K7  BertOutput.dense       (+bias +residual)                called at oscar/modeling_bert.py:120
//   K9  img_embedding + location_embeds  tasks/viewpoint_select/encoder.py:277-279 (K-concatenated)
//   K11/K12/K13/K14 pooler and heads     tasks/viewpoint_select/encoder.py:296,377,381,391
// With pre-transposed operands the same kernel serves dgrad (dX = dY . W) and wgrad (dW = dY^T . X).
//
// gfx950 design (one workgroup = 4 waves = one 128x128 output tile, BK = 64):
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging); LDS tiles are
//     [128 rows][64 k] bf16 with the 16-B chunk index XOR-swizzled by (row>>1)&7, applied on the
//     per-lane SOURCE address (the DMA destination is lane-linear), which makes every
//     ds_read_b128 fragment read conflict-free;
//   * two LDS buffers, one s_barrier per K-step: fragments of tile k are read to registers, then
//     the DMA of tile k+1 is issued and flies under the 32 MFMAs of tile k;
//   * MFMA v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (W rows feed the A port, activation
//     rows the B port) so that a lane's accumulators are 4 consecutive output columns of one
//     output row; W rows are permuted at staging time so the 4 N-subtiles of a wave interleave to
//     16 consecutive columns per lane -> the epilogue stores 32 contiguous bytes per lane and
//     bias / GELU / residual are applied in registers, no LDS round trip;
//   * block ids are remapped so that the tiles sharing an activation row-panel run on one XCD
//     (one L2).
// This file holds the 2-waves-per-SIMD kernels (v2, v4, v5, v6 below) and the dispatcher / shape table; the
// one-wave-per-SIMD 256x256 kernels with AGPR accumulators are in gemm_v7.hip (variants 15, 16).
#include <mutex>
#include "gemm_common.hpp"

// ================================================================================================
// v2: the same tile / fragment / epilogue design with
//   * a NSTAGE-deep LDS ring fed by global_load_lds, retired by a COUNTED s_waitcnt vmcnt(N) so the
//     DMA of the next NSTAGE-2 tiles stays in flight across the barrier;
//   * fragment reads as inline-asm ds_read_b128 (hipcc would otherwise put a vmcnt(0) in front of
//     every LDS read that follows an LDS-DMA and drain the ring), waited for by hand;
//   * BK = 64 (2 or 3 stages) or BK = 32 with 3 stages = 48 KiB LDS -> 3 workgroups per CU, which
//     also softens wave quantisation (768 slots instead of 512);
//   * grouped tile order: each XCD walks bands of 8 row-tiles, column-tile major inside a band, so
//     the ~64 tiles resident on an XCD share 8 activation panels and ~8 weight panels in its L2.
template <int BK>
__device__ __forceinline__ int swz(int row) {
  return BK == 64 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3);
}

__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int BK, int NSTAGE, int ACT, bool OUT_F32>
__global__ __launch_bounds__(256, (BK == 32 ? 3 : 2)) void gemm_nt_bf16_v2(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = BK / 8;             // 16-B chunks per tile row
  constexpr int ROWB = BK * 2;           // bytes per tile row
  constexpr int TILE = 128 * ROWB;       // bytes per operand tile
  constexpr int STAGE = 2 * TILE;        // X tile then W tile
  constexpr int NP = BK / 16;            // 1-KiB DMA pieces per operand per wave per stage
  constexpr int G = 2 * NP;              // DMA instructions per stage per wave
  constexpr int KS = BK / 32;            // MFMA k-substeps per stage

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-contiguous chunk of the grouped tile order
  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  const int t_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int band_tiles = 8 * g.tiles_n;
  const int band = t_id / band_tiles;
  const int within = t_id - band * band_tiles;
  const int rows_left = g.tiles_m - band * 8;
  const int band_h = rows_left < 8 ? rows_left : 8;
  const int bn = within / band_h;
  const int bm = band * 8 + (within - bn * band_h);
  const int m0 = bm * GEMM_BM, n0 = bn * GEMM_BN;

  const bf16_t* a_src[NP];
  const bf16_t* w_src[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = (wave * NP + i) * 64 + lane;
    const int row = p / CH;
    const int c = (p % CH) ^ swz<BK>(row);
    int am = m0 + row;
    am = am < g.M ? am : g.M - 1;
    a_src[i] = g.A + (long)am * g.lda + c * 8;
    int wnrow = n0 + (row & 64) + 16 * ((row >> 2) & 3) + 4 * ((row >> 4) & 3) + (row & 3);
    wnrow = wnrow < g.N ? wnrow : g.N - 1;
    w_src[i] = g.W + (long)wnrow * g.ldw + c * 8;
  }

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned x_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xr = 64 * wm + 16 * i + (lane & 15);
    x_off[i] = lds0 + xr * ROWB + (((lane >> 4) ^ swz<BK>(xr)) << 4);
    const int wr = 64 * wn + 16 * i + (lane & 15);
    w_off[i] = lds0 + TILE + wr * ROWB + (((lane >> 4) ^ swz<BK>(wr)) << 4);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;

  // prologue: tiles 0 .. NSTAGE-2
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) {
    if (st < nk) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        glds16(a_src[i] + st * BK, smem + st * STAGE + (wave * NP + i) * 1024);
        glds16(w_src[i] + st * BK, smem + st * STAGE + TILE + (wave * NP + i) * 1024);
      }
    }
  }

  int slot = 0;                 // ring slot of tile kt
  int pslot = NSTAGE - 1;       // ring slot the prefetch of tile kt+NSTAGE-1 goes to
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed (mine); tiles kt+1 .. kt+NSTAGE-2 may stay in flight
    if (NSTAGE > 2 && kt + NSTAGE - 2 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * G) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    const unsigned so = slot * STAGE;
    u32x4 xf[KS][4], wf[KS][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wf[0][i] = lds_read_b128(w_off[i] + so);
      xf[0][i] = lds_read_b128(x_off[i] + so);
    }

    if (kt + NSTAGE - 1 < nk) {
      char* nX = smem + pslot * STAGE;
      const int koff = (kt + NSTAGE - 1) * BK;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        glds16(a_src[i] + koff, nX + (wave * NP + i) * 1024);
        glds16(w_src[i] + koff, nX + TILE + (wave * NP + i) * 1024);
      }
    }

    if (KS == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wf[KS - 1][i] = lds_read_b128((w_off[i] + so) ^ 64);
        xf[KS - 1][i] = lds_read_b128((x_off[i] + so) ^ 64);
      }
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0][t]),
                                                             __builtin_bit_cast(bf16x8, xf[0][mt]), acc[mt][t], 0, 0, 0);
    if (KS == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[KS - 1][t]),
                                                               __builtin_bit_cast(bf16x8, xf[KS - 1][mt]), acc[mt][t], 0, 0, 0);
    }
    slot = (slot + 1 == NSTAGE) ? 0 : slot + 1;
    pslot = (pslot + 1 == NSTAGE) ? 0 : pslot + 1;
  }

  gemm_epilogue<ACT, OUT_F32>(g, acc, lane, m0 + 64 * wm, n0 + 64 * wn);
}

// ================================================================================================
// v4: 256 x BN tile (BN = 192 or 256), 8 waves as 4(M) x 2(N), wave tile 64 x BN/2.  The 128x128 tile is
// bound by the L2 -> LDS path (~50 GB/s per CU measured with the MFMAs removed: 64 FLOP per staged
// byte caps it near 750 TF/s); this tile stages 110-128 FLOP per byte and reads 0.38-0.42 LDS
// fragments per MFMA instead of 0.5.  Same LDS images, swizzle, swapped-operand MFMA and epilogue
// scheme as above; the output columns of a wave are handled in 64-column groups of T = 4 (or, for the
// last 32 columns of the 96-wide wave tile, T = 2) N-subtiles.
template <int T, int ACT, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_grp(const GemmArgs& g, f32x4 (&acc)[4][T], int lane, int row0, int col0) {
  // lane (j = lane&15, gq = lane>>4) owns rows row0+16mt+j and the 4T consecutive columns col0 + 4T*gq ..
  constexpr int W = 4 * T;
  const int gq = lane >> 4;
  const int nb = col0 + W * gq;
  if (nb >= g.N) return;
  const bool full = (nb + W <= g.N);
  float bv[W];
#pragma unroll
  for (int i = 0; i < W; ++i) bv[i] = 0.f;
  if (g.bias) {
#pragma unroll
    for (int i = 0; i < W; ++i)
      if (full || nb + i < g.N) bv[i] = g.bias[nb + i];
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = row0 + 16 * mt + (lane & 15);
    if (m >= g.M) continue;
    const long orow = g.grp_rows ? (long)(m / g.grp_rows) * g.grp_stride + (m % g.grp_rows) : (long)m;
    float v[W];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * t + e] = acc[mt][t][e] + bv[4 * t + e];
    if (full) {
      if (g.C2) {
#pragma unroll
        for (int h = 0; h < W / 8; ++h) {
          u32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float a0 = v[8 * h + 2 * i], a1 = v[8 * h + 2 * i + 1];
            o[i] = pack_bf16x2((ACT == ACT_GELU) ? gelu_erf_grad(a0) : a0, (ACT == ACT_GELU) ? gelu_erf_grad(a1) : a1);
          }
          *(u32x4*)(g.C2 + orow * g.ldc2 + nb + 8 * h) = o;
        }
      }
#pragma unroll
      for (int i = 0; i < W; ++i) v[i] = apply_act<ACT>(v[i]);
      if (g.drop.thresh) {
        const uint32_t e0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)nb;
        vt_drop_run<W>(g.drop, e0, v);
      }
      if (g.R) {
#pragma unroll
        for (int h = 0; h < W / 8; ++h) {
          const u32x4 rr = *(const u32x4*)(g.R + orow * g.ldr + nb + 8 * h);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float r0 = bf16lo(rr[i]), r1 = bf16hi(rr[i]);
            v[8 * h + 2 * i] = (ACT == ACT_MUL) ? v[8 * h + 2 * i] * r0 : v[8 * h + 2 * i] + r0;
            v[8 * h + 2 * i + 1] = (ACT == ACT_MUL) ? v[8 * h + 2 * i + 1] * r1 : v[8 * h + 2 * i + 1] + r1;
          }
        }
      }
      if (OUT_F32) {
        f32x4* cp = (f32x4*)((float*)g.C + orow * g.ldc + nb);
#pragma unroll
        for (int i = 0; i < W / 4; ++i) cp[i] = (f32x4){v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
      } else {
#pragma unroll
        for (int h = 0; h < W / 8; ++h) {
          u32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(v[8 * h + 2 * i], v[8 * h + 2 * i + 1]);
          *(u32x4*)((bf16_t*)g.C + orow * g.ldc + nb + 8 * h) = o;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < W; ++i) {
        if (nb + i < g.N) {
          if (g.C2) g.C2[orow * g.ldc2 + nb + i] = f32_to_bf16((ACT == ACT_GELU) ? gelu_erf_grad(v[i]) : v[i]);
          float x = apply_act<ACT>(v[i]);
          if (g.drop.thresh) x = vt_keep(g.drop, (uint32_t)m * (uint32_t)g.N + (uint32_t)(nb + i)) ? x * g.drop.scale : 0.f;
          if (g.R) {
            const float rr = bf16_to_f32(g.R[orow * g.ldr + nb + i]);
            x = (ACT == ACT_MUL) ? x * rr : x + rr;
          }
          if (OUT_F32) ((float*)g.C)[orow * g.ldc + nb + i] = x;
          else ((bf16_t*)g.C)[orow * g.ldc + nb + i] = f32_to_bf16(x);
        }
      }
    }
  }
}

template <int BN, int ACT, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_v4(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BK = 64;
  constexpr int XT = BM * 128;          // X tile bytes
  constexpr int WT = BN * 128;          // W tile bytes
  constexpr int STAGE = XT + WT;
  constexpr int WN = BN / 2;            // columns per wave
  constexpr int NT = WN / 16;           // N-subtiles per wave (6 or 8)
  constexpr int NPX = 4;                // X DMA pieces per wave per stage (32 pieces / 8 waves)
  constexpr int NPW = BN / 64;          // W DMA pieces per wave per stage (3 or 4)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  const int t_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int band_tiles = 4 * g.tiles_n;  // bands of 4 row-tiles (1024 rows)
  const int band = t_id / band_tiles;
  const int within = t_id - band * band_tiles;
  const int rows_left = g.tiles_m - band * 4;
  const int band_h = rows_left < 4 ? rows_left : 4;
  const int bn = within / band_h;
  const int bm = band * 4 + (within - bn * band_h);
  const int m0 = bm * BM, n0 = bn * BN;

  const bf16_t* a_src[NPX];
  const bf16_t* w_src[NPW];
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int p = (wave * NPX + i) * 64 + lane;
    const int row = p >> 3;
    const int c = (p & 7) ^ ((row >> 1) & 7);
    int am = m0 + row;
    am = am < g.M ? am : g.M - 1;
    a_src[i] = g.A + (long)am * g.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = (wave * NPW + i) * 64 + lane;
    const int row = p >> 3;                       // LDS row 0 .. BN-1
    const int c = (p & 7) ^ ((row >> 1) & 7);
    const int wv = row / WN, rr = row - wv * WN;  // owning wave column, row inside its range
    const int gb = rr < 64 ? 0 : 64;              // 64-column group base
    const int Tg = (rr < 64 || WN == 128) ? 4 : 2;
    const int lr = rr - gb, t = lr >> 4, i16 = lr & 15;
    int wnrow = n0 + wv * WN + gb + (4 * Tg) * (i16 >> 2) + 4 * t + (i16 & 3);
    wnrow = wnrow < g.N ? wnrow : g.N - 1;
    w_src[i] = g.W + (long)wnrow * g.ldw + c * 8;
  }

  int x_off[4], w_off[NT];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xr = 64 * wm + 16 * i + (lane & 15);
    x_off[i] = xr * 128 + (((lane >> 4) ^ ((xr >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int wr = WN * wn + 16 * i + (lane & 15);
    w_off[i] = XT + wr * 128 + (((lane >> 4) ^ ((wr >> 1) & 7)) << 4);
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  auto stage = [&](int kt, int buf) {
    char* d = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NPX; ++i) glds16(a_src[i] + kt * BK, d + (wave * NPX + i) * 1024);
#pragma unroll
    for (int i = 0; i < NPW; ++i) glds16(w_src[i] + kt * BK, d + XT + (wave * NPW + i) * 1024);
  };

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* sb = smem + (kt & 1) * STAGE;
    bf16x8 xf[4], wf[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wf[i] = *(const bf16x8*)(sb + w_off[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8*)(sb + x_off[i]);
    bf16x8 xg[4], wg[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wg[i] = *(const bf16x8*)(sb + (w_off[i] ^ 64));
#pragma unroll
    for (int i = 0; i < 4; ++i) xg[i] = *(const bf16x8*)(sb + (x_off[i] ^ 64));
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[mt], acc[mt][t], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[t], xg[mt], acc[mt][t], 0, 0, 0);
  }

  // epilogue: 64-column groups of the wave's columns
  {
    f32x4 a0[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) a0[mt][t] = acc[mt][t];
    gemm_epilogue_grp<4, ACT, OUT_F32>(g, a0, lane, m0 + 64 * wm, n0 + WN * wn);
  }
  if (NT == 8) {
    f32x4 a1[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) a1[mt][t] = acc[mt][(NT == 8 ? 4 : 0) + t];
    gemm_epilogue_grp<4, ACT, OUT_F32>(g, a1, lane, m0 + 64 * wm, n0 + WN * wn + 64);
  } else {
    f32x4 a1[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) a1[mt][t] = acc[mt][4 + t];
    gemm_epilogue_grp<2, ACT, OUT_F32>(g, a1, lane, m0 + 64 * wm, n0 + WN * wn + 64);
  }
}

// ================================================================================================
// v5: 256x256 tile, BK = 32, 4-stage LDS ring, 8 waves as 2(M) x 4(N), wave tile 128 x 64, two phases
// per K-step.  LDS traffic (DMA writes + fragment reads) is what bounds the kernels above, so here the
// fragment reads of the NEXT phase and the DMA issue are placed in front of each phase's 16 MFMAs and
// retired by counted waits only:
//   phase 0: read A(rows 64..127 of the wave) of tile k | issue X pieces of tile k+3 | lgkmcnt(4) | 16 MFMA (rows 0..63)
//   phase 1: vmcnt(6): tile k+1 landed | lgkmcnt(0) | barrier | read A(rows 0..63) + W of tile k+1 |
//            issue W pieces of tile k+3 | 16 MFMA (rows 64..127)
// One barrier per K-step; every wave has retired all its LDS reads before it, so the DMA of tile k+3
// may overwrite the stage of tile k-1.  Fragment reads are inline asm (hipcc would drain the ring in
// front of compiler-visible LDS reads); DMA past the last tile re-loads the last tile so the vmcnt
// arithmetic is uniform.
#define V5_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))

template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_v5(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, BK = 32, STAGE = 32768, XT = 16384;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  const int t_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int band_tiles = 4 * g.tiles_n;
  const int band = t_id / band_tiles;
  const int within = t_id - band * band_tiles;
  const int rows_left = g.tiles_m - band * 4;
  const int band_h = rows_left < 4 ? rows_left : 4;
  const int bn = within / band_h;
  const int bm = band * 4 + (within - bn * band_h);
  const int m0 = bm * BM, n0 = bn * BN;

  // DMA pieces: 1 KiB = 16 rows x 64 B; wave w moves pieces 2w, 2w+1 of each operand tile
  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (2 * wave + i) + (lane >> 2);
    const int c = (lane & 3) ^ ((-(row >> 2)) & 3);
    int am = m0 + row;
    am = am < g.M ? am : g.M - 1;
    a_src[i] = g.A + (long)am * g.lda + c * 8;
    int wnrow = n0 + (row & 192) + 16 * ((row >> 2) & 3) + 4 * ((row >> 4) & 3) + (row & 3);
    wnrow = wnrow < g.N ? wnrow : g.N - 1;
    w_src[i] = g.W + (long)wnrow * g.ldw + c * 8;
  }
  const int nk = g.K / BK;
  auto dma_x = [&](int kt, int st) {
    const int kk = (kt < nk ? kt : nk - 1) * BK;
    glds16(a_src[0] + kk, smem + st * STAGE + (2 * wave) * 1024);
    glds16(a_src[1] + kk, smem + st * STAGE + (2 * wave + 1) * 1024);
  };
  auto dma_w = [&](int kt, int st) {
    const int kk = (kt < nk ? kt : nk - 1) * BK;
    glds16(w_src[0] + kk, smem + st * STAGE + XT + (2 * wave) * 1024);
    glds16(w_src[1] + kk, smem + st * STAGE + XT + (2 * wave + 1) * 1024);
  };

  // fragment addresses (stage 0): the swizzle depends on (lane&15)>>2 only, so sub-tiles are immediates
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int i16 = lane & 15;
  const unsigned fsw = (unsigned)((((lane >> 4) ^ ((-(i16 >> 2)) & 3)) << 4));
  const unsigned a_addr0 = lds0 + (128 * wm + i16) * 64 + fsw;        // + 1024 * mt
  const unsigned w_addr0 = lds0 + XT + (64 * wn + i16) * 64 + fsw;    // + 1024 * t

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 A0[4], A1[4], Wa[4], Wb[4];

  // prologue: tiles 0,1,2 in flight; tile 0 landed; first fragments read
  dma_x(0, 0); dma_w(0, 0);
  dma_x(1, 1); dma_w(1, 1);
  dma_x(2, 2); dma_w(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  V5_DSR(A0[0], a_addr0, 0); V5_DSR(A0[1], a_addr0, 1024); V5_DSR(A0[2], a_addr0, 2048); V5_DSR(A0[3], a_addr0, 3072);
  V5_DSR(Wa[0], w_addr0, 0); V5_DSR(Wa[1], w_addr0, 1024); V5_DSR(Wa[2], w_addr0, 2048); V5_DSR(Wa[3], w_addr0, 3072);

  auto kstep = [&](int kt, u32x4 (&Wc)[4], u32x4 (&Wn)[4]) {
    const unsigned so = (unsigned)(kt & 3) * STAGE;
    // ---- phase 0 ----
    {
      const unsigned aa = a_addr0 + so;
      V5_DSR(A1[0], aa, 4096); V5_DSR(A1[1], aa, 5120); V5_DSR(A1[2], aa, 6144); V5_DSR(A1[3], aa, 7168);
    }
    dma_x(kt + 3, (kt + 3) & 3);
    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wc[t]),
                                                             __builtin_bit_cast(bf16x8, A0[mt]), acc[mt][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 1 ----
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) {
      const unsigned sn = (unsigned)((kt + 1) & 3) * STAGE;
      const unsigned aa = a_addr0 + sn, ww = w_addr0 + sn;
      V5_DSR(A0[0], aa, 0); V5_DSR(A0[1], aa, 1024); V5_DSR(A0[2], aa, 2048); V5_DSR(A0[3], aa, 3072);
      V5_DSR(Wn[0], ww, 0); V5_DSR(Wn[1], ww, 1024); V5_DSR(Wn[2], ww, 2048); V5_DSR(Wn[3], ww, 3072);
    } else {
      // keep the LDS-op count of the step uniform for the next (non-existent) lgkmcnt(4): nothing to do
    }
    dma_w(kt + 3, (kt + 3) & 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[4 + mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wc[t]),
                                                                 __builtin_bit_cast(bf16x8, A1[mt]), acc[4 + mt][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int kt = 0; kt < nk; kt += 2) {
    kstep(kt, Wa, Wb);
    if (kt + 1 < nk) kstep(kt + 1, Wb, Wa);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the tail DMAs before the LDS goes away

  {
    f32x4 lo[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) lo[mt][t] = acc[mt][t];
    gemm_epilogue<ACT, OUT_F32>(g, lo, lane, m0 + 128 * wm, n0 + 64 * wn);
  }
  {
    f32x4 hi[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) hi[mt][t] = acc[4 + mt][t];
    gemm_epilogue<ACT, OUT_F32>(g, hi, lane, m0 + 128 * wm + 64, n0 + 64 * wn);
  }
}

// ================================================================================================
// v6: the v2 structure (128x128 tile, BK 64, 2-stage ring) with EIGHT waves of 32x64 each: <= 128 VGPRs,
// so two workgroups put 4 waves on every SIMD.  PMC showed the MFMA pipes only 35-41 % busy with two
// waves per SIMD (LDS 10 % busy, no bank conflicts): more, smaller waves cover each other's barrier and
// DMA-issue time at the price of 0.75 instead of 0.5 LDS fragment reads per MFMA.
template <int ACT, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_m2(const GemmArgs& g, f32x4 (&acc)[2][4], int lane, int row0, int col0) {
  // same as gemm_epilogue for a 32-row wave tile: pad to 4 row-subtiles with rows >= M masked off
  f32x4 a4[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { a4[0][t] = acc[0][t]; a4[1][t] = acc[1][t]; a4[2][t] = acc[0][t]; a4[3][t] = acc[0][t]; }
  GemmArgs g2 = g;
  const int lim = row0 + 32;
  g2.M = g.M < lim ? g.M : lim;   // rows row0+32.. (the two padding subtiles) fall outside and are skipped
  gemm_epilogue<ACT, OUT_F32>(g2, a4, lane, row0, col0);
}

template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(512, 4) void gemm_nt_bf16_v6(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = 128 * 128, STAGE = 2 * TILE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves, wave tile 32 x 64

  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  const int t_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int band_tiles = 8 * g.tiles_n;
  const int band = t_id / band_tiles;
  const int within = t_id - band * band_tiles;
  const int rows_left = g.tiles_m - band * 8;
  const int band_h = rows_left < 8 ? rows_left : 8;
  const int bn = within / band_h;
  const int bm = band * 8 + (within - bn * band_h);
  const int m0 = bm * GEMM_BM, n0 = bn * GEMM_BN;

  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (wave * 2 + i) * 64 + lane;
    const int row = p >> 3;
    const int c = (p & 7) ^ ((row >> 1) & 7);
    int am = m0 + row;
    am = am < g.M ? am : g.M - 1;
    a_src[i] = g.A + (long)am * g.lda + c * 8;
    int wnrow = n0 + (row & 64) + 16 * ((row >> 2) & 3) + 4 * ((row >> 4) & 3) + (row & 3);
    wnrow = wnrow < g.N ? wnrow : g.N - 1;
    w_src[i] = g.W + (long)wnrow * g.ldw + c * 8;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned x_off[2], w_off[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int xr = 32 * wm + 16 * i + (lane & 15);
    x_off[i] = lds0 + xr * 128 + (((lane >> 4) ^ ((xr >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int wr = 64 * wn + 16 * i + (lane & 15);
    w_off[i] = lds0 + TILE + wr * 128 + (((lane >> 4) ^ ((wr >> 1) & 7)) << 4);
  }
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    glds16(a_src[i], smem + (wave * 2 + i) * 1024);
    glds16(w_src[i], smem + TILE + (wave * 2 + i) * 1024);
  }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned so = (kt & 1) * STAGE;
    u32x4 xf[2][2], wf[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[0][i] = lds_read_b128(w_off[i] + so);
#pragma unroll
    for (int i = 0; i < 2; ++i) xf[0][i] = lds_read_b128(x_off[i] + so);
    if (kt + 1 < nk) {
      char* nX = smem + ((kt + 1) & 1) * STAGE;
      const int koff = (kt + 1) * 64;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        glds16(a_src[i] + koff, nX + (wave * 2 + i) * 1024);
        glds16(w_src[i] + koff, nX + TILE + (wave * 2 + i) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[1][i] = lds_read_b128((w_off[i] + so) ^ 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) xf[1][i] = lds_read_b128((x_off[i] + so) ^ 64);
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0][t]),
                                                             __builtin_bit_cast(bf16x8, xf[0][mt]), acc[mt][t], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[1][t]),
                                                             __builtin_bit_cast(bf16x8, xf[1][mt]), acc[mt][t], 0, 0, 0);
  }
  gemm_epilogue_m2<ACT, OUT_F32>(g, acc, lane, m0 + 32 * wm, n0 + 64 * wn);
}

// ---- launchers ---------------------------------------------------------------------------------
// variant: 1 = 128x128 tile, 4 waves, BK 64, 2-stage ring (v2); 14 = the same tile with 8 waves of 32x64 (v6);
//          9 / 10 = 256x192 / 256x256 tile, 8 waves (v4); 11 = 256x256, BK 32, 4-stage ring, phased (v5);
//          15 / 16 = 256x256 tile, 4 waves of 128x128 with AGPR accumulators, one tile per workgroup / persistent
//          (gemm_v7.hip); 18 .. 21 = the persistent kernel on 224- / 192- / 160- / 128-row tiles (fewer, better balanced rounds when the
//          256-row tiling leaves the last round mostly empty).
int vt_gemm_v7_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn);  // gemm_v7.hip (tile height 32 * mtn)
int vt_gemm_v8_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn, bool shared_tiles = false);  // gemm_v7.hip (persistent; tile height 32 * mtn)
#ifdef VT_EXPERIMENTAL_GEMM   // tools/experiments (make gemmlab): the measured-negative redesigns of round 4, variants 24 .. 27; not in the product
int vt_gemm_v10_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream);           // gemm_v10.hip (two persistent 256x128-tile workgroups per CU)
int vt_gemm_v11_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream);           // gemm_v11.hip (eight waves on shared 256x256 stages)
int vt_gemm_v12_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn);  // gemm_v12.hip (short tiles on three operand stages)
#endif
static void* g_gemm_trace = nullptr;
void vt_gemm_set_trace(void* p) { g_gemm_trace = p; }
static int g_gemm_variant = -1;  // -1: table / heuristic (tuning hook only; set through vt_debug_set_gemm_variant)
static int g_gemm_tail_split = 0;   // opt-in: the persistent kernel's half-empty last round goes to the 128x128-tile kernel
// v >= 0: force a variant; -1: automatic; -2: automatic WITH the tail launch (measured +0.4 % on the step: opt-in)
void vt_gemm_set_variant(int v) {
  if (v == -2) { g_gemm_tail_split = 1; g_gemm_variant = -1; return; }
  if (v == -1) g_gemm_tail_split = 0;
  g_gemm_variant = v;
}

// Shape -> variant table filled by the host-side autotuner (visitron_amd.ops.autotune_linear) before
// the shapes are used; read-only afterwards.  One table PER DEVICE (the calling thread's current device), guarded by a
// mutex: threads driving different GPUs of one process (torch.nn.DataParallel) tune and look up independently.
// The key is (M, N, K, what the epilogue does): kind = act | residual << 4 | second output << 5 | fp32 output << 6 |
// deferred-LayerNorm mode << 8 -- a plain dgrad and the out-proj with its residual are the same (M, N, K, act) and not
// the same kernel time (VT_TUNE_KIND below is what vt_gemm_dispatch looks up; vt_gemm_tune takes the same number).
#define VT_TUNE_KIND(act, has_r, has_c2, out_f32, ln_mode) \
  ((act) | ((has_r) ? 16 : 0) | ((has_c2) ? 32 : 0) | ((out_f32) ? 64 : 0) | ((ln_mode) << 8))
struct TuneEntry { int M, N, K, kind, variant; };
struct TuneTable { TuneEntry e[512]; int n; };
static TuneTable g_tune_dev[VT_MAX_DEVICES];
static std::mutex g_tune_mu;
void vt_gemm_tune_set(int M, int N, int K, int kind, int variant) {
  const int dev = vt_current_device();
  if (dev < 0) return;
  std::lock_guard<std::mutex> lock(g_tune_mu);
  TuneEntry* g_tune = g_tune_dev[dev].e;
  int& g_ntune = g_tune_dev[dev].n;
  for (int i = 0; i < g_ntune; ++i)
    if (g_tune[i].M == M && g_tune[i].N == N && g_tune[i].K == K && g_tune[i].kind == kind) { g_tune[i].variant = variant; return; }
  if (g_ntune < 512) g_tune[g_ntune++] = TuneEntry{M, N, K, kind, variant};
}
// ln_only: the caller can only run the 256x256-tile kernels (deferred-LayerNorm epilogues)
int vt_gemm_pick_variant(int M, int N, int K, int kind) {
  const int dev = vt_current_device();
  const bool ln_only = (kind >> 8) != 0;
  {
    std::lock_guard<std::mutex> lock(g_tune_mu);
    const TuneEntry* g_tune = g_tune_dev[dev < 0 ? 0 : dev].e;
    const int g_ntune = dev < 0 ? 0 : g_tune_dev[dev].n;
    for (int i = 0; i < g_ntune; ++i)
      if (g_tune[i].M == M && g_tune[i].N == N && g_tune[i].K == K && g_tune[i].kind == kind) return g_tune[i].variant;
    // a row count the tuner has not seen (compacted batches change it every step): the entry of the same (N, K, kind)
    // whose M is nearest, within 25 %; failing that the same shape under another epilogue of the same activation
    for (int pass = 0; pass < 2; ++pass) {
      int best = -1;
      long best_d = 0;
      for (int i = 0; i < g_ntune; ++i) {
        const bool same = pass == 0 ? g_tune[i].kind == kind
                                    : ((g_tune[i].kind & 15) == (kind & 15) && (g_tune[i].kind >> 8) == (kind >> 8));
        if (g_tune[i].N == N && g_tune[i].K == K && same) {
          const long d = g_tune[i].M > M ? g_tune[i].M - M : M - g_tune[i].M;
          if (4 * d <= g_tune[i].M && (best < 0 || d < best_d)) { best = i; best_d = d; }
        }
      }
      if (best >= 0) return g_tune[best].variant;
    }
  }
  if (ln_only) {   // untuned: persistent 256-row tiles when they fill the chip, else one tile per workgroup
    const int cus = vt_device_cus();
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    return t256 >= 2L * (cus > 0 ? cus : 256) ? 16 : 23;
  }
  // heuristic: wave-quantisation efficiency x measured relative rate of each tile
  auto eff = [](long tiles, int slots) { const double w = (double)tiles / slots; return w / (double)((long)(w + 0.999)); };
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const long t192 = (long)((M + 255) / 256) * ((N + 191) / 192);
  const double gelu_pen = ((kind & 15) == ACT_GELU) ? 0.84 : 1.0;
  const double s1 = eff(t128, 512) * 1.0;
  const double s9 = (N >= 192 && M >= 256) ? eff(t192, 256) * 1.12 * gelu_pen : 0.0;
  return s9 > s1 ? 9 : GEMM_DEFAULT_VARIANT;
}

template <typename K>
static int launch_kernel_v4(K kern, GemmArgs g, int bn, hipStream_t stream) {
  const int lds_bytes = 2 * (256 + bn) * 128;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return VT_ERR_HIP;
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + bn - 1) / bn;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), lds_bytes, stream, g);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

template <typename K>
static int launch_kernel(K kern, const GemmArgs& g, int lds_bytes, hipStream_t stream) {
  // hipFuncSetAttribute is idempotent and cheap; called per launch to stay free of per-kernel statics
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(256), lds_bytes, stream, g);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// ---- variant 33: split-K with the WHOLE epilogue behind the reduction (round 6) ----------------------------------------
// Small batches leave a long-K product a handful of 256x256 tiles, each a chain of K / 64 dependent K-steps of ~0.8 us whatever
// M is (the reference's own batch sizes: 2 x 767 or 8 x 511 rows per GPU; [912, 768] x 3 072 takes 38 us for 3.6 GFLOP).
// ksplit copies of the tile list of the one-tile-per-workgroup kernel each take a share of the K-steps and leave their
// accumulators RAW in the caller's GEMM workspace (vt_gemm_set_workspace; AGPR-sourced stores, no epilogue); one further
// kernel sums a tile's copies IN ORDER (deterministic) and runs the ordinary register epilogue -- bias, activation, dropout,
// residual / rebuilt-LayerNorm residual, second output, fp16 / bf16 / fp32 store, row remap -- on the sums (epi_row_direct, the
// code the register-epilogue kernels run: every epilogue kind is served).  The autotuner decides per (M, N, K, kind).
int vt_gemm_splitk_tiles_launch(const GemmArgs& g, int act, int out_f32, int ks, hipStream_t stream);   // gemm_v7.hip
template <int ACT, bool OUT_F32>
static int launch_splitk_epi(const GemmArgs& g, hipStream_t stream) {
  const int cus = vt_device_cus();
  if (cus <= 0) return VT_ERR_HIP;
  const long tiles = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
  const int nk = g.K >> 6;
  // copies of the tile list: a copy's chain is nk / ks K-steps of ~0.8 us, the epilogue kernel reads ks x tiles x 256 KiB at
  // ~5 TB/s (0.05 us per copy and tile): the sum is smallest near ks = sqrt(16 nk / tiles) -- within one round of workgroups,
  // at least three K-steps per copy
  long ks = (long)(sqrtf(16.0f * (float)nk / (float)tiles) + 0.5f);
  if (ks > cus / tiles) ks = cus / tiles;
  if (ks > nk / 3) ks = nk / 3;
  if (ks > 16) ks = 16;
  if (ks < 2 || (g.K & 63)) return VT_ERR_UNSUPPORTED;
  return vt_gemm_splitk_tiles_launch(g, ACT, OUT_F32 ? 1 : 0, (int)ks, stream);
}

template <int ACT, bool OUT_F32>
static int launch_gemm(const GemmArgs& g, int variant, hipStream_t stream) {
  switch (variant) {
    case 1: return launch_kernel(gemm_nt_bf16_v2<64, 2, ACT, OUT_F32>, g, 2 * 32768, stream);
    // 35: the same 128x128-tile kernel on a ring of THREE stages (96 KiB: one workgroup per CU).  For launches of fewer tiles
    // than CUs, where a tile's K-steps are a chain paced by the load latency and two stages keep ONE K-tile in flight: the
    // long-K products of small batches (round 6; four stages measured no better than three)
    case 35: return launch_kernel(gemm_nt_bf16_v2<64, 3, ACT, OUT_F32>, g, 3 * 32768, stream);
    case 9: return launch_kernel_v4(gemm_nt_bf16_v4<192, ACT, OUT_F32>, g, 192, stream);
    case 10: return launch_kernel_v4(gemm_nt_bf16_v4<256, ACT, OUT_F32>, g, 256, stream);
    case 11: {
      GemmArgs g5 = g;
      g5.tiles_m = (g.M + 255) / 256;
      g5.tiles_n = (g.N + 255) / 256;
      auto kern = gemm_nt_bf16_v5<ACT, OUT_F32>;
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768) != hipSuccess) return VT_ERR_HIP;
      hipLaunchKernelGGL(kern, dim3(g5.tiles_m * g5.tiles_n), dim3(512), 4 * 32768, stream, g5);
      return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
    }
    case 14: {
      auto kern = gemm_nt_bf16_v6<ACT, OUT_F32>;
      hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), 2 * 32768, stream, g);
      return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
    }
    case 15: return vt_gemm_v7_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 8);
    case 22: return vt_gemm_v7_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 7);   // one tile per workgroup, 224-row tiles
    case 23: return vt_gemm_v7_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 6);   // ... 192-row tiles (mid-size batches: one fuller round)
    case 16: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 8);
    case 18: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 7);   // the persistent kernel on 224-row tiles
    case 19: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 6);   // ... on 192-row tiles
    case 20: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 5);   // ... on 160-row tiles
    case 21: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 4);   // ... on 128-row tiles (small batches)
    // 28 .. 32: the persistent kernel (256 .. 128-row tiles) with its left-over tiles SHARED along K among the workgroups a
    // last round would leave idle (GemmArgs::sk_parts; needs vt_gemm_set_workspace)
    case 33: return launch_splitk_epi<ACT, OUT_F32>(g, stream);
    case 28: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 8, true);
    case 29: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 7, true);
    case 30: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 6, true);
    case 31: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 5, true);
    case 32: return vt_gemm_v8_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 4, true);
#ifdef VT_EXPERIMENTAL_GEMM
    case 24: return vt_gemm_v10_launch(g, ACT, OUT_F32 ? 1 : 0, stream);     // two co-resident persistent workgroups per CU, 256x128 tiles
    case 25: return vt_gemm_v11_launch(g, ACT, OUT_F32 ? 1 : 0, stream);     // eight waves (two groups of four) on shared 256x256 stages
    case 26: return vt_gemm_v12_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 5);  // persistent, 160-row tiles on three operand stages
    case 27: return vt_gemm_v12_launch(g, ACT, OUT_F32 ? 1 : 0, stream, 4);  // ... 128-row tiles
#endif
    default: return VT_ERR_UNSUPPORTED;
  }
}

// Host entry used by the C ABI (capi.hip).  Returns a VT_* code; never synchronises.
int vt_gemm_dispatch(const void* A, long lda, const void* W, long ldw, const float* bias, const void* R, long ldr,
                     void* C, long ldc, int M, int N, int K, int act, int out_mode, int grp_rows, int grp_stride,
                     hipStream_t stream, void* C2 = nullptr, long ldc2 = 0, const DropCfg* drop = nullptr,
                     const VtLnResidual* rln = nullptr) {
  // out_mode: bit 0 fp32 output; bit 1 C written as fp16 (saturating) instead of bf16; bit 2 the residual R holds fp16
  // (GemmArgs::c_f16 / r_f16: the training layer's higher-precision residual stream)
  const int out_f32 = out_mode & 1, c_f16 = (out_mode >> 1) & 1, r_f16 = (out_mode >> 2) & 1;
  if ((out_mode & ~7) || (c_f16 && out_f32) || (r_f16 && (!R || act == ACT_MUL))) return VT_ERR_UNSUPPORTED;
  if (!A || !W || !C) return VT_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K % GEMM_BK) != 0) return VT_ERR_BAD_SHAPE;
  if ((lda % 8) || (ldw % 8) || (R && (ldr % 8)) || (ldc % (out_f32 ? 4 : 8)) || (C2 && (ldc2 % 8))) return VT_ERR_BAD_ALIGN;
  if (act == ACT_MUL && !R) return VT_ERR_NULL;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)R | (uintptr_t)bias) & 15) return VT_ERR_BAD_ALIGN;
  if (grp_rows < 0 || (grp_rows > 0 && grp_stride < grp_rows)) return VT_ERR_BAD_SHAPE;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.R = (const bf16_t*)R; g.C = C;
  g.lda = lda; g.ldw = ldw; g.ldr = ldr; g.ldc = ldc; g.C2 = (bf16_t*)C2; g.ldc2 = ldc2;
  g.M = M; g.N = N; g.K = K;
  g.grp_rows = grp_rows; g.grp_stride = grp_stride;
  if (drop) g.drop = *drop; else { g.drop.thresh = 0; g.drop.seed = 0; g.drop.scale = 1.0f; }
  g.trace = (unsigned long long*)g_gemm_trace;
  g.ln_mode = 0; g.ln_np = 0; g.ln_rows = 0; g.ln_inv_n = 0.f; g.ln_eps = 0.f; g.ln_stats = nullptr; g.colv = nullptr;
  g.Rs = nullptr; g.Cs = nullptr; g.stats_out = nullptr; g.ldrs = 0; g.ldcs = 0; g.ksplit = 0; g.reverse = 0; g.c_plane = 0;
  g.r_f16 = r_f16; g.c_f16 = c_f16;
  if (rln) {   // residual = LayerNorm(R) from saved row statistics (GemmArgs::r_mean)
    if (!rln->mean || !rln->rstd || !rln->gamma || !rln->beta) return VT_ERR_NULL;
    if (!r_f16 || act != ACT_NONE || grp_rows || (N % 16)) return VT_ERR_UNSUPPORTED;
    if ((((uintptr_t)rln->gamma | (uintptr_t)rln->beta) & 15) || (((uintptr_t)rln->mean | (uintptr_t)rln->rstd) & 3)) return VT_ERR_BAD_ALIGN;
    g.r_mean = rln->mean; g.r_rstd = rln->rstd; g.r_gamma = rln->gamma; g.r_beta = rln->beta;
  }
  if (g.drop.thresh && (long)M * N >= (1L << 32)) return VT_ERR_UNSUPPORTED;
  g.tiles_m = (M + GEMM_BM - 1) / GEMM_BM;
  g.tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
  int variant = g_gemm_variant >= 0 ? g_gemm_variant
                                    : vt_gemm_pick_variant(M, N, K, VT_TUNE_KIND(act, R != nullptr, C2 != nullptr, out_f32 != 0, 0));
  if ((r_f16 || c_f16) && (variant == 9 || variant == 10)) variant = 1;   // the 256 x 192 / 256 kernels' grouped epilogue reads / writes bf16 only
  auto launch = [&](const GemmArgs& ga, int v) {
    switch (act * 2 + (out_f32 ? 1 : 0)) {
      case 0: return launch_gemm<ACT_NONE, false>(ga, v, stream);
      case 1: return launch_gemm<ACT_NONE, true>(ga, v, stream);
      case 2: return launch_gemm<ACT_GELU, false>(ga, v, stream);
      case 3: return launch_gemm<ACT_GELU, true>(ga, v, stream);
      case 4: return launch_gemm<ACT_TANH, false>(ga, v, stream);
      case 5: return launch_gemm<ACT_TANH, true>(ga, v, stream);
      case 6: return launch_gemm<ACT_MUL, false>(ga, v, stream);
      default: return VT_ERR_UNSUPPORTED;
    }
  };
  // Tail rows of the persistent kernel.  T = 256x256 tiles over `cus` CUs: F full rounds and a last round with R tiles.
  // When that last round is at most half full, its rows go to the 128x128-tile kernel instead (two workgroups per
  // CU: R * 4 <= 2 * cus small tiles = one round of 0.58 of the big tile's time); the persistent kernel keeps the rows
  // of the F full rounds.  Dropout: element index = m * N + n and the hash is linear in (index/2 + seed), so the row
  // offset of the second launch is a seed offset.  Not when a variant is forced (tuning) or rows are remapped.
  if (variant == 16 && g_gemm_variant < 0 && g_gemm_tail_split && grp_rows == 0 && (N & 1) == 0) {
    const int cus = vt_device_cus();
    const long tn = (N + 255) / 256, tm = (M + 255) / 256, T = tm * tn;
    if (cus > 0 && T > cus) {
      const long F = T / cus, Rt = T - F * cus;
      const long m1 = (F * cus) / tn;           // row tiles of the full rounds
      const long M1 = m1 * 256;
      if (Rt > 0 && 2 * Rt <= cus && M1 > 0 && M1 < M) {
        GemmArgs g1 = g, g2 = g;
        g1.M = (int)M1;
        g1.tiles_m = (int)((M1 + GEMM_BM - 1) / GEMM_BM);
        g2.M = (int)(M - M1);
        g2.tiles_m = (g2.M + GEMM_BM - 1) / GEMM_BM;
        g2.A = g.A + M1 * lda;
        if (g.R) g2.R = g.R + M1 * ldr;
        if (g.r_mean) { g2.r_mean = g.r_mean + M1; g2.r_rstd = g.r_rstd + M1; }
        g2.C = out_f32 ? (void*)((float*)C + M1 * ldc) : (void*)((bf16_t*)C + M1 * ldc);
        if (g.C2) g2.C2 = g.C2 + M1 * ldc2;
        g2.drop.seed = g.drop.seed + (uint32_t)((M1 * N) >> 1);
        const int rc = launch(g1, 16);
        if (rc) return rc;
        return launch(g2, 1);
      }
    }
  }
  const int rc = launch(g, variant);
  // a table entry tuned at a neighbouring row count may name the split-K variant where this M leaves it nothing to split
  // (or no workspace is registered on this device): the default kernel instead -- never when the variant was forced
  if (rc == VT_ERR_UNSUPPORTED && variant == 33 && g_gemm_variant < 0) return launch(g, GEMM_DEFAULT_VARIANT);
  return rc;
}

// ---- deferred-LayerNorm GEMMs (GemmArgs::ln_mode; gemm_v7_ln.hip) --------------------------------------------------
int vt_gemm_ln_launch(const GemmArgs& g, int act, int variant, hipStream_t stream);
// mode 1: C = act(rstd_r (A W^T - mean_r colv) + bias), statistics of A's rows (row length K) from stats_in;
// mode 2: v = A W^T + bias + colv * ((Rs - mean_r) rstd_r) (row length N) -> Cs (fp16), C (bf16), stats_out.
int vt_gemm_ln_dispatch(const void* A, long lda, const void* W, long ldw, const float* bias, const float* colv,
                        const float* stats_in, int np, long stat_rows, float eps, int ln_mode, const void* Rs, long ldrs,
                        void* C, long ldc, void* Cs, long ldcs, float* stats_out, int M, int N, int K, int act,
                        hipStream_t stream) {
  if (!A || !W || !C || !bias || !colv || !stats_in) return VT_ERR_NULL;
  if (ln_mode != 1 && ln_mode != 2) return VT_ERR_UNSUPPORTED;
  if (ln_mode == 2 && (!Rs || !Cs || !stats_out)) return VT_ERR_NULL;
  if (M <= 0 || N <= 0 || K < 128 || (K % GEMM_BK) != 0 || (N & 127)) return VT_ERR_BAD_SHAPE;
  const int row_len = ln_mode == 1 ? K : N;   // the LayerNorm runs over the rows of A (mode 1) / of the stream (mode 2)
  if ((row_len & 127) || np != row_len / 128 || np > 8 || stat_rows < M || (stat_rows & 1)) return VT_ERR_BAD_SHAPE;
  if ((lda % 8) || (ldw % 8) || (ldc % 8) || (ln_mode == 2 && ((ldrs % 8) || (ldcs % 8)))) return VT_ERR_BAD_ALIGN;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)colv | (uintptr_t)stats_in | (uintptr_t)Rs |
       (uintptr_t)Cs | (uintptr_t)stats_out) & 15)
    return VT_ERR_BAD_ALIGN;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.R = nullptr; g.C = C; g.C2 = nullptr;
  g.lda = lda; g.ldw = ldw; g.ldr = 0; g.ldc = ldc; g.ldc2 = 0;
  g.M = M; g.N = N; g.K = K; g.grp_rows = 0; g.grp_stride = 0; g.tiles_m = 0; g.tiles_n = 0;
  g.trace = nullptr;
  g.drop.thresh = 0; g.drop.seed = 0; g.drop.scale = 1.0f;
  g.ln_mode = ln_mode; g.ln_np = np; g.ln_rows = (int)stat_rows; g.ln_inv_n = 1.0f / (float)row_len; g.ln_eps = eps;
  g.ln_stats = stats_in; g.colv = colv; g.Rs = (const uint16_t*)Rs; g.Cs = (uint16_t*)Cs; g.stats_out = stats_out; g.ldrs = ldrs; g.ldcs = ldcs;
  g.ksplit = 0; g.reverse = 0; g.c_plane = 0; g.r_f16 = 0; g.c_f16 = 0;
  int variant = g_gemm_variant >= 0 ? g_gemm_variant : vt_gemm_pick_variant(M, N, K, VT_TUNE_KIND(act, 0, 0, 0, ln_mode));
  if (variant != 15 && variant != 16 && (variant < 18 || variant > 23) && (variant < 28 || variant > 32)) variant = 16;   // only the 256x256-tile kernels
  return vt_gemm_ln_launch(g, act, variant, stream);
}

// ---- split-K: C[M,N] (bf16) = A[M,K] W[N,K]^T for a long K and few output tiles --------------------------------------
// ksplit copies of the 256x256 tile list of the one-tile-per-workgroup kernel, each over its share of the K-steps, write fp32
// planes into `ws` (ksplit * M * N floats); splitk_reduce_bf16 sums them.  No bias / activation / residual.
int vt_gemm_v7_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn);

__global__ __launch_bounds__(256) void splitk_reduce_bf16(const float* __restrict__ ws, long plane, int ksplit, bf16_t* __restrict__ C,
                                                          long ldc, long M, int N) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;   // element index in [M, N] (N % 4 == 0)
  if (i >= M * N) return;
  f32x4 acc = *(const f32x4*)(ws + i);
  for (int s = 1; s < ksplit; ++s) {
    const f32x4 v = *(const f32x4*)(ws + (long)s * plane + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += v[e];
  }
  const long row = i / N;
  const int col = (int)(i - row * N);
  u32x2 o;
  o[0] = pack_bf16x2(acc[0], acc[1]);
  o[1] = pack_bf16x2(acc[2], acc[3]);
  *(u32x2*)(C + row * ldc + col) = o;
}

int vt_gemm_splitk_dispatch(const void* A, long lda, const void* W, long ldw, void* C, long ldc, float* ws, int M, int N, int K,
                            int ksplit, hipStream_t stream) {
  if (!A || !W || !C || !ws) return VT_ERR_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (N % 4) || ksplit < 2 || ksplit > 64 || ksplit > K / 64) return VT_ERR_BAD_SHAPE;
  if ((lda % 8) || (ldw % 8) || (ldc % 4) || (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)ws) & 15)) return VT_ERR_BAD_ALIGN;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = nullptr; g.R = nullptr; g.C = ws; g.C2 = nullptr;
  g.lda = lda; g.ldw = ldw; g.ldr = 0; g.ldc = N; g.ldc2 = 0;
  g.M = M; g.N = N; g.K = K; g.grp_rows = 0; g.grp_stride = 0; g.tiles_m = 0; g.tiles_n = 0;
  g.trace = nullptr;
  g.drop.thresh = 0; g.drop.seed = 0; g.drop.scale = 1.0f;
  g.ln_mode = 0; g.ln_np = 0; g.ln_rows = 0; g.ln_inv_n = 0.f; g.ln_eps = 0.f; g.ln_stats = nullptr; g.colv = nullptr;
  g.Rs = nullptr; g.Cs = nullptr; g.stats_out = nullptr; g.ldrs = 0; g.ldcs = 0;
  g.ksplit = ksplit; g.reverse = 0; g.c_plane = (long)M * N; g.r_f16 = 0; g.c_f16 = 0;
  const int rc = vt_gemm_v7_launch(g, ACT_NONE, 1, stream, 8);
  if (rc) return rc;
  const long n4 = ((long)M * N + 3) / 4;
  hipLaunchKernelGGL(splitk_reduce_bf16, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, ws, (long)M * N, ksplit,
                     (bf16_t*)C, ldc, (long)M, N);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
