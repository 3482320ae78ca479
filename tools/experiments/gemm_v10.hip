// NT GEMM, variant 24: TWO co-resident workgroups per CU, each a persistent 4-wave workgroup on 256 x 128 output tiles
// (wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 tiles, 128 accumulator registers, <= 256 registers per wave).
//
// Same contraction and epilogues as gemm_bf16.hip (every nn.Linear of oscar/modeling_bert.py:43-45,94,119-120 and their
// dgrads).  Why a second design beside the one-wave-per-SIMD 256 x 256 kernel (gemm_v7_kernels.hpp): with ONE wave per SIMD
// nothing overlaps a tile's epilogue -- at K = 768 the K loop is 16.4 us and the epilogue 4.3 (plain) .. 15 us (GELU + saved
// GELU') during which the matrix pipes idle -- and the wave's own LDS-DMA issue costs a quarter of its K-step.  Here every
// SIMD hosts one wave of each of two independent workgroups: while one is in its epilogue (vector ALU, stores) or issuing
// its operand DMA, the other's MFMAs keep the pipe busy; the hardware arbitrates, no protocol between the two.  The price
// is 1.5 x the operand bytes per FLOP into LDS (48 KiB per 256 x 128 x 64 against 64 KiB per 256 x 256 x 64):
// tools/dma_rate.hip measures 120-130 GB/s per CU for L2-resident operands (57-67 with half of them from beyond L2), the
// K loop at full MFMA rate needs ~90.
//
// K pipeline (per workgroup): BK = 32, a ring of THREE 24 KiB stages (X image 256 rows x 64 B, W image 128 rows x 64 B;
// 16-B chunks XOR-swizzled by (row >> 2) & 3 on the DMA source address) = 72 KiB of LDS, so two workgroups fit a CU.  Per
// K-step and wave: 6 LDS-DMA pieces, 12 fragment reads, 32 MFMAs, ONE barrier:
//   phase 0: read the X fragments of rows 64..127 (step k) | lgkmcnt(4) | 16 MFMAs (rows 0..63)
//   phase 1: vmcnt(6): step k+1 has landed (mine) | lgkmcnt(0) | barrier: step k+1 landed for every wave AND every wave
//            has retired its reads of step k's stage | read X rows 0..63 + W fragments of step k+1 | issue the 6 pieces of
//            step k+3 into the stage step k just left | 16 MFMAs (rows 64..127)
// The ring runs straight across tile boundaries (the DMA cursor is three K-steps ahead, into the next tile); only the
// fragment prefetch is held back over the epilogue (it would pin 64 more registers under it).
#include "gemm_common.hpp"

#define V10_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define V10_XT 16384
#define V10_STAGE 24576
#define V10_LDS_BYTES (3 * V10_STAGE)

template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_v10(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- this workgroup's tiles: each XCD owns a contiguous chunk of the grouped tile order (bands of 8 row tiles,
  // column-tile major inside), its workgroups take the chunk's tiles round-robin
  const int T = g.tiles_m * g.tiles_n;
  const int nwg = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7;
  const int nx = (nwg - xcd + 7) >> 3;
  const int ng = nwg < 8 ? nwg : 8;
  const int c0 = (int)((long)T * xcd / ng), c1 = (int)((long)T * (xcd + 1) / ng);
  const int first = c0 + (b >> 3);
  if (first >= c1) return;   // uniform
  const int band_tiles = 8 * g.tiles_n;
  const int nk = g.K >> 5;
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int band = t / band_tiles;
    const int within = t - band * band_tiles;
    const int rows_left = g.tiles_m - band * 8;
    const int band_h = rows_left < 8 ? rows_left : 8;
    const int bn = within / band_h;
    m0 = (band * 8 + (within - bn * band_h)) * 256;
    n0 = bn * 128;
  };

  // ---- DMA addressing: a piece = 16 image rows x 64 B; lane -> row lane >> 2, 16-B chunk (lane & 3) ^ swz(row),
  // swz(row) = (-(row >> 2)) & 3 = (-(lane >> 4)) & 3.  X pieces 4 wave .. 4 wave + 3, W pieces 2 wave, 2 wave + 1; W image
  // row r <- W row 64 (r >> 6) + 16 ((r >> 2) & 3) + 4 ((r >> 4) & 3) + (r & 3) (a lane's four N-subtiles interleave to 16
  // consecutive output columns, as in gemm_bf16.hip)
  const int rl = lane >> 2;
  const int cch = (lane & 3) ^ ((-(rl >> 2)) & 3);
  const int vx = rl * (int)g.lda * 2 + cch * 16;
  const int vw = (16 * (rl >> 2) + (rl & 3)) * (int)g.ldw * 2 + cch * 16;
  int sx[4], sw[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) sx[i] = 16 * (4 * wave + i) * (int)g.lda * 2;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = 2 * wave + j;
    sw[j] = (64 * (p >> 2) + 4 * (p & 3)) * (int)g.ldw * 2;
  }
  int cur_t = first, cur_kt = 0;
  const char* cur_x = nullptr;
  const char* cur_w = nullptr;
  unsigned cur_xb = 0, cur_wb = 0;
  auto cursor_tile = [&]() {
    if (cur_t < c1) {
      int m0, n0;
      tile_origin(cur_t, m0, n0);
      const int rows_x = g.M - m0 < 256 ? g.M - m0 : 256;
      const int rows_w = g.N - n0 < 128 ? g.N - n0 : 128;
      cur_x = (const char*)(g.A + (long)m0 * g.lda);
      cur_w = (const char*)(g.W + (long)n0 * g.ldw);
      cur_xb = (unsigned)(((long)(rows_x - 1) * g.lda + g.K) * 2);
      cur_wb = (unsigned)(((long)(rows_w - 1) * g.ldw + g.K) * 2);
    } else {
      cur_xb = 0; cur_wb = 0;   // past the last tile: null descriptors, the pieces read nothing (and still count in vmcnt)
    }
  };
  auto dma_step = [&](int slot) {
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(cur_x + cur_kt * 64), 0, cur_xb ? (int)(cur_xb - cur_kt * 64) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(cur_w + cur_kt * 64), 0, cur_wb ? (int)(cur_wb - cur_kt * 64) : 0, 0x00020000);
    char* d = smem + slot * V10_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(d + (4 * wave + i) * 1024), 16, vx, sx[i], 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(d + V10_XT + (2 * wave + j) * 1024), 16, vw, sw[j], 0, 0);
    if (++cur_kt == nk) { cur_kt = 0; cur_t += nx; cursor_tile(); }
  };

  // ---- fragment addresses (stage 0): image row 128 wm + 16 mt + (lane & 15) / 64 wn + 16 t + (lane & 15), chunk (lane >> 4) ^ swz
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int i16 = lane & 15;
  const unsigned fsw = (unsigned)((((lane >> 4) ^ ((-(i16 >> 2)) & 3)) << 4));
  const unsigned a_addr0 = lds0 + (128 * wm + i16) * 64 + fsw;             // + 1024 mt
  const unsigned w_addr0 = lds0 + V10_XT + (64 * wn + i16) * 64 + fsw;     // + 1024 t

  f32x4 acc[8][4];
  u32x4 A0[4], A1[4], Wa[4], Wb[4];

  cursor_tile();
  dma_step(0); dma_step(1); dma_step(2);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int slot = 0;   // ring slot of the current K-step

  // one K-step; Wc: this step's W fragments (with A0 already in registers), Wn: the next step's; fetch_next: read them
  auto kstep = [&](u32x4 (&Wc)[4], u32x4 (&Wn)[4], bool fetch_next) {
    const unsigned so = (unsigned)slot * V10_STAGE;
    const int nslot = slot == 2 ? 0 : slot + 1;
    // ---- phase 0 ----
    {
      const unsigned aa = a_addr0 + so;
      V10_DSR(A1[0], aa, 4096); V10_DSR(A1[1], aa, 5120); V10_DSR(A1[2], aa, 6144); V10_DSR(A1[3], aa, 7168);
    }
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
    if (fetch_next) {
      const unsigned sn = (unsigned)nslot * V10_STAGE;
      const unsigned aa = a_addr0 + sn, ww = w_addr0 + sn;
      V10_DSR(A0[0], aa, 0); V10_DSR(A0[1], aa, 1024); V10_DSR(A0[2], aa, 2048); V10_DSR(A0[3], aa, 3072);
      V10_DSR(Wn[0], ww, 0); V10_DSR(Wn[1], ww, 1024); V10_DSR(Wn[2], ww, 2048); V10_DSR(Wn[3], ww, 3072);
    }
    dma_step(slot);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[4 + mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wc[t]),
                                                                 __builtin_bit_cast(bf16x8, A1[mt]), acc[4 + mt][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  };

  for (int t = first; t < c1; t += nx) {
    int m0, n0;
    tile_origin(t, m0, n0);
    // first fragments of the tile (its step 0 has landed for every wave: prologue barrier / the previous step's barrier)
    {
      const unsigned aa = a_addr0 + (unsigned)slot * V10_STAGE, ww = w_addr0 + (unsigned)slot * V10_STAGE;
      V10_DSR(A0[0], aa, 0); V10_DSR(A0[1], aa, 1024); V10_DSR(A0[2], aa, 2048); V10_DSR(A0[3], aa, 3072);
      V10_DSR(Wa[0], ww, 0); V10_DSR(Wa[1], ww, 1024); V10_DSR(Wa[2], ww, 2048); V10_DSR(Wa[3], ww, 3072);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; kt += 2) {   // K % 64 == 0: an even number of steps
      kstep(Wa, Wb, true);
      kstep(Wb, Wa, kt + 2 < nk);
    }
    {
      f32x4 lo[4][4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) lo[mt][tt] = acc[mt][tt];
      gemm_epilogue<ACT, OUT_F32>(g, lo, lane, m0 + 128 * wm, n0 + 64 * wn);
    }
    {
      f32x4 hi[4][4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) hi[mt][tt] = acc[4 + mt][tt];
      gemm_epilogue<ACT, OUT_F32>(g, hi, lane, m0 + 128 * wm + 64, n0 + 64 * wn);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's (empty) pieces retire before the LDS goes away
}

// Two workgroups per compute unit (72 KiB of LDS, <= 256 registers each)
int vt_gemm_persistent_cus();   // gemm_v7.hip: compute units the persistent grids may use (vt_gemm_reserve_cus)

template <int ACT, bool OUT_F32>
static int launch_v10(const GemmArgs& g, hipStream_t stream) {
  GemmArgs ga = g;
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 128L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  if (ACT == ACT_MUL && !g.R) return VT_ERR_NULL;
  ga.tiles_m = (g.M + 255) / 256;
  ga.tiles_n = (g.N + 127) / 128;
  const int cus = vt_gemm_persistent_cus();
  if (cus <= 0) return VT_ERR_HIP;
  const long tiles = (long)ga.tiles_m * ga.tiles_n;
  const int grid = (int)(tiles < 2L * cus ? tiles : 2L * cus);
  auto kern = gemm_nt_bf16_v10<ACT, OUT_F32>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V10_LDS_BYTES) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), V10_LDS_BYTES, stream, ga);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_gemm_v10_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream) {
  switch (act * 2 + (out_f32 ? 1 : 0)) {
    case 0: return launch_v10<ACT_NONE, false>(g, stream);
    case 1: return launch_v10<ACT_NONE, true>(g, stream);
    case 2: return launch_v10<ACT_GELU, false>(g, stream);
    case 3: return launch_v10<ACT_GELU, true>(g, stream);
    case 4: return launch_v10<ACT_TANH, false>(g, stream);
    case 5: return launch_v10<ACT_TANH, true>(g, stream);
    case 6: return launch_v10<ACT_MUL, false>(g, stream);
    default: return VT_ERR_UNSUPPORTED;
  }
}
