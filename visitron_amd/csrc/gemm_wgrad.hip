// Weight gradients of the nn.Linear layers on the path:  dW[N,K] (+)= dY[M,N]^T . X[M,K]  (fp32 out),
// optionally db[N] (+)= column sums of dY.  These are the wgrad halves of `loss.backward()`
// (tasks/viewpoint_select/pretrain.py:191) for query/key/value, BertSelfOutput.dense,
// BertIntermediate.dense, BertOutput.dense (oscar/modeling_bert.py:43-45,94,119,120), the region
// projection, the MLM decoder / transform and the small heads (tasks/viewpoint_select/encoder.py).
//
// gfx950 design.  Both operands are row-major with the REDUCTION index (token m) as the row, so
// neither can feed an MFMA port with a contiguous read; the 64-row x 128-column tiles are staged
// as they lie in HBM (buffer_load ... lds, 16 B per lane, no VGPR staging, out-of-range rows read
// as zero through the buffer descriptor's bounds check -- that is the M tail) and the MFMA
// fragments are taken with ds_read_b64_tr_b16, the hardware transposing LDS read.  LDS rows are
// 256 B = one full bank window, so the 32-byte group index of a row is XOR-swizzled with
// (row&3)|((row>>3)&1)<<2 (applied on the DMA source address) to make the transposed reads
// conflict-free.  One workgroup = 4 waves = a 128(n) x 128(k) tile of dW reduced over ALL of M
// (no split-K, no atomics, bitwise reproducible); the launch is GROUPED: up to 8 (dY, X, dW)
// problems -- e.g. the four weight matrices of one encoder layer, 432 tiles -- share one grid so
// the chip is filled in a single round.  MFMA operands are arranged (A = X^T, B = dY^T) so that a
// lane ends up with 4 consecutive k of one n: 16-byte fp32 stores.  Bias gradients ride along as
// an extra MFMA against an all-ones A fragment in the k-tile-0 workgroups.
#include "common.hpp"

#include "wgrad_common.hpp"

#define WG_TILE_BYTES (64 * 256)   // X tile: 64 rows x 128 columns
// TN = width of the dY tile (n-range of the output tile): 128 (4 waves) or 256 (8 waves, 85 instead of 64
// FLOP per staged byte; a layer's four matrices are then 216 tiles = one round on 256 CUs)
template <int TN> struct WgCfg {
  static constexpr int PITCH_Y = TN * 2;               // bytes per dY tile row
  static constexpr int TILE_Y = 64 * PITCH_Y;
  static constexpr int STAGE = TILE_Y + WG_TILE_BYTES; // dY tile then X tile
  static constexpr int LDS = 2 * STAGE;
  static constexpr int WAVES = TN / 32;                // (TN/64) x 2
  static constexpr int NPY = (TILE_Y / 1024) / WAVES;  // dY DMA pieces per wave per stage (4)
  static constexpr int NPX = 16 / WAVES;               // X DMA pieces per wave per stage (4 or 2)
};

__device__ __forceinline__ int wg_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ bf16x8 tr_frag(unsigned addr, int pitch) {
  // rows m..m+3 and m+4..m+7 of a 16-column block, delivered column-major: 8 bf16 along m per lane
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)addr);
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(addr + 4 * pitch));
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int TN>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a) {
  using C = WgCfg<TN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;

  // XCD-contiguous remap: blocks b and b+8 share an XCD (and its L2); give every XCD one contiguous
  // range of the tile list, whose order is problem-major then n-tile-major, so the workgroups that
  // share a dY column panel (same n-tile, every k-tile) stream it through ONE L2 instead of eight.
  // (PMC: 1.65 GB fetched per layer launch before the remap vs 0.36 GB of operands.)
  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  const int t_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  // which problem / tile (wave-uniform scan over <= 8 problems)
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAX_PROBLEMS; ++i)
    if (i < a.nprob && t_id >= a.p[i].tile_begin) pi = i;
  const WgradProblem& P = a.p[pi];
  const int lt = t_id - P.tile_begin;
  const int bn = lt / P.tiles_k, bk = lt - bn * P.tiles_k;
  const int n0 = bn * TN, k0 = bk * 128;
  const int M = a.M;

  // buffer descriptors: base at the tile's first column; rows >= M (and columns past the row end of
  // the last row) fall outside num_records and read as zero
  const int ncols_y = (P.N - n0) < TN ? (P.N - n0) : TN;
  const int ncols_x = (P.K - k0) < 128 ? (P.K - k0) : 128;
  // whole 16-B chunks must be in range (a partially out-of-range dwordx4 reads as zero): round the
  // column count up to 8; ld % 8 == 0 keeps that inside the last row's storage
  const long bytes_y = ((long)(M - 1) * P.ldy + ((ncols_y + 7) & ~7)) * 2;
  const long bytes_x = ((long)(M - 1) * P.ldx + ((ncols_x + 7) & ~7)) * 2;
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(P.dY + n0), 0, (int)bytes_y, 0x00020000);
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(P.X + k0), 0, (int)bytes_x, 0x00020000);

  // DMA pieces of 1 KiB: the X tile (256-B rows) in 16 pieces of 4 rows; the dY tile in pieces of
  // 1024 / PITCH_Y rows.  LDS slot (row, cs) holds logical 16-B chunk c = cs ^ (swz(row) << 1) (the XOR
  // stays inside a 256-B window).  Columns past the matrix edge are redirected out of range (zero fill).
  int voff_y[C::NPY], voff_x[C::NPX];
#pragma unroll
  for (int i = 0; i < C::NPY; ++i) {
    const int pce = wave * C::NPY + i;
    constexpr int RPP = 1024 / C::PITCH_Y;        // rows per piece (4 or 2)
    constexpr int CPR = C::PITCH_Y / 16;          // chunks per row (16 or 32)
    const int row = pce * RPP + lane / CPR;
    const int c = (lane % CPR) ^ (wg_swz(row) << 1);
    voff_y[i] = (c * 8 < ncols_y) ? (int)(row * P.ldy * 2 + c * 16) : 0x7fffffff;
  }
#pragma unroll
  for (int i = 0; i < C::NPX; ++i) {
    const int pce = wave * C::NPX + i;
    const int row = pce * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (wg_swz(row) << 1);
    voff_x[i] = (c * 8 < ncols_x) ? (int)(row * P.ldx * 2 + c * 16) : 0x7fffffff;
  }
  const int step_y = (int)(64 * P.ldy * 2), step_x = (int)(64 * P.ldx * 2);

  // fragment read addresses (k-substep 0, first 4-row block): lane (i16 = lane&15 -> q = i16>>2 row in
  // block, p = i16&3 4-column piece; g = lane>>4 -> rows 8g..8g+7 of the 32-row substep)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, g = lane >> 4;
  const int frow = 8 * g + q4;  // + 32*ks (+4 for the second block: same swizzle)
  const int fsw = wg_swz(frow) << 5;
  unsigned y_addr[4], x_addr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    y_addr[t] = lds0 + frow * C::PITCH_Y + ((((64 * wn + 16 * t) * 2) + 8 * p4) ^ fsw);
    x_addr[t] = lds0 + C::TILE_Y + frow * 256 + ((((64 * wk + 16 * t) * 2) + 8 * p4) ^ fsw);
  }

  f32x4 acc[4][4];
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = (P.db != nullptr) && (bk == 0) && (wk == 0);  // wave-uniform
  typedef __attribute__((ext_vector_type(8))) short short8v;
  const short one = (short)0x3F80;  // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, (short8v){one, one, one, one, one, one, one, one});

  const int nk = (M + 63) >> 6;
#pragma unroll
  for (int i = 0; i < C::NPY; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, LDS_PTR(smem + (wave * C::NPY + i) * 1024), 16, voff_y[i], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < C::NPX; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + C::TILE_Y + (wave * C::NPX + i) * 1024), 16, voff_x[i], 0, 0, 0);

  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned so = (kt & 1) * C::STAGE;

    bf16x8 yf[2][4], xf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        yf[ks][t] = tr_frag(y_addr[t] + so + ks * (32 * C::PITCH_Y), C::PITCH_Y);
        xf[ks][t] = tr_frag(x_addr[t] + so + ks * (32 * 256), 256);
      }

    if (kt + 1 < nk) {
      char* nb = smem + ((kt + 1) & 1) * C::STAGE;
      const int sy = (kt + 1) * step_y, sx = (kt + 1) * step_x;
#pragma unroll
      for (int i = 0; i < C::NPY; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, LDS_PTR(nb + (wave * C::NPY + i) * 1024), 16, voff_y[i], sy, 0, 0);
#pragma unroll
      for (int i = 0; i < C::NPX; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(nb + C::TILE_Y + (wave * C::NPX + i) * 1024), 16, voff_x[i], sx, 0, 0);
    }

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks][t], yf[ks][nt], acc[nt][t], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[ks][nt], accb[nt], 0, 0, 0);
      }
    }
  }

  // epilogue: lane (j = lane&15, g) holds dW[n0 + 64wn + 16nt + j][k0 + 64wk + 16t + 4g .. +3]
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = n0 + 64 * wn + 16 * nt + (lane & 15);
    if (n >= P.N) continue;
    float* orow = P.dW + (long)n * P.ldw;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = k0 + 64 * wk + 16 * t + 4 * g;
      if (k + 4 <= P.K) {
        f32x4 v = acc[nt][t];
        if (P.accumulate) {
          const f32x4 o = *(const f32x4*)(orow + k);
          v = (f32x4){v[0] + o[0], v[1] + o[1], v[2] + o[2], v[3] + o[3]};
        }
        *(f32x4*)(orow + k) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < P.K) orow[k + e] = acc[nt][t][e] + (P.accumulate ? orow[k + e] : 0.f);
      }
    }
    if (do_bias && g == 0) {
      const float s = accb[nt][0];  // every row of the ones-product is the column sum; take row 0
      P.db[n] = P.accumulate ? P.db[n] + s : s;
    }
  }
}

__global__ __launch_bounds__(256, 2) void gemm_wgrad_tn_bf16(WgradArgs a) { wgrad_body<128>(a); }
__global__ __launch_bounds__(512, 2) void gemm_wgrad_tn_bf16_w256(WgradArgs a) { wgrad_body<256>(a); }

// Host entry.  problems: array of `nprob` WgradProblem-like records filled by capi.hip.
static int g_wgrad_tn = 0;  // 0: choose; 128 / 256: forced tile width, -8 never / 8 always the persistent kernel (tuning hook)
void vt_wgrad_set_tile(int tn) { g_wgrad_tn = tn; }

template <int TN>
static int wgrad_launch(WgradArgs& a, int total, hipStream_t stream) {
  if (TN == 256) {
    if (hipFuncSetAttribute((const void*)gemm_wgrad_tn_bf16_w256, hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<256>::LDS) != hipSuccess) return VT_ERR_HIP;
    hipLaunchKernelGGL(gemm_wgrad_tn_bf16_w256, dim3(total), dim3(512), WgCfg<256>::LDS, stream, a);
  } else {
    if (hipFuncSetAttribute((const void*)gemm_wgrad_tn_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<128>::LDS) != hipSuccess) return VT_ERR_HIP;
    hipLaunchKernelGGL(gemm_wgrad_tn_bf16, dim3(total), dim3(256), WgCfg<128>::LDS, stream, a);
  }
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_wgrad_v8_dispatch(WgradArgs& a, hipStream_t stream, bool force);   // gemm_wgrad_v8.hip

int vt_wgrad_dispatch(WgradArgs& a, hipStream_t stream) {
  if (a.nprob <= 0 || a.nprob > WG_MAX_PROBLEMS || a.M <= 0) return VT_ERR_BAD_SHAPE;
  // 128-wide n-tiles by default: the 256-wide form (8 waves, one workgroup per CU) measured 24 % SLOWER on
  // the encoder layer group (15.2 vs 12.2 ms per step at B=256) and is kept only behind the tuning hook
  int tn = 128;
  if (g_wgrad_tn == 128 || g_wgrad_tn == 256) tn = g_wgrad_tn;
  int total = 0;
  for (int i = 0; i < a.nprob; ++i) {
    WgradProblem& P = a.p[i];
    if (!P.dY || !P.X || !P.dW) return VT_ERR_NULL;
    if (P.N <= 0 || P.K <= 0) return VT_ERR_BAD_SHAPE;
    if ((P.ldy % 8) || (P.ldx % 8) || (P.ldw % 4) || (P.K % 4)) return VT_ERR_BAD_ALIGN;
    if (((uintptr_t)P.dY | (uintptr_t)P.X | (uintptr_t)P.dW) & 15) return VT_ERR_BAD_ALIGN;
    if ((long)a.M * P.ldy * 2 >= (1L << 31) || (long)a.M * P.ldx * 2 >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  }
  // groups whose matrices are whole 256x256 tiles: the persistent stream-K kernel (long reductions, every CU busy)
  if (g_wgrad_tn == 0 || g_wgrad_tn == 8) {   // (8: the persistent kernel wherever it is eligible, whatever the row count)
    const int rc8 = vt_wgrad_v8_dispatch(a, stream, g_wgrad_tn == 8);
    if (rc8 != VT_ERR_UNSUPPORTED) return rc8;
  }
  for (int i = 0; i < a.nprob; ++i) {
    WgradProblem& P = a.p[i];
    P.tiles_k = (P.K + 127) / 128;
    P.tile_begin = total;
    total += ((P.N + tn - 1) / tn) * P.tiles_k;
  }
  return tn == 256 ? wgrad_launch<256>(a, total, stream) : wgrad_launch<128>(a, total, stream);
}
