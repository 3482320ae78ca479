// NT GEMM, variants 26 / 27: the persistent 256-column-tile kernel for SHORT tiles (160 / 128 rows, MTN = 5 / 4 row blocks
// per wave) on a ring of THREE operand stages.  A 128-row tile's K-step holds 0.5 us of MFMAs; with two 64 KiB stages the
// step cannot be shorter than the time an operand piece takes to land divided by its 1.3-step head start, and it pays three
// barriers (X image free, W image free, next stage landed) -- 1.6 us per K-step at the M = 8 208 shapes (B = 36, the
// per-GPU share of BASELINE configs[3]).  A stage of a short tile is 48 / 52 KiB, so three fit into the 160 KiB of LDS:
//   * the stage a K-step has just finished reading is refilled with the K-tile THREE ahead, so every piece has two full
//     steps to land;
//   * one barrier per step: "the next stage has landed for every wave" is also "every wave is done with the stage two
//     back" (its last fragments were read a phase earlier), so the image-free barriers are gone.
// Same images, swizzle, fragment layout, MFMA order (i = 8 nt + mt) and straight-line epilogue as gemm_v7_kernels.hpp, hence
// bitwise the same results as variants 20 / 21.  bf16 / fp16 output in whole 64-column slabs only; other calls go to
// variant 20 / 21.
#include "gemm_v7_kernels.hpp"

#define V12_STAGE(MTN) (32768 + 4096 * (MTN))
#define V12_LDS_BYTES(MTN) (3 * V12_STAGE(MTN) + 4096)   // three stages + 1 KiB per wave: the tile's bias values

// X pieces: piece p = wave * MTN + i of the X image (rows 8p .. 8p+7), parity p & 1 selects the swizzle -- with MTN odd the
// parity of a wave's i-th piece depends on the wave, so the per-lane offset is vx0 ^ (parity << 6) (the two swizzles differ
// in bit 2 of the 16-byte chunk index; row pitch K * 2 bytes is a multiple of 128).
#define V12_DMA_X(rs, so_, i)                                                                              \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (so_) + xdst + (i) * 1024), 16, vx[0] ^ (((wpar + (i)) & 1) << 6), \
                                           sxb + (i) * pstep, 0, 0)
#define V12_DMA_W(rs, so_, i)                                                                              \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (so_) + 4096 * MTN + wave * 8192 + (i) * 1024), 16, vw[(i) & 1], sw[i], 0, 0)

// One K-step.  `so`: byte offset of the stage that holds the current K-tile (its k-substep-0 fragments are in set 0), `sn`:
// the next K-tile's stage; (rx, rw) describe the K-tile three ahead, which goes into `so` behind the barrier.
#define V12_STEP(MFMA_A)                                                                                   \
  {                                                                                                        \
    const unsigned xa1 = xb1 + so, wa1 = wb1 + so, xn0 = xb0 + sn, wn0 = wb0 + sn;                         \
    asm volatile("" : "+v"(vx[0]), "+v"(vx[1]), "+v"(vw[0]), "+v"(vw[1]));                                 \
    _Pragma("unroll") for (int i = 0; i < 64; ++i) {                                                       \
      MFMA_A(0, i);                                                                                        \
      if (i < 2 * MTN && (i & 1)) V7_LDSR(xf[1][i >> 1], xa1, (i >> 1) * 2048);                            \
      if (i >= 16 && i < 32 && (i & 1)) V7_LDSR(wf[1][(i - 16) >> 1], wa1, ((i - 16) >> 1) * 2048);        \
    }                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    _Pragma("unroll") for (int i = 0; i < 64; ++i) {                                                       \
      V7_MFMA(1, i);                                                                                       \
      if (i == 24) {   /* the next K-tile has landed (the MTN + 8 pieces of the one after it may be in flight) */ \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MTN + 8) : "memory");                                     \
        __builtin_amdgcn_s_barrier();                                                                      \
      }                                                                                                    \
      if (i >= 26 && i < 26 + 2 * (MTN + 8) && !(i & 1)) {                                                 \
        const int j = (i - 26) >> 1;                                                                       \
        if (j < MTN) V7_LDSR(xf[0][j], xn0, j * 2048);                                                     \
        else V7_LDSR(wf[0][j - MTN], wn0, (j - MTN) * 2048);                                               \
      }                                                                                                    \
      if (i >= 27 && i < 27 + 2 * (MTN + 8) && (i & 1)) {                                                  \
        const int j = (i - 27) >> 1;                                                                       \
        if (j < MTN) V12_DMA_X(rx, so, j);                                                                 \
        else V12_DMA_W(rw, so, j - MTN);                                                                   \
      }                                                                                                    \
    }                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    so = sn;                                                                                               \
    sn = sn + S3 == 3 * S3 ? 0u : sn + S3;                                                                 \
  }

template <int ACT, bool HAS_R, int MTN>
__global__ __launch_bounds__(256, 1) void gemm_nt_bf16_v12(GemmArgs g) {
  constexpr int TH = 32 * MTN;
  constexpr unsigned S3 = V12_STAGE(MTN);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int T = g.tiles_m * g.tiles_n;
  const int nwg = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7;
  const int nx = (nwg - xcd + 7) >> 3;
  const int ng = nwg < 8 ? nwg : 8;
  const int c0 = (int)((long)T * xcd / ng), c1 = (int)((long)T * (xcd + 1) / ng);
  const int first = c0 + (b >> 3);
  const int band_tiles = 8 * g.tiles_n;
  const int nk = g.K >> 6;

  auto tile_origin = [&](int t, int& m0, int& n0) {   // grouped order (see gemm_nt_bf16_v8)
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
  const int pstep = 8 * (int)g.lda * 2;       // one piece = 8 rows of the operand
  const int sxb = wave * MTN * pstep;         // this wave's first X piece
  const int wpar = (wave * MTN) & 1;          // its parity
  int sw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sw[i] = (64 * wave + 32 * (i & 1) + 4 * (i >> 1)) * (int)g.ldw * 2;
  const unsigned xdst = wave * MTN * 1024;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned fr = (lane & 15) * 128 + ((((unsigned)lane >> 4) ^ (((unsigned)lane & 15) >> 1)) << 4);
  // fragment bases inside a stage (k-substep 0; substep 1 = the same ^ 64: stage offsets are multiples of 128)
  unsigned xb0 = lds0 + wm * (MTN * 2048) + fr, xb1 = xb0 ^ 64;
  unsigned wb0 = lds0 + 4096 * MTN + wn * 16384 + fr, wb1 = wb0 ^ 64;
  unsigned so = 0, sn = S3;

  int cur_t = first, cur_kt = 0;
  const char* cur_x = nullptr;
  const char* cur_w = nullptr;
  unsigned cur_xb = 0, cur_wb = 0;
  auto cursor_tile = [&]() {
    if (cur_t < c1) {
      int m0, n0;
      tile_origin(cur_t, m0, n0);
      const int rows_x = g.M - m0 < TH ? g.M - m0 : TH;
      const int rows_w = g.N - n0 < 256 ? g.N - n0 : 256;
      cur_x = (const char*)(g.A + (long)m0 * g.lda);
      cur_w = (const char*)(g.W + (long)n0 * g.ldw);
      cur_xb = (unsigned)(((long)(rows_x - 1) * g.lda + g.K) * 2);
      cur_wb = (unsigned)(((long)(rows_w - 1) * g.ldw + g.K) * 2);
    } else {
      cur_xb = 0; cur_wb = 0;   // past the last tile: null descriptors
    }
  };
  auto cursor_next = [&]() {
    if (++cur_kt == nk) { cur_kt = 0; cur_t += nx; cursor_tile(); }
  };

  f32x4 acc[8][8];   // rows [0, MTN) are used
  u32x4 xf[2][8], wf[2][8];

  if (first >= c1) return;
  unsigned long long* tr = g.trace ? g.trace + (long)b * 64 : nullptr;
  int tri = 2;
#define V12_TRACE_RT() { if (tr && tid == 0 && tri < 40) tr[tri++] = __builtin_amdgcn_s_memrealtime(); }
  if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_memrealtime(); tr[1] = __builtin_amdgcn_s_memtime(); }
  cursor_tile();
  {   // prologue: K-tiles 0, 1, 2 in flight; substep-0 fragments of K-tile 0 in set 0
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
      cursor_next();
#pragma unroll
      for (int i = 0; i < MTN; ++i) V12_DMA_X(rx, s * S3, i);
#pragma unroll
      for (int i = 0; i < 8; ++i) V12_DMA_W(rw, s * S3, i);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (MTN + 8)) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < MTN) V7_LDSR(xf[0][i], xb0, i * 2048);
      V7_LDSR(wf[0][i], wb0, i * 2048);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  for (int t = first; t < c1; t += nx) {
    int m0, n0;
    tile_origin(t, m0, n0);
    V12_TRACE_RT();
    {  // first K-step of the tile; the bias piece goes first (older than every piece a later wait counts)
      __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
      cursor_next();
      {
        const int bn_ = g.N - n0 < 256 ? g.N - n0 : 256;
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bias ? g.bias + n0 : nullptr), 0, g.bias ? bn_ * 4 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(smem + 3 * S3 + wave * 1024), 16, lane * 16, 0, 0, 0);
      }
      V12_STEP(V7_MFMA0)
    }
    for (int kt = 1; kt < nk; ++kt) {
      __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
      cursor_next();
      V12_STEP(V7_MFMA)
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    V12_TRACE_RT();
    // (the epilogue finds its bias slot at lds0 + 2 * V7_STAGE + wave KiB: shift the base so that this is 3 * S3)
    v7_epilogue_fast<ACT, HAS_R, MTN>(g, acc, lane, wave, m0, n0, lds0 + 3 * S3 - 2 * V7_STAGE);
    V12_TRACE_RT();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tr && tid == 0) { tr[62] = __builtin_amdgcn_s_memrealtime(); tr[63] = __builtin_amdgcn_s_memtime(); }
}

int vt_gemm_persistent_cus();   // gemm_v7.hip
int vt_gemm_v8_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn);

template <int ACT, int MTN>
static int launch_v12(const GemmArgs& g, hipStream_t stream) {
  GemmArgs ga = g;
  ga.tiles_n = (g.N + 255) / 256;
  ga.tiles_m = (g.M + 32 * MTN - 1) / (32 * MTN);
  if (ACT == ACT_MUL && !g.R) return VT_ERR_NULL;
  const int cus = vt_gemm_persistent_cus();
  if (cus <= 0) return VT_ERR_HIP;
  const long tiles = (long)ga.tiles_m * ga.tiles_n;
  const int grid = (int)(tiles < cus ? tiles : cus);
  const bool has_r = g.R || ACT == ACT_MUL;
  constexpr bool NO_R = ACT == ACT_MUL;   // (ACT_MUL always reads R)
  void (*kern)(GemmArgs) = &gemm_nt_bf16_v12<ACT, NO_R, MTN>;
  if (has_r) kern = &gemm_nt_bf16_v12<ACT, true, MTN>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V12_LDS_BYTES(MTN)) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), V12_LDS_BYTES(MTN), stream, ga);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_gemm_v12_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn) {
  const bool fast = !out_f32 && (g.N & 63) == 0 && g.grp_rows == 0 && act != ACT_TANH;
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  // (odd MTN: the X pieces' swizzle is taken from one per-lane offset by an xor, which needs the row pitch to be a multiple of 128 B)
  if (!fast || (mtn != 4 && (g.lda & 63))) return vt_gemm_v8_launch(g, act, out_f32, stream, mtn);
  switch (act * 2 + (mtn == 4 ? 1 : 0)) {
    case ACT_NONE * 2 + 0: return launch_v12<ACT_NONE, 5>(g, stream);
    case ACT_NONE * 2 + 1: return launch_v12<ACT_NONE, 4>(g, stream);
    case ACT_GELU * 2 + 0: return launch_v12<ACT_GELU, 5>(g, stream);
    case ACT_GELU * 2 + 1: return launch_v12<ACT_GELU, 4>(g, stream);
    case ACT_MUL * 2 + 0: return launch_v12<ACT_MUL, 5>(g, stream);
    case ACT_MUL * 2 + 1: return launch_v12<ACT_MUL, 4>(g, stream);
    default: return VT_ERR_UNSUPPORTED;
  }
}
