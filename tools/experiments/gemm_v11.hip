// NT GEMM, variant 25: the persistent 256x256-tile kernel of gemm_v7_kernels.hpp re-cut for EIGHT waves -- two groups of
// four, wave tile 64 x 128 (the MTN = 4 instantiation of the hand-ordered K-step stream), 128 accumulators in AGPRs and at
// most 128 VGPRs per wave, two waves per SIMD -- with the operand stages SHARED by the groups (the 64 KiB stage is the
// 256-row X image + the 256-row W image, as before, so bytes per FLOP and bytes in flight are those of the one-wave-per-SIMD
// kernel).  It is the first half of the "second wave group in the opposite phase" design: here the groups
// still run in phase (upper / lower half of the same tile); what it measures is the K-step rate with two waves per SIMD --
// 24 fragment reads per wave and step instead of 32 (192 KiB of LDS reads per step and CU against 128), eight DMA pieces
// per wave instead of sixteen.
#include "gemm_v7_kernels.hpp"

#define V11_LDS_BYTES (2 * V7_STAGE + 8192)   // two operand stages + 1 KiB per wave: the tile's bias values
#define V11_DMA_X(rs, d, i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (d) + (i) * 1024), 16, vx[(i) & 1], sx[i], 0, 0)
#define V11_DMA_W(rs, d, i) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (d) + V7_WOFF + (i) * 1024), 16, vw[(i) & 1], sw[i], 0, 0)
// One K-step (see V7_STEP_): the same three barriers; per wave 4 X + 8 W fragment reads per k-substep and 4 + 4 DMA pieces.
// VMW = 8: the eight pieces issued so far in this step may stay in flight when the next K-tile's are waited for.
#define V11_STEP_(VMW, MFMA_A)                                                                            \
  {                                                                                                       \
    const unsigned xn0 = xa0 ^ V7_STAGE, wn0 = wa0 ^ V7_STAGE;                                            \
    asm volatile("" : "+v"(vx[0]), "+v"(vx[1]), "+v"(vw[0]), "+v"(vw[1]));                                \
    _Pragma("unroll") for (int i = 0; i < 64; ++i) {                                                      \
      MFMA_A(0, i);                                                                                       \
      if (i < 8 && (i & 1)) V7_LDSR(xf[1][i >> 1], xa1, (i >> 1) * 2048);                                 \
      if (i == 20) {                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                     \
      }                                                                                                   \
      if (i >= 22 && i < 30 && !(i & 1)) V11_DMA_X(rx, dst, (i - 22) >> 1);                               \
      if (i >= 22 && i < 38 && (i & 1)) V7_LDSR(wf[1][(i - 22) >> 1], wa1, ((i - 22) >> 1) * 2048);       \
      if (i == 50) {                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                     \
      }                                                                                                   \
      if (i == 52) V11_DMA_W(rw, dst, 0);                                                                 \
      if (i == 58) V11_DMA_W(rw, dst, 1);                                                                 \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 64; ++i) {                                                      \
      V7_MFMA(1, i);                                                                                      \
      if (i == 4) V11_DMA_W(rw, dst, 2);                                                                  \
      if (i == 10) V11_DMA_W(rw, dst, 3);                                                                 \
      if (i == 24) {                                                                                      \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");                                        \
        __builtin_amdgcn_s_barrier();                                                                     \
      }                                                                                                   \
      if (i >= 26 && i < 58 && !(i & 1)) {                                                                \
        const int j = (i - 26) >> 1;                                                                      \
        if (j < 8) { if (j < MTN) V7_LDSR(xf[0][j], xn0, j * 2048); }                                     \
        else V7_LDSR(wf[0][j - 8], wn0, (j - 8) * 2048);                                                  \
      }                                                                                                   \
    }                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    xa0 ^= V7_STAGE; xa1 ^= V7_STAGE; wa0 ^= V7_STAGE; wa1 ^= V7_STAGE; dst ^= V7_STAGE;                  \
  }

// Timing experiment (GemmArgs::trace slot 256*64+2 != 0; results are NOT valid then): group B "paused" -- its waves keep
// the step's three barriers and their eight DMA pieces but issue no MFMAs and read no fragments, optionally running
// `fill` x 8 independent v_fma_f32 in each of the three gaps (the vector work of an epilogue cut into pieces).  It measures
// what the opposite-phase design rests on: the K-step of one group computing alone, and what a partner doing vector work at
// the same barrier cadence costs it.
#define V11_FILL(n)                                                                                       \
  for (int r_ = 0; r_ < (n); ++r_)                                                                        \
    asm volatile("v_fma_f32 %0, %8, %9, %0\n\tv_fma_f32 %1, %8, %9, %1\n\tv_fma_f32 %2, %8, %9, %2\n\tv_fma_f32 %3, %8, %9, %3\n\t" \
                 "v_fma_f32 %4, %8, %9, %4\n\tv_fma_f32 %5, %8, %9, %5\n\tv_fma_f32 %6, %8, %9, %6\n\tv_fma_f32 %7, %8, %9, %7"     \
                 : "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]), "+v"(fl[3]), "+v"(fl[4]), "+v"(fl[5]), "+v"(fl[6]), "+v"(fl[7])     \
                 : "v"(fc1), "v"(fc2))
#define V11_PSTEP_(VMW)                                                                                   \
  {                                                                                                       \
    asm volatile("" : "+v"(vx[0]), "+v"(vx[1]), "+v"(vw[0]), "+v"(vw[1]));                                \
    V11_FILL(fill);                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) V11_DMA_X(rx, dst, i);                                  \
    V11_FILL(fill);                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) V11_DMA_W(rw, dst, i);                                  \
    V11_FILL(fill);                                                                                       \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");                                            \
    __builtin_amdgcn_s_barrier();                                                                         \
    xa0 ^= V7_STAGE; xa1 ^= V7_STAGE; wa0 ^= V7_STAGE; wa1 ^= V7_STAGE; dst ^= V7_STAGE;                  \
  }

template <int ACT, bool HAS_R, bool DBG = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_v11(GemmArgs g) {
  constexpr int MTN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3, wn = wave & 1;   // group, wave inside the group (2 (M) x 2 (N))

  const int T = g.tiles_m * g.tiles_n;
  const int nwg = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7;
  const int nx = (nwg - xcd + 7) >> 3;
  const int ng = nwg < 8 ? nwg : 8;
  const int c0 = (int)((long)T * xcd / ng), c1 = (int)((long)T * (xcd + 1) / ng);
  const int first = c0 + (b >> 3);
  const int band_tiles = 8 * g.tiles_n;
  const int nk = g.K >> 6;

  auto tile_origin = [&](int t, int& m0, int& n0) {   // grouped order of the 256 x 256 tiles (see gemm_nt_bf16_v8)
    const int band = t / band_tiles;
    const int within = t - band * band_tiles;
    const int rows_left = g.tiles_m - band * 8;
    const int band_h = rows_left < 8 ? rows_left : 8;
    const int bn = within / band_h;
    m0 = (band * 8 + (within - bn * band_h)) * 256;
    n0 = bn * 256;
  };

  // DMA: piece p = 4 * wave + i of an operand image (rows 8p .. 8p+7); swizzle by piece parity as in v7
  int vx[2], vw[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int c = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
    vx[par] = (lane >> 3) * (int)g.lda * 2 + c * 16;
    vw[par] = (16 * (lane >> 5) + ((lane >> 3) & 3)) * (int)g.ldw * 2 + c * 16;
  }
  int sx[4], sw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int iw = 4 * (wave & 1) + i;   // piece inside the 64-row block of the W image (row permutation of v7)
    sx[i] = (wave * 4 + i) * 8 * (int)g.lda * 2;
    sw[i] = (64 * (wave >> 1) + 32 * (iw & 1) + 4 * (iw >> 1)) * (int)g.ldw * 2;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned fr = (lane & 15) * 128 + ((((unsigned)lane >> 4) ^ (((unsigned)lane & 15) >> 1)) << 4);
  // X image rows 64 * (wave >> 1) .. + 63 (group g owns rows 128 g .. 128 g + 127), W image rows 128 wn .. + 127
  unsigned xa0 = lds0 + (wave >> 1) * 8192 + fr, xa1 = xa0 ^ 64;
  unsigned wa0 = lds0 + V7_WOFF + wn * 16384 + fr, wa1 = wa0 ^ 64;
  unsigned dst = wave * 4096;

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
      cur_xb = 0; cur_wb = 0;
    }
  };
  auto cursor_next = [&]() {
    if (++cur_kt == nk) { cur_kt = 0; cur_t += nx; cursor_tile(); }
  };

  f32x4 acc[8][8];   // rows [0, MTN) are used
  u32x4 xf[2][8], wf[2][8];

  if (first >= c1) return;
  const int dbg = DBG && g.trace ? (int)g.trace[256 * 64 + 2] : 0;   // timing experiment (see V11_PSTEP_): its own instantiation
  const bool paused = DBG && dbg != 0 && grp == 1;
  const int fill = dbg == 2 ? 12 : dbg == 3 ? 32 : 0;
  float fl[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  float fc1 = 0.5f + lane * 1e-3f, fc2 = 0.25f;   // (distinct source registers: one register three times costs bank-conflict cycles)
  asm volatile("" : "+v"(fc1), "+v"(fc2));
  unsigned long long* tr = g.trace ? g.trace + (long)b * 64 : nullptr;
  int tri = 2;
#define V11_TRACE_RT() { if (tr && tid == 0 && tri < 40) tr[tri++] = __builtin_amdgcn_s_memrealtime(); }
  if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_memrealtime(); tr[1] = __builtin_amdgcn_s_memtime(); }
  cursor_tile();
  {
    __amdgpu_buffer_rsrc_t rx0 = V8_RSRC_X(), rw0 = V8_RSRC_W();
    cursor_next();
    __amdgpu_buffer_rsrc_t rx1 = V8_RSRC_X(), rw1 = V8_RSRC_W();
    cursor_next();
#pragma unroll
    for (int i = 0; i < 4; ++i) V11_DMA_X(rx0, dst, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) V11_DMA_W(rw0, dst, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) V11_DMA_X(rx1, dst + V7_STAGE, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) V11_DMA_W(rw1, dst + V7_STAGE, i);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < MTN) V7_LDSR(xf[0][i], xa0, i * 2048);
      V7_LDSR(wf[0][i], wa0, i * 2048);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  if (DBG && paused) {
    // the paused group's whole tile walk, kept apart from the computing path (a branch per step made the compiler park the
    // accumulators in scratch): the same steps, barriers and pieces, no MFMAs, no epilogue
    for (int t = first; t < c1; t += nx) {
      int m0, n0;
      tile_origin(t, m0, n0);
      for (int kt = 0; kt < nk; ++kt) {
        __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
        cursor_next();
        if (kt == 0) {
          const int bn_ = g.N - n0 < 256 ? g.N - n0 : 256;
          __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bias ? g.bias + n0 : nullptr), 0, g.bias ? bn_ * 4 : 0, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(smem + 2 * V7_STAGE + wave * 1024), 16, lane * 16, 0, 0, 0);
        }
        V11_PSTEP_(8)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  for (int t = first; t < c1; t += nx) {
    int m0, n0;
    tile_origin(t, m0, n0);
    V11_TRACE_RT();
    {  // first K-step of the tile; the bias piece goes first, so the step's landing wait covers it too
      __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
      cursor_next();
      {
        const int bn_ = g.N - n0 < 256 ? g.N - n0 : 256;
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bias ? g.bias + n0 : nullptr), 0, g.bias ? bn_ * 4 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(smem + 2 * V7_STAGE + wave * 1024), 16, lane * 16, 0, 0, 0);
      }
      V11_STEP_(8, V7_MFMA0)
    }
    for (int kt = 1; kt < nk; ++kt) {
      __amdgpu_buffer_rsrc_t rx = V8_RSRC_X(), rw = V8_RSRC_W();
      cursor_next();
      V11_STEP_(8, V7_MFMA)
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    V11_TRACE_RT();
    // the group's 128 x 256 half of the tile through the straight-line epilogue (its bias slot: 4 KiB per group)
    if (m0 + 128 * grp < g.M) v7_epilogue_fast<ACT, HAS_R, MTN>(g, acc, lane, wq, m0 + 128 * grp, n0, lds0 + 4096 * grp);
    V11_TRACE_RT();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tr && tid == 0) { tr[62] = __builtin_amdgcn_s_memrealtime(); tr[63] = __builtin_amdgcn_s_memtime(); }
}

int vt_gemm_persistent_cus();   // gemm_v7.hip
int vt_gemm_v8_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn);

template <int ACT>
static int launch_v11(const GemmArgs& g, hipStream_t stream) {
  GemmArgs ga = g;
  ga.tiles_n = (g.N + 255) / 256;
  ga.tiles_m = (g.M + 255) / 256;
  if (ACT == ACT_MUL && !g.R) return VT_ERR_NULL;
  const int cus = vt_gemm_persistent_cus();
  if (cus <= 0) return VT_ERR_HIP;
  const long tiles = (long)ga.tiles_m * ga.tiles_n;
  const int grid = (int)(tiles < cus ? tiles : cus);
  const bool has_r = g.R || ACT == ACT_MUL;
  void (*kern)(GemmArgs) = has_r ? gemm_nt_bf16_v11<ACT, true> : gemm_nt_bf16_v11<ACT, ACT == ACT_MUL>;
  if (g.trace && ACT == ACT_NONE && !has_r) kern = gemm_nt_bf16_v11<ACT_NONE, false, true>;   // tools/v8_trace.py: with the paused-group modes
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V11_LDS_BYTES) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), V11_LDS_BYTES, stream, ga);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// bf16 output in whole 64-column slabs without row remap only (the encoder's shapes); everything else goes to variant 16
int vt_gemm_v11_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream) {
  const bool fast = !out_f32 && (g.N & 63) == 0 && g.grp_rows == 0 && act != ACT_TANH;
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  if (!fast) return vt_gemm_v8_launch(g, act, out_f32, stream, 8);
  switch (act) {
    case ACT_NONE: return launch_v11<ACT_NONE>(g, stream);
    case ACT_GELU: return launch_v11<ACT_GELU>(g, stream);
    case ACT_MUL: return launch_v11<ACT_MUL>(g, stream);
    default: return VT_ERR_UNSUPPORTED;
  }
}
