// NT GEMM, 256x256 tile with 128x128 wave tiles and AGPR accumulators (variants 15, 16, 18 .. 23 of vt_gemm_dispatch):
// launchers of the kernels in gemm_v7_kernels.hpp with the plain epilogues.
#include "gemm_v7_kernels.hpp"

// Compute units the persistent grid leaves free (vt_gemm_reserve_cus): with a collective running beside the backward (one
// process per GPU, RCCL kernels on a few CUs) a persistent workgroup mapped onto a busy CU would stall its share of the
// tiles for a kernel time; launched with cus - k workgroups the grid fits beside it.  0 by default.
static std::atomic<int> g_reserved_cus{0};
void vt_gemm_set_reserved_cus(int k) { g_reserved_cus.store(k < 0 ? 0 : k, std::memory_order_relaxed); }
int vt_gemm_persistent_cus() {
  const int cus = vt_device_cus();   // of the calling thread's current device
  if (cus <= 0) return -1;
  const int k = g_reserved_cus.load(std::memory_order_relaxed);
  return cus - k >= 8 ? cus - k : cus;
}
static int v8_grid(int tiles) {
  const int cus = vt_gemm_persistent_cus();
  if (cus <= 0) return -1;
  return tiles < cus ? tiles : cus;
}

template <int ACT, bool OUT_F32>
static int launch_v8(const GemmArgs& g, hipStream_t stream, int mtn) {
  GemmArgs g8 = g;
  g8.tiles_n = (g.N + 255) / 256;
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  // bf16 output in whole 64-column slabs without row remap: the straight-line epilogue, specialised on the residual.
  // (Round 5 kept residual GEMMs with N % 128 != 0 away from it: a wave whose second column half lies past N never waited
  // for that half's ring loads.  Round 6 drains the ring in the kernel -- V7_HALF's else branch -- and the guard is gone;
  // tests/test_gpu_round5.py runs N = 832 / 384 / 640 with a residual and with ACT_MUL on every 256x256-tile variant.)
  const bool fast = !OUT_F32 && (g.N & 63) == 0 && g.grp_rows == 0;
  if (ACT == ACT_MUL && !g.R) return VT_ERR_NULL;
  if (!fast || OUT_F32 || ACT == ACT_TANH) mtn = 8;   // the shorter tiles exist for the encoder's own (bf16, fast-epilogue) shapes
  else if (g.r_mean && mtn == 8) mtn = 7;             // a rebuilt LayerNorm residual: not in the 256-row instantiation (v7_epilogue_fast)
  const int th = 32 * mtn;
  g8.tiles_m = (g.M + th - 1) / th;
  const int grid = v8_grid(g8.tiles_m * g8.tiles_n);
  if (grid <= 0) return VT_ERR_HIP;
  const bool has_r = g.R || ACT == ACT_MUL;
  void (*kern)(GemmArgs) = nullptr;
  if (!fast) kern = gemm_nt_bf16_v8<ACT, OUT_F32, false, false>;
  else if (mtn == 8) kern = has_r ? gemm_nt_bf16_v8<ACT, OUT_F32, !OUT_F32, true> : gemm_nt_bf16_v8<ACT, OUT_F32, !OUT_F32, ACT == ACT_MUL>;
#define V8_PICK(MT)                                                                                                  \
  else if (mtn == MT) kern = has_r ? gemm_nt_bf16_v8<ACT, OUT_F32, !OUT_F32, true, (OUT_F32 || ACT == ACT_TANH) ? 8 : MT>   \
                                   : gemm_nt_bf16_v8<ACT, OUT_F32, !OUT_F32, ACT == ACT_MUL, (OUT_F32 || ACT == ACT_TANH) ? 8 : MT>;
  V8_PICK(7) V8_PICK(6) V8_PICK(5) V8_PICK(4)
#undef V8_PICK
  else return VT_ERR_UNSUPPORTED;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V7_LDS_BYTES) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), V7_LDS_BYTES, stream, g8);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

template <int ACT, bool OUT_F32>
static int launch_v7(const GemmArgs& g, hipStream_t stream, int mtn) {
  GemmArgs g7 = g;
  g7.tiles_n = (g.N + 255) / 256;
  // operand panels are addressed with 32-bit byte offsets inside a tile's row panel
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  const bool fast = !OUT_F32 && (g.N & 63) == 0 && g.grp_rows == 0;   // as in launch_v8
  if (ACT == ACT_MUL && !g.R) return VT_ERR_NULL;
  if (!fast || OUT_F32 || ACT == ACT_TANH) mtn = 8;
  g7.tiles_m = (g.M + 32 * mtn - 1) / (32 * mtn);
  const bool has_r = g.R || ACT == ACT_MUL;
  void (*kern)(GemmArgs) = nullptr;
  if (!fast) kern = gemm_nt_bf16_v7<ACT, OUT_F32, false, false>;
  else if (mtn == 8) kern = has_r ? gemm_nt_bf16_v7<ACT, OUT_F32, !OUT_F32, true> : gemm_nt_bf16_v7<ACT, OUT_F32, !OUT_F32, ACT == ACT_MUL>;
#define V7_PICK(MT)                                                                                                  \
  else if (mtn == MT) kern = has_r ? gemm_nt_bf16_v7<ACT, OUT_F32, !OUT_F32, true, (OUT_F32 || ACT == ACT_TANH) ? 8 : MT>   \
                                   : gemm_nt_bf16_v7<ACT, OUT_F32, !OUT_F32, ACT == ACT_MUL, (OUT_F32 || ACT == ACT_TANH) ? 8 : MT>;
  V7_PICK(7) V7_PICK(6)
#undef V7_PICK
  else return VT_ERR_UNSUPPORTED;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V7_LDS_BYTES) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(g7.tiles_m * g7.tiles_n * (g7.ksplit > 1 ? g7.ksplit : 1)), dim3(256), V7_LDS_BYTES, stream, g7);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_gemm_v8_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn) {
#ifdef V7_ONE
  return launch_v8<ACT_NONE, false>(g, stream, mtn);
#else
  switch (act * 2 + (out_f32 ? 1 : 0)) {
    case 0: return launch_v8<ACT_NONE, false>(g, stream, mtn);
    case 1: return launch_v8<ACT_NONE, true>(g, stream, mtn);
    case 2: return launch_v8<ACT_GELU, false>(g, stream, mtn);
    case 3: return launch_v8<ACT_GELU, true>(g, stream, mtn);
    case 4: return launch_v8<ACT_TANH, false>(g, stream, mtn);
    case 5: return launch_v8<ACT_TANH, true>(g, stream, mtn);
    case 6: return launch_v8<ACT_MUL, false>(g, stream, mtn);
    case 7: return launch_v8<ACT_MUL, true>(g, stream, mtn);
    default: return VT_ERR_UNSUPPORTED;
  }
#endif
}

int vt_gemm_v7_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn) {
#ifdef V7_ONE
  return launch_v7<ACT_NONE, false>(g, stream, mtn);
#else
  switch (act * 2 + (out_f32 ? 1 : 0)) {
    case 0: return launch_v7<ACT_NONE, false>(g, stream, mtn);
    case 1: return launch_v7<ACT_NONE, true>(g, stream, mtn);
    case 2: return launch_v7<ACT_GELU, false>(g, stream, mtn);
    case 3: return launch_v7<ACT_GELU, true>(g, stream, mtn);
    case 4: return launch_v7<ACT_TANH, false>(g, stream, mtn);
    case 5: return launch_v7<ACT_TANH, true>(g, stream, mtn);
    case 6: return launch_v7<ACT_MUL, false>(g, stream, mtn);
    case 7: return launch_v7<ACT_MUL, true>(g, stream, mtn);
    default: return VT_ERR_UNSUPPORTED;
  }
#endif
}
