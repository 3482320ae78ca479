// Deferred-LayerNorm GEMMs of the inference path (GemmArgs::ln_mode 1 / 2, see gemm_common.hpp): launchers of the
// 256x256-tile kernels of gemm_v7_kernels.hpp with the epilogues of gemm_v7_ln_epilogue.hpp.  One translation unit of
// its own so that it compiles beside gemm_v7.hip.
#include "gemm_v7_kernels.hpp"

int vt_gemm_persistent_cus();   // gemm_v7.hip: the device's CU count less the reserved ones
int vt_gemm_v8_take_region(GemmArgs& g);   // gemm_v7.hip: shared-tile workspace of this launch (0: none registered)

template <int ACT, int LNM>
static int launch_ln(const GemmArgs& g, int persistent, int mtn, hipStream_t stream, bool shared_tiles) {
  GemmArgs ga = g;
  if (shared_tiles && !persistent) return VT_ERR_UNSUPPORTED;
  if (shared_tiles && mtn <= 5 && !vt_gemm_v8_take_region(ga)) return VT_ERR_UNSUPPORTED;   // (the region: 160- / 128-row tiles only)
  ga.tiles_n = (g.N + 255) / 256;
  ga.tiles_m = (g.M + 32 * mtn - 1) / (32 * mtn);
  const int tiles = ga.tiles_m * ga.tiles_n;
  void (*kern)(GemmArgs) = nullptr;
  int grid = tiles;
  if (persistent) {
    const int cus = vt_gemm_persistent_cus();
    if (cus <= 0) return VT_ERR_HIP;
    grid = tiles < cus ? tiles : cus;
    if (ga.sk_parts > 1) {   // the stream-K region: the whole grid (launch_v8 in gemm_v7.hip)
      if (cus > 8 * V8_SK_WGS_PER_XCD) ga.sk_parts = 0;
      else grid = cus;
    }
    switch (mtn) {
      case 8: kern = gemm_nt_bf16_v8<ACT, false, true, false, 8, LNM>; break;
      case 7: kern = gemm_nt_bf16_v8<ACT, false, true, false, 7, LNM>; break;
      case 6: kern = gemm_nt_bf16_v8<ACT, false, true, false, 6, LNM>; break;
      case 5: kern = gemm_nt_bf16_v8<ACT, false, true, false, 5, LNM>; break;
      case 4: kern = gemm_nt_bf16_v8<ACT, false, true, false, 4, LNM>; break;
      default: return VT_ERR_UNSUPPORTED;
    }
  } else {
    switch (mtn) {
      case 8: kern = gemm_nt_bf16_v7<ACT, false, true, false, 8, LNM>; break;
      case 7: kern = gemm_nt_bf16_v7<ACT, false, true, false, 7, LNM>; break;
      case 6: kern = gemm_nt_bf16_v7<ACT, false, true, false, 6, LNM>; break;
      default: return VT_ERR_UNSUPPORTED;
    }
  }
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V7_LDS_BYTES_LN) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), V7_LDS_BYTES_LN, stream, ga);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// variant: the numbering of vt_gemm_dispatch (15 / 22 / 23: one tile per workgroup on 256- / 224- / 192-row tiles; 16, 18 .. 21:
// persistent on 256- .. 128-row tiles)
int vt_gemm_ln_launch(const GemmArgs& g, int act, int variant, hipStream_t stream) {
  int persistent, mtn;
  const bool sk = variant >= 28 && variant <= 32;   // the persistent kernel sharing its left-over tiles (gemm_v7.hip)
  if (sk) variant = variant == 28 ? 16 : variant - 11;   // 29 .. 32 -> 18 .. 21
  switch (variant) {
    case 15: persistent = 0; mtn = 8; break;
    case 22: persistent = 0; mtn = 7; break;
    case 23: persistent = 0; mtn = 6; break;
    case 16: persistent = 1; mtn = 8; break;
    case 18: persistent = 1; mtn = 7; break;
    case 19: persistent = 1; mtn = 6; break;
    case 20: persistent = 1; mtn = 5; break;
    case 21: persistent = 1; mtn = 4; break;
    default: return VT_ERR_UNSUPPORTED;
  }
  if ((g.K & 63) || g.K < 128 || (g.N & 127) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31))
    return VT_ERR_UNSUPPORTED;
  if (g.ln_mode == 1) {
    if (act == ACT_NONE) return launch_ln<ACT_NONE, 1>(g, persistent, mtn, stream, sk);
    if (act == ACT_GELU) return launch_ln<ACT_GELU, 1>(g, persistent, mtn, stream, sk);
    return VT_ERR_UNSUPPORTED;
  }
  if (g.ln_mode == 2 && act == ACT_NONE) return launch_ln<ACT_NONE, 2>(g, persistent, mtn, stream, sk);
  return VT_ERR_UNSUPPORTED;
}
