// NT GEMM, 256x256 tile with 128x128 wave tiles and AGPR accumulators (variants 15, 16, 18 .. 23 of vt_gemm_dispatch):
// launchers of the kernels in gemm_v7_kernels.hpp with the plain epilogues.
#include <atomic>
#include <cstdlib>
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
// ---- workspace of the stream-K region (GemmArgs::sk_parts) ------------------------------------------------------------
// Caller-owned device memory, registered per device (vt_gemm_set_workspace): `regions` regions of V8_SK_REGION_BYTES, each
// = 8 x 32 workgroup slots x 256 KiB of fp32 accumulators, then 256 arrival counters and one error counter (the
// caller hands the memory over ZEROED; the kernels leave the counters at zero).  Launches take the regions round-robin, so
// that many launches may be in flight on different streams of one device; launches of one stream never overlap.
struct SkWorkspace { char* base; int regions; };
static SkWorkspace g_sk_ws[VT_MAX_DEVICES];
static std::atomic<unsigned> g_sk_ctr{0};
int vt_gemm_set_workspace_impl(void* base, long bytes) {
  const int dev = vt_current_device();
  if (dev < 0 || dev >= VT_MAX_DEVICES) return VT_ERR_HIP;
  if (base && (((uintptr_t)base & 255) || bytes < V8_SK_REGION_BYTES)) return VT_ERR_BAD_ALIGN;
  g_sk_ws[dev].base = (char*)base;
  g_sk_ws[dev].regions = base ? (int)(bytes / V8_SK_REGION_BYTES) : 0;
  return VT_OK;
}
long vt_gemm_workspace_region_bytes_impl() { return V8_SK_REGION_BYTES; }
int vt_gemm_has_workspace() {
  const int dev = vt_current_device();
  return dev >= 0 && dev < VT_MAX_DEVICES && g_sk_ws[dev].regions > 0;
}
// Sum of the regions' error counters (bounded waits of a finishing workgroup that ran out), cleared on read.  Blocking.
int vt_gemm_shared_tile_timeouts_impl(unsigned* out) {
  *out = 0;
  const int dev = vt_current_device();
  if (dev < 0 || dev >= VT_MAX_DEVICES || g_sk_ws[dev].regions <= 0) return VT_OK;
  for (int r = 0; r < g_sk_ws[dev].regions; ++r) {
    unsigned v = 0;
    char* p = g_sk_ws[dev].base + (long)(r + 1) * V8_SK_REGION_BYTES - 4096 + 2048;
    if (hipMemcpy(&v, p, 4, hipMemcpyDeviceToHost) != hipSuccess) return VT_ERR_HIP;
    if (v) {
      const unsigned zero = 0;
      if (hipMemcpy(p, &zero, 4, hipMemcpyHostToDevice) != hipSuccess) return VT_ERR_HIP;
      *out += v;
    }
  }
  return VT_OK;
}
// the regions' error counters on the current device (vt_step_counters): how many, at most `max`
int vt_gemm_sk_counter_ptrs(unsigned** ptrs, int max) {
  const int dev = vt_current_device();
  if (dev < 0 || dev >= VT_MAX_DEVICES || g_sk_ws[dev].regions <= 0) return 0;
  int n = 0;
  for (int r = 0; r < g_sk_ws[dev].regions && n < max; ++r)
    ptrs[n++] = (unsigned*)(g_sk_ws[dev].base + (long)(r + 1) * V8_SK_REGION_BYTES - 4096 + 2048);
  return n;
}
// fills the shared-tile fields of a launch's arguments; false: no workspace on this device
static bool v8_take_region(GemmArgs& g) {
  const int dev = vt_current_device();
  if (dev < 0 || dev >= VT_MAX_DEVICES || g_sk_ws[dev].regions <= 0) return false;
  static const int parts_env = [] { const char* e = getenv("VT_GEMM_SK"); return e ? atoi(e) : 2; }();   // 0 / 1: the region off
  char* reg = g_sk_ws[dev].base + (long)(g_sk_ctr.fetch_add(1) % (unsigned)g_sk_ws[dev].regions) * V8_SK_REGION_BYTES;
  g.sk_ws = (float*)reg;
  g.sk_sem = (int*)(reg + V8_SK_REGION_BYTES - 4096);
  g.sk_err = (unsigned*)(reg + V8_SK_REGION_BYTES - 4096 + 2048);
  g.sk_parts = parts_env;
  return true;
}
int vt_gemm_v8_take_region(GemmArgs& g) { return v8_take_region(g) ? 1 : 0; }   // gemm_v7_ln.hip
// one region of the workspace as plain scratch (the split-K planes of variant 33, gemm_bf16.hip); null: none / too small
void* vt_gemm_take_scratch(long bytes) {
  const int dev = vt_current_device();
  if (dev < 0 || dev >= VT_MAX_DEVICES || g_sk_ws[dev].regions <= 0 || bytes > V8_SK_REGION_BYTES - 4096) return nullptr;
  return g_sk_ws[dev].base + (long)(g_sk_ctr.fetch_add(1) % (unsigned)g_sk_ws[dev].regions) * V8_SK_REGION_BYTES;
}

static int v8_grid(int tiles) {
  const int cus = vt_gemm_persistent_cus();
  if (cus <= 0) return -1;
  return tiles < cus ? tiles : cus;
}

template <int ACT, bool OUT_F32>
static int launch_v8(const GemmArgs& g, hipStream_t stream, int mtn, bool shared_tiles) {
  GemmArgs g8 = g;
  g8.tiles_n = (g.N + 255) / 256;
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  // bf16 output in whole 64-column slabs without row remap: the straight-line epilogue, specialised on the residual.
  // (Round 5 kept residual GEMMs with N % 128 != 0 away from it: a wave whose second column half lies past N never waited
  // for that half's ring loads.  Round 6 drains the ring in the kernel -- V7_HALF's else branch -- and the guard is gone;
  // tests/test_gpu_round5.py runs N = 832 / 384 / 640 with a residual and with ACT_MUL on every 256x256-tile variant.)
  const bool fast = !OUT_F32 && (g.N & 63) == 0 && g.grp_rows == 0;
  if (ACT == ACT_MUL && !g.R) return VT_ERR_NULL;
  // variants 28 .. 32 need vt_gemm_set_workspace; the register-epilogue kernels do not share tiles (they run as 16 .. 21)
  if (shared_tiles && !vt_gemm_has_workspace()) return VT_ERR_UNSUPPORTED;
  if (!fast || OUT_F32 || ACT == ACT_TANH) mtn = 8;   // the shorter tiles exist for the encoder's own (bf16, fast-epilogue) shapes
  else if (g.r_mean && mtn == 8) mtn = 7;             // a rebuilt LayerNorm residual: not in the 256-row instantiation (v7_epilogue_fast)
  // the stream-K region exists in the fast-epilogue kernels on 160- and 128-row tiles (gemm_nt_bf16_v8: SK_OK); the other
  // variants of 28 .. 32 run as their plain twins
  if (shared_tiles && fast && mtn <= 5 && !v8_take_region(g8)) return VT_ERR_UNSUPPORTED;
  const int th = 32 * mtn;
  g8.tiles_m = (g.M + th - 1) / th;
  {   // experiment switch: VT_GEMM_REVERSE_K = k walks the tiles of launches with K >= k backwards (0 / unset: never)
    static const int rev_k = [] { const char* e = getenv("VT_GEMM_REVERSE_K"); return e ? atoi(e) : 0; }();
    g8.reverse = (rev_k > 0 && g.K >= rev_k) ? 1 : 0;
  }
  int grid = v8_grid(g8.tiles_m * g8.tiles_n);
  if (grid <= 0) return VT_ERR_HIP;
  // the stream-K region spreads a chunk's tiles over every workgroup of the XCD: launch the whole grid even for fewer
  // tiles than CUs (the region's slots are laid out for 32 workgroups per XCD)
  if (g8.sk_parts > 1) {
    const int cus = vt_gemm_persistent_cus();
    if (cus > 8 * V8_SK_WGS_PER_XCD) g8.sk_parts = 0;
    else grid = cus;
  }
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

// ---- variant 33: split-K over raw partial tiles + one epilogue kernel (see gemm_bf16.hip, launch_splitk_epi) -------------
// Workgroup (tile, mt, nh) of splitk_tiles_epilogue: the 16 mt-rows of each wave of the 256x256 tile, one 64-column half;
// thread tid reads, per copy, the 16-byte groups the GEMM's lane tid stored (coalesced), adds the copies in order and hands
// the 16 sums of a row to the register epilogue.
template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(256) void splitk_tiles_epilogue(GemmArgs g, int ksplit) {
  const int tile = blockIdx.x >> 4, mt = (blockIdx.x >> 1) & 7, nh = blockIdx.x & 1;
  const int bm = tile / g.tiles_n, bn = tile - bm * g.tiles_n;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, gq = lane >> 4, j = lane & 15;
  const int row = bm * 256 + 128 * wm + 16 * mt + j;
  const int nb = bn * 256 + 128 * wn + 64 * nh + 16 * gq;
  if (row >= g.M || nb >= g.N) return;
  const long stride = (long)g.tiles_m * g.tiles_n * (V8_SK_PART_BYTES / 4);   // floats between two copies of a tile
  const float* base = g.sk_ws + (long)tile * (V8_SK_PART_BYTES / 4) + (long)(8 * mt + 4 * nh) * 1024 + (long)tid * 4;
  f32x4 a[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) a[t] = *(const f32x4*)(base + t * 1024);
  int s = 1;
  for (; s + 1 < ksplit; s += 2) {   // two copies' loads in flight; the sum stays in copy order
    f32x4 u[4], v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u[t] = *(const f32x4*)(base + s * stride + t * 1024);
      v[t] = *(const f32x4*)(base + (s + 1) * stride + t * 1024);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[t][e] = (a[t][e] + u[t][e]) + v[t][e];
  }
  if (s < ksplit) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 u = *(const f32x4*)(base + s * stride + t * 1024);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[t][e] += u[e];
    }
  }
  const bool full = nb + 16 <= g.N;
  float bv[16];
  epi_load_bias(g, nb, full, bv);
  epi_row_direct<ACT, OUT_F32>(g, a, bv, row, nb, full);
}

template <int ACT, bool OUT_F32>
static int launch_splitk_tiles(const GemmArgs& g, int ks, hipStream_t stream) {
  GemmArgs gs = g;
  gs.tiles_m = (g.M + 255) / 256;
  gs.tiles_n = (g.N + 255) / 256;
  const int ntile = gs.tiles_m * gs.tiles_n;
  if ((g.K & 63) || 256L * g.lda * 2 + 2L * g.K >= (1L << 31) || 256L * g.ldw * 2 + 2L * g.K >= (1L << 31)) return VT_ERR_UNSUPPORTED;
  float* ws = (float*)vt_gemm_take_scratch((long)ks * ntile * V8_SK_PART_BYTES);
  if (!ws) return VT_ERR_UNSUPPORTED;
  gs.sk_ws = ws;
  gs.ksplit = ks;
  GemmArgs gk = gs;   // the K-step copies: no epilogue operands at all
  gk.bias = nullptr; gk.R = nullptr; gk.C2 = nullptr; gk.drop.thresh = 0; gk.r_mean = nullptr; gk.grp_rows = 0;
  auto kern = gemm_nt_bf16_v7<ACT_NONE, false, true, false, 8, 0>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V7_LDS_BYTES) != hipSuccess) return VT_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(ntile * ks), dim3(256), V7_LDS_BYTES, stream, gk);
  hipLaunchKernelGGL((splitk_tiles_epilogue<ACT, OUT_F32>), dim3(ntile * 16), dim3(256), 0, stream, gs, ks);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

int vt_gemm_splitk_tiles_launch(const GemmArgs& g, int act, int out_f32, int ks, hipStream_t stream) {
  switch (act * 2 + (out_f32 ? 1 : 0)) {
    case 0: return launch_splitk_tiles<ACT_NONE, false>(g, ks, stream);
    case 1: return launch_splitk_tiles<ACT_NONE, true>(g, ks, stream);
    case 2: return launch_splitk_tiles<ACT_GELU, false>(g, ks, stream);
    case 3: return launch_splitk_tiles<ACT_GELU, true>(g, ks, stream);
    case 4: return launch_splitk_tiles<ACT_TANH, false>(g, ks, stream);
    case 5: return launch_splitk_tiles<ACT_TANH, true>(g, ks, stream);
    case 6: return launch_splitk_tiles<ACT_MUL, false>(g, ks, stream);
    case 7: return launch_splitk_tiles<ACT_MUL, true>(g, ks, stream);
    default: return VT_ERR_UNSUPPORTED;
  }
}

int vt_gemm_v8_launch(const GemmArgs& g, int act, int out_f32, hipStream_t stream, int mtn, bool sk) {
#ifdef V7_ONE
  return launch_v8<ACT_NONE, false>(g, stream, mtn, sk);
#else
  switch (act * 2 + (out_f32 ? 1 : 0)) {
    case 0: return launch_v8<ACT_NONE, false>(g, stream, mtn, sk);
    case 1: return launch_v8<ACT_NONE, true>(g, stream, mtn, sk);
    case 2: return launch_v8<ACT_GELU, false>(g, stream, mtn, sk);
    case 3: return launch_v8<ACT_GELU, true>(g, stream, mtn, sk);
    case 4: return launch_v8<ACT_TANH, false>(g, stream, mtn, sk);
    case 5: return launch_v8<ACT_TANH, true>(g, stream, mtn, sk);
    case 6: return launch_v8<ACT_MUL, false>(g, stream, mtn, sk);
    case 7: return launch_v8<ACT_MUL, true>(g, stream, mtn, sk);
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
