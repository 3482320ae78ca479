// Row kernels of the deferred-LayerNorm inference path (see GemmArgs::ln_mode in gemm_common.hpp).
//   ln_apply_rows   the one place where a LayerNorm of that path is materialised: y = (v - mean) * rstd * gamma + beta from
//                   the fp16 stream v and its partial row statistics -- BertOutput.LayerNorm of the LAST layer
//                   (oscar/modeling_bert.py:120; the encoder's output, :161-169), written as bf16 (the pooler's and the
//                   heads' GEMM operand) and / or fp32 (what the caller is handed).
//   ln_stream_init  layer-0 input: the embedding output x0 (fp32) -> the stream (fp16), its bf16 copy and the identity
//                   statistics (mean 0, rstd 1: x0 is not normalised again), so that the first layer runs the same kernels
//                   as the others.
#include "common.hpp"

struct LnApplyArgs {
  const uint16_t* v; long ldv;   // fp16 rows
  const float* stats; int np; long stat_rows;   // [np][stat_rows][2]
  const float* gamma; const float* beta;
  bf16_t* y16; long ldy16;
  float* y32; long ldy32;
  long M; int H;
  float eps;
};

__device__ __forceinline__ float h2f(uint32_t bits) { return (float)__builtin_bit_cast(_Float16, (uint16_t)bits); }
__device__ __forceinline__ uint32_t f2h2(float lo, float hi) {   // saturating, as the GEMM epilogue packs the stream
  lo = __builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f);
  hi = __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f);
  return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)lo) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)hi) << 16);
}

// one wave per row, 4 columns per lane and pass
__global__ __launch_bounds__(256) void ln_apply_rows(LnApplyArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  float s = 0.f, q = 0.f;
  for (int p = 0; p < a.np; ++p) {
    const float* st = a.stats + ((long)p * a.stat_rows + row) * 2;
    s += st[0];
    q += st[1];
  }
  const float invH = 1.0f / (float)a.H;
  const float mean = s * invH;
  const float rstd = rsqrtf(fmaxf(q * invH - mean * mean, 0.f) + a.eps);
  const uint16_t* vr = a.v + row * a.ldv;
  for (int col = lane * 4; col < a.H; col += 256) {
    const u32x2 xw = *(const u32x2*)(vr + col);
    const f32x4 x = {h2f(xw[0] & 0xffffu), h2f(xw[0] >> 16), h2f(xw[1] & 0xffffu), h2f(xw[1] >> 16)};
    const f32x4 g = *(const f32x4*)(a.gamma + col), b = *(const f32x4*)(a.beta + col);
    f32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = (x[i] - mean) * rstd * g[i] + b[i];
    if (a.y32) *(f32x4*)(a.y32 + row * a.ldy32 + col) = y;
    if (a.y16) {
      u32x2 o;
      o[0] = pack_bf16x2(y[0], y[1]);
      o[1] = pack_bf16x2(y[2], y[3]);
      *(u32x2*)(a.y16 + row * a.ldy16 + col) = o;
    }
  }
}

int vt_ln_apply_dispatch(const void* v, long ldv, const float* stats, int np, long stat_rows, const float* gamma,
                         const float* beta, float eps, void* y16, long ldy16, float* y32, long ldy32, long M, int H,
                         hipStream_t stream) {
  if (!v || !stats || !gamma || !beta || (!y16 && !y32)) return VT_ERR_NULL;
  if (M <= 0 || H <= 0 || (H & 3) || np <= 0 || np > 8 || stat_rows < M) return VT_ERR_BAD_SHAPE;
  if ((ldv & 3) || (y16 && (ldy16 & 3)) || (y32 && (ldy32 & 3))) return VT_ERR_BAD_ALIGN;
  if (((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y32) & 15) return VT_ERR_BAD_ALIGN;
  if (((uintptr_t)y16 | (uintptr_t)v) & 7) return VT_ERR_BAD_ALIGN;
  LnApplyArgs a;
  a.v = (const uint16_t*)v; a.ldv = ldv; a.stats = stats; a.np = np; a.stat_rows = stat_rows; a.gamma = gamma; a.beta = beta;
  a.y16 = (bf16_t*)y16; a.ldy16 = ldy16; a.y32 = y32; a.ldy32 = ldy32; a.M = M; a.H = H; a.eps = eps;
  hipLaunchKernelGGL(ln_apply_rows, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}

// x0 fp32 [M, H] -> fp16 stream + bf16 copy; statistics slice 0 = (0, H * (1 - eps)) so that mean = 0 and rstd = 1, the
// other slices 0
__global__ __launch_bounds__(256) void ln_stream_init(const float* __restrict__ x, long ldx, uint16_t* __restrict__ s16, long lds,
                                                      bf16_t* __restrict__ y16, long ldy, float* __restrict__ stats, int np,
                                                      long stat_rows, long M, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  for (int col = lane * 4; col < H; col += 256) {
    const f32x4 v = *(const f32x4*)(x + row * ldx + col);
    u32x2 o;
    o[0] = pack_bf16x2(v[0], v[1]);
    o[1] = pack_bf16x2(v[2], v[3]);
    *(u32x2*)(y16 + row * ldy + col) = o;
    u32x2 hh;
    hh[0] = f2h2(v[0], v[1]);
    hh[1] = f2h2(v[2], v[3]);
    *(u32x2*)(s16 + row * lds + col) = hh;
  }
  if (lane < np) {
    float* st = stats + ((long)lane * stat_rows + row) * 2;
    st[0] = 0.f;
    st[1] = lane == 0 ? (float)H * (1.0f - eps) : 0.f;
  }
}

int vt_ln_stream_init_dispatch(const float* x, long ldx, void* s16, long lds, void* y16, long ldy, float* stats, int np,
                               long stat_rows, long M, int H, float eps, hipStream_t stream) {
  if (!x || !s16 || !y16 || !stats) return VT_ERR_NULL;
  if (M <= 0 || H <= 0 || (H & 3) || np <= 0 || np > 8 || stat_rows < M) return VT_ERR_BAD_SHAPE;
  if ((ldx & 3) || (ldy & 3) || (lds & 3) || ((uintptr_t)x & 15) || (((uintptr_t)y16 | (uintptr_t)s16) & 7)) return VT_ERR_BAD_ALIGN;
  hipLaunchKernelGGL(ln_stream_init, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x, ldx, (uint16_t*)s16, lds,
                     (bf16_t*)y16, ldy, stats, np, stat_rows, M, H, eps);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
