// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in memory

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte asynchronous global -> LDS copy (global_load_lds_dwordx4).  The LDS destination is
// wave-uniform base + lane*16; the global source address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even f32 -> bf16 via the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// two round-to-nearest-even conversions in ONE v_cvt_pk_bf16_f32 (converting the halves separately costs two
// converts, a shift and an or: 4 instructions per output dword in every epilogue)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2_t v = __builtin_convertvector((f32x2){lo, hi}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- fp16 (the higher-precision copies of the residual stream: 11 significant bits against bf16's 8) ----------------
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float f16lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
__device__ __forceinline__ float f16hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
// two fp32 -> one dword of fp16, round to nearest even, saturating at the largest finite fp16 (a pre-LayerNorm sum of that
// size does not occur in a BERT-class model; it must not become an infinity that the next LayerNorm turns into NaN)
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  lo = __builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f);
  hi = __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f);
  const f16x2_t v = __builtin_convertvector((f32x2){lo, hi}, f16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint16_t f32_to_f16bits(float f) {
  return __builtin_bit_cast(uint16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f));
}
__device__ __forceinline__ float f16bits_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), enough for fp32-tolerance parity
// (1e-3) and far below bf16 resolution; ~12 VALU ops instead of libm's erff.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float y = 1.061405429f;
  y = y * t - 1.453152027f;
  y = y * t + 1.421413741f;
  y = y * t - 0.284496736f;
  y = y * t + 0.254829592f;
  y = 1.0f - y * t * __expf(-ax * ax);
  return copysignf(y, x);
}
// erf-GELU, the reference's hidden_act == "gelu": 0.5 x (1 + erf(x / sqrt 2))
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// d/dx of the erf-GELU: Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// erf-GELU and its derivative in one go (training forward saves the derivative): they share the reciprocal, the
// polynomial and the exponential (exp(-x^2/2) is both erf's tail factor and the normal pdf); same values as
// gelu_erf / gelu_erf_grad
__device__ __forceinline__ void gelu_erf_both(float x, float& g, float& dg) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float y = 1.061405429f;
  y = y * t - 1.453152027f;
  y = y * t + 1.421413741f;
  y = y * t - 0.284496736f;
  y = y * t + 0.254829592f;
  const float e = __expf(-ax * ax);
  const float erf_x = copysignf(1.0f - y * t * e, x);
  const float cdf = 0.5f * (1.0f + erf_x);
  g = x * cdf;
  dg = cdf + x * (0.3989422804014327f * e);
}
// The same for two elements with packed fp32 arithmetic (v_pk_mul/fma/add_f32: two lanes' worth per issue slot);
// the reciprocal, the exponential and the sign transfer stay per element.  At one wave per SIMD the GEMM epilogue
// has nothing to overlap its VALU work with, so the instruction count is its time.
__device__ __forceinline__ void gelu_erf_both2(f32x2 x, f32x2& g, f32x2& dg) {
  const f32x2 ax = __builtin_elementwise_abs(x) * 0.70710678118654752f;
  const f32x2 den = ax * 0.3275911f + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 y = t * 0.5307027145f - 0.7265760135f;      // the erf polynomial with its coefficients halved
  y = y * t + 0.7107068705f;
  y = y * t - 0.142248368f;
  y = y * t + 0.127414796f;
  const f32x2 q = y * t;                              // 0.5 * (1 - |erf|) / exp(-x^2/2)
  const f32x2 a2 = (ax * ax) * -1.4426950408889634f;  // -x^2/2 in base 2
  const f32x2 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
  const f32x2 h = 0.5f - q * e;                       // 0.5 * |erf|
  const f32x2 hs = {copysignf(h[0], x[0]), copysignf(h[1], x[1])};
  const f32x2 cdf = hs + 0.5f;
  g = x * cdf;
  dg = cdf + x * (e * 0.3989422804014327f);
}
// Four elements at a time: the same arithmetic as gelu_erf_both2 on two independent pairs written as 4-vector
// expressions, so consecutive packed instructions belong to different dependency chains (a packed fp32 result needs a
// wait state before a dependent VALU read; with a single pair hipcc fills it with s_nop, 20 % of the epilogue's issue
// slots at one wave per SIMD).
__device__ __forceinline__ void gelu_erf_both4(f32x4 x, f32x4& g, f32x4& dg) {
  const f32x4 ax = __builtin_elementwise_abs(x) * 0.70710678118654752f;
  const f32x4 den = ax * 0.3275911f + 1.0f;
  const f32x4 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1]), __builtin_amdgcn_rcpf(den[2]),
                   __builtin_amdgcn_rcpf(den[3])};
  f32x4 y = t * 0.5307027145f - 0.7265760135f;
  y = y * t + 0.7107068705f;
  y = y * t - 0.142248368f;
  y = y * t + 0.127414796f;
  const f32x4 q = y * t;
  const f32x4 a2 = (ax * ax) * -1.4426950408889634f;
  const f32x4 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1]), __builtin_amdgcn_exp2f(a2[2]),
                   __builtin_amdgcn_exp2f(a2[3])};
  const f32x4 h = 0.5f - q * e;
  const f32x4 hs = {copysignf(h[0], x[0]), copysignf(h[1], x[1]), copysignf(h[2], x[2]), copysignf(h[3], x[3])};
  const f32x4 cdf = hs + 0.5f;
  g = x * cdf;
  dg = cdf + x * (e * 0.3989422804014327f);
}
// erf-GELU without transcendentals, four elements at a time, for the bf16 epilogues of the INFERENCE path (the training
// forward keeps gelu_erf_both4: a polynomial form with the derivative measured no faster there -- 314-320 us either way at
// M = 51 200: Phi and x phi(x) as two polynomials are as many issue slots as the reciprocal and the exponential both results
// share; the fp32 path keeps gelu_erf).  Phi(x) = 0.5 + xc P(xc^2) with
// xc = x clamped to +-4.5 and P a degree-8 minimax fit of (Phi(x) - 0.5) / x over |x| <= 4.5 (weighted for the absolute
// error of Phi); evaluated in fp32 by Horner: |Phi - exact| <= 2.6e-5, |gelu - exact| <= 1.2e-4 over all x (checked on a
// 2e6-point grid over [-9, 9]; the rounding of the result to bf16 is 2e-3 relative).  Packed fma / mul on pairs of
// elements: ~8 issue slots per element against ~16 with the reciprocal and the exponential of the erf form -- with one wave
// per SIMD an epilogue's instruction count is its time.
__device__ __forceinline__ f32x4 gelu_poly4(f32x4 x) {
  f32x4 xc;
#pragma unroll
  for (int i = 0; i < 4; ++i) xc[i] = __builtin_amdgcn_fmed3f(x[i], -4.5f, 4.5f);
  const f32x4 t = xc * xc;
  f32x4 p = t * 3.805696024e-11f - 4.002319340e-09f;
  p = p * t + 1.846167212e-07f;
  p = p * t - 4.959739163e-06f;
  p = p * t + 8.727667042e-05f;
  p = p * t - 1.076739372e-03f;
  p = p * t + 9.729491361e-03f;
  p = p * t - 6.624043805e-02f;
  p = p * t + 3.988664888e-01f;
  f32x4 phi = xc * p + 0.5f;
#pragma unroll
  for (int i = 0; i < 4; ++i) phi[i] = fmaxf(phi[i], 0.f);   // the fit dips 1.4e-5 below zero at the clamp
  return x * phi;
}
__device__ __forceinline__ float tanh_fast(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); saturates cleanly for |x| large
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// A residual operand that is a LayerNorm never written out (GemmArgs::r_mean, vt_layer_acts::ln_residual_mode): row
// statistics [M] and the LayerNorm's weight / bias [N], all fp32
struct VtLnResidual { const float* mean; const float* rstd; const float* gamma; const float* beta; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum on the vector ALU alone (no LDS-crossbar shuffles, whose six dependent round trips per reduction were the
// latency chain of the LayerNorm kernels): four DPP adds inside a 16-lane row (xor 1, xor 2, mirror within 8, mirror within
// 16), then v_permlane16_swap / v_permlane32_swap pair the rows and the halves.  Every lane ends with the total; the order
// of the additions differs from wave_sum's butterfly (last-bit differences).
__device__ __forceinline__ float wave_sum_valu(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
  const auto p16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(p16[0]) + __uint_as_float(p16[1]);
  const auto p32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p32[0]) + __uint_as_float(p32[1]);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- dropout: counter-based hash RNG --------------------------------------------------------------
// One 32-bit hash serves TWO neighbouring elements: keep(element i) = 16 bits of hash32(site_seed, i >> 1) (low half
// for even i, high half for odd i) >= p * 2^16; kept values are scaled by 1/(1-p).  Nothing is stored: the backward
// kernels recompute the same mask from (seed, index).  site_seed mixes the step seed with a site id (layer, which
// dropout), see vt_site_seed.  The hash is two multiply / xor-shift rounds; the first multiply is linear in the index
// (amortised over a run, see vt_hash_pre); the finisher takes two 24-bit multiplies: v_mul_u32_u24 issues at the full
// rate, v_mul_lo_u32 at a quarter of it, and the attention kernels -- VALU-bound, half of it this hash -- have nothing
// to overlap it with (8 issue slots per word against 12 for one 32-bit multiply and three xor-shifts).  Measured on
// 4 M elements per stream (keep rate, lag correlations up to 65537, agreement between the streams of two heads): at the
// sampling noise, 0.0006 / 0.0013, where the 32-bit-multiply finisher showed 0.004 / 0.002.
struct DropCfg {
  uint32_t thresh;   // p * 2^16, 0 = no dropout
  uint32_t seed;     // site seed
  float scale;       // 1 / (1 - p)
};
// hash32(seed, idx) = fin((idx + seed) * C1).  The first multiply is LINEAR in idx, so a run of pair indices
// base + k needs it once: pre(base + k) = pre(base) + k * C1 (an add of a constant); only the second multiply of
// the finishing rounds is paid per hash word.
#define VT_HASH_C1 0x9E3779B1u
__host__ __device__ __forceinline__ uint32_t vt_hash_pre(uint32_t seed, uint32_t idx) { return (idx + seed) * VT_HASH_C1; }
__host__ __device__ __forceinline__ uint32_t vt_hash_fin(uint32_t x) {
  x ^= x >> 15;
  x = (x & 0xffffffu) * 0x85EBCBu;   // low 32 bits of a 24 x 24-bit product (v_mul_u32_u24)
  x ^= x >> 12;
  x = (x & 0xffffffu) * 0xC2B2AFu;
  x ^= x >> 16;
  return x;
}
// the two 16-bit fields of a hash word against the threshold, without extracting them: the high field decides
// h >= thresh << 16 on the whole word, the low field the same after a shift
__host__ __device__ __forceinline__ bool vt_keep_lo(uint32_t h, uint32_t thresh) { return (h << 16) >= (thresh << 16); }
__host__ __device__ __forceinline__ bool vt_keep_hi(uint32_t h, uint32_t thresh) { return h >= (thresh << 16); }
__host__ __device__ __forceinline__ uint32_t vt_hash32(uint32_t seed, uint32_t idx) { return vt_hash_fin(vt_hash_pre(seed, idx)); }
__host__ __device__ __forceinline__ bool vt_keep(const DropCfg& d, uint32_t idx) {
  const uint32_t h = vt_hash32(d.seed, idx >> 1);
  return (idx & 1u) ? vt_keep_hi(h, d.thresh) : vt_keep_lo(h, d.thresh);
}
// elements idx (even) and idx + 1 from one hash
__host__ __device__ __forceinline__ void vt_keep2(const DropCfg& d, uint32_t idx_even, bool& k0, bool& k1) {
  const uint32_t h = vt_hash32(d.seed, idx_even >> 1);
  k0 = vt_keep_lo(h, d.thresh);
  k1 = vt_keep_hi(h, d.thresh);
}
// v[0..N) *= keep / (1-p) for N consecutive elements starting at e0 (N even); pairs share a hash when e0 is even
// the two keep flags of the pair whose pre-multiplied index is x (see vt_hash_pre)
__host__ __device__ __forceinline__ void vt_keep2_pre(const DropCfg& d, uint32_t x, bool& k0, bool& k1) {
  const uint32_t h = vt_hash_fin(x);
  k0 = vt_keep_lo(h, d.thresh);
  k1 = vt_keep_hi(h, d.thresh);
}
template <int N>
__device__ __forceinline__ void vt_drop_run(const DropCfg& d, uint32_t e0, float (&v)[N]) {
  if ((e0 & 1u) == 0) {
    const uint32_t x0 = vt_hash_pre(d.seed, e0 >> 1);   // consecutive pairs: one multiply for the run
#pragma unroll
    for (int i = 0; i < N; i += 2) {
      bool k0, k1;
      vt_keep2_pre(d, x0 + (uint32_t)(i >> 1) * VT_HASH_C1, k0, k1);
      v[i] = k0 ? v[i] * d.scale : 0.f;
      v[i + 1] = k1 ? v[i + 1] * d.scale : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = vt_keep(d, e0 + i) ? v[i] * d.scale : 0.f;
  }
}
__host__ __device__ __forceinline__ uint32_t vt_site_seed(uint64_t step_seed, uint32_t site) {
  uint64_t x = step_seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return (uint32_t)(x >> 32);
}
__host__ __forceinline__ DropCfg vt_make_drop(float p, uint64_t step_seed, uint32_t site) {
  DropCfg d;
  d.thresh = p > 0.f ? (uint32_t)(p * 65536.0f) : 0u;
  d.seed = vt_site_seed(step_seed, site);
  d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}
// ---- attention-probability dropout (oscar/modeling_bert.py:62): one hash word per FOUR neighbouring keys ------------------
// The attention forward is bound by vector issue and half of its slots were this decision.  For the attention sites a hash
// word serves the keys 4m .. 4m+3 of a query -- byte j against an 8-bit threshold, one SDWA byte compare each, no shifts --
// which needs the row pitch of the element index (q * pitch + key) to be a multiple of 4 (the sequence's length rounded up)
// and quantises the drop probability to 1/256: p = 0.1 runs as 26/256 = 0.1016, and the scale 1/(1-p) uses the quantised
// value (unbiased).  DropCfg::thresh then holds p * 2^8.  Everything else (hidden-state dropout in the GEMM epilogues, the
// embeddings) keeps the 16-bit pairs above.
//
// EXACT-p FORM (round 6; the DEFAULT since ABI 12, vt_set_attn_dropout_bits(8) / VT_ATTN_DROPOUT_BITS=8 selects the form
// above): the attention sites take the 16-bit pairs as well -- element idx = q * pitch + key reads field idx & 1 of
// hash32(seed, idx >> 1), the function vt_keep of every other site -- so p = 0.1 runs as 6 554 / 65 536 = 0.100006 at twice
// the hash words per score tile (measured: attention forward 107 -> 112 us per layer at B = 256, the step +0.15 %).
// DropCfg::thresh then holds (p * 2^16) << 16, i.e. any value above 255 says "wide" to the kernels (vt_attn_wide) and both
// fields compare against it without extraction (vt_keep_hi / vt_keep_lo's form).
__host__ __device__ __forceinline__ bool vt_attn_wide(const DropCfg& d) { return d.thresh > 255u; }
__host__ __device__ __forceinline__ bool vt_keep_attn(const DropCfg& d, uint32_t idx) {
  if (vt_attn_wide(d)) {
    const uint32_t h = vt_hash32(d.seed, idx >> 1);
    return ((idx & 1u) ? h : (h << 16)) >= d.thresh;
  }
  return ((vt_hash32(d.seed, idx >> 2) >> (8u * (idx & 3u))) & 0xffu) >= d.thresh;
}
// p in (0, 1/512) runs as 1/256 (never silently as "no dropout"); p > 255.5/256 -- where the quantised value would be 1, the
// scale 1 / (1 - p) infinite and every output 0 * inf -- is refused by the entry points (vt_attn_drop_ok): VT_ERR_UNSUPPORTED.
// bits = 8 (default) or 16 (exact-p mode: steps of 1/65536, the same rules at that step).
__host__ __forceinline__ bool vt_attn_drop_ok(float p, int bits = 8) {
  const float n = bits == 16 ? 65536.0f : 256.0f;
  return p >= 0.f && p * n + 0.5f < n;
}
__host__ __forceinline__ uint32_t vt_attn_drop_steps(float p, int bits) {   // p in steps of 2^-bits, 1 .. 2^bits - 1
  const float n = bits == 16 ? 65536.0f : 256.0f;
  const uint32_t top = bits == 16 ? 65535u : 255u;
  uint32_t q = (uint32_t)(p * n + 0.5f);
  return q < 1u ? 1u : (q > top ? top : q);
}
__host__ __forceinline__ float vt_attn_drop_p(float p, int bits = 8) {
  if (!(p > 0.f)) return 0.f;
  return (float)vt_attn_drop_steps(p, bits) / (bits == 16 ? 65536.0f : 256.0f);
}
__host__ __forceinline__ DropCfg vt_make_drop_attn(float p, uint64_t step_seed, uint32_t site, int bits = 8) {
  DropCfg d;
  const float pq = vt_attn_drop_p(p, bits);
  d.thresh = pq > 0.f ? (bits == 16 ? vt_attn_drop_steps(p, 16) << 16 : vt_attn_drop_steps(p, 8)) : 0u;
  d.seed = vt_site_seed(step_seed, site);
  d.scale = pq > 0.f ? 1.0f / (1.0f - pq) : 1.0f;
  return d;
}
// ---- weight prefetch (round 6; rowops.hip has the why) -----------------------------------------------------------------
// Up to four byte ranges that a kernel's spare workgroups (or a launch of their own) read and drop, so that the lines sit in
// the Infinity Cache when the next GEMM's one round of tiles asks for them.
struct PrefetchArgs {
  const void* p[4];
  long bytes[4];
  int n;
};
#ifdef __HIPCC__
// workgroup `wg` of `nwg` (256 lanes each): 16 KiB blocks round-robin, four 16-byte loads per lane in flight
__device__ __forceinline__ void vt_prefetch_role(const PrefetchArgs& a, int wg, int nwg) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  const int lane = threadIdx.x & 255;
  for (int r = 0; r < a.n; ++r) {
    const char* base = (const char*)a.p[r];
    const long bytes = a.bytes[r] & ~15L;
    const long nblk = (bytes + 16383) >> 14;
    for (long blk = wg; blk < nblk; blk += nwg) {
      const long off = (blk << 14) + lane * 16;
      u4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
      if (off < bytes) v0 = *(const u4*)(base + off);
      if (off + 4096 < bytes) v1 = *(const u4*)(base + off + 4096);
      if (off + 8192 < bytes) v2 = *(const u4*)(base + off + 8192);
      if (off + 12288 < bytes) v3 = *(const u4*)(base + off + 12288);
      asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3));   // the loads must be issued and waited for; nothing is kept
    }
  }
}
#endif

// dropout sites: layer l uses 8*l + {0: attention probs, 1: attention.output dropout, 2: output dropout};
// 0xE0 = embeddings, 0xE1 = image embedding
#define VT_SITE_ATTN(l) (8u * (l) + 0u)
#define VT_SITE_SELFOUT(l) (8u * (l) + 1u)
#define VT_SITE_OUT(l) (8u * (l) + 2u)
#define VT_SITE_EMB 0xE0u
#define VT_SITE_IMG 0xE1u

// ---- per-device host-side state ---------------------------------------------------------------------------
// One process may drive several GPUs from several threads (torch.nn.DataParallel, pretrain.py:93-94): everything the
// host side remembers between calls is kept PER DEVICE (the calling thread's current device) and behind atomics.
#define VT_MAX_DEVICES 64
inline int vt_current_device() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < VT_MAX_DEVICES) ? d : -1;
}
// compute units of the current device (0 on error)
inline int vt_device_cus() {
  static std::atomic<int> cus[VT_MAX_DEVICES];
  const int d = vt_current_device();
  if (d < 0) return 0;
  int c = cus[d].load(std::memory_order_relaxed);
  if (!c) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) != hipSuccess) return 0;
    c = p.multiProcessorCount;
    cus[d].store(c, std::memory_order_relaxed);
  }
  return c;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel call site, device): function attributes belong to
// the device's copy of the code object, so a process-wide "done" flag would skip the second GPU.
struct VtLdsAttrOnce {
  std::atomic<bool> done[VT_MAX_DEVICES];
  bool set(const void* kern, int bytes) {
    const int d = vt_current_device();
    if (d < 0) return false;
    if (done[d].load(std::memory_order_acquire)) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done[d].store(true, std::memory_order_release);
    return true;
  }
};

// error codes of the C ABI (include/visitron_hip.h)
#define VT_OK 0
#define VT_ERR_BAD_SHAPE (-1)
#define VT_ERR_BAD_ALIGN (-2)
#define VT_ERR_NULL (-3)
#define VT_ERR_UNSUPPORTED (-4)
#define VT_ERR_HIP (-5)
