// Fused multi-head self-attention forward over the mixed [text | region] sequence, head size 64.
//
// Replaces, per layer, oscar/modeling_bert.py:47-72 (CaptionBertSelfAttention.forward after the
// three projections): head split (:47-49), scores = Q.K^T (:52), / sqrt(64) (:53), + additive
// mask (:55), softmax over keys (:58), [* head_mask (:65-66)], context = P.V (:68), head merge
// (:70-72).  The [B,12,S,S] score/prob tensors and the four permute copies never exist in HBM.
// The additive mask is computed here from the caller's raw mask exactly as
// tasks/viewpoint_select/encoder.py:238-241 does: (1.0 - mask) * -10000.0 (any numeric mask).
//
// gfx950 design.  One workgroup = 8 waves = 256 queries of one (batch, head) (the whole sequence at
// S = 228, so K and V are staged once per head); each wave owns 32
// queries and keeps the QUERY on the MFMA lane for both products (v_mfma_f32_32x32x16_bf16):
//   S^T[key][query]  = K . Q^T        A = K rows (ds_read_b128 from a swizzled LDS tile), B = Q (registers)
//   O^T[d][query]   += V^T . P^T      B = P^T taken straight from the S^T accumulators (no LDS,
//                                     no lane movement), A = V^T via ds_read_b64_tr_b16 from the
//                                     row-major V tile (hardware transpose)
// so the softmax row statistics are lane-local (16 keys per lane per tile + one cross-half
// shuffle), and the normalised context row is stored as 2 x 32 contiguous bytes per lane.
// K and V of the (batch, head) are staged once per 256-key chunk by global_load_lds_dwordx4
// (swizzles applied on the source address).  Online softmax in fp32; the masked score is formed
// with the reference's roundings (fma(acc, 1/8, mask)) and (score - max) is exact before exp2.
#include "common.hpp"

struct AttnArgs {
  const bf16_t* qkv;        // [B*S, ld_qkv]  q | k | v, each nh*64 wide
  const float* mask;        // [B, S] raw mask (1 = attend) or additive bias, or null
  int mask_additive;        // 0: mask is raw, bias = (1-mask)*-10000;  1: mask already is the additive bias;
                            // 2: mask is an additive PER-QUERY bias [B, S, S] (the reference's 3-D attention_mask,
                            //    tasks/viewpoint_select/encoder.py:226-229); forward / probabilities only
  const float* head_scale;  // [nh] head_mask multipliers, or null
  bf16_t* ctx;              // [B*S, ld_ctx]
  float* lse;               // [B, nh, S] natural-log log-sum-exp of the masked scores (for backward), or null
  long ld_qkv, ld_ctx;
  int B, S, nh;
  float scale;              // 1 / sqrt(head_size)
  DropCfg drop;             // dropout on the attention probabilities (oscar/modeling_bert.py:62), the attention sites'
                            // form (common.hpp, vt_keep_attn): per (b,h) the seed is hash32(drop.seed, b*nh+h), the element
                            // index is q * S' + key with S' = the sequence's length rounded up to a MULTIPLE OF 4, and the
                            // keys 4m .. 4m+3 of a query share one hash word (byte j >= drop.thresh = p * 2^8).  (With
                            // pitch S an odd-length sequence -- every other one of a compacted batch -- paid a whole hash
                            // per element in round 3.)
  // compacted rows (training without the padding rows): sequence b holds seq_len[b] <= S rows starting at row
  // seq_start[b]; lse keeps its [B, nh, S] layout.  Null: every sequence has S rows, sequence b starts at row b * S.
  const int* seq_start;
  const int* seq_len;
  // Training: the keep decisions of the probability dropout, written out for the backward kernel (which otherwise spends a
  // third of its vector instructions re-deriving them from the hash).  keep_bits[((b*nh + h) * nqb + qb) * kpitch + key]:
  // bit j = keep(query 32 qb + j, key), nqb = ceil(S / 32), kpitch = S rounded up to 32.  Null: nothing written.
  uint32_t* keep_bits;
  // Weight prefetch riding along (round 6, common.hpp vt_prefetch_role; rowops.hip has the why): the workgroups of the grid's
  // z-slices B .. gridDim.z - 1 compute no attention but read `pf`'s byte ranges -- the weights of the GEMMs that follow --
  // and drop them, which leaves the lines in the Infinity Cache.  pf.n == 0 / gridDim.z == B: none.
  PrefetchArgs pf;
};

#define LOG2E 1.4426950408889634f
#define ATT_KCHUNK 256
#define ATT_SK 0
#define ATT_SV (ATT_KCHUNK * 128)
#define ATT_SBIAS (2 * ATT_KCHUNK * 128)
#define ATT_LDS_BYTES (2 * ATT_KCHUNK * 128 + ATT_KCHUNK * 4)

// Sixteen lanes of the VGPR `dst` = sixteen wave-uniform words (hipcc has no builtin for v_writelane_b32): ml[j] / mh[j] are
// the low / high halves of the lane mask of compare j (j = 0 .. 7), written to lanes La_j / Lb_j.  The values are lane masks
// v_cmp instructions have just written: a VALU write of an SGPR needs wait states before v_writelane reads it (measured:
// without them the first write of a group carried the previous compare's mask), and hipcc's hazard recognizer does not look
// inside an asm statement -- hence the s_nop, ONE for all sixteen (round 3 paid one per group of four).
#define ATT_WL_STR2(x) #x
#define ATT_WL_STR(x) ATT_WL_STR2(x)
#define ATT_WRITELANE16(dst, ml, mh, A0, B0, A1, B1, A2, B2, A3, B3, A4, B4, A5, B5, A6, B6, A7, B7)                        \
  asm volatile("s_nop 4\n\t"                                                                                               \
               "v_writelane_b32 %0, %1, " ATT_WL_STR(A0) "\n\tv_writelane_b32 %0, %2, " ATT_WL_STR(B0) "\n\t"                 \
               "v_writelane_b32 %0, %3, " ATT_WL_STR(A1) "\n\tv_writelane_b32 %0, %4, " ATT_WL_STR(B1) "\n\t"                 \
               "v_writelane_b32 %0, %5, " ATT_WL_STR(A2) "\n\tv_writelane_b32 %0, %6, " ATT_WL_STR(B2) "\n\t"                 \
               "v_writelane_b32 %0, %7, " ATT_WL_STR(A3) "\n\tv_writelane_b32 %0, %8, " ATT_WL_STR(B3) "\n\t"                 \
               "v_writelane_b32 %0, %9, " ATT_WL_STR(A4) "\n\tv_writelane_b32 %0, %10, " ATT_WL_STR(B4) "\n\t"                \
               "v_writelane_b32 %0, %11, " ATT_WL_STR(A5) "\n\tv_writelane_b32 %0, %12, " ATT_WL_STR(B5) "\n\t"               \
               "v_writelane_b32 %0, %13, " ATT_WL_STR(A6) "\n\tv_writelane_b32 %0, %14, " ATT_WL_STR(B6) "\n\t"               \
               "v_writelane_b32 %0, %15, " ATT_WL_STR(A7) "\n\tv_writelane_b32 %0, %16, " ATT_WL_STR(B7)                       \
               : "+v"(dst)                                                                                                   \
               : "s"(ml[0]), "s"(mh[0]), "s"(ml[1]), "s"(mh[1]), "s"(ml[2]), "s"(mh[2]), "s"(ml[3]), "s"(mh[3]),             \
                 "s"(ml[4]), "s"(mh[4]), "s"(ml[5]), "s"(mh[5]), "s"(ml[6]), "s"(mh[6]), "s"(ml[7]), "s"(mh[7]))

__device__ __forceinline__ bf16x8 tr_pair(const char* p) {
  // two transposed 4x16 block reads (keys k..k+3 and k+8..k+11), 8 bf16 per lane
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p + 1024));
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// KEEP: also write the dropout keep words (AttnArgs::keep_bits; training with probability dropout)
// MASK3: mask_additive == 2, an additive bias per (query, key) [B, S, S] (the reference's 3-D masks); the common kernels
// carry none of its per-element loads and branches
//
// The softmax runs on the RAW accumulators: the per-key bias is staged in LDS divided by the scale (bias / scale: exact
// for head size 64, scale = 1/8) and is the INITIAL VALUE of the score accumulators, so after the four MFMAs
//   acc = q.k + bias / scale,   score * log2(e) = acc * scale2,   scale2 = scale * log2(e) > 0,
// the running maximum is taken over acc itself and p = exp2(fma(acc, scale2, -max * scale2)): one fma per element where
// round 3 spent a fma (scale + bias) and a subtract.  (The reference rounds fma(q.k, 1/8, mask) once; here a masked key's
// q.k is added to -80 000 in fp32 -- its probability is 0 either way -- and an unmasked key's bias is 0: same values.)
// WIDE: the exact-p mode of the probability dropout (common.hpp: 16-bit fields, TWO hash words per four neighbouring keys)
template <bool KEEP, bool MASK3, bool WIDE = false>
__global__ __launch_bounds__(512, 4) void attention_fwd_d64(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h2 = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  if (b >= a.B) {   // a spare workgroup (uniform): two 256-lane halves, each one prefetch workgroup
    const int nwg = (int)((gridDim.z - a.B) * gridDim.y * gridDim.x) * 2;
    const int wg = (int)(((b - a.B) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2 + (tid >> 8);
    vt_prefetch_role(a.pf, wg, nwg);
    return;
  }
  const int Smax = a.S, H = a.nh * 64;
  // (1 .. a.S: the contract of include/visitron_hip.h; a length outside it is clamped into it rather than indexed with)
  const int S = a.seq_len ? (a.seq_len[b] < 1 ? 1 : (a.seq_len[b] > a.S ? a.S : a.seq_len[b])) : a.S;   // this sequence's rows
  const long row0 = a.seq_start ? (long)a.seq_start[b] : (long)b * a.S;
  if ((int)blockIdx.x * 256 >= S) return;                             // uniform: a query block past a short sequence
  const int q0 = blockIdx.x * 256 + wave * 32;
  const bool wave_active = q0 < S;  // wave-uniform

  const bf16_t* base = a.qkv + row0 * a.ld_qkv + head * 64;

  // Q fragments: B operand, lane (r, h2) holds Q[q0+r][16*ds + 8*h2 .. +7]
  bf16x8 qf[4];
  {
    int qr = q0 + r;
    qr = qr < S ? qr : S - 1;
    const bf16_t* qp = base + (long)qr * a.ld_qkv + 8 * h2;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = *(const bf16x8*)(qp + 16 * ds);
  }

  f32x16 o0, o1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;   // running maximum (of the raw accumulators) and running sum
  const float scale2 = a.scale * LOG2E;
  const float inv_scale = 1.0f / a.scale;
  DropCfg dr = a.drop;
  dr.seed = vt_hash32(a.drop.seed, (uint32_t)(b * a.nh + head));
  const uint32_t Sp = (uint32_t)(S + 3) & ~3u;                        // row pitch of the dropout element index: a multiple of 4
  const uint32_t q_elem = (uint32_t)(q0 + r) * Sp;
  const float* mrow3 = MASK3 ? a.mask + ((long)b * Smax + ((q0 + r) < S ? (q0 + r) : S - 1)) * Smax : nullptr;

  // lane-constant LDS offsets
  const int k_row_off = r * 128;                       // + kt*4096
  const int k_swz = (r >> 1) & 7;                      // (row>>1)&7 with row = 32kt + r  (32kt adds a multiple of 16 to row>>1)
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, dhalf = (lane >> 4) & 1;
  const int v_row = 4 * h2 + q4;                       // + 32kt + 16s2
  const int v_colb = 2 * (16 * (p4 & 1) + 8 * dhalf + 4 * (p4 >> 1));  // + 64dt
  const int v_swz = ((v_row >> 1) & 1) << 6;           // 64-B half swizzle; invariant under +16s2 / +32kt / +8

  for (int kc = 0; kc < S; kc += ATT_KCHUNK) {
    const int rows = (S - kc) < ATT_KCHUNK ? (S - kc) : ATT_KCHUNK;
    const int ntiles = (rows + 31) >> 5;
    if (kc > 0) __syncthreads();  // previous chunk fully consumed

    // ---- stage K, V (8-row pieces, one 1-KiB DMA each) and the additive bias ----
    for (int j = wave; j < ntiles * 4; j += 8) {
      const int row = 8 * j + (lane >> 3);
      int kr = kc + row;
      kr = kr < S ? kr : S - 1;
      const bf16_t* src = base + (long)kr * a.ld_qkv;
      const int cs = lane & 7;
      glds16(src + H + ((cs ^ ((row >> 1) & 7)) << 3), smem + ATT_SK + j * 1024);
      glds16(src + 2 * H + ((cs ^ (((row >> 1) & 1) << 2)) << 3), smem + ATT_SV + j * 1024);
    }
    if (tid < ATT_KCHUNK) {
      const int key = kc + tid;
      float bias = -INFINITY;
      if (key < S) {
        float add = 0.f;
        if (a.mask && !MASK3) {
          const float mval = a.mask[(long)b * Smax + key];
          add = a.mask_additive ? mval : (1.0f - mval) * -10000.0f;
        }
        bias = add * inv_scale;   // in units of the raw accumulator (see the kernel's header)
      }
      ((float*)(smem + ATT_SBIAS))[tid] = bias;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (wave_active) {
      // this wave's row of the keep words (uniform base; see AttnArgs::keep_bits)
      uint32_t* keep_tile = KEEP ? a.keep_bits : nullptr;
      if (KEEP) {
        const long nqb = (Smax + 31) >> 5;
        keep_tile += ((((long)b * a.nh + head) * nqb + (q0 >> 5)) * nqb << 5) + kc;
      }
      for (int kt = 0; kt < ntiles; ++kt) {
        // ---- S^T tile = K[32 keys] . Q^T, on top of the bias ----
        f32x16 sacc;
        const float* bp = (const float*)(smem + ATT_SBIAS) + kt * 32 + 4 * h2;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 bv = *(const f32x4*)(bp + 8 * g4);
          if (MASK3) {   // per-query bias row of this lane's query; the LDS bias is 0 / -inf (key range)
            const int key0 = kc + kt * 32 + 8 * g4 + 4 * h2;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (key0 + e < S) bv[e] += mrow3[key0 + e] * inv_scale;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) sacc[4 * g4 + e] = bv[e];
        }
        const char* kp = smem + ATT_SK + kt * 4096 + k_row_off;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          const bf16x8 kf = *(const bf16x8*)(kp + (((2 * ds + h2) ^ k_swz) << 4));
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], sacc, 0, 0, 0);
        }
        // ---- online softmax in the exp2 domain on the raw accumulators.  The kernel is VALU-bound
        // (profiles/r01/attention_sq_counters.txt): every instruction taken out of this loop shows. ----
        float tmax = sacc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, sacc[i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // the running maximum moves in the first tile or two and then rarely: the rescale of the 32 output registers (and
        // its exp2) is skipped when no lane's maximum changed (wave-uniform branch; exact: alpha would be 1)
        if (__builtin_amdgcn_ballot_w64(tmax > m_run) != 0) {
          const float m_new = fmaxf(m_run, tmax);
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale2);
          m_run = m_new;
          l_run *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
        const float negm = -m_run * scale2;
        // (scalar arrays from here on: updating elements of the f32x16 accumulator tuple in place made hipcc copy the whole
        // tuple -- 16 v_mov_b64 per tile -- to keep the undropped values for the row sum)
        float pv[16];
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          pv[i] = __builtin_amdgcn_exp2f(fmaf(sacc[i], scale2, negm));
          psum += pv[i];
        }
        l_run += psum;
        // a lane's elements 4 g4 .. 4 g4 + 3 are the four neighbouring keys 8 g4 + 4 h2 .. + 3: ONE hash word each (q * S' and
        // the key offsets are multiples of 4); the words of a tile sit at fixed offsets from one word index, so the hash's
        // first multiply is paid once per tile
        const uint32_t xb = dr.thresh ? vt_hash_pre(dr.seed, (q_elem + (uint32_t)(kc + kt * 32 + 4 * h2)) >> (WIDE ? 1 : 2)) : 0u;

        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          typedef __attribute__((ext_vector_type(8))) short short8v;
          u32x4 pbw;
          float pm[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pm[j] = pv[8 * s2 + j];
          if (dr.thresh) {  // drop probabilities AFTER the row sum was taken (the normaliser uses all of them)
            uint32_t ml[8], mh[8];   // lane masks of the eight compares: low half = this wave's 32 queries against key
                                     // (i&3) + 8(i>>2), high half the same queries against that key + 4
#pragma unroll
            for (int j = 0; j < 8; j += 4) {   // elements i .. i + 3 are four neighbouring keys: one hash word, a byte each
              const int i = 8 * s2 + j;          // key offset of element i: 8 (i >> 2) + 4 h2, word index + 2 (i >> 2)
              const uint32_t hw = vt_hash_fin(xb + (uint32_t)((WIDE ? 4 : 2) * (i >> 2)) * VT_HASH_C1);
              // exact-p mode: the word of keys 0 / 1 of the four, and the next word for keys 2 / 3
              const uint32_t hw1 = WIDE ? vt_hash_fin(xb + (uint32_t)(4 * (i >> 2) + 1) * VT_HASH_C1) : 0u;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t hx = e < 2 ? hw : hw1;
                const bool k = WIDE ? ((e & 1) ? hx : (hx << 16)) >= dr.thresh
                                    : ((hw >> (8 * e)) & 0xffu) >= dr.thresh;   // (an SDWA byte compare: no shift, no mask)
                pm[j + e] = k ? pm[j + e] : 0.f;     // the 1 / (1 - p) factor is uniform: applied once to O below
                if (KEEP) {
                  const uint64_t m = __builtin_amdgcn_ballot_w64(k);
                  ml[j + e] = (uint32_t)m; mh[j + e] = (uint32_t)(m >> 32);
                }
              }
            }
            if (KEEP) {
              // a compare's lane mask IS the word the backward wants; each half goes to the lane of its key: lane 16 s2 + l,
              // l < 16, ends up with the keep word of key 32 kt + 16 s2 + l over this wave's 32 queries.  All sixteen
              // v_writelane_b32 behind ONE wait (a VALU write of an SGPR needs wait states before v_writelane reads it, and
              // hipcc's hazard recognizer does not look inside an asm statement)
              int kw = 0;
              if (s2 == 0) ATT_WRITELANE16(kw, ml, mh, 0, 4, 1, 5, 2, 6, 3, 7, 8, 12, 9, 13, 10, 14, 11, 15);
              else ATT_WRITELANE16(kw, ml, mh, 16, 20, 17, 21, 18, 22, 19, 23, 24, 28, 25, 29, 26, 30, 27, 31);
              if ((lane >> 4) == s2) keep_tile[kt * 32 + lane] = (uint32_t)kw;
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) pbw[j] = pack_bf16x2(pm[2 * j], pm[2 * j + 1]);
          const bf16x8 pb = __builtin_bit_cast(bf16x8, pbw);
          const char* vp = smem + ATT_SV + (kt * 32 + 16 * s2 + v_row) * 128;
          const bf16x8 vf0 = tr_pair(vp + ((v_colb) ^ v_swz));
          const bf16x8 vf1 = tr_pair(vp + ((v_colb + 64) ^ v_swz));
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pb, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pb, o1, 0, 0, 0);
        }
      }
    }
  }

  if (!wave_active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const int q = q0 + r;
  if (q >= S) return;
  float inv = 1.0f / l_tot;
  if (dr.thresh) inv *= dr.scale;   // dropout's 1 / (1 - p)
  if (a.lse && h2 == 0) a.lse[((long)b * a.nh + head) * Smax + q] = (m_run * scale2 + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  if (a.head_scale) inv *= a.head_scale[head];
  bf16_t* op = a.ctx + (row0 + q) * a.ld_ctx + head * 64 + 16 * h2;
  u32x4 w0, w1, w2, w3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w0[i] = pack_bf16x2(o0[2 * i] * inv, o0[2 * i + 1] * inv);
    w1[i] = pack_bf16x2(o0[8 + 2 * i] * inv, o0[8 + 2 * i + 1] * inv);
    w2[i] = pack_bf16x2(o1[2 * i] * inv, o1[2 * i + 1] * inv);
    w3[i] = pack_bf16x2(o1[8 + 2 * i] * inv, o1[8 + 2 * i + 1] * inv);
  }
  ((u32x4*)op)[0] = w0;
  ((u32x4*)op)[1] = w1;
  ((u32x4*)(op + 32))[0] = w2;
  ((u32x4*)(op + 32))[1] = w3;
}

int vt_attention_fwd_dispatch(const void* qkv, long ld_qkv, const float* mask, int mask_additive, const float* head_scale, void* ctx,
                              long ld_ctx, float* lse, int B, int S, int nh, int head_size, hipStream_t stream,
                              const DropCfg* drop = nullptr, const int* seq_start = nullptr, const int* seq_len = nullptr,
                              uint32_t* keep_bits = nullptr, const PrefetchArgs* pf = nullptr) {
  if (!qkv || !ctx) return VT_ERR_NULL;
  if (head_size != 64) return VT_ERR_UNSUPPORTED;
  if (B <= 0 || S <= 0 || nh <= 0 || B > 65535 || nh > 65535) return VT_ERR_BAD_SHAPE;
  if ((ld_qkv % 8) || (ld_ctx % 8) || ld_qkv < 3L * nh * 64 || ld_ctx < (long)nh * 64) return VT_ERR_BAD_ALIGN;
  if (((uintptr_t)qkv | (uintptr_t)ctx) & 15) return VT_ERR_BAD_ALIGN;
  static VtLdsAttrOnce attr, attr_keep, attr_m3, attr_keep_m3, attr_w, attr_keep_w, attr_m3_w, attr_keep_m3_w;
  if (!attr.set((const void*)attention_fwd_d64<false, false>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  if (!attr_keep.set((const void*)attention_fwd_d64<true, false>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  if (!attr_m3.set((const void*)attention_fwd_d64<false, true>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  if (!attr_keep_m3.set((const void*)attention_fwd_d64<true, true>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  if (!attr_w.set((const void*)attention_fwd_d64<false, false, true>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  if (!attr_keep_w.set((const void*)attention_fwd_d64<true, false, true>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  if (!attr_m3_w.set((const void*)attention_fwd_d64<false, true, true>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  if (!attr_keep_m3_w.set((const void*)attention_fwd_d64<true, true, true>, ATT_LDS_BYTES)) return VT_ERR_HIP;
  AttnArgs a;
  a.qkv = (const bf16_t*)qkv; a.mask = mask; a.mask_additive = mask_additive; a.head_scale = head_scale; a.ctx = (bf16_t*)ctx; a.lse = lse;
  a.ld_qkv = ld_qkv; a.ld_ctx = ld_ctx; a.B = B; a.S = S; a.nh = nh;
  a.scale = 1.0f / sqrtf((float)head_size);
  if (drop) a.drop = *drop; else { a.drop.thresh = 0; a.drop.seed = 0; a.drop.scale = 1.0f; }
  if ((seq_start == nullptr) != (seq_len == nullptr)) return VT_ERR_NULL;
  if (seq_start && mask) return VT_ERR_UNSUPPORTED;   // compacted rows carry no masked keys
  a.seq_start = seq_start; a.seq_len = seq_len;
  a.keep_bits = keep_bits;
  dim3 grid((S + 255) / 256, nh, B);
  a.pf.n = 0;
  for (int i = 0; i < 4; ++i) { a.pf.p[i] = nullptr; a.pf.bytes[i] = 0; }
  if (pf && pf->n > 0 && B < 65535 - 8) {   // spare z-slices: about one prefetch workgroup (half a block) per 16 KiB, at most 8 slices
    long blocks16k = 0;
    for (int i = 0; i < pf->n; ++i) blocks16k += (pf->bytes[i] + 16383) >> 14;
    const long per_slice = 2L * grid.x * grid.y;
    long z = (blocks16k + per_slice - 1) / per_slice;
    z = z < 1 ? 1 : (z > 8 ? 8 : z);
    a.pf = *pf;
    grid.z += (unsigned)z;
  }
  const bool keep = keep_bits && a.drop.thresh, m3 = mask && mask_additive == 2;
  const bool wide = vt_attn_wide(a.drop);   // exact-p mode (common.hpp): 16-bit fields
#define ATT_FWD_LAUNCH(K_, M_)                                                                                        \
  do {                                                                                                                \
    if (wide) hipLaunchKernelGGL((attention_fwd_d64<K_, M_, true>), grid, dim3(512), ATT_LDS_BYTES, stream, a);       \
    else hipLaunchKernelGGL((attention_fwd_d64<K_, M_, false>), grid, dim3(512), ATT_LDS_BYTES, stream, a);           \
  } while (0)
  if (keep && m3) ATT_FWD_LAUNCH(true, true);
  else if (keep) ATT_FWD_LAUNCH(true, false);
  else if (m3) ATT_FWD_LAUNCH(false, true);
  else ATT_FWD_LAUNCH(false, false);
#undef ATT_FWD_LAUNCH
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}


// ---- attention probabilities for `output_attentions` (oscar/modeling_bert.py:58-66, 74-79) ------------------------
// probs[b,h,q,k] = exp(q.k / 8 + bias[k] - lse[b,h,q]) [* head_scale[h]] in fp32, from the packed projection output and
// the log-sum-exp the fused forward kernel saved; eval mode (no dropout).  A diagnostic output: one workgroup per
// (query, head, batch), plain VALU dot products, the row written contiguously.
__global__ __launch_bounds__(256) void attention_probs_d64(AttnArgs a, float* __restrict__ probs) {
  __shared__ float qs[64];
  const int q = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int S = a.S, H = a.nh * 64;
  const bf16_t* base = a.qkv + (long)b * S * a.ld_qkv + head * 64;
  if (threadIdx.x < 64) qs[threadIdx.x] = bf16_to_f32(base[(long)q * a.ld_qkv + threadIdx.x]);
  __syncthreads();
  const float lse = a.lse[((long)b * a.nh + head) * S + q];
  const float hs = a.head_scale ? a.head_scale[head] : 1.0f;
  float* out = probs + (((long)b * a.nh + head) * S + q) * S;
  for (int key = threadIdx.x; key < S; key += 256) {
    const u32x4* kp = (const u32x4*)(base + (long)key * a.ld_qkv + H);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 w = kp[c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dot = fmaf(qs[8 * c + 2 * i], bf16lo(w[i]), dot);
        dot = fmaf(qs[8 * c + 2 * i + 1], bf16hi(w[i]), dot);
      }
    }
    float add = 0.f;
    if (a.mask) {
      const float mval = a.mask_additive == 2 ? a.mask[((long)b * S + q) * S + key] : a.mask[(long)b * S + key];
      add = a.mask_additive ? mval : (1.0f - mval) * -10000.0f;
    }
    out[key] = __expf(fmaf(dot, a.scale, add) - lse) * hs;
  }
}

int vt_attention_probs_dispatch(const void* qkv, long ld_qkv, const float* mask, int mask_additive, const float* head_scale,
                                const float* lse, float* probs, int B, int S, int nh, int head_size, hipStream_t stream) {
  if (!qkv || !lse || !probs) return VT_ERR_NULL;
  if (head_size != 64) return VT_ERR_UNSUPPORTED;
  if (B <= 0 || S <= 0 || nh <= 0 || B > 65535 || nh > 65535) return VT_ERR_BAD_SHAPE;
  if ((ld_qkv % 8) || ld_qkv < 3L * nh * 64 || ((uintptr_t)qkv & 15)) return VT_ERR_BAD_ALIGN;
  AttnArgs a;
  a.qkv = (const bf16_t*)qkv; a.mask = mask; a.mask_additive = mask_additive; a.head_scale = head_scale; a.ctx = nullptr;
  a.lse = const_cast<float*>(lse);
  a.ld_qkv = ld_qkv; a.ld_ctx = 0; a.B = B; a.S = S; a.nh = nh; a.seq_start = nullptr; a.seq_len = nullptr; a.keep_bits = nullptr;
  a.scale = 1.0f / sqrtf((float)head_size);
  a.drop.thresh = 0; a.drop.seed = 0; a.drop.scale = 1.0f;
  hipLaunchKernelGGL(attention_probs_d64, dim3(S, nh, B), dim3(256), 0, stream, a, probs);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
