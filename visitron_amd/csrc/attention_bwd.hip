// Backward of the fused self-attention (head size 64): given dCtx, recompute P from Q, K and
// the forward's log-sum-exp and produce dQ | dK | dV in the packed [B*S, 3H] layout.  This is the
// autograd of oscar/modeling_bert.py:47-72 inside `loss.backward()` (tasks/viewpoint_select/
// pretrain.py:191):
//     P  = softmax(QK^T/8 + mask)      dV = P^T dO        dP = dO V^T
//     dS = P * (dP - delta)            dQ = dS K / 8      dK = dS^T Q / 8,   delta = rowsum(dO * O)
// The S x S matrices never exist in HBM.
//
// gfx950 design.  One workgroup = 4 waves = one (batch, head); wave w owns keys 64w..64w+63 for the
// whole kernel: their K and V fragments live in registers (MFMA B operands) and their dK^T / dV^T
// (64 d x 64 keys each) accumulate in 128 registers, so dK and dV need no cross-wave or cross-
// workgroup sum.  The workgroup sweeps 32-query slices (Q and dO slices double-buffered in LDS by
// global_load_lds).  All score-shaped products keep the KEY on the MFMA lane (v_mfma_f32_32x32x16_bf16):
//   S[q][key]  = Q . K^T       dP[q][key] = dO . V^T        (A = LDS rows, B = registers)
// so P and dS tiles are, as they stand in the accumulators, the B operands of
//   dV^T[d][key] += dO^T . P   dK^T[d][key] += Q^T . dS     (A = ds_read_b64_tr_b16 of the slices)
// Only dS crosses LDS, once per slice, as a [4-query group][key] image that every wave then reads
// (transposed) to form dQ^T[d][q] = K^T . dS^T over ALL keys -- wave w produces d rows 16w..16w+15
// with v_mfma_f32_16x16x32_bf16 -- so dQ needs no reduction either, and for S <= 256 everything is
// bitwise reproducible (no atomics).  For S > 256 the keys are split into blocks of 256 with one
// workgroup each (grid.z); dK/dV stay private, and the dQ partials of the key blocks are combined by
// fp32 atomics into a scratch [B*S,H] slab: each slice's dQ^T is transposed through LDS so that a
// wave-instruction adds 256 contiguous bytes of one row (the full-rate atomic shape), then a small
// kernel rounds the slab to bf16.
#include "common.hpp"
#include <type_traits>

#define LOG2E 1.4426950408889634f

// A sequence's row count as the kernels use it: 1 .. S.  The engine always keeps the [CLS] row (seq_len >= 1) and never
// exceeds the padded length; a C-ABI caller that hands 0 or more than S would otherwise turn the unsigned "last row" clamps
// below (S - 1) into ~4 G rows of offset in LDS-DMA loads that are not range-checked.  Out-of-contract lengths therefore
// compute (meaningless but in-bounds) rows instead of faulting; include/visitron_hip.h states the contract.
__device__ __forceinline__ int vt_clamp_len(int len, int S) { return len < 1 ? 1 : (len > S ? S : len); }

struct AttnBwdArgs {
  const bf16_t* qkv;    // [B*S, ld_qkv]
  const bf16_t* dctx;   // [B*S, ld_d]   gradient of the context
  const float* mask;    // [B,S] raw or additive (as forward), or null
  int mask_additive;
  const float* lse;     // [B, nh, S] natural-log LSE from the forward
  const float* delta;   // [B, nh, S] rowsum(dO * O)
  bf16_t* dqkv;         // [B*S, ld_dqkv]  dq | dk | dv
  float* dq32;          // [B*S, nh*64] fp32 accumulation slab (zeroed) when S > 256, else null
  long ld_qkv, ld_d, ld_dqkv;
  int B, S, nh;
  const int* seq_start;   // compacted rows (see attention_fwd.hip): sequence b = rows seq_start[b] .. + seq_len[b];
  const int* seq_len;     // lse / delta keep their [B, nh, S] layout.  Null: sequence b = rows b*S .. b*S + S
  float scale;          // 1 / sqrt(head_size)
  DropCfg drop;         // the forward's attention-probability dropout (same seed / indexing)
  // the forward's keep decisions (AttnArgs::keep_bits in attention_fwd.hip): word [((b*nh + h) * nqb + qb) * kpitch + key],
  // bit j = keep(query 32 qb + j, key).  Null: the 8-wave kernel re-derives them from the hash like the 4-wave kernel does.
  const uint32_t* keep_bits;
  // the forward's context rows [B*S, ld_ctx] for the kernels that form delta = rowsum(dO o O) themselves (8-wave, DELTA)
  const bf16_t* ctx; long ld_ctx; float delta_mul;
};

// LDS map (bytes)
#define AB_K 0                      // [256 keys][128 B], 32-B groups XOR ((row>>1)&1 | ((row>>3)&1)<<1)
#define AB_Q (32768)                // 2 x [32 q][128 B], chunk XOR (row>>1)&7
#define AB_DO (AB_Q + 2 * 4096)     // 2 x [32 q][128 B], same
#define AB_DS (AB_DO + 2 * 4096)    // 2 x [8 q-groups][256 keys][8 B]
#define AB_ROW (AB_DS + 2 * 16384)  // 2 x (lse[32] | delta[32]) fp32
#define AB_DQ (AB_ROW + 2 * 256)     // [32 q][64 d] fp32 staging for the atomic dQ path
#define AB_LDS_BYTES (AB_DQ + 8192)

typedef __attribute__((ext_vector_type(8))) short short8v;

__device__ __forceinline__ bf16x8 tr_pair_b(unsigned addr, int second_off) {
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)addr);
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(addr + second_off));
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s2) {
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = pack_bf16x2(v[8 * s2 + 2 * j], v[8 * s2 + 2 * j + 1]);
  return __builtin_bit_cast(bf16x8, r);
}

// Keep flags of a lane's 16 score elements in the backward tiles, re-derived from the hash (the path without the forward's
// keep words: tests, VT_ATTN_KEEP_BITS=0): element i = query qbase + (i&3) + 8*(i>>2), this lane's key.  The attention
// sites' function (common.hpp, vt_keep_attn): byte key & 3 of the hash word of (query, key >> 2), row pitch Sp = the
// sequence's length rounded up to a multiple of 4.  The word index is linear in the query, so the hash's first multiply is
// taken once and the other queries are reached by adding multiples of (Sp / 4) * C1.
// Exact-p mode (vt_attn_wide: 16-bit fields, two keys per word): the same walk with key >> 1 / Sp >> 1 and field key & 1.
__device__ __forceinline__ void attn_bwd_keep16(const DropCfg& dr, uint32_t qbase, uint32_t key, uint32_t Sp, bool (&keep)[16]) {
  const bool wide = vt_attn_wide(dr);
  const uint32_t lg = wide ? 1u : 2u;
  const uint32_t sh = wide ? 16u * (key & 1u) : 8u * (key & 3u);
  const uint32_t fmask = wide ? 0xffffu : 0xffu, th = wide ? dr.thresh >> 16 : dr.thresh;
  const uint32_t qstep = (Sp >> lg) * VT_HASH_C1;
  const uint32_t x0 = vt_hash_pre(dr.seed, qbase * (Sp >> lg) + (key >> lg));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t h = vt_hash_fin(x0 + (uint32_t)((i & 3) + 8 * (i >> 2)) * qstep);
    keep[i] = ((h >> sh) & fmask) >= th;
  }
}

__global__ __launch_bounds__(256, 1) void attention_bwd_d64(AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h2 = lane >> 5;
  const int b = blockIdx.y, head = blockIdx.x;
  const int kb0 = blockIdx.z * 256;   // first key of this workgroup's key block
  const int Smax = a.S, H = a.nh * 64;
  const int S = a.seq_len ? vt_clamp_len(a.seq_len[b], a.S) : a.S;                       // this sequence's rows
  const long row0 = a.seq_start ? (long)a.seq_start[b] : (long)b * a.S;
  if (kb0 >= S) return;                                               // uniform: a key block past a short sequence
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const bf16_t* base = a.qkv + row0 * a.ld_qkv + head * 64;
  const bf16_t* dobase = a.dctx + row0 * a.ld_d + head * 64;
  const float* lse_p = a.lse + ((long)b * a.nh + head) * Smax;
  const float* del_p = a.delta + ((long)b * a.nh + head) * Smax;
  const int nslices = (S + 31) >> 5;
  DropCfg dr = a.drop;
  dr.seed = vt_hash32(a.drop.seed, (uint32_t)(b * a.nh + head));
  const float ds_scale = a.scale * dr.scale;   // 1 / sqrt(d) times dropout's 1 / (1-p) (1 when off)
  const int kcount = (S - kb0) < 256 ? (S - kb0) : 256;
  const int nkt = (kcount + 31) >> 5;  // 32-key steps of this key block for dQ

  // ---- K tile (all keys) -> LDS for the dQ product ----
  for (int j = wave; j < nkt * 4; j += 4) {  // 8-row pieces
    const int row = 8 * j + (lane >> 3);
    int kr = (kb0 + row) < S ? (kb0 + row) : S - 1;
    const int cs = lane & 7;
    const int sw = (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;  // 32-B group XOR, in 16-B chunk units
    glds16(base + (long)kr * a.ld_qkv + H + ((cs ^ sw) << 3), smem + AB_K + j * 1024);
  }
  // ---- slice loader: Q and dO rows q0..q0+31 (one 1-KiB piece = 8 rows; waves 0..3 take rows 8w..8w+7) ----
  // The row constants (lse | delta) are fetched BEFORE the DMA is issued (vmcnt retires in order: a
  // later register load would make the wave wait for the DMA too) and written to LDS by store_rows().
  auto load_rows = [&](int sl) -> float {
    float v = 0.f;
    if (tid < 64) {
      const int qq = sl * 32 + (tid & 31);
      if (tid < 32) v = qq < S ? lse_p[qq] : INFINITY;  // +inf => P = 0 for padded queries
      else v = qq < S ? del_p[qq] : 0.f;
    }
    return v;
  };
  auto store_rows = [&](float v, int buf) {
    if (tid < 64) ((float*)(smem + AB_ROW + buf * 256))[tid] = v;
  };
  auto load_slice = [&](int sl, int buf) {
    const int row = 8 * wave + (lane >> 3);
    int q = sl * 32 + row;
    q = q < S ? q : S - 1;
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    glds16(base + (long)q * a.ld_qkv + (c << 3), smem + AB_Q + buf * 4096 + wave * 1024);
    glds16(dobase + (long)q * a.ld_d + (c << 3), smem + AB_DO + buf * 4096 + wave * 1024);
  };
  store_rows(load_rows(0), 0);
  load_slice(0, 0);

  // ---- this wave's K and V fragments (B operands; lane = key) and its key biases ----
  bf16x8 kf[2][4], vf[2][4];
  float kbias[2];
  const float* mq3[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = kb0 + 64 * wave + 32 * kt + r;
    const int kr = key < S ? key : S - 1;
    const bf16_t* kp = base + (long)kr * a.ld_qkv + H + 8 * h2;
    const bf16_t* vp = base + (long)kr * a.ld_qkv + 2 * H + 8 * h2;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      kf[kt][ds] = *(const bf16x8*)(kp + 16 * ds);
      vf[kt][ds] = *(const bf16x8*)(vp + 16 * ds);
    }
    float add = -INFINITY;
    if (key < S) {
      add = 0.f;
      if (a.mask && a.mask_additive != 2) {
        const float mval = a.mask[(long)b * Smax + key];
        add = a.mask_additive ? mval : (1.0f - mval) * -10000.0f;
      }
    }
    kbias[kt] = add;
    // mask_additive == 2: an additive bias per (query, key) [B, S, S] (the reference's 3-D attention masks,
    // encoder.py:228-229): this lane's key column; the query row is added per element below
    mq3[kt] = (a.mask && a.mask_additive == 2) ? a.mask + (long)b * Smax * Smax + kr : nullptr;
  }

  f32x16 dv[2][2], dk[2][2];  // [d-tile][key-tile]: lane = key, reg <-> d = 32dt + 16h2 + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dv[i][j][e] = 0.f; dk[i][j][e] = 0.f; }

  // lane-constant addressing
  const int a_row = r * 128;                                  // Q / dO row read (A operand of S, dP)
  const int a_swz = (r >> 1) & 7;
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, dhalf = (lane >> 4) & 1, g = lane >> 4;
  // transposed slice reads (A operand of dV^T / dK^T): rows q = 16*s2 + 4*h2 + q4 (+8), cols d-perm
  const int t_row = 4 * h2 + q4;
  const int t_col = 2 * (16 * (p4 & 1) + 8 * dhalf + 4 * (p4 >> 1));  // + 64*dt, within the 128-B row
  // slices use the 16-B chunk swizzle (row>>1)&7: as a byte XOR on bits 4..6
  const int t_sw0 = (((t_row) >> 1) & 7) << 4;          // rows 4h2+q4        (s2 = 0)
  const int t_sw1 = (((t_row + 8) >> 1) & 7) << 4;      // rows +8
  const int t_sw2 = (((t_row + 16) >> 1) & 7) << 4;     // s2 = 1
  const int t_sw3 = (((t_row + 24) >> 1) & 7) << 4;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int sl = 0; sl < nslices; ++sl) {
    const int buf = sl & 1;

    const char* sQ = smem + AB_Q + buf * 4096;
    const char* sDO = smem + AB_DO + buf * 4096;
    const float* rowv = (const float*)(smem + AB_ROW + buf * 256);
    char* sDS = smem + AB_DS + buf * 16384;

    // A fragments of this slice: Q and dO rows (lane (r, h2): row r, d = 16ds + 8h2 ..)
    bf16x8 qa[4], da[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      qa[ds] = *(const bf16x8*)(sQ + a_row + (((2 * ds + h2) ^ a_swz) << 4));
      da[ds] = *(const bf16x8*)(sDO + a_row + (((2 * ds + h2) ^ a_swz) << 4));
    }
    // per-register row constants: q = (reg&3) + 8(reg>>2) + 4h2
    f32x4 lse4[4], del4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      lse4[g4] = *(const f32x4*)(rowv + 8 * g4 + 4 * h2);
      del4[g4] = *(const f32x4*)(rowv + 32 + 8 * g4 + 4 * h2);
    }
    // transposed A operands of the slice: [s2][dt] for dO^T (dV) and Q^T (dK)
    bf16x8 doT[2][2], qT[2][2];
    {
      const unsigned bq = lds0 + AB_Q + buf * 4096 + t_row * 128;
      const unsigned bd = lds0 + AB_DO + buf * 4096 + t_row * 128;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int cb = t_col + 64 * dt;
        // rows +8 => +1024 bytes, but the swizzle differs between the two blocks: two explicit reads
        short4v q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + (cb ^ t_sw0)));
        short4v q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + 1024 + (cb ^ t_sw1)));
        short4v q2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + 2048 + (cb ^ t_sw2)));
        short4v q3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + 3072 + (cb ^ t_sw3)));
        qT[0][dt] = __builtin_bit_cast(bf16x8, (short8v){q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]});
        qT[1][dt] = __builtin_bit_cast(bf16x8, (short8v){q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]});
        short4v d0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + (cb ^ t_sw0)));
        short4v d1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + 1024 + (cb ^ t_sw1)));
        short4v d2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + 2048 + (cb ^ t_sw2)));
        short4v d3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + 3072 + (cb ^ t_sw3)));
        doT[0][dt] = __builtin_bit_cast(bf16x8, (short8v){d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]});
        doT[1][dt] = __builtin_bit_cast(bf16x8, (short8v){d2[0], d2[1], d2[2], d2[3], d3[0], d3[1], d3[2], d3[3]});
      }
    }

    // prefetch the next slice now that this slice's operands are in registers (issuing the DMA before
    // those LDS reads would make hipcc wait for it first); buffer buf^1 was last read in slice sl-1,
    // which every wave left through the barrier below
    float next_rowv = 0.f;
    if (sl + 1 < nslices) {
      next_rowv = load_rows(sl + 1);
      load_slice(sl + 1, buf ^ 1);
    }

#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      // S = Q K^T and dP = dO V^T for keys 64w + 32kt .. +31 (lane = key)
      f32x16 sacc, dpacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; dpacc[e] = 0.f; }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ds], kf[kt][ds], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ds], vf[kt][ds], dpacc, 0, 0, 0);
      }
      // P = exp(s/8 + mask - lse);  dS = P (dP - delta) / 8
      f32x16 pacc;
      bool keep[16];
      if (dr.thresh) attn_bwd_keep16(dr, (uint32_t)(sl * 32 + 4 * h2), (uint32_t)(kb0 + 64 * wave + 32 * kt + r), (uint32_t)(S + 3) & ~3u, keep);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g4 + e;
          float s = fmaf(sacc[i], a.scale, kbias[kt]);
          if (mq3[kt]) {   // per-query bias row of element i's query (clamped: rows past S carry dO = 0)
            const int qi = sl * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
            s += mq3[kt][(long)(qi < S ? qi : S - 1) * Smax];
          }
          const float p = __builtin_amdgcn_exp2f((s - lse4[g4][e]) * LOG2E);
          float dpv = dpacc[i], pv = p;
          if (dr.thresh) {  // O = (P * mask / (1-p)) V: dP and the P that feeds dV carry the mask, dS keeps P
            dpv = keep[i] ? dpv : 0.f;
            pv = keep[i] ? p : 0.f;
          }
          // dropout's uniform 1 / (1-p) is not applied per element: dV takes it once at the end, and in
          // dS = P (mask dP / (1-p) - delta) / 8 it moves outside the bracket: ds_scale = 1 / (8 (1-p)), the row constant arrives
          // as delta (1-p) (attn_delta_rows)
          pacc[i] = pv;
          sacc[i] = p * (dpv - del4[g4][e]) * ds_scale;  // dS' (scales folded in)
        }
      // dV^T += dO^T P ; dK^T += Q^T dS'   (k = q, 2 steps of 16)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack8(pacc, s2);
        const bf16x8 sb = pack8(sacc, s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[s2][dt], pb, dv[dt][kt], 0, 0, 0);
          dk[dt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s2][dt], sb, dk[dt][kt], 0, 0, 0);
        }
      }
      // dS' -> LDS image [q-group G][key][4 q] (8 B per (G,key)), key index XOR-swizzled by G
      {
        const int key = 64 * wave + 32 * kt + r;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int G = 2 * g4 + h2;
          const int kx = key ^ (((G >> 1) & 1) << 2) ^ ((G & 1) << 4);
          u32x2 w;
          w[0] = pack_bf16x2(sacc[4 * g4 + 0], sacc[4 * g4 + 1]);
          w[1] = pack_bf16x2(sacc[4 * g4 + 2], sacc[4 * g4 + 3]);
          *(u32x2*)(sDS + (G * 256 + kx) * 8) = w;
        }
      }
    }

    if (sl + 1 < nslices) store_rows(next_rowv, buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next slice's DMA (issued above) has landed
    __syncthreads();                                   // ... and every wave's dS' is in the image

    // ---- dQ^T[d][q] = sum_key K^T[d][key] dS'^T[key][q]; wave w: d = 16w .. 16w+15, both 16-q tiles ----
    f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
    {
      const unsigned kbase = lds0 + AB_K;
      const unsigned dsb = lds0 + AB_DS + buf * 16384;
      for (int ks = 0; ks < nkt; ++ks) {
        const int krow = 32 * ks + 8 * g + q4;  // first block row; second block = +4
        const int ksw = ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 5);
        const int ksw2 = (((((krow + 4) >> 1) & 1) | ((((krow + 4) >> 3) & 1) << 1)) << 5);
        const int kcol = 2 * (16 * wave) + 8 * p4;
        short4v k0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(kbase + krow * 128 + (kcol ^ ksw)));
        short4v k1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(kbase + (krow + 4) * 128 + (kcol ^ ksw2)));
        const bf16x8 ka = __builtin_bit_cast(bf16x8, (short8v){k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]});
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int G = 4 * qt + p4;
          const int gsw = (((G >> 1) & 1) << 2) ^ ((G & 1) << 4);
          short4v s0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(dsb + (G * 256 + (krow ^ gsw)) * 8));
          short4v s1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(dsb + (G * 256 + ((krow + 4) ^ gsw)) * 8));
          const bf16x8 sbq = __builtin_bit_cast(bf16x8, (short8v){s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]});
          if (qt == 0) dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, sbq, dq0, 0, 0, 0);
          else dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, sbq, dq1, 0, 0, 0);
        }
      }
    }
    // store dQ: lane (q_local = lane&15, g): d = 16w + 4g .. +3
    if (a.dq32 == nullptr) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int q = sl * 32 + 16 * qt + (lane & 15);
        if (q < S) {
          const f32x4 v = qt == 0 ? dq0 : dq1;
          u32x2 w;
          w[0] = pack_bf16x2(v[0], v[1]);
          w[1] = pack_bf16x2(v[2], v[3]);
          *(u32x2*)(a.dqkv + (row0 + q) * a.ld_dqkv + head * 64 + 16 * wave + 4 * g) = w;
        }
      }
    } else {
      // several key blocks: transpose the slice's dQ^T through LDS, then add whole 256-byte row
      // segments atomically (lane = d) -- the atomic shape that runs at the full chip-wide rate
      float* st = (float*)(smem + AB_DQ);
      *(f32x4*)(st + (lane & 15) * 64 + 16 * wave + 4 * g) = dq0;
      *(f32x4*)(st + (16 + (lane & 15)) * 64 + 16 * wave + 4 * g) = dq1;
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int ql = 8 * wave + rr;
        const int q = sl * 32 + ql;
        if (q < S) atomicAdd(a.dq32 + (row0 + q) * H + head * 64 + lane, st[ql * 64 + lane]);
      }
      // the staging tile is rewritten only after the next slice's barrier, which every wave reaches
      // after these reads
    }
    // no barrier needed here: the next slice writes the OTHER dS image / reads the other Q,dO buffers,
    // and the barrier inside the next slice orders this slice's dQ reads before the image is reused.
  }

  // ---- dK, dV of this wave's keys: lane = key, reg <-> d = 32dt + 16h2 + reg ----
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = kb0 + 64 * wave + 32 * kt + r;
    if (key >= S) continue;
    bf16_t* orow = a.dqkv + (row0 + key) * a.ld_dqkv + head * 64 + 16 * h2;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      u32x4 k0, k1, v0, v1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        k0[i] = pack_bf16x2(dk[dt][kt][2 * i], dk[dt][kt][2 * i + 1]);
        k1[i] = pack_bf16x2(dk[dt][kt][8 + 2 * i], dk[dt][kt][8 + 2 * i + 1]);
        v0[i] = pack_bf16x2(dv[dt][kt][2 * i] * dr.scale, dv[dt][kt][2 * i + 1] * dr.scale);
        v1[i] = pack_bf16x2(dv[dt][kt][8 + 2 * i] * dr.scale, dv[dt][kt][8 + 2 * i + 1] * dr.scale);
      }
      ((u32x4*)(orow + H + 32 * dt))[0] = k0;
      ((u32x4*)(orow + H + 32 * dt))[1] = k1;
      ((u32x4*)(orow + 2 * H + 32 * dt))[0] = v0;
      ((u32x4*)(orow + 2 * H + 32 * dt))[1] = v1;
    }
  }
}

// 8-wave form: wave w owns keys 32w..32w+31 (one key tile), <= 256 VGPRs, two waves per SIMD.
// BITS: the dropout keep flags come from the words the forward wrote (one 4-byte load per lane and slice, one bit-field
// extract and two ands per element) instead of the hash (about ten vector instructions per element, a third of the loop)
// DELTA: the row constant delta = rowsum(dO o O) (x (1-p) under dropout) is formed here, a slice ahead, from 8 bytes of dO
// and of O per thread (16 threads per query), instead of by a separate pass over both tensors (attn_delta_rows)
template <bool BITS, bool DELTA>
__global__ __launch_bounds__(512, 2) void attention_bwd_d64_w8(AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h2 = lane >> 5;
  const int b = blockIdx.y, head = blockIdx.x;
  const int kb0 = blockIdx.z * 256;   // first key of this workgroup's key block
  const int Smax = a.S, H = a.nh * 64;
  const int S = a.seq_len ? vt_clamp_len(a.seq_len[b], a.S) : a.S;                       // this sequence's rows
  const long row0 = a.seq_start ? (long)a.seq_start[b] : (long)b * a.S;
  if (kb0 >= S) return;                                               // uniform: a key block past a short sequence
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const bf16_t* base = a.qkv + row0 * a.ld_qkv + head * 64;
  const bf16_t* dobase = a.dctx + row0 * a.ld_d + head * 64;
  const float* lse_p = a.lse + ((long)b * a.nh + head) * Smax;
  const float* del_p = a.delta + ((long)b * a.nh + head) * Smax;
  const int nslices = (S + 31) >> 5;
  DropCfg dr = a.drop;
  dr.seed = vt_hash32(a.drop.seed, (uint32_t)(b * a.nh + head));
  const float ds_scale = a.scale * dr.scale;   // 1 / sqrt(d) times dropout's 1 / (1-p) (1 when off): applied to dK once at the
                                               // end and to each slice's four dQ values, not to every dS element
  const int kcount = (S - kb0) < 256 ? (S - kb0) : 256;
  const int nkt = (kcount + 31) >> 5;  // 32-key steps of this key block for dQ

  // ---- K tile (all keys) -> LDS for the dQ product ----
  for (int j = wave; j < nkt * 4; j += 8) {  // 8-row pieces
    const int row = 8 * j + (lane >> 3);
    int kr = (kb0 + row) < S ? (kb0 + row) : S - 1;
    const int cs = lane & 7;
    const int sw = (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;  // 32-B group XOR, in 16-B chunk units
    glds16(base + (long)kr * a.ld_qkv + H + ((cs ^ sw) << 3), smem + AB_K + j * 1024);
  }
  // ---- slice loader: Q and dO rows q0..q0+31 (one 1-KiB piece = 8 rows; waves 0..3 take rows 8w..8w+7) ----
  // The row constants (lse | delta) are fetched BEFORE the DMA is issued (vmcnt retires in order: a
  // later register load would make the wave wait for the DMA too) and written to LDS by store_rows().
  auto load_rows = [&](int sl) -> float {
    float v = 0.f;
    if (tid < 64) {
      const int qq = sl * 32 + (tid & 31);
      if (tid < 32) v = qq < S ? lse_p[qq] : INFINITY;  // +inf => P = 0 for padded queries
      else if (!DELTA) v = qq < S ? del_p[qq] : 0.f;
    }
    return v;
  };
  auto store_rows = [&](float v, int buf) {
    if (tid < (DELTA ? 32 : 64)) ((float*)(smem + AB_ROW + buf * 256))[tid] = v;
  };
  // DELTA: thread t = (query t >> 4 of the slice, 4 head columns 4 (t & 15)): its 8 bytes of dO and of O
  const bf16_t* obase = DELTA ? a.ctx + row0 * a.ld_ctx + head * 64 + 4 * (tid & 15) : nullptr;
  auto load_do_o = [&](int sl, u32x2& xd, u32x2& xo) {
    int q = sl * 32 + (tid >> 4);
    q = q < S ? q : S - 1;
    xd = *(const u32x2*)(dobase + (long)q * a.ld_d + 4 * (tid & 15));
    xo = *(const u32x2*)(obase + (long)q * a.ld_ctx);
  };
  auto store_delta = [&](u32x2 xd, u32x2 xo, int buf) {
    float v = bf16lo(xd[0]) * bf16lo(xo[0]) + bf16hi(xd[0]) * bf16hi(xo[0]) + bf16lo(xd[1]) * bf16lo(xo[1]) + bf16hi(xd[1]) * bf16hi(xo[1]);
    // sum over the 16 lanes of a query (one DPP row): xor 1, xor 2, mirror within 8, mirror within 16
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
    if ((tid & 15) == 0) ((float*)(smem + AB_ROW + buf * 256))[32 + (tid >> 4)] = v * a.delta_mul;
  };
  auto load_slice = [&](int sl, int buf) {
    const int pw = wave & 3;
    const int row = 8 * pw + (lane >> 3);
    int q = sl * 32 + row;
    q = q < S ? q : S - 1;
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    if (wave < 4) glds16(base + (long)q * a.ld_qkv + (c << 3), smem + AB_Q + buf * 4096 + pw * 1024);
    else glds16(dobase + (long)q * a.ld_d + (c << 3), smem + AB_DO + buf * 4096 + pw * 1024);
  };
  // the keep words of this lane's key: one per 32-query slice, fetched a slice ahead like the row constants
  const uint32_t* keep_p = nullptr;
  if (BITS) {
    const long nqb = (Smax + 31) >> 5, kpitch = nqb << 5;
    const long key = kb0 + 32 * wave + r;
    keep_p = a.keep_bits + ((long)b * a.nh + head) * nqb * kpitch + (key < kpitch ? key : kpitch - 1);
  }
  const long keep_step = (long)((Smax + 31) >> 5) << 5;
  uint32_t kw_next = BITS ? keep_p[0] : 0u;
  u32x2 nd = {0u, 0u}, no = {0u, 0u};
  if (DELTA) { load_do_o(0, nd, no); store_delta(nd, no, 0); }
  store_rows(load_rows(0), 0);
  load_slice(0, 0);

  // ---- this wave's K and V fragments (B operands; lane = key) and its key biases ----
  bf16x8 kf[1][4], vf[1][4];
  float kbias[1];
  const float* mq3[1];
#pragma unroll
  for (int kt = 0; kt < 1; ++kt) {
    const int key = kb0 + 32 * wave + r;
    const int kr = key < S ? key : S - 1;
    const bf16_t* kp = base + (long)kr * a.ld_qkv + H + 8 * h2;
    const bf16_t* vp = base + (long)kr * a.ld_qkv + 2 * H + 8 * h2;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      kf[kt][ds] = *(const bf16x8*)(kp + 16 * ds);
      vf[kt][ds] = *(const bf16x8*)(vp + 16 * ds);
    }
    float add = -INFINITY;
    if (key < S) {
      add = 0.f;
      if (a.mask && a.mask_additive != 2) {
        const float mval = a.mask[(long)b * Smax + key];
        add = a.mask_additive ? mval : (1.0f - mval) * -10000.0f;
      }
    }
    kbias[kt] = add;
    // mask_additive == 2: an additive bias per (query, key) [B, S, S] (the reference's 3-D attention masks,
    // encoder.py:228-229): this lane's key column; the query row is added per element below
    mq3[kt] = (a.mask && a.mask_additive == 2) ? a.mask + (long)b * Smax * Smax + kr : nullptr;
  }

  f32x16 dv[2][1], dk[2][1];  // [d-tile][key-tile]: lane = key, reg <-> d = 32dt + 16h2 + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 1; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dv[i][j][e] = 0.f; dk[i][j][e] = 0.f; }

  // lane-constant addressing
  const int a_row = r * 128;                                  // Q / dO row read (A operand of S, dP)
  const int a_swz = (r >> 1) & 7;
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, dhalf = (lane >> 4) & 1, g = lane >> 4;
  // transposed slice reads (A operand of dV^T / dK^T): rows q = 16*s2 + 4*h2 + q4 (+8), cols d-perm
  const int t_row = 4 * h2 + q4;
  const int t_col = 2 * (16 * (p4 & 1) + 8 * dhalf + 4 * (p4 >> 1));  // + 64*dt, within the 128-B row
  // slices use the 16-B chunk swizzle (row>>1)&7: as a byte XOR on bits 4..6
  const int t_sw0 = (((t_row) >> 1) & 7) << 4;          // rows 4h2+q4        (s2 = 0)
  const int t_sw1 = (((t_row + 8) >> 1) & 7) << 4;      // rows +8
  const int t_sw2 = (((t_row + 16) >> 1) & 7) << 4;     // s2 = 1
  const int t_sw3 = (((t_row + 24) >> 1) & 7) << 4;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int sl = 0; sl < nslices; ++sl) {
    const int buf = sl & 1;

    const char* sQ = smem + AB_Q + buf * 4096;
    const char* sDO = smem + AB_DO + buf * 4096;
    const float* rowv = (const float*)(smem + AB_ROW + buf * 256);
    char* sDS = smem + AB_DS + buf * 16384;

    // A fragments of this slice: Q and dO rows (lane (r, h2): row r, d = 16ds + 8h2 ..)
    bf16x8 qa[4], da[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      qa[ds] = *(const bf16x8*)(sQ + a_row + (((2 * ds + h2) ^ a_swz) << 4));
      da[ds] = *(const bf16x8*)(sDO + a_row + (((2 * ds + h2) ^ a_swz) << 4));
    }
    // per-register row constants (q = (reg&3) + 8(reg>>2) + 4h2) are read from LDS where they are used:
    // holding them across the MFMA chains costs 32 registers this kernel does not have
    // transposed A operands of the slice: [s2][dt] for dO^T (dV) and Q^T (dK)
    bf16x8 doT[2][2], qT[2][2];
    {
      const unsigned bq = lds0 + AB_Q + buf * 4096 + t_row * 128;
      const unsigned bd = lds0 + AB_DO + buf * 4096 + t_row * 128;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int cb = t_col + 64 * dt;
        // rows +8 => +1024 bytes, but the swizzle differs between the two blocks: two explicit reads
        short4v q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + (cb ^ t_sw0)));
        short4v q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + 1024 + (cb ^ t_sw1)));
        short4v q2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + 2048 + (cb ^ t_sw2)));
        short4v q3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bq + 3072 + (cb ^ t_sw3)));
        qT[0][dt] = __builtin_bit_cast(bf16x8, (short8v){q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]});
        qT[1][dt] = __builtin_bit_cast(bf16x8, (short8v){q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]});
        short4v d0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + (cb ^ t_sw0)));
        short4v d1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + 1024 + (cb ^ t_sw1)));
        short4v d2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + 2048 + (cb ^ t_sw2)));
        short4v d3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(bd + 3072 + (cb ^ t_sw3)));
        doT[0][dt] = __builtin_bit_cast(bf16x8, (short8v){d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]});
        doT[1][dt] = __builtin_bit_cast(bf16x8, (short8v){d2[0], d2[1], d2[2], d2[3], d3[0], d3[1], d3[2], d3[3]});
      }
    }

    // prefetch the next slice now that this slice's operands are in registers (issuing the DMA before
    // those LDS reads would make hipcc wait for it first); buffer buf^1 was last read in slice sl-1,
    // which every wave left through the barrier below
    float next_rowv = 0.f;
    const uint32_t kw_cur = kw_next >> (4 * h2);   // bit (i&3) + 8(i>>2) = element i's query
    if (sl + 1 < nslices) {
      next_rowv = load_rows(sl + 1);
      if (DELTA) load_do_o(sl + 1, nd, no);
      if (BITS) kw_next = keep_p[(long)(sl + 1) * keep_step];
      load_slice(sl + 1, buf ^ 1);
    }

#pragma unroll
    for (int kt = 0; kt < 1; ++kt) {
      // S = Q K^T and dP = dO V^T for keys 32w .. 32w+31 (lane = key)
      f32x16 sacc, dpacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; dpacc[e] = 0.f; }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ds], kf[kt][ds], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ds], vf[kt][ds], dpacc, 0, 0, 0);
      }
      // P = exp(s/8 + mask - lse);  dS = P (dP - delta) / 8
      f32x16 pacc;
      // hash path (no keep words): the flags of a group of four queries are derived right before their use -- sixteen flags
      // held across the tile cost three spilled registers at this kernel's 128-register budget
      uint32_t hx0 = 0, hstep = 0, hsh = 0, hmask = 0xffu, hth = dr.thresh;
      if (!BITS && dr.thresh) {
        const bool wide = vt_attn_wide(dr);   // exact-p mode: 16-bit fields, two keys per word
        const uint32_t lg = wide ? 1u : 2u;
        const uint32_t Sp4 = ((uint32_t)(S + 3) & ~3u) >> lg, key = (uint32_t)(kb0 + 32 * wave + r);
        hsh = wide ? 16u * (key & 1u) : 8u * (key & 3u);
        hmask = wide ? 0xffffu : 0xffu;
        hth = wide ? dr.thresh >> 16 : dr.thresh;
        hstep = Sp4 * VT_HASH_C1;
        hx0 = vt_hash_pre(dr.seed, (uint32_t)(sl * 32 + 4 * h2) * Sp4 + (key >> lg));
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 lse4_g = *(const f32x4*)(rowv + 8 * g4 + 4 * h2);
        const f32x4 del4_g = *(const f32x4*)(rowv + 32 + 8 * g4 + 4 * h2);
        bool keep[16];
        if (!BITS && dr.thresh) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            keep[4 * g4 + e] = ((vt_hash_fin(hx0 + (uint32_t)(e + 8 * g4) * hstep) >> hsh) & hmask) >= hth;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g4 + e;
          float s = fmaf(sacc[i], a.scale, kbias[kt]);
          if (mq3[kt]) {   // per-query bias row of element i's query (clamped: rows past S carry dO = 0)
            const int qi = sl * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
            s += mq3[kt][(long)(qi < S ? qi : S - 1) * Smax];
          }
          const float p = __builtin_amdgcn_exp2f((s - lse4_g[e]) * LOG2E);
          float dpv = dpacc[i], pv = p;
          if (BITS) {   // (launched only with dropout on)
            const int km = __builtin_amdgcn_sbfe((int)kw_cur, (i & 3) + 8 * (i >> 2), 1);   // 0 / -1
            dpv = __int_as_float(__float_as_int(dpv) & km);
            pv = __int_as_float(__float_as_int(p) & km);
          } else if (dr.thresh) {  // O = (P * mask / (1-p)) V: dP and the P that feeds dV carry the mask, dS keeps P
            dpv = keep[i] ? dpv : 0.f;
            pv = keep[i] ? p : 0.f;
          }
          // dropout's uniform 1 / (1-p) is not applied per element: dV takes it once at the end, and in
          // dS = P (mask dP / (1-p) - delta) / 8 it moves outside the bracket: ds_scale = 1 / (8 (1-p)), the row constant arrives
          // as delta (1-p) (attn_delta_rows); ds_scale itself multiplies the dK accumulators at the end and dQ per slice
          // (348 -> 337 us per launch at B = 256; measured on the same box and NOT taken: log2(e) folded into scale / bias / lse
          // -- 16 multiplies fewer, 364-374 us, as in round 2 --, dS as fma(P o M, dP, -P delta) 340-349 us, the previous
          // slice's dQ product issued under this slice's element-wise work 384 us at 256 registers)
          pacc[i] = pv;
          sacc[i] = p * (dpv - del4_g[e]);   // dS' up to ds_scale
        }
      }
      // dV^T += dO^T P ; dK^T += Q^T dS'   (k = q, 2 steps of 16)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack8(pacc, s2);
        const bf16x8 sb = pack8(sacc, s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[s2][dt], pb, dv[dt][kt], 0, 0, 0);
          dk[dt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s2][dt], sb, dk[dt][kt], 0, 0, 0);
        }
      }
      // dS' -> LDS image [q-group G][key][4 q] (8 B per (G,key)), key index XOR-swizzled by G
      {
        const int key = 32 * wave + r;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int G = 2 * g4 + h2;
          const int kx = key ^ (((G >> 1) & 1) << 2) ^ ((G & 1) << 4);
          u32x2 w;
          w[0] = pack_bf16x2(sacc[4 * g4 + 0], sacc[4 * g4 + 1]);
          w[1] = pack_bf16x2(sacc[4 * g4 + 2], sacc[4 * g4 + 3]);
          *(u32x2*)(sDS + (G * 256 + kx) * 8) = w;
        }
      }
    }

    if (sl + 1 < nslices) {
      store_rows(next_rowv, buf ^ 1);
      if (DELTA) store_delta(nd, no, buf ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next slice's DMA (issued above) has landed
    __syncthreads();                                   // ... and every wave's dS' is in the image

    // ---- dQ^T[d][q] = sum_key K^T[d][key] dS'^T[key][q]; wave w: d = 16w .. 16w+15, both 16-q tiles ----
    f32x4 dq0 = {0.f, 0.f, 0.f, 0.f};
    const int dqd = wave >> 1, dqq = wave & 1;   // this wave's 16(d) x 16(q) tile of dQ^T
    {
      const unsigned kbase = lds0 + AB_K;
      const unsigned dsb = lds0 + AB_DS + buf * 16384;
      for (int ks = 0; ks < nkt; ++ks) {
        const int krow = 32 * ks + 8 * g + q4;  // first block row; second block = +4
        const int ksw = ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 5);
        const int ksw2 = (((((krow + 4) >> 1) & 1) | ((((krow + 4) >> 3) & 1) << 1)) << 5);
        const int kcol = 2 * (16 * dqd) + 8 * p4;
        short4v k0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(kbase + krow * 128 + (kcol ^ ksw)));
        short4v k1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(kbase + (krow + 4) * 128 + (kcol ^ ksw2)));
        const bf16x8 ka = __builtin_bit_cast(bf16x8, (short8v){k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]});
        {
          const int qt = dqq;
          const int G = 4 * qt + p4;
          const int gsw = (((G >> 1) & 1) << 2) ^ ((G & 1) << 4);
          short4v s0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(dsb + (G * 256 + (krow ^ gsw)) * 8));
          short4v s1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(uintptr_t)(dsb + (G * 256 + ((krow + 4) ^ gsw)) * 8));
          const bf16x8 sbq = __builtin_bit_cast(bf16x8, (short8v){s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]});
          dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, sbq, dq0, 0, 0, 0);
        }
      }
    }
    // store dQ: lane (q_local = lane&15, g): d = 16w + 4g .. +3
    dq0 *= ds_scale;
    if (a.dq32 == nullptr) {
      const int q = sl * 32 + 16 * dqq + (lane & 15);
      if (q < S) {
        u32x2 w;
        w[0] = pack_bf16x2(dq0[0], dq0[1]);
        w[1] = pack_bf16x2(dq0[2], dq0[3]);
        *(u32x2*)(a.dqkv + (row0 + q) * a.ld_dqkv + head * 64 + 16 * dqd + 4 * g) = w;
      }
    } else {
      // several key blocks: transpose the slice's dQ^T through LDS, then add whole 256-byte row
      // segments atomically (lane = d) -- the atomic shape that runs at the full chip-wide rate
      float* st = (float*)(smem + AB_DQ);
      *(f32x4*)(st + (16 * dqq + (lane & 15)) * 64 + 16 * dqd + 4 * g) = dq0;
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int ql = 4 * wave + rr;
        const int q = sl * 32 + ql;
        if (q < S) atomicAdd(a.dq32 + (row0 + q) * H + head * 64 + lane, st[ql * 64 + lane]);
      }
      // the staging tile is rewritten only after the next slice's barrier, which every wave reaches
      // after these reads
    }
    // no barrier needed here: the next slice writes the OTHER dS image / reads the other Q,dO buffers,
    // and the barrier inside the next slice orders this slice's dQ reads before the image is reused.
  }

  // ---- dK, dV of this wave's keys: lane = key, reg <-> d = 32dt + 16h2 + reg ----
#pragma unroll
  for (int kt = 0; kt < 1; ++kt) {
    const int key = kb0 + 32 * wave + r;
    if (key >= S) continue;
    bf16_t* orow = a.dqkv + (row0 + key) * a.ld_dqkv + head * 64 + 16 * h2;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      u32x4 k0, k1, v0, v1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        k0[i] = pack_bf16x2(dk[dt][kt][2 * i] * ds_scale, dk[dt][kt][2 * i + 1] * ds_scale);
        k1[i] = pack_bf16x2(dk[dt][kt][8 + 2 * i] * ds_scale, dk[dt][kt][8 + 2 * i + 1] * ds_scale);
        v0[i] = pack_bf16x2(dv[dt][kt][2 * i] * dr.scale, dv[dt][kt][2 * i + 1] * dr.scale);
        v1[i] = pack_bf16x2(dv[dt][kt][8 + 2 * i] * dr.scale, dv[dt][kt][8 + 2 * i + 1] * dr.scale);
      }
      ((u32x4*)(orow + H + 32 * dt))[0] = k0;
      ((u32x4*)(orow + H + 32 * dt))[1] = k1;
      ((u32x4*)(orow + 2 * H + 32 * dt))[0] = v0;
      ((u32x4*)(orow + 2 * H + 32 * dt))[1] = v1;
    }
  }
}

// ================================================================================================
// 16-wave form (round 4): wave w owns the 16 keys 16w .. 16w+15, every product on v_mfma_f32_16x16x32_bf16, <= 128
// registers, FOUR waves per SIMD.  The 8-wave kernel above sits at 235 registers and 83 KB of LDS -- one workgroup per CU,
// two waves per SIMD that reach the slice barrier together, 40 % of their cycles parked at waits
// (profiles/r03/attention_sq_counters.txt) -- and spends ~13 vector instructions per score element.  Here:
//   * a wave's S / dP / P / dS tiles are 16 keys x 32 queries (8 elements per lane); its K and V fragments (B operands,
//     lane = key) are 16 registers, its dK^T / dV^T accumulators 32;
//   * one iteration = 64 queries in two sub-slices of 32 (S, dP, element-wise, dV^T, dK^T per sub-slice), then ONE barrier
//     and the dQ product of all 64 queries: 16 output tiles of 16 d x 16 queries, one per wave -- half as many barriers
//     per (batch, head) as with 32-query slices;
//   * element-wise work per score element: c = key bias - lse (per element: the two live on different axes), p =
//     exp2(fma(acc, scale * log2 e, c)) -- the scale, the bias, the log-sum-exp and the change of base in ONE fma --, the
//     keep bit (bit-field extract, two ands), dS = p (m dP - delta): 9-10 instructions.
// The A operands that contract over the queries (dO^T, Q^T for dV^T, dK^T) come from the row-major slices by
// ds_read_b64_tr_b16 in the k-slot order the accumulators already have: lane group g of a P / dS tile pair holds the queries
// {4g..4g+3, 16+4g..16+4g+3} of the sub-slice, so B = (tile 0 registers | tile 1 registers) needs no lane movement.
// dS crosses LDS once, in the [4-query group][key][4 queries] image of the kernels above.  Serves S <= 256 (one key block),
// per-key masks or none, dropout through the forward's keep words (or none), delta formed in the kernel; everything else
// stays with the kernels above.
#define AW_K 0                       // [256 keys][128 B]
#define AW_Q 32768                   // [256 q][128 B]: the whole sequence's Q rows (this head)
#define AW_DO (AW_Q + 32768)         // [256 q][128 B]: its dO rows
#define AW_DS (AW_DO + 32768)        // [16 q-groups][256 keys][8 B]: the dS of one iteration (64 queries)
#define AW_ROW (AW_DS + 32768)       // lse2[256] | delta[256] fp32
#define AW_LDS_BYTES (AW_ROW + 2048)

// Everything a (batch, head) needs is fetched ONCE, in the prologue, and stays in LDS (K, Q, dO: 3 x 32 KB at S <= 256): a
// workgroup that owns a whole compute unit has nothing to hide a global load behind, and the slice-by-slice kernels above pay
// one exposed load latency per slice (8 per (batch, head): 25 us per workgroup for ~5 us of MFMA and ~5 us of vector work).
// AW_LAB (tools/experiments/Makefile builds libvisitron_hip_awlab<N>.so with -DAW_LAB=<bits>; 0 / undefined in the product):
// timing-only removals -- 1 no iteration loop (prologue + epilogue), 2 no dQ phase, 4 trivial element-wise work, 8 no dV^T / dK^T
// products (and their transposed reads), 16 no S / dP products (and their row reads), 32 no workgroup barriers in the loop
#ifndef AW_LAB
#define AW_LAB 0
#endif
template <bool BITS>
__global__ __launch_bounds__(1024) void attention_bwd_d64_w16(AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane & 15, g = lane >> 4;          // this lane's key inside the wave's 16; k-slot group
  const int b = blockIdx.y, head = blockIdx.x;
  const int Smax = a.S, H = a.nh * 64;
  const int S = a.seq_len ? vt_clamp_len(a.seq_len[b], a.S) : a.S;
  const long row0 = a.seq_start ? (long)a.seq_start[b] : (long)b * a.S;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // wave-uniform bases (64-bit, in SGPRs) + 32-bit per-lane byte offsets: per-lane 64-bit pointers cost two registers each
  // in a kernel that has none to spare
  const bf16_t* base = a.qkv + row0 * a.ld_qkv + head * 64;
  const char* qb8 = (const char*)base;
  const char* dob8 = (const char*)(a.dctx + row0 * a.ld_d + head * 64);
  const char* ob8 = (const char*)(a.ctx + row0 * a.ld_ctx + head * 64);
  char* dq8 = (char*)(a.dqkv + row0 * a.ld_dqkv + head * 64);
  const unsigned ldq_b = (unsigned)a.ld_qkv * 2u, ldd_b = (unsigned)a.ld_d * 2u, ldo_b = (unsigned)a.ld_ctx * 2u, lddq_b = (unsigned)a.ld_dqkv * 2u;
  const float* lse_p = a.lse + ((long)b * a.nh + head) * Smax;
  const int niter = (S + 63) >> 6;
  const int nkt = (S + 31) >> 5;                   // 32-key steps of the dQ product
  const float drop_scale = a.drop.thresh ? a.drop.scale : 1.0f;
  const float ds_scale = a.scale * drop_scale;     // 1 / sqrt(d) times dropout's 1 / (1-p): applied to dK once and to dQ per tile
  const float scale2 = a.scale * LOG2E;

  // ---- prologue.  Register loads first (row constants, the dO / O chunks of delta), then the DMA, then the LDS stores: an
  // LDS access behind an LDS-DMA makes hipcc wait for the DMA, which would put a second load latency behind the first.
  float* rowc = (float*)(smem + AW_ROW);
  const float lse_v = (tid < 256 && tid < S) ? lse_p[tid] * LOG2E : INFINITY;   // +inf past the sequence: P = 0
  u32x2 xd[4], xo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {                      // thread t: query 64 i + (t >> 4), head columns 4 (t & 15) .. + 3
    const int qi = 64 * i + (tid >> 4);
    const unsigned q = qi < S ? qi : S - 1;
    xd[i] = *(const u32x2*)(dob8 + (q * ldd_b + 8u * (tid & 15)));
    xo[i] = *(const u32x2*)(ob8 + (q * ldo_b + 8u * (tid & 15)));
  }
  // K rows (image of the kernels above), Q rows and dO rows (chunk swizzle (row >> 1) & 7) -> LDS by DMA, 8-row pieces of
  // 1 KiB; rows past the sequence repeat its last row (their P is 0)
  for (int j = wave; j < nkt * 4; j += 16) {
    const int row = 8 * j + (lane >> 3);
    const int kr = row < S ? row : S - 1;
    const int cs = lane & 7;
    const int sw = (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
    glds16(base + (long)kr * a.ld_qkv + H + ((cs ^ sw) << 3), smem + AW_K + j * 1024);
  }
  for (int j = wave; j < niter * 8; j += 16) {
    const int row = 8 * j + (lane >> 3);
    const unsigned q = row < S ? row : S - 1;
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    glds16(qb8 + (q * ldq_b + 16u * c), smem + AW_Q + j * 1024);
    glds16(dob8 + (q * ldd_b + 16u * c), smem + AW_DO + j * 1024);
  }
  // the keep words of this lane's key: one per 32-query sub-slice, fetched an iteration ahead
  const int key = 16 * wave + kl;
  const unsigned keep_step = (unsigned)((Smax + 31) >> 5) << 5;
  const uint32_t* keep_u = BITS ? a.keep_bits + ((long)b * a.nh + head) * ((Smax + 31) >> 5) * (long)keep_step : nullptr;   // uniform
  const unsigned keep_k = (unsigned)key < keep_step ? (unsigned)key : keep_step - 1u;
  auto keep_word = [&](int qb) -> uint32_t { return keep_u[(unsigned)qb * keep_step + keep_k]; };
  const int nqb_s = (S + 31) >> 5;                 // sub-slices that hold queries of this sequence
  uint32_t kwn0 = BITS ? keep_word(0) : 0u, kwn1 = (BITS && nqb_s > 1) ? keep_word(1) : 0u;

  // ---- this wave's K and V fragments (B operands: lane = key, 8 head columns at 32 ks + 8 g) and its key bias ----
  const int kr = key < S ? key : S - 1;
  bf16x8 kf[2], vf[2];
  {
    const bf16_t* kp = base + (long)kr * a.ld_qkv + H + 8 * g;
    const bf16_t* vp = base + (long)kr * a.ld_qkv + 2 * H + 8 * g;
    kf[0] = *(const bf16x8*)kp; kf[1] = *(const bf16x8*)(kp + 32);
    vf[0] = *(const bf16x8*)vp; vf[1] = *(const bf16x8*)(vp + 32);
  }
  float kb2 = -INFINITY;                           // key bias in the exp2 domain; -inf: a key past the sequence (p = 0)
  if (key < S) {
    float add = 0.f;
    if (a.mask) {
      const float mval = a.mask[(long)b * Smax + key];
      add = a.mask_additive ? mval : (1.0f - mval) * -10000.0f;
    }
    kb2 = add * LOG2E;
  }
  f32x4 dv[4], dk[4];   // [d tile]: lane = key, register j <-> head column 16 dt + 4 g + j
#pragma unroll
  for (int i = 0; i < 4; ++i) { dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // lane-constant addressing.  Row reads (A of S, dP): row 16 t + kl of the sub-slice, 16-byte chunk 4 ks + g, chunk
  // swizzle (row >> 1) & 7.  Transposed reads (A of dV^T, dK^T): lane 4 q' + p of a 16-lane group supplies row 4 g + q' (+ 16 t),
  // head columns 16 dt + 4 p .. + 3.
  const int qp = kl >> 2, pp = kl & 3;
  const int tr_row = 4 * g + qp;
  const int dq_dt = wave >> 2, dq_qt = wave & 3;   // this wave's 16 (d) x 16 (q) tile of the iteration's dQ^T

  // row constants of every query -> LDS: lse2 = lse log2(e), delta = rowsum(dO o O) x (1 - p_drop) (sum over the 16 lanes of a
  // query, one DPP row: xor 1, xor 2, mirror within 8, mirror within 16)
  if (tid < 256) rowc[tid] = lse_v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v = bf16lo(xd[i][0]) * bf16lo(xo[i][0]) + bf16hi(xd[i][0]) * bf16hi(xo[i][0]) + bf16lo(xd[i][1]) * bf16lo(xo[i][1]) +
              bf16hi(xd[i][1]) * bf16hi(xo[i][1]);
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
    if ((tid & 15) == 0) rowc[256 + 64 * i + (tid >> 4)] = v * a.delta_mul;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int it = 0; it < ((AW_LAB & 1) ? 0 : niter); ++it) {
    const unsigned sQ = lds0 + AW_Q + it * 8192, sDO = lds0 + AW_DO + it * 8192;
    const float* rowv = rowc + 64 * it;
    char* sDS = smem + AW_DS;
    const uint32_t kw0 = kwn0, kw1 = kwn1;
    if (BITS && it + 1 < niter) {                   // the next iteration's keep words
      kwn0 = keep_word(2 * it + 2);
      kwn1 = (2 * it + 3 < nqb_s) ? keep_word(2 * it + 3) : 0u;
    }

#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      if (it * 64 + ss * 32 >= S) break;           // uniform: the sub-slice holds no query of this sequence
      const unsigned qb_ = sQ + ss * 4096, db_ = sDO + ss * 4096;
      // ---- S = Q K^T and dP = dO V^T for this wave's 16 keys (lane = key; register j of tile t <-> query 16 t + 4 g + j) ----
      f32x4 sacc[2], dpacc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = 16 * t + kl;
        const unsigned ro = row * 128, sw = (row >> 1) & 7;
        const bf16x8 q0 = *(const bf16x8*)(smem + (qb_ - lds0) + ro + (((0 + g) ^ sw) << 4));
        const bf16x8 q1 = *(const bf16x8*)(smem + (qb_ - lds0) + ro + (((4 + g) ^ sw) << 4));
        const bf16x8 d0 = *(const bf16x8*)(smem + (db_ - lds0) + ro + (((0 + g) ^ sw) << 4));
        const bf16x8 d1 = *(const bf16x8*)(smem + (db_ - lds0) + ro + (((4 + g) ^ sw) << 4));
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (AW_LAB & 16) { sacc[t] = (f32x4){kb2, kb2, 0.f, 0.f}; dpacc[t] = sacc[t]; continue; }
        sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, kf[0], z, 0, 0, 0);
        sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, kf[1], sacc[t], 0, 0, 0);
        dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, vf[0], z, 0, 0, 0);
        dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, vf[1], dpacc[t], 0, 0, 0);
      }
      // ---- P = exp2(acc scale2 + key bias - lse2);  dS = P (keep dP - delta) ----
      const uint32_t kw = (ss == 0 ? kw0 : kw1) >> (4 * g);   // bit 16 t + j = the keep flag of query 16 t + 4 g + j
      float pm[8], dsv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 l4 = *(const f32x4*)(rowv + 32 * ss + 16 * t + 4 * g);
        const f32x4 e4 = *(const f32x4*)(rowv + 256 + 32 * ss + 16 * t + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (AW_LAB & 4) { pm[4 * t + j] = sacc[t][j]; dsv[4 * t + j] = dpacc[t][j] + l4[j] + e4[j]; continue; }
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[t][j], scale2, kb2 - l4[j]));
          float dpv = dpacc[t][j], pv = p;
          if (BITS) {   // (launched only with dropout on)
            const int km = __builtin_amdgcn_sbfe((int)kw, 16 * t + j, 1);   // 0 / -1
            dpv = __int_as_float(__float_as_int(dpv) & km);
            pv = __int_as_float(__float_as_int(p) & km);
          }
          pm[4 * t + j] = pv;
          dsv[4 * t + j] = p * (dpv - e4[j]);        // up to ds_scale; delta arrives times (1 - p_drop), see the kernels above
        }
      }
      u32x4 pw4, sw4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pw4[i] = pack_bf16x2(pm[2 * i], pm[2 * i + 1]);
        sw4[i] = pack_bf16x2(dsv[2 * i], dsv[2 * i + 1]);
      }
      const bf16x8 pb = __builtin_bit_cast(bf16x8, pw4), sb = __builtin_bit_cast(bf16x8, sw4);
      // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 32 queries, k-slot order = the tiles' own) ----
#pragma unroll
      for (int dt = 0; dt < ((AW_LAB & 8) ? 0 : 4); ++dt) {
        const int cb = 32 * dt + 8 * pp;            // byte offset of head columns 16 dt + 4 pp inside the 128-byte row
        const int r0 = tr_row, r1 = tr_row + 16;
        const unsigned o0 = r0 * 128 + ((((cb >> 4) ^ ((r0 >> 1) & 7)) << 4) | (cb & 8));
        const unsigned o1 = r1 * 128 + ((((cb >> 4) ^ ((r1 >> 1) & 7)) << 4) | (cb & 8));
        const bf16x8 dot = tr_pair_b(db_ + o0, (int)(o1 - o0));
        const bf16x8 qt = tr_pair_b(qb_ + o0, (int)(o1 - o0));
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pb, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, sb, dk[dt], 0, 0, 0);
      }
      // ---- dS -> LDS image [q-group G][key][4 q] (8 B per (G, key)), key index XOR-swizzled by G ----
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int G = 8 * ss + 4 * t + g;
        const int kx = key ^ (((G >> 1) & 1) << 2) ^ ((G & 1) << 4);
        u32x2 w;
        w[0] = sw4[2 * t]; w[1] = sw4[2 * t + 1];
        *(u32x2*)(sDS + (G * 256 + kx) * 8) = w;
      }
    }

    if (!(AW_LAB & 32)) __syncthreads();   // every wave's dS is in the image

    // ---- dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]: this wave's tile = head columns 16 dq_dt .., queries 16 dq_qt .. ----
    if (!(AW_LAB & 2) && it * 64 + 16 * dq_qt < S) {   // uniform: the tile holds a query of this sequence
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
      const unsigned kbase = lds0 + AW_K, dsb = lds0 + AW_DS;
      const int G = 4 * dq_qt + pp;
      const int gsw = (((G >> 1) & 1) << 2) ^ ((G & 1) << 4);
      for (int ks = 0; ks < nkt; ++ks) {
        const int krow = 32 * ks + 8 * g + qp;         // first block row; second block = +4
        const int ksw = ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 5);
        const int ksw2 = (((((krow + 4) >> 1) & 1) | ((((krow + 4) >> 3) & 1) << 1)) << 5);
        const int kcol = 2 * (16 * dq_dt) + 8 * pp;
        const bf16x8 ka = tr_pair_b(kbase + krow * 128 + (kcol ^ ksw), 4 * 128 + ((kcol ^ ksw2) - (kcol ^ ksw)));
        const unsigned s0a = dsb + (G * 256 + (krow ^ gsw)) * 8, s1a = dsb + (G * 256 + ((krow + 4) ^ gsw)) * 8;
        const bf16x8 sbq = tr_pair_b(s0a, (int)(s1a - s0a));
        dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, sbq, dq, 0, 0, 0);
      }
      const int q = it * 64 + 16 * dq_qt + kl;         // lane (query kl of the tile, g): head columns 16 dq_dt + 4 g .. + 3
      if (q < S) {
        u32x2 w;
        w[0] = pack_bf16x2(dq[0] * ds_scale, dq[1] * ds_scale);
        w[1] = pack_bf16x2(dq[2] * ds_scale, dq[3] * ds_scale);
        *(u32x2*)(dq8 + ((unsigned)q * lddq_b + 2u * (16 * dq_dt + 4 * g))) = w;
      }
    }
    if (!(AW_LAB & 32) && it + 1 < niter) __syncthreads();   // the (single) dS image is rewritten by the next iteration
  }

  // ---- dK, dV of this wave's keys: lane = key, register j of tile dt <-> head column 16 dt + 4 g + j ----
  if (key < S) {
    char* orow = dq8 + ((unsigned)key * lddq_b + 8u * g);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      u32x2 kq, vq;
      kq[0] = pack_bf16x2(dk[dt][0] * ds_scale, dk[dt][1] * ds_scale);
      kq[1] = pack_bf16x2(dk[dt][2] * ds_scale, dk[dt][3] * ds_scale);
      vq[0] = pack_bf16x2(dv[dt][0] * drop_scale, dv[dt][1] * drop_scale);
      vq[1] = pack_bf16x2(dv[dt][2] * drop_scale, dv[dt][3] * drop_scale);
      *(u32x2*)(orow + 2 * (H + 16 * dt)) = kq;
      *(u32x2*)(orow + 2 * (2 * H + 16 * dt)) = vq;
    }
  }
}

// ================================================================================================
// 16-wave form, PERSISTENT and software-pipelined (round 5).  profiles/r05/attention_bwd_ablation.txt: the kernel above with
// its iteration loop removed -- prologue and epilogue alone -- takes 137 of its 329 us: a workgroup owns its compute unit, every
// workgroup of the grid fetches its 175 KB (Q, K, V, dO, O) at the same time at HBM's full rate while no MFMA runs, then all of
// them compute while HBM idles.  Here one workgroup per compute unit walks its share of the (batch, head) pairs and the loads of
// what comes next run under the arithmetic of what is there:
//   * Q, dO and O rows travel in 64-query slots (24 KB: two slots), fetched TWO iterations ahead by LDS-DMA across pair
//     boundaries, the slot's lse as one 256-byte piece; dO goes through a buffer descriptor, so rows past the sequence read
//     as ZERO -- their dP, delta, dS and dV terms vanish whatever P is -- while Q and lse repeat the last row (a finite P);
//   * delta = rowsum(dO o O) of a landed slot is the DIAGONAL of dO O^T: four waves take two MFMAs each in the dQ phase of
//     the iteration before the slot's use (an all-thread pass over the slot cost 28 us per launch, a pre-pass kernel 38);
//   * the K image is double-buffered (the next pair's keys land during the current pair's first iteration); the next pair's V
//     fragments, key bias and first keep words are fetched into registers during the pair's last dQ phase -- the only point
//     where 16 registers are free in a 128-register kernel -- and its K fragments come from the landed image;
//   * dK / dV leave at the pair switch, dQ per iteration as before.
// The LDS-DMA instructions are issued from inline asm: hipcc orders every LDS access behind a pending LDS-DMA it knows of
// (s_waitcnt vmcnt(0)), which would expose the latency this kernel exists to hide; the landing wait is the explicit
// vmcnt(0) in front of the iteration's dS barrier, a full iteration after issue.  Arithmetic, tile shapes, LDS images and
// the order of the sums over keys and queries are those of attention_bwd_d64_w16; delta and the exponent's bias are rounded
// differently (fp32 MFMA sum; one fma), so results agree to rounding, not bitwise (tests/test_gpu_round5.py).
#define AP_K 0                        // 2 x [256 keys][128 B]
#define AP_RING 65536                 // 2 slots x (Q [64 q][128 B] | dO [64 q][128 B] | O [64 q][128 B]), 16-B chunk XOR AP_SWZ(row)
#define AP_SLOT 24576
// Chunk swizzle of the slot images.  The round-4 images used (row >> 1) & 7: conflict-free for the row reads (ds_read_b128) but
// 2-way conflicting on EVERY transposed read of the dV^T / dK^T operands (rows 4 g + q' of a 32-lane group: rows 0 and 2 landed in
// the same 32-byte bank group) -- SQ_LDS_BANK_CONFLICT 1.16e7 of 5.9e7 LDS cycles per launch.  row & 6 is conflict-free for both
// patterns (exhaustive check over the xor-linear swizzles: profiles/r05/attention_bwd_persistent.txt).
#define AP_SWZ(row) ((row) & 6)
#define AP_DS (AP_RING + 2 * AP_SLOT) // [16 q-groups][256 keys][8 B]
#define AP_ROW (AP_DS + 32768)        // 2 slots x (lse[64] | delta[64]) fp32
#define AP_KEEP (AP_ROW + 1024)       // 2 slots x 2 sub-slices x [256 keys] keep words
#define AP_META (AP_KEEP + 4096)      // [B <= 1024] x (sequence length, first row): scalar metadata the loop reads from LDS
#define AP_MAX_B 1024
#define AP_LDS_BYTES (AP_META + 8 * AP_MAX_B)

// 16 bytes per lane global -> LDS (wave-uniform LDS byte address in M0), invisible to the compiler's waitcnt insertion
__device__ __forceinline__ void ap_dma16(const void* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}
__device__ __forceinline__ void ap_dma4(const void* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}
// the same through a buffer descriptor: bytes at or past `bytes` read as ZERO (rows past the sequence)
__device__ __forceinline__ u32x4 ap_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void ap_bdma16(u32x4 rs, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ void ap_bdma4(u32x4 rs, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}

// AP_LAB (tools/experiments, timing only; 0 / undefined in the product): 1 no LDS-DMA after the prologue (arithmetic on stale
// slots), 4 no arithmetic (the transfer pipeline alone), 8 no dQ stores
#ifndef AP_LAB
#define AP_LAB 0
#endif
// 1: this wave's K fragments are re-read from the key image in every sub-slice instead of living in registers for the pair
#ifndef AP_KF_LDS
#define AP_KF_LDS 1
#endif
template <bool BITS>
__global__ __launch_bounds__(1024) void attention_bwd_d64_w16p(AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane & 15, g = lane >> 4;
  const int Smax = a.S, H = a.nh * 64;
  const int npairs = a.B * a.nh, G = gridDim.x;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned ldq_b = (unsigned)a.ld_qkv * 2u, ldd_b = (unsigned)a.ld_d * 2u, ldo_b = (unsigned)a.ld_ctx * 2u, lddq_b = (unsigned)a.ld_dqkv * 2u;
  const float drop_scale = a.drop.thresh ? a.drop.scale : 1.0f;
  const float ds_scale = a.scale * drop_scale;
  const float scale2 = a.scale * LOG2E;
  const int key = 16 * wave + kl;
  const unsigned keep_step = (unsigned)((Smax + 31) >> 5) << 5;
  const long keep_pair = (long)((Smax + 31) >> 5) * (long)keep_step;   // keep words per (batch, head)

  // Per-batch metadata -> LDS once.  Inside the loop NOTHING is loaded from global memory by compiler-visible instructions
  // except at a pair's end: hipcc fetches even uniform values with vector loads once the kernel contains stores, and waits
  // for each with vmcnt(0) -- which would also wait for every LDS-DMA piece in flight, the latency this kernel hides.
  int* meta = (int*)(smem + AP_META);
  if (tid < a.B) {
    meta[2 * tid] = a.seq_len ? vt_clamp_len(a.seq_len[tid], a.S) : a.S;
    meta[2 * tid + 1] = a.seq_start ? a.seq_start[tid] : tid * a.S;
  }
  __syncthreads();
  auto pair_S = [&](int p) -> int { return __builtin_amdgcn_readfirstlane(meta[2 * (p / a.nh)]); };
  auto pair_rows = [&](int p) -> long { return (long)__builtin_amdgcn_readfirstlane(meta[2 * (p / a.nh) + 1]); };
  // this workgroup's k-th pair, or npairs when its list has ended.  (A length-balanced order -- batches ranked by length in
  // the kernel, dealt boustrophedon -- was built and measured: no gain, 272 -> 277 us on the compacted batch.  A pair's work
  // is quantised in 64-query iterations, so sequences of 193 .. 256 rows all cost four of them.)
  auto pair_at = [&](int k) -> int {
    const int i = k * G + (int)blockIdx.x;
    return i < npairs ? i : npairs;
  };

  // ---- the slot cursor: (pair, iteration) of the next 64-query slot to fetch; runs two iterations ahead of the arithmetic ----
  int ck = 0, cp = pair_at(0), cit = 0, cS = pair_S(cp);
  auto issue_slot = [&](int slot) {
    if (cp >= npairs) return;                       // uniform: past this workgroup's last pair
    const long r0 = pair_rows(cp);
    const int head = cp % a.nh;
    const unsigned sb = lds0 + AP_RING + slot * AP_SLOT;
    const int pw = wave & 7;
    const int row = 8 * pw + (lane >> 3);           // slot-relative query row of this lane's 16 bytes
    const int qa = cit * 64 + row;
    const unsigned q = qa < cS ? qa : cS - 1;       // Q rows past the sequence repeat its last row (dO reads zeros there)
    const unsigned c16 = 16u * ((lane & 7) ^ AP_SWZ(row));
    if (wave < 8) {
      ap_dma16(a.qkv + r0 * a.ld_qkv + head * 64, q * ldq_b + c16, sb + pw * 1024);
      ap_dma16(a.ctx + r0 * a.ld_ctx + head * 64, q * ldo_b + c16, sb + 16384 + pw * 1024);
      if (wave == 0) {                              // the slot's 64 log-sum-exps (natural log, as the forward left them)
        const int qi = cit * 64 + lane;
        ap_dma4(a.lse + (long)cp * Smax, 4u * (unsigned)(qi < cS ? qi : cS - 1), lds0 + AP_ROW + slot * 512);
      }
    } else {
      const void* db = a.dctx + r0 * a.ld_d + head * 64;
      ap_bdma16(ap_rsrc(db, (unsigned)(cS - 1) * ldd_b + 128u), (unsigned)qa * ldd_b + c16, sb + 8192 + pw * 1024);
    }
    if (BITS && (pw >= 1 && pw <= 4)) {             // keep words of the slot's two 32-query sub-slices: 4 pieces of 64 keys each
      const int ss = wave >> 3, j = pw - 1;
      const int qb = 2 * cit + ss;
      if (qb * 32 < cS && 64 * j < (int)keep_step)
        ap_bdma4(ap_rsrc(a.keep_bits + (long)cp * keep_pair + (long)qb * keep_step, 4u * keep_step), 4u * (unsigned)(64 * j + lane),
                 lds0 + AP_KEEP + slot * 2048 + ss * 1024 + j * 256);
    }
    if (++cit * 64 >= cS) { cit = 0; cp = pair_at(++ck); cS = cp < npairs ? pair_S(cp) : 0; }
  };
  // the key image of pair p -> K[par]: 8-row pieces of 1 KiB, rows past the sequence repeat its last row
  auto issue_keys = [&](int p, int par) {
    const int S = pair_S(p);
    const bf16_t* kb = a.qkv + pair_rows(p) * a.ld_qkv + (p % a.nh) * 64 + H;
    const int np = ((S + 31) >> 5) * 4;
    for (int j = wave; j < np; j += 16) {
      const int row = 8 * j + (lane >> 3);
      const unsigned kr = row < S ? row : S - 1;
      const int sw = (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
      ap_dma16(kb, kr * ldq_b + 16u * ((lane & 7) ^ sw), lds0 + AP_K + par * 32768 + j * 1024);
    }
  };
  // this lane's V fragments and raw mask value of pair p: plain loads into registers, issued BEFORE the step's LDS-DMA pieces
  // (a compiler-placed wait for them then drains nothing younger) and consumed behind the pair's final drain
  auto load_vm = [&](int p, bf16x8 (&vf)[2], float& mval) {
    const int S = pair_S(p), b = p / a.nh;
    const int kr = key < S ? key : S - 1;
    const bf16_t* vp = a.qkv + (pair_rows(p) + kr) * a.ld_qkv + (p % a.nh) * 64 + 2 * H + 8 * g;
    vf[0] = *(const bf16x8*)vp; vf[1] = *(const bf16x8*)(vp + 32);
    mval = a.mask ? a.mask[(long)b * Smax + kr] : 1.0f;
  };
  auto key_bias = [&](int p, float mval) -> float {  // in the exp2 domain; -inf: a key past the sequence (p = 0)
    if (key >= pair_S(p)) return -INFINITY;
    const float add = !a.mask ? 0.f : a.mask_additive ? mval : (1.0f - mval) * -10000.0f;
    return add * LOG2E;
  };
  // delta of a landed slot: waves 0 .. 3 take 16 queries each; D = dO_tile O_tile^T (16 x 16, two MFMAs over the 64 head
  // columns), delta[q] = D[q][q] x (1 - p_drop): lane (kl, g) holds rows 4 g .. 4 g + 3 of column kl
  auto slot_delta = [&](int slot) {
    if (wave >= 4) return;
    const int row = 16 * wave + kl;
    const unsigned ro = row * 128, sw = AP_SWZ(row);
    const char* sd = smem + AP_RING + slot * AP_SLOT + 8192 + ro;
    const char* so = smem + AP_RING + slot * AP_SLOT + 16384 + ro;
    const bf16x8 d0 = *(const bf16x8*)(sd + (((0 + g) ^ sw) << 4)), d1 = *(const bf16x8*)(sd + (((4 + g) ^ sw) << 4));
    const bf16x8 o0 = *(const bf16x8*)(so + (((0 + g) ^ sw) << 4)), o1 = *(const bf16x8*)(so + (((4 + g) ^ sw) << 4));
    f32x4 dd = {0.f, 0.f, 0.f, 0.f};
    dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, o0, dd, 0, 0, 0);
    dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, o1, dd, 0, 0, 0);
    const int r = kl & 3;
    const float v = r == 0 ? dd[0] : r == 1 ? dd[1] : r == 2 ? dd[2] : dd[3];
    if (g == (kl >> 2)) ((float*)(smem + AP_ROW + slot * 512))[64 + row] = v * a.delta_mul;
  };
  // K fragments of this wave's 16 keys from the landed image (B operands: lane = key, 8 head columns at 32 ks + 8 g)
  auto read_kf = [&](int par, bf16x8 (&kf)[2]) {
    const int sw = ((((key >> 1) & 1) | (((key >> 3) & 1) << 1)) << 1);
    const char* kr = smem + AP_K + par * 32768 + key * 128;
    kf[0] = *(const bf16x8*)(kr + (((0 + g) ^ sw) << 4));
    kf[1] = *(const bf16x8*)(kr + (((4 + g) ^ sw) << 4));
  };

  // lane-constant addressing (see attention_bwd_d64_w16)
  const int qp = kl >> 2, pp = kl & 3;
  const int tr_row = 4 * g + qp;
  const int dq_dt = wave >> 2, dq_qt = wave & 3;

  // ---- prologue: the first pair's keys, slots 0 and 1, its V fragments / bias / keep words; then slot 0's row constants ----
  int p = pair_at(0);
  bf16x8 kf[2], vf[2];
  float kb2, mv;
  load_vm(p, vf, mv);
  issue_keys(p, 0);
  issue_slot(0);
  issue_slot(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  slot_delta(0);
  if (!AP_KF_LDS) read_kf(0, kf);
  kb2 = key_bias(p, mv);
  __syncthreads();

  int par = 0, J = 0;                               // K buffer of the current pair; global iteration count (slot = J & 1)
  for (int k = 0; p < npairs; ++k) {
    const int S = pair_S(p);
    const int pn = pair_at(k + 1);
    const bool has_next = pn < npairs;
    const int niter = (S + 63) >> 6, nkt = (S + 31) >> 5;
    char* dq8 = (char*)(a.dqkv + pair_rows(p) * a.ld_dqkv + (p % a.nh) * 64);
    f32x4 dv[4], dk[4];   // [d tile]: lane = key, register j <-> head column 16 dt + 4 g + j
#pragma unroll
    for (int i = 0; i < 4; ++i) { dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    bf16x8 vfn[2];
    float mvn = 1.0f;

    // one 64-query iteration; LAST (compile-time) = the pair's last: the next pair's V fragments, bias and keep words are
    // fetched in its dQ phase (two instantiations, so that those registers are not live across the score phase)
    auto iteration = [&](const int it, auto last_tag) {
      constexpr bool last = decltype(last_tag)::value;
      const int slot = J & 1;
      const unsigned sQ = lds0 + AP_RING + slot * AP_SLOT, sDO = sQ + 8192;
      const float* rowv = (const float*)(smem + AP_ROW + slot * 512);
      char* sDS = smem + AP_DS;
      const uint32_t* keepw = (const uint32_t*)(smem + AP_KEEP + slot * 2048);

#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        if ((AP_LAB & 4) || it * 64 + ss * 32 >= S) break;   // uniform: the sub-slice holds no query of this sequence
        const unsigned qb_ = sQ + ss * 4096, db_ = sDO + ss * 4096;
        // ---- S = Q K^T and dP = dO V^T for this wave's 16 keys (lane = key; register j of tile t <-> query 16 t + 4 g + j) ----
        f32x4 sacc[2], dpacc[2];
        if (AP_KF_LDS) read_kf(par, kf);            // (re-read per sub-slice: eight registers the pair's end needs for the next V)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = 16 * t + kl;
          const unsigned ro = row * 128, sw = AP_SWZ(row);
          const bf16x8 q0 = *(const bf16x8*)(smem + (qb_ - lds0) + ro + (((0 + g) ^ sw) << 4));
          const bf16x8 q1 = *(const bf16x8*)(smem + (qb_ - lds0) + ro + (((4 + g) ^ sw) << 4));
          const bf16x8 d0 = *(const bf16x8*)(smem + (db_ - lds0) + ro + (((0 + g) ^ sw) << 4));
          const bf16x8 d1 = *(const bf16x8*)(smem + (db_ - lds0) + ro + (((4 + g) ^ sw) << 4));
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, kf[0], z, 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, kf[1], sacc[t], 0, 0, 0);
          dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, vf[0], z, 0, 0, 0);
          dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, vf[1], dpacc[t], 0, 0, 0);
        }
        // ---- P = exp2(acc scale2 + key bias - lse2);  dS = P (keep dP - delta) ----
        const uint32_t kw = BITS ? keepw[ss * 256 + key] >> (4 * g) : 0u;   // bit 16 t + j = the keep flag of query 16 t + 4 g + j
        float pm[8], dsv[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 l4 = *(const f32x4*)(rowv + 32 * ss + 16 * t + 4 * g);
          const f32x4 e4 = *(const f32x4*)(rowv + 64 + 32 * ss + 16 * t + 4 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float pr = __builtin_amdgcn_exp2f(fmaf(sacc[t][j], scale2, fmaf(l4[j], -LOG2E, kb2)));
            float dpv = dpacc[t][j], pv = pr;
            if (BITS) {   // (launched only with dropout on)
              const int km = __builtin_amdgcn_sbfe((int)kw, 16 * t + j, 1);   // 0 / -1
              dpv = __int_as_float(__float_as_int(dpv) & km);
              pv = __int_as_float(__float_as_int(pr) & km);
            }
            pm[4 * t + j] = pv;
            dsv[4 * t + j] = pr * (dpv - e4[j]);
          }
        }
        u32x4 pw4, sw4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pw4[i] = pack_bf16x2(pm[2 * i], pm[2 * i + 1]);
          sw4[i] = pack_bf16x2(dsv[2 * i], dsv[2 * i + 1]);
        }
        const bf16x8 pb = __builtin_bit_cast(bf16x8, pw4), sb = __builtin_bit_cast(bf16x8, sw4);
        // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 32 queries, k-slot order = the tiles' own) ----
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int cb = 32 * dt + 8 * pp;
          const int r0 = tr_row, r1 = tr_row + 16;
          const unsigned o0 = r0 * 128 + ((((cb >> 4) ^ AP_SWZ(r0)) << 4) | (cb & 8));
          const unsigned o1 = r1 * 128 + ((((cb >> 4) ^ AP_SWZ(r1)) << 4) | (cb & 8));
          const bf16x8 dot = tr_pair_b(db_ + o0, (int)(o1 - o0));
          const bf16x8 qt = tr_pair_b(qb_ + o0, (int)(o1 - o0));
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pb, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, sb, dk[dt], 0, 0, 0);
        }
        // ---- dS -> LDS image [q-group G][key][4 q] (8 B per (G, key)), key index XOR-swizzled by G ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int Gq = 8 * ss + 4 * t + g;
          const int kx = key ^ (((Gq >> 1) & 1) << 2) ^ ((Gq & 1) << 4);
          u32x2 w;
          w[0] = sw4[2 * t]; w[1] = sw4[2 * t + 1];
          *(u32x2*)(sDS + (Gq * 256 + kx) * 8) = w;
        }
      }

      // slot J + 1 (and, behind it, the next pair's keys) has had a whole iteration to land
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // every wave's dS is in the image; nobody reads slot J's rows any more; slot J + 1 is visible
      if (last && has_next) load_vm(pn, vfn, mvn);
      if (!(AP_LAB & 1)) issue_slot(slot);  // global iteration J + 2 -> the slot just freed
      if (!(AP_LAB & 1) && it == 0 && has_next) issue_keys(pn, par ^ 1);

      // ---- dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]: this wave's tile = head columns 16 dq_dt .., queries 16 dq_qt .. ----
      if (!(AP_LAB & 4) && it * 64 + 16 * dq_qt < S) {   // uniform: the tile holds a query of this sequence
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        const unsigned kbase = lds0 + AP_K + par * 32768, dsb = lds0 + AP_DS;
        const int Gq = 4 * dq_qt + pp;
        const int gsw = (((Gq >> 1) & 1) << 2) ^ ((Gq & 1) << 4);
        for (int ks = 0; ks < nkt; ++ks) {
          const int krow = 32 * ks + 8 * g + qp;         // first block row; second block = +4
          const int ksw = ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 5);
          const int ksw2 = (((((krow + 4) >> 1) & 1) | ((((krow + 4) >> 3) & 1) << 1)) << 5);
          const int kcol = 2 * (16 * dq_dt) + 8 * pp;
          const bf16x8 ka = tr_pair_b(kbase + krow * 128 + (kcol ^ ksw), 4 * 128 + ((kcol ^ ksw2) - (kcol ^ ksw)));
          const unsigned s0a = dsb + (Gq * 256 + (krow ^ gsw)) * 8, s1a = dsb + (Gq * 256 + ((krow + 4) ^ gsw)) * 8;
          const bf16x8 sbq = tr_pair_b(s0a, (int)(s1a - s0a));
          dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, sbq, dq, 0, 0, 0);
        }
        const int q = it * 64 + 16 * dq_qt + kl;         // lane (query kl of the tile, g): head columns 16 dq_dt + 4 g .. + 3
        if (!(AP_LAB & 8) && q < S) {
          u32x2 w;
          w[0] = pack_bf16x2(dq[0] * ds_scale, dq[1] * ds_scale);
          w[1] = pack_bf16x2(dq[2] * ds_scale, dq[3] * ds_scale);
          *(u32x2*)(dq8 + ((unsigned)q * lddq_b + 2u * (16 * dq_dt + 4 * g))) = w;
        }
      }
      // delta of global iteration J + 1's slot (landed and published by this iteration's dS barrier)
      if (!last || has_next) slot_delta(slot ^ 1);
      if (last) {
        // the next pair's keys (when this pair had a single iteration), V fragments and bias must be there before the switch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- dK, dV of this wave's keys: lane = key, register j of tile dt <-> head column 16 dt + 4 g + j ----
        if (key < S) {
          char* orow = dq8 + ((unsigned)key * lddq_b + 8u * g);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            u32x2 kq, vq;
            kq[0] = pack_bf16x2(dk[dt][0] * ds_scale, dk[dt][1] * ds_scale);
            kq[1] = pack_bf16x2(dk[dt][2] * ds_scale, dk[dt][3] * ds_scale);
            vq[0] = pack_bf16x2(dv[dt][0] * drop_scale, dv[dt][1] * drop_scale);
            vq[1] = pack_bf16x2(dv[dt][2] * drop_scale, dv[dt][3] * drop_scale);
            *(u32x2*)(orow + 2 * (H + 16 * dt)) = kq;
            *(u32x2*)(orow + 2 * (2 * H + 16 * dt)) = vq;
          }
        }
      }
      __syncthreads();   // the (single) dS image and slot J + 1's row constants; at a pair's end also the next pair's keys
      ++J;
    };
    for (int it = 0; it + 1 < niter; ++it) iteration(it, std::false_type{});
    iteration(niter - 1, std::true_type{});
    if (has_next) {
      vf[0] = vfn[0]; vf[1] = vfn[1]; kb2 = key_bias(pn, mvn);
      par ^= 1;
      if (!AP_KF_LDS) read_kf(par, kf);
    }
    p = pn;
  }
}

// delta[b,h,s] = mul * sum_d dO[b,s,h,d] * O[b,s,h,d]   (one wave per token row; 64 d per head = 8 lanes x 8).
// grid = (ceil(S / 4), B): row s of sequence b, which starts at seq_start[b] (or b * S) and has seq_len[b] (or S) rows.
__global__ __launch_bounds__(256) void attn_delta_rows(const bf16_t* __restrict__ d_o, long ld_d, const bf16_t* __restrict__ o,
                                                       long ld_o, float* __restrict__ delta, int B, int S, int nh,
                                                       const int* __restrict__ seq_start, const int* __restrict__ seq_len,
                                                       float mul) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y, s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int len = seq_len ? seq_len[b] : S;
  if (s >= len) return;
  const long row = (seq_start ? (long)seq_start[b] : (long)b * S) + s;
  const int nchunks = nh * 8;  // 8-element chunks per row
  for (int c = lane; c < nchunks; c += 64) {
    const u32x4 x = *(const u32x4*)(d_o + row * ld_d + c * 8);
    const u32x4 y = *(const u32x4*)(o + row * ld_o + c * 8);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += bf16lo(x[i]) * bf16lo(y[i]) + bf16hi(x[i]) * bf16hi(y[i]);
    // reduce the 8 lanes of a head (c>>3 == head): lanes are consecutive
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if ((c & 7) == 0) delta[((long)b * nh + (c >> 3)) * S + s] = acc * mul;   // mul = 1 - p (see the backward's dS)
  }
}

// dq (bf16, columns 0..H-1 of the packed dqkv rows) = round(dq32)
__global__ __launch_bounds__(256) void attn_dq_round(const float* __restrict__ dq32, bf16_t* __restrict__ dqkv, long ld_dqkv,
                                                     long rows, int H) {
  const int cpr = H >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cpr) return;
  const long row = i / cpr;
  const int col = (int)(i - row * cpr) * 8;
  const f32x4 x0 = *(const f32x4*)(dq32 + row * H + col), x1 = *(const f32x4*)(dq32 + row * H + col + 4);
  u32x4 o;
  o[0] = pack_bf16x2(x0[0], x0[1]); o[1] = pack_bf16x2(x0[2], x0[3]);
  o[2] = pack_bf16x2(x1[0], x1[1]); o[3] = pack_bf16x2(x1[2], x1[3]);
  *(u32x4*)(dqkv + row * ld_dqkv + col) = o;
}

// 8 (default): the 8-wave kernel forming delta = rowsum(dO o O) itself; 10: the same kernel behind a separate
// attn_delta_rows pass (the form before round 3's last change: 330-338 against 320 us per launch at B = 256); 4: the 4-wave kernel
// 16 (default): the 16-wave kernel where it serves (one key block, no per-query bias, keep words or no dropout), else as 8
// 17 (default): the persistent, software-pipelined 16-wave kernel where the 16-wave kernel serves; 16: the one-pair-per-workgroup form
static int g_attn_bwd_waves = 17;
void vt_attn_bwd_set_waves(int w) { g_attn_bwd_waves = (w == 4 || w == 10 || w == 8 || w == 16) ? w : 17; }

int vt_attention_bwd_dispatch(const void* qkv, long ld_qkv, const void* dctx, long ld_d, const void* ctx, long ld_ctx,
                              const float* mask, int mask_additive, const float* lse, float* delta_ws, void* dqkv,
                              long ld_dqkv, float* dq32_ws, int B, int S, int nh, int head_size, hipStream_t stream,
                              const DropCfg* drop = nullptr, const int* seq_start = nullptr, const int* seq_len = nullptr,
                              long rows_total = 0, const uint32_t* keep_bits = nullptr) {
  if (mask_additive == 2 && (!mask || seq_start)) return VT_ERR_NULL;   // per-query bias [B, S, S]
  if (!qkv || !dctx || !ctx || !lse || !delta_ws || !dqkv) return VT_ERR_NULL;
  if (head_size != 64) return VT_ERR_UNSUPPORTED;
  const int nkb = (S + 255) / 256;
  if (nkb > 1 && !dq32_ws) return VT_ERR_NULL;
  if (nkb > 65535) return VT_ERR_BAD_SHAPE;
  if (B <= 0 || S <= 0 || nh <= 0 || B > 65535 || nh > 65535) return VT_ERR_BAD_SHAPE;
  if ((ld_qkv % 8) || (ld_d % 8) || (ld_ctx % 8) || (ld_dqkv % 8)) return VT_ERR_BAD_ALIGN;
  if (((uintptr_t)qkv | (uintptr_t)dctx | (uintptr_t)ctx | (uintptr_t)dqkv) & 15) return VT_ERR_BAD_ALIGN;
  static VtLdsAttrOnce attr4;
  if (!attr4.set((const void*)attention_bwd_d64, AB_LDS_BYTES)) return VT_ERR_HIP;
  if ((seq_start == nullptr) != (seq_len == nullptr)) return VT_ERR_NULL;
  if (seq_start && (mask || rows_total <= 0)) return VT_ERR_UNSUPPORTED;   // compacted rows carry no masked keys
  const long rows = seq_start ? rows_total : (long)B * S;
  const float delta_mul = (drop && drop->thresh) ? 1.0f / drop->scale : 1.0f;
  const bool w16 = g_attn_bwd_waves >= 16 && nkb == 1 && mask_additive != 2 && (keep_bits || !(drop && drop->thresh));
  if (g_attn_bwd_waves != 8 && g_attn_bwd_waves < 16)
    hipLaunchKernelGGL(attn_delta_rows, dim3((unsigned)((S + 3) / 4), B), dim3(256), 0, stream, (const bf16_t*)dctx, ld_d,
                       (const bf16_t*)ctx, ld_ctx, delta_ws, B, S, nh, seq_start, seq_len, delta_mul);
  AttnBwdArgs a;
  a.qkv = (const bf16_t*)qkv; a.dctx = (const bf16_t*)dctx; a.mask = mask; a.mask_additive = mask_additive;
  a.lse = lse; a.delta = delta_ws; a.dqkv = (bf16_t*)dqkv;
  a.dq32 = nkb > 1 ? dq32_ws : nullptr;
  if (nkb > 1 && hipMemsetAsync(dq32_ws, 0, (size_t)rows * nh * 64 * sizeof(float), stream) != hipSuccess) return VT_ERR_HIP;
  a.ld_qkv = ld_qkv; a.ld_d = ld_d; a.ld_dqkv = ld_dqkv; a.B = B; a.S = S; a.nh = nh;
  a.seq_start = seq_start; a.seq_len = seq_len;
  a.scale = 1.0f / sqrtf((float)head_size);
  if (drop) a.drop = *drop; else { a.drop.thresh = 0; a.drop.seed = 0; a.drop.scale = 1.0f; }
  a.keep_bits = a.drop.thresh ? keep_bits : nullptr;
  a.ctx = (const bf16_t*)ctx; a.ld_ctx = ld_ctx; a.delta_mul = delta_mul;
  if (w16 && g_attn_bwd_waves == 17 && (long)B * nh <= 65535 && B <= AP_MAX_B) {
    static VtLdsAttrOnce attr17, attr17b;
    if (!attr17.set((const void*)attention_bwd_d64_w16p<false>, AP_LDS_BYTES)) return VT_ERR_HIP;
    if (!attr17b.set((const void*)attention_bwd_d64_w16p<true>, AP_LDS_BYTES)) return VT_ERR_HIP;
    const int cus = vt_device_cus();
    if (cus <= 0) return VT_ERR_HIP;
    const long pairs = (long)B * nh;
    const unsigned grid = (unsigned)(pairs < cus ? pairs : cus);
    if (a.keep_bits) hipLaunchKernelGGL((attention_bwd_d64_w16p<true>), dim3(grid), dim3(1024), AP_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((attention_bwd_d64_w16p<false>), dim3(grid), dim3(1024), AP_LDS_BYTES, stream, a);
  } else if (w16) {
    static VtLdsAttrOnce attr16, attr16b;
    if (!attr16.set((const void*)attention_bwd_d64_w16<false>, AW_LDS_BYTES)) return VT_ERR_HIP;
    if (!attr16b.set((const void*)attention_bwd_d64_w16<true>, AW_LDS_BYTES)) return VT_ERR_HIP;
    if (a.keep_bits) hipLaunchKernelGGL((attention_bwd_d64_w16<true>), dim3(nh, B), dim3(1024), AW_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((attention_bwd_d64_w16<false>), dim3(nh, B), dim3(1024), AW_LDS_BYTES, stream, a);
  } else if (g_attn_bwd_waves == 4) {
    hipLaunchKernelGGL(attention_bwd_d64, dim3(nh, B, nkb), dim3(256), AB_LDS_BYTES, stream, a);
  } else {
    static VtLdsAttrOnce attr8, attr8b, attr8d, attr8bd;
    if (!attr8.set((const void*)attention_bwd_d64_w8<false, false>, AB_LDS_BYTES)) return VT_ERR_HIP;
    if (!attr8b.set((const void*)attention_bwd_d64_w8<true, false>, AB_LDS_BYTES)) return VT_ERR_HIP;
    if (!attr8d.set((const void*)attention_bwd_d64_w8<false, true>, AB_LDS_BYTES)) return VT_ERR_HIP;
    if (!attr8bd.set((const void*)attention_bwd_d64_w8<true, true>, AB_LDS_BYTES)) return VT_ERR_HIP;
    const dim3 grid(nh, B, nkb);
    if (g_attn_bwd_waves == 8 || g_attn_bwd_waves >= 16) {
      if (a.keep_bits) hipLaunchKernelGGL((attention_bwd_d64_w8<true, true>), grid, dim3(512), AB_LDS_BYTES, stream, a);
      else hipLaunchKernelGGL((attention_bwd_d64_w8<false, true>), grid, dim3(512), AB_LDS_BYTES, stream, a);
    } else {
      if (a.keep_bits) hipLaunchKernelGGL((attention_bwd_d64_w8<true, false>), grid, dim3(512), AB_LDS_BYTES, stream, a);
      else hipLaunchKernelGGL((attention_bwd_d64_w8<false, false>), grid, dim3(512), AB_LDS_BYTES, stream, a);
    }
  }
  if (nkb > 1) {
    const long n = rows * (nh * 8);
    hipLaunchKernelGGL(attn_dq_round, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dq32_ws, (bf16_t*)dqkv, ld_dqkv,
                       rows, nh * 64);
  }
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
