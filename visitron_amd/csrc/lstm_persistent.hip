// The LSTM recurrence of OscarEncoder.forward (tasks/viewpoint_select/agent_models.py:285-301: nn.LSTM over the packed
// trunk output) as ONE persistent launch instead of one launch per position.
//
// The step-per-launch form (rollout.hip: lstm_step_kernel) is bound by the 7 us a dependent launch costs, 511 times per
// instruction.  Here hs / 16 workgroups stay resident for the whole sequence; workgroup j owns hidden units
// 16 j .. 16 j + 15: their 64 rows of W_hh (4 gates) sit in its REGISTERS as MFMA A fragments for all T steps, their cell
// and hidden state in registers too.  Per step a workgroup needs every unit's previous hidden state (the B operand):
// the workgroups exchange it through a small bf16 buffer in global memory, hand-off form per MI355X_MICROARCH.md
// (Valid forms, table row 3):
//   producer  16-byte write-through (sc1) stores of the workgroup's slice, each 128-byte line written whole by one store
//             instruction of one wave (exchange layout [workgroup][batch][16 units]: 2 KiB contiguous per workgroup);
//             every storing wave drains (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds 1 to a counter with an
//             agent-scope atomic;
//   consumer  ONE lane polls the counter with relaxed agent-scope loads (sc1) until epoch * workgroups is reached
//             (bounded: a timeout sets an error word and every workgroup leaves the loop); workgroup barrier; then
//             every load of the exchanged bytes is a 16-byte sc1 load to registers.
// Two exchange buffers alternate: a workgroup can only start step i+1 after all have published step i, i.e. after all
// have finished READING the buffer step i+1 overwrites.  The arithmetic is the step kernel's: the fp32 state is rounded
// to bf16 for the recurrent product (here when it is published instead of when it is loaded: same values), gates and
// state update in fp32, packed-sequence rule (a row past its length keeps its state, its output position is zero).
#include "common.hpp"
#include "rollout_args.hpp"

struct LstmPersistArgs {
  const float* xproj; long ldx_b, ldx_t;   // padded layout: row (b, t) at xproj + b * ldx_b + t * ldx_t (xrow_start == null)
  const int* xrow_start;                   // compacted layout: row (b, t) at xproj + (xrow_start[b] + t) * ldx_t
  float* h; float* c;                      // [B, hs] fp32: initial state in, final state out
  const bf16_t* w_hh;                      // [4 hs, hs]
  const int* lengths;                      // [B] or null
  float* seq_out; long lds_b, lds_t;       // optional [B, T, hs] view
  bf16_t* xchg;                            // 2 x [hs / 16][Bp][16] bf16 exchange buffers (Bp = batch rounded up to 16)
  unsigned* sync;                          // [0] arrival counter, [1] timeout flag (zeroed by the launcher)
  int B, hs, T, reverse;
};

__device__ __forceinline__ float lp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float lp_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}
__device__ __forceinline__ u32x4 lp_load16_sc1(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void lp_store16_sc1(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// KS = 32-wide K steps per wave (hs = 128 KS); NBT = batch tiles of 16 (B <= 16 NBT)
template <int KS, int NBT>
__global__ __launch_bounds__(256, 1) void lstm_persistent_kernel(LstmPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lp_smem[];
  // [wave][batch tile][gate][lane]: partial gate sums | the step's new hidden values of this workgroup's 16 units, by
  // batch row | the poll's verdict
  f32x4 (*part)[NBT][4][64] = (f32x4 (*)[NBT][4][64])lp_smem;
  float (*hstage)[16] = (float (*)[16])(lp_smem + 4 * NBT * 4 * 64 * 16);
  int& s_flag = *(int*)(lp_smem + 4 * NBT * 4 * 64 * 16 + NBT * 16 * 16 * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x, nwg = gridDim.x, h0 = j * 16;
  const int hs = a.hs, Bp = NBT * 16;
  const int kq = hs >> 2;
  const int kl = (lane >> 4) * 8;
  const long xslab = (long)nwg * Bp * 16;      // elements per exchange buffer

  // W_hh fragments: gate g, k-step ks: row g*hs + h0 + (lane&15), k = wave*kq + 32 ks + kl .. +7
  bf16x8 wf[4][KS];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[g][ks] = *(const bf16x8*)(a.w_hh + ((long)g * hs + h0 + (lane & 15)) * hs + wave * kq + 32 * ks + kl);

  // this thread's elements: hidden = h0 + 4 (lane / 16) + wave, batch = 16 bt + lane % 16
  const int hl = 4 * (lane >> 4) + wave, hid = h0 + hl;
  float hreg[NBT], creg[NBT];
  int len[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    const int b = 16 * bt + (lane & 15);
    const bool v = b < a.B;
    hreg[bt] = v ? a.h[(long)b * hs + hid] : 0.f;
    creg[bt] = v ? a.c[(long)b * hs + hid] : 0.f;
    len[bt] = v ? (a.lengths ? a.lengths[b] : a.T) : 0;
  }

  // publish a slice: hstage (fp32, [batch][16 units]) -> bf16 -> xchg[buf][j][batch][16]; threads 0 .. 2 Bp - 1, 16 B each
  auto publish = [&](int buf) {
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) hstage[16 * bt + (lane & 15)][hl] = hreg[bt];
    __syncthreads();
    if (tid < 2 * Bp) {
      const int b = tid >> 1, half = tid & 1;
      const float* s = &hstage[b][8 * half];
      u32x4 w;
      w[0] = pack_bf16x2(s[0], s[1]); w[1] = pack_bf16x2(s[2], s[3]);
      w[2] = pack_bf16x2(s[4], s[5]); w[3] = pack_bf16x2(s[6], s[7]);
      lp_store16_sc1(a.xchg + (long)buf * xslab + ((long)j * Bp + b) * 16 + 8 * half, w);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the workgroup signals
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  publish(0);   // epoch 1: the initial state

  bool finished = true;
  for (int i = 0; i < a.T; ++i) {
    const int t = a.reverse ? a.T - 1 - i : i;
    // this step's input projections do not depend on the exchange: requested before the wait, used after it
    float xg[NBT][4];
    bool act[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      const int b = 16 * bt + (lane & 15);
      act[bt] = b < a.B && t < len[bt];
      const long xr = a.xrow_start ? ((long)(act[bt] ? a.xrow_start[b] : 0) + t) * a.ldx_t : (long)b * a.ldx_b + (long)t * a.ldx_t;
#pragma unroll
      for (int g = 0; g < 4; ++g) xg[bt][g] = act[bt] ? a.xproj[xr + (long)g * hs + hid] : 0.f;
    }
    // ---- wait until every workgroup has published epoch i + 1 (the state before this step) ----
    if (tid == 0) {
      const unsigned want = (unsigned)(i + 1) * (unsigned)nwg;
      int ok = 1;
      unsigned spins = 0;
      while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 20) || __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tell the others, leave
          ok = 0;
          break;
        }
      }
      s_flag = ok;
    }
    __syncthreads();
    if (!s_flag) { finished = false; break; }   // uniform

    // ---- gates' recurrent part: W_hh slice (registers) x h_prev (exchange buffer, sc1 loads) ----
    const bf16_t* xb = a.xchg + (long)(i & 1) * xslab;
    f32x4 acc[NBT][4];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[bt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 hb[KS][NBT];
    // (the asm loads below are invisible to the compiler's own vmcnt bookkeeping: nothing of its may be in flight
    // around them -- drained before, everything drained after; all KS x NBT loads go out together: one round trip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = wave * kq + 32 * ks + kl;          // 8 consecutive units: inside one workgroup's 16
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt)
        hb[ks][bt] = lp_load16_sc1(xb + ((long)(k >> 4) * Bp + 16 * bt + (lane & 15)) * 16 + (k & 15));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) {
        asm volatile("" : "+v"(hb[ks][bt]));
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[bt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][ks], __builtin_bit_cast(bf16x8, hb[ks][bt]), acc[bt][g], 0, 0, 0);
      }
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
      for (int g = 0; g < 4; ++g) part[wave][bt][g][lane] = acc[bt][g];
    __syncthreads();

    // ---- gate arithmetic for this thread's elements ----
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      const int b = 16 * bt + (lane & 15);
      if (b >= a.B) continue;
      const bool active = act[bt];
      float hn = hreg[bt];
      if (active) {
        float gate[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float s = xg[bt][g];
#pragma unroll
          for (int w = 0; w < 4; ++w) s += ((const float*)&part[w][bt][g][lane])[wave];
          gate[g] = s;
        }
        const float cn = lp_sigmoid(gate[1]) * creg[bt] + lp_sigmoid(gate[0]) * lp_tanh(gate[2]);
        hn = lp_sigmoid(gate[3]) * lp_tanh(cn);
        creg[bt] = cn;
        hreg[bt] = hn;
      }
      if (a.seq_out) a.seq_out[(long)b * a.lds_b + (long)t * a.lds_t + hid] = active ? hn : 0.f;
    }
    if (i + 1 < a.T) publish((i + 1) & 1);   // epoch i + 2 (its barriers also fence the LDS arrays for the next step)
  }

  // final state (fp32); after a timeout the caller's state is left as it was (the launcher's caller then takes the
  // step-per-launch form from it)
  if (!finished) return;
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    const int b = 16 * bt + (lane & 15);
    if (b < a.B) { a.h[(long)b * hs + hid] = hreg[bt]; a.c[(long)b * hs + hid] = creg[bt]; }
  }
}

// workspace bytes the caller hands in: two exchange buffers + the sync words
long vt_lstm_persistent_ws_bytes(int B, int hs) {
  const long Bp = (B + 15) / 16 * 16;
  return 256 + 2L * (hs / 16) * Bp * 16 * 2;
}

template <int KS, int NBT>
static int launch_lp2(const LstmPersistArgs& a, hipStream_t stream) {
  const int lds = 4 * NBT * 4 * 64 * 16 + NBT * 16 * 16 * 4 + 16;
  if (hipFuncSetAttribute((const void*)lstm_persistent_kernel<KS, NBT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return VT_ERR_HIP;
  hipLaunchKernelGGL((lstm_persistent_kernel<KS, NBT>), dim3(a.hs / 16), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_HIP;
}
template <int KS>
static int launch_lp(const LstmPersistArgs& a, int nbt, hipStream_t stream) {
  switch (nbt) {
    case 1: return launch_lp2<KS, 1>(a, stream);
    case 2: return launch_lp2<KS, 2>(a, stream);
    case 3: return launch_lp2<KS, 3>(a, stream);
    case 4: return launch_lp2<KS, 4>(a, stream);
    default: return VT_ERR_UNSUPPORTED;
  }
}

// Returns VT_ERR_UNSUPPORTED when the shape is outside what the persistent form serves (B > 64, hs not 128 / 256 / 512 /
// 1024, more workgroups than compute units): the caller then issues the step launches.
int vt_lstm_persistent_dispatch(LstmPersistArgs a, void* ws, long ws_bytes, hipStream_t stream) {
  if (!a.xproj || !a.h || !a.c || !a.w_hh || !ws) return VT_ERR_NULL;
  if (a.B <= 0 || a.T <= 0 || a.hs <= 0) return VT_ERR_BAD_SHAPE;
  if (a.xrow_start && !a.lengths) return VT_ERR_NULL;
  if (a.B > 64 || (a.hs != 128 && a.hs != 256 && a.hs != 512 && a.hs != 1024)) return VT_ERR_UNSUPPORTED;
  if (a.hs / 16 > vt_device_cus()) return VT_ERR_UNSUPPORTED;   // every workgroup must be resident at once
  if (ws_bytes < vt_lstm_persistent_ws_bytes(a.B, a.hs)) return VT_ERR_BAD_SHAPE;
  if ((((uintptr_t)a.w_hh) | ((uintptr_t)ws)) & 15) return VT_ERR_BAD_ALIGN;
  a.sync = (unsigned*)ws;
  a.xchg = (bf16_t*)((char*)ws + 256);
  if (hipMemsetAsync(ws, 0, 256, stream) != hipSuccess) return VT_ERR_HIP;   // counter, timeout word: zero before EVERY launch
  const int nbt = (a.B + 15) / 16;
  switch (a.hs) {
    case 128: return launch_lp<1>(a, nbt, stream);
    case 256: return launch_lp<2>(a, nbt, stream);
    case 512: return launch_lp<4>(a, nbt, stream);
    default: return launch_lp<8>(a, nbt, stream);
  }
}
