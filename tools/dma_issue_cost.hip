// What does ONE LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) cost the wave that issues it
// inside a back-to-back MFMA stream, and does the cost come from the four waves of a workgroup colliding on the CU's one
// vector-memory front end?  (The NT GEMM's K-step carries 16 pieces per wave and 128 MFMAs; removing the pieces takes it
// from 1.40-1.50 to 1.12 us -- DESIGN.md section 4 -- so the piece's issue cost IS the K-step's non-MFMA time.)
//
// One 256-thread workgroup per CU (one wave per SIMD, 128 KiB of LDS so that nothing else is resident), every wave runs
// `iters` blocks of 32 independent v_mfma_f32_16x16x32_bf16 with DMA pieces placed at compile-time slots:
//   mode 0  no pieces                                   (cycles per MFMA: the floor)
//   mode 1  every wave, one piece every P MFMAs, all waves in the SAME slots           (what the GEMM does)
//   mode 2  every wave, one piece every P MFMAs, wave w shifted by w * P / 4 slots     (staggered)
//   mode 3  only wave 0 issues, one piece every P MFMAs                                 (no partner traffic)
//   mode 4  as 1, plus one ds_read_b128 per 4 MFMAs                                     (the K-step's fragment reads)
//   mode 5  as 2, plus one ds_read_b128 per 4 MFMAs
//   mode 6  as 1 with the pieces in bursts: 8 pieces in 16 consecutive MFMA gaps, then 112 MFMAs without (the X phase)
// Source rows: a 64 KiB window per workgroup (L2 hits after the first pass).  Output: shader cycles per MFMA and the
// extra cycles per piece against mode 0.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_issue_cost.hip -o /tmp/dma_issue_cost && /tmp/dma_issue_cost
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define MFMA(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(i) & 15]) : "v"(a), "v"(b))

// one block = 128 MFMAs (a K-step's worth); P = MFMAs per piece; OFF = this wave's slot shift; READS = ds_read_b128 per 4 MFMAs
template <int P, int OFF, bool DMA, bool READS, bool BURST, bool BAR = false>
__device__ __forceinline__ void block(f32x4 (&acc)[16], bf16x8 a, bf16x8 b, __amdgpu_buffer_rsrc_t rs, char* smem, int wave,
                                      int voff, u32x4 (&frag)[4], unsigned raddr) {
#pragma unroll
  for (int i = 0; i < 128; ++i) {
    MFMA(i);
    if (BAR && (i == 20 || i == 50 || i == 88)) __builtin_amdgcn_s_barrier();   // the K-step's three barriers
    if (DMA) {
      if (!BURST) {
        if ((i % P) == OFF) {
          const int k = i / P;   // piece index within the block
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + wave * 16384 + (k & 15) * 1024), 16, voff, (k & 15) * 4096, 0, 0);
        }
      } else {
        if (i >= 22 && i < 38 && !(i & 1)) {
          const int k = (i - 22) >> 1;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + wave * 16384 + k * 1024), 16, voff, k * 4096, 0, 0);
        }
        if (i >= 52 && !(i & 3) && i < 84) {
          const int k = 8 + ((i - 52) >> 2);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + wave * 16384 + k * 1024), 16, voff, k * 4096, 0, 0);
        }
      }
    }
    if (READS && (i & 3) == 1)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[(i >> 2) & 3]) : "v"(raddr), "n"(((i >> 2) & 7) * 2048));
  }
  // the pieces of this block may stay in flight through the next one (the GEMM waits two K-tiles back)
  if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(128 / P > 16 ? 16 : 128 / P) : "memory");
  if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int P, int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, unsigned long long* clk, int iters, size_t nwin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((lane + i) & 3); b[i] = (__bf16)(0.5f + (lane & 1)); }
  // 64 KiB window of this workgroup, 16 KiB per wave; a piece = 8 rows of 128 B, rows 512 B apart (a K-panel's pitch)
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 255) * 65536 + wave * 16384), 0, 16384 + 49152, 0x00020000);
  const int voff = (lane >> 3) * 512 + (lane & 7) * 16;
  u32x4 frag[4];
  const unsigned raddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 65536 + (lane & 15) * 128 + (lane >> 4) * 16;
  constexpr bool dma = MODE != 0, reads = MODE == 4 || MODE == 5 || MODE >= 11, stag = MODE == 2 || MODE == 5;
  constexpr bool burst = MODE == 6 || MODE == 9 || MODE == 10 || MODE == 12, bar = MODE >= 7 && MODE != 13;
  constexpr bool stream = MODE == 8 || MODE == 10 || MODE == 12 || MODE == 13;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
      if (wave == 0) block<P, 0, true, false, false>(acc, a, b, rs, smem, wave, voff, frag, raddr);
      else block<P, 0, false, false, false>(acc, a, b, rs, smem, wave, voff, frag, raddr);
    } else if (stag) {
      switch (wave) {   // wave-uniform: four copies of the stream with shifted slots
        case 0: block<P, 0, dma, reads, false>(acc, a, b, rs, smem, wave, voff, frag, raddr); break;
        case 1: block<P, P / 4, dma, reads, false>(acc, a, b, rs, smem, wave, voff, frag, raddr); break;
        case 2: block<P, 2 * P / 4, dma, reads, false>(acc, a, b, rs, smem, wave, voff, frag, raddr); break;
        default: block<P, 3 * P / 4, dma, reads, false>(acc, a, b, rs, smem, wave, voff, frag, raddr); break;
      }
    } else {
      if (stream) {   // a fresh 64 KiB window per block and workgroup: every piece misses L1 and (first touch) L2
        const size_t win = ((size_t)it * gridDim.x + blockIdx.x) % nwin;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + win * 65536 + wave * 16384), 0, 16384 + 49152, 0x00020000);
      }
      block<P, 0, dma, reads, burst, bar>(acc, a, b, rs, smem, wave, voff, frag, raddr);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (reads) s += __uint_as_float(frag[0][0] ^ frag[1][1] ^ frag[2][2] ^ frag[3][3]) * 1e-30f;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
}

static char* g_src;
static size_t g_nwin;
static float* g_out;
static unsigned long long* g_clk;

template <int P, int MODE>
static double run(int iters, double* us) {
  (void)hipFuncSetAttribute((const void*)k<P, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<P, MODE><<<256, 256, 131072>>>(g_src, g_out, g_clk, iters, g_nwin);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k<P, MODE><<<256, 256, 131072>>>(g_src, g_out, g_clk, iters, g_nwin);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  *us = ms * 1e3 / 3;
  static unsigned long long h[1024];
  (void)hipMemcpy(h, g_clk, sizeof(h), hipMemcpyDeviceToHost);
  // median over workgroups of the slowest wave
  double v[256];
  for (int b = 0; b < 256; ++b) { unsigned long long m = 0; for (int w = 0; w < 4; ++w) m = h[b * 4 + w] > m ? h[b * 4 + w] : m; v[b] = (double)m; }
  for (int i = 0; i < 256; ++i) for (int j = i + 1; j < 256; ++j) if (v[j] < v[i]) { double t = v[i]; v[i] = v[j]; v[j] = t; }
  return v[128];
}

template <int P>
static void sweep(int iters, double base_cyc) {
  const char* names[7] = {"none", "same slots", "staggered", "wave 0 only", "same + reads", "staggered + reads", "GEMM burst"};
  double us;
  const double nm = iters * 128.0;
#define ONE(MODE, pieces_per_block)                                                                          \
  {                                                                                                          \
    const double c = run<P, MODE>(iters, &us);                                                               \
    printf("P=%2d  %-18s %8.1f us  %6.2f cycles/MFMA  %7.1f extra cycles per piece (per issuing wave)\n", P, names[MODE], us, \
           c / nm, (c - base_cyc) / (iters * (double)(pieces_per_block)));                                   \
  }
  ONE(1, 128 / P) ONE(2, 128 / P) ONE(3, 128 / P) ONE(4, 128 / P) ONE(5, 128 / P)
#undef ONE
}

int main() {
  g_nwin = 24576;   // 1.5 GiB of 64 KiB windows (beyond the 256 MiB Infinity Cache)
  (void)hipMalloc(&g_src, g_nwin * 65536 + 65536);
  (void)hipMemset(g_src, 1, g_nwin * 65536 + 65536);
  (void)hipMalloc(&g_out, 256 * 256 * 4);
  (void)hipMalloc(&g_clk, 1024 * 8);
  const int iters = 2000;
  double us;
  for (int warm = 0; warm < 3; ++warm) run<8, 0>(iters, &us);
  const double base = run<8, 0>(iters, &us);
  printf("no pieces: %8.1f us  %6.2f cycles/MFMA (shader clock %.2f GHz)\n", us, base / (iters * 128.0), base / us * 1e-3);
  sweep<8>(iters, base);     // 16 pieces per 128 MFMAs: the GEMM's count, evenly spread
  sweep<4>(iters, base);     // 32 pieces per 128 MFMAs
  sweep<16>(iters, base);    // 8 pieces per 128 MFMAs
  {
    const double c = run<8, 6>(iters, &us);
    printf("GEMM placement (8 pieces in MFMA gaps 22..37 step 2, 8 in 52..83 step 4): %8.1f us  %6.2f cycles/MFMA  %7.1f extra cycles per piece\n",
           us, c / (iters * 128.0), (c - base) / (iters * 16.0));
  }
  {
    const char* nm[14] = {"", "", "", "", "", "", "", "same slots + 3 barriers", "same slots + 3 barriers, streamed source", "GEMM burst + 3 barriers",
                          "GEMM burst + 3 barriers, streamed source", "same slots + barriers + reads", "GEMM burst + barriers + reads, streamed", "same slots, streamed source, no barriers"};
    double c;
#define TWO(MODE) c = run<8, MODE>(iters, &us); printf("%-52s %8.1f us  %6.2f cycles/MFMA  %7.1f extra cycles per piece\n", nm[MODE], us, c / (iters * 128.0), (c - base) / (iters * 16.0));
    TWO(7) TWO(8) TWO(9) TWO(10) TWO(11) TWO(12) TWO(13)
#undef TWO
  }
  {   // reads alone
    const double c0 = run<8, 0>(iters, &us);
    printf("(re-measured floor %6.2f cycles/MFMA)\n", c0 / (iters * 128.0));
  }
  return 0;
}
