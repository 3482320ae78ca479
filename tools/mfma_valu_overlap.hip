// Do a wave's MFMAs and ANOTHER wave's vector instructions on the same SIMD overlap?  (The question under every
// "hide the epilogue under the other group's K loop" design: DESIGN.md 5a.)  One workgroup of eight waves per CU = two waves
// per SIMD; waves 0-3 issue independent v_mfma_f32_16x16x32_bf16 back to back, waves 4-7 independent v_fma_f32, each alone
// and both together, no memory traffic, no barriers.  If the pipes overlap, "both" costs max(a, b); if the MFMA holds the
// SIMD's vector pipeline against other waves as well, it costs a + b.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// mode bit 0: the MFMA waves work; bit 1: the VALU waves work.  valu_per_mfma8: v_fma_f32 per 8 MFMAs of the partner.
__global__ __launch_bounds__(512, 1) void overlap(float* out, unsigned long long* clk, int iters, int mode, int valu_reps) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();   // shader clock
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane & 3); b[i] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)   // eight independent accumulators: back-to-back issue
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 17 && threadIdx.x == 0) clk[0] = __builtin_amdgcn_s_memtime() - t0;
  } else {
    if (!(mode & 2)) return;
    float f[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    float c1 = 0.5f + lane * 1e-3f, c2 = 0.25f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    for (int it = 0; it < iters; ++it) {
      for (int r = 0; r < valu_reps; ++r) {
        asm volatile("v_fma_f32 %0, %8, %9, %0\n\tv_fma_f32 %1, %8, %9, %1\n\tv_fma_f32 %2, %8, %9, %2\n\tv_fma_f32 %3, %8, %9, %3\n\t"
                     "v_fma_f32 %4, %8, %9, %4\n\tv_fma_f32 %5, %8, %9, %5\n\tv_fma_f32 %6, %8, %9, %6\n\tv_fma_f32 %7, %8, %9, %7\n\t"
                     "v_fma_f32 %0, %8, %9, %0\n\tv_fma_f32 %1, %8, %9, %1\n\tv_fma_f32 %2, %8, %9, %2\n\tv_fma_f32 %3, %8, %9, %3\n\t"
                     "v_fma_f32 %4, %8, %9, %4\n\tv_fma_f32 %5, %8, %9, %5\n\tv_fma_f32 %6, %8, %9, %6\n\tv_fma_f32 %7, %8, %9, %7"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                     : "v"(c1), "v"(c2));   // (three different source registers: the same one three times costs bank-conflict cycles)
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7];
    if (blockIdx.x == 17 && threadIdx.x == 256) clk[1] = __builtin_amdgcn_s_memtime() - t0;
  }
}

static unsigned long long* g_clk;
static float run(float* d, int iters, int mode, int valu_reps, double* cyc_m, double* cyc_v) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  overlap<<<256, 512>>>(d, g_clk, iters, mode, valu_reps);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) overlap<<<256, 512>>>(d, g_clk, iters, mode, valu_reps);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  (void)hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
  *cyc_m = (double)h[0]; *cyc_v = (double)h[1];   // shader-clock cycles of one MFMA wave / one VALU wave (last launch)
  return ms * 1e3f / 5;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 512 * 4);
  (void)hipMalloc(&g_clk, 16);
  (void)hipMemset(g_clk, 0, 16);
  const int iters = 20000;   // x 8 MFMAs per wave
  for (int valu_reps = 1; valu_reps <= 4; valu_reps *= 2) {
    double cm, cv, cbm, cbv, x0, x1;
    const float tm = run(d, iters, 1, valu_reps, &cm, &x0), tv = run(d, iters, 2, valu_reps, &x1, &cv);
    const float tb = run(d, iters, 3, valu_reps, &cbm, &cbv);
    const double nm = iters * 8.0, nv = iters * 16.0 * valu_reps;
    printf("%2d v_fma_f32 per 8 MFMAs: alone: MFMA wave %7.1f us = %.1f shader cycles per MFMA, VALU wave %7.1f us = %.1f per v_fma | "
           "together %7.1f us: MFMA wave %.1f cycles per MFMA, VALU wave %.1f per v_fma; wall %.2f x max, %.2f x sum\n",
           16 * valu_reps, tm, cm / nm, tv, cv / nv, tb, cbm / nm, cbv / nv, tb / (tm > tv ? tm : tv), tb / (tm + tv));
  }
  (void)hipFree(d);
  return 0;
}
