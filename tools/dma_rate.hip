// L2 / HBM -> LDS rate of the operand DMA (buffer_load ... lds, 1 KiB pieces of 8 rows x 128 B, the NT GEMM's fill shape)
// with nothing else in the kernel: what one CU can take in per second, by workgroups per CU, waves per workgroup and
// stage size.  This bounds every tiling of the NT GEMM: a 256 x 256 x 64 K-step stages 64 KiB for 8.4 MFLOP, a
// 256 x 128 x 64 one 48 KiB for 4.2 MFLOP (1.5 x the bytes per FLOP).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
//   mode 0: every workgroup reads the same rows (L2-resident, the W operand's case)
//   mode 1: every workgroup walks a row panel of its own, K-tile by K-tile (the X operand's case: each line once, from HBM)
//   mode 2: half the pieces from a shared block, half from the workgroup's own panel (a GEMM's mix)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int WAVES, int PPW, int NSTAGE>
__global__ __launch_bounds__(WAVES * 64) void dma_rate(const char* src, long stride, long total_bytes, int iters, int mode, int ksteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int STAGE = WAVES * PPW * 1024;
  constexpr int ROWS = STAGE / 128;
  const long own = ((long)blockIdx.x * ROWS * stride) % (total_bytes - (long)ROWS * stride);
  __amdgpu_buffer_rsrc_t rs_shared = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(ROWS * stride), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_own = __builtin_amdgcn_make_buffer_rsrc((void*)(src + own), 0, (int)(ROWS * stride), 0x00020000);
  const int voff = (lane >> 3) * (int)stride + (((lane & 7) ^ ((lane >> 4) & 7)) << 4);
  for (int it = 0; it < iters; ++it) {
    const int st = it % NSTAGE;
    const int kt = it % ksteps;
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      const int piece = wave * PPW + p;
      const int soff = piece * 8 * (int)stride + kt * 128;
      const bool shared = mode == 0 || (mode == 2 && (p & 1));
      if (shared) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_shared, LDS_PTR(smem + st * STAGE + piece * 1024), 16, voff, soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_own, LDS_PTR(smem + st * STAGE + piece * 1024), 16, voff, soff, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 1) * PPW) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int WAVES, int PPW, int NSTAGE>
static void run(const char* d, long stride, long total, int wg_per_cu, int mode, const char* what) {
  constexpr int STAGE = WAVES * PPW * 1024;
  const int lds = STAGE * NSTAGE;
  auto kern = dma_rate<WAVES, PPW, NSTAGE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = 256 * wg_per_cu, iters = 480, ksteps = (int)(stride / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) kern<<<grid, WAVES * 64, lds>>>(d, stride, total, iters, mode, ksteps);
  hipEventRecord(e0);
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) kern<<<grid, WAVES * 64, lds>>>(d, stride, total, iters, mode, ksteps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * iters * STAGE * reps;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-34s mode %d: %d WG/CU x %d waves, stage %3d KiB x %d (%3d KiB LDS): %7.1f GB/s per CU, %6.2f TB/s chip, %.3f us per stage\n",
         what, mode, wg_per_cu, WAVES, STAGE / 1024, NSTAGE, lds / 1024, gbs / 256, gbs / 1e3, ms * 1e3 / reps / iters);
}

int main() {
  const long stride = 1536;                    // K = 768 bf16
  const long total = 1L << 30;
  char* d; hipMalloc(&d, total); hipMemset(d, 1, total);
  for (int mode = 0; mode < 3; ++mode) {
    run<4, 16, 2>(d, stride, total, 1, mode, "v8 shape (256x256x64)");
    run<4, 12, 2>(d, stride, total, 1, mode, "256x128x64, one WG");
    run<4, 6, 3>(d, stride, total, 2, mode, "256x128x32 x 3 stages, two WGs");
    run<4, 12, 1>(d, stride, total, 2, mode, "256x128x64 x 1 stage, two WGs");
    run<8, 8, 2>(d, stride, total, 1, mode, "8 waves, 64 KiB stages");
    run<8, 6, 3>(d, stride, total, 1, mode, "8 waves, 48 KiB x 3");
    run<4, 8, 2>(d, stride, total, 2, mode, "128x128x64 (32 KiB), two WGs");
    run<4, 8, 2>(d, stride, total, 4, mode, "128x128x64 (32 KiB), four WGs");
  }
  hipFree(d);
  return 0;
}
