// Store-only rates of the bf16 epilogue patterns (256x256 tile, 4 waves of 128x128, slab = 16 rows x 64 columns = 128 B per
// row): hipcc --offload-arch=gfx950 -O3 tools/store_pattern_bf16.hip -o /tmp/sp && /tmp/sp
//   A: today's register layout -- lane (j = lane & 15, gq = lane >> 4) owns 32 B of row j: two stores of 16 B, i.e. every
//      instruction writes 16-byte pieces at a 32-byte stride
//   C: after a permlane16_swap + permlane32_swap of the two pieces: an instruction writes 64 contiguous bytes of each of 16 rows
//   D: 128 contiguous bytes of each of 8 rows per instruction (would need a row exchange as well)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int PAT>
__global__ __launch_bounds__(256) void pat(unsigned short* out, long ld, long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tn = ld / 256;
  const long m0 = (blockIdx.x / tn) * 256, n0 = (blockIdx.x % tn) * 256;
  const int wm = wave >> 1, wn = wave & 1;
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (int h = 0; h < 16; ++h) {
    const long r0 = m0 + 128 * wm + 16 * (h >> 1), c0 = n0 + 128 * wn + 64 * (h & 1);
    if (PAT == 0) {
      const long row = r0 + (lane & 15);
      if (row < rows) { u32x4* p = (u32x4*)(out + row * ld + c0 + 16 * (lane >> 4)); p[0] = v; p[1] = v; }
    } else if (PAT == 1) {
      const long row = r0 + (lane & 15);
      if (row < rows) { u32x4* p = (u32x4*)(out + row * ld + c0 + 8 * (lane >> 4)); p[0] = v; p[4] = v; }
    } else {
      for (int i = 0; i < 2; ++i) {
        const long row = r0 + 8 * i + (lane >> 3);
        if (row < rows) *(u32x4*)(out + row * ld + c0 + 8 * (lane & 7)) = v;
      }
    }
  }
}
int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long M : {14592L, 58368L}) {
    const long N = 768;
    unsigned short* d; hipMalloc(&d, (size_t)M * N * 2);
    const int tiles = (int)(((M + 255) / 256) * (N / 256));
    for (int p = 0; p < 3; ++p) {
      auto run = [&]() { if (p == 0) pat<0><<<tiles, 256>>>(d, N, M); else if (p == 1) pat<1><<<tiles, 256>>>(d, N, M); else pat<2><<<tiles, 256>>>(d, N, M); };
      for (int rep = 0; rep < 3; ++rep) run();
      hipEventRecord(e0);
      for (int rep = 0; rep < 20; ++rep) run();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("M = %ld (%d workgroups): pattern %c: %.1f us per pass, %.2f TB/s\n", M, tiles, "ACD"[p], ms / 20 * 1e3, (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
    }
    hipFree(d);
  }
  return 0;
}
