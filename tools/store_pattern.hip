#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
// A: the GEMM register epilogue's pattern: lane (j = lane&15, gq = lane>>4) writes 64 B of row j at column gq*16, as 4 stores of 16 B
__global__ __launch_bounds__(256) void pat_a(float* out, long ld, long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tile = blockIdx.x;   // 256x256 tile
  const long tn = ld / 256;
  const long m0 = (tile / tn) * 256, n0 = (tile % tn) * 256;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (int h = 0; h < 16; ++h) {
    const long r0 = m0 + 128 * wm + 16 * (h >> 1), c0 = n0 + 128 * wn + 64 * (h & 1);
    const long row = r0 + (lane & 15);
    if (row < rows) {
      f32x4* cp = (f32x4*)(out + row * ld + c0 + 16 * (lane >> 4));
      for (int i = 0; i < 4; ++i) cp[i] = v;
    }
  }
}
// B: after a transpose: each store instruction writes 4 rows x 256 contiguous bytes (lane -> row lane>>4, 16 B chunk lane&15)
__global__ __launch_bounds__(256) void pat_b(float* out, long ld, long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tile = blockIdx.x;
  const long tn = ld / 256;
  const long m0 = (tile / tn) * 256, n0 = (tile % tn) * 256;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (int h = 0; h < 16; ++h) {
    const long r0 = m0 + 128 * wm + 16 * (h >> 1), c0 = n0 + 128 * wn + 64 * (h & 1);
    for (int i = 0; i < 4; ++i) {
      const long row = r0 + 4 * i + (lane >> 4);
      if (row < rows) *(f32x4*)(out + row * ld + c0 + 4 * (lane & 15)) = v;
    }
  }
}
int main() {
  const long M = 4272, N = 30528 / 256 * 256;   // 30464 columns (119 tiles)
  float* d; hipMalloc(&d, (size_t)M * 30528 * 4);
  const int tiles = ((M + 255) / 256) * (N / 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pat = 0; pat < 2; ++pat) {
    for (int rep = 0; rep < 3; ++rep) { if (pat == 0) pat_a<<<tiles, 256>>>(d, 30528, M); else pat_b<<<tiles, 256>>>(d, 30528, M); }
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) { if (pat == 0) pat_a<<<tiles, 256>>>(d, 30528, M); else pat_b<<<tiles, 256>>>(d, 30528, M); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("pattern %c: %.1f us per pass, %.2f TB/s\n", pat ? 'B' : 'A', ms / 10 * 1e3, (double)M * N * 4 / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
