// fp32 MFMA issue-rate calibration: WAVES waves per block of pure v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // pseudo-random operands (full-range mantissas, mixed signs): the sustained clock depends on data
  float av[4], bv[4];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
  for (int i = 0; i < 4; ++i) {
    h = h * 1664525u + 1013904223u; av[i] = ((int)h) * (a / 2147483648.0f);
    h = h * 1664525u + 1013904223u; bv[i] = ((int)h) * (b / 2147483648.0f);
  }
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(i + u) & 3], bv[(i + 2 * u + 1) & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    int grid = 256 * blocks_per_cu, iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)grid * 4 * iters * 4 * 4096.0;   // waves * iters * NACC * flop per mfma
      printf("blocks/CU %d: %.2f ms  %.1f TFLOP/s  (=> %.2f GHz at 64 FLOP/clk/SIMD)\n", blocks_per_cu, ms,
             flop / ms / 1e9, flop / ms / 1e9 * 1e12 / (1024.0 * 64) / 1e9);
    }
  }
  return 0;
}
