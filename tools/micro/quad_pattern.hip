// Does the order in which a quad of lanes covers a row's 128 bytes change the global-load rate?
// Pattern 0 (what OperandX3 does for a k-contiguous operand): lane c of a quad loads floats [8c, 8c+4) then [8c+4, 8c+8)
//   -> one dwordx4 instruction covers 16 B out of every 32 of the line.
// Pattern 1: lane c loads [4c, 4c+4) then [16+4c, 16+4c+4) -> one instruction covers a contiguous 64 B per quad.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/quad_pattern.hip -o /tmp/quad_pattern && /tmp/quad_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PAT>
__global__ void __launch_bounds__(512) rd(const float* __restrict__ A, int64_t ld, int K, int rows, float* out) {
  const int t = threadIdx.x, r = t >> 2, c = t & 3;                  // 128 rows x 4 chunks per K-step of 32
  const int row = (blockIdx.x * 128 + r) % rows;
  const float* p = A + (int64_t)row * ld + (PAT ? 4 * c : 8 * c);
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
#pragma unroll 4
  for (int k = 0; k < K; k += 32) {
    const float4 a = *reinterpret_cast<const float4*>(p + k);
    const float4 b = *reinterpret_cast<const float4*>(p + k + (PAT ? 16 : 4));
    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
  }
  out[blockIdx.x * 512 + t] = s0.x + s0.y + s0.z + s0.w + s1.x + s1.y + s1.z + s1.w;
}

int main() {
  const int rows = 4096, K = 4096;
  const int64_t ld = K;
  float *A, *out;
  hipMalloc(&A, sizeof(float) * rows * ld);
  hipMalloc(&out, sizeof(float) * 4096 * 512);
  hipMemset(A, 0, sizeof(float) * rows * ld);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    for (int pat = 0; pat < 2; ++pat) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) {
          if (pat) hipLaunchKernelGGL(rd<1>, dim3(blocks), dim3(512), 0, 0, A, ld, K, rows, out);
          else hipLaunchKernelGGL(rd<0>, dim3(blocks), dim3(512), 0, 0, A, ld, K, rows, out);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = 10.0 * blocks * 128.0 * K * 4;
      printf("blocks %4d pattern %d: %7.1f us per launch, %6.2f TB/s, %5.1f B/clk/CU at 2.4 GHz\n", blocks, pat,
             best * 100.0, bytes / best / 1e9, bytes / (best * 1e-3) / 256 / 2.4e9 * (256.0 / (blocks < 256 ? blocks : 256)));
    }
  }
  return 0;
}
