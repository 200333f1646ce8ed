// Issue cost of candidate instructions for the operand cut of gemm_h2.hip beside v_mfma_f32_32x32x16_f16, one wave per
// SIMD: per loop iteration one MFMA + K copies of the instruction (independent destinations), cycles per iteration
// by s_memtime.   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_beside_mfma.hip -o /tmp/vbm && /tmp/vbm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define REP4(x) x x x x
template <int OP, int K>
__global__ void __launch_bounds__(256) kern(float* out, long long* cyc, int iters, float s) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i - 3.f); }
  f16v acc = {0}, acc2 = {0}, acc3 = {0}, acc4 = {0};
  float x0 = threadIdx.x * 1.25f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f;
  unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
  float f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
   for (int rep = 0; rep < 4; ++rep) {
    if (rep == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (rep == 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
    if (rep == 2) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
    if (rep == 3) acc4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc4, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#define ONE(D, F, X)                                                                                       \
    if (OP == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(D) : "v"(X), "s"(s));                 \
    if (OP == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(F) : "v"(X), "s"(s), "v"(D)); \
    if (OP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(D) : "v"(X), "v"(X));                   \
    if (OP == 3) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(F) : "v"(D));                                  \
    if (OP == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(F) : "v"(D)); \
    if (OP == 5) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(F) : "v"(X), "v"(D));                        \
    if (OP == 6) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(F) : "s"(s), "v"(X));                          \
    if (OP == 7) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(F) : "v"(X), "s"(s), "v"(X));              \
    if (OP == 8) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(D) : "v"(X), "v"(X));                \
    if (OP == 9) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(D) : "v"(X), "s"(s));                 \
    if (OP == 10) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(F) : "v"(X), "v"(F));                         \
    if (OP == 11) asm volatile("v_and_b32 %0, 0xffffe000, %1" : "=v"(D) : "v"(X));                         \
    if (OP == 12) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(D) : "v"(X), "v"(X), "s"(0x07060302u));   \
    if (OP == 13) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double*)&F##d) : "v"(*(double*)&F##d), "v"(*(double*)&F##d));
    double f0d = 1, f1d = 1, f2d = 1, f3d = 1, f4d = 1, f5d = 1;
    if (K > 0) { ONE(d0, f0, x0) }
    if (K > 1) { ONE(d1, f1, x1) }
    if (K > 2) { ONE(d2, f2, x2) }
    if (K > 3) { ONE(d3, f3, x3) }
    if (K > 4) { ONE(d4, f4, x4) }
    if (K > 5) { ONE(d5, f5, x5) }
    if (K > 6) { ONE(d0, f0, x1) }
    if (K > 7) { ONE(d1, f1, x2) }
    __builtin_amdgcn_sched_barrier(0);
   }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i] + acc2[i] + acc3[i] + acc4[i];
  out[blockIdx.x * 256 + threadIdx.x] = r + f0 + f1 + f2 + f3 + f4 + f5 + d0 + d1 + d2 + d3 + d4 + d5;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int K>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000, blocks = 256;
  kern<OP, K><<<blocks, 256>>>(out, cyc, iters, 0.5f);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  kern<OP, K><<<blocks, 256>>>(out, cyc, iters, 0.5f);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < blocks; ++i) m += h[i];
  printf("%-24s K=%d  %7.1f ns/MFMA  %7.1f memtime-ticks/MFMA\n", name, K, ms * 1e6 / iters / 4, m / blocks / iters / 4);
}
#define ROW(OP, name) run<OP, 0>(name, out, cyc); run<OP, 2>(name, out, cyc); run<OP, 4>(name, out, cyc); run<OP, 6>(name, out, cyc); run<OP, 8>(name, out, cyc);
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  ROW(0, "v_fma_mixlo_f16") ROW(9, "v_fma_mixhi_f16") ROW(1, "v_fma_mix_f32") ROW(2, "v_cvt_pk_f16_f32") ROW(8, "v_cvt_pkrtz_f16_f32")
  ROW(3, "v_cvt_f32_f16") ROW(4, "v_cvt_f32_f16_sdwa") ROW(5, "v_ldexp_f32") ROW(6, "v_mul_f32") ROW(7, "v_fma_f32") ROW(10, "v_sub_f32")
  ROW(11, "v_and_b32") ROW(12, "v_perm_b32") ROW(13, "v_pk_mul_f32")
  return 0;
}
