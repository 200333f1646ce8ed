// Magnitude records: max |x| over the finite elements of a tensor, left by the kernel that writes the tensor for the
// fp32 GEMM that reads it next (gemm_h2.hip places its operands in fp16's range with a power-of-two scale derived
// from it).  A record is 8 bytes: the fp32 bit pattern of the maximum in the low word (for x >= 0 the order of the
// bit patterns is the order of the values), an epoch tag in the high word, and it is only ever raised — one 64-bit
// integer atomicMax per wave or workgroup, so the result does not depend on the order of arrival and needs no
// reset: a writer of a later epoch (the device-side update counter, see mapx_amax_epoch_source) outranks whatever
// an earlier step left.  Readers take the low word.  Non-finite elements are left out: the scale then comes from
// the finite ones, and an infinity or NaN makes its own output rows non-finite instead of everybody's.
#pragma once
#include "common.h"

namespace mapx {

typedef unsigned long long amax_rec;

const int32_t* amax_epoch_ptr();      // runtime.cpp: the pointer set by mapx_amax_epoch_source (NULL: epoch 0)

__device__ inline uint32_t finite_abs_bits(float v) {
  const uint32_t u = __float_as_uint(v) & 0x7fffffffu;
  return u < 0x7f800000u ? u : 0u;
}
__device__ inline uint32_t amax4(uint32_t m, float a, float b, float c, float d) {
  const uint32_t x = max(finite_abs_bits(a), finite_abs_bits(b)), y = max(finite_abs_bits(c), finite_abs_bits(d));
  return max(m, max(x, y));
}
__device__ inline uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, kWave));
  return v;
}
// one lane of a wave (or one thread of a workgroup) publishes; `ahead`: the record is for epoch + ahead (the
// optimizer writes the weights of the NEXT step)
__device__ inline void amax_publish(amax_rec* rec, uint32_t bits, const int32_t* epoch, int ahead = 0) {
  const unsigned long long tag = epoch ? (unsigned long long)(uint32_t)(*epoch + ahead) : 0ull;
  atomicMax(rec, (tag << 32) | bits);
}
// every lane calls; the wave's maximum goes out once
__device__ inline void amax_publish_wave(amax_rec* rec, uint32_t bits, const int32_t* epoch, int ahead = 0) {
  bits = wave_max_u32(bits);
  if ((threadIdx.x & 63) == 0) amax_publish(rec, bits, epoch, ahead);
}

// Reader: the power-of-two exponent n with amax * 2^n in [2^14, 2^15) — the largest value then rounds to at most
// 32768 < 65504, fp16's largest.  No record, a zero or a non-finite maximum: n = 0.  |n| <= 126 so that 2^n is a
// normal float (a tensor whose maximum is below 2^-112 keeps fewer bits: fp32 training values are 25 orders of
// magnitude above that).
__device__ inline int h2_scale_exp(const float* amax) {
  if (!amax) return 0;
  const uint32_t b = __float_as_uint(*amax) & 0x7fffffffu;
  const int e = (int)(b >> 23);
  if (b == 0u || e == 255) return 0;
  int n = 14 - (e - 127);                // e == 0 (a subnormal maximum): 141, capped below
  return n > 126 ? 126 : (n < -126 ? -126 : n);
}
__device__ inline float pow2f(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }

}  // namespace mapx
