// Magnitude records: max |x| over the finite elements of a tensor, left by the kernel that writes the tensor for the
// fp32 GEMM that reads it next (gemm_h2.hip places its operands in fp16's range with a power-of-two scale derived
// from it).  A record is 64 slots of 8 bytes (512 B): in each slot the fp32 bit pattern of a maximum in the low word
// (for x >= 0 the order of the bit patterns is the order of the values) and an epoch tag in the high word.  Slots are
// only ever raised — one 64-bit integer atomicMax per wave, into the slot the wave's number picks — so the result
// does not depend on the order of arrival and needs no reset: a writer of a later epoch (mapx_amax_epoch_source; the
// step counter) outranks whatever an earlier step left.  Readers take the maximum over the 64 low words.  Every wave
// of a writing launch publishes (zeros too), and a record is written by the same launch geometry every epoch (a
// captured step replays the same grids) or is zeroed first (eager launches: mapx.ops.amax_record): so every slot
// holds either this epoch's value or nothing, and the maximum is exact — the eager and the replayed step scale alike.
// Why 64 slots: same-address atomics serialise in L2 at ~12 ns each — 1 k waves of a GEMM epilogue on ONE word cost
// 12 us, the 5.9 k waves of the embedding gather 69 us (measured, round 4); spread over 64 words they cost < 1 us.
// Non-finite elements are left out: the scale then comes from the finite ones, and an infinity or NaN makes its own
// output rows non-finite instead of everybody's.
#pragma once
#include "common.h"

namespace mapx {

typedef unsigned long long amax_rec;
constexpr int kAmaxSlots = 64;

const int32_t* amax_epoch_ptr();      // runtime.cpp: the word set by mapx_amax_epoch_source (NULL: epoch 0)

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
// the slot this wave raises
__device__ inline int amax_slot() {
  return (int)(((blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) * (blockDim.x >> 6) +
                (threadIdx.x >> 6)) & (kAmaxSlots - 1));
}
// one lane publishes into `slot`; `ahead`: the record is for epoch + ahead (the optimizer writes the weights of
// the NEXT step)
__device__ inline void amax_publish(amax_rec* rec, uint32_t bits, const int32_t* epoch, int ahead = 0, int slot = 0) {
  const unsigned long long tag = epoch ? (unsigned long long)(uint32_t)(*epoch + ahead) : 0ull;
  atomicMax(rec + (slot & (kAmaxSlots - 1)), (tag << 32) | bits);
}
// every lane of the wave calls; the wave's maximum goes out once
__device__ inline void amax_publish_wave(amax_rec* rec, uint32_t bits, const int32_t* epoch, int ahead = 0) {
  bits = wave_max_u32(bits);
  if ((threadIdx.x & 63) == 0) amax_publish(rec, bits, epoch, ahead, amax_slot());
}

// every thread of the WORKGROUP calls (it meets at a barrier); the workgroup's maximum goes out once: the form for
// grids of thousands of workgroups (the embedding gather: 5.9 k waves on 64 slots still cost 15 us)
__device__ inline void amax_publish_block(amax_rec* rec, uint32_t bits, const int32_t* epoch, int ahead = 0) {
  __shared__ uint32_t wave_bits[16];
  bits = wave_max_u32(bits);
  if ((threadIdx.x & 63) == 0) wave_bits[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = wave_bits[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = max(m, wave_bits[w]);
    amax_publish(rec, m, epoch, ahead, (int)(blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y));
  }
  __syncthreads();            // (a second record published by the same kernel reuses wave_bits)
}

// Reader (every lane of a wave calls): the bit pattern of the tensor's maximum
__device__ inline uint32_t amax_read(const float* rec) {
  if (!rec) return 0u;
  return wave_max_u32(reinterpret_cast<const uint32_t*>(rec)[2 * (threadIdx.x & 63)] & 0x7fffffffu);
}
// ... and the power-of-two exponent n with amax * 2^n in [2^14, 2^15) — the largest value then rounds to at most
// 32768 < 65504, fp16's largest.  No record, a zero or a non-finite maximum: n = 0.  |n| <= 126 so that 2^n is a
// normal float (a tensor whose maximum is below 2^-112 keeps fewer bits: fp32 training values are 25 orders of
// magnitude above that).
__device__ inline int h2_scale_exp(const float* rec) {
  const uint32_t b = amax_read(rec);
  const int e = (int)(b >> 23);
  if (b == 0u || e == 255) return 0;
  int n = 14 - (e - 127);                // e == 0 (a subnormal maximum): 141, capped below
  return n > 126 ? 126 : (n < -126 ? -126 : n);
}
__device__ inline float pow2f(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }

}  // namespace mapx
