// Shared helpers for the mapx gfx950 kernels: status codes, launch checks, Philox RNG.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define MAPX_OK 0
#define MAPX_EINVAL (-1)
#define MAPX_EHIP (-2)
#define MAPX_EWORKSPACE (-3)

namespace mapx {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MAPX_EHIP;
  }
  return MAPX_OK;
}

#define MAPX_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::mapx::set_error(__VA_ARGS__);    \
      return MAPX_EINVAL;                \
    }                                    \
  } while (0)

#define MAPX_HIP(call)                                                   \
  do {                                                                   \
    hipError_t e_ = (call);                                              \
    if (e_ != hipSuccess) {                                              \
      ::mapx::set_error("%s: %s", #call, hipGetErrorString(e_));         \
      return MAPX_EHIP;                                                  \
    }                                                                    \
  } while (0)

constexpr int kWave = 64;

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound kernels: cap the grid at 256 CUs x 8 blocks and grid-stride the rest.
inline int grid_for(int64_t work_items, int block, int max_blocks = 2048) {
  int64_t g = ceil_div(work_items, block);
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG: (seed, offset, element index) -> 4 x u32, no state in memory, so a
// kernel replay (hipGraph) with a bumped offset word gives a fresh, reproducible stream.
struct Philox4 {
  uint32_t x, y, z, w;
};

__device__ inline Philox4 philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// u32 -> uniform integer in [0, n) by multiply-shift (bias < n / 2^32).
__device__ inline uint32_t bounded(uint32_t r, uint32_t n) {
  return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32);
}
// u32 -> uniform float in [0, 1) with 24 random bits.
__device__ inline float unit_float(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// Wave64 butterfly sum over `width` lanes (width a power of two <= 64).
template <int WIDTH>
__device__ inline float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

}  // namespace mapx
