// Magnitude records of tensors no mapx kernel produced (amax.h): one pass over a [rows, cols] fp32 matrix with row
// stride ld.  The tensors the step's own kernels write carry their record from the kernel that wrote them.
#include "../../include/mapx_hip.h"
#include "amax.h"

namespace mapx {

__global__ void __launch_bounds__(256) amax_f32_kernel(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                       amax_rec* __restrict__ rec, const int32_t* __restrict__ epoch,
                                                       int ahead) {
  uint32_t m = 0;
  if (ld == cols && cols % 4 == 0 && (uintptr_t)x % 16 == 0) {
    const int64_t n4 = rows * cols / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      m = amax4(m, v.x, v.y, v.z, v.w);
    }
  } else {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
      m = max(m, finite_abs_bits(x[(i / cols) * ld + i % cols]));
  }
  amax_publish_block(rec, m, epoch, ahead);
}

}  // namespace mapx

extern "C" int mapx_amax_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, void* record, int reset,
                             hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(record && rows >= 0 && cols >= 0 && ld >= cols, "amax_f32: bad arguments");
  if (reset) MAPX_HIP(hipMemsetAsync(record, 0, kAmaxSlots * sizeof(amax_rec), stream));
  if (rows == 0 || cols == 0) return MAPX_OK;
  MAPX_REQUIRE(x, "amax_f32: null tensor");
  hipLaunchKernelGGL(amax_f32_kernel, dim3(grid_for(rows * cols / 4 + 1, 256, 1024)), dim3(256), 0, stream, x, rows, cols,
                     ld, static_cast<amax_rec*>(record), amax_epoch_ptr(), 0);
  return check_launch("amax_f32");
}
