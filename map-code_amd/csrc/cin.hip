// Compressed Interaction Network pieces (xDeepFM; reference code/layers.py:696-721).
//   X_{i+1}[b,o,:] = bias[o] + sum_{h,m} W[o, h*H_i + m] * X_0[b,h,:] * X_i[b,m,:]
// is a GEMM over rows r = (b, d) once every activation is kept "embedding-major"
// (Xt[b, d, m] = X[b, m, d]): the Hadamard row  had[r, h*H + m] = X0t[r, h] * Xt_i[r, m]  is an
// outer product of two short contiguous vectors, the 1x1 convolution is  Xt_{i+1} = had W^T + b
// (mapx_gemm_f32, output already embedding-major for the next layer), and sum-pooling over the
// embedding axis adds the E rows of a sample.  The kernels here are the glue around that GEMM:
// the [B,F,E] <-> [B,E,F] transpose, the outer product and its backward, the pooling and its
// backward.  All HBM-bound streams; the Hadamard matrix is materialised (R x F*H floats).
#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

// out[b, c, r] = x[b, r, c]
__global__ void __launch_bounds__(256) transpose_batched_kernel(const float* __restrict__ x, int64_t B,
                                                                int R, int C, float* __restrict__ out) {
  const int64_t per = (int64_t)R * C, n = B * per;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / per;
    const int o = (int)(t - b * per);
    const int c = o / R, r = o - c * R;
    out[t] = x[b * per + (int64_t)r * C + c];
  }
}

// had[r * ld + h*H + m] = x0t[r, h] * xi[r, m]; columns F*H .. ld - 1 (the padding that gives the 1x1 convolution's GEMM
// 16-byte rows and a K that is a multiple of 8: F*H = 529 or 1150 left it the scalar operand path) are zeroed.
// A thread owns up to four COLUMNS (k = t, t + 256, ...): their (h, m) are worked out once, the rows are walked
// without a division (the element-per-thread form spent its time on 64-bit t / ld, k / H: 159 us for 302 MB).
__global__ void __launch_bounds__(256) cin_outer_fwd_kernel(const float* __restrict__ x0t, int F,
                                                            const float* __restrict__ xi, int H, int64_t R,
                                                            float* __restrict__ had, int64_t ld) {
  constexpr int CPT = 8;                                  // columns per thread; blockIdx.y: groups of 2048 columns
  const int K = F * H, kb = blockIdx.y * (256 * CPT);
  int hh[CPT], mm[CPT];
  bool in[CPT], live[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int k = kb + threadIdx.x + 256 * j;
    live[j] = k < ld;
    in[j] = k < K;
    hh[j] = in[j] ? k / H : 0;
    mm[j] = in[j] ? k - hh[j] * H : 0;
  }
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    const float* __restrict__ a = x0t + r * F;
    const float* __restrict__ c = xi + r * H;
    float* __restrict__ o = had + r * ld;
#pragma unroll
    for (int j = 0; j < CPT; ++j)
      if (live[j]) o[kb + threadIdx.x + 256 * j] = in[j] ? a[hh[j]] * c[mm[j]] : 0.f;
  }
}

// one wave per row r, ONE pass over the row of dhad:  lane m (and m + 64 ...) holds its column of all F rows h,
//   dxi[r, m] = sum_h dhad[r,h,m] x0t[r,h]  per lane;   dx0t[r, h] (+)= sum_m dhad[r,h,m] xi[r,m]  by a wave reduction
// per h, fixed order.  (The two-pass form read the 302-MB matrix twice: 232 us.)  H <= 256 (four columns per lane).
__global__ void __launch_bounds__(256) cin_outer_bwd_kernel(const float* __restrict__ dhad,
                                                            const float* __restrict__ x0t, int F,
                                                            const float* __restrict__ xi, int H, int64_t R,
                                                            float* __restrict__ dx0t, int accumulate_x0,
                                                            float* __restrict__ dxi, int64_t ld) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* __restrict__ g = dhad + r * ld;
  const float* __restrict__ a = x0t + r * F;
  const float* __restrict__ c = xi + r * H;
  constexpr int CM = 4;                                    // columns per lane: H <= 256
  float cv[CM], sv[CM];
#pragma unroll
  for (int q = 0; q < CM; ++q) {
    const int m = lane + 64 * q;
    cv[q] = m < H ? c[m] : 0.f;
    sv[q] = 0.f;
  }
  for (int h = 0; h < F; ++h) {
    const float ah = a[h];
    float p = 0.f;
#pragma unroll
    for (int q = 0; q < CM; ++q) {
      const int m = lane + 64 * q;
      if (64 * q < H) {                                    // (wave-uniform)
        const float gv = m < H ? g[h * H + m] : 0.f;
        sv[q] += gv * ah;
        p += gv * cv[q];
      }
    }
    for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, kWave);
    if (lane == 0) dx0t[r * F + h] = accumulate_x0 ? dx0t[r * F + h] + p : p;
  }
#pragma unroll
  for (int q = 0; q < CM; ++q) {
    const int m = lane + 64 * q;
    if (m < H) dxi[r * H + m] = sv[q];
  }
}

// out[b*ld_out + o] = sum_d xt[b, d, o]
__global__ void __launch_bounds__(256) cin_pool_fwd_kernel(const float* __restrict__ xt, int64_t B, int E,
                                                           int H, float* __restrict__ out, int64_t ld_out) {
  const int64_t n = B * H;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / H;
    const int o = (int)(t - b * H);
    float s = 0.f;
    for (int d = 0; d < E; ++d) s += xt[(b * E + d) * H + o];
    out[b * ld_out + o] = s;
  }
}

// dxt[b, d, o] (+)= g[b*ld_g + o]
__global__ void __launch_bounds__(256) cin_pool_bwd_kernel(const float* __restrict__ g, int64_t ld_g,
                                                           int64_t B, int E, int H, float* __restrict__ dxt,
                                                           int accumulate) {
  const int64_t per = (int64_t)E * H, n = B * per;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / per;
    const int o = (int)((t - b * per) % H);
    const float v = g[b * ld_g + o];
    dxt[t] = accumulate ? dxt[t] + v : v;
  }
}

}  // namespace mapx

extern "C" int mapx_transpose_batched(const float* x, int64_t B, int R, int C, float* out, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(B >= 0 && R > 0 && C > 0, "transpose_batched: bad sizes");
  if (B == 0) return MAPX_OK;
  MAPX_REQUIRE(x && out, "transpose_batched: null pointer");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(grid_for(B * R * C, 256)), dim3(256), 0, stream, x, B, R, C, out);
  return check_launch("transpose_batched");
}

extern "C" int mapx_cin_outer_fwd(const float* x0t, int F, const float* xi, int H, int64_t R, float* had,
                                  int64_t ld_had, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(R >= 0 && F > 0 && H > 0 && ld_had >= (int64_t)F * H, "cin_outer_fwd: bad sizes");
  if (R == 0) return MAPX_OK;
  MAPX_REQUIRE(x0t && xi && had, "cin_outer_fwd: null pointer");
  hipLaunchKernelGGL(cin_outer_fwd_kernel, dim3((unsigned)(R < 8192 ? R : 8192), (unsigned)((ld_had + 2047) / 2048)),
                     dim3(256), 0, stream, x0t, F, xi, H, R, had, ld_had);
  return check_launch("cin_outer_fwd");
}

extern "C" int mapx_cin_outer_bwd(const float* dhad, int64_t ld_dhad, const float* x0t, int F, const float* xi, int H,
                                  int64_t R, float* dx0t, int accumulate_x0, float* dxi, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(R >= 0 && F > 0 && H > 0 && H <= 256 && R < (1LL << 33) && ld_dhad >= (int64_t)F * H,
               "cin_outer_bwd: bad sizes (H <= 256)");
  if (R == 0) return MAPX_OK;
  MAPX_REQUIRE(dhad && x0t && xi && dx0t && dxi, "cin_outer_bwd: null pointer");
  hipLaunchKernelGGL(cin_outer_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, dhad, x0t, F, xi, H,
                     R, dx0t, accumulate_x0, dxi, ld_dhad);
  return check_launch("cin_outer_bwd");
}

extern "C" int mapx_cin_pool_fwd(const float* xt, int64_t B, int E, int H, float* out, int64_t ld_out,
                                 hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(B >= 0 && E > 0 && H > 0 && ld_out >= H, "cin_pool_fwd: bad sizes");
  if (B == 0) return MAPX_OK;
  MAPX_REQUIRE(xt && out, "cin_pool_fwd: null pointer");
  hipLaunchKernelGGL(cin_pool_fwd_kernel, dim3(grid_for(B * H, 256)), dim3(256), 0, stream, xt, B, E, H, out, ld_out);
  return check_launch("cin_pool_fwd");
}

extern "C" int mapx_cin_pool_bwd(const float* g, int64_t ld_g, int64_t B, int E, int H, float* dxt, int accumulate,
                                 hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(B >= 0 && E > 0 && H > 0 && ld_g >= H, "cin_pool_bwd: bad sizes");
  if (B == 0) return MAPX_OK;
  MAPX_REQUIRE(g && dxt, "cin_pool_bwd: null pointer");
  hipLaunchKernelGGL(cin_pool_bwd_kernel, dim3(grid_for(B * E * H, 256)), dim3(256), 0, stream, g, ld_g, B, E, H, dxt,
                     accumulate);
  return check_launch("cin_pool_bwd");
}
