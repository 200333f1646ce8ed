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
// 16-byte rows and a K that is a multiple of 8: F*H = 529 or 1150 left it the scalar operand path) are zeroed
__global__ void __launch_bounds__(256) cin_outer_fwd_kernel(const float* __restrict__ x0t, int F,
                                                            const float* __restrict__ xi, int H, int64_t R,
                                                            float* __restrict__ had, int64_t ld) {
  const int64_t K = (int64_t)F * H, n = R * ld;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / ld;
    const int k = (int)(t - r * ld);
    const int h = k / H, m = k - h * H;
    had[t] = k < K ? x0t[r * F + h] * xi[r * H + m] : 0.f;
  }
}

// one wave per row r:  dxi[r, m] = sum_h dhad[r,h,m] x0t[r,h];  dx0t[r, h] (+)= sum_m dhad[r,h,m] xi[r,m]
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
  for (int m0 = 0; m0 < H; m0 += 64) {                  // dxi: lanes over m, loop over h
    const int m = m0 + lane;
    if (m < H) {
      float s = 0.f;
      for (int h = 0; h < F; ++h) s += g[h * H + m] * a[h];
      dxi[r * H + m] = s;
    }
  }
  for (int h = 0; h < F; ++h) {                         // dx0t: wave reduction over m, fixed order
    float s = 0.f;
    for (int m = lane; m < H; m += 64) s += g[h * H + m] * c[m];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, kWave);
    if (lane == 0) dx0t[r * F + h] = accumulate_x0 ? dx0t[r * F + h] + s : s;
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
  hipLaunchKernelGGL(cin_outer_fwd_kernel, dim3(grid_for(R * ld_had, 256)), dim3(256), 0, stream, x0t, F, xi, H, R, had,
                     ld_had);
  return check_launch("cin_outer_fwd");
}

extern "C" int mapx_cin_outer_bwd(const float* dhad, int64_t ld_dhad, const float* x0t, int F, const float* xi, int H,
                                  int64_t R, float* dx0t, int accumulate_x0, float* dxi, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(R >= 0 && F > 0 && H > 0 && R < (1LL << 33) && ld_dhad >= (int64_t)F * H, "cin_outer_bwd: bad sizes");
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
