// Options of the hot-path classes that the DCNv2 run scripts leave off but the CLI accepts:
// dropout (reference code/layers.py:95 Embeddings.dropout, :183 MLPBlock's nn.Dropout) and the
// embeddings' LayerNorm (layers.py:92-94, 99-100).  HBM-bound elementwise / row kernels.
#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

// Inverted dropout, mask never stored: element i keeps its value iff Philox(seed, offset (+ *offset_dev),
// i / 4)[i % 4] >= p * 2^32, scaled by 1 / (1 - p).  Forward and backward call the same kernel with the
// same (seed, offset): the mask is regenerated, not saved (0 bytes of activation memory, graph-replay
// safe through the device-side offset).
__global__ void __launch_bounds__(256) dropout_kernel(const float* __restrict__ x, int64_t n, float p, float scale,
                                                      uint64_t seed, uint64_t offset,
                                                      const int32_t* __restrict__ offset_dev,
                                                      float* __restrict__ out) {
  const uint64_t off = offset + (offset_dev ? (uint64_t)(uint32_t)*offset_dev : 0ull);
  const uint32_t thr = (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f);
  const int64_t n4 = (n + 3) / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const Philox4 r = philox4x32_10(seed, (uint64_t)i, off);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t j = 4 * i + e;
      if (j < n) out[j] = rr[e] >= thr ? x[j] * scale : 0.f;
    }
  }
}

// LayerNorm over the last dimension E (<= 64, E % 4 == 0) of [R, E]: one lane group of E/4 lanes per row.
//   y = (x - mean) * rstd * w + b,  rstd = 1 / sqrt(var + eps)  (biased variance, as nn.LayerNorm)
template <int LG>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, int64_t R, int E,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float eps, float* __restrict__ y,
                                                            float* __restrict__ stats /* [R,2] mean, rstd */) {
  const int lig = threadIdx.x % LG;
  const bool live = lig * 4 < E;
  const float4 wv = live ? *reinterpret_cast<const float4*>(w + 4 * lig) : make_float4(0, 0, 0, 0);
  const float4 bv = live ? *reinterpret_cast<const float4*>(b + 4 * lig) : make_float4(0, 0, 0, 0);
  for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG; row < R;
       row += ((int64_t)gridDim.x * blockDim.x) / LG) {
    const float4 v = live ? *reinterpret_cast<const float4*>(x + row * E + 4 * lig) : make_float4(0, 0, 0, 0);
    const float mean = group_sum<LG>(v.x + v.y + v.z + v.w) / E;
    const float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    const float var = group_sum<LG>(live ? d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w : 0.f) / E;
    const float rstd = rsqrtf(var + eps);
    if (live)
      *reinterpret_cast<float4*>(y + row * E + 4 * lig) =
          make_float4(d.x * rstd * wv.x + bv.x, d.y * rstd * wv.y + bv.y, d.z * rstd * wv.z + bv.z, d.w * rstd * wv.w + bv.w);
    if (lig == 0) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w;  also writes dy * xhat (its column sum is dw)
template <int LG>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ stats,
                                                            int64_t R, int E, float* __restrict__ dx,
                                                            float* __restrict__ dyxhat) {
  const int lig = threadIdx.x % LG;
  const bool live = lig * 4 < E;
  const float4 wv = live ? *reinterpret_cast<const float4*>(w + 4 * lig) : make_float4(0, 0, 0, 0);
  for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG; row < R;
       row += ((int64_t)gridDim.x * blockDim.x) / LG) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float4 v = live ? *reinterpret_cast<const float4*>(x + row * E + 4 * lig) : make_float4(0, 0, 0, 0);
    const float4 gy = live ? *reinterpret_cast<const float4*>(dy + row * E + 4 * lig) : make_float4(0, 0, 0, 0);
    const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    const float4 g = make_float4(gy.x * wv.x, gy.y * wv.y, gy.z * wv.z, gy.w * wv.w);
    const float mg = group_sum<LG>(live ? g.x + g.y + g.z + g.w : 0.f) / E;
    const float mgx = group_sum<LG>(live ? g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w : 0.f) / E;
    if (live) {
      *reinterpret_cast<float4*>(dx + row * E + 4 * lig) =
          make_float4(rstd * (g.x - mg - xh.x * mgx), rstd * (g.y - mg - xh.y * mgx), rstd * (g.z - mg - xh.z * mgx),
                      rstd * (g.w - mg - xh.w * mgx));
      *reinterpret_cast<float4*>(dyxhat + row * E + 4 * lig) = make_float4(gy.x * xh.x, gy.y * xh.y, gy.z * xh.z, gy.w * xh.w);
    }
  }
}

// The reference's other activations (code/layers.py:13-80 get_act: tanh, sigmoid, none, elu, leu, gelu (erf),
// gelu_new (tanh form), swish, mish; `hidden_act` of MLPBlock, layers.py:173-188) as one elementwise pass
// after the plain Linear (+ bias) GEMM, and dz = dy f'(z) recomputed from the saved pre-activation in backward.
// Not the benchmarked path (every DCNv2 script uses relu, which is fused into the GEMM epilogue): HBM-bound.
enum { kActTanh = 1, kActSigmoid, kActNone, kActElu, kActLeu, kActGelu, kActGeluNew, kActSwish, kActMish };

__device__ inline float act_value(int kind, float z) {
  switch (kind) {
    case kActTanh: return tanhf(z);
    case kActSigmoid: return 1.f / (1.f + expf(-z));
    case kActElu: return z > 0.f ? z : expm1f(z);
    case kActLeu: return z > 0.f ? logf(z + 1.f) : expf(z) - 1.f;                 // layers.py:22-27 (alpha = 1)
    case kActGelu: return z * 0.5f * (1.f + erff(z * 0.70710678118654752f));       // layers.py:36-37
    case kActGeluNew: {                                                            // layers.py:41-42
      const float u = 0.79788456080286536f * (z + 0.044715f * z * z * z);
      return 0.5f * z * (1.f + tanhf(u));
    }
    case kActSwish: return z / (1.f + expf(-z));                                   // layers.py:46-47
    case kActMish: {                                                               // layers.py:51-52
      const float sp = fmaxf(z, 0.f) + log1pf(expf(-fabsf(z)));
      return z * tanhf(sp);
    }
    default: return z;
  }
}

__device__ inline float act_slope(int kind, float z) {
  switch (kind) {
    case kActTanh: { const float t = tanhf(z); return 1.f - t * t; }
    case kActSigmoid: { const float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
    case kActElu: return z > 0.f ? 1.f : expf(z);
    case kActLeu: return z > 0.f ? 1.f / (z + 1.f) : expf(z);
    case kActGelu:
      return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
    case kActGeluNew: {
      const float u = 0.79788456080286536f * (z + 0.044715f * z * z * z), t = tanhf(u);
      return 0.5f * (1.f + t) + 0.5f * z * (1.f - t * t) * 0.79788456080286536f * (1.f + 3.f * 0.044715f * z * z);
    }
    case kActSwish: { const float s = 1.f / (1.f + expf(-z)); return s + z * s * (1.f - s); }
    case kActMish: {
      const float sp = fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))), t = tanhf(sp), s = 1.f / (1.f + expf(-z));
      return t + z * (1.f - t * t) * s;
    }
    default: return 1.f;
  }
}

// y[m, :] (row stride ldy) = f(z[m, :]);  dz = dy * f'(z)  (dy row stride ld_dy)
__global__ void __launch_bounds__(256) act_fwd_kernel(int kind, const float* __restrict__ z, int64_t M, int N,
                                                      float* __restrict__ y, int64_t ldy) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N;
    y[m * ldy + (i - m * N)] = act_value(kind, z[i]);
  }
}
__global__ void __launch_bounds__(256) act_bwd_kernel(int kind, const float* __restrict__ dy, int64_t ld_dy,
                                                      const float* __restrict__ z, int64_t M, int N,
                                                      float* __restrict__ dz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N;
    dz[i] = dy[m * ld_dy + (i - m * N)] * act_slope(kind, z[i]);
  }
}

}  // namespace mapx

extern "C" int mapx_dropout(const float* x, int64_t n, float p, uint64_t seed, uint64_t offset,
                            const int32_t* offset_dev, float* out, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, "dropout: bad arguments (0 <= p < 1)");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(x && out, "dropout: null pointer");
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, stream, x, n, p, 1.f / (1.f - p), seed,
                     offset, offset_dev, out);
  return check_launch("dropout");
}

extern "C" int mapx_layernorm_fwd(const float* x, int64_t R, int E, const float* w, const float* b, float eps, float* y,
                                  float* stats, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(R >= 0 && E >= 4 && E <= 64 && E % 4 == 0, "layernorm_fwd: E must be a multiple of 4 in [4, 64]");
  if (R == 0) return MAPX_OK;
  MAPX_REQUIRE(x && w && b && y && stats, "layernorm_fwd: null pointer");
  const int lg = E <= 16 ? 4 : (E <= 32 ? 8 : 16);
  const int grid = grid_for(R * lg, 256);
  if (lg == 4) hipLaunchKernelGGL(layernorm_fwd_kernel<4>, dim3(grid), dim3(256), 0, stream, x, R, E, w, b, eps, y, stats);
  else if (lg == 8) hipLaunchKernelGGL(layernorm_fwd_kernel<8>, dim3(grid), dim3(256), 0, stream, x, R, E, w, b, eps, y, stats);
  else hipLaunchKernelGGL(layernorm_fwd_kernel<16>, dim3(grid), dim3(256), 0, stream, x, R, E, w, b, eps, y, stats);
  return check_launch("layernorm_fwd");
}

extern "C" int mapx_layernorm_bwd(const float* dy, const float* x, const float* w, const float* stats, int64_t R, int E,
                                  float* dx, float* dy_xhat, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(R >= 0 && E >= 4 && E <= 64 && E % 4 == 0, "layernorm_bwd: E must be a multiple of 4 in [4, 64]");
  if (R == 0) return MAPX_OK;
  MAPX_REQUIRE(dy && x && w && stats && dx && dy_xhat, "layernorm_bwd: null pointer");
  const int lg = E <= 16 ? 4 : (E <= 32 ? 8 : 16);
  const int grid = grid_for(R * lg, 256);
  if (lg == 4) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(grid), dim3(256), 0, stream, dy, x, w, stats, R, E, dx, dy_xhat);
  else if (lg == 8) hipLaunchKernelGGL(layernorm_bwd_kernel<8>, dim3(grid), dim3(256), 0, stream, dy, x, w, stats, R, E, dx, dy_xhat);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<16>, dim3(grid), dim3(256), 0, stream, dy, x, w, stats, R, E, dx, dy_xhat);
  return check_launch("layernorm_bwd");
}

extern "C" int mapx_act_fwd(int kind, const float* z, int64_t M, int N, float* y, int64_t ldy, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(kind >= kActTanh && kind <= kActMish, "act_fwd: unknown activation %d", kind);
  MAPX_REQUIRE(M >= 0 && N > 0 && ldy >= N, "act_fwd: bad sizes");
  if (M == 0) return MAPX_OK;
  MAPX_REQUIRE(z && y, "act_fwd: null pointer");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(M * N, 256)), dim3(256), 0, stream, kind, z, M, N, y, ldy);
  return check_launch("act_fwd");
}

extern "C" int mapx_act_bwd(int kind, const float* dy, int64_t ld_dy, const float* z, int64_t M, int N, float* dz,
                            hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(kind >= kActTanh && kind <= kActMish, "act_bwd: unknown activation %d", kind);
  MAPX_REQUIRE(M >= 0 && N > 0 && ld_dy >= N, "act_bwd: bad sizes");
  if (M == 0) return MAPX_OK;
  MAPX_REQUIRE(dy && z && dz, "act_bwd: null pointer");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(M * N, 256)), dim3(256), 0, stream, kind, dy, ld_dy, z, M, N, dz);
  return check_launch("act_bwd");
}
