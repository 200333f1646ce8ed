// First-order (LR) and second-order (FM) terms of the DeepFM backbone:
//   LR   reference code/models.py:129-143: logit[b] = sum_f w[ids[b,f]] + bias, w = Embedding(V, 1)
//   FM   reference code/layers.py:123-131 (InnerProductLayer, output='product_sum'):
//        fm[b] = 0.5 * sum_e ( (sum_f x[b,f,e])^2 - sum_f x[b,f,e]^2 )
// Both read what the embedding gather already touches (ids, x [B,F,E]); they are HBM-bound
// elementwise / small-reduction kernels.  The LR weight lives in the same row table as the
// embedding (scalar-per-row secondary, like the NCE bias), so its gradient is the EXTRA
// column of the embedding's segment reduction (mapx_seg_reduce_rows_extra, segplan.hip).
#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

// 32 lanes per batch row: lane l sums fields l, l+32, ...; fixed-order butterfly.
__global__ void __launch_bounds__(256) lr_sum_kernel(const int64_t* __restrict__ ids, int64_t B, int F,
                                                     const float* __restrict__ w, int64_t V,
                                                     float* __restrict__ out, int* __restrict__ err) {
  const int l = threadIdx.x & 31;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5; b < B;
       b += ((int64_t)gridDim.x * blockDim.x) >> 5) {
    float v = 0.f;
    for (int f = l; f < F; f += 32) {
      const int64_t id = ids[b * F + f];
      const bool ok = (id >= 0) & (id < V);
      if (!ok && err) atomicOr(err, 1);
      v += ok ? w[id] : 0.f;
    }
    v = group_sum<32>(v);
    if (l == 0) out[b] = v;
  }
}

// One lane per (row, e): walks the F fields (the E lanes of a row read E consecutive floats).
// s[b,e] = sum_f x is kept for backward.  E <= 64 and a power of two (host-checked).
template <int E>
__global__ void __launch_bounds__(256) fm_fwd_kernel(const float* __restrict__ x, int64_t B, int F,
                                                     float* __restrict__ out, float* __restrict__ s) {
  const int e = threadIdx.x % E;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / E; b < B;
       b += ((int64_t)gridDim.x * blockDim.x) / E) {
    const float* __restrict__ xb = x + b * F * E + e;
    float sum = 0.f, sq = 0.f;
    for (int f = 0; f < F; ++f) {
      const float v = xb[(int64_t)f * E];
      sum += v;
      sq += v * v;
    }
    s[b * E + e] = sum;
    const float t = group_sum<E>(0.5f * (sum * sum - sq));
    if (e == 0) out[b] = t;
  }
}

// dx[b,f,e] = g[b] * (s[b,e] - x[b,f,e])
__global__ void __launch_bounds__(256) fm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ s,
                                                     const float* __restrict__ x, int64_t B, int F, int E,
                                                     float* __restrict__ dx) {
  const int64_t n4 = B * F * E / 4;
  const int e4 = E / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / ((int64_t)F * e4);
    const int c = (int)(i % e4);
    const float gb = g[b];
    const float4 sv = reinterpret_cast<const float4*>(s)[b * e4 + c];
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(dx)[i] = make_float4(gb * (sv.x - xv.x), gb * (sv.y - xv.y),
                                                   gb * (sv.z - xv.z), gb * (sv.w - xv.w));
  }
}

}  // namespace mapx

extern "C" int mapx_lr_sum_fwd(const int64_t* ids, int64_t B, int F, const float* w, int64_t V, float* out,
                               int* err_flag, hipStream_t stream) {
  MAPX_REQUIRE(B >= 0 && F > 0 && V > 0, "lr_sum_fwd: bad sizes");
  if (B == 0) return MAPX_OK;
  MAPX_REQUIRE(ids && w && out, "lr_sum_fwd: null pointer");
  hipLaunchKernelGGL(mapx::lr_sum_kernel, dim3(mapx::grid_for(B * 32, 256)), dim3(256), 0, stream, ids, B, F, w,
                     V, out, err_flag);
  return mapx::check_launch("lr_sum_fwd");
}

extern "C" int mapx_fm_fwd(const float* x, int64_t B, int F, int E, float* out, float* s, hipStream_t stream) {
  MAPX_REQUIRE(B >= 0 && F > 0, "fm_fwd: bad sizes");
  MAPX_REQUIRE(E == 4 || E == 8 || E == 16 || E == 32 || E == 64, "fm_fwd: embed_size %d (4, 8, 16, 32, 64)", E);
  if (B == 0) return MAPX_OK;
  MAPX_REQUIRE(x && out && s, "fm_fwd: null pointer");
  const int grid = mapx::grid_for(B * E, 256);
#define MAPX_FM(E_) hipLaunchKernelGGL(mapx::fm_fwd_kernel<E_>, dim3(grid), dim3(256), 0, stream, x, B, F, out, s)
  switch (E) {
    case 4: MAPX_FM(4); break;
    case 8: MAPX_FM(8); break;
    case 16: MAPX_FM(16); break;
    case 32: MAPX_FM(32); break;
    default: MAPX_FM(64); break;
  }
#undef MAPX_FM
  return mapx::check_launch("fm_fwd");
}

extern "C" int mapx_fm_bwd(const float* g, const float* s, const float* x, int64_t B, int F, int E, float* dx,
                           hipStream_t stream) {
  MAPX_REQUIRE(B >= 0 && F > 0 && E > 0 && E % 4 == 0, "fm_bwd: bad sizes");
  if (B == 0) return MAPX_OK;
  MAPX_REQUIRE(g && s && x && dx, "fm_bwd: null pointer");
  MAPX_REQUIRE(((uintptr_t)s % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dx % 16 == 0),
               "fm_bwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(mapx::fm_bwd_kernel, dim3(mapx::grid_for(B * F * E / 4, 256)), dim3(256), 0, stream, g, s, x,
                     B, F, E, dx);
  return mapx::check_launch("fm_bwd");
}
