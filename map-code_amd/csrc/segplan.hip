// Segment plan: sort n int32 keys (table row ids) and describe the runs of equal keys.
// Built once per step per table from the ids alone, BEFORE the forward pass, so the same
// plan serves the lazy-optimizer catch-up of the touched rows (optim.hip) and the
// deterministic gradient reduce-by-key after the backward pass (segreduce.h).
//
// The radix sort and the prefix scan are rocPRIM device primitives (header-only, compiled
// into this library); everything downstream is hand-written.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/mapx_hip.h"
#include "common.h"
#include "segreduce.h"

namespace mapx {

struct HeadFlag {
  const int32_t* sk;
  __host__ __device__ inline int32_t operator()(int32_t j) const {
    return (j == 0 || sk[j] != sk[j - 1]) ? 1 : 0;
  }
};

__global__ void __launch_bounds__(256) seg_mark_kernel(const int32_t* __restrict__ sk,
                                                       const int32_t* __restrict__ rank, int64_t n,
                                                       int32_t* __restrict__ uniq,
                                                       int32_t* __restrict__ seg_start,
                                                       int32_t* __restrict__ n_uniq) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = rank[j];
    if (j == 0 || sk[j] != sk[j - 1]) {
      uniq[r - 1] = sk[j];
      seg_start[r - 1] = (int32_t)j;
    }
    if (j == n - 1) {
      *n_uniq = r;
      seg_start[r] = (int32_t)n;
    }
  }
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static int key_bits_for(int64_t V) {
  int b = 1;
  while (((int64_t)1 << b) < V && b < 31) ++b;
  return b;
}

static size_t sort_temp_bytes(int64_t n, int bits) {
  size_t sz = 0;
  rocprim::counting_iterator<int32_t> iota(0);
  (void)rocprim::radix_sort_pairs(nullptr, sz, (const int32_t*)nullptr, (int32_t*)nullptr, iota,
                                  (int32_t*)nullptr, (size_t)n, 0u, (unsigned)bits, hipStream_t(0));
  return sz;
}

static size_t scan_temp_bytes(int64_t n) {
  size_t sz = 0;
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int32_t>(0),
                                             HeadFlag{nullptr});
  (void)rocprim::inclusive_scan(nullptr, sz, in, (int32_t*)nullptr, (size_t)n,
                                rocprim::plus<int32_t>(), hipStream_t(0));
  return sz;
}

}  // namespace mapx

extern "C" size_t mapx_seg_plan_workspace_bytes(int64_t n, int64_t V) {
  if (n <= 0) return 256;
  const int bits = mapx::key_bits_for(V);
  size_t a = mapx::sort_temp_bytes(n, bits), b = mapx::scan_temp_bytes(n);
  return mapx::align_up(a > b ? a : b) + 256;
}

extern "C" int mapx_seg_plan(const int32_t* keys, int64_t n, int64_t V, void* ws, size_t ws_bytes,
                             int32_t* sorted_keys, int32_t* perm, int32_t* rank, int32_t* uniq,
                             int32_t* seg_start, int32_t* n_uniq, hipStream_t stream) {
  MAPX_REQUIRE(n >= 0 && n < (1LL << 31) && V > 0 && V < (1LL << 31), "seg_plan: bad sizes");
  MAPX_REQUIRE(n_uniq && seg_start, "seg_plan: null output");
  if (n == 0) {
    MAPX_HIP(hipMemsetAsync(n_uniq, 0, sizeof(int32_t), stream));
    MAPX_HIP(hipMemsetAsync(seg_start, 0, sizeof(int32_t), stream));
    return MAPX_OK;
  }
  MAPX_REQUIRE(keys && sorted_keys && perm && rank && uniq && ws, "seg_plan: null pointer");
  const int bits = mapx::key_bits_for(V);
  size_t need_sort = mapx::sort_temp_bytes(n, bits), need_scan = mapx::scan_temp_bytes(n);
  if (ws_bytes < need_sort || ws_bytes < need_scan) {
    mapx::set_error("seg_plan: workspace %zu < %zu bytes", ws_bytes,
                    need_sort > need_scan ? need_sort : need_scan);
    return MAPX_EWORKSPACE;
  }
  rocprim::counting_iterator<int32_t> iota(0);
  MAPX_HIP(rocprim::radix_sort_pairs(ws, need_sort, keys, sorted_keys, iota, perm, (size_t)n, 0u,
                                     (unsigned)bits, stream));
  auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<int32_t>(0),
                                                mapx::HeadFlag{sorted_keys});
  MAPX_HIP(rocprim::inclusive_scan(ws, need_scan, flags, rank, (size_t)n, rocprim::plus<int32_t>(),
                                   stream));
  hipLaunchKernelGGL(mapx::seg_mark_kernel, dim3(mapx::grid_for(n, 256)), dim3(256), 0, stream,
                     sorted_keys, rank, n, uniq, seg_start, n_uniq);
  return mapx::check_launch("seg_plan");
}

// Generic reduce-by-key of dense rows: out[u, :] = sum over the positions of key uniq[u] of
// src[position, :], in sorted-position order.  Embedding-table gradient: src = dL/dX0
// viewed as [B*F, E], keys = input_ids.flatten() (reference: aten::embedding_dense_backward).
extern "C" size_t mapx_seg_reduce_workspace_bytes(int64_t n, int W) {
  return mapx::seg_reduce_partial_bytes(n, W, true) + 256;
}

extern "C" int mapx_seg_reduce_rows(int64_t n, const int32_t* perm, const int32_t* rank,
                                    const int32_t* seg_start, const float* src, int W, float* out,
                                    void* ws, size_t ws_bytes, hipStream_t stream) {
  MAPX_REQUIRE(n >= 0, "seg_reduce_rows: n < 0");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(perm && rank && seg_start && src && out, "seg_reduce_rows: null pointer");
  MAPX_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)ws % 16 == 0),
               "seg_reduce_rows: pointers must be 16-byte aligned");
  mapx::SegPlanView pl{n, perm, rank, seg_start};
  mapx::RowsContrib c{src, W};
  return mapx::seg_reduce_launch<false>(pl, c, W, out, nullptr, ws, ws_bytes, stream,
                                        "seg_reduce_rows");
}
