// Segment plan: sort n int32 keys (table row ids) and describe the runs of equal keys.
// Built once per step per table from the ids alone, BEFORE the forward pass, so the same
// plan serves the lazy-optimizer catch-up of the touched rows (optim.hip) and the
// deterministic gradient reduce-by-key after the backward pass (segreduce.h).
//
// The sort is the hand-written LSD radix sort of radixsort.h (8-bit digits: two launches per pass, two more for the
// runs; 9- and 12-bit digits and a scan launch per pass were measured and lost).  rocPRIM (header-only, compiled
// into this library) serves the comparison path MAPX_SORT=1 only.
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/mapx_hip.h"
#include "common.h"
#include "radixsort.h"
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
      n_uniq[0] = r;
      n_uniq[1] = 0;      // owner counter of the one segment reduction that will use this plan
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

static size_t offsets_scan_temp_bytes(int64_t m) {
  size_t sz = 0;
  (void)rocprim::exclusive_scan(nullptr, sz, (const int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)m,
                                rocprim::plus<int32_t>(), hipStream_t(0));
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

namespace mapx {
// rocPRIM's radix sort with the merge-sort shortcut disabled (MergeSortLimit = 0): Onesweep
// (one histogram launch + one scan + one decoupled-look-back pass per digit) at every size.
using OnesweepCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                               rocprim::default_config, 0>;
static size_t onesweep_temp_bytes(int64_t n, int bits) {
  size_t sz = 0;
  rocprim::counting_iterator<int32_t> iota(0);
  (void)rocprim::radix_sort_pairs<OnesweepCfg>(nullptr, sz, (const int32_t*)nullptr, (int32_t*)nullptr, iota,
                                               (int32_t*)nullptr, (size_t)n, 0u, (unsigned)bits,
                                               hipStream_t(0));
  return sz;
}
// measured inside the full step (MI355X): hand-written passes 1.70 ms/step, Onesweep 1.76 ms/step
static int sort_mode() {   // 0 = hand-written LSD passes (radixsort.h, default), 1 = rocPRIM Onesweep
  static int m = [] { const char* e = getenv("MAPX_SORT"); return e ? atoi(e) : 0; }();
  return m;
}

struct PlanWs {   // carve-up of the caller's workspace
  int32_t *tk, *tv, *bh, *off;
  void* scan;
  size_t scan_bytes, total;
};
static PlanWs plan_ws(void* ws, int64_t n) {
  PlanWs w;
  const size_t hist = (size_t)(1 << kSortBits) * radix_ld(radix_blocks(n));
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes); return at; };
  char* base = static_cast<char*>(ws);
  w.tk = reinterpret_cast<int32_t*>(base + take((size_t)n * 4));
  w.tv = reinterpret_cast<int32_t*>(base + take((size_t)n * 4));
  w.bh = reinterpret_cast<int32_t*>(base + take(hist * 4));
  w.off = reinterpret_cast<int32_t*>(base + take(hist * 4));
  const size_t a = offsets_scan_temp_bytes((int64_t)hist), b = scan_temp_bytes(n);
  const size_t c = onesweep_temp_bytes(n, 31);
  w.scan_bytes = a > b ? a : b;
  if (c > w.scan_bytes) w.scan_bytes = c;
  w.scan = base + take(w.scan_bytes);
  w.total = o;
  return w;
}
}  // namespace mapx

extern "C" size_t mapx_seg_plan_workspace_bytes(int64_t n, int64_t V) {
  (void)V;
  if (n <= 0) return 256;
  return mapx::plan_ws(nullptr, n).total + 256;
}

extern "C" int mapx_seg_plan(const int32_t* keys, int64_t n, int64_t V, void* ws, size_t ws_bytes,
                             int32_t* sorted_keys, int32_t* perm, int32_t* rank, int32_t* uniq,
                             int32_t* seg_start, int32_t* n_uniq, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(n >= 0 && n < (1LL << 31) && V > 0 && V < (1LL << 31), "seg_plan: bad sizes");
  MAPX_REQUIRE(n_uniq && seg_start, "seg_plan: null output");
  if (n == 0) {
    MAPX_HIP(hipMemsetAsync(n_uniq, 0, 2 * sizeof(int32_t), stream));
    MAPX_HIP(hipMemsetAsync(seg_start, 0, sizeof(int32_t), stream));
    return MAPX_OK;
  }
  MAPX_REQUIRE(keys && sorted_keys && perm && rank && uniq && ws, "seg_plan: null pointer");
  MAPX_REQUIRE((uintptr_t)ws % 256 == 0, "seg_plan: workspace must be 256-byte aligned");
  const PlanWs w = plan_ws(ws, n);
  if (ws_bytes < w.total) {
    set_error("seg_plan: workspace %zu < %zu bytes", ws_bytes, w.total);
    return MAPX_EWORKSPACE;
  }
  const int bits = key_bits_for(V), passes = radix_passes(bits), nblocks = radix_blocks(n);
  if (sort_mode() == 1) {
    rocprim::counting_iterator<int32_t> iota(0);
    size_t sb = w.scan_bytes;
    MAPX_HIP(rocprim::radix_sort_pairs<OnesweepCfg>(w.scan, sb, keys, sorted_keys, iota, perm, (size_t)n, 0u,
                                                    (unsigned)bits, stream));
  }
  // ping-pong so that the last pass lands in (sorted_keys, perm)
  const int32_t* src_k = keys;
  const int32_t* src_v = nullptr;
  for (int p = 0; p < passes && sort_mode() == 0; ++p) {
    const int shift = p * kSortBits;
    const int db = (bits - shift) < kSortBits ? (bits - shift) : kSortBits;
    const int bins = 1 << db;
    const bool to_out = ((passes - 1 - p) % 2) == 0;
    int32_t* dst_k = to_out ? sorted_keys : w.tk;
    int32_t* dst_v = to_out ? perm : w.tv;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(256), 0, stream, src_k, n, shift, bins,
                       nblocks, w.bh);
    if (p == 0)
      hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(nblocks), dim3(256), 0, stream, src_k, src_v,
                         n, shift, db, nblocks, (const int32_t*)w.bh, dst_k, dst_v);
    else
      hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(nblocks), dim3(256), 0, stream, src_k, src_v,
                         n, shift, db, nblocks, (const int32_t*)w.bh, dst_k, dst_v);
    src_k = dst_k;
    src_v = dst_v;
  }
  if (sort_mode() == 1) {   // comparison path: device scan + mark
    auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<int32_t>(0),
                                                  HeadFlag{sorted_keys});
    size_t sb = w.scan_bytes;
    MAPX_HIP(rocprim::inclusive_scan(w.scan, sb, flags, rank, (size_t)n, rocprim::plus<int32_t>(), stream));
    hipLaunchKernelGGL(seg_mark_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, sorted_keys, rank, n,
                       uniq, seg_start, n_uniq);
  } else {
    hipLaunchKernelGGL(seg_count_kernel, dim3(nblocks), dim3(256), 0, stream, (const int32_t*)sorted_keys, n,
                       w.off);
    hipLaunchKernelGGL(seg_mark_tiles_kernel, dim3(nblocks), dim3(256), 0, stream, (const int32_t*)sorted_keys,
                       n, (const int32_t*)w.off, rank, uniq, seg_start, n_uniq);
  }
  return check_launch("seg_plan");
}

// ---------------------------------------------------------------------------------------------
// The plans of up to kMaxSortProbs key lists (the step's tables) from ONE chain of launches
// (radixsort.h, multi-problem form).  Workspace per problem: two ping-pong arrays, one histogram
// matrix per pass, the tile counts.
namespace mapx {
struct MultiWs {
  size_t tk, tv, bh[kMaxPasses], cnt, total;
};
static MultiWs multi_ws(int64_t n, int passes, size_t at) {
  MultiWs w;
  size_t o = at;
  auto take = [&](size_t bytes) { size_t a = o; o = align_up(o + bytes); return a; };
  const size_t hist = (size_t)(1 << kSortBits) * radix_ld(radix_blocks(n));
  w.tk = take((size_t)n * 4);
  w.tv = take((size_t)n * 4);
  for (int p = 0; p < kMaxPasses; ++p) w.bh[p] = p < 1 ? take(hist * 4) : 0;
  (void)passes;
  w.cnt = take((size_t)radix_blocks(n) * 4);
  w.total = o;
  return w;
}
}  // namespace mapx

extern "C" size_t mapx_seg_plan_multi_workspace_bytes(int count, const int64_t* n, const int64_t* V) {
  using namespace mapx;
  size_t at = 0;
  for (int q = 0; q < count; ++q)
    if (n[q] > 0) at = multi_ws(n[q], radix_passes(key_bits_for(V[q])), at).total;
  return at + 256;
}

extern "C" int mapx_seg_plan_multi(int count, const int32_t* const* keys, const int64_t* n, const int64_t* V,
                                   void* ws, size_t ws_bytes, int32_t* const* sorted_keys, int32_t* const* perm,
                                   int32_t* const* rank, int32_t* const* uniq, int32_t* const* seg_start,
                                   int32_t* const* n_uniq, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(count >= 1 && count <= kMaxSortProbs, "seg_plan_multi: 1..%d key lists per call", kMaxSortProbs);
  MAPX_REQUIRE(keys && n && V && sorted_keys && perm && rank && uniq && seg_start && n_uniq, "seg_plan_multi: null array");
  SortProbs ps;
  memset(&ps, 0, sizeof(ps));
  int blocks = 0, max_passes = 0;
  size_t at = 0;
  char* base = static_cast<char*>(ws);
  for (int q = 0; q < count; ++q) {
    MAPX_REQUIRE(n[q] >= 0 && n[q] < (1LL << 31) && V[q] > 0 && V[q] < (1LL << 31), "seg_plan_multi: bad sizes");
    MAPX_REQUIRE(n_uniq[q] && seg_start[q], "seg_plan_multi: null output");
    if (n[q] == 0) {          // an empty list: its plan is {0 runs}; it takes no part in the launches
      MAPX_HIP(hipMemsetAsync(n_uniq[q], 0, 2 * sizeof(int32_t), stream));
      MAPX_HIP(hipMemsetAsync(seg_start[q], 0, sizeof(int32_t), stream));
      continue;
    }
    MAPX_REQUIRE(keys[q] && sorted_keys[q] && perm[q] && rank[q] && uniq[q] && ws, "seg_plan_multi: null pointer");
    MAPX_REQUIRE((uintptr_t)ws % 256 == 0, "seg_plan_multi: workspace must be 256-byte aligned");
    SortProb& pr = ps.p[ps.count];
    pr.keys = keys[q]; pr.n = n[q];
    pr.nblocks = radix_blocks(n[q]); pr.block0 = blocks;
    pr.bits = key_bits_for(V[q]); pr.passes = radix_passes(pr.bits);
    MAPX_REQUIRE(pr.passes <= kMaxPasses, "seg_plan_multi: keys wider than %d bits", kMaxPasses * kSortBits);
    const MultiWs w = multi_ws(n[q], pr.passes, at);
    if (ws_bytes < w.total) {
      set_error("seg_plan_multi: workspace %zu < %zu bytes", ws_bytes, w.total);
      return MAPX_EWORKSPACE;
    }
    at = w.total;
    pr.tk = reinterpret_cast<int32_t*>(base + w.tk);
    pr.tv = reinterpret_cast<int32_t*>(base + w.tv);
    pr.bh[0] = reinterpret_cast<int32_t*>(base + w.bh[0]);
    pr.cnt = reinterpret_cast<int32_t*>(base + w.cnt);
    pr.sorted_keys = sorted_keys[q]; pr.perm = perm[q]; pr.rank = rank[q]; pr.uniq = uniq[q];
    pr.seg_start = seg_start[q]; pr.n_uniq = n_uniq[q];
    blocks += pr.nblocks;
    if (pr.passes > max_passes) max_passes = pr.passes;
    ++ps.count;
  }
  if (ps.count == 0) return MAPX_OK;
  for (int p = 0; p < max_passes; ++p) {
    hipLaunchKernelGGL(radix_hist_mp_kernel, dim3(blocks), dim3(256), 0, stream, ps, p);
    if (p == 0) hipLaunchKernelGGL(radix_scatter_mp_kernel<true>, dim3(blocks), dim3(256), 0, stream, ps, p);
    else hipLaunchKernelGGL(radix_scatter_mp_kernel<false>, dim3(blocks), dim3(256), 0, stream, ps, p);
  }
  hipLaunchKernelGGL(seg_count_mp_kernel, dim3(blocks), dim3(256), 0, stream, ps);
  hipLaunchKernelGGL(seg_mark_mp_kernel, dim3(blocks), dim3(256), 0, stream, ps);
  return check_launch("seg_plan_multi");
}

namespace mapx {
// rank of every key of `lists` sorted lists in their stable merge
__global__ void __launch_bounds__(256) merge_rank_kernel(const int32_t* __restrict__ keys, int lists,
                                                         int64_t len, int32_t* __restrict__ sorted_keys,
                                                         int32_t* __restrict__ perm) {
  const int64_t n = (int64_t)lists * len;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / len);
    const uint32_t k = (uint32_t)keys[e];
    int64_t pos = e - (int64_t)r * len;
    for (int q = 0; q < lists; ++q) {
      if (q == r) continue;
      const int32_t* __restrict__ L = keys + (int64_t)q * len;
      int64_t lo = 0, hi = len;            // q < r: #{L <= k};  q > r: #{L < k}
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const uint32_t v = (uint32_t)L[mid];
        const bool left = q < r ? v <= k : v < k;
        lo = left ? mid + 1 : lo;
        hi = left ? hi : mid;
      }
      pos += lo;
    }
    sorted_keys[pos] = (int32_t)k;
    perm[pos] = (int32_t)e;
  }
}
}  // namespace mapx

extern "C" int mapx_seg_plan_merge(const int32_t* keys, int lists, int64_t len, void* ws, size_t ws_bytes,
                                   int32_t* sorted_keys, int32_t* perm, int32_t* rank, int32_t* uniq,
                                   int32_t* seg_start, int32_t* n_uniq, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(lists >= 1 && len >= 0 && (int64_t)lists * len < (1LL << 31), "seg_plan_merge: bad sizes");
  MAPX_REQUIRE(n_uniq && seg_start, "seg_plan_merge: null output");
  const int64_t n = (int64_t)lists * len;
  if (n == 0) {
    MAPX_HIP(hipMemsetAsync(n_uniq, 0, 2 * sizeof(int32_t), stream));
    MAPX_HIP(hipMemsetAsync(seg_start, 0, sizeof(int32_t), stream));
    return MAPX_OK;
  }
  MAPX_REQUIRE(keys && sorted_keys && perm && rank && uniq && ws, "seg_plan_merge: null pointer");
  MAPX_REQUIRE((uintptr_t)ws % 256 == 0, "seg_plan_merge: workspace must be 256-byte aligned");
  const PlanWs w = plan_ws(ws, n);
  if (ws_bytes < w.total) {
    set_error("seg_plan_merge: workspace %zu < %zu bytes", ws_bytes, w.total);
    return MAPX_EWORKSPACE;
  }
  const int nblocks = radix_blocks(n);
  hipLaunchKernelGGL(merge_rank_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, keys, lists, len,
                     sorted_keys, perm);
  hipLaunchKernelGGL(seg_count_kernel, dim3(nblocks), dim3(256), 0, stream, (const int32_t*)sorted_keys, n,
                     w.off);
  hipLaunchKernelGGL(seg_mark_tiles_kernel, dim3(nblocks), dim3(256), 0, stream, (const int32_t*)sorted_keys,
                     n, (const int32_t*)w.off, rank, uniq, seg_start, n_uniq);
  return check_launch("seg_plan_merge");
}

// Generic reduce-by-key of dense rows: out[u, :] = sum over the positions of key uniq[u] of
// src[position, :], in sorted-position order.  Embedding-table gradient: src = dL/dX0
// viewed as [B*F, E], keys = input_ids.flatten() (reference: aten::embedding_dense_backward).
extern "C" size_t mapx_seg_reduce_workspace_bytes(int64_t n, int W) {
  return mapx::seg_reduce_partial_bytes(n, W, true) + 256;
}

extern "C" int mapx_seg_reduce_rows(int64_t n, const int32_t* perm, const int32_t* rank,
                                    const int32_t* seg_start, const float* src, const float* src2_opt, int W,
                                    float* out, void* ws, size_t ws_bytes, int32_t* zeroed_counter_opt,
                                    hipStream_t stream) {
  MAPX_REQUIRE(n >= 0, "seg_reduce_rows: n < 0");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(perm && rank && seg_start && src && out, "seg_reduce_rows: null pointer");
  MAPX_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)src2_opt % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                   ((uintptr_t)ws % 16 == 0),
               "seg_reduce_rows: pointers must be 16-byte aligned");
  mapx::SegPlanView pl{n, perm, rank, seg_start};
  if (src2_opt) {
    mapx::Rows2Contrib c2{src, src2_opt, W};
    return mapx::seg_reduce_launch<false>(pl, c2, W, out, nullptr, ws, ws_bytes, zeroed_counter_opt, stream,
                                          "seg_reduce_rows");
  }
  mapx::RowsContrib c{src, W};
  return mapx::seg_reduce_launch<false>(pl, c, W, out, nullptr, ws, ws_bytes, zeroed_counter_opt, stream,
                                        "seg_reduce_rows");
}

// Same reduction with an extra scalar per run: out_extra[u] = sum over the run of
// extra[position / group] (DeepFM: the LR weight shares the embedding's ids, its gradient
// dL/dlr[b] is common to the F positions of batch row b).
extern "C" int mapx_seg_reduce_rows_bf16(int64_t n, const int32_t* perm, const int32_t* rank,
                                         const int32_t* seg_start, const mapx_bf16* src, const mapx_bf16* src2_opt,
                                         int W, float* out, void* ws, size_t ws_bytes,
                                         int32_t* zeroed_counter_opt, hipStream_t stream) {
  MAPX_REQUIRE(n >= 0, "seg_reduce_rows_bf16: n < 0");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(perm && rank && seg_start && src && out, "seg_reduce_rows_bf16: null pointer");
  MAPX_REQUIRE((uintptr_t)src % 8 == 0 && (uintptr_t)src2_opt % 8 == 0 && (uintptr_t)out % 16 == 0 &&
                   (uintptr_t)ws % 16 == 0,
               "seg_reduce_rows_bf16: pointers must be 8 / 16-byte aligned");
  mapx::SegPlanView pl{n, perm, rank, seg_start};
  if (src2_opt) {
    mapx::Rows2Bf16Contrib c2{reinterpret_cast<const __bf16*>(src), reinterpret_cast<const __bf16*>(src2_opt), W};
    return mapx::seg_reduce_launch<false>(pl, c2, W, out, nullptr, ws, ws_bytes, zeroed_counter_opt, stream,
                                          "seg_reduce_rows_bf16");
  }
  mapx::RowsBf16Contrib c{reinterpret_cast<const __bf16*>(src), W};
  return mapx::seg_reduce_launch<false>(pl, c, W, out, nullptr, ws, ws_bytes, zeroed_counter_opt, stream,
                                        "seg_reduce_rows_bf16");
}

extern "C" int mapx_seg_reduce_rows_extra(int64_t n, const int32_t* perm, const int32_t* rank,
                                          const int32_t* seg_start, const float* src, int W, int64_t ld_src,
                                          const float* extra, int group, int64_t extra_stride, float* out,
                                          float* out_extra,
                                          void* ws, size_t ws_bytes, int32_t* zeroed_counter_opt,
                                          hipStream_t stream) {
  MAPX_REQUIRE(n >= 0 && group >= 1 && ld_src >= W && ld_src % 4 == 0 && extra_stride >= 1,
               "seg_reduce_rows_extra: bad sizes");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(perm && rank && seg_start && src && extra && out && out_extra, "seg_reduce_rows_extra: null pointer");
  MAPX_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)ws % 16 == 0),
               "seg_reduce_rows_extra: pointers must be 16-byte aligned");
  mapx::SegPlanView pl{n, perm, rank, seg_start};
  mapx::RowsExtraContrib c{src, W, ld_src, extra, group, extra_stride};
  return mapx::seg_reduce_launch<true>(pl, c, W, out, out_extra, ws, ws_bytes, zeroed_counter_opt, stream,
                                       "seg_reduce_rows_extra");
}

// Data-parallel exchange (mapx/parallel.py): the first n_uniq (id, gradient row) pairs of a rank's
// sparse gradient as a fixed-size message of `maxc` entries: keys_out[i] = uniq[i], rows_out[i] =
// {rows0[i, :] * scale, rows1[i] * scale, 0, 0, 0} (rows1 optional: then W0 columns only); entries
// at and beyond *n_uniq are id `pad_id` with a zero row.  One launch instead of a dozen tensor ops.
namespace mapx {
__global__ void __launch_bounds__(256) pack_sparse_kernel(const int32_t* __restrict__ uniq,
                                                          const float* __restrict__ rows0, int W0,
                                                          const float* __restrict__ rows1,
                                                          const int32_t* __restrict__ n_uniq, int64_t cap,
                                                          int64_t maxc, float scale, int32_t pad_id,
                                                          int32_t* __restrict__ keys_out,
                                                          float* __restrict__ rows_out) {
  const int Wp = rows1 ? W0 + 4 : W0;
  const int per = Wp / 4;
  int64_t n = *n_uniq;
  if (n > cap) n = cap;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < maxc * per;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / per;
    const int c = (int)(t - i * per) * 4;
    const bool live = i < n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
      if (c < W0) {
        v = *reinterpret_cast<const float4*>(rows0 + i * W0 + c);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      } else {
        v.x = rows1[i] * scale;
      }
    }
    *reinterpret_cast<float4*>(rows_out + i * Wp + c) = v;
    if (c == 0) keys_out[i] = live ? uniq[i] : pad_id;
  }
}
__global__ void __launch_bounds__(64) publish_i32_kernel(const int32_t* __restrict__ src, int n,
                                                         int32_t* __restrict__ stamp,
                                                         volatile int32_t* __restrict__ host_out) {
  const int lane = threadIdx.x;
  int v = lane < n ? src[lane] : 0;
  if (lane < n) host_out[lane] = v;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  __threadfence_system();
  if (lane == 0) {
    const int s = *stamp + 1;
    *stamp = s;
    __hip_atomic_store((int32_t*)host_out + n + 1, v + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((int32_t*)host_out + n, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace mapx

extern "C" int mapx_pack_sparse(const int32_t* uniq, const float* rows0, int W0, const float* rows1_opt,
                                const int32_t* n_uniq, int64_t cap, int64_t maxc, float scale,
                                int32_t pad_id, int32_t* keys_out, float* rows_out, hipStream_t stream) {
  MAPX_REQUIRE(uniq && rows0 && n_uniq && keys_out && rows_out, "pack_sparse: null pointer");
  MAPX_REQUIRE(W0 > 0 && W0 % 4 == 0 && cap >= 0 && maxc >= 0, "pack_sparse: bad sizes");
  if (maxc == 0) return MAPX_OK;
  const int Wp = rows1_opt ? W0 + 4 : W0;
  hipLaunchKernelGGL(mapx::pack_sparse_kernel, dim3(mapx::grid_for(maxc * (Wp / 4), 256)), dim3(256), 0, stream,
                     uniq, rows0, W0, rows1_opt, n_uniq, cap, maxc, scale, pad_id, keys_out, rows_out);
  return mapx::check_launch("pack_sparse");
}

extern "C" int mapx_publish_i32(const int32_t* src, int n, int32_t* stamp_dev, int32_t* host_out,
                                hipStream_t stream) {
  MAPX_REQUIRE(src && stamp_dev && host_out, "publish_i32: null pointer");
  MAPX_REQUIRE(n >= 0 && n <= 64, "publish_i32: n must be in [0, 64]");
  hipLaunchKernelGGL(mapx::publish_i32_kernel, dim3(1), dim3(64), 0, stream, src, n, stamp_dev, host_out);
  return mapx::check_launch("publish_i32");
}
