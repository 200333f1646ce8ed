// Evaluation metrics on the device: ROC-AUC, log-loss, mean logit / mean probability of one
// eval pass (reference code/trainer.py:163-199 collects logits on the host as Python lists and
// calls sklearn.metrics.roc_auc_score / log_loss on float64 copies of the fp32 sigmoid).
//
//   p32   = sigmoid(logit) in fp32                    (trainer.py:190, torch.sigmoid on fp32)
//   AUC   = P(score+ > score-) + 0.5 P(score+ == score-) on p32, ties included — what the
//           trapezoid of sklearn's ROC curve evaluates to.  Computed exactly in integers:
//           sort by p32, one scan carrying {negatives so far, start of the current tie run},
//           and per tie run  2U += positives_in_run * (negatives_before + negatives_through).
//   loss  = mean of -log(clip(p)) for positives, -log(clip(1 - p)) for negatives, in fp64 with
//           sklearn's clip to [eps, 1 - eps], eps = 2^-52 (log_loss on a float64 array).
// All reductions run in a fixed order (block partials, then one block), integers by atomics.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

constexpr int kEvalBlock = 256;
constexpr int kEvalMaxBlocks = 1024;

struct RunState {
  int32_t neg;     // negatives among the sorted elements [0, i]
  int32_t start;   // index where the tie run containing i starts
};
struct RunCombine {
  __host__ __device__ RunState operator()(const RunState& a, const RunState& b) const {
    return RunState{a.neg + b.neg, a.start > b.start ? a.start : b.start};
  }
};
struct RunInput {           // element i of the sorted sequence -> its scan input
  const uint32_t* keys;
  const uint8_t* pos;
  __host__ __device__ RunState operator()(int32_t i) const {
    const bool head = (i == 0) || (keys[i] != keys[i - 1]);
    return RunState{pos[i] ? 0 : 1, head ? i : 0};
  }
};

__device__ inline double block_sum_f64(double v, double* smem) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < kEvalBlock / 64; ++w) t += smem[w];
  return t;   // valid on thread 0
}

// keys = bits of the fp32 probability (non-negative floats order like unsigned ints);
// per-block partial sums of {loss, logit, probability, positives} in fp64.
__global__ void __launch_bounds__(kEvalBlock) eval_prepare_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, int64_t n,
    uint32_t* __restrict__ keys, uint8_t* __restrict__ pos, double* __restrict__ partial) {
  __shared__ double smem[kEvalBlock / 64];
  const double eps = 2.220446049250313e-16;
  double ll = 0.0, sx = 0.0, sp = 0.0, np = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kEvalBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kEvalBlock) {
    const float x = logits[i];
    const float p32 = 1.0f / (1.0f + expf(-x));
    const bool y = labels[i] > 0.5f;
    keys[i] = __float_as_uint(p32);
    pos[i] = y ? 1 : 0;
    const double p = (double)p32;
    double q = y ? p : 1.0 - p;
    q = q < eps ? eps : (q > 1.0 - eps ? 1.0 - eps : q);
    ll -= log(q);
    sx += (double)x;
    sp += p;
    np += y ? 1.0 : 0.0;
  }
  double r;
  r = block_sum_f64(ll, smem); if (threadIdx.x == 0) partial[4 * blockIdx.x + 0] = r;
  r = block_sum_f64(sx, smem); if (threadIdx.x == 0) partial[4 * blockIdx.x + 1] = r;
  r = block_sum_f64(sp, smem); if (threadIdx.x == 0) partial[4 * blockIdx.x + 2] = r;
  r = block_sum_f64(np, smem); if (threadIdx.x == 0) partial[4 * blockIdx.x + 3] = r;
}

// One term per tie run, added at the run's last element.
__global__ void __launch_bounds__(kEvalBlock) eval_runs_kernel(const uint32_t* __restrict__ keys,
                                                               const RunState* __restrict__ st, int64_t n,
                                                               unsigned long long* __restrict__ u2) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * kEvalBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kEvalBlock) {
    const bool last = (i == n - 1) || (keys[i + 1] != keys[i]);
    if (!last) continue;
    const RunState e = st[i];
    const int64_t neg_before = e.start > 0 ? st[e.start - 1].neg : 0;
    const int64_t neg_through = e.neg;
    const int64_t pos_run = (i - e.start + 1) - (neg_through - neg_before);
    acc += (unsigned long long)pos_run * (unsigned long long)(neg_before + neg_through);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(u2, acc);
}

// out6 = {auc, logloss, mean logit, mean probability, positives, negatives}
__global__ void eval_finalize_kernel(const double* __restrict__ partial, int nblocks, int64_t n,
                                     const unsigned long long* __restrict__ u2, double* __restrict__ out6) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double ll = 0.0, sx = 0.0, sp = 0.0, np = 0.0;
  for (int b = 0; b < nblocks; ++b) {
    ll += partial[4 * b + 0];
    sx += partial[4 * b + 1];
    sp += partial[4 * b + 2];
    np += partial[4 * b + 3];
  }
  const double nn = (double)n - np;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  out6[0] = (np > 0.0 && nn > 0.0) ? ((double)(*u2) * 0.5) / (np * nn) : nan;
  out6[1] = ll / (double)n;
  out6[2] = sx / (double)n;
  out6[3] = sp / (double)n;
  out6[4] = np;
  out6[5] = nn;
}

constexpr unsigned kProbBits = 30;   // fp32 values in [0, 1] have bit patterns <= 0x3F800000 < 2^30

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct EvalLayout {
  size_t keys_in, keys_out, pos_in, pos_out, st, partial, u2, temp, temp_bytes, total;
};

static EvalLayout eval_layout(int64_t n) {
  EvalLayout L{};
  size_t sort_bytes = 0, scan_bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (const uint8_t*)nullptr, (uint8_t*)nullptr, (size_t)n, 0u, kProbBits,
                                  hipStream_t(0));
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int32_t>(0), RunInput{nullptr, nullptr});
  (void)rocprim::inclusive_scan(nullptr, scan_bytes, in, (RunState*)nullptr, (size_t)n, RunCombine(),
                                hipStream_t(0));
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
  L.keys_in = take((size_t)n * 4);
  L.keys_out = take((size_t)n * 4);
  L.pos_in = take((size_t)n);
  L.pos_out = take((size_t)n);
  L.st = take((size_t)n * sizeof(RunState));
  L.partial = take((size_t)kEvalMaxBlocks * 4 * sizeof(double));
  L.u2 = take(sizeof(unsigned long long));
  L.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  L.temp = take(L.temp_bytes);
  L.total = off;
  return L;
}

}  // namespace mapx

extern "C" size_t mapx_eval_metrics_workspace_bytes(int64_t n) {
  return n > 0 ? mapx::eval_layout(n).total : 256;
}

extern "C" int mapx_eval_metrics(const float* logits, const float* labels, int64_t n, double* out6, void* ws,
                                 size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(n >= 1, "eval_metrics: no examples");
  MAPX_REQUIRE(n < ((int64_t)1 << 31), "eval_metrics: at most 2^31 - 1 examples");
  MAPX_REQUIRE(logits && labels && out6 && ws, "eval_metrics: null pointer");
  const EvalLayout L = eval_layout(n);
  if (ws_bytes < L.total) {
    set_error("eval_metrics: workspace %zu B < %zu B", ws_bytes, L.total);
    return MAPX_EWORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(base + L.keys_in);
  uint32_t* keys = reinterpret_cast<uint32_t*>(base + L.keys_out);
  uint8_t* pos_in = reinterpret_cast<uint8_t*>(base + L.pos_in);
  uint8_t* pos = reinterpret_cast<uint8_t*>(base + L.pos_out);
  RunState* st = reinterpret_cast<RunState*>(base + L.st);
  double* partial = reinterpret_cast<double*>(base + L.partial);
  unsigned long long* u2 = reinterpret_cast<unsigned long long*>(base + L.u2);
  const int nblocks = grid_for(n, kEvalBlock, kEvalMaxBlocks);
  MAPX_HIP(hipMemsetAsync(u2, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(eval_prepare_kernel, dim3(nblocks), dim3(kEvalBlock), 0, stream, logits, labels, n,
                     keys_in, pos_in, partial);
  size_t tb = L.temp_bytes;
  MAPX_HIP(rocprim::radix_sort_pairs(base + L.temp, tb, keys_in, keys, pos_in, pos, (size_t)n, 0u, kProbBits,
                                     stream));
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int32_t>(0), RunInput{keys, pos});
  tb = L.temp_bytes;
  MAPX_HIP(rocprim::inclusive_scan(base + L.temp, tb, in, st, (size_t)n, RunCombine(), stream));
  hipLaunchKernelGGL(eval_runs_kernel, dim3(nblocks), dim3(kEvalBlock), 0, stream, keys, st, n, u2);
  hipLaunchKernelGGL(eval_finalize_kernel, dim3(1), dim3(64), 0, stream, partial, nblocks, n, u2, out6);
  return check_launch("eval_metrics");
}
