// transformers-4.26 AdamW (the optimizer reference code/trainer.py:60-85 constructs), as
// (1) one fused dense sweep per flat parameter group and (2) a lazy, EXACT row-sparse form
// for the [V,*] tables.
//
// Update number s (1-based), gradient g, on fp32 state (p, m, v):
//     m = b1*m + (1-b1)*g ;  v = b2*v + (1-b2)*g*g
//     p = p - step_s * m / (sqrt(v) + eps)        step_s = lr_s*sqrt(1-b2^s)/(1-b1^s)
//     p = p - lr_s*wd * p                          (decoupled decay AFTER the Adam update)
// with lr_s = lr0 * lambda(s-1) (scheduler stepped after the optimizer, trainer.py:328-329).
// (step_s, lr_s) come from a device table sched[s-1] = {step_s, lr_s} computed once on the
// host in double, and the update counter lives in device memory, so a captured hipGraph
// replays the whole step without host-side scalars.
//
// The reference never uses sparse gradients, so every table row takes a zero-gradient
// update on every step it is not touched (m, v decay; p moves by the decaying momentum and
// shrinks by the weight decay): 49*V parameters * 28 B per step.  The lazy form keeps
// `last[row]` = number of updates already applied to that row and replays the missing
// zero-gradient updates IN REGISTERS, with the same fp32 operations in the same order,
// when the row is next needed (catch-up before the forward gather; again, as a no-op,
// inside the gradient update) or when the table is flushed (checkpoint / eval).
// A replay of n zero-gradient steps is applied in CLOSED FORM, O(1) per element for any n (see
// replay_coef below): host fp64 tables over the schedule give the decay product, b1^n, b2^n and
// seven per-row coefficients of the accumulated Adam displacement.  It differs from the
// reference's n successive fp32 roundings by <= n * 2^-24 relative (and from the same steps done
// in fp64 by 2e-14).  Only gaps that run past the end of the tables (steps beyond the schedule)
// are replayed step by step, with a closed-form tail once the Adam term can no longer move p.
#include "../../include/mapx_hip.h"
#include "amax.h"
#include "common.h"
#include "lazy_adam.h"

namespace mapx {

// SHADOW: also store the updated parameter as bf16 (the weight operand of the bf16 GEMMs,
// gemm_bf16.hip): the conversion rides on the one pass that touches every parameter anyway.
template <bool SHADOW>
__global__ void __launch_bounds__(256) adamw_dense_kernel(float* __restrict__ p,
                                                          const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          int64_t n, const float2* __restrict__ sched,
                                                          int sched_len, const int32_t* __restrict__ done,
                                                          AdamHyper h, float wd, __bf16* __restrict__ shadow,
                                                          const int64_t* __restrict__ seg_off, int nseg,
                                                          amax_rec* __restrict__ seg_amax,
                                                          const int32_t* __restrict__ epoch) {
  int s = *done;  // updates applied so far; this is update s+1 -> sched[s]
  if (s >= sched_len) s = sched_len - 1;
  const float2 sc = sched[s];
  const float step = sc.x, decay = sc.y * wd;
  const int64_t n4 = n / 4;
  // A block takes ONE contiguous range of float4s.  With magnitude records to keep (amax.h: the parameters this launch
  // writes are the next step's GEMM operands; parameter z = elements [seg_off[z], seg_off[z + 1]), starts multiples of 8)
  // a thread tracks the maximum of the parameter it is in, the block's threads meet in LDS (a range touches a few
  // parameters), and one thread per touched parameter raises its record: ~1 global atomic per block.  (One atomic per
  // wave and float4 — 15 k on twenty addresses — made this kernel 90 us instead of 9.)
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x, i0 = (int64_t)blockIdx.x * per;
  const int64_t i1 = i0 + per < n4 ? i0 + per : n4;
  constexpr int kSlots = 16;
  __shared__ uint32_t smax[kSlots];
  int seg = 0, seg_first = 0;
  uint32_t mx = 0;
  if (seg_amax) {
    if (threadIdx.x < kSlots) smax[threadIdx.x] = 0;
    int lo = 0, hi = nseg;
    while (hi - lo > 1) {                       // the parameter the block's range starts in
      const int mid = (lo + hi) >> 1;
      if (seg_off[mid] <= 4 * i0) lo = mid; else hi = mid;
    }
    seg = seg_first = lo;
    __syncthreads();
  }
  auto flush = [&]() {
    if (mx == 0) return;
    if (seg - seg_first < kSlots) atomicMax(&smax[seg - seg_first], mx);
    else amax_publish(seg_amax + (int64_t)seg * kAmaxSlots, mx, epoch, 1, blockIdx.x);
    mx = 0;
  };
  for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    adam_elem(pv.x, mv.x, vv.x, gv.x, step, decay, h);
    adam_elem(pv.y, mv.y, vv.y, gv.y, step, decay, h);
    adam_elem(pv.z, mv.z, vv.z, gv.z, step, decay, h);
    adam_elem(pv.w, mv.w, vv.w, gv.w, step, decay, h);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (seg_amax) {
      while (seg + 1 < nseg && 4 * i >= seg_off[seg + 1]) {
        flush();
        ++seg;
      }
      mx = amax4(mx, pv.x, pv.y, pv.z, pv.w);
    }
    if (SHADOW) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t o;
      o[0] = (__bf16)pv.x; o[1] = (__bf16)pv.y; o[2] = (__bf16)pv.z; o[3] = (__bf16)pv.w;
      reinterpret_cast<bf16x4_t*>(shadow)[i] = o;
    }
  }
  if (seg_amax) {
    flush();
    __syncthreads();
    if (threadIdx.x < kSlots && smax[threadIdx.x] != 0 && seg_first + threadIdx.x < nseg)
      amax_publish(seg_amax + (int64_t)(seg_first + threadIdx.x) * kAmaxSlots, smax[threadIdx.x], epoch, 1, blockIdx.x);
  }
  // tail (n % 4)
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      adam_elem(p[i], m[i], v[i], g[i], step, decay, h);
      if (SHADOW) shadow[i] = (__bf16)p[i];
      if (seg_amax) amax_publish(seg_amax + (int64_t)(nseg - 1) * kAmaxSlots, finite_abs_bits(p[i]), epoch, 1, threadIdx.x);
    }
}

// ... and, when the step walks an epoch's permutation by a device-side cursor (trainer.GraphedStep), the
// cursor's move to the next batch: one launch at the end of a step instead of two.
// ... and the epoch of the magnitude records (amax.h): what the next step's kernels write outranks this step's.
__global__ void step_advance_kernel(int32_t* done, int64_t* cursor, int64_t stride, int32_t* amax_epoch) {
  *done += 1;
  if (cursor) *cursor += stride;
  if (amax_epoch) *amax_epoch += 1;
}

// ---------------------------------------------------------------------------- lazy tables
// Row strides of the moments: a row's m and v may sit side by side in ONE record (m0 = base, v0 = base + W0,
// ld_mv0 = 2 W0; the scalar table: m1 = base, v1 = base + 1, ld_mv1 = 2).  Random rows cost per ROW ACCESS, not per
// byte — 86 k random 256-byte records read in the time of 86 k 128-byte rows (tools/micro/row_width.py: 21.5 vs
// 21.0 us; two separate 128-byte rows: 42 us) — so the update touches 2 random places per row instead of 3.
struct TableGroup {
  float* p0; float* m0; float* v0; int W0; float wd0;   // main table [V, W0], W0 % 4 == 0
  float* p1; float* m1; float* v1; float wd1;           // optional scalar-per-row table [V] or null
  int32_t* last;                                        // [V] updates applied to each row
  int64_t ld_mv0, ld_mv1;                               // row stride (floats) of m0 / v0 and of m1 / v1
};

constexpr int kRowBusy = -1;

// One row's worth of work for one lane group: catch-up through `target` updates, then (grad0)
// update target+1.  gi = index of the row's gradient in grad0/grad1.
// LATE_FROM (the gradient update): `from` = last[r] is fetched HERE, together with the row's p, m, v and gradient
// instead of ahead of them — the row id leads to ONE round of random loads, not to last[r] and then to the rows
// (the kernels are chains of dependent misses: 3 round trips -> 2).
template <int LG, bool LATE_FROM = false>
__device__ inline void table_adam_row(const TableGroup& tg, int64_t r, int from, int target,
                                      int64_t gi, const float* __restrict__ grad0,
                                      const float* __restrict__ grad1,
                                      const float2* __restrict__ sched, int sched_len,
                                      const AdamHyper& h, const ReplayAux& ax, int lig) {
  float4 gfirst = make_float4(0.f, 0.f, 0.f, 0.f);
  float p1v = 0.f, m1v = 0.f, v1v = 0.f, g1v = 0.f;
  if (LATE_FROM) {
    if (lig * 4 < tg.W0) gfirst = *reinterpret_cast<const float4*>(grad0 + gi * tg.W0 + 4 * lig);
    if (tg.p1 && lig == 0) {
      p1v = tg.p1[r]; m1v = tg.m1[r * tg.ld_mv1]; v1v = tg.v1[r * tg.ld_mv1];
      g1v = grad1[gi];
    }
  }
  for (int sub = lig; sub * 4 < tg.W0; sub += LG) {
    float* pp = tg.p0 + r * tg.W0 + 4 * sub;
    float* pm = tg.m0 + r * tg.ld_mv0 + 4 * sub;
    float* pv = tg.v0 + r * tg.ld_mv0 + 4 * sub;
    float4 p = *reinterpret_cast<float4*>(pp);
    float4 m = *reinterpret_cast<float4*>(pm);
    float4 v = *reinterpret_cast<float4*>(pv);
    if (LATE_FROM && sub == lig) from = tg.last[r];
    replay4(p, m, v, from, target, sched, sched_len, tg.wd0, h, ax);
    if (grad0) {
      const float2 sc = sched[target < sched_len ? target : sched_len - 1];
      const float4 g = (LATE_FROM && sub == lig) ? gfirst
                                                 : *reinterpret_cast<const float4*>(grad0 + gi * tg.W0 + 4 * sub);
      const float decay = sc.y * tg.wd0;
      adam_elem(p.x, m.x, v.x, g.x, sc.x, decay, h);
      adam_elem(p.y, m.y, v.y, g.y, sc.x, decay, h);
      adam_elem(p.z, m.z, v.z, g.z, sc.x, decay, h);
      adam_elem(p.w, m.w, v.w, g.w, sc.x, decay, h);
    }
    *reinterpret_cast<float4*>(pp) = p;
    *reinterpret_cast<float4*>(pm) = m;
    *reinterpret_cast<float4*>(pv) = v;
  }
  if (tg.p1 && lig == 0) {
    float p, m, v;
    if (LATE_FROM) { p = p1v; m = m1v; v = v1v; }
    else { p = tg.p1[r]; m = tg.m1[r * tg.ld_mv1]; v = tg.v1[r * tg.ld_mv1]; }
    int s = from;
    if (ax.rows > 3 && target < ax.len) {
      if (target > from) {
        ReplayCoef c;
        replay_coef(from, target, ax, tg.wd1 != 0.f, c);
        replay_elem_closed(p, m, v, c, h.eps);
      }
      s = target;
    }
    for (; s < target; ++s) {
      const float2 sc = sched[s < sched_len ? s : sched_len - 1];
      if (adam_elem_zero_grad(p, m, v, sc.x, sc.y * tg.wd1, h)) { ++s; break; }
    }
    if (s < target) {
      float fp, fm, fv;
      closed_form_tail(s, target, ax, tg.wd1 != 0.f, fp, fm, fv);
      p *= fp; m *= fm; v *= fv;
    }
    if (grad1) {
      const float2 sc = sched[target < sched_len ? target : sched_len - 1];
      adam_elem(p, m, v, LATE_FROM ? g1v : grad1[gi], sc.x, sc.y * tg.wd1, h);
    }
    tg.p1[r] = p; tg.m1[r * tg.ld_mv1] = m; tg.v1[r * tg.ld_mv1] = v;
  }
  // the row's lanes sit in one wave and have all read `from` already (program order)
  if (lig == 0) tg.last[r] = grad0 ? target + 1 : target;
}

// rows == nullptr: rows are row_begin + i (range sweep / flush), count = n_rows.
// rows != nullptr: rows[i], count = *n_rows_dev (device, <= capacity n_rows).
// grad == nullptr: catch-up only, rows become current through `*done` updates.
// grad != nullptr: catch-up through *done, then apply update *done+1 with grad rows
//                  grad0[i, :] (and grad1[i]); last = *done + 1.
template <int LG>
__global__ void __launch_bounds__(256) table_adam_kernel(TableGroup tg, const int32_t* __restrict__ rows,
                                                         int64_t row_begin, int64_t n_rows,
                                                         const int32_t* __restrict__ n_rows_dev,
                                                         const float* __restrict__ grad0,
                                                         const float* __restrict__ grad1,
                                                         const float2* __restrict__ sched,
                                                         int sched_len, const int32_t* __restrict__ done,
                                                         AdamHyper h, ReplayAux ax) {
  const int64_t count = n_rows_dev ? (int64_t)*n_rows_dev : n_rows;
  const int target = *done;
  const int lig = threadIdx.x % LG;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG; i < count;
       i += ((int64_t)gridDim.x * blockDim.x) / LG) {
    const int64_t r = rows ? (int64_t)rows[i] : row_begin + i;
    if (r < 0) continue;                 // padding entry of a gathered gradient list (mapx.parallel)
    if (grad0) {
      table_adam_row<LG, true>(tg, r, 0, target, i, grad0, grad1, sched, sched_len, h, ax, lig);
      continue;
    }
    const int from = tg.last[r];
    if (from >= target) continue;
    table_adam_row<LG>(tg, r, from, target, i, grad0, grad1, sched, sched_len, h, ax, lig);
  }
}

// Catch-up straight from the RAW id list of a batch (rows may repeat; no sort is needed
// before the forward pass).  Phase 1, one entry per lane: a stale row is claimed with one
// atomicCAS on last[row] (old value -> BUSY; repeats and other blocks lose and skip) and
// pushed on the block's LDS work list.  Rows touched by the previous step are already
// current and cost one read of last[], so the hot ids never reach the CAS.  Phase 2: the
// block's lane groups drain the list, dense, so a wave is not held up by skipped entries.
template <int LG>
__global__ void __launch_bounds__(256) table_adam_raw_kernel(TableGroup tg, const int32_t* __restrict__ rows,
                                                             int64_t n_rows,
                                                             const float2* __restrict__ sched,
                                                             int sched_len, const int32_t* __restrict__ done,
                                                             AdamHyper h, ReplayAux ax) {
  __shared__ int list_row[256];
  __shared__ int list_from[256];
  __shared__ int list_n;
  const int target = *done;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < n_rows; base += (int64_t)gridDim.x * 256) {
    if (threadIdx.x == 0) list_n = 0;
    __syncthreads();
    const int64_t i = base + threadIdx.x;
    if (i < n_rows) {
      const int r = rows[i];
      const int from = tg.last[r];
      if (from >= 0 && from < target && atomicCAS(&tg.last[r], from, kRowBusy) == from) {
        const int k = atomicAdd(&list_n, 1);
        list_row[k] = r;
        list_from[k] = from;
      }
    }
    __syncthreads();
    const int n = list_n;
    for (int k = threadIdx.x / LG; k < n; k += 256 / LG)
      table_adam_row<LG>(tg, list_row[k], list_from[k], target, 0, nullptr, nullptr, sched, sched_len,
                         h, ax, threadIdx.x % LG);
    __syncthreads();
  }
}


}  // namespace mapx

extern "C" int mapx_adamw_dense(float* p, const float* g, float* m, float* v, int64_t n,
                                const float* sched, int sched_len, const int32_t* done,
                                double beta1, double beta2, double eps, double weight_decay,
                                const int64_t* seg_off_opt, int nseg, void* seg_amax_opt, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(p && g && m && v && sched && done && n >= 0 && sched_len > 0, "adamw_dense: bad arguments");
  MAPX_REQUIRE(!seg_amax_opt || (seg_off_opt && nseg >= 1), "adamw_dense: magnitude records need the segment table");
  MAPX_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                   ((uintptr_t)v % 16 == 0),
               "adamw_dense: pointers must be 16-byte aligned");
  if (n == 0) return MAPX_OK;
  hipLaunchKernelGGL(adamw_dense_kernel<false>, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, stream, p, g, m,
                     v, n, reinterpret_cast<const float2*>(sched), sched_len, done,
                     make_hyper(beta1, beta2, eps), (float)weight_decay, (__bf16*)nullptr, seg_off_opt, nseg,
                     static_cast<amax_rec*>(seg_amax_opt), amax_epoch_ptr());
  return check_launch("adamw_dense");
}

extern "C" int mapx_adamw_dense_shadow(float* p, const float* g, float* m, float* v, int64_t n,
                                       const float* sched, int sched_len, const int32_t* done,
                                       double beta1, double beta2, double eps, double weight_decay,
                                       mapx_bf16* shadow, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(p && g && m && v && sched && done && shadow && n >= 0 && sched_len > 0, "adamw_dense_shadow: bad arguments");
  MAPX_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                   ((uintptr_t)v % 16 == 0) && ((uintptr_t)shadow % 8 == 0),
               "adamw_dense_shadow: pointers must be 16-byte (shadow: 8-byte) aligned");
  if (n == 0) return MAPX_OK;
  hipLaunchKernelGGL(adamw_dense_kernel<true>, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, stream, p, g, m,
                     v, n, reinterpret_cast<const float2*>(sched), sched_len, done,
                     make_hyper(beta1, beta2, eps), (float)weight_decay, reinterpret_cast<__bf16*>(shadow),
                     (const int64_t*)nullptr, 0, (amax_rec*)nullptr, (const int32_t*)nullptr);
  return check_launch("adamw_dense_shadow");
}

extern "C" int mapx_step_advance(int32_t* done, int64_t* cursor_opt, int64_t cursor_stride, hipStream_t stream) {
  MAPX_REQUIRE(done, "step_advance: null");
  hipLaunchKernelGGL(mapx::step_advance_kernel, dim3(1), dim3(1), 0, stream, done, cursor_opt, cursor_stride,
                     const_cast<int32_t*>(mapx::amax_epoch_ptr()));
  return mapx::check_launch("step_advance");
}

namespace mapx {
// coef[(decayed ? 0 : len) + from] = replay_coef(from, *done) for every from < *done (lazy_adam.h: LazyRows)
__global__ void __launch_bounds__(256) replay_coef_table_kernel(ReplayAux ax, const int32_t* __restrict__ done,
                                                                float* __restrict__ coef) {
  const int to = *done;
  if (!(ax.rows > 3 && to < ax.len)) return;           // (gaps past the schedule tables: readers replay step by step)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * (int64_t)to;
       i += (int64_t)gridDim.x * blockDim.x) {
    const bool decayed = i < to;
    const int from = (int)(decayed ? i : i - to);
    ReplayCoef c;
    replay_coef(from, to, ax, decayed, c);
    float4* q = reinterpret_cast<float4*>(coef + ((size_t)(decayed ? 0 : ax.len) + from) * kCoefFloats);
    q[0] = make_float4(c.fp, c.fm, c.fv, c.T[0]);
    q[1] = make_float4(c.T[1], c.T[2], c.T[3], c.T[4]);
    q[2] = make_float4(c.T[5], c.T[6], 0.f, 0.f);
  }
}
}  // namespace mapx

extern "C" size_t mapx_replay_coef_table_bytes(int aux_len) {
  return (size_t)2 * (size_t)(aux_len > 0 ? aux_len : 0) * mapx::kCoefFloats * sizeof(float);
}

extern "C" int mapx_replay_coef_table(const double* aux, int aux_len, int aux_rows, double beta1, double beta2,
                                      const int32_t* done, float* coef, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(aux && done && coef && aux_len > 1 && (uintptr_t)coef % 16 == 0, "replay_coef_table: bad arguments");
  MAPX_REQUIRE(aux_rows == 3 + 2 * (kJ + 1), "replay_coef_table: the closed form needs the %d-row tables", 3 + 2 * (kJ + 1));
  const double beta = sqrt(beta2);
  const ReplayAux ax{aux, aux_len, aux_rows, beta1 / beta, 1.0 / beta};
  hipLaunchKernelGGL(replay_coef_table_kernel, dim3(64), dim3(256), 0, stream, ax, done, coef);
  return check_launch("replay_coef_table");
}

extern "C" int mapx_table_adam(float* p0, float* m0, float* v0, int64_t ld_mv0, int W0, float wd0, float* p1,
                               float* m1, float* v1, int64_t ld_mv1, float wd1, int32_t* last,
                               const int32_t* rows, int64_t row_begin, int64_t n_rows,
                               const int32_t* n_rows_dev, const float* grad0, const float* grad1,
                               const float* sched, int sched_len, const int32_t* done,
                               const double* aux, int aux_len, int aux_rows, double beta1, double beta2,
                               double eps, int rows_may_repeat, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(p0 && m0 && v0 && last && sched && done && aux && aux_len > 1,
               "table_adam: null pointer");
  MAPX_REQUIRE(W0 > 0 && W0 % 4 == 0, "table_adam: row width %d must be a multiple of 4", W0);
  MAPX_REQUIRE(ld_mv0 >= W0 && ld_mv0 % 4 == 0 && (uintptr_t)m0 % 16 == 0 && (uintptr_t)v0 % 16 == 0 &&
                   (uintptr_t)p0 % 16 == 0,
               "table_adam: moment rows must be 16-byte aligned (ld_mv0 %% 4 == 0, >= W0)");
  MAPX_REQUIRE(!p1 || ld_mv1 >= 1, "table_adam: ld_mv1 < 1");
  MAPX_REQUIRE(!p1 || (m1 && v1), "table_adam: secondary state missing");
  MAPX_REQUIRE(!(grad0 && p1) || grad1, "table_adam: secondary gradient missing");
  MAPX_REQUIRE(!rows_may_repeat || (rows && !grad0), "table_adam: repeated rows only in catch-up mode");
  if (n_rows <= 0) return MAPX_OK;
  TableGroup tg{p0, m0, v0, W0, wd0, p1, m1, v1, wd1, last, ld_mv0, ld_mv1};
  const int lg = (W0 <= 16) ? 4 : (W0 <= 32 ? 8 : 16);
  const int grid = grid_for(n_rows * lg, 256, 4096);
  const AdamHyper h = make_hyper(beta1, beta2, eps);
  const float2* sc = reinterpret_cast<const float2*>(sched);
  MAPX_REQUIRE(aux_rows == 3 || aux_rows == 3 + 2 * (kJ + 1), "table_adam: aux must have 3 or %d rows",
               3 + 2 * (kJ + 1));
  const double beta = sqrt(beta2);
  const ReplayAux ax{aux, aux_len, aux_rows, beta1 / beta, 1.0 / beta};
#define MAPX_TA(LG_)                                                                              \
  do {                                                                                            \
    if (rows_may_repeat)                                                                          \
      hipLaunchKernelGGL(table_adam_raw_kernel<LG_>, dim3(grid_for(n_rows, 256, 4096)), dim3(256), 0, \
                         stream, tg, rows, n_rows, sc, sched_len, done, h, ax);                    \
    else                                                                                          \
      hipLaunchKernelGGL(table_adam_kernel<LG_>, dim3(grid), dim3(256), 0, stream, tg, rows,       \
                         row_begin, n_rows, n_rows_dev, grad0, grad1, sc, sched_len, done, h, ax);  \
  } while (0)
  if (lg == 4) MAPX_TA(4);
  else if (lg == 8) MAPX_TA(8);
  else MAPX_TA(16);
#undef MAPX_TA
  return check_launch("table_adam");
}
