// Embedding row gather (reference code/layers.py:97-102 -> nn.Embedding forward, flattened
// to [B, F*E] in models.py:308) and its helpers.  HBM-bound: 8 B of id + 2*E*4 B per row.
#include "../../include/mapx_hip.h"
#include "amax.h"
#include "common.h"
#include "lazy_adam.h"

namespace mapx {

// One float4 per thread: E/4 consecutive lanes cover one row (E = 16 -> 4 lanes x 16 B =
// one 64-B segment), 16 rows per wave-instruction; ids are re-read by the E/4 lanes of a
// row from the same cache line.  Out-of-range ids set *err and produce zeros (the
// reference raises IndexError on CPU; the host layer turns *err into the same exception).
template <int VEC>
__global__ void __launch_bounds__(256) emb_gather_kernel(const int64_t* __restrict__ ids,
                                                         int64_t n, const float* __restrict__ table,
                                                         int64_t V, int E, float* __restrict__ out,
                                                         int* __restrict__ err, amax_rec* __restrict__ amax_out,
                                                         const int32_t* __restrict__ epoch) {
  const int per_row = E / VEC;
  const int64_t total = n * per_row;
  uint32_t amx = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / per_row;
    const int c = (int)(t - row * per_row) * VEC;
    const int64_t id = ids[row];
    const bool ok = (id >= 0) & (id < V);
    if (!ok && err) atomicOr(err, 1);
    if (VEC == 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(table + id * E + c);
      *reinterpret_cast<float4*>(out + row * E + c) = v;
      amx = amax4(amx, v.x, v.y, v.z, v.w);
    } else {
      const float v = ok ? table[id * E + c] : 0.f;
      out[row * E + c] = v;
      amx = max(amx, finite_abs_bits(v));
    }
  }
  if (amax_out) amax_publish_block(amax_out, amx, epoch);      // max |out| for the products that read it (amax.h)
}

// The gather of a training step without a catch-up pass in front of it (lazy_adam.h: LazyRows): last[id] is read beside
// the row and a stale row's missing zero-gradient updates are replayed in registers from its m | v record — the bits
// the catch-up pass would have stored — and nothing is written; the gradient update that ends the step is the row's
// one read-modify-write.  OUT = float or __bf16 (8 columns per thread then).
template <class OUT>
__global__ void __launch_bounds__(256) emb_gather_lazy_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                              const float* __restrict__ table, int64_t V, int E,
                                                              OUT* __restrict__ out, int* __restrict__ err,
                                                              amax_rec* __restrict__ amax_out,
                                                              const int32_t* __restrict__ epoch, LazyRows lz) {
  constexpr bool kHalf = sizeof(OUT) == 2;
  constexpr int CPT = kHalf ? 8 : 4;                       // columns per thread
  const int per_row = E / CPT;
  const int64_t total = n * per_row;
  const int target = *lz.done;
  uint32_t amx = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / per_row;
    const int c = (int)(t - row * per_row) * CPT;
    const int64_t id = ids[row];
    const bool ok = (id >= 0) & (id < V);
    if (!ok && err) atomicOr(err, 1);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (ok) {
      const int from = lz.last[id];
      a = *reinterpret_cast<const float4*>(table + id * E + c);
      if (kHalf) b = *reinterpret_cast<const float4*>(table + id * E + c + 4);
      if (from >= 0 && from < target) {
        const float* pm = lz.m0 + id * lz.ld_mv0 + c;
        const float* pv = lz.v0 + id * lz.ld_mv0 + c;
        const float4 m0 = *reinterpret_cast<const float4*>(pm), v0 = *reinterpret_cast<const float4*>(pv);
        float4 m1 = m0, v1 = v0;
        if (kHalf) { m1 = *reinterpret_cast<const float4*>(pm + 4); v1 = *reinterpret_cast<const float4*>(pv + 4); }
        lazy_replay4(lz, a, m0, v0, from, target);
        if (kHalf) lazy_replay4(lz, b, m1, v1, from, target);
      }
    }
    if constexpr (kHalf) {
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      bf16x8_t o;
      o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
      o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
      *reinterpret_cast<bf16x8_t*>(out + row * E + c) = o;
    } else {
      *reinterpret_cast<float4*>(out + row * E + c) = a;
      amx = amax4(amx, a.x, a.y, a.z, a.w);
    }
  }
  if (!kHalf && amax_out) amax_publish_block(amax_out, amx, epoch);
}

// The same gather with bf16 output rows (bf16 compute mode: the trunk's first GEMMs read bf16; the
// table stays fp32).  8 columns per thread: two 16-B table reads, one 16-B store.
__global__ void __launch_bounds__(256) emb_gather_bf16_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                              const float* __restrict__ table, int64_t V, int E,
                                                              __bf16* __restrict__ out, int* __restrict__ err) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  const int per_row = E / 8;
  const int64_t total = n * per_row;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / per_row;
    const int c = (int)(t - row * per_row) * 8;
    const int64_t id = ids[row];
    const bool ok = (id >= 0) & (id < V);
    if (!ok && err) atomicOr(err, 1);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (ok) {
      a = *reinterpret_cast<const float4*>(table + id * E + c);
      b = *reinterpret_cast<const float4*>(table + id * E + c + 4);
    }
    bf16x8_t o;
    o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
    o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
    *reinterpret_cast<bf16x8_t*>(out + row * E + c) = o;
  }
}

// int64 ids -> int32 keys for the sort / segment machinery (V < 2^31), range-checked.
__global__ void __launch_bounds__(256) ids_to_i32_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                         int64_t V, int32_t* __restrict__ out,
                                                         int* __restrict__ err) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = ids[t];
    const bool ok = (id >= 0) & (id < V);
    if (!ok && err) atomicOr(err, 1);
    out[t] = ok ? (int32_t)id : 0;
  }
}

}  // namespace mapx

namespace mapx {
// mapx_lazy_rows -> LazyRows (false + error set: incomplete); W = row width the state must cover
bool lazy_rows_from(const mapx_lazy_rows* q, int W, const char* what, LazyRows* lz) {
  if (!(q->m0 && q->v0 && q->last && q->sched && q->done && q->aux && q->coef_opt && q->aux_len > 1 && q->sched_len > 0 &&
        (!q->m1) == (!q->v1) && q->ld_mv0 >= W && q->ld_mv0 % 4 == 0 && (uintptr_t)q->m0 % 16 == 0 &&
        (uintptr_t)q->v0 % 16 == 0 && (!q->m1 || q->ld_mv1 >= 1) && (uintptr_t)q->coef_opt % 16 == 0 &&
        q->aux_rows == 3 + 2 * (kJ + 1))) {
    set_error("%s: incomplete lazy-row state (17-row aux and the coefficient table are required)", what);
    return false;
  }
  const double beta = sqrt(q->beta2);
  *lz = LazyRows{q->m0, q->v0, q->ld_mv0, q->wd0, q->m1, q->v1, q->ld_mv1, q->wd1, q->last,
                 reinterpret_cast<const float2*>(q->sched), q->sched_len, q->done,
                 make_hyper(q->beta1, q->beta2, q->eps),
                 ReplayAux{q->aux, q->aux_len, q->aux_rows, q->beta1 / beta, 1.0 / beta}, q->coef_opt};
  return true;
}
}  // namespace mapx

extern "C" int mapx_emb_gather_fwd(const int64_t* ids, int64_t n, const float* table, int64_t V,
                                   int E, float* out, int* err_flag, void* amax_out_opt,
                                   const mapx_lazy_rows* lazy_opt, hipStream_t stream) {
  MAPX_REQUIRE(n >= 0 && V > 0 && E > 0, "emb_gather_fwd: bad sizes n=%lld V=%lld E=%d",
               (long long)n, (long long)V, E);
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(ids && table && out, "emb_gather_fwd: null pointer");
  const bool vec = (E % 4 == 0) && ((uintptr_t)table % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const int64_t total = n * (vec ? E / 4 : E);
  static const int cap = [] { const char* e = getenv("MAPX_GATHER_GRID"); return e ? atoi(e) : 512; }();   // (2048 blocks: 6.2 us with the record's block reduction, 512: 5.0)
  const int grid = mapx::grid_for(total, 256, cap);
  if (lazy_opt) {
    MAPX_REQUIRE(vec, "emb_gather_fwd: rows are read through their pending updates for 16-byte rows only (E %% 4 == 0)");
    mapx::LazyRows lz;
    if (!mapx::lazy_rows_from(lazy_opt, E, "emb_gather_fwd", &lz)) return MAPX_EINVAL;
    hipLaunchKernelGGL(mapx::emb_gather_lazy_kernel<float>, dim3(grid), dim3(256), 0, stream, ids, n, table, V, E, out,
                       err_flag, static_cast<mapx::amax_rec*>(amax_out_opt), mapx::amax_epoch_ptr(), lz);
    return mapx::check_launch("emb_gather_fwd");
  }
  if (vec)
    hipLaunchKernelGGL(mapx::emb_gather_kernel<4>, dim3(grid), dim3(256), 0, stream, ids, n, table,
                       V, E, out, err_flag, static_cast<mapx::amax_rec*>(amax_out_opt), mapx::amax_epoch_ptr());
  else
    hipLaunchKernelGGL(mapx::emb_gather_kernel<1>, dim3(grid), dim3(256), 0, stream, ids, n, table,
                       V, E, out, err_flag, static_cast<mapx::amax_rec*>(amax_out_opt), mapx::amax_epoch_ptr());
  return mapx::check_launch("emb_gather_fwd");
}

extern "C" int mapx_emb_gather_fwd_bf16(const int64_t* ids, int64_t n, const float* table, int64_t V, int E,
                                        mapx_bf16* out, int* err_flag, const mapx_lazy_rows* lazy_opt,
                                        hipStream_t stream) {
  MAPX_REQUIRE(n >= 0 && V > 0 && E > 0 && E % 8 == 0, "emb_gather_fwd_bf16: bad sizes n=%lld V=%lld E=%d (E %% 8 == 0)",
               (long long)n, (long long)V, E);
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(ids && table && out && (uintptr_t)table % 16 == 0 && (uintptr_t)out % 16 == 0,
               "emb_gather_fwd_bf16: null or unaligned pointer");
  if (lazy_opt) {
    mapx::LazyRows lz;
    if (!mapx::lazy_rows_from(lazy_opt, E, "emb_gather_fwd_bf16", &lz)) return MAPX_EINVAL;
    hipLaunchKernelGGL(mapx::emb_gather_lazy_kernel<__bf16>, dim3(mapx::grid_for(n * (E / 8), 256)), dim3(256), 0, stream,
                       ids, n, table, V, E, reinterpret_cast<__bf16*>(out), err_flag, (mapx::amax_rec*)nullptr,
                       (const int32_t*)nullptr, lz);
    return mapx::check_launch("emb_gather_fwd_bf16");
  }
  hipLaunchKernelGGL(mapx::emb_gather_bf16_kernel, dim3(mapx::grid_for(n * (E / 8), 256)), dim3(256), 0, stream, ids, n,
                     table, V, E, reinterpret_cast<__bf16*>(out), err_flag);
  return mapx::check_launch("emb_gather_fwd_bf16");
}

extern "C" int mapx_ids_to_i32(const int64_t* ids, int64_t n, int64_t V, int32_t* out,
                               int* err_flag, hipStream_t stream) {
  MAPX_REQUIRE(n >= 0 && V > 0 && V < (1LL << 31), "ids_to_i32: bad sizes");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(ids && out, "ids_to_i32: null pointer");
  hipLaunchKernelGGL(mapx::ids_to_i32_kernel, dim3(mapx::grid_for(n, 256)), dim3(256), 0, stream,
                     ids, n, V, out, err_flag);
  return mapx::check_launch("ids_to_i32");
}
