// Deterministic reduce-by-key of gradient rows ("sparse embedding grad"): replaces the
// dense [V,E] zero-fill + index_add of aten::embedding_dense_backward / index_select
// backward (reference code/layers.py:86, nce/index_linear.py:99-100).
//
// A SegPlan (segplan.hip) gives, for n keys: the sorted order `perm`, the 1-based segment
// rank of every sorted position, and the segment starts.  Reduction is chunk-aligned so
// that skewed keys (id 3 = <mask> receives B*L rows, 2-value fields receive ~B/2 each)
// cost no more than uniform ones:
//   pass A  a group of W/4 lanes walks CH consecutive SORTED positions, accumulating in
//           registers and flushing whenever the segment rank changes.  Segments that lie
//           inside the chunk go straight to out[rank-1]; the piece that continues from
//           the previous chunk goes to part_head[chunk], the piece that continues into
//           the next chunk to part_tail[chunk].
//   pass B  one wave per OWNER chunk (the chunk whose tail piece starts a spanning segment;
//           pass A lists them): tail + the following chunks' head pieces, strided over the
//           wave's lane groups and combined in a fixed order.
// No float atomics anywhere: sums are bit-reproducible, so data-parallel replicas that
// apply the same merged gradient stay bit-identical.
#pragma once
#include <cstdlib>

#include "common.h"

namespace mapx {

constexpr int kSegChunk = 32;  // sorted positions per lane group in pass A ...
// ... and 16 for small problems (round 3): the embedding gradient's 94 k positions are 46 workgroups at 32 — a chain of
// dependent misses per group with most of the chip idle (19 us); at 16 they are 92 and the walk is half as long
// (13.5 us).  The NCE table's 639 k positions lose at 16 (37 -> 48 us: twice the partial rows and owners).
constexpr int kSegChunkSmall = 16;
constexpr int64_t kSegSmallN = 1 << 18;
inline int seg_chunk_for(int64_t n) {
  static const int64_t small_n = [] { const char* e = getenv("MAPX_SEG_SMALL_N"); return e ? atoll(e) : kSegSmallN; }();
  return n <= small_n ? kSegChunkSmall : kSegChunk;
}

struct SegPlanView {
  int64_t n;
  const int32_t* perm;       // [n]  sorted position -> original position
  const int32_t* rank;       // [n]  1-based segment id of each sorted position
  const int32_t* seg_start;  // [U+1] first sorted position of each segment; [U] = n
};

// Contribution functors: value of row `p` (original position), float4 column `sub`.
struct RowsContrib {
  const float* src;  // [n, W]
  int W;
  __device__ inline void prepare() {}
  __device__ inline float4 operator()(int32_t p, int sub, float& extra) const {
    extra = 0.f;
    return *reinterpret_cast<const float4*>(src + (int64_t)p * W + 4 * sub);
  }
};

// Rows that reach the table as the SUM of two gradient tensors (DCNv2: dL/dX0 of the deep tower and of the
// cross tower): a + b is formed here, element by element, instead of by an elementwise launch in front of
// the reduction — the same fp32 additions in the same order, so the result is bit-identical.
struct Rows2Contrib {
  const float* src;   // [n, W]
  const float* src2;  // [n, W]
  int W;
  __device__ inline void prepare() {}
  __device__ inline float4 operator()(int32_t p, int sub, float& extra) const {
    extra = 0.f;
    const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)p * W + 4 * sub);
    const float4 b = *reinterpret_cast<const float4*>(src2 + (int64_t)p * W + 4 * sub);
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
};

// bf16 gradient rows (bf16 compute mode: dL/dX0 leaves the trunk's backward GEMMs in bf16);
// the sums are fp32 like everything else here.
struct RowsBf16Contrib {
  const __bf16* src;  // [n, W]
  int W;
  __device__ inline void prepare() {}
  __device__ inline float4 operator()(int32_t p, int sub, float& extra) const {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    extra = 0.f;
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(src + (int64_t)p * W + 4 * sub);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
};

// Two bf16 gradient tensors summed on the fly; the sum is rounded to bf16 first, as the elementwise
// bf16 addition it replaces would have done.
struct Rows2Bf16Contrib {
  const __bf16* src;
  const __bf16* src2;
  int W;
  __device__ inline void prepare() {}
  __device__ inline float4 operator()(int32_t p, int sub, float& extra) const {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    extra = 0.f;
    const bf16x4_t a = *reinterpret_cast<const bf16x4_t*>(src + (int64_t)p * W + 4 * sub);
    const bf16x4_t b = *reinterpret_cast<const bf16x4_t*>(src2 + (int64_t)p * W + 4 * sub);
    float4 r;
    r.x = (float)(__bf16)((float)a[0] + (float)b[0]);
    r.y = (float)(__bf16)((float)a[1] + (float)b[1]);
    r.z = (float)(__bf16)((float)a[2] + (float)b[2]);
    r.w = (float)(__bf16)((float)a[3] + (float)b[3]);
    return r;
  }
};

// NCE output-table gradient, never materialised per (target, sample) pair:
// d emb[idx[t,j]] += dlogit[t,j] * h[t,:]   d bias[idx[t,j]] += dlogit[t,j]
// (backward of reference nce/index_linear.py:99-102; p = t*(K+1)+j)
// Rows plus one scalar per GROUP of consecutive positions: embedding gradient rows dL/dX[b,f,:]
// together with the LR gradient dL/dlr[b] that every field of row b shares (group = F).
struct RowsExtraContrib {
  const float* src;    // [n, ld]: the first W columns of a row are reduced
  int W;
  int64_t ld;          // row stride of src (>= W)
  const float* extra;  // scalar of position p: extra[(p / group) * estride]
  int group;
  int64_t estride;
  __device__ inline void prepare() {}
  __device__ inline float4 operator()(int32_t p, int sub, float& ex) const {
    ex = extra[(int64_t)(p / group) * estride];
    return *reinterpret_cast<const float4*>(src + (int64_t)p * ld + 4 * sub);
  }
};

struct NceContrib {
  const float* dlogit;  // [T*(K+1)]
  const float* h;       // [T, P]
  int K1, P;
  const float* gscale;  // optional device scalar: incoming gradient of the loss (dlogit is per unit loss)
  float gs;             // its value, fetched once per thread by prepare() (a load + wait per call otherwise)
  __device__ inline void prepare() { gs = gscale ? *gscale : 1.f; }
  __device__ inline float4 operator()(int32_t p, int sub, float& extra) const {
    const float d = dlogit[p] * gs;
    const int t = p / K1;
    float4 v = *reinterpret_cast<const float4*>(h + (int64_t)t * P + 4 * sub);
    extra = d;
    return make_float4(d * v.x, d * v.y, d * v.z, d * v.w);
  }
};

__device__ inline void add4(float4& a, const float4& b) {
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

// W = row width in floats (multiple of 4, <= 256).  Row layout of out/part_*: WS floats
// per row where WS = W (+4 if EXTRA: the extra scalar lives at column W).
// The chunk's 32 (rank, perm) pairs are loaded once, LG-wide, and handed round by shuffles;
// contributions are fetched 8 entries ahead of the accumulate/flush walk (which is serial by
// nature), so the random reads behind `perm` overlap instead of paying their latency 32 times.
// A chunk whose tail run continues into the next chunk appends itself to `owners`.
template <int LG, bool EXTRA, class Contrib, int BATCH = 8, int CH = kSegChunk>
__global__ void __launch_bounds__(256) seg_reduce_pass_a(SegPlanView pl, Contrib contrib, int W,
                                                         float* __restrict__ out,
                                                         float* __restrict__ out_extra,
                                                         float* __restrict__ part_head,
                                                         float* __restrict__ part_tail,
                                                         int32_t* __restrict__ owners,
                                                         int32_t* __restrict__ n_owners) {
  constexpr int NPL = CH / LG;          // entries each lane preloads; BATCH = contributions in flight per walk step
  __shared__ int wg_owners, wg_base;    // this workgroup's owner chunks: counted in LDS, appended with ONE atomic
  if (threadIdx.x == 0) wg_owners = 0;
  __syncthreads();
  int own_slot = -1;
  contrib.prepare();
  const int WS = EXTRA ? W + 4 : W;
  const int lane = threadIdx.x % kWave;
  const int lig = lane % LG, gbase = lane - lig;
  const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG;
  const int64_t ngroups = ceil_div(pl.n, CH);
  const bool active = group < ngroups;         // whole groups are active or not; shuffles need all lanes
  const int64_t j0 = active ? group * CH : 0;
  const int64_t j1 = active ? ((j0 + CH < pl.n) ? j0 + CH : pl.n) : 0;
  int myrank[NPL], myperm[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int64_t j = j0 + i * LG + lig;
    myrank[i] = (j < j1) ? pl.rank[j] : -1;
    myperm[i] = (j < j1) ? pl.perm[j] : 0;
  }
  const int prev_rank = (active && j0 > 0) ? pl.rank[j0 - 1] : -1;
  const int next_rank = (active && j1 < pl.n) ? pl.rank[j1] : -2;
  for (int sub0 = 0; sub0 * 4 < W; sub0 += LG) {   // wave-uniform rounds (shuffles inside);
    const int sub = sub0 + lig;                    // W > 4*LG: further column rounds
    const bool live = sub * 4 < W;                 // W < 4*LG: idle lanes still take part in shuffles
    int cur = __shfl(myrank[0], gbase, kWave);
    bool started_inside = (j0 == 0) || (prev_rank != cur);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accx = 0.f;
#pragma unroll
    for (int e0 = 0; e0 < CH; e0 += BATCH) {
      float4 v[BATCH];
      float ex[BATCH];
      int rk[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int e = e0 + u;
        rk[u] = __shfl(myrank[e / LG], gbase + (e % LG), kWave);
        const int p = __shfl(myperm[e / LG], gbase + (e % LG), kWave);
        // always load (p is a valid position even in the padding, idle lanes re-read column 0):
        // a load inside a conditional block is waited for at the end of that block, which
        // serialises the batch into 16 memory round trips
        v[u] = contrib(p, live ? sub : 0, ex[u]);
      }
      // Retire the batch's loads HERE, once: the walk below stores conditionally (a run ends),
      // and with a store possibly in flight the compiler cannot count how many memory operations
      // are younger than a load, so it would put `s_waitcnt vmcnt(0)` in front of every later use
      // of v[] / ex[] — each one draining the store just issued (measured: 42 us for the NCE
      // table gradient instead of 12).  After these moves the values are plain ALU results.
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3\n\tv_mov_b32 %4, %4"
                     : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w), "+v"(ex[u]));
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (rk[u] < 0) continue;                 // past the end of the array
        if (rk[u] != cur) {
          float* dst = started_inside ? out + (int64_t)(cur - 1) * W : part_head + group * WS;
          if (live) *reinterpret_cast<float4*>(dst + 4 * sub) = acc;
          if (EXTRA && sub == 0) {
            if (started_inside) out_extra[cur - 1] = accx;
            else part_head[group * WS + W] = accx;
          }
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
          accx = 0.f;
          cur = rk[u];
          started_inside = true;
        }
        add4(acc, v[u]);
        if (EXTRA) accx += ex[u];
      }
    }
    if (!active || !live) continue;
    const bool ends = (j1 == pl.n) || (next_rank != cur);
    float* dst;
    float* dstx;
    if (started_inside && ends) {
      dst = out + (int64_t)(cur - 1) * W;
      dstx = out_extra + (cur - 1);
    } else if (!started_inside) {
      dst = part_head + group * WS;
      dstx = part_head + group * WS + W;
    } else {
      dst = part_tail + group * WS;
      dstx = part_tail + group * WS + W;
      if (sub == 0) own_slot = atomicAdd(&wg_owners, 1);               // this chunk owns a spanning run
    }
    *reinterpret_cast<float4*>(dst + 4 * sub) = acc;
    if (EXTRA && sub == 0) *dstx = accx;
  }
  // The owner list through one returning atomic per WORKGROUP (its chunks' owners counted in LDS).  Rounds 1-3: one per
  // owner chunk, all on the same counter — 4.6 k same-address returning atomics per NCE step = 10 of the kernel's 30 us
  // (tools/micro/seg_reduce_floor.hip, variants C and D; measured here: 30.0 -> 19.7 us).  One per wave (2.2 k) is as
  // slow as one per chunk (28.9 us); per-chunk flags instead of a list made this pass as fast (18.5 us) and pass B,
  // which then visits every chunk, 6-9 us slower.
  __syncthreads();
  if (threadIdx.x == 0 && wg_owners > 0) wg_base = atomicAdd(n_owners, wg_owners);
  __syncthreads();
  if (own_slot >= 0) owners[wg_base + own_slot] = (int32_t)group;
}

// One wave per OWNER chunk (listed by pass A): sum = tail[c] + head[c+1] + ... + head[c_last].
template <int LG, bool EXTRA, int CH = kSegChunk>
__global__ void __launch_bounds__(256) seg_reduce_pass_b(SegPlanView pl, int W,
                                                         float* __restrict__ out,
                                                         float* __restrict__ out_extra,
                                                         const float* __restrict__ part_head,
                                                         const float* __restrict__ part_tail,
                                                         const int32_t* __restrict__ owners,
                                                         const int32_t* __restrict__ n_owners) {
  const int WS = EXTRA ? W + 4 : W;
  const int lane = threadIdx.x % kWave;
  const int nown = *n_owners;
  for (int64_t oi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave; oi < nown;
       oi += ((int64_t)gridDim.x * blockDim.x) / kWave) {
  const int64_t c = owners[oi];
  const int64_t j1 = c * CH + CH;
  const int cur = pl.rank[j1 - 1];
  const int64_t c_last = (pl.seg_start[cur] - 1) / CH;
  constexpr int G = kWave / LG;                 // lane groups per wave
  const int g = lane / LG, lig = lane % LG;
  for (int sub0 = 0; sub0 * 4 < W; sub0 += LG) {   // wave-uniform trip count (shuffles inside)
    const int sub = sub0 + lig;
    const bool live = sub * 4 < W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accx = 0.f;
    if (live) {
      // A hot key's run spans hundreds of chunks (the most frequent feature value draws ~2 % of all noise
      // samples) and ONE wave walks it: the walk's length in memory round trips is this kernel's duration.
      // 16 loads in flight per lane group (4 before: 12-19 us for the NCE table inside the step), added in order.
      constexpr int NF = 16;
      int64_t cc = c + 1 + g;
      for (; cc + (NF - 1) * G <= c_last; cc += NF * G) {
        float4 l[NF];
        float x[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
          l[u] = *reinterpret_cast<const float4*>(part_head + (cc + u * G) * WS + 4 * sub);
          x[u] = (EXTRA && sub == 0) ? part_head[(cc + u * G) * WS + W] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NF; ++u) {
          add4(acc, l[u]);
          accx += x[u];
        }
      }
      for (; cc + 3 * G <= c_last; cc += 4 * G) {   // 4 loads in flight, added in order
        const float4 l0 = *reinterpret_cast<const float4*>(part_head + cc * WS + 4 * sub);
        const float4 l1 = *reinterpret_cast<const float4*>(part_head + (cc + G) * WS + 4 * sub);
        const float4 l2 = *reinterpret_cast<const float4*>(part_head + (cc + 2 * G) * WS + 4 * sub);
        const float4 l3 = *reinterpret_cast<const float4*>(part_head + (cc + 3 * G) * WS + 4 * sub);
        float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
        if (EXTRA && sub == 0) {
          x0 = part_head[cc * WS + W]; x1 = part_head[(cc + G) * WS + W];
          x2 = part_head[(cc + 2 * G) * WS + W]; x3 = part_head[(cc + 3 * G) * WS + W];
        }
        add4(acc, l0); add4(acc, l1); add4(acc, l2); add4(acc, l3);
        accx += x0; accx += x1; accx += x2; accx += x3;
      }
      for (; cc <= c_last; cc += G) {
        add4(acc, *reinterpret_cast<const float4*>(part_head + cc * WS + 4 * sub));
        if (EXTRA && sub == 0) accx += part_head[cc * WS + W];
      }
    }
    // combine the G strided partial sums in group order 0..G-1 (fixed -> reproducible)
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    float totx = 0.f;
    if (g == 0 && live) {
      tot = *reinterpret_cast<const float4*>(part_tail + c * WS + 4 * sub);
      if (EXTRA && sub == 0) totx = part_tail[c * WS + W];
    }
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      const int src = gg * LG + lig;
      tot.x += __shfl(acc.x, src, kWave);
      tot.y += __shfl(acc.y, src, kWave);
      tot.z += __shfl(acc.z, src, kWave);
      tot.w += __shfl(acc.w, src, kWave);
      if (EXTRA) totx += __shfl(accx, src, kWave);
    }
    if (g == 0 && live) {
      *reinterpret_cast<float4*>(out + (int64_t)(cur - 1) * W + 4 * sub) = tot;
      if (EXTRA && sub == 0) out_extra[cur - 1] = totx;
    }
  }
  }
}

// Bytes of partial storage seg_reduce needs for n keys of row width W (+extra).
inline size_t seg_reduce_partial_bytes(int64_t n, int W, bool extra) {
  const int WS = extra ? W + 4 : W;
  const size_t chunks = (size_t)ceil_div(n > 0 ? n : 1, seg_chunk_for(n));
  return chunks * WS * sizeof(float) * 2 + (chunks + 64) * sizeof(int32_t);   // + owner list, counter
}

template <bool EXTRA, class Contrib>
int seg_reduce_launch(const SegPlanView& pl, const Contrib& contrib, int W, float* out,
                      float* out_extra, void* ws, size_t ws_bytes, int32_t* zeroed_counter, hipStream_t stream,
                      const char* what) {
  if (pl.n == 0) return MAPX_OK;
  MAPX_REQUIRE(W % 4 == 0 && W >= 4 && W <= 256, "%s: row width %d must be a multiple of 4", what, W);
  const size_t need = seg_reduce_partial_bytes(pl.n, W, EXTRA);
  if (ws_bytes < need || !ws) {
    set_error("%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    return MAPX_EWORKSPACE;
  }
  const int WS = EXTRA ? W + 4 : W;
  const int ch = seg_chunk_for(pl.n);
  const int64_t nchunks = ceil_div(pl.n, ch);
  float* part_head = static_cast<float*>(ws);
  float* part_tail = part_head + nchunks * WS;
  int32_t* n_owners = reinterpret_cast<int32_t*>(part_tail + nchunks * WS);
  int32_t* owners = n_owners + 16;
  if (zeroed_counter) {
    n_owners = zeroed_counter;       // the plan's own counter, zeroed when the plan was built: no memset launch
  } else if (hipMemsetAsync(n_owners, 0, sizeof(int32_t), stream) != hipSuccess) {
    set_error("%s: memset failed", what);
    return MAPX_EHIP;
  }
  // lane-group width: 4 lanes for 16-float rows, 8 for 32-float rows, 16 beyond
  const int lg = (W <= 16) ? 4 : (W <= 32 ? 8 : 16);
  const int64_t threads_a = nchunks * lg;
  const int grid_a = (int)ceil_div(threads_a, 256);
  int64_t gb = ceil_div(nchunks * kWave, 256);
  const int grid_b = (int)(gb > 1024 ? 1024 : gb);
  // (8 contributions in flight per walk step; 16 and 32 were measured and change nothing: 38.3 / 38.8 / 42.7 us for
  // the NCE table's reduction, 20.4 / 22.5 / 17.9 us for the embedding's — the walk is not bound by its loads in flight)
#define MAPX_SEG_LAUNCH2(LG_, CH_)                                                                      \
  hipLaunchKernelGGL((seg_reduce_pass_a<LG_, EXTRA, Contrib, 8, CH_>), dim3(grid_a), dim3(256), 0,         \
                     stream, pl, contrib, W, out, out_extra, part_head, part_tail, owners,               \
                     n_owners);                                                                           \
  hipLaunchKernelGGL((seg_reduce_pass_b<LG_, EXTRA, CH_>), dim3(grid_b), dim3(256), 0, stream, pl,        \
                     W, out, out_extra, part_head, part_tail, owners, n_owners)
#define MAPX_SEG_LAUNCH(LG_)                                                                              \
  do {                                                                                                    \
    if (ch == kSegChunkSmall) { MAPX_SEG_LAUNCH2(LG_, kSegChunkSmall); }                                  \
    else { MAPX_SEG_LAUNCH2(LG_, kSegChunk); }                                                            \
  } while (0)
  if (lg == 4) { MAPX_SEG_LAUNCH(4); }
  else if (lg == 8) { MAPX_SEG_LAUNCH(8); }
  else { MAPX_SEG_LAUNCH(16); }
#undef MAPX_SEG_LAUNCH
#undef MAPX_SEG_LAUNCH2
  return check_launch(what);
}

}  // namespace mapx
