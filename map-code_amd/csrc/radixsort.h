// Stable LSD radix sort of (int32 key, int32 value = original position) pairs, 8 bits per
// pass, for the segment plans (n <= ~1 M keys of <= 26 bits).  rocPRIM's device sort picks
// block-sort + ~13 merge launches per sort at these sizes (0.11 ms per call, launch-latency
// bound); three passes of {histogram, scan, scatter} do the same job in short launches.
//
//   histogram  one block per tile of 4096 keys: LDS histogram of the pass digit, written
//              digit-major [digit][block] so that an exclusive scan gives every (digit, block)
//              its base offset;
//   scatter    the block re-reads its tile; each of the 4 waves ranks its own contiguous
//              quarter with wave ballots (lanes holding equal digits find each other in 8
//              ballots, rank = popcount of lower lanes) on a private LDS counter row, so no
//              barrier is needed inside the ranking; after one barrier every item knows
//              offset[digit][block] + counts of earlier waves + its own rank.  Equal keys keep
//              their input order (stable), which makes the later reduce-by-key deterministic.
#pragma once
#include "common.h"

namespace mapx {

constexpr int kSortTile = 4096;   // keys per block
constexpr int kSortItems = 16;    // per thread
constexpr int kSortBits = 8;      // digit width (12-bit digits: 4096x156 scattered histogram cells cost more than the extra pass;
                                  // 9-bit digits for Criteo's 26-bit keys, 3 passes instead of 4, two digits per thread in
                                  // the scans: the step 0.846 vs 0.821 ms bf16, 1.223 vs 1.197 fp32 — round 3, not kept)

// leading dimension of the digit-major histogram matrix: a multiple of 4 so that a row can be
// read with 16-byte loads
inline __host__ __device__ int radix_ld(int nblocks) { return (nblocks + 3) & ~3; }

__global__ void __launch_bounds__(256) radix_hist_kernel(const int32_t* __restrict__ keys, int64_t n,
                                                         int shift, int bins, int nblocks,
                                                         int32_t* __restrict__ bh) {
  __shared__ int h[1 << kSortBits];
  for (int d = threadIdx.x; d < bins; d += 256) h[d] = 0;
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = t0 + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[((uint32_t)keys[i] >> shift) & (bins - 1)], 1);
  }
  __syncthreads();
  const int ld = radix_ld(nblocks);
  for (int d = threadIdx.x; d < bins; d += 256) bh[(int64_t)d * ld + blockIdx.x] = h[d];
}

// Exclusive scan of the digit-major histogram matrix (bins x nblocks <= 64 K ints) by ONE block:
// replaces a rocPRIM device scan (2 launches) by a 3-us kernel.
__global__ void __launch_bounds__(1024) radix_scan_kernel(const int32_t* __restrict__ bh, int total,
                                                          int32_t* __restrict__ off) {
  __shared__ int wsum[16];
  const int per = (total + 1023) / 1024;
  const int lo = threadIdx.x * per;
  const int hi = lo + per < total ? lo + per : total;
  int s = 0;
  for (int i = lo; i < hi; ++i) s += bh[i];
  // block exclusive scan of the 1024 thread sums: wave scan + scan of the 16 wave totals
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += wsum[i];
  int run = base + incl - s;
  for (int i = lo; i < hi; ++i) {
    off[i] = run;
    run += bh[i];
  }
}

// Scatter of one pass.  Ranking as described above (per-wave ballots on private counters);
// then the tile is laid out in LDS in its block-local sorted order and written from there, so
// that consecutive lanes store to consecutive global addresses (runs of one digit) instead of
// 4096 scattered 4-byte stores per block.
// The global offset of (digit d, this block) = sum of all smaller digits' totals + the counts of
// digit d in earlier blocks is computed HERE from the digit-major histogram matrix (thread d
// reads row d with 16-byte loads, then one block scan over the digits): a pass is two launches
// (histogram, scatter) instead of four (+ a 2-launch device scan of the matrix) — the sort is a
// dependent chain of tiny kernels, its length is its cost.
template <bool IOTA>
__global__ void __launch_bounds__(256) radix_scatter_kernel(const int32_t* __restrict__ keys,
                                                            const int32_t* __restrict__ vals,
                                                            int64_t n, int shift, int bits, int nblocks,
                                                            const int32_t* __restrict__ bh,
                                                            int32_t* __restrict__ keys_out,
                                                            int32_t* __restrict__ vals_out) {
  constexpr int MAXB = 1 << kSortBits;
  __shared__ int cntw[4 * MAXB];       // per-wave digit counts -> per-wave exclusive prefixes
  __shared__ int lbase[MAXB];          // block-local exclusive start of every digit
  __shared__ int gbase[MAXB];          // global start of (digit, this block)
  __shared__ int wtot[4], gtot[4];
  __shared__ int32_t skey[kSortTile], sval[kSortTile];
  const int bins = 1 << bits;
  for (int d = threadIdx.x; d < 4 * bins; d += 256) cntw[d] = 0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // digit d = threadIdx.x: total over all blocks and count in the blocks before this one
  int gsum = 0, gpre = 0;
  if (threadIdx.x < bins) {
    const int ld = radix_ld(nblocks);
    const int4* __restrict__ row = reinterpret_cast<const int4*>(bh + (int64_t)threadIdx.x * ld);
    const int me = blockIdx.x;
    for (int b4 = 0; b4 < ld / 4; b4 += 8) {
      int4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = (b4 + u < ld / 4) ? row[b4 + u] : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = (b4 + u) * 4;        // columns >= nblocks of the padded row are never written: mask them
        const int c0 = b + 0 < nblocks ? x[u].x : 0, c1 = b + 1 < nblocks ? x[u].y : 0;
        const int c2 = b + 2 < nblocks ? x[u].z : 0, c3 = b + 3 < nblocks ? x[u].w : 0;
        gsum += c0 + c1 + c2 + c3;
        gpre += (b + 0 < me ? c0 : 0) + (b + 1 < me ? c1 : 0) + (b + 2 < me ? c2 : 0) + (b + 3 < me ? c3 : 0);
      }
    }
  }
  {
    int incl = gsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    if (lane == 63) gtot[w] = incl;
    __syncthreads();                     // also publishes the zeroed cntw
    int base = 0;
    for (int i = 0; i < w; ++i) base += gtot[i];
    if (threadIdx.x < bins) gbase[threadIdx.x] = base + incl - gsum + gpre;
  }
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t tile0 = (int64_t)blockIdx.x * kSortTile;
  const int64_t seg0 = tile0 + w * (kSortTile / 4);
  int32_t key[kSortItems], val[kSortItems], lr[kSortItems];
  int* mine = cntw + w * bins;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = seg0 + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys[i] : 0;
    val[r] = valid ? (IOTA ? (int32_t)i : vals[i]) : 0;
    const int d = ((uint32_t)key[r] >> shift) & (bins - 1);
    uint64_t m = __ballot(valid);
    for (int k = 0; k < bits; ++k) {
      const bool bit = (d >> k) & 1;
      const uint64_t b = __ballot(bit);
      m &= bit ? b : ~b;
    }
    const int base = valid ? mine[d] : 0;                    // all lanes of a group read first ...
    lr[r] = base + __popcll(m & lt);
    if (valid && (m & lt) == 0) mine[d] = base + __popcll(m);   // ... then its lowest lane updates
  }
  __syncthreads();
  // digit d = threadIdx.x (bins <= 256): wave prefixes in place, block-local exclusive base
  {
    int tot = 0;
    if (threadIdx.x < bins) {
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        const int c = cntw[ww * bins + threadIdx.x];
        cntw[ww * bins + threadIdx.x] = tot;
        tot += c;
      }
    }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += wtot[i];
    if (threadIdx.x < bins) lbase[threadIdx.x] = base + incl - tot;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = seg0 + r * 64 + lane;
    if (i < n) {
      const int d = ((uint32_t)key[r] >> shift) & (bins - 1);
      const int pos = lbase[d] + cntw[w * bins + d] + lr[r];
      skey[pos] = key[r];
      sval[pos] = val[r];
    }
  }
  __syncthreads();
  const int count = (int)((n - tile0 < kSortTile) ? (n - tile0) : kSortTile);
  for (int i = threadIdx.x; i < count; i += 256) {
    const int32_t k = skey[i];
    const int d = ((uint32_t)k >> shift) & (bins - 1);
    const int64_t g = (int64_t)gbase[d] + (i - lbase[d]);
    keys_out[g] = k;
    vals_out[g] = sval[i];
  }
}

// Runs of equal keys in the sorted sequence, two launches:
//   seg_count  heads (first position of a run) per tile of kSortTile positions;
//   seg_mark   each block adds the head counts of the tiles before it, scans its own tile and
//              writes rank[j] (1-based run index), uniq[run], seg_start[run]; the last position
//              closes seg_start and sets n_uniq = {number of runs, 0}.
__global__ void __launch_bounds__(256) seg_count_kernel(const int32_t* __restrict__ sk, int64_t n,
                                                        int32_t* __restrict__ cnt) {
  __shared__ int wsum[4];
  const int64_t t0 = (int64_t)blockIdx.x * kSortTile;
  int c = 0;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t j = t0 + r * 256 + threadIdx.x;
    if (j < n && (j == 0 || sk[j] != sk[j - 1])) ++c;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, kWave);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(256) seg_mark_tiles_kernel(const int32_t* __restrict__ sk, int64_t n,
                                                             const int32_t* __restrict__ cnt,
                                                             int32_t* __restrict__ rank,
                                                             int32_t* __restrict__ uniq,
                                                             int32_t* __restrict__ seg_start,
                                                             int32_t* __restrict__ n_uniq) {
  __shared__ int wsum[4];
  __shared__ int before_s;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // runs that start in earlier tiles
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) part += cnt[b];
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, kWave);
  if (lane == 0) wsum[w] = part;
  __syncthreads();
  if (threadIdx.x == 0) before_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  int run = before_s;
  // the tile in 16 rounds of 256 consecutive positions: block scan of the head flags per round
  const int64_t t0 = (int64_t)blockIdx.x * kSortTile;
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t j = t0 + r * 256 + threadIdx.x;
    const bool in = j < n;
    const int32_t k = in ? sk[j] : 0;
    const bool head = in && (j == 0 || k != sk[j - 1]);
    const unsigned long long hb = __ballot(head);
    const int lower = __popcll(hb & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    __syncthreads();
    if (lane == 0) wsum[w] = __popcll(hb);
    __syncthreads();
    int base = run;
    for (int i = 0; i < w; ++i) base += wsum[i];
    const int rj = base + lower + (head ? 1 : 0);
    if (in) {
      rank[j] = rj;
      if (head) {
        uniq[rj - 1] = k;
        seg_start[rj - 1] = (int32_t)j;
      }
      if (j == n - 1) {
        n_uniq[0] = rj;
        n_uniq[1] = 0;      // owner counter of the one segment reduction that will use this plan
        seg_start[rj] = (int32_t)n;
      }
    }
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

// ---------------------------------------------------------------------------------------------
// Multi-problem form: the segment plans of ALL tables of a step (embedding ids + sampled NCE ids)
// sorted by ONE sequence of launches — a sort is a dependent chain of tiny kernels, so two sorts
// one after the other cost twice the chain; here block b of every launch works on problem
// q = (b >= p[1].block0), and the chain is as long as one sort's: 8 launches for both tables of
// a 3-pass sort (V <= 2^24) instead of 8 per table.
// (Tried: one launch per pass, the scatter of pass p also counting the pass-(p+1) digit of every key
// into the histogram cell of the tile the key lands in with integer atomics.  Correct, but the
// sampled ids are unigram-skewed: popular ids put thousands of adds from every block on one L2
// address — 0.9 ms per pass, and still 0.4 ms per plan with one add per run of equal cells.)
constexpr int kMaxSortProbs = 2;
constexpr int kMaxPasses = 4;
struct SortProb {
  const int32_t* keys;   // [n] input
  int64_t n;
  int nblocks, block0;   // tiles of kSortTile keys; first block of this problem inside a launch
  int bits, passes;      // key width, ceil(bits / 8)
  int32_t* bh[kMaxPasses];   // bh[0]: digit-major histogram matrix [256][radix_ld(nblocks)] of the running pass
  int32_t *tk, *tv;      // ping-pong buffers [n]
  int32_t *sorted_keys, *perm, *rank, *uniq, *seg_start, *n_uniq, *cnt;   // outputs; cnt [nblocks] scratch
};
struct SortProbs {
  SortProb p[kMaxSortProbs];
  int count;
};

__device__ inline int sort_prob_of(const SortProbs& ps, int& local_block) {
  int q = 0;
  if (ps.count > 1 && (int)blockIdx.x >= ps.p[1].block0) q = 1;
  local_block = (int)blockIdx.x - ps.p[q].block0;
  return q;
}
__device__ inline int pass_bits(int bits, int pass) {
  const int left = bits - pass * kSortBits;
  return left < kSortBits ? left : kSortBits;
}

__global__ void __launch_bounds__(256) radix_hist_mp_kernel(SortProbs ps, int pass) {
  int lb;
  const SortProb& pr = ps.p[sort_prob_of(ps, lb)];
  if (pass >= pr.passes) return;
  __shared__ int h[1 << kSortBits];
  const int shift = pass * kSortBits, bins = 1 << pass_bits(pr.bits, pass);
  const bool to_out = ((pr.passes - 1 - pass) % 2) == 0;        // this pass's INPUT is the other buffer
  const int32_t* __restrict__ keys = pass == 0 ? pr.keys : (to_out ? pr.tk : pr.sorted_keys);
  for (int d = threadIdx.x; d < bins; d += 256) h[d] = 0;
  __syncthreads();
  const int64_t t0 = (int64_t)lb * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = t0 + r * 256 + threadIdx.x;
    if (i < pr.n) atomicAdd(&h[((uint32_t)keys[i] >> shift) & (bins - 1)], 1);
  }
  __syncthreads();
  const int ld = radix_ld(pr.nblocks);
  for (int d = threadIdx.x; d < bins; d += 256) pr.bh[0][(int64_t)d * ld + lb] = h[d];
}

template <bool FIRST>
__global__ void __launch_bounds__(256) radix_scatter_mp_kernel(SortProbs ps, int pass) {
  int lb;
  const SortProb& pr = ps.p[sort_prob_of(ps, lb)];
  if (pass >= pr.passes) return;                      // a narrower table is done already (block-uniform)
  constexpr int MAXB = 1 << kSortBits;
  __shared__ int cntw[4 * MAXB];
  __shared__ int lbase[MAXB];
  __shared__ int gbase[MAXB];
  __shared__ int wtot[4], gtot[4];
  __shared__ int32_t skey[kSortTile], sval[kSortTile];
  const int shift = pass * kSortBits, bits = pass_bits(pr.bits, pass), bins = 1 << bits;
  const int64_t n = pr.n;
  const int nblocks = pr.nblocks;
  // ping-pong so that the last pass lands in (sorted_keys, perm)
  const bool to_out = ((pr.passes - 1 - pass) % 2) == 0;
  const int32_t* __restrict__ keys = FIRST ? pr.keys : (to_out ? pr.tk : pr.sorted_keys);
  const int32_t* __restrict__ vals = FIRST ? nullptr : (to_out ? pr.tv : pr.perm);
  int32_t* __restrict__ keys_out = to_out ? pr.sorted_keys : pr.tk;
  int32_t* __restrict__ vals_out = to_out ? pr.perm : pr.tv;
  const int32_t* __restrict__ bh = pr.bh[0];

  for (int d = threadIdx.x; d < 4 * bins; d += 256) cntw[d] = 0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ld = radix_ld(nblocks);
  int gsum = 0, gpre = 0;
  if (threadIdx.x < bins) {
    const int4* __restrict__ row = reinterpret_cast<const int4*>(bh + (int64_t)threadIdx.x * ld);
    for (int b4 = 0; b4 < ld / 4; b4 += 8) {
      int4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = (b4 + u < ld / 4) ? row[b4 + u] : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = (b4 + u) * 4;
        const int c0 = b + 0 < nblocks ? x[u].x : 0, c1 = b + 1 < nblocks ? x[u].y : 0;
        const int c2 = b + 2 < nblocks ? x[u].z : 0, c3 = b + 3 < nblocks ? x[u].w : 0;
        gsum += c0 + c1 + c2 + c3;
        gpre += (b + 0 < lb ? c0 : 0) + (b + 1 < lb ? c1 : 0) + (b + 2 < lb ? c2 : 0) + (b + 3 < lb ? c3 : 0);
      }
    }
  }
  {
    int incl = gsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    if (lane == 63) gtot[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += gtot[i];
    if (threadIdx.x < bins) gbase[threadIdx.x] = base + incl - gsum + gpre;
  }
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t tile0 = (int64_t)lb * kSortTile;
  const int64_t seg0 = tile0 + w * (kSortTile / 4);
  int32_t key[kSortItems], val[kSortItems], lr[kSortItems];
  int* mine = cntw + w * bins;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = seg0 + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys[i] : 0;
    val[r] = valid ? (FIRST ? (int32_t)i : vals[i]) : 0;
    const int d = ((uint32_t)key[r] >> shift) & (bins - 1);
    uint64_t m = __ballot(valid);
    for (int k = 0; k < bits; ++k) {
      const bool bit = (d >> k) & 1;
      const uint64_t b = __ballot(bit);
      m &= bit ? b : ~b;
    }
    const int base = valid ? mine[d] : 0;
    lr[r] = base + __popcll(m & lt);
    if (valid && (m & lt) == 0) mine[d] = base + __popcll(m);
  }
  __syncthreads();
  {
    int tot = 0;
    if (threadIdx.x < bins) {
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        const int c = cntw[ww * bins + threadIdx.x];
        cntw[ww * bins + threadIdx.x] = tot;
        tot += c;
      }
    }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += wtot[i];
    if (threadIdx.x < bins) lbase[threadIdx.x] = base + incl - tot;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = seg0 + r * 64 + lane;
    if (i < n) {
      const int d = ((uint32_t)key[r] >> shift) & (bins - 1);
      const int pos = lbase[d] + cntw[w * bins + d] + lr[r];
      skey[pos] = key[r];
      sval[pos] = val[r];
    }
  }
  __syncthreads();
  const int count = (int)((n - tile0 < kSortTile) ? (n - tile0) : kSortTile);
  for (int i = threadIdx.x; i < count; i += 256) {
    const int32_t k = skey[i];
    const int d = ((uint32_t)k >> shift) & (bins - 1);
    const int64_t g = (int64_t)gbase[d] + (i - lbase[d]);
    keys_out[g] = k;
    vals_out[g] = sval[i];
  }
}

__global__ void __launch_bounds__(256) seg_count_mp_kernel(SortProbs ps) {
  int lb;
  const SortProb& pr = ps.p[sort_prob_of(ps, lb)];
  __shared__ int wsum[4];
  const int32_t* __restrict__ sk = pr.sorted_keys;
  const int64_t t0 = (int64_t)lb * kSortTile;
  int c = 0;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t j = t0 + r * 256 + threadIdx.x;
    if (j < pr.n && (j == 0 || sk[j] != sk[j - 1])) ++c;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, kWave);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) pr.cnt[lb] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(256) seg_mark_mp_kernel(SortProbs ps) {
  int lb;
  const SortProb& pr = ps.p[sort_prob_of(ps, lb)];
  __shared__ int wsum[4];
  __shared__ int before_s;
  const int32_t* __restrict__ sk = pr.sorted_keys;
  const int64_t n = pr.n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int part = 0;
  for (int b = threadIdx.x; b < lb; b += 256) part += pr.cnt[b];
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, kWave);
  if (lane == 0) wsum[w] = part;
  __syncthreads();
  if (threadIdx.x == 0) before_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  int run = before_s;
  const int64_t t0 = (int64_t)lb * kSortTile;
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t j = t0 + r * 256 + threadIdx.x;
    const bool in = j < n;
    const int32_t k = in ? sk[j] : 0;
    const bool head = in && (j == 0 || k != sk[j - 1]);
    const unsigned long long hb = __ballot(head);
    const int lower = __popcll(hb & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    __syncthreads();
    if (lane == 0) wsum[w] = __popcll(hb);
    __syncthreads();
    int base = run;
    for (int i = 0; i < w; ++i) base += wsum[i];
    const int rj = base + lower + (head ? 1 : 0);
    if (in) {
      pr.rank[j] = rj;
      if (head) {
        pr.uniq[rj - 1] = k;
        pr.seg_start[rj - 1] = (int32_t)j;
      }
      if (j == n - 1) {
        pr.n_uniq[0] = rj;
        pr.n_uniq[1] = 0;
        pr.seg_start[rj] = (int32_t)n;
      }
    }
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

inline int radix_passes(int bits) { return (bits + kSortBits - 1) / kSortBits; }
inline int radix_blocks(int64_t n) { return (int)ceil_div(n > 0 ? n : 1, kSortTile); }
// ints of workspace: two ping-pong arrays + histogram + offsets
inline size_t radix_ws_ints(int64_t n) {
  return 2 * (size_t)n + 2 * (size_t)(1 << kSortBits) * radix_ld(radix_blocks(n));
}

}  // namespace mapx
