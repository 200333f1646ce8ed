// Vocabulary builder of one categorical field: what reference
// data_preprocess/proc_avazu.py:237-262 and proc_criteo.py:147-163 do per field with a Python Counter —
//   for k, v in Counter(feat).most_common():  if v >= n_core: feat_map[name-k] = len(feat_map)
//   feat_map[name-<oov>] = len(feat_map);  feat_ids = [feat_map.get(name-f, oov) for f in feat]
// i.e. count every distinct raw value, rank the values by DESCENDING count (ties: the value that occurs first in
// the column comes first — Counter keeps insertion order and most_common() sorts stably), keep those seen at
// least n_core times, give them consecutive ids from the field's base, one more id to <oov>, and translate
// the column.  Integer work, bit-exact.
//
// On the GPU: (1) count — one pass over the column into an open-addressing hash table keyed by the raw
// 64-bit value (atomicCAS on the key, atomicAdd on the count, atomicMin on the first position; integer atomics,
// so the result does not depend on the order of arrival); (2) compact the occupied slots into an entry list;
// (3) the host sorts the entries by (count descending, first position ascending) with the library's stable
// radix sort (mapx_seg_plan: first by first position, then by max_count - count); (4) assign ranks and the
// n-core cut; (5) map every row through slot -> entry -> rank -> id.  HBM-bound: 8 B of key + a random
// 16-byte table slot per row.
#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

constexpr int64_t kVocabEmpty = INT64_MIN;       // key value no column may hold

__device__ inline uint64_t vocab_hash(uint64_t x) {   // splitmix64 finaliser: raw ids are hashes or small ints
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

__global__ void __launch_bounds__(256) vocab_table_init_kernel(int64_t* __restrict__ tkey, int32_t* __restrict__ tcount,
                                                               int32_t* __restrict__ tfirst, int64_t cap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    tkey[i] = kVocabEmpty;
    tcount[i] = 0;
    tfirst[i] = INT32_MAX;
  }
}

__global__ void __launch_bounds__(256) vocab_count_kernel(const int64_t* __restrict__ keys, int64_t n,
                                                          int64_t* __restrict__ tkey, int32_t* __restrict__ tcount,
                                                          int32_t* __restrict__ tfirst, int64_t cap,
                                                          int32_t* __restrict__ slot_of_row, int* __restrict__ err) {
  const uint64_t mask = (uint64_t)cap - 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = keys[i];
    if (k == kVocabEmpty) {
      atomicOr(err, 1);
      slot_of_row[i] = 0;
      continue;
    }
    uint64_t s = vocab_hash((uint64_t)k) & mask;
    int64_t probes = 0;
    for (;;) {
      unsigned long long* cell = reinterpret_cast<unsigned long long*>(tkey + s);
      long long seen = (long long)*reinterpret_cast<volatile unsigned long long*>(cell);
      if (seen == kVocabEmpty)
        seen = (long long)atomicCAS(cell, (unsigned long long)kVocabEmpty, (unsigned long long)k);
      if (seen == kVocabEmpty || seen == k) break;          // claimed now, or this value's slot already
      s = (s + 1) & mask;
      if (++probes >= cap) {                                // table full: the caller sized it too small
        atomicOr(err, 2);
        break;
      }
    }
    atomicAdd(tcount + s, 1);
    atomicMin(tfirst + s, (int32_t)i);
    slot_of_row[i] = (int32_t)s;
  }
}

// occupied slots -> entries (any order: the ranking below is a total order), largest count on the side
__global__ void __launch_bounds__(256) vocab_compact_kernel(const int64_t* __restrict__ tkey,
                                                            const int32_t* __restrict__ tcount,
                                                            const int32_t* __restrict__ tfirst, int64_t cap,
                                                            int32_t* __restrict__ slot_entry,
                                                            int64_t* __restrict__ ekey, int32_t* __restrict__ ecount,
                                                            int32_t* __restrict__ efirst,
                                                            int32_t* __restrict__ n_entries_maxc) {
  const int lane = threadIdx.x & 63;
  int mymax = 0;
  for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < cap;
       i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + lane;
    const bool occ = i < cap && tkey[i] != kVocabEmpty;
    const unsigned long long m = __ballot(occ);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(n_entries_maxc, __popcll(m));
    base = __shfl(base, 0, kWave);
    if (occ) {
      const int e = base + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
      slot_entry[i] = e;
      ekey[e] = tkey[i];
      ecount[e] = tcount[i];
      efirst[e] = tfirst[i];
      mymax = max(mymax, tcount[i]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) mymax = max(mymax, __shfl_xor(mymax, o, kWave));
  if (lane == 0 && mymax > 0) atomicMax(n_entries_maxc + 1, mymax);
}

// second sort key of the entries listed in first-position order: max_count - count (ascending = count descending)
__global__ void __launch_bounds__(256) vocab_rank_keys_kernel(const int32_t* __restrict__ ecount,
                                                              const int32_t* __restrict__ by_first, int64_t U,
                                                              int32_t maxc, int32_t* __restrict__ keys2) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < U; r += (int64_t)gridDim.x * blockDim.x)
    keys2[r] = maxc - ecount[by_first[r]];
}

// rank r (0 = most frequent) holds entry by_first[by_count[r]]; entries seen >= n_core times are kept
__global__ void __launch_bounds__(256) vocab_assign_kernel(const int32_t* __restrict__ by_first,
                                                           const int32_t* __restrict__ by_count,
                                                           const int32_t* __restrict__ ecount,
                                                           const int64_t* __restrict__ ekey, int64_t U, int32_t n_core,
                                                           int32_t* __restrict__ rank_of_entry,
                                                           int64_t* __restrict__ ranked_keys,
                                                           int32_t* __restrict__ ranked_counts,
                                                           int32_t* __restrict__ n_kept) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < U; r += (int64_t)gridDim.x * blockDim.x) {
    const int32_t e = by_first[by_count[r]];
    const int32_t c = ecount[e];
    rank_of_entry[e] = (int32_t)r;
    ranked_keys[r] = ekey[e];
    ranked_counts[r] = c;
    if (c >= n_core) {                       // counts descend with r: the last kept rank closes the vocabulary
      const bool last = (r == U - 1) || (ecount[by_first[by_count[r + 1]]] < n_core);
      if (last) *n_kept = (int32_t)r + 1;
    }
  }
}

__global__ void __launch_bounds__(256) vocab_map_kernel(const int32_t* __restrict__ slot_of_row,
                                                        const int32_t* __restrict__ slot_entry,
                                                        const int32_t* __restrict__ rank_of_entry,
                                                        const int32_t* __restrict__ n_kept, int64_t n, int64_t base,
                                                        int64_t* __restrict__ ids_out, int64_t ld_out) {
  const int32_t kept = *n_kept;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = rank_of_entry[slot_entry[slot_of_row[i]]];
    ids_out[i * ld_out] = base + (r < kept ? r : kept);     // base + kept = the field's <oov> id
  }
}

}  // namespace mapx

extern "C" int mapx_vocab_table_init(int64_t* table_keys, int32_t* table_count, int32_t* table_first, int64_t capacity,
                                     hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(table_keys && table_count && table_first, "vocab_table_init: null pointer");
  MAPX_REQUIRE(capacity >= 64 && (capacity & (capacity - 1)) == 0 && capacity <= ((int64_t)1 << 31),
               "vocab_table_init: capacity must be a power of two in [64, 2^31]");
  hipLaunchKernelGGL(vocab_table_init_kernel, dim3(grid_for(capacity, 256)), dim3(256), 0, stream, table_keys,
                     table_count, table_first, capacity);
  return check_launch("vocab_table_init");
}

extern "C" int mapx_vocab_count(const int64_t* keys, int64_t n, int64_t* table_keys, int32_t* table_count,
                                int32_t* table_first, int64_t capacity, int32_t* slot_of_row, int* err_flag,
                                hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "vocab_count: at most 2^31 - 1 rows");
  MAPX_REQUIRE(capacity >= 64 && (capacity & (capacity - 1)) == 0, "vocab_count: capacity must be a power of two >= 64");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(keys && table_keys && table_count && table_first && slot_of_row && err_flag, "vocab_count: null pointer");
  hipLaunchKernelGGL(vocab_count_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream, keys, n, table_keys,
                     table_count, table_first, capacity, slot_of_row, err_flag);
  return check_launch("vocab_count");
}

extern "C" int mapx_vocab_compact(const int64_t* table_keys, const int32_t* table_count, const int32_t* table_first,
                                  int64_t capacity, int32_t* slot_entry, int64_t* entry_keys, int32_t* entry_count,
                                  int32_t* entry_first, int32_t* n_entries_maxc, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(table_keys && table_count && table_first && slot_entry && entry_keys && entry_count && entry_first &&
                   n_entries_maxc,
               "vocab_compact: null pointer");
  MAPX_REQUIRE(capacity >= 64 && capacity % 64 == 0, "vocab_compact: capacity must be a multiple of 64");
  MAPX_HIP(hipMemsetAsync(n_entries_maxc, 0, 2 * sizeof(int32_t), stream));
  hipLaunchKernelGGL(vocab_compact_kernel, dim3(grid_for(capacity, 256)), dim3(256), 0, stream, table_keys, table_count,
                     table_first, capacity, slot_entry, entry_keys, entry_count, entry_first, n_entries_maxc);
  return check_launch("vocab_compact");
}

extern "C" int mapx_vocab_rank_keys(const int32_t* entry_count, const int32_t* by_first, int64_t n_entries,
                                    int32_t max_count, int32_t* keys2, hipStream_t stream) {
  using namespace mapx;
  if (n_entries <= 0) return MAPX_OK;
  MAPX_REQUIRE(entry_count && by_first && keys2 && max_count >= 1, "vocab_rank_keys: bad arguments");
  hipLaunchKernelGGL(vocab_rank_keys_kernel, dim3(grid_for(n_entries, 256)), dim3(256), 0, stream, entry_count, by_first,
                     n_entries, max_count, keys2);
  return check_launch("vocab_rank_keys");
}

extern "C" int mapx_vocab_assign(const int32_t* by_first, const int32_t* by_count, const int32_t* entry_count,
                                 const int64_t* entry_keys, int64_t n_entries, int32_t n_core, int32_t* rank_of_entry,
                                 int64_t* ranked_keys, int32_t* ranked_counts, int32_t* n_kept, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(n_kept, "vocab_assign: null pointer");
  MAPX_HIP(hipMemsetAsync(n_kept, 0, sizeof(int32_t), stream));
  if (n_entries <= 0) return MAPX_OK;
  MAPX_REQUIRE(by_first && by_count && entry_count && entry_keys && rank_of_entry && ranked_keys && ranked_counts,
               "vocab_assign: null pointer");
  hipLaunchKernelGGL(vocab_assign_kernel, dim3(grid_for(n_entries, 256)), dim3(256), 0, stream, by_first, by_count,
                     entry_count, entry_keys, n_entries, n_core, rank_of_entry, ranked_keys, ranked_counts, n_kept);
  return check_launch("vocab_assign");
}

extern "C" int mapx_vocab_map(const int32_t* slot_of_row, const int32_t* slot_entry, const int32_t* rank_of_entry,
                              const int32_t* n_kept, int64_t n, int64_t base, int64_t* ids_out, int64_t ld_out,
                              hipStream_t stream) {
  using namespace mapx;
  if (n <= 0) return MAPX_OK;
  MAPX_REQUIRE(slot_of_row && slot_entry && rank_of_entry && n_kept && ids_out && ld_out >= 1, "vocab_map: bad arguments");
  hipLaunchKernelGGL(vocab_map_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream, slot_of_row, slot_entry,
                     rank_of_entry, n_kept, n, base, ids_out, ld_out);
  return check_launch("vocab_map");
}
