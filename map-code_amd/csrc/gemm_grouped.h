// Arguments of the grouped feat_encoder GEMMs (gemm.hip: fp32 MFMA kernels; gemm_x3.hip: the same
// products on the bf16 matrix cores).  Layout and meaning of the slot order: gemm.hip, "Grouped GEMMs".
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace mapx {

struct GroupedArgs {
  const float* A; int64_t lda;       // FWD: final [B, K]          DW: dh [slots, 32]
  const float* B; int64_t ldb;       // FWD: W [F*32, K]           DW: final [B, N]
  float* C; int64_t ldc;             // FWD: h [slots, 32]         DW: dW [F*32, N]
  const float* bias;                 // FWD only
  const int32_t* rowmap;             // [slots]
  const int32_t* tile_group;         // FWD: [slots/128] field of the tile, -1 = unused
  const int32_t* group_start;        // DW: [F+1] first slot of each field's group (multiples of 128)
  int F;                             // number of fields (x3 kernels: block -> tile order)
  int K, N, nrows;                   // FWD: K = D+H;  DW: N = D+H;  nrows = B (bounds of rowmap values)
  float* zero_out;                   // FWD, optional: [slots, 32] buffer cleared tile by tile (dh_slots)
  const float* gscale;               // DW, optional device scalar multiplied into the result
  const float* amax_a;               // magnitude records of A and B (amax.h): both given -> the two-piece fp16
  const float* amax_b;               // kernels (gemm_grouped_h2.hip), else the six-product bf16 ones (gemm_x3.hip)
};

// gemm_x3.hip
hipError_t enc_grouped_fwd_x3_launch(const GroupedArgs& g, int cap_slots, hipStream_t stream);
hipError_t enc_grouped_dw_x3_launch(const GroupedArgs& g, int F, hipStream_t stream);
// gemm_grouped_h2.hip
hipError_t enc_grouped_fwd_h2_launch(const GroupedArgs& g, int cap_slots, hipStream_t stream);
hipError_t enc_grouped_dw_h2_launch(const GroupedArgs& g, int F, hipStream_t stream);

#ifdef __HIPCC__
// FWD: block -> 128-slot tile, XCD-aware (8 waves; every thread of the block gets the same answer, -1 = no tile).
// Tiles sorted by (eighth of their group they lie in, field, index in the group): a group's slots are in
// batch-row order, so equal eighths of different groups hold the same batch rows.  Block b takes sorted position
// (b % 8) * per + b / 8 — a bijection on [0, 8 per) that covers the used tiles.  (Computed by the whole block at
// once: wave x sums the tiles below key x over the fields, its lanes.)
__device__ inline int grouped_fwd_tile(const GroupedArgs& a, int lane, int wave) {
  constexpr int BM = 128;
  int tile = blockIdx.x;
  if (a.group_start) {
    const int used = a.group_start[a.F] / BM, per = (used + 7) / 8;
    if ((int)gridDim.x >= 8 * per && a.F <= 64) {
      const int i = blockIdx.x / 8, pos = (blockIdx.x % 8) * per + i;
      if (i >= per || pos >= used) return -1;
      __shared__ int below[8];                     // tiles with key < x:  sum_f ceil(x nf / 8)
      const int gs0 = lane < a.F ? a.group_start[lane] : 0, gs1 = lane < a.F ? a.group_start[lane + 1] : 0;
      const int nf = (gs1 - gs0) / BM;
      int v = (wave * nf + 7) >> 3;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) below[wave] = v;
      __syncthreads();
      int X = 0;
#pragma unroll
      for (int x = 1; x < 8; ++x)
        if (below[x] <= pos) X = x;
      const int q = pos - below[X];                // index among the tiles of key X, ordered by (field, j)
      const int j0 = (X * nf + 7) >> 3, c = (((X + 1) * nf + 7) >> 3) - j0;
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
      }
      const bool hit = q >= incl - c && q < incl;
      const unsigned long long m = __ballot(hit);
      if (m == 0) return -1;
      tile = __shfl(gs0 / BM + j0 + (q - (incl - c)), __ffsll((long long)m) - 1);
    }
  }
  return tile;
}

#endif

}  // namespace mapx
