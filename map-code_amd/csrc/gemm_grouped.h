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
};

// gemm_x3.hip
hipError_t enc_grouped_fwd_x3_launch(const GroupedArgs& g, int cap_slots, hipStream_t stream);
hipError_t enc_grouped_dw_x3_launch(const GroupedArgs& g, int F, hipStream_t stream);

}  // namespace mapx
