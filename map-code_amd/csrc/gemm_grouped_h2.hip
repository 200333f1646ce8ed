// The grouped feat_encoder products (gemm.hip "Grouped GEMMs": reference models.py:74-75, only the masked fields' blocks
// of the encoder) in the two-piece fp16 arithmetic of gemm_h2.hip: gemm_x3.hip's gemm_grouped_x3_kernel with two fp16
// planes per operand instead of three bf16 ones and three MFMAs per K16 step instead of six — same slot layout, same
// XCD-aware tile order, same eight waves splitting every K-step between two sets of four, same outputs.  Taken when
// both operands come with a magnitude record (`final`: raised by both towers' last kernels; the encoder weight: by
// AdamW; dh: by the NCE forward kernel).
#include "gemm_grouped.h"
#include "gemm_h2_common.h"

namespace mapx {

template <bool DW>
__global__ void __launch_bounds__(512) gemm_grouped_h2_kernel(GroupedArgs a) {
  constexpr int NT = 512;
  constexpr int BM = DW ? 32 : 128, BN = DW ? 128 : 32;
  using OpA = OperandH2<BM, 1, !DW, NT>;     // FWD: k-contiguous gathered rows;  DW: dh, [k = slot][p]
  using OpB = OperandH2<BN, 1, !DW, NT>;     // FWD: the field's 32 weight rows;  DW: gathered rows, [k = slot][n]
  static_assert(OpA::NV == 1 && OpB::NV == 1, "one chunk per thread per operand");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16_t* const smem = reinterpret_cast<f16_t*>(smem_raw);
  const int na = __builtin_amdgcn_readfirstlane(h2_scale_exp(a.amax_a));
  const int nb = __builtin_amdgcn_readfirstlane(h2_scale_exp(a.amax_b));
  const float sA = pow2f(na), sB = pow2f(nb);
  const int dn = -(na + nb);
  constexpr int kBuf = OpA::LDS_ELEMS + OpB::LDS_ELEMS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int w4 = wave & 3, half = wave >> 2;       // output tile of the wave, k16 half of every K-step
  const int abase = DW ? 0 : w4 * 32, bbase = DW ? w4 * 32 : 0;

  // Block -> tile, XCD-aware (workgroups go round-robin to the 8 XCDs, each with its own 4 MB L2).  Both
  // products read every row of `final` once per masked field of that row (~7 times): in the launch order
  // of the plain grids each XCD sees all of `final` (22 MB) and re-fetches it through the fabric; ordered
  // so that one XCD works on one eighth of the batch rows (FWD) or on one or two 128-column slices
  // (DW), the re-reads hit its L2.
  int f, kbeg, kend, n0 = 0, slot0 = 0;
  if (DW) {
    const int nb = gridDim.x, per = nb / 8;
    int lin = blockIdx.x;
    if (lin < per * 8) lin = (lin % 8) * per + lin / 8;
    f = lin % a.F;                                   // column-slice-major: consecutive blocks share their columns
    n0 = (lin / a.F) * BN;
    kbeg = a.group_start[f];
    kend = a.group_start[f + 1];
  } else {
    if (a.zero_out) {
      float4* z = reinterpret_cast<float4*>(a.zero_out + (int64_t)blockIdx.x * BM * 32);
      for (int i = threadIdx.x; i < BM * 32 / 4; i += NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int tile = grouped_fwd_tile(a, lane, wave);
    if (tile < 0) return;
    f = a.tile_group[tile];
    if (f < 0) return;
    slot0 = tile * BM;
    kbeg = 0;
    kend = a.K;
  }
  const float* __restrict__ Bb = DW ? a.B : a.B + (int64_t)f * 32 * a.ldb;

  f32x16 acc, cor;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = cor[r] = 0.f;

  // this thread's chunk of each operand (coordinates inside the tile never change)
  int atr, atc, btr, btc;
  OpA::coords(threadIdx.x, atr, atc);
  OpB::coords(threadIdx.x, btr, btc);
  const bool a_mine = !OpA::PARTIAL || threadIdx.x < OpA::TOTAL;
  const bool b_mine = !OpB::PARTIAL || threadIdx.x < OpB::TOTAL;
  int64_t arow = 0;
  bool arow_ok = false;
  if (!DW) {
    const int row = a.rowmap[slot0 + atr];
    arow_ok = row >= 0;
    arow = (int64_t)(row >= 0 ? row : 0) * a.lda + atc;
  }
  OpA la[2];
  OpB lb[2];
  int bnext = -1;                                    // DW: rowmap entry of this thread's row of the next tile to load
#define MAPX_GX_ROW(t) (DW ? ((kbeg + (t) * kXBK + btr) < kend ? a.rowmap[kbeg + (t) * kXBK + btr] : -1) : 0)
#define MAPX_GX_LOAD(SET, t, BROW)                                                                   \
  do {                                                                                               \
    const int k0 = kbeg + (t) * kXBK;                                                                \
    if (a_mine) {                                                                                    \
      const float* q;                                                                                \
      if (DW) {                                                                                      \
        la[SET].ok[0] = (k0 + atr) < kend;                                                           \
        q = a.A + (la[SET].ok[0] ? (int64_t)(k0 + atr) * a.lda + atc : 0);                           \
      } else {                                                                                       \
        la[SET].ok[0] = arow_ok && (k0 + atc) < kend;                                                \
        q = a.A + (la[SET].ok[0] ? arow + k0 : 0);                                                   \
      }                                                                                              \
      la[SET].r[0][0] = *reinterpret_cast<const float4*>(q);                                         \
      la[SET].r[0][1] = *reinterpret_cast<const float4*>(q + 4);                                     \
    }                                                                                                \
    if (b_mine) {                                                                                    \
      const float* q;                                                                                \
      if (DW) {                                                                                      \
        const int row = (BROW);                                                                      \
        lb[SET].ok[0] = row >= 0 && (n0 + btc) < a.N;                                                \
        q = Bb + (lb[SET].ok[0] ? (int64_t)row * a.ldb + n0 + btc : 0);                              \
      } else {                                                                                       \
        lb[SET].ok[0] = (k0 + btc) < kend;                                                           \
        q = Bb + (int64_t)btr * a.ldb + (lb[SET].ok[0] ? k0 + btc : 0);                              \
      }                                                                                              \
      lb[SET].r[0][0] = *reinterpret_cast<const float4*>(q);                                         \
      lb[SET].r[0][1] = *reinterpret_cast<const float4*>(q + 4);                                     \
    }                                                                                                \
  } while (0)
#define MAPX_GX_STORE(SET, buf)                                                                      \
  do {                                                                                               \
    la[SET].store_masked(smem + (buf) * kBuf, sA);                                                   \
    lb[SET].store_masked(smem + (buf) * kBuf + OpA::LDS_ELEMS, sB);                                  \
  } while (0)
  const int nk = (kend - kbeg + kXBK - 1) / kXBK;
  if (nk > 0) MAPX_GX_LOAD(0, 0, MAPX_GX_ROW(0));
  if (nk > 1) MAPX_GX_LOAD(1, 1, MAPX_GX_ROW(1));
  if (nk > 0) MAPX_GX_STORE(0, 0);
  if (nk > 2) MAPX_GX_LOAD(0, 2, MAPX_GX_ROW(2));
  if (nk > 3) bnext = MAPX_GX_ROW(3);
  __syncthreads();
  // K-step kt on LDS buffer CUR = kt & 1 (literal): the wave's half of tile kt, then cut + store of tile
  // kt+1 (register set CUR^1, landed), then the loads of tile kt+3 into that set and the row index of kt+4.
#define MAPX_GX_KSTEP(CUR, kt, STEADY)                                                               \
  do {                                                                                               \
    const f16_t* const As_cur = smem + (CUR) * kBuf;                                                 \
    const f16_t* const Bs_cur = As_cur + OpA::LDS_ELEMS;                                             \
    const f16x8 ah = OpA::frag1(As_cur, 0, abase, lane, half, 0), bh = OpB::frag1(Bs_cur, 0, bbase, lane, half, 0); \
    const f16x8 al = OpA::frag1(As_cur, 1, abase, lane, half, 0), bl = OpB::frag1(Bs_cur, 1, bbase, lane, half, 0); \
    cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cor, 0, 0, 0);                              \
    cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cor, 0, 0, 0);                              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);                              \
    if ((STEADY) || (kt) + 1 < nk) MAPX_GX_STORE((CUR) ^ 1, (CUR) ^ 1);                              \
    /* the row index of tile kt+4 is ISSUED before the loads of tile kt+3 (memory returns in order: */ \
    /* the wait for it next step then leaves those loads in flight)                                  */ \
    int bnew = -1;                                                                                   \
    if (DW && ((STEADY) || (kt) + 4 < nk)) bnew = MAPX_GX_ROW((kt) + 4);                             \
    if ((STEADY) || (kt) + 3 < nk) MAPX_GX_LOAD((CUR) ^ 1, (kt) + 3, bnext);                         \
    bnext = bnew;                                                                                    \
    __syncthreads();                                                                                 \
  } while (0)
  int kt = 0;
  for (; kt + 5 < nk; kt += 2) {
    MAPX_GX_KSTEP(0, kt, true);
    MAPX_GX_KSTEP(1, kt + 1, true);
  }
  for (; kt < nk; kt += 2) {
    MAPX_GX_KSTEP(0, kt, false);
    if (kt + 1 < nk) MAPX_GX_KSTEP(1, kt + 1, false);
  }
#undef MAPX_GX_KSTEP
#undef MAPX_GX_STORE
#undef MAPX_GX_LOAD
#undef MAPX_GX_ROW

  // the second half's partial tiles go through LDS (every K-step ended on a barrier: the buffers are free)
  float* const part = reinterpret_cast<float*>(smem_raw);      // [4 tiles][16 registers][64 lanes]
  if (half == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(w4 * 16 + r) * 64 + lane] = __builtin_fmaf(cor[r], 0x1p-11f, acc[r]);
  }
  __syncthreads();
  if (half == 1) return;
  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int n = (DW ? n0 : 0) + bbase + l31;
  if (DW ? (n < a.N) : true) {
    const float bn = DW ? 0.f : a.bias[f * 32 + n];
    const float gs = (DW && a.gscale) ? *a.gscale : 1.f;
    float* __restrict__ Cb = DW ? a.C + (int64_t)f * 32 * a.ldc : a.C + (int64_t)slot0 * a.ldc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = abase + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const float v = __builtin_ldexpf(__builtin_fmaf(cor[r], 0x1p-11f, acc[r]) + part[(w4 * 16 + r) * 64 + lane], dn);
      Cb[(int64_t)m * a.ldc + n] = DW ? v * gs : v + bn;
    }
  }
}


template <bool DW>
static hipError_t launch_grouped_h2(const GroupedArgs& g, dim3 grid, hipStream_t stream) {
  constexpr int BM = DW ? 32 : 128, BN = DW ? 128 : 32;
  using OpA = OperandH2<BM, 1, !DW, 512>;
  using OpB = OperandH2<BN, 1, !DW, 512>;
  constexpr size_t lds = (size_t)2 * (OpA::LDS_ELEMS + OpB::LDS_ELEMS) * sizeof(f16_t);
  static_assert(lds >= 4 * 16 * 64 * sizeof(float) && lds <= 160 * 1024, "LDS budget");
  auto* fn = &gemm_grouped_h2_kernel<DW>;
  static hipError_t raised =
      hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, stream, g);
  return hipSuccess;
}

hipError_t enc_grouped_fwd_h2_launch(const GroupedArgs& g, int cap_slots, hipStream_t stream) {
  return launch_grouped_h2<false>(g, dim3(cap_slots / 128), stream);
}
hipError_t enc_grouped_dw_h2_launch(const GroupedArgs& g, int F, hipStream_t stream) {
  return launch_grouped_h2<true>(g, dim3(((g.N + 127) / 128) * F), stream);
}

}  // namespace mapx
