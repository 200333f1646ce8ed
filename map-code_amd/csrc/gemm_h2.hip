// fp32 GEMM on the fp16 matrix cores with TWO pieces per operand element and THREE MFMAs per product (round 4).
// Every operand tensor comes with its magnitude record (amax.h): a power of two s = 2^n brings max |x| into
// [2^14, 2^15), and each element is cut as
//   s x = hi + 2^-11 lo,   hi = rn_fp16(s x),   lo = rn_fp16(2^11 (s x - hi))          (11 + 11 significant bits,
// pieces of either sign; the residual s x - hi is exact in fp32), so that
//   a b = 2^-(na+nb) [ hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) ] + O(2^-22 |a b|):
// the leading products go to one fp32 MFMA accumulator, the two corrections to a second one, both are exact
// (11 x 11 bits) and meet once, in the epilogue, where the scales are undone.  What is dropped (lo lo, and what
// rn_fp16 leaves of lo) is below 2^-22 |a b| per product and has no preferred sign; against fp64 the products measure
// 0.6e-7 .. 1.4e-7 of sum_k |a_k b_k| on normal data for any K (the six-product bf16 kernel of gemm_x3.hip: the
// same; a plain fp32 FMA chain: 1.5e-7 .. 2e-7) — tests/test_kernels_gpu.py, unchanged bounds.  Range: elements
// below 2^-29 of their tensor's maximum lose relative (not absolute) precision gradually — the error stays below
// 2^-51 of the maximum per element — where gemm_x3.hip carries fp32's exponent range per element; the caller
// (mapx_gemm_f32) therefore uses this family only for operands whose record it was given.
// Why: three v_mfma_f32_32x32x16_f16 (96 cycles) per product tile instead of six (192), two 2-byte planes in LDS per
// operand instead of three, 3 instead of 5.5 VALU instructions per element for the cut.  The dense layers
// (CrossNetV2 layers.py:197-201, MLPBlock layers.py:173-188, feat_encoder / pred_rfd models.py:74,119-124 and all
// their backward products) are where the fp32 step's time is.
//
// Structure: gemm_x3.hip's 4-wave layouts with the hand-woven K-step (one wave per SIMD; slot z = MFMA z + its
// share of the next tile's cut + at most one memory instruction), K remainder first, edge rows clamped.  Only the
// vector-load case with >= 2 K-steps per slab is built here; everything else stays on gemm_x3.hip.
#include "gemm_h2_common.h"

namespace mapx {

// order in which a k16 half's 2 (WMT + WNT) fragments are read = order in which the MFMAs first need them (tiles
// row-major; per tile lo.hi, hi.lo, hi.hi).  what = 0: operand (0 A, 1 B), 1: plane (0 hi, 1 lo), 2: tile
__host__ __device__ constexpr int h2_frag_order(int q, int what, int wnt) {
  constexpr int first[4][2] = {{0, 1}, {1, 0}, {0, 0}, {1, 1}};          // tile (0,0): A lo, B hi, A hi, B lo
  if (q < 4) return what == 0 ? first[q][0] : what == 1 ? first[q][1] : 0;
  const int r = (q - 4) / 2, w = (q - 4) % 2;
  const bool isB = r < wnt - 1;                       // columns 1 .. WNT-1 of B first (tiles (0, j)), then rows of A
  // a new column of B is needed hi first (lo_a hi_b), a new row of A lo first
  return what == 0 ? (isB ? 1 : 0) : what == 1 ? (isB ? w : 1 - w) : (isB ? r + 1 : r - (wnt - 1) + 1);
}

template <int WR, int WC, int WMT, int WNT, bool A_KC, bool B_KC>
__global__ void __launch_bounds__(64 * WR * WC) gemm_f32h2_kernel(GemmX3Args a_in) {
  constexpr int BM = 32 * WMT * WR, BN = 32 * WNT * WC, NT = 64 * WR * WC;
  GemmX3Args a = a_in;
  if (gridDim.z > 1) {                        // batched: problem blockIdx.z of gridDim.z equal-shaped ones
    a.A = a_in.Az[blockIdx.z];
    a.B = a_in.Bz[blockIdx.z];
    a.C = a_in.Cz[blockIdx.z] + (int64_t)blockIdx.z * a_in.batch_slabs;
    a.amax_a = a_in.amax_az[blockIdx.z];
    a.amax_b = a_in.amax_bz[blockIdx.z];
  }
  using OpA = OperandH2<BM, WMT, A_KC, NT>;
  using OpB = OperandH2<BN, WNT, B_KC, NT>;
  static_assert(!OpA::PARTIAL && !OpB::PARTIAL, "whole rounds of chunks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16_t* const smem = reinterpret_cast<f16_t*>(smem_raw);
  constexpr int kBuf = OpA::LDS_ELEMS + OpB::LDS_ELEMS;

  // the operands' scales (uniform: scalar registers)
  const int na = __builtin_amdgcn_readfirstlane(h2_scale_exp(a.amax_a));
  const int nb = __builtin_amdgcn_readfirstlane(h2_scale_exp(a.amax_b));
  const float sA = pow2f(na), sB = pow2f(nb), k2048 = 2048.f;

  const int nb_tiles = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x, ks = blockIdx.y;
  if (a.xcd_slices) {                         // split-K: an XCD works on ONE k-slice (gemm_x3.hip)
    const int L = blockIdx.x + blockIdx.y * nb_tiles, c = L & 7, slot = L >> 3, ns = gridDim.y;
    ks = c % ns;
    lin = (c / ns) * (nb_tiles / (8 / ns)) + slot;
  } else {
    const int per = nb_tiles / 8;
    if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order
  }
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = ks * a.k_chunk;
  const int kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
  float* __restrict__ C = a.C + (int64_t)ks * a.slab_stride;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int l31 = lane & 31, kh = lane >> 5;
  const int abase = wr * 32 * WMT, bbase = wc * 32 * WNT;

  f32x16 acc[WMT][WNT], cor[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = cor[i][j][r] = 0.f;

  // Two register sets per operand: set (t & 1) carries tile t from its global load (issued inside K-step t-3) to
  // its cut + LDS store (inside K-step t-1).  The K range's remainder goes FIRST (tile 0 is the partial one, cut
  // by the bounds-checked prologue); where the loop would run out of tiles it re-loads the last one.
  OpA la[2];
  OpB lb[2];
  const int nk = (kend - kbeg + kXBK - 1) / kXBK;            // >= 2: the launcher's promise
  const int wk0 = kbeg + (kend - kbeg) - kXBK * (nk - 1);
  {
    auto wload = [&](auto& oa, auto& ob, int t) __attribute__((always_inline)) {
      const int tc = t < nk - 1 ? t : nk - 1, k0 = tc == 0 ? kbeg : wk0 + kXBK * (tc - 1);
      oa.load(a.A, a.lda, m0, a.M, k0, tc == 0 ? wk0 : kend);
      ob.load(a.B, a.ldb, n0, a.N, k0, tc == 0 ? wk0 : kend);
    };
    wload(la[0], lb[0], 0);
    wload(la[1], lb[1], 1);
    la[0].store_masked(smem, sA);
    lb[0].store_masked(smem + OpA::LDS_ELEMS, sB);
    wload(la[0], lb[0], 2);
  }
  __syncthreads();

  // slots of the woven K-step: MFMAs, chunks, units of the cut (a chunk = 4 pairs = 2 pair groups x 3 stages),
  // fragments per k16 half, units before the first MFMA, slots that carry units (the last 4: the last chunk's two
  // LDS stores and two global loads)
  constexpr int kNM = 6 * WMT * WNT, kNCH = OpA::NV + OpB::NV, kU = 8 * kNCH, kFR = 2 * (WMT + WNT);
#ifdef MAPX_H2_PRE
  constexpr int kPre = MAPX_H2_PRE < kU - 1 ? MAPX_H2_PRE : kU - 1, kS = kNM - 4;      // experiment: units ahead of the MFMAs
#else
  constexpr int kPre = kU >= 32 ? 4 : 2, kS = kNM - 4;
#endif
  static_assert(kS >= 1 && kU > kPre, "slot budget");
#ifdef MAPX_H2_ABLATE
  constexpr int kDbg = MAPX_H2_ABLATE;       // 2 no cut / stores / loads, 4 no MFMAs, 16 no LDS stores (cut kept), 32 no global loads, 64 no cut VALU
#else
  constexpr int kDbg = 0;
#endif

  // per chunk: element offset from the K-step's (uniform) operand base, out-of-range rows / columns of an edge tile
  // clamped to 0 (what they contribute lands in outputs the epilogue does not store), and the LDS offset of plane 0
  int64_t goffA[OpA::NV], goffB[OpB::NV];
  int soffA[OpA::NV], soffB[OpB::NV];
#pragma unroll
  for (int i = 0; i < OpA::NV; ++i) {
    int tr, tc;
    OpA::coords(threadIdx.x + i * NT, tr, tc);
    const bool in = (A_KC ? m0 + tr : m0 + tc) < a.M;
    goffA[i] = A_KC ? (int64_t)(in ? m0 + tr : 0) * a.lda + tc : (int64_t)tr * a.lda + (in ? m0 + tc : 0);
    soffA[i] = OpA::lds_off(tr, tc);
  }
#pragma unroll
  for (int i = 0; i < OpB::NV; ++i) {
    int tr, tc;
    OpB::coords(threadIdx.x + i * NT, tr, tc);
    const bool in = (B_KC ? n0 + tr : n0 + tc) < a.N;
    goffB[i] = B_KC ? (int64_t)(in ? n0 + tr : 0) * a.ldb + tc : (int64_t)tr * a.ldb + (in ? n0 + tc : 0);
    soffB[i] = OpA::LDS_ELEMS + OpB::lds_off(tr, tc);
  }

#define MAPX_H_WSTORE(CUR, c, pl)                                                                      \
  do {                                                                                                 \
    constexpr bool isA_ = (c) < OpA::NV;                                                               \
    constexpr int i_ = isA_ ? (c) : (c) - OpA::NV, plane_ = isA_ ? OpA::PLANE : OpB::PLANE;            \
    if (kDbg & 16) break;                                                                              \
    f16_t* const d_ = smem + ((CUR) ^ 1) * kBuf + (isA_ ? soffA[i_] : soffB[i_]) + (pl) * plane_;      \
    const uint32_t* const w_ = (pl) == 0 ? cH[(c) & 1] : cL[(c) & 1];                                  \
    *reinterpret_cast<uint4*>(d_) = make_uint4(w_[0], w_[1], w_[2], w_[3]);                            \
  } while (0)
#define MAPX_H_WLOAD(CUR, c, hf)                                                                       \
  do {                                                                                                 \
    if (kDbg & 32) break;                                                                              \
    constexpr bool isA_ = (c) < OpA::NV;                                                               \
    constexpr int i_ = isA_ ? (c) : (c) - OpA::NV;                                                     \
    const float* const q_ = (isA_ ? wA + goffA[i_] : wB + goffB[i_]) + 4 * (hf);                       \
    if (isA_) la[(CUR) ^ 1].r[i_][hf] = *reinterpret_cast<const float4*>(q_);                          \
    else lb[(CUR) ^ 1].r[i_][hf] = *reinterpret_cast<const float4*>(q_);                               \
  } while (0)
  // unit u of the cut of tile kt+1 (register set CUR^1): chunk u / 8, pair group (u % 8) / 4, stage u % 4; the
  // first four units of chunk c also carry chunk c-1's two LDS stores and two global loads (its registers are free:
  // unit 0 of each pair group was their last reader)
#define MAPX_H_CUT_UNIT(CUR, u)                                                                        \
  do {                                                                                                 \
    constexpr int c_ = (u) / 8, pg_ = ((u) % 8) / 4, st_ = (u) % 4, m_ = (u) % 8;                      \
    constexpr bool isA_ = c_ < OpA::NV;                                                                \
    constexpr int i_ = isA_ ? c_ : c_ - OpA::NV;                                                       \
    if ((kDbg & 128) && !isA_) {           /* ablation: operand B is not staged at all */              \
      if (c_ == OpA::NV && m_ < 2) MAPX_H_WSTORE(CUR, (c_ > 0 ? c_ - 1 : 0), (m_ < 2 ? m_ : 0));       \
      if (c_ == OpA::NV && m_ >= 2 && m_ < 4) MAPX_H_WLOAD(CUR, (c_ > 0 ? c_ - 1 : 0), (m_ >= 2 && m_ < 4 ? m_ - 2 : 0)); \
      break;                                                                                           \
    }                                                                                                  \
    if (kDbg & 64) {        /* ablation: no VALU, the raw bits are stored */                           \
      const float4 v_ = isA_ ? la[(CUR) ^ 1].r[i_][pg_] : lb[(CUR) ^ 1].r[i_][pg_];                    \
      if (st_ == 0) { cH[c_ & 1][2 * pg_] = __float_as_uint(v_.x); cH[c_ & 1][2 * pg_ + 1] = __float_as_uint(v_.y);    \
                      cL[c_ & 1][2 * pg_] = __float_as_uint(v_.z); cL[c_ & 1][2 * pg_ + 1] = __float_as_uint(v_.w); }  \
    } else {                                                                                           \
    if (st_ == 0) {                                                                                    \
      const float4 v_ = isA_ ? la[(CUR) ^ 1].r[i_][pg_] : lb[(CUR) ^ 1].r[i_][pg_];                    \
      h2_unit0(v_.x, v_.y, v_.z, v_.w, isA_ ? sA : sB, cr);                                            \
    }                                                                                                  \
    if (st_ == 1) h2_unit1(cr, cH[c_ & 1][2 * pg_], cH[c_ & 1][2 * pg_ + 1]);                          \
    if (st_ == 2) h2_unit2(cr, cH[c_ & 1][2 * pg_], cH[c_ & 1][2 * pg_ + 1], k2048);                   \
    if (st_ == 3) h2_unit3(cr, k2048, cL[c_ & 1][2 * pg_], cL[c_ & 1][2 * pg_ + 1]);                   \
    }                                                                                                  \
    if (c_ > 0 && m_ < 2) MAPX_H_WSTORE(CUR, (c_ > 0 ? c_ - 1 : 0), m_);                               \
    if (c_ > 0 && m_ >= 2 && m_ < 4) MAPX_H_WLOAD(CUR, (c_ > 0 ? c_ - 1 : 0), (m_ >= 2 && m_ < 4 ? m_ - 2 : 0)); \
  } while (0)
#define MAPX_H_KSTEP(CUR, kt)                                                                          \
  do {                                                                                                 \
    const f16_t* const As_cur = smem + (CUR) * kBuf;                                                   \
    const f16_t* const Bs_cur = As_cur + OpA::LDS_ELEMS;                                               \
    f16x8 fa[2][2][WMT], fb[2][2][WNT];           /* [k16 half][plane hi / lo][tile] */                \
    unroll_seq([&](auto qc) __attribute__((always_inline)) {                                           \
      constexpr int q = decltype(qc)::value, op = h2_frag_order(q, 0, WNT), pl = h2_frag_order(q, 1, WNT), \
                    t = h2_frag_order(q, 2, WNT);                                                      \
      if (op == 0) fa[0][pl][t] = OpA::frag1(As_cur, pl, abase, lane, 0, t);                           \
      else fb[0][pl][t] = OpB::frag1(Bs_cur, pl, bbase, lane, 0, t);                                   \
    }, std::make_integer_sequence<int, kFR>{});                                                        \
    uint32_t cH[2][4], cL[2][4];                  /* [chunk parity][pair] */                           \
    CutRegs cr;                                                                                        \
    const int wk_ = wk0 + kXBK * (((kt) + 3 < nk - 1 ? (kt) + 3 : nk - 1) - 1);   /* tile min(kt+3, nk-1) */ \
    const float* const wA = a.A + (int64_t)wk_ * (A_KC ? 1 : a.lda);                                   \
    const float* const wB = a.B + (int64_t)wk_ * (B_KC ? 1 : a.ldb);                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (!(kDbg & 2)) {                                                                                 \
      unroll_seq([&](auto uc) __attribute__((always_inline)) {                                         \
        MAPX_H_CUT_UNIT(CUR, decltype(uc)::value);                                                     \
      }, std::make_integer_sequence<int, kPre>{});                                                     \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    unroll_seq([&](auto zc) __attribute__((always_inline)) {                                           \
      constexpr int z = decltype(zc)::value;                                                           \
      constexpr int h = z / (kNM / 2), t3 = (z % (kNM / 2)) / 3, i = t3 / WNT, j = t3 % WNT, term = z % 3; \
      if (!(kDbg & 4)) {                                                                               \
        if (term == 0) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[h][1][i], fb[h][0][j], cor[i][j], 0, 0, 0); \
        if (term == 1) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[h][0][i], fb[h][1][j], cor[i][j], 0, 0, 0); \
        if (term == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[h][0][i], fb[h][0][j], acc[i][j], 0, 0, 0); \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if constexpr (z < kNM / 2) {            /* the second k16 half's fragments, under the first half's MFMAs */ \
        constexpr int q0 = z * kFR / (kNM / 2), q1 = (z + 1) * kFR / (kNM / 2);                        \
        unroll_seq([&](auto qc) __attribute__((always_inline)) {                                       \
          constexpr int q = q0 + decltype(qc)::value, op = h2_frag_order(q, 0, WNT), pl = h2_frag_order(q, 1, WNT), \
                        t = h2_frag_order(q, 2, WNT);                                                  \
          if (op == 0) fa[1][pl][t] = OpA::frag1(As_cur, pl, abase, lane, 1, t);                       \
          else fb[1][pl][t] = OpB::frag1(Bs_cur, pl, bbase, lane, 1, t);                               \
        }, std::make_integer_sequence<int, q1 - q0>{});                                                \
      }                                                                                                \
      if (!(kDbg & 2)) {                                                                               \
        if constexpr (z < kS) {                                                                        \
          constexpr int u0 = kPre + z * (kU - kPre) / kS, u1 = kPre + (z + 1) * (kU - kPre) / kS;      \
          unroll_seq([&](auto uc) __attribute__((always_inline)) {                                     \
            MAPX_H_CUT_UNIT(CUR, u0 + decltype(uc)::value);                                            \
          }, std::make_integer_sequence<int, u1 - u0>{});                                              \
        }                                                                                              \
        if constexpr (z >= kS && z < kS + 2) { if (!(kDbg & 128)) MAPX_H_WSTORE(CUR, kNCH - 1, (z >= kS && z < kS + 2 ? z - kS : 0)); } \
        if constexpr (z >= kS + 2) { if (!(kDbg & 128)) MAPX_H_WLOAD(CUR, kNCH - 1, (z >= kS + 2 ? z - kS - 2 : 0)); } \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }, std::make_integer_sequence<int, kNM>{});                                                        \
    __syncthreads();                                                                                   \
  } while (0)

  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    MAPX_H_KSTEP(0, kt);
    MAPX_H_KSTEP(1, kt + 1);
  }
  if (kt < nk) MAPX_H_KSTEP(0, kt);
#undef MAPX_H_KSTEP
#undef MAPX_H_CUT_UNIT
#undef MAPX_H_WLOAD
#undef MAPX_H_WSTORE

  // (acc + 2^-11 cor) 2^-(na + nb) -> the fp32 tile in LDS -> the epilogues of gemm_x3_common.h
  float* const tile = reinterpret_cast<float*>(smem_raw);
  constexpr int LDT = BN + 4;
  const int dn = -(na + nb);
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(abase + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2)) * LDT + bbase + 32 * j + l31] =
            __builtin_ldexpf(__builtin_fmaf(cor[i][j][r], 0x1p-11f, acc[i][j][r]), dn);
  __syncthreads();
  epilogue_dispatch<BM, BN, NT>(a, C, tile, m0, n0);
}

template <int WR, int WC, int WMT, int WNT, bool A_KC, bool B_KC>
static hipError_t launch_one_h2(const GemmX3Args& a, int nsplit, hipStream_t stream, int batch) {
  constexpr int BM = 32 * WR * WMT, BN = 32 * WC * WNT, NT = 64 * WR * WC;
  using OpA = OperandH2<BM, WMT, A_KC, NT>;
  using OpB = OperandH2<BN, WNT, B_KC, NT>;
  constexpr size_t ops = (size_t)2 * (OpA::LDS_ELEMS + OpB::LDS_ELEMS) * sizeof(f16_t);
  constexpr size_t epi = ((size_t)BM * (BN + 4) + 4 * 256) * sizeof(float);     // the fp32 tile + the column-sum rows
  constexpr size_t lds = ops > epi ? ops : epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto* fn = &gemm_f32h2_kernel<WR, WC, WMT, WNT, A_KC, B_KC>;
  static hipError_t raised = lds > 65536
      ? hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipSuccess;
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(fn, dim3(a.tiles_m * a.tiles_n, nsplit, batch), dim3(NT), lds, stream, a);
  return hipSuccess;
}

template <bool A_KC, bool B_KC>
static hipError_t launch_layout_h2(GemmX3Args& a, int tile, int nsplit, hipStream_t stream, int batch) {
  if (tile == 3) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
    return launch_one_h2<2, 2, 2, 2, A_KC, B_KC>(a, nsplit, stream, batch);
  }
  if (tile == 2) {            // 8 waves, two per SIMD
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
    return launch_one_h2<2, 4, 2, 1, A_KC, B_KC>(a, nsplit, stream, batch);
  }
  if (tile == 1) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 63) / 64;
    return launch_one_h2<2, 2, 2, 1, A_KC, B_KC>(a, nsplit, stream, batch);
  }
  a.tiles_m = (a.M + 63) / 64; a.tiles_n = (a.N + 63) / 64;
  return launch_one_h2<2, 2, 1, 1, A_KC, B_KC>(a, nsplit, stream, batch);
}

// Called by gemm_f32x3_launch (gemm_x3.hip) with the argument block it has prepared (operands, epilogue, split-K
// slabs, k_chunk) when both operands come with a magnitude record.  Returns false when this family does not
// build the case (scalar-load operands, a slab of one K-step, the 8-wave layout asked for): the caller goes on
// with its own kernels.
bool gemm_f32h2_try(GemmX3Args& g, int a_kc, int b_kc, bool vec, int tile, bool hinted, bool xcd_on, int nsplit,
                    int batch, hipStream_t stream, hipError_t* err) {
  static const bool on = [] { const char* e = getenv("MAPX_GEMM_H2"); return !e || atoi(e) != 0; }();
  if (!on || !vec) return false;
  if (g.K - (int64_t)g.k_chunk * (nsplit - 1) <= kXBK) return false;        // every slab >= 2 K-steps
  if (tile != 0 && tile != 1 && tile != 2) tile = 3;
  if (!hinted && g.epi != MAPX_EPI_BWD_FUSED && g.epi != MAPX_EPI_RELU_MASK_COLSUM) {
    // This family's own choice of layout (tools/gemm_h2_bench.py, round 4): with half the MFMA time per K-step the
    // 128 x 128 tile pays from 128 workgroups on (the six-product kernels: 160), and a problem too narrow for it
    // (N = 368) takes 128 x 64 before 64 x 64 when that gives >= 160 workgroups (1000 x 1000 x 4096 unsplit: 128
    // of them took 83 us, 256 tiles of 64 x 64 take 71).  (The two epilogues that leave one partial row per 128-row tile keep
    // what the caller chose.)
    auto blocks = [&](int bm, int bn) { return ceil_div(g.M, bm) * ceil_div(g.N, bn) * nsplit * batch; };
    static const int forced = [] { const char* e = getenv("MAPX_H2_TILE"); return e ? atoi(e) : -1; }();   // A/B switch
    int mine = blocks(128, 128) >= 128 ? 3 : (blocks(128, 64) >= 160 ? 1 : 0);
    if (forced >= 0 && mine == 3) mine = forced;
    if (mine != tile && nsplit > 1 && batch == 1) {       // the k-slice dealing depends on the tile grid (gemm_x3.hip)
      const int bm = mine == 0 ? 64 : 128, bn = mine == 3 ? 128 : 64;
      const int64_t nb1 = ceil_div(g.M, bm) * ceil_div(g.N, bn);
      g.xcd_slices = (xcd_on && 8 % nsplit == 0 && nb1 % (8 / nsplit) == 0 && (nb1 * nsplit) % 8 == 0) ? 1 : 0;
    }
    tile = mine;
  }
  if (a_kc && b_kc) *err = launch_layout_h2<true, true>(g, tile, nsplit, stream, batch);
  else if (a_kc) *err = launch_layout_h2<true, false>(g, tile, nsplit, stream, batch);
  else *err = launch_layout_h2<false, false>(g, tile, nsplit, stream, batch);
  return true;
}

}  // namespace mapx
