// fp32 GEMM on the bf16 matrix cores: every fp32 operand element is cut into three bf16 pieces,
//   a = a_hi + a_mid + a_lo      (8 + 8 + 8 significant bits, each piece rounded to nearest)
// and a product a*b is formed from the six piece products that matter,
//   a*b = a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_mid b_mid + a_lo b_hi) + O(2^-24 |a b|),
// each of them EXACT in the MFMA's fp32 accumulator (8 x 8 significant bits), summed in fp32: the
// leading products in one accumulator, the five corrections in a second one (they would be rounded
// away against a grown sum), the two added once at the end.  What is dropped (a_mid b_lo, a_lo b_mid,
// a_lo b_lo) is below 2^-23 of |a b| — the size of one fp32 rounding, which the plain fp32 FMA chain
// of v_mfma_f32_32x32x2_f32 makes once per product as well — and has no preferred sign (pieces are
// rounded to nearest).  Six v_mfma_f32_32x32x16_bf16 do the work of eight v_mfma_f32_32x32x2_f32 in 192 instead of
// 512 cycles: the dense layers of the step (CrossNetV2 layers.py:197-201, MLPBlock layers.py:173-188,
// feat_encoder / pred_rfd / fc_out models.py:74,119-124,304, and all their backward products) are
// MFMA-bound, so this is where the fp32 step's time is.  Inputs, outputs, bias, epilogues and the
// accumulation stay fp32; tensors in HBM are plain fp32 (the cut happens between the global load and
// the LDS store, in the shadow of the MFMAs).  Measured error against fp64: see tests (inside the bound the
// fp32-MFMA instruction's own rounding chain gave; that kernel family, round 1's, was removed in round 3).
//
// Interface of mapx_gemm_f32 (gemm.hip): operand storage flags (A_KC / B_KC), epilogues, split-K slabs.
// Tiling: 256 threads = 2x2 waves, block tile (64 WMT) x (64 WNT), BK = 32; per operand three bf16
// planes in LDS, laid out like gemm_bf16.hip's single plane (k-contiguous: [row][32 + 8], one
// ds_read_b128 per fragment, conflict-free at the 80-byte row stride; k-strided: [k][rows + 32], two
// ds_read_b64_tr_b16 per fragment).  One barrier per K-step, two LDS buffers; the epilogue goes
// through LDS as fp32 rows and moves every operand with 16-byte accesses.
#include "amax.h"
#include "gemm_x3_common.h"
#include "gemm_grouped.h"

namespace mapx {

// One operand: global fp32 tile -> registers (chunks of 8 floats) -> three bf16 planes in LDS -> fragments.
// VEC: leading dimension % 4 == 0, 16-B aligned base, contiguous extent % 8 == 0 (a chunk is all-in or
// all-out); otherwise 8 scalar loads with per-element predicates.
template <int ROWS, int T, bool KC, bool VEC, int NT>
struct OperandX3 {
  static constexpr int LD = KC ? kXBK + 8 : ROWS + 32;            // bf16 elements per stored row
  static constexpr int PLANE = KC ? ROWS * LD : kXBK * LD;         // elements of one plane
  static constexpr int LDS_ELEMS = 3 * PLANE;
  static constexpr int CPR = KC ? kXBK / 8 : ROWS / 8;             // chunks per stored row
  static constexpr int TOTAL = ROWS * kXBK / 8;                    // chunks per tile
  static constexpr int NV = (TOTAL + NT - 1) / NT;                 // chunks per thread per tile (NT threads)
  static constexpr bool PARTIAL = TOTAL % NT != 0;                 // the last round is for the first waves only
  static_assert(!PARTIAL || TOTAL % 64 == 0, "a partial round must end on a wave boundary");
  float4 r[NV][2];
  bool ok[NV];

  __device__ static inline void coords(int f, int& row, int& col) {
    row = f / CPR;
    col = (f % CPR) * 8;
  }

  __device__ inline void load_one(int i, const float* __restrict__ g, int64_t ld, int row0, int nrows, int k0,
                                  int kend) {
    int tr, tc;
    coords(threadIdx.x + i * NT, tr, tc);
    const int gr = (KC ? row0 : k0) + tr, gc = (KC ? k0 : row0) + tc;
    const int rlim = KC ? nrows : kend, clim = KC ? kend : nrows;
    const bool rok = gr < rlim;
    const float* p = g + (int64_t)(rok ? gr : 0) * ld;
    if (VEC) {
      ok[i] = rok && gc < clim;
      const float* q = p + (ok[i] ? gc : 0);
      r[i][0] = *reinterpret_cast<const float4*>(q);
      r[i][1] = *reinterpret_cast<const float4*>(q + 4);
    } else {
      ok[i] = true;
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool oke = rok && gc + e < clim;
        const float v = p[oke ? gc + e : 0];
        x[e] = oke ? v : 0.f;
      }
      r[i][0] = make_float4(x[0], x[1], x[2], x[3]);
      r[i][1] = make_float4(x[4], x[5], x[6], x[7]);
    }
  }
  __device__ inline void load(const float* __restrict__ g, int64_t ld, int row0, int nrows, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (PARTIAL && threadIdx.x + i * NT >= TOTAL) break;
      load_one(i, g, ld, row0, nrows, k0, kend);
    }
  }

  // cut + store (zero fill applied here, so that no ALU touches a load's result before it is needed)
  template <bool MASK>
  __device__ inline void store(bf16_t* __restrict__ s) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tr, tc;
      if (PARTIAL && threadIdx.x + i * NT >= TOTAL) break;
      coords(threadIdx.x + i * NT, tr, tc);
      const bool keep = !MASK || ok[i];
      const float x[8] = {keep ? r[i][0].x : 0.f, keep ? r[i][0].y : 0.f, keep ? r[i][0].z : 0.f, keep ? r[i][0].w : 0.f,
                          keep ? r[i][1].x : 0.f, keep ? r[i][1].y : 0.f, keep ? r[i][1].z : 0.f, keep ? r[i][1].w : 0.f};
      uint4 hi, mid, lo;
      cut3(x, hi, mid, lo);
      bf16_t* d = s + tr * LD + tc;
      *reinterpret_cast<uint4*>(d) = hi;
      *reinterpret_cast<uint4*>(d + PLANE) = mid;
      *reinterpret_cast<uint4*>(d + 2 * PLANE) = lo;
    }
  }

  // fragment of k16-step s2 (k = 16 s2 + 8 (lane >> 5) + j) of plane `pl` for the wave's tile t
  __device__ static inline bf16x8 frag1(const bf16_t* __restrict__ s, int pl, int base, int lane, int s2, int t) {
    const int l31 = lane & 31, kh = lane >> 5;
    const bf16_t* sp = s + pl * PLANE;
    if (KC) return *reinterpret_cast<const bf16x8*>(sp + (base + 32 * t + l31) * LD + 16 * s2 + 8 * kh);
    const int q = (lane >> 2) & 3, p = lane & 3, half = (lane >> 4) & 1;
    const bf16_t* a0 = sp + (16 * s2 + 8 * kh + q) * LD + base + 32 * t + 16 * half + 4 * p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * LD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  __device__ static inline void frags(const bf16_t* __restrict__ s, int pl, int base, int lane, int s2,
                                      bf16x8 (&f)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) f[t] = frag1(s, pl, base, lane, s2, t);
  }
};

// WR x WC waves, each owning WMT x WNT MFMA tiles of 32 x 32: block tile (32 WMT WR) x (32 WNT WC).
// 8 waves (two per SIMD) on a 128 x 128 tile: while one wave of a SIMD cuts and stores its share of
// the next tile (VALU + LDS), its partner's MFMAs keep the matrix pipe busy — the overlap that one
// wave per SIMD only gets from a perfect instruction interleave.
template <int WR, int WC, int WMT, int WNT, bool A_KC, bool B_KC, bool VEC>
__global__ void __launch_bounds__(64 * WR * WC) gemm_f32x3_kernel(GemmX3Args a_in) {
  constexpr int BM = 32 * WMT * WR, BN = 32 * WNT * WC, NT = 64 * WR * WC;
  GemmX3Args a = a_in;
  if (gridDim.z > 1) {                        // batched: problem blockIdx.z of gridDim.z equal-shaped ones
    a.A = a_in.Az[blockIdx.z];
    a.B = a_in.Bz[blockIdx.z];
    a.C = a_in.Cz[blockIdx.z] + (int64_t)blockIdx.z * a_in.batch_slabs;
  }
  using OpA = OperandX3<BM, WMT, A_KC, VEC, NT>;
  using OpB = OperandX3<BN, WNT, B_KC, VEC, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* const smem = reinterpret_cast<bf16_t*>(smem_raw);
  constexpr int kBuf = OpA::LDS_ELEMS + OpB::LDS_ELEMS;

  const int nb = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x, ks = blockIdx.y;
  if (a.xcd_slices) {
    // Split-K: workgroups reach the XCDs round-robin in launch order (x fastest), so XCD c = L % 8.  Dealing
    // the tiles as below gave every XCD all K of 1/8 of the tiles: its L2 fetched one operand WHOLE (8 x 16 MB
    // for the 1000 x 1000 x 4096 weight gradient, PMC: 138 MB per launch for 36 MB of operands and output).
    // Here XCD c takes k-slice c % nsplit of the tiles of group c / nsplit: 1/nsplit of both operands' k range,
    // the rows / columns of 1/(8/nsplit) of the tiles.
    const int L = blockIdx.x + blockIdx.y * nb, c = L & 7, slot = L >> 3, ns = gridDim.y;
    ks = c % ns;
    lin = (c / ns) * (nb / (8 / ns)) + slot;
  } else {
    const int per = nb / 8;
    if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order (see gemm.hip)
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

  // Two accumulators per output tile: `acc` takes the leading products a_hi b_hi, `cor` the five
  // correction products (2^-8 and 2^-16 of the leading ones).  Added into ONE accumulator the
  // corrections are rounded away as soon as it has grown (same-sign data, K = 4096: 190 ulp of error,
  // measured); kept apart they sum among their own size and join the result once, in the epilogue.
  f32x16 acc[WMT][WNT], cor[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = cor[i][j][r] = 0.f;

  // Two register sets per operand: set (t & 1) carries tile t from its global load (issued at the
  // start of K-step t-2) to its cut + LDS store (inside K-step t-1).  A K-step is written
  // compute-first — fragments and the 48 MFMAs of tile kt, then the cut (~180 VALU instructions) and
  // the 12 LDS stores of tile kt+1 — and, in the branch-free steady-state body, a
  // sched_group_barrier pattern deals those VALU / LDS instructions out between the MFMAs: an MFMA
  // holds the SIMD's issue port for 8 of its 32 cycles, the cut runs in the other 24.
  OpA la[2];
  OpB lb[2];
  const int nk = (kend - kbeg + kXBK - 1) / kXBK;
  const bool interior = VEC && (m0 + BM <= a.M) && (n0 + BN <= a.N);
#define MAPX_X_LOAD(SET, t)                                              \
  do {                                                                   \
    la[SET].load(a.A, a.lda, m0, a.M, kbeg + (t) * kXBK, kend);          \
    lb[SET].load(a.B, a.ldb, n0, a.N, kbeg + (t) * kXBK, kend);          \
  } while (0)
#define MAPX_X_STORE(SET, buf, MASK)                                     \
  do {                                                                   \
    la[SET].template store<MASK>(smem + (buf) * kBuf);                   \
    lb[SET].template store<MASK>(smem + (buf) * kBuf + OpA::LDS_ELEMS);  \
  } while (0)
  constexpr bool kWeave = (WR * WC == 4 && VEC);
  // the woven K-step (below): MFMA slots, units of the cut, fragments per k16 half, slots that carry units
  constexpr int kNM = 12 * WMT * WNT, kNCH = OpA::NV + OpB::NV, kU = 12 * kNCH, kFR = 3 * (WMT + WNT);
  constexpr int kPre = 6, kS = kNM - 4;
  // The woven layout wants every K-step of its loop alike.  So (nk >= 2) the K range's remainder goes
  // FIRST: tile 0 is the partial one, [kbeg, kbeg + rem) with rem in 8..32, cut here with the bounds-
  // checked load and the masked store; tiles 1.. are full ones starting at wk0 = kbeg + rem.  And where the
  // loop would run out of tiles it repeats the last one: a cut into the LDS buffer nobody reads any more,
  // loads of a tile that is in bounds.
  // (nk >= 2 is the launcher's promise: it gives slabs of one K-step to the 8-wave layout.)
  const int wk0 = kbeg + (kend - kbeg) - kXBK * (nk - 1);
  if constexpr (kWeave) {
    auto wload = [&](auto& oa, auto& ob, int t) __attribute__((always_inline)) {
      const int tc = t < nk - 1 ? t : nk - 1, k0 = tc == 0 ? kbeg : wk0 + kXBK * (tc - 1);
      oa.load(a.A, a.lda, m0, a.M, k0, tc == 0 ? wk0 : kend);
      ob.load(a.B, a.ldb, n0, a.N, k0, tc == 0 ? wk0 : kend);
    };
    wload(la[0], lb[0], 0);
    wload(la[1], lb[1], 1);
    MAPX_X_STORE(0, 0, true);
    wload(la[0], lb[0], 2);
  } else {
    if (nk > 0) MAPX_X_LOAD(0, 0);
    if (nk > 1) MAPX_X_LOAD(1, 1);
    if (nk > 0) MAPX_X_STORE(0, 0, true);
    if (nk > 2) MAPX_X_LOAD(0, 2);
  }
  __syncthreads();
  // K-step on LDS buffer CUR (= kt & 1, literal): at its start set CUR^1 holds tile kt+1 (landed),
  // set CUR holds tile kt+2 (in flight, issued one step ago)
  // Timing experiments (tools/x3_ablate.sh): -DMAPX_X3_ABLATE=<bits> builds the K loop without some of its
  // phases (1 no global loads, 2 no cut / LDS stores / loads, 4 no MFMAs; woven loop only: 8 B's cut
  // skipped as if B came pre-split, 16 no LDS stores, 32 no global loads); results are then wrong.
#ifdef MAPX_X3_ABLATE
  constexpr int kDbg = MAPX_X3_ABLATE;       // compile-time: a run-time switch would put branches into the slots
#else
  constexpr int kDbg = 0;
#endif
#define MAPX_X_COMPUTE(CUR)                                                                            \
  do {                                                                                                 \
    const bf16_t* const As_cur = smem + (CUR) * kBuf;                                                  \
    const bf16_t* const Bs_cur = As_cur + OpA::LDS_ELEMS;                                              \
    _Pragma("unroll") for (int s2 = 0; s2 < kXBK / 16; ++s2) {                                         \
      bf16x8 ah[WMT], am[WMT], al[WMT], bh[WNT], bm[WNT], bl[WNT];                                     \
      OpA::frags(As_cur, 0, abase, lane, s2, ah);                                                      \
      OpB::frags(Bs_cur, 0, bbase, lane, s2, bh);                                                      \
      OpA::frags(As_cur, 1, abase, lane, s2, am);                                                      \
      OpB::frags(Bs_cur, 1, bbase, lane, s2, bm);                                                      \
      OpA::frags(As_cur, 2, abase, lane, s2, al);                                                      \
      OpB::frags(Bs_cur, 2, bbase, lane, s2, bl);                                                      \
      if (!(kDbg & 4))                                                                                \
      _Pragma("unroll") for (int i = 0; i < WMT; ++i)                                                  \
        _Pragma("unroll") for (int j = 0; j < WNT; ++j) {                                              \
          f32x16 c = cor[i][j];                 /* smallest terms first */                             \
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], c, 0, 0, 0);                       \
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], c, 0, 0, 0);                       \
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], c, 0, 0, 0);                       \
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], c, 0, 0, 0);                       \
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], c, 0, 0, 0);                       \
          cor[i][j] = c;                                                                               \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);       \
        }                                                                                              \
    }                                                                                                  \
  } while (0)
#define MAPX_X_STAGE(CUR, kt, STEADY, MASK)                                                            \
  do {                                                                                                 \
    if (((STEADY) || (kt) + 1 < nk) && !(kDbg & 2)) MAPX_X_STORE((CUR) ^ 1, (CUR) ^ 1, MASK);         \
    if (((STEADY) || (kt) + 3 < nk) && !(kDbg & 1)) MAPX_X_LOAD((CUR) ^ 1, (kt) + 3);                 \
  } while (0)
  // LATE (literal) swaps the two phases.  Tried for waves 4-7 (so one SIMD resident's MFMAs run beside
  // the other's cut): 1.97 us per K-step against 1.90 without — the residents are not in lockstep here.
#define MAPX_X_KSTEP(CUR, kt, STEADY, MASK, LATE)                                                      \
  do {                                                                                                 \
    if (LATE) {                                                                                        \
      MAPX_X_STAGE(CUR, kt, STEADY, MASK);                                                             \
      MAPX_X_COMPUTE(CUR);                                                                             \
    } else {                                                                                           \
      MAPX_X_COMPUTE(CUR);                                                                             \
      MAPX_X_STAGE(CUR, kt, STEADY, MASK);                                                             \
    }                                                                                                  \
    __syncthreads();                                                                                   \
  } while (0)
  // One wave per SIMD (kWeave: the 4-wave layouts): nothing but the ORDER of the wave's own instructions
  // can put the cut of tile kt+1 into the shadow of tile kt's MFMAs, and left alone the compiler emits all
  // MFMAs, then the whole cut.  An MFMA holds the SIMD's vector issue for 8 of its 32 cycles and every
  // other instruction costs about 4 (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): 5 instructions
  // per MFMA gap hide.  So a K-step is kNM = 12 WMT WNT slots fenced by sched_barrier(0); slot z = MFMA z +
  // its share of the cut's UNITS (chunks x 4 pairs x 3 pieces, of 5, 5 and 1 VALU) + at most one memory
  // instruction.  For the 128 x 128 tile (48 MFMAs, 4 chunks = 48 units):
  //   before MFMA 0   the first k16 half's 12 fragment reads, then units 0..5 under their latency
  //   slots 0..41     unit 6 + z;  even slots < 24: a fragment of the second k16 half (for MFMAs 24..47);
  //                   the 1-VALU units of chunk c carry chunk c-1's three LDS stores and its two global loads
  //   slots 42..45    the last chunk's stores and loads
  // (128 x 64: 24 slots for 36 units, 1 or 2 per slot; 64 x 64: 12 slots for 24 units — those two are bound
  // by the cut's issue slots, not by the MFMAs.)
  // Out-of-range rows / columns of an edge tile are not masked here: their addresses are clamped to the
  // operand's first row / column and what they contribute lands in outputs the epilogue does not store
  // (only the K bound needs zeros, and the prologue has dealt with it).
#define MAPX_X_WSTORE(CUR, c, pl)                                                                      \
  do {                                                                                                 \
    if (kDbg & 16) break;                                                                              \
    constexpr bool isA_ = (c) < OpA::NV;                                                               \
    constexpr int i_ = isA_ ? (c) : (c) - OpA::NV, plane_ = isA_ ? OpA::PLANE : OpB::PLANE;            \
    bf16_t* const d_ = smem + ((CUR) ^ 1) * kBuf + (isA_ ? soffA[i_] : soffB[i_]) + (pl) * plane_;     \
    const uint32_t* const w_ = (pl) == 0 ? cH[(c) & 1] : (pl) == 1 ? cM[(c) & 1] : cL[(c) & 1];        \
    *reinterpret_cast<uint4*>(d_) = make_uint4(w_[0], w_[1], w_[2], w_[3]);                            \
  } while (0)
#define MAPX_X_WLOAD(CUR, c)                                                                           \
  do {                                                                                                 \
    if (kDbg & 32) break;                                                                              \
    constexpr bool isA_ = (c) < OpA::NV;                                                               \
    constexpr int i_ = isA_ ? (c) : (c) - OpA::NV;                                                     \
    const float* const q_ = isA_ ? wA + goffA[i_] : wB + goffB[i_];                                    \
    if (isA_) {                                                                                        \
      la[(CUR) ^ 1].r[i_][0] = *reinterpret_cast<const float4*>(q_);                                   \
      la[(CUR) ^ 1].r[i_][1] = *reinterpret_cast<const float4*>(q_ + 4);                               \
    } else {                                                                                           \
      lb[(CUR) ^ 1].r[i_][0] = *reinterpret_cast<const float4*>(q_);                                   \
      lb[(CUR) ^ 1].r[i_][1] = *reinterpret_cast<const float4*>(q_ + 4);                               \
    }                                                                                                  \
  } while (0)
#define MAPX_X_CUT_UNIT(CUR, u)                                                                                  \
  do {                                                                                                 \
    constexpr int c_ = (u) / 12, e_ = ((u) % 12) / 3, st_ = (u) % 3;                                   \
    constexpr bool isA_ = c_ < OpA::NV;                                                                \
    constexpr int i_ = isA_ ? c_ : c_ - OpA::NV;                                                       \
    if ((kDbg & 8) && !isA_) {      /* ablation: B as if it came pre-split (no VALU; wrong results) */   \
      const float4 v_ = lb[(CUR) ^ 1].r[i_][e_ >> 1];                                                  \
      if (st_ == 0) cH[c_ & 1][e_] = __float_as_uint(v_.x);                                            \
      if (st_ == 1) cM[c_ & 1][e_] = __float_as_uint(v_.y);                                            \
      if (st_ == 2) cL[c_ & 1][e_] = __float_as_uint(v_.z);                                            \
    } else if (st_ == 0) {                                                                             \
      const float4 v_ = isA_ ? la[(CUR) ^ 1].r[i_][e_ >> 1] : lb[(CUR) ^ 1].r[i_][e_ >> 1];            \
      piece((e_ & 1) ? v_.z : v_.x, (e_ & 1) ? v_.w : v_.y, cH[c_ & 1][e_], cr0, cr1);                 \
    } else if (st_ == 1) {                                                                             \
      float t0_, t1_;                                                                                  \
      piece(cr0, cr1, cM[c_ & 1][e_], t0_, t1_);                                                       \
      cr0 = t0_; cr1 = t1_;                                                                            \
    } else {                                                                                           \
      cL[c_ & 1][e_] = pk_bf16(cr0, cr1);                                                              \
    }                                                                                                  \
    if (st_ == 2) {                                                                                    \
      if (c_ > 0 && e_ < 3) MAPX_X_WSTORE(CUR, (c_ > 0 ? c_ - 1 : 0), e_);                             \
      if (c_ > 0 && e_ == 3) MAPX_X_WLOAD(CUR, (c_ > 0 ? c_ - 1 : 0));                                 \
    }                                                                                                  \
  } while (0)
#define MAPX_X_WSLOTS(CUR)                                                                             \
  unroll_seq([&](auto zc) __attribute__((always_inline)) {                                             \
    constexpr int z = decltype(zc)::value;                                                             \
    constexpr int h = z / (kNM / 2), t4 = (z % (kNM / 2)) / 6, i = t4 / WNT, j = t4 % WNT, term = z % 6; \
    if (!(kDbg & 4)) {                                                                                 \
    if (term == 0) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][2][i], fb[h][0][j], cor[i][j], 0, 0, 0); \
    if (term == 1) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0][i], fb[h][2][j], cor[i][j], 0, 0, 0); \
    if (term == 2) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][1][i], fb[h][1][j], cor[i][j], 0, 0, 0); \
    if (term == 3) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][1][i], fb[h][0][j], cor[i][j], 0, 0, 0); \
    if (term == 4) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0][i], fb[h][1][j], cor[i][j], 0, 0, 0); \
    if (term == 5) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0][i], fb[h][0][j], acc[i][j], 0, 0, 0); \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if constexpr (z < kNM / 2) {                                                                       \
      constexpr int q0 = z * kFR / (kNM / 2), q1 = (z + 1) * kFR / (kNM / 2);                          \
      if constexpr (q0 < q1) {                                                                         \
        constexpr int op = frag_order(q0, 0, WNT), pl = frag_order(q0, 1, WNT), t = frag_order(q0, 2, WNT); \
        if (op == 0) fa[1][pl][t] = OpA::frag1(As_cur, pl, abase, lane, 1, t);                         \
        else fb[1][pl][t] = OpB::frag1(Bs_cur, pl, bbase, lane, 1, t);                                 \
      }                                                                                                \
    }                                                                                                  \
    if (!(kDbg & 2)) {                                                                                 \
      if constexpr (z < kS) {                                                                          \
        constexpr int u0 = kPre + z * (kU - kPre) / kS, u1 = kPre + (z + 1) * (kU - kPre) / kS;        \
        if constexpr (u0 < u1) MAPX_X_CUT_UNIT(CUR, u0);                                               \
        if constexpr (u0 + 1 < u1) MAPX_X_CUT_UNIT(CUR, (u0 + 1 < u1 ? u0 + 1 : 0));                   \
        if constexpr (u0 + 2 < u1) MAPX_X_CUT_UNIT(CUR, (u0 + 2 < u1 ? u0 + 2 : 0));                   \
        static_assert(u1 - u0 <= 3, "at most three units of the cut per slot");                        \
      }                                                                                                \
      if constexpr (z >= kS && z < kS + 3) MAPX_X_WSTORE(CUR, kNCH - 1, (z >= kS && z < kS + 3 ? z - kS : 0)); \
      if constexpr (z == kS + 3) MAPX_X_WLOAD(CUR, kNCH - 1);                                          \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  }, std::make_integer_sequence<int, kNM>{})
#define MAPX_X_KSTEP_WEAVE(CUR, kt)                                                                             \
  do {                                                                                                 \
    const bf16_t* const As_cur = smem + (CUR) * kBuf;                                                  \
    const bf16_t* const Bs_cur = As_cur + OpA::LDS_ELEMS;                                              \
    bf16x8 fa[2][3][WMT], fb[2][3][WNT];          /* [k16 half][plane hi/mid/lo][tile] */              \
    unroll_seq([&](auto qc) __attribute__((always_inline)) {                                           \
      constexpr int q = decltype(qc)::value, op = frag_order(q, 0, WNT), pl = frag_order(q, 1, WNT), t = frag_order(q, 2, WNT); \
      if (op == 0) fa[0][pl][t] = OpA::frag1(As_cur, pl, abase, lane, 0, t);                           \
      else fb[0][pl][t] = OpB::frag1(Bs_cur, pl, bbase, lane, 0, t);                                   \
    }, std::make_integer_sequence<int, kFR>{});                                                        \
    uint32_t cH[2][4], cM[2][4], cL[2][4];        /* [chunk parity][pair] */                           \
    float cr0 = 0.f, cr1 = 0.f;                                                                        \
    const int wk_ = wk0 + kXBK * (((kt) + 3 < nk - 1 ? (kt) + 3 : nk - 1) - 1);   /* tile min(kt+3, nk-1) */ \
    const float* const wA = a.A + (int64_t)wk_ * (A_KC ? 1 : a.lda);                                   \
    const float* const wB = a.B + (int64_t)wk_ * (B_KC ? 1 : a.ldb);                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    static_assert(kPre == 6 && kU >= 12, "the units before the first MFMA are written out");           \
    if (!(kDbg & 2)) {                                                                                 \
      MAPX_X_CUT_UNIT(CUR, 0); MAPX_X_CUT_UNIT(CUR, 1); MAPX_X_CUT_UNIT(CUR, 2);                       \
      MAPX_X_CUT_UNIT(CUR, 3); MAPX_X_CUT_UNIT(CUR, 4); MAPX_X_CUT_UNIT(CUR, 5);                       \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    MAPX_X_WSLOTS(CUR);                                                                                             \
    __syncthreads();                                                                                   \
  } while (0)
#define MAPX_X_LOOPS(LATE)                                                                             \
  do {                                                                                                 \
    int kt = 0;                                                                                        \
    if (interior) {                                                                                    \
      for (; kt + 4 < nk; kt += 2) {                                                                   \
        MAPX_X_KSTEP(0, kt, true, false, LATE);                                                        \
        MAPX_X_KSTEP(1, kt + 1, true, false, LATE);                                                    \
      }                                                                                                \
    } else {                                                                                           \
      for (; kt + 4 < nk; kt += 2) {                                                                   \
        MAPX_X_KSTEP(0, kt, true, true, LATE);                                                         \
        MAPX_X_KSTEP(1, kt + 1, true, true, LATE);                                                     \
      }                                                                                                \
    }                                                                                                  \
    for (; kt < nk; kt += 2) {                                                                         \
      MAPX_X_KSTEP(0, kt, false, true, LATE);                                                          \
      if (kt + 1 < nk) MAPX_X_KSTEP(1, kt + 1, false, true, LATE);                                     \
    }                                                                                                  \
  } while (0)
  if constexpr (kWeave) {
    // per chunk: element offset from the K-step's (uniform) operand base, with the out-of-range row / column
    // of an edge tile clamped to 0 — the chunk's `ok` flag (set by the prologue's loads) zero-fills it at the cut
    int64_t goffA[OpA::NV], goffB[OpB::NV];
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      int tr, tc;
      OpA::coords(threadIdx.x + i * NT, tr, tc);
      const bool in = (A_KC ? m0 + tr : m0 + tc) < a.M;
      goffA[i] = A_KC ? (int64_t)(in ? m0 + tr : 0) * a.lda + tc : (int64_t)tr * a.lda + (in ? m0 + tc : 0);
    }
#pragma unroll
    for (int i = 0; i < OpB::NV; ++i) {
      int tr, tc;
      OpB::coords(threadIdx.x + i * NT, tr, tc);
      const bool in = (B_KC ? n0 + tr : n0 + tc) < a.N;
      goffB[i] = B_KC ? (int64_t)(in ? n0 + tr : 0) * a.ldb + tc : (int64_t)tr * a.ldb + (in ? n0 + tc : 0);
    }
    int soffA[OpA::NV], soffB[OpB::NV];          // LDS element offset of the chunk inside a buffer (plane 0)
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      int tr, tc;
      OpA::coords(threadIdx.x + i * NT, tr, tc);
      soffA[i] = tr * OpA::LD + tc;
    }
#pragma unroll
    for (int i = 0; i < OpB::NV; ++i) {
      int tr, tc;
      OpB::coords(threadIdx.x + i * NT, tr, tc);
      soffB[i] = OpA::LDS_ELEMS + tr * OpB::LD + tc;
    }
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      MAPX_X_KSTEP_WEAVE(0, kt);
      MAPX_X_KSTEP_WEAVE(1, kt + 1);
    }
    if (kt < nk) MAPX_X_KSTEP_WEAVE(0, kt);
  } else {
    MAPX_X_LOOPS(false);
  }
#undef MAPX_X_LOOPS
#undef MAPX_X_KSTEP_WEAVE
#undef MAPX_X_WSLOTS
#undef MAPX_X_CUT_UNIT
#undef MAPX_X_WSTORE
#undef MAPX_X_WLOAD
#undef MAPX_X_COMPUTE
#undef MAPX_X_STAGE
#undef MAPX_X_KSTEP
#undef MAPX_X_LOAD
#undef MAPX_X_STORE

  float* const tile = reinterpret_cast<float*>(smem_raw);
  constexpr int LDT = BN + 4;
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(abase + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2)) * LDT + bbase + 32 * j + l31] = acc[i][j][r] + cor[i][j][r];
  __syncthreads();
  epilogue_dispatch<BM, BN, NT>(a, C, tile, m0, n0);
}

template <int WR, int WC, int WMT, int WNT, bool A_KC, bool B_KC, bool VEC>
static hipError_t launch_one_x3(const GemmX3Args& a, int nsplit, hipStream_t stream, int batch = 1) {
  constexpr int BM = 32 * WMT * WR, BN = 32 * WNT * WC, NT = 64 * WR * WC;
  using OpA = OperandX3<BM, WMT, A_KC, VEC, NT>;
  using OpB = OperandX3<BN, WNT, B_KC, VEC, NT>;
  constexpr size_t lds = (size_t)2 * (OpA::LDS_ELEMS + OpB::LDS_ELEMS) * sizeof(bf16_t);
  static_assert(lds >= (size_t)BM * (BN + 4) * sizeof(float), "the epilogue's fp32 tile must fit");
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto* fn = &gemm_f32x3_kernel<WR, WC, WMT, WNT, A_KC, B_KC, VEC>;
  static hipError_t raised = lds > 65536
      ? hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipSuccess;
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(fn, dim3(a.tiles_m * a.tiles_n, nsplit, batch), dim3(NT), lds, stream, a);
  return hipSuccess;
}

// tile 2 (and 3): 128 x 128 by 8 waves (2 x 4, wave tile 64 x 32; 4 waves of 64 x 64 would need 256
// accumulator registers for the two sets and measured 15 % slower with one); tile 1: 128 x 64 by 4
// waves; tile 0: 64 x 64 by 4 waves
template <bool A_KC, bool B_KC>
static hipError_t launch_layout_x3(GemmX3Args& a, bool vec, int tile, int nsplit, hipStream_t stream, int batch = 1) {
#define MAPX_X3(WR_, WC_, WM, WN) (vec ? launch_one_x3<WR_, WC_, WM, WN, A_KC, B_KC, true>(a, nsplit, stream, batch) \
                                       : launch_one_x3<WR_, WC_, WM, WN, A_KC, B_KC, false>(a, nsplit, stream, batch))
  // (128x128 by 4 waves with 64x64 wave tiles — 2/3 of the LDS fragment reads per MFMA — measured 8-14 %
  // slower on every shape of the step: one wave per SIMD leaves the cut nothing to hide behind.)
  if (tile == 3) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
    return MAPX_X3(2, 2, 2, 2);
  }
  if (tile == 2) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
    return MAPX_X3(2, 4, 2, 1);
  }
  if (tile == 1) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 63) / 64;
    return MAPX_X3(2, 2, 2, 1);
  }
  a.tiles_m = (a.M + 63) / 64; a.tiles_n = (a.N + 63) / 64;
  return MAPX_X3(2, 2, 1, 1);
#undef MAPX_X3
}

__global__ void __launch_bounds__(256) splitk_reduce_x3_kernel(const float* __restrict__ slabs, int64_t slab_stride,
                                                               int nsplit, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slabs[s * slab_stride + i];
    out[i] = v;
  }
}
// 16-byte form (n % 4 == 0, aligned slabs and output): a quarter of the memory instructions; same order of adds
__global__ void __launch_bounds__(256) splitk_reduce_x3_v4_kernel(const float* __restrict__ slabs, int64_t slab_stride,
                                                                  int nsplit, int64_t n4, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(slabs)[i];
    v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f;          // (0 + x first, as the scalar form adds)
    for (int s = 1; s < nsplit; ++s) {
      const float4 x = reinterpret_cast<const float4*>(slabs + s * slab_stride)[i];
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

bool gemm_f32h2_try(GemmX3Args& g, int a_kc, int b_kc, bool vec, int tile, bool hinted, bool xcd_on, int nsplit,
                    int batch, hipStream_t stream, hipError_t* err);      // gemm_h2.hip

bool gemm_f32h2w_try(GemmX3Args& g, int a_kc, bool vec, const void* planes, int nsplit, int batch, hipStream_t stream,
                     hipError_t* err);     // gemm_h2w.hip

static bool xcd_slices_enabled() {
  static const bool on = [] { const char* e = getenv("MAPX_XCD_SLICES"); return !e || atoi(e) != 0; }();
  return on;
}

// Called by mapx_gemm_f32 (gemm.hip) when the split-bf16 path is selected.  Same contract.
int gemm_f32x3_launch(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, int epi, const float* bias, const float* aux1, int64_t ld1,
                      const float* aux2, int64_t ld2, float* out2, int64_t ldo2, int nsplit, int tile_hint, void* ws,
                      size_t ws_bytes, int* nsplit_deferred, hipStream_t stream, const GemmX3Extra* ex) {
  GemmX3Args g{};
  const int batch = ex && ex->batch > 1 ? ex->batch : 1;
  if (ex) {
    g.aux3 = ex->aux3; g.ld3 = ex->ld3; g.mask = ex->mask; g.ldm = ex->ldm; g.out3 = ex->out3; g.ldo3 = ex->ldo3;
    g.out4 = ex->out4; g.ldo4 = ex->ldo4; g.c0 = ex->c0; g.flags = ex->flags;
    g.amax_a = ex->amax_a; g.amax_b = ex->amax_b; g.amax_c = ex->amax_c; g.amax_c2 = ex->amax_c2;
    for (int z = 0; z < batch && batch > 1; ++z) {
      g.Az[z] = ex->Az[z]; g.Bz[z] = ex->Bz[z]; g.Cz[z] = ex->Cz[z];
      g.amax_az[z] = ex->amax_az[z]; g.amax_bz[z] = ex->amax_bz[z];
    }
  }
  g.epoch = amax_epoch_ptr();
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias;
  g.aux1 = aux1; g.ld1 = ld1; g.aux2 = aux2; g.ld2 = ld2; g.out2 = out2; g.ldo2 = ldo2;
  g.k_chunk = K > 0 ? K : kXBK; g.slab_stride = 0;
  if (nsplit > 1) {
    const size_t need = (size_t)batch * nsplit * M * N * sizeof(float);
    if (!ws || ws_bytes < need) {
      set_error("gemm_f32: split-K workspace %zu < %zu", ws_bytes, need);
      return MAPX_EWORKSPACE;
    }
    const int kc = (int)ceil_div(ceil_div(K, nsplit), 64) * 64;
    g.k_chunk = kc;
    g.batch_slabs = (int64_t)nsplit * M * N;      // (the requested count: the caller's slab layout)
    nsplit = (int)ceil_div(K, kc);
    g.C = static_cast<float*>(ws);
    for (int z = 0; z < batch && batch > 1; ++z) g.Cz[z] = static_cast<float*>(ws);
    g.ldc = N;
    g.slab_stride = (int64_t)M * N;
    g.amax_c = g.amax_c2 = nullptr;          // slabs hold partial sums: their maxima say nothing about C
  }

  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) &&
                   (g.k_chunk % 8 == 0) && (a_kc ? (K % 8 == 0) : (M % 8 == 0)) && (b_kc ? (K % 8 == 0) : (N % 8 == 0));
  auto blocks = [&](int bm, int bn) { return ceil_div(M, bm) * ceil_div(N, bn) * nsplit * batch; };
  const int64_t big = blocks(128, 128);
  // 128 x 128 layout of the NT / NN / TN products, three digits (3: 4 waves, hand-woven K-step; 2: 8 waves).
  // Alone the two are within 5 % of each other; inside the step's graph the 4-wave layout is worth 10 % of
  // the whole step (0.958 vs 1.069 ms): its one wave per SIMD and ~360 of 512 registers leave room for the
  // small kernels the other queue runs beside a GEMM, where the 8-wave layout (2 x 250 registers per SIMD)
  // owns the CU and every such kernel waits for a CU to drain.
  static const int big_tile = [] { const char* e = getenv("MAPX_X3_TILE"); return e ? atoi(e) : 333; }();
  const int cls = (a_kc && b_kc) ? 0 : a_kc ? 1 : 2;
  int tile = (big >= 160) ? (cls == 0 ? big_tile / 100 : cls == 1 ? big_tile / 10 % 10 : big_tile % 10) : 0;
  if (tile_hint >= 0 && (tile_hint & 255) <= 3) tile = tile_hint & 255;
  if (epi == MAPX_EPI_BWD_FUSED && tile == 0) tile = 1;       // one partial row per 128-row tile
  if (epi == MAPX_EPI_RELU_MASK_COLSUM) {
    auto al16 = [](const void* p, int64_t ld) { return p && (uintptr_t)p % 16 == 0 && ld % 4 == 0; };
    if (!(N % 4 == 0 && nsplit <= 1 && al16(C, ldc) && al16(aux1, ld1) && al16(out2, ldo2))) {
      set_error("gemm_f32: EPI_RELU_MASK_COLSUM needs N %% 4 == 0, no split-K, 16-byte aligned C / aux1 / out2");
      return MAPX_EINVAL;
    }
    if (tile == 0) tile = 1;          // one partial row per 128-row tile
  }
  // the woven K loop (4-wave layouts with vector loads) wants >= 2 K-steps in every slab
  if (tile != 2 && vec && K - (int64_t)g.k_chunk * (nsplit - 1) <= kXBK) tile = 2;
  if (nsplit > 1 && batch == 1 && xcd_slices_enabled()) {
    const int bm = tile == 0 ? 64 : 128, bn = (tile == 0 || tile == 1) ? 64 : 128;
    const int64_t nb1 = ceil_div(M, bm) * ceil_div(N, bn);
    g.xcd_slices = (8 % nsplit == 0 && nb1 % (8 / nsplit) == 0 && (nb1 * nsplit) % 8 == 0) ? 1 : 0;
  }
  hipError_t e = hipSuccess;
  // both operands with a magnitude record: the two-piece fp16 arithmetic (gemm_h2.hip), where it builds the case
  bool scaled = batch > 1 ? true : (g.amax_a && g.amax_b);
  for (int z = 0; z < batch && batch > 1; ++z) scaled = scaled && g.amax_az[z] && g.amax_bz[z];
  const bool hinted = tile_hint >= 0 && (tile_hint & 255) <= 3;
  // operand B's pieces already in HBM (a weight matrix: gemm_h2w.hip), else both operands cut in the kernel
  if (!hinted && ex && ex->b_planes && gemm_f32h2w_try(g, a_kc, vec, ex->b_planes, nsplit, batch, stream, &e)) {
  } else
  if (!(scaled && gemm_f32h2_try(g, a_kc, b_kc, vec, tile, hinted, xcd_slices_enabled(), nsplit, batch, stream, &e))) {
    if (a_kc && b_kc) e = launch_layout_x3<true, true>(g, vec, tile, nsplit, stream, batch);
    else if (a_kc) e = launch_layout_x3<true, false>(g, vec, tile, nsplit, stream, batch);
    else e = launch_layout_x3<false, false>(g, vec, tile, nsplit, stream, batch);
  }
  MAPX_HIP(e);
  if (batch > 1) {            // the caller sums the slabs of all problems with one launch (mapx_gemm_f32_batched)
    if (nsplit_deferred) *nsplit_deferred = nsplit > 1 ? nsplit : 0;
    return check_launch("gemm_f32 (3 x bf16, batched)");
  }
  if (nsplit_deferred) *nsplit_deferred = nsplit > 1 ? nsplit : 0;
  if (nsplit > 1 && !nsplit_deferred) {
    MAPX_REQUIRE(ldc == N, "gemm_f32: split-K output must be dense (ldc == N)");
    const int64_t n = (int64_t)M * N;
    if (n % 4 == 0 && g.slab_stride % 4 == 0 && (uintptr_t)ws % 16 == 0 && (uintptr_t)C % 16 == 0)
      hipLaunchKernelGGL(splitk_reduce_x3_v4_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, stream,
                         static_cast<const float*>(ws), g.slab_stride, nsplit, n / 4, C);
    else
      hipLaunchKernelGGL(splitk_reduce_x3_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                         static_cast<const float*>(ws), g.slab_stride, nsplit, n, C);
  }
  return check_launch("gemm_f32 (3 x bf16)");
}

// ------------------------------------------------------------------------------------------
// Grouped feat_encoder products (gemm.hip, "Grouped GEMMs": models.py:74-75) on the bf16 matrix cores.
// Same slot layout, same two products, same outputs as gemm_grouped_kernel:
//   FWD  h[slot, 0:32]   = final[rowmap[slot], :] . W[32 f : 32 f + 32, :]^T + bias    tile 128 slots x 32
//   DW   dW[32 f + p, n] = sum_{slot in group f} dh[slot, p] final[rowmap[slot], n]     tile 32 x 128 columns
// A tile has only four 32 x 32 MFMA tiles, so the block runs EIGHT waves and splits every K-step of 32
// between two sets of four: waves 0-3 take its first k16 half, waves 4-7 the second (each wave 6 MFMAs per
// K-step on its own accumulator pair), and the two partial tiles meet in LDS once, after the K loop.
// That puts two waves on every SIMD — one's cut of the next tile (VALU + LDS stores) beside the other's
// MFMAs — where four waves would leave each SIMD a single in-order instruction stream.
// Global loads: 640 chunks of 8 floats per K-step for 512 threads (the 128-row operand one chunk per
// thread, the 32-row operand by waves 0-1); gathered rows as in gemm_grouped_kernel (FWD resolves its
// rowmap entries once, DW fetches them one K-step ahead of the row loads that need them).
template <bool DW>
__global__ void __launch_bounds__(512) gemm_grouped_x3_kernel(GroupedArgs a) {
  constexpr int NT = 512;
  constexpr int BM = DW ? 32 : 128, BN = DW ? 128 : 32;
  using OpA = OperandX3<BM, 1, !DW, true, NT>;     // FWD: k-contiguous gathered rows;  DW: dh, [k = slot][p]
  using OpB = OperandX3<BN, 1, !DW, true, NT>;     // FWD: the field's 32 weight rows;  DW: gathered rows, [k = slot][n]
  static_assert(OpA::NV == 1 && OpB::NV == 1, "one chunk per thread per operand");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* const smem = reinterpret_cast<bf16_t*>(smem_raw);
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
    la[SET].template store<true>(smem + (buf) * kBuf);                                               \
    lb[SET].template store<true>(smem + (buf) * kBuf + OpA::LDS_ELEMS);                              \
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
    const bf16_t* const As_cur = smem + (CUR) * kBuf;                                                \
    const bf16_t* const Bs_cur = As_cur + OpA::LDS_ELEMS;                                            \
    bf16x8 ah[1], am[1], al[1], bh[1], bm[1], bl[1];                                                 \
    OpA::frags(As_cur, 0, abase, lane, half, ah);                                                    \
    OpB::frags(Bs_cur, 0, bbase, lane, half, bh);                                                    \
    OpA::frags(As_cur, 1, abase, lane, half, am);                                                    \
    OpB::frags(Bs_cur, 1, bbase, lane, half, bm);                                                    \
    OpA::frags(As_cur, 2, abase, lane, half, al);                                                    \
    OpB::frags(Bs_cur, 2, bbase, lane, half, bl);                                                    \
    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], bh[0], cor, 0, 0, 0);                       \
    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bl[0], cor, 0, 0, 0);                       \
    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], bm[0], cor, 0, 0, 0);                       \
    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], bh[0], cor, 0, 0, 0);                       \
    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bm[0], cor, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bh[0], acc, 0, 0, 0);                       \
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
    for (int r = 0; r < 16; ++r) part[(w4 * 16 + r) * 64 + lane] = acc[r] + cor[r];
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
      const float v = (acc[r] + cor[r]) + part[(w4 * 16 + r) * 64 + lane];
      Cb[(int64_t)m * a.ldc + n] = DW ? v * gs : v + bn;
    }
  }
}

template <bool DW>
static hipError_t launch_grouped_x3(const GroupedArgs& g, dim3 grid, hipStream_t stream) {
  constexpr int BM = DW ? 32 : 128, BN = DW ? 128 : 32;
  using OpA = OperandX3<BM, 1, !DW, true, 512>;
  using OpB = OperandX3<BN, 1, !DW, true, 512>;
  constexpr size_t lds = (size_t)2 * (OpA::LDS_ELEMS + OpB::LDS_ELEMS) * sizeof(bf16_t);
  static_assert(lds >= 4 * 16 * 64 * sizeof(float) && lds <= 160 * 1024, "LDS budget");
  auto* fn = &gemm_grouped_x3_kernel<DW>;
  static hipError_t raised =
      hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, stream, g);
  return hipSuccess;
}

hipError_t enc_grouped_fwd_x3_launch(const GroupedArgs& g, int cap_slots, hipStream_t stream) {
  return launch_grouped_x3<false>(g, dim3(cap_slots / 128), stream);
}
hipError_t enc_grouped_dw_x3_launch(const GroupedArgs& g, int F, hipStream_t stream) {
  return launch_grouped_x3<true>(g, dim3(((g.N + 127) / 128) * F), stream);
}

}  // namespace mapx
