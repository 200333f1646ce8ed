// fp32 MFMA GEMM family for the dense part of DCNv2: CrossNetV2 (code/layers.py:197-201),
// MLPBlock (layers.py:173-188), feat_encoder / pred_rfd / fc_out (models.py:74,119-124,304)
// and their backward products.  MFMA-bound (v_mfma_f32_32x32x2_f32: exact fp32, 64
// FLOP/clk/SIMD).
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(k,n) )
//
// Operand storage is described per operand, so that forward, dX and dW all run on the same
// kernel without a transposed copy of anything:
//   A_KC : A(m,k) = A[m*lda + k]  (k contiguous)      else A(m,k) = A[k*lda + m]
//   B_KC : B(k,n) = B[n*ldb + k]  (k contiguous)      else B(k,n) = B[k*ldb + n]
//   forward  Y = X W^T      : A_KC (X [B,in]),   B_KC (W [out,in])
//   dX = dY W               : A_KC (dY [B,out]), B_NC (W [out,in] read as [K=out, N=in])
//   dW = dY^T X             : A_MC (dY [B,out] read as [K=B, M=out]), B_NC (X [B,in])
//
// Tiling: 256 threads = 2x2 waves; a wave owns WMT x WNT MFMA tiles of 32x32; block tile
// (64*WMT) x (64*WNT), BK = 32.  Each operand keeps its GLOBAL orientation in LDS, so the
// global->LDS path is a straight 16-byte copy (global_load_dwordx4 -> ds_write_b128, whole
// 128-B lines per 8 lanes) for every layout:
//   k-contiguous operand  -> LDS [row][BK+4]: a lane fetches 4 consecutive k with ONE
//                            ds_read_b128 (row stride 36 floats = conflict-free);
//   k-strided operand     -> LDS [k][rows+4]: a lane fetches its k values with ds_read_b32
//                            (32 consecutive floats per half-wave = conflict-free).
// The MFMA contracts k in a permuted order that both layouts share: within a group of 8 k
// values, MFMA s (0..3) pairs k = 8q+s on lanes 0-31 with k = 8q+4+s on lanes 32-63.
// Global loads of tile t+1 are issued before the MFMAs of tile t and written to the other
// LDS buffer after them (one barrier per K-step); operand fragments of k-group q+1 are
// fetched before the MFMAs of group q.  Block ids are remapped so that each XCD (private
// 4 MiB L2) works on consecutive tiles of the same A row-panel.
#include "../../include/mapx_hip.h"
#include "common.h"
#include "gemm_grouped.h"

namespace mapx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int M, N, K;
  int epi;
  const float* bias;                 // [N]
  const float* aux1; int64_t ld1;    // CROSS: Xi   ADD / RELU_MASK: aux
  const float* aux2; int64_t ld2;    // CROSS: X0
  float* out2; int64_t ldo2;         // CROSS: u = W Xi + b (kept for backward)
  int k_chunk;                       // split-K: K range per blockIdx.y (multiple of BK)
  int64_t slab_stride;               // split-K: C offset per split
  int tiles_m, tiles_n;
  int dbg;                           // timing experiments only (tile_hint >> 8): 1 = no loads in the loop, 2 = no LDS stores
};

constexpr int kBK = 32;   // default K-step

// One operand's staging: global tile -> registers -> LDS, and LDS -> MFMA fragments.
// Branch-free loads: every lane always loads (from a clamped, valid address); what lies
// outside the matrix is zeroed when the registers are written to LDS, so no ALU op touches
// the loaded registers before the MFMAs of the current tile have been issued.
// VEC requires: leading dimension % 4 == 0, 16-B aligned base, and the contiguous extent
// (K for k-contiguous operands, M|N otherwise) % 4 == 0: a float4 is all-in or all-out.
template <int ROWS /*BM or BN*/, int T /*32-row MFMA tiles per wave*/, bool KC, bool VEC, int BK>
struct Operand {
  static constexpr int LD = KC ? BK + 4 : ROWS + 4;
  static constexpr int LDS_FLOATS = KC ? ROWS * LD : BK * LD;
  static constexpr int NV = ROWS * BK / 4 / 256;   // float4 per thread per tile
  float4 r[NV];
  bool ok[NV];
  // loop-invariant per-thread state (set once by init): element offset of float4 #i inside
  // K-step 0 with the out-of-matrix direction clamped, and whether that direction is valid.
  unsigned off[NV];
  bool inb[NV];

  __device__ static inline void coords(int f, int& row, int& col) {
    // (row, col) of float4 #f in the tile's storage order; col is the contiguous index
    if (KC) { row = f / (BK / 4); col = (f % (BK / 4)) << 2; }       // [ROWS][BK]
    else { constexpr int PER = ROWS / 4; row = f / PER; col = (f % PER) << 2; }   // [BK][ROWS]
  }

  __device__ inline void init(int64_t ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tr, tc;
      coords(threadIdx.x + i * 256, tr, tc);
      if (KC) {
        inb[i] = row0 + tr < nrows;
        off[i] = (unsigned)((inb[i] ? row0 + tr : 0) * ld + tc);
      } else {
        inb[i] = row0 + tc < nrows;
        off[i] = (unsigned)(tr * ld + (inb[i] ? row0 + tc : 0));
      }
    }
  }

  // Full K-step (k0 + BK <= kend): `gk` = operand base advanced to this K-step, wave-uniform
  // (g + k0 for k-contiguous storage, g + k0*ld otherwise): no per-lane address arithmetic.
  __device__ inline void load_full_one(const float* __restrict__ gk, int i) {
    ok[i] = inb[i];
    r[i] = *reinterpret_cast<const float4*>(gk + off[i]);
  }
  __device__ inline void load_full(const float* __restrict__ gk) {
#pragma unroll
    for (int i = 0; i < NV; ++i) load_full_one(gk, i);
  }

  // General K-step (K tail, unaligned operands): clamped addresses + per-element predicates.
  __device__ inline void load(const float* __restrict__ g, int64_t ld, int row0, int nrows,
                              int k0, int kend) {
#pragma unroll
    for (int i = 0; i < NV; ++i) load_one(g, ld, row0, nrows, k0, kend, i);
  }
  __device__ inline void load_one(const float* __restrict__ g, int64_t ld, int row0, int nrows,
                                  int k0, int kend, int i) {
    {
      int tr, tc;
      coords(threadIdx.x + i * 256, tr, tc);
      const int gr = (KC ? row0 : k0) + tr, gc = (KC ? k0 : row0) + tc;
      const int rlim = KC ? nrows : kend, clim = KC ? kend : nrows;
      const bool rok = gr < rlim;
      const float* p = g + (int64_t)(rok ? gr : 0) * ld;
      if (VEC) {
        ok[i] = rok && (gc < clim);
        r[i] = *reinterpret_cast<const float4*>(p + (ok[i] ? gc : 0));
      } else {
        ok[i] = true;
        const bool o0 = rok && gc + 0 < clim, o1 = rok && gc + 1 < clim;
        const bool o2 = rok && gc + 2 < clim, o3 = rok && gc + 3 < clim;
        const float x0 = p[o0 ? gc + 0 : 0], x1 = p[o1 ? gc + 1 : 0];
        const float x2 = p[o2 ? gc + 2 : 0], x3 = p[o3 ? gc + 3 : 0];
        r[i] = make_float4(o0 ? x0 : 0.f, o1 ? x1 : 0.f, o2 ? x2 : 0.f, o3 ? x3 : 0.f);
      }
    }
  }

  template <bool MASK>
  __device__ inline void store_one(float* __restrict__ s, int i) const {
    int tr, tc;
    coords(threadIdx.x + i * 256, tr, tc);
    const float4 v = (!MASK || ok[i]) ? r[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(s + tr * LD + tc) = v;
  }
  template <bool MASK>
  __device__ inline void store(float* __restrict__ s) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) store_one<MASK>(s, i);
  }

  // fragments of k-group q (8 k values) for this wave's T tiles: f[t][s], s = MFMA step
  __device__ static inline void frags(const float* __restrict__ s, int base, int l31, int kh,
                                      int q, float (&f)[T][4]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (KC) {
        const float4 v =
            *reinterpret_cast<const float4*>(s + (base + 32 * t + l31) * LD + 8 * q + 4 * kh);
        f[t][0] = v.x; f[t][1] = v.y; f[t][2] = v.z; f[t][3] = v.w;
      } else {
        const float* p = s + (8 * q + 4 * kh) * LD + base + 32 * t + l31;
        f[t][0] = p[0]; f[t][1] = p[LD]; f[t][2] = p[2 * LD]; f[t][3] = p[3 * LD];
      }
    }
  }
};

// C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// Element (i, j, r) of a wave's accumulators sits at row mrow + ROW(r), column n of the output,
// ROW(r) = (r&3) + 8*(r>>2): all addresses are `base + 32-bit element offset` with offsets that
// differ by compile-time multiples of the leading dimension (host-checked: every operand spans
// < 2^31 elements), so an element costs one add and one store instead of a 64-bit multiply.
// Control flow is kept out of the element loops — one lane-mask test per 32-column group, one
// wave-uniform test per 32-row band (ROWS_OK) — because every conditional block between a
// load and its use makes the compiler fall back to `s_waitcnt vmcnt(0)` in front of each store,
// which drains the previous store first: 64 serialised stores cost 4 us per launch.
// Per band the auxiliary operands of all elements are fetched first, the stores follow.
template <int EPI, int WMT, int WNT, bool ROWS_OK>
__device__ inline void epilogue_band(const GemmArgs& a, float* __restrict__ C, f32x16 (&acc)[WMT][WNT],
                                     const float (&bias_r)[WNT], int i, uint32_t mrow, int nbase, int l31) {
  constexpr bool kAux1 = EPI == MAPX_EPI_BIAS_CROSS || EPI == MAPX_EPI_ADD || EPI == MAPX_EPI_RELU_MASK;
  constexpr bool kAux2 = EPI == MAPX_EPI_BIAS_CROSS;
  const float* __restrict__ aux1 = a.aux1;
  const float* __restrict__ aux2 = a.aux2;
  float* __restrict__ out2 = a.out2;
  const uint32_t ldc = (uint32_t)a.ldc, ld1 = (uint32_t)a.ld1, ld2 = (uint32_t)a.ld2, ldo = (uint32_t)a.ldo2;
#define MAPX_ROW(r) ((uint32_t)(((r) & 3) + 8 * ((r) >> 2)))
  float x1[WNT][16], x2[WNT][16];
  if (kAux1) {
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      // clamped, always-valid addresses: the loads carry no predicate (values of dead lanes are unused)
      uint32_t n = (uint32_t)(nbase + 32 * j + l31);
      n = (int)n < a.N ? n : (uint32_t)(a.N - 1);
      // lanes of the upper half-wave start 4 rows lower: in a partial band their first row may
      // already be outside the matrix, so the load base is clamped as well (stores are guarded)
      const uint32_t mload = (ROWS_OK || (int)mrow < a.M) ? mrow : (uint32_t)(a.M - 1);
      const uint32_t o1 = mload * ld1 + n, o2 = mload * ld2 + n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t dr = ROWS_OK ? MAPX_ROW(r) : ((int)(mrow + MAPX_ROW(r)) < a.M ? MAPX_ROW(r) : 0u);   // mrow >= M: 0
        x1[j][r] = aux1[o1 + dr * ld1];
        if (kAux2) x2[j][r] = aux2[o2 + dr * ld2];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < WNT; ++j) {
    const uint32_t n = (uint32_t)(nbase + 32 * j + l31);
    if ((int)n < a.N) {
      const uint32_t oc = mrow * ldc + n, oo = mrow * ldo + n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r];
        if (EPI >= MAPX_EPI_BIAS && EPI <= MAPX_EPI_BIAS_CROSS) v += bias_r[j];
        if (EPI == MAPX_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        float u = v;
        if (EPI == MAPX_EPI_BIAS_CROSS) v = x1[j][r] + x2[j][r] * v;
        if (EPI == MAPX_EPI_ADD) v += x1[j][r];
        if (EPI == MAPX_EPI_RELU_MASK) v = x1[j][r] > 0.f ? v : 0.f;
        if (ROWS_OK || (int)(mrow + MAPX_ROW(r)) < a.M) {
          if (EPI == MAPX_EPI_BIAS_CROSS) out2[oo + MAPX_ROW(r) * ldo] = u;
          C[oc + MAPX_ROW(r) * ldc] = v;
        }
      }
    }
  }
#undef MAPX_ROW
}

template <int EPI, int WMT, int WNT>
__device__ inline void epilogue(const GemmArgs& a, float* __restrict__ C, f32x16 (&acc)[WMT][WNT],
                                const float (&bias_r)[WNT], int mbase, int nbase, int l31, int kh) {
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
    const uint32_t mrow = (uint32_t)(mbase + 32 * i + 4 * kh);
    if (mbase + 32 * i + 32 <= a.M)      // wave-uniform: the whole 32-row band is inside the matrix
      epilogue_band<EPI, WMT, WNT, true>(a, C, acc, bias_r, i, mrow, nbase, l31);
    else if (mbase + 32 * i < a.M)
      epilogue_band<EPI, WMT, WNT, false>(a, C, acc, bias_r, i, mrow, nbase, l31);
  }
}

template <int WMT, int WNT, bool A_KC, bool B_KC, bool VEC, int BK>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs a) {
  constexpr int BM = 64 * WMT, BN = 64 * WNT;
  using OpA = Operand<BM, WMT, A_KC, VEC, BK>;
  using OpB = Operand<BN, WNT, B_KC, VEC, BK>;
  __shared__ __attribute__((aligned(16))) float As[2][OpA::LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[2][OpB::LDS_FLOATS];

  // XCD-aware tile order: blocks b, b+8, ... share an XCD; give each XCD a contiguous run
  // of tiles so that the tiles of one A row-panel hit the same L2.
  const int nb = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x;
  const int per = nb / 8;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int kbeg = blockIdx.y * a.k_chunk;
  const int kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
  float* __restrict__ C = a.C + (int64_t)blockIdx.y * a.slab_stride;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int abase = wr * 32 * WMT, bbase = wc * 32 * WNT;

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // bias of this lane's output columns, fetched now so that its latency hides behind the K loop
  float bias_r[WNT];
#pragma unroll
  for (int j = 0; j < WNT; ++j) {
    const int n = n0 + bbase + 32 * j + l31;
    bias_r[j] = (a.epi >= MAPX_EPI_BIAS && a.epi <= MAPX_EPI_BIAS_CROSS && n < a.N) ? a.bias[n] : 0.f;
  }

  // Two register sets per operand: set (t & 1) carries tile t from its global load (issued during
  // K-step t-2) to its LDS store (during K-step t-1).  Loads and stores are cut into per-k-group
  // slices and interleaved with the MFMAs of the tile being computed, so that they issue in
  // the shadow of the 64-cycle MFMAs instead of before / after the MFMA block.
  OpA la[2];
  OpB lb[2];
  la[0].init(a.lda, m0, a.M);
  lb[0].init(a.ldb, n0, a.N);
#pragma unroll
  for (int i = 0; i < OpA::NV; ++i) { la[1].off[i] = la[0].off[i]; la[1].inb[i] = la[0].inb[i]; }
#pragma unroll
  for (int i = 0; i < OpB::NV; ++i) { lb[1].off[i] = lb[0].off[i]; lb[1].inb[i] = lb[0].inb[i]; }
  const int nk = (kend - kbeg + BK - 1) / BK;
  const int nk_full = VEC ? (kend - kbeg) / BK : 0;          // K-steps on the fast load path
  // interior tiles need no zero-fill of out-of-matrix rows (block-uniform -> scalar branch)
  const bool interior = VEC && (m0 + BM <= a.M) && (n0 + BN <= a.N);
  const int64_t astep = A_KC ? 1 : a.lda, bstep = B_KC ? 1 : a.ldb;
  constexpr int NQ = BK / 8;                                  // k-groups per K-step
  constexpr int SA = (OpA::NV + NQ - 1) / NQ, SB = (OpB::NV + NQ - 1) / NQ;   // slice sizes

  // slice `q` of the global loads of tile `t` into register set `set`.  FULL (literal true):
  // the caller guarantees a full K-step -> no branch, the slice stays in the MFMAs' basic block.
#define MAPX_LOAD_SLICE(set, t, q, FULL)                                                         \
  do {                                                                                           \
    const int k0_ = kbeg + (t) * BK;                                                             \
    const bool full_ = (FULL) || (t) < nk_full;                                                  \
    _Pragma("unroll") for (int i_ = (q) * SA; i_ < ((q) + 1) * SA && i_ < OpA::NV; ++i_) {       \
      if (full_) la[set].load_full_one(a.A + (int64_t)k0_ * astep, i_);                          \
      else la[set].load_one(a.A, a.lda, m0, a.M, k0_, kend, i_);                                 \
    }                                                                                            \
    _Pragma("unroll") for (int i_ = (q) * SB; i_ < ((q) + 1) * SB && i_ < OpB::NV; ++i_) {       \
      if (full_) lb[set].load_full_one(a.B + (int64_t)k0_ * bstep, i_);                          \
      else lb[set].load_one(a.B, a.ldb, n0, a.N, k0_, kend, i_);                                 \
    }                                                                                            \
  } while (0)
  // slice `q` of the LDS stores of the tile held in register set `set` into buffer `buf`
#define MAPX_STORE_SLICE(set, buf, q, MASK)                                                      \
  do {                                                                                           \
    _Pragma("unroll") for (int i_ = (q) * SA; i_ < ((q) + 1) * SA && i_ < OpA::NV; ++i_)         \
      la[set].template store_one<MASK>(As[buf], i_);                                             \
    _Pragma("unroll") for (int i_ = (q) * SB; i_ < ((q) + 1) * SB && i_ < OpB::NV; ++i_)         \
      lb[set].template store_one<MASK>(Bs[buf], i_);                                             \
  } while (0)

  if (nk > 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) MAPX_LOAD_SLICE(0, 0, q, false);
#pragma unroll
    for (int q = 0; q < NQ; ++q) MAPX_STORE_SLICE(0, 0, q, true);
    if (nk > 1) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) MAPX_LOAD_SLICE(1, 1, q, false);
    }
  }
  __syncthreads();

  // One K-step: MFMAs on LDS buffer SET; store tile kt+1 (register set SET^1) into the other
  // buffer; load tile kt+2 into set SET.  SET = kt & 1 is a literal (loop unrolled by 2).
  // STEADY (literal): tiles kt+1 and kt+2 exist and are full K-steps -> the body is one basic
  // block and the sched_group_barrier pattern puts one memory instruction behind every MFMA.
#define MAPX_KSTEP(SET, kt, STEADY, MASK)                                                        \
  do {                                                                                           \
    float af[2][WMT][4], bf[2][WNT][4];                                                          \
    OpA::frags(As[SET], abase, l31, kh, 0, af[0]);                                               \
    OpB::frags(Bs[SET], bbase, l31, kh, 0, bf[0]);                                               \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                             \
      const int c = q & 1;                                                                       \
      if (q + 1 < NQ) {                                                                          \
        OpA::frags(As[SET], abase, l31, kh, q + 1, af[c ^ 1]);                                   \
        OpB::frags(Bs[SET], bbase, l31, kh, q + 1, bf[c ^ 1]);                                   \
      }                                                                                          \
      if ((STEADY) || (kt) + 1 < nk) MAPX_STORE_SLICE((SET) ^ 1, (SET) ^ 1, q, MASK);            \
      if ((STEADY) || (kt) + 2 < nk) MAPX_LOAD_SLICE(SET, (kt) + 2, q, STEADY);                  \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                              \
        _Pragma("unroll") for (int i = 0; i < WMT; ++i)                                          \
          _Pragma("unroll") for (int j = 0; j < WNT; ++j)                                        \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i][s], bf[c][j][s], acc[i][j], 0, 0, 0); \
      if (STEADY) {                                                                              \
        _Pragma("unroll") for (int z = 0; z < 4 * WMT * WNT; ++z) {                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
          __builtin_amdgcn_sched_group_barrier(0x330, 1, 0);                                     \
        }                                                                                        \
      }                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
    __syncthreads();                                                                             \
  } while (0)

  int kt = 0;
  if (interior) {
    for (; kt + 3 < nk_full; kt += 2) {
      MAPX_KSTEP(0, kt, true, false);
      MAPX_KSTEP(1, kt + 1, true, false);
    }
  } else {
    for (; kt + 3 < nk_full; kt += 2) {
      MAPX_KSTEP(0, kt, true, true);
      MAPX_KSTEP(1, kt + 1, true, true);
    }
  }
  for (; kt < nk; kt += 2) {
    MAPX_KSTEP(0, kt, false, true);
    if (kt + 1 < nk) MAPX_KSTEP(1, kt + 1, false, true);
  }
#undef MAPX_KSTEP
#undef MAPX_LOAD_SLICE
#undef MAPX_STORE_SLICE

  const int mbase = m0 + abase, nbase = n0 + bbase;
  // The bias registers were loaded before the K loop; passing them through an ALU move here
  // retires that load for the compiler's wait-count bookkeeping (otherwise it re-waits, with
  // vmcnt(0), in front of every store of the epilogue).
#pragma unroll
  for (int j = 0; j < WNT; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(bias_r[j]) : "v"(bias_r[j]));
  switch (a.epi) {
    case MAPX_EPI_BIAS: epilogue<MAPX_EPI_BIAS>(a, C, acc, bias_r, mbase, nbase, l31, kh); break;
    case MAPX_EPI_BIAS_RELU: epilogue<MAPX_EPI_BIAS_RELU>(a, C, acc, bias_r, mbase, nbase, l31, kh); break;
    case MAPX_EPI_BIAS_CROSS: epilogue<MAPX_EPI_BIAS_CROSS>(a, C, acc, bias_r, mbase, nbase, l31, kh); break;
    case MAPX_EPI_ADD: epilogue<MAPX_EPI_ADD>(a, C, acc, bias_r, mbase, nbase, l31, kh); break;
    case MAPX_EPI_RELU_MASK: epilogue<MAPX_EPI_RELU_MASK>(a, C, acc, bias_r, mbase, nbase, l31, kh); break;
    default: epilogue<MAPX_EPI_NONE>(a, C, acc, bias_r, mbase, nbase, l31, kh); break;
  }
}

// ------------------------------------------------------------------------------------------
// Grouped GEMMs of the MFP head's feat_encoder (models.py:74-75).  The reference computes all
// F*P encoder outputs per row and then gathers the L masked fields' P-blocks: 74 % of the
// forward GEMM and of its weight-gradient GEMM is never read.  Here targets (b, l) are sorted
// by field (groups padded to 128 slots; rowmap[slot] = batch row or -1), and
//   FWD  h[slot, :]      = final[rowmap[slot], :] . W[f*P:(f+1)*P, :]^T + bias[f*P:(f+1)*P]
//        one block per 128-slot tile (single field f = tile_group[tile]); 4x1 waves, tile 128x32;
//        A rows gathered through rowmap, B = the field's 32 weight rows.
//   DW   dW[f*P + p, n]  = sum_{slot in group f} dh[slot, p] * final[rowmap[slot], n]
//        one block per (field, 128-column tile); 1x4 waves, tile 32x128; K runs over the group's
//        slots, B rows gathered through rowmap.  Every weight row is written (zeros for fields
//        nobody masked), so no split-K and no zero-fill pass.
// P = 32 only (the reference default); other proj sizes use the dense path.
template <bool DW>
__global__ void __launch_bounds__(256) gemm_grouped_kernel(GroupedArgs a) {
  constexpr int BK = 32;
  constexpr int BM = DW ? 32 : 128, BN = DW ? 128 : 32;
  using OpA = Operand<BM, 1, !DW, true, BK>;     // FWD: k-contiguous rows; DW: [k][m] storage
  using OpB = Operand<BN, 1, !DW, true, BK>;     // FWD: k-contiguous rows; DW: [k][n] storage
  __shared__ __attribute__((aligned(16))) float As[2][OpA::LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[2][OpB::LDS_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int abase = DW ? 0 : wave * 32, bbase = DW ? wave * 32 : 0;

  int f, kbeg, kend, n0 = 0, slot0 = 0;
  if (DW) {
    f = blockIdx.y;
    kbeg = a.group_start[f];
    kend = a.group_start[f + 1];
    n0 = blockIdx.x * BN;
  } else {
    if (a.zero_out) {                // saves the separate fill launch of the slot-ordered dL/dh buffer
      float4* z = reinterpret_cast<float4*>(a.zero_out + (int64_t)blockIdx.x * BM * 32);
      for (int i = threadIdx.x; i < BM * 32 / 4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f = a.tile_group[blockIdx.x];
    if (f < 0) return;               // capacity tile beyond the used slots
    slot0 = blockIdx.x * BM;
    kbeg = 0;
    kend = a.K;
  }
  const float* __restrict__ Ab = DW ? a.A : a.A;
  const float* __restrict__ Bb = DW ? a.B : a.B + (int64_t)f * 32 * a.ldb;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // two register sets: tile t+2 is loaded while tile t is computed and tile t+1 is stored, so a
  // gathered load (rowmap, then the row it names) has two K-steps to land (one wave per SIMD)
  OpA la[2];
  OpB lb[2];
  // FWD: the gathered A rows do not change along K -> resolve rowmap once
  int64_t arow[OpA::NV];
  bool arow_ok[OpA::NV];
  if (!DW) {
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      int tr, tc;
      OpA::coords(threadIdx.x + i * 256, tr, tc);
      const int row = a.rowmap[slot0 + tr];
      arow_ok[i] = row >= 0;
      arow[i] = (int64_t)(row >= 0 ? row : 0) * a.lda + tc;
    }
  }
  // DW: the gathered B rows change every K-step -> their rowmap entries are fetched one K-step
  // before the row loads that need them, so no load waits on another load
  int brow[2][OpB::NV];
#define MAPX_G_ROWS(set, k0_)                                                                     \
  do {                                                                                            \
    if (DW) {                                                                                     \
      _Pragma("unroll") for (int i = 0; i < OpB::NV; ++i) {                                       \
        int tr, tc;                                                                               \
        OpB::coords(threadIdx.x + i * 256, tr, tc);                                               \
        brow[set][i] = ((k0_) + tr) < kend ? a.rowmap[(k0_) + tr] : -1;                           \
      }                                                                                           \
    }                                                                                             \
  } while (0)
#define MAPX_G_LOAD(set, k0_)                                                                     \
  do {                                                                                            \
    const int k0 = (k0_);                                                                         \
    _Pragma("unroll") for (int i = 0; i < OpA::NV; ++i) {                                         \
      int tr, tc;                                                                                 \
      OpA::coords(threadIdx.x + i * 256, tr, tc);                                                 \
      if (DW) {                                                                                   \
        la[set].ok[i] = true;                                                                     \
        la[set].r[i] = *reinterpret_cast<const float4*>(Ab + (int64_t)(k0 + tr) * a.lda + tc);    \
      } else {                                                                                    \
        la[set].ok[i] = arow_ok[i] && (k0 + tc) < kend;                                           \
        la[set].r[i] = *reinterpret_cast<const float4*>(Ab + (la[set].ok[i] ? arow[i] + k0 : 0));  \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < OpB::NV; ++i) {                                         \
      int tr, tc;                                                                                 \
      OpB::coords(threadIdx.x + i * 256, tr, tc);                                                 \
      if (DW) {                                                                                   \
        const int row = brow[set][i];           /* resolved one K-step earlier */                 \
        lb[set].ok[i] = row >= 0 && (n0 + tc) < a.N;                                              \
        lb[set].r[i] = *reinterpret_cast<const float4*>(                                          \
            Bb + (lb[set].ok[i] ? (int64_t)row * a.ldb + n0 + tc : 0));                           \
      } else {                                                                                    \
        lb[set].ok[i] = (k0 + tc) < kend;                                                         \
        lb[set].r[i] = *reinterpret_cast<const float4*>(                                          \
            Bb + (int64_t)tr * a.ldb + (lb[set].ok[i] ? k0 + tc : 0));                            \
      }                                                                                           \
    }                                                                                             \
  } while (0)
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    MAPX_G_ROWS(0, kbeg);
    MAPX_G_LOAD(0, kbeg);
    la[0].template store<true>(As[0]);
    lb[0].template store<true>(Bs[0]);
    if (nk > 1) {
      MAPX_G_ROWS(1, kbeg + BK);
      MAPX_G_LOAD(1, kbeg + BK);
    }
    if (nk > 2) MAPX_G_ROWS(0, kbeg + 2 * BK);
  }
  __syncthreads();
#define MAPX_G_KSTEP(SET, kt)                                                                     \
  do {                                                                                            \
    if ((kt) + 2 < nk) MAPX_G_LOAD(SET, kbeg + ((kt) + 2) * BK);                                  \
    if ((kt) + 3 < nk) MAPX_G_ROWS((SET) ^ 1, kbeg + ((kt) + 3) * BK);                            \
    float af[2][1][4], bf[2][1][4];                                                               \
    OpA::frags(As[SET], abase, l31, kh, 0, af[0]);                                                \
    OpB::frags(Bs[SET], bbase, l31, kh, 0, bf[0]);                                                \
    _Pragma("unroll") for (int q = 0; q < BK / 8; ++q) {                                          \
      const int c = q & 1;                                                                        \
      if (q + 1 < BK / 8) {                                                                       \
        OpA::frags(As[SET], abase, l31, kh, q + 1, af[c ^ 1]);                                    \
        OpB::frags(Bs[SET], bbase, l31, kh, q + 1, bf[c ^ 1]);                                    \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                               \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0][s], bf[c][0][s], acc, 0, 0, 0);       \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    if ((kt) + 1 < nk) {                                                                          \
      la[(SET) ^ 1].template store<true>(As[(SET) ^ 1]);                                          \
      lb[(SET) ^ 1].template store<true>(Bs[(SET) ^ 1]);                                          \
    }                                                                                             \
    __syncthreads();                                                                              \
  } while (0)
  for (int kt = 0; kt < nk; kt += 2) {
    MAPX_G_KSTEP(0, kt);
    if (kt + 1 < nk) MAPX_G_KSTEP(1, kt + 1);
  }
#undef MAPX_G_KSTEP
#undef MAPX_G_LOAD
#undef MAPX_G_ROWS
  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int n = (DW ? n0 : 0) + bbase + l31;
  if (DW ? (n < a.N) : true) {
    const float bn = DW ? 0.f : a.bias[f * 32 + n];
    const float gs = (DW && a.gscale) ? *a.gscale : 1.f;
    float* __restrict__ Cb = DW ? a.C + (int64_t)f * 32 * a.ldc : a.C + (int64_t)slot0 * a.ldc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = abase + (r & 3) + 8 * (r >> 2) + 4 * kh;
      Cb[(int64_t)m * a.ldc + n] = DW ? acc[r] * gs : acc[r] + bn;
    }
  }
}

// Padded by-field slot layout of the T = B*L targets, straight from masked_index (one block,
// one launch; the targets' field ids are a key space of F <= 64 values, so a counting sort in
// LDS replaces a general radix sort + run detection + layout = 9 launches).
//   slots are ordered by field, inside a field by target index t (stable => the summation order
//   of the grouped dW GEMM is fixed); every present field's group is padded to a multiple of 128.
//   hpos[t] = slot of target t;  rowmap[slot] = batch row t / L, or -1 for padding;
//   tile_group[slot / 128] = field of that 128-slot tile, or -1;  group_start[f], f = 0..F.
constexpr int kLayoutThreads = 1024;
constexpr int kLayoutMaxT = 96 * 1024;   // field ids of all targets cached in LDS as bytes
__global__ void __launch_bounds__(kLayoutThreads) enc_group_layout_kernel(
    const int64_t* __restrict__ masked_index, int T, int L, int F, int cap_slots, int32_t* __restrict__ rowmap,
    int32_t* __restrict__ hpos, int32_t* __restrict__ tile_group, int32_t* __restrict__ group_start) {
  constexpr int NW = kLayoutThreads / 64;
  __shared__ int wave_cnt[NW][64];     // targets of field f in wave w's chunk, then the wave's running offset
  __shared__ int gstart[64 + 1];       // padded start of field f's group
  extern __shared__ uint8_t fld[];     // [T]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rounds = (T + kLayoutThreads - 1) / kLayoutThreads;      // same for every wave
  const int wbase = wave * rounds * 64;                              // wave w owns targets [wbase, wbase + rounds*64)
  for (int i = threadIdx.x; i < NW * 64; i += kLayoutThreads) (&wave_cnt[0][0])[i] = 0;
  for (int t = threadIdx.x; t < T; t += kLayoutThreads) {            // coalesced, all loads independent
    int f = (int)masked_index[t];
    fld[t] = (uint8_t)(f < 0 ? 0 : (f >= F ? F - 1 : f));
  }
  for (int s = threadIdx.x; s < cap_slots; s += kLayoutThreads) rowmap[s] = -1;
  __syncthreads();
  for (int r = 0; r < rounds; ++r) {
    const int t = wbase + r * 64 + lane;
    if (t < T) atomicAdd(&wave_cnt[wave][fld[t]], 1);
  }
  __syncthreads();
  if (threadIdx.x < 64) {              // thread f: exclusive prefix over the waves, field total
    const int f = threadIdx.x;
    int run = 0;
    for (int w = 0; w < NW; ++w) {
      const int c = wave_cnt[w][f];
      wave_cnt[w][f] = run;
      run += c;
    }
    gstart[f] = f < F ? (run + 127) / 128 * 128 : 0;                // padded length for now
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int f = 0; f < F; ++f) {
      const int len = gstart[f];
      gstart[f] = run;
      group_start[f] = run;
      run += len;
    }
    gstart[F] = run;
    group_start[F] = run;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < cap_slots / 128; k += kLayoutThreads) {
    int g = -1;
    for (int f = 0; f < F; ++f)
      if (gstart[f] <= k * 128 && k * 128 < gstart[f + 1]) g = f;
    tile_group[k] = g;
  }
  // placement: a wave walks its chunk in target order; lanes of one round that share a field
  // rank themselves by lane id, then the field's running offset advances by their number
  for (int r = 0; r < rounds; ++r) {
    const int t = wbase + r * 64 + lane;
    const bool live = t < T;
    const int f = live ? fld[t] : 0;
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const unsigned long long m = __ballot(live && ((f >> b) & 1));
      peers &= ((f >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    // LDS operations of one wave execute in order: the next round's read sees this round's
    // update without a hardware wait; only the compiler has to keep the order
    int base = 0;
    if (live) base = wave_cnt[wave][f];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live) {
      const int s = gstart[f] + base + rank;
      hpos[t] = s;
      rowmap[s] = t / L;
      if (rank == 0) wave_cnt[wave][f] = base + __popcll(peers);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// The same layout with one workgroup PER FIELD (grid = F).  The single-workgroup kernel above takes
// 30-50 us at the head of the cross tower's stream, and since the NCE sampling joined it there that
// chain is the longer one of the forward pass.  Every workgroup histograms all T field ids (LDS
// atomics; the ids are cached as bytes in LDS), so each knows the padded start of its own field's
// group without talking to the others; then its 4 waves place the targets of field f in target
// order: wave w owns a contiguous quarter of the targets, counts its matches first (ballots), and
// after one barrier walks the quarter again handing out consecutive slots.  Stable, like the
// kernel above, so the dW summation order is unchanged.
constexpr int kLayoutMwThreads = 1024;
constexpr int kLayoutBatch = 8;          // field ids a thread fetches before it processes any of them
__global__ void __launch_bounds__(kLayoutMwThreads) enc_group_layout_mw_kernel(
    const int64_t* __restrict__ masked_index, int T, int L, int F, int cap_slots, int32_t* __restrict__ rowmap,
    int32_t* __restrict__ hpos, int32_t* __restrict__ tile_group, int32_t* __restrict__ group_start) {
  constexpr int NW = kLayoutMwThreads / 64;
  __shared__ int hist[64];
  __shared__ int wave_match[NW];
  __shared__ int gs[64 + 1];
  extern __shared__ uint8_t fld[];     // [T]
  const int f = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  // pass 1: histogram of every field + this wave's number of targets of field f.  16 waves share the
  // T targets (the 4-wave version walked 96 rounds per wave, each one a dependent global load: 52 us
  // alone, 100+ us beside the towers' GEMMs — the longest link of the forward pass's side chain); the
  // ids of kLayoutBatch rounds are fetched together, so a wave pays the memory latency T / 8192 times.
  const int per = ((T + kLayoutMwThreads - 1) / kLayoutMwThreads) * 64;      // targets per wave, a multiple of 64
  const int w0 = wave * per, w1 = min(w0 + per, T);
  int mine = 0;
  for (int t0 = w0 + lane; t0 < w0 + per; t0 += 64 * kLayoutBatch) {
    int g[kLayoutBatch];
#pragma unroll
    for (int u = 0; u < kLayoutBatch; ++u) {
      const int t = t0 + 64 * u;
      g[u] = (t < w1 && t < w0 + per) ? (int)masked_index[t] : -1;
    }
#pragma unroll
    for (int u = 0; u < kLayoutBatch; ++u) {
      const int t = t0 + 64 * u;
      const bool live = t < w1 && t < w0 + per;
      int gg = g[u];
      gg = gg < 0 ? 0 : (gg >= F ? F - 1 : gg);
      if (live) {
        fld[t] = (uint8_t)gg;
        atomicAdd(&hist[gg], 1);
      }
      mine += __popcll(__ballot(live && gg == f));
    }
  }
  if (lane == 0) wave_match[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int g = 0; g < F; ++g) {
      gs[g] = run;
      run += (hist[g] + 127) / 128 * 128;
    }
    gs[F] = run;
  }
  __syncthreads();
  const int start = gs[f], cnt = hist[f], padded = gs[f + 1];
  if (threadIdx.x == 0) {
    group_start[f] = start;
    if (f == F - 1) group_start[F] = gs[F];
  }
  // padding slots of this group, the capacity behind the last group, and the tiles' fields
  for (int sl = start + cnt + threadIdx.x; sl < padded; sl += kLayoutMwThreads) rowmap[sl] = -1;
  for (int k = start / 128 + threadIdx.x; k < padded / 128; k += kLayoutMwThreads) tile_group[k] = f;
  if (f == F - 1) {
    for (int sl = gs[F] + threadIdx.x; sl < cap_slots; sl += kLayoutMwThreads) rowmap[sl] = -1;
    for (int k = gs[F] / 128 + threadIdx.x; k < cap_slots / 128; k += kLayoutMwThreads) tile_group[k] = -1;
  }
  // pass 2: placement in target order (field ids now come from LDS)
  int base = start;
  for (int w = 0; w < wave; ++w) base += wave_match[w];
  for (int t = w0 + lane; t < w0 + per; t += 64) {
    const bool hit = t < w1 && fld[t] == f;
    const unsigned long long peers = __ballot(hit);
    if (hit) {
      const int sl = base + __popcll(peers & ((1ull << lane) - 1ull));
      hpos[t] = sl;
      rowmap[sl] = t / L;
    }
    base += __popcll(peers);
  }
}

// out[i] = sum_s slabs[s][i] in slab order (deterministic split-K combine)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ slabs,
                                                            int64_t slab_stride, int nsplit,
                                                            int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slabs[s * slab_stride + i];
    out[i] = v;
  }
}

// Deferred slab sums: dst[i] = sum_s src[s*stride + i] for up to kMaxSumTasks independent
// tasks in ONE launch (split-K slabs of the weight-gradient GEMMs and the row-chunk partials
// of the bias-gradient column sums of a whole backward pass; 14 tiny launches -> 1).
constexpr int kMaxSumTasks = 32;
struct SumTasks {
  mapx_sum_task t[kMaxSumTasks];
};
__global__ void __launch_bounds__(256) sum_tasks_kernel(SumTasks tasks) {
  const mapx_sum_task tk = tasks.t[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tk.n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    int s = 0;
    // The kernel is a latency chain (a column has one thread, a task a few blocks): 32 loads in flight per
    // round trip — 128 row-chunk partials of a bias gradient are 4 round trips, not 16 — added in slab order.
    for (; s + 32 <= tk.nsplit; s += 32) {
      float x[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) x[u] = tk.src[(s + u) * tk.stride + i];
#pragma unroll
      for (int u = 0; u < 32; ++u) v += x[u];
    }
    for (; s + 8 <= tk.nsplit; s += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = tk.src[(s + u) * tk.stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; s < tk.nsplit; ++s) v += tk.src[s * tk.stride + i];
    tk.dst[i] = v;
  }
}

// column sums of X [M,N] (bias gradients): stage 1 = 32 row chunks -> partial[32][N]
constexpr int kColChunks = 128;
__global__ void __launch_bounds__(256) colsum_stage1_kernel(const float* __restrict__ x, int64_t ld,
                                                            int M, int N, float* __restrict__ part) {
  // block = 64 columns x 4 row lanes
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int rows_per = (M + kColChunks - 1) / kColChunks;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
  float v = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) v += x[(int64_t)r * ld + c];
  __shared__ float s[4][64];
  s[rl][threadIdx.x & 63] = v;
  __syncthreads();
  if (rl == 0 && c < N)
    part[(int64_t)blockIdx.y * N + c] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] +
                                        s[3][threadIdx.x];
}
__global__ void __launch_bounds__(256) colsum_stage2_kernel(const float* __restrict__ part, int N,
                                                            float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float v = 0.f;
  for (int i = 0; i < kColChunks; ++i) v += part[(int64_t)i * N + c];
  out[c] = v;
}

// CrossNetV2 backward, elementwise part of one layer (layers.py:200 differentiated):
//   t = g * x0 (feeds the dXi / dW GEMMs and db),  dx0 (+)= g * u
__global__ void __launch_bounds__(256) cross_bwd_pre_kernel(const float* __restrict__ g,
                                                            const float* __restrict__ x0,
                                                            const float* __restrict__ u, int64_t n4,
                                                            float* __restrict__ t,
                                                            float* __restrict__ dx0, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    const float4 xv = reinterpret_cast<const float4*>(x0)[i];
    const float4 uv = reinterpret_cast<const float4*>(u)[i];
    reinterpret_cast<float4*>(t)[i] = make_float4(gv.x * xv.x, gv.y * xv.y, gv.z * xv.z, gv.w * xv.w);
    float4 d = make_float4(gv.x * uv.x, gv.y * uv.y, gv.z * uv.z, gv.w * uv.w);
    if (accumulate) {
      const float4 o = reinterpret_cast<const float4*>(dx0)[i];
      d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w;
    }
    reinterpret_cast<float4*>(dx0)[i] = d;
  }
}

// Elementwise backward steps fused with the bias-gradient column sum (one pass over the
// data instead of elementwise kernel + colsum stage 1):
//   OP 0 (ReLU layer):   dz = y > 0 ? dy : 0                      db = colsum(dz)
//   OP 1 (cross layer):  t = g * x0 ; dx0 (+)= g * u (+ g)        db = colsum(t)
//                        accumulate bit 0: add to the dx0 already there; bit 1: also add g
// Block = CL column lanes (x float4) x 256 / CL row lanes over one of kColChunks row chunks; stage 2 adds the
// chunks.  CL = 64 or 32, whichever wastes fewer lanes on the last column block (N = 368: 92 float4 columns
// are 2 blocks of 64 with 28 % of the lanes idle, or 3 blocks of 32 with 4 %).
template <int OP, int CL>
__global__ void __launch_bounds__(256) ew_colsum_kernel(const float* __restrict__ a, int64_t lda,
                                                        const float* __restrict__ b, int64_t ldb,
                                                        const float* __restrict__ c, int M, int N,
                                                        float* __restrict__ o1, float* __restrict__ o2,
                                                        int accumulate, float* __restrict__ part) {
  // CL lanes x float4 columns per block row; RL row lanes; N % 4 == 0, lda % 4 == 0 (host-checked)
  constexpr int RL = 256 / CL;
  const int cl = threadIdx.x % CL;
  const int col = (blockIdx.x * CL + cl) * 4;
  const int rl = threadIdx.x / CL;
  const int rows_per = (M + kColChunks - 1) / kColChunks;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < N) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += RL) {
      const int64_t i = ((int64_t)r * N + col) >> 2;
      const float4 av = reinterpret_cast<const float4*>(a)[((int64_t)r * lda + col) >> 2];   // a may be a column slice
      const float4 bv = reinterpret_cast<const float4*>(b)[((int64_t)r * ldb + col) >> 2];
      if (OP == 0) {
        const float4 dz = make_float4(bv.x > 0.f ? av.x : 0.f, bv.y > 0.f ? av.y : 0.f,
                                      bv.z > 0.f ? av.z : 0.f, bv.w > 0.f ? av.w : 0.f);
        reinterpret_cast<float4*>(o1)[i] = dz;
        v.x += dz.x; v.y += dz.y; v.z += dz.z; v.w += dz.w;
      } else {
        const float4 cv = reinterpret_cast<const float4*>(c)[i];
        const float4 t = make_float4(av.x * bv.x, av.y * bv.y, av.z * bv.z, av.w * bv.w);
        float4 d = make_float4(av.x * cv.x, av.y * cv.y, av.z * cv.z, av.w * cv.w);
        if (accumulate & 1) {
          const float4 o = reinterpret_cast<const float4*>(o2)[i];
          d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w;
        }
        if (accumulate & 2) {   // first cross layer: Xi IS X0, its gradient g joins the X0 total
          d.x += av.x; d.y += av.y; d.z += av.z; d.w += av.w;
        }
        reinterpret_cast<float4*>(o1)[i] = t;
        reinterpret_cast<float4*>(o2)[i] = d;
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    }
  }
  __shared__ float4 s[RL][CL];
  s[rl][cl] = v;
  __syncthreads();
  if (rl == 0 && col < N) {
    float4 t = s[0][cl];
#pragma unroll
    for (int k = 1; k < RL; ++k) {       // fixed order: bit-reproducible
      const float4 q = s[k][cl];
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * N + col) = t;
  }
}

// 32 instead of 64 column lanes per block when that leaves fewer lanes of the last column block idle
static inline bool ew_narrow_lanes(int N) {
  const int c4 = (N + 3) / 4;
  return ((c4 + 31) / 32) * 32 < ((c4 + 63) / 64) * 64;
}

// ---------------------------------------------------------------------------------------------
// Weight-gradient GEMM with a deep K-step:  C[m,n] = sum_k A[k*lda + m] * B[k*ldb + n]
// (dW = dY^T X: both operands k-strided).  Output tiles of 64x64 put exactly one block on each
// of the 256 CUs for the step's 1000x1000 gradients; at BK = 32 such a block spends a third of
// its time in the per-K-step fixed costs (barrier, LDS round trip, staging latency) because a
// K-step is only 16 MFMAs per wave.  Here a K-step is 128 deep — 64 MFMAs per wave between
// barriers (hipBLASLt picks the same shape for this problem: MT64x64x128) — and the operands go
// from global memory straight into LDS (global_load_lds_dwordx4: no staging registers, no
// ds_write, nothing for the MFMA stream to wait on until the end of the step).
//   * a wave-instruction of the LDS-DMA writes 64 lanes x 16 B = 1 KB contiguous = 4 rows k of a
//     tile (row = 64 consecutive floats); thread t fetches, for i in 0..7, row (t >> 4) + 16 i;
//   * LDS image: rows 0..63 dense from the buffer's base, rows 64..127 dense behind them.  An
//     MFMA operand register takes row j on lanes 0-31 and row 64+j on lanes 32-63 (any pairing
//     of k values is a valid contraction order as long as A and B use the same one); every
//     ds_read is 32 consecutive floats per half-wave: conflict-free without padding.
//     2 operands x 2 buffers x 32 KB = 128 KB of the CU's 160.
// Needs K-chunks that are multiples of 128 (no zero fill on the DMA path) and M, N multiples of 4.
// Measured (1000x1000x4096): 81-87 us against 96-100 for the BK = 32 kernel; 2.1-2.4 us per K-step
// of 64 MFMAs (1.95 at the MFMA rate): what is left is the L2 -> CU fill rate a 64x64 tile needs
// (512 B per k per CU, 6.2 TB/s chip-wide at this speed).
constexpr int kDeepBK = 128;
constexpr int kDeepTile = kDeepBK * 64;                     // floats per operand buffer
constexpr size_t kDeepLds = (size_t)4 * kDeepTile * sizeof(float);

__device__ inline void deep_dma16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ void __launch_bounds__(256, 1) gemm_tn_deep_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // buffer b: A at 2b tiles, B one tile behind
  const int nb = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x;
  const int per = nb / 8;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order (see gemm_f32_kernel)
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int abase = (wave >> 1) * 32, bbase = (wave & 1) * 32;

  const int kbeg = blockIdx.y * a.k_chunk;
  const int kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
  const int nk = (kend - kbeg) / kDeepBK;                  // host-checked: whole K-steps only

  // DMA map: this lane's source for instruction i of a K-step is row (wave*4 + lane/16) + 16 i,
  // floats [4 (lane & 15), +4); its wave's destination is the 1 KB that holds those 4 rows
  const int scol = (lane & 15) * 4;
  const int ca = min(m0 + scol, a.M - 4), cb = min(n0 + scol, a.N - 4);     // clamped: never stored from
  const float* __restrict__ ga = a.A + (int64_t)(kbeg + wave * 4 + (lane >> 4)) * a.lda + ca;
  const float* __restrict__ gb = a.B + (int64_t)(kbeg + wave * 4 + (lane >> 4)) * a.ldb + cb;
  const int64_t sa = 16 * a.lda, sb = 16 * a.ldb;          // row stride between two instructions
  const int ldst = wave * 4 * 64;                          // floats: first of this wave's 4 rows
#define MAPX_DEEP_DMA(buf, i)                                                                  \
  do {                                                                                         \
    deep_dma16(ga + (i) * sa, smem + (buf) * 2 * kDeepTile + ldst + (i) * 16 * 64);            \
    deep_dma16(gb + (i) * sb, smem + ((buf) * 2 + 1) * kDeepTile + ldst + (i) * 16 * 64);      \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) MAPX_DEEP_DMA(0, i);
    ga += (int64_t)kDeepBK * a.lda;
    gb += (int64_t)kDeepBK * a.ldb;
  }
  __syncthreads();                                         // (drains the DMA: vmcnt(0) + barrier)
  const int fa = kh * 64 * 64 + abase + l31, fb = kh * 64 * 64 + bbase + l31;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // the last step has nothing to prefetch: it re-fetches its own rows into the idle buffer, so
    // that the group bodies stay branch-free (a branch between an LDS read and its MFMA makes the
    // compiler wait for ALL outstanding LDS traffic at the join)
    if (kt + 1 == nk) {
      ga -= (int64_t)kDeepBK * a.lda;
      gb -= (int64_t)kDeepBK * a.ldb;
    }
    const float* __restrict__ pa = smem + cur * 2 * kDeepTile + fa;
    const float* __restrict__ pb = smem + (cur * 2 + 1) * kDeepTile + fb;
    // 64 k-pairs in 8 groups of 8 MFMAs, fragments two groups ahead in three register sets.  The
    // compiler waits for ALL outstanding LDS reads at the first MFMA of a group (with LDS-DMA in
    // flight it does not count them), so that MFMA comes BEFORE the reads of group g+2 are issued:
    // what is outstanding at the wait was issued a whole group (512 cycles) ago.
    float av[3][8], bv[3][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[0][u] = pa[u * 64]; bv[0][u] = pb[u * 64];
      av[1][u] = pa[(8 + u) * 64]; bv[1][u] = pb[(8 + u) * 64];
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g % 3][0], bv[g % 3][0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 < 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          av[(g + 2) % 3][u] = pa[((g + 2) * 8 + u) * 64];
          bv[(g + 2) % 3][u] = pb[((g + 2) * 8 + u) * 64];
        }
      }
      if (g < 4) {                            // the whole next K-step is in flight after half of this one
        MAPX_DEEP_DMA(cur ^ 1, 2 * g);
        MAPX_DEEP_DMA(cur ^ 1, 2 * g + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      // (one accumulation chain: a second, independent chain was measured and changes nothing —
      // a dependent 32x32x2 MFMA issues back to back)
#pragma unroll
      for (int u = 1; u < 8; ++u)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g % 3][u], bv[g % 3][u], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    ga += (int64_t)kDeepBK * a.lda;
    gb += (int64_t)kDeepBK * a.ldb;
    __syncthreads();
  }
#undef MAPX_DEEP_DMA

  float* __restrict__ C = a.C + (int64_t)blockIdx.y * a.slab_stride;
  const int n = n0 + bbase + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + abase + 4 * kh + (r & 3) + 8 * (r >> 2);     // C/D map of the 32x32 MFMA
    if (m < a.M && n < a.N) C[(int64_t)m * a.ldc + n] = acc[r];
  }
}

static hipError_t deep_raise_lds() {
  static hipError_t done = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_deep_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDeepLds);
  return done;
}

static bool tn_deep_enabled() {
  static int v = [] { const char* e = getenv("MAPX_DW_DEEP"); return e ? atoi(e) : 1; }();
  return v != 0;
}

template <int WMT, int WNT, bool A_KC, bool B_KC, int BK>
static void launch_tile(const GemmArgs& a, bool vec, int nsplit, hipStream_t stream) {
  dim3 grid(a.tiles_m * a.tiles_n, nsplit);
  if (vec)
    hipLaunchKernelGGL((gemm_f32_kernel<WMT, WNT, A_KC, B_KC, true, BK>), grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<WMT, WNT, A_KC, B_KC, false, BK>), grid, dim3(256), 0, stream, a);
}

template <bool A_KC, bool B_KC>
static void launch_layout(GemmArgs& a, bool vec, int tile, int nsplit, hipStream_t stream) {
  if (tile == 2 || tile == 3) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
    if (tile == 3) launch_tile<2, 2, A_KC, B_KC, 64>(a, vec, nsplit, stream);
    else launch_tile<2, 2, A_KC, B_KC, 32>(a, vec, nsplit, stream);
  } else if (tile == 1) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 63) / 64;
    launch_tile<2, 1, A_KC, B_KC, 32>(a, vec, nsplit, stream);
  } else {
    a.tiles_m = (a.M + 63) / 64; a.tiles_n = (a.N + 63) / 64;
    if (tile == 4) launch_tile<1, 1, A_KC, B_KC, 64>(a, vec, nsplit, stream);
    else launch_tile<1, 1, A_KC, B_KC, 32>(a, vec, nsplit, stream);
  }
}

// gemm_x3.hip: the same product on the bf16 matrix cores (three bf16 pieces per fp32 operand, six MFMAs)
int gemm_f32x3_launch(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, int epi, const float* bias, const float* aux1, int64_t ld1,
                      const float* aux2, int64_t ld2, float* out2, int64_t ldo2, int nsplit, int tile_hint, void* ws,
                      size_t ws_bytes, int* nsplit_deferred, hipStream_t stream);

// MAPX_GEMM = x3 (default): fp32 GEMMs as 3 x bf16 split products (gemm_x3.hip); mfma32: v_mfma_f32_32x32x2_f32
static int gemm_mode() {
  static int m = [] {
    const char* e = getenv("MAPX_GEMM");
    return (e && strcmp(e, "mfma32") == 0) ? 0 : 1;
  }();
  return m;
}

}  // namespace mapx

extern "C" int mapx_gemm_f32_mode(void) { return mapx::gemm_mode(); }

extern "C" size_t mapx_gemm_splitk_workspace_bytes(int M, int N, int nsplit) {
  return nsplit > 1 ? (size_t)nsplit * M * N * sizeof(float) : 0;
}

extern "C" int mapx_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, int epi,
                             const float* bias, const float* aux1, int64_t ld1, const float* aux2,
                             int64_t ld2, float* out2, int64_t ldo2, int nsplit, int tile_hint,
                             void* ws, size_t ws_bytes, int* nsplit_deferred, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_f32: negative size");
  if (M == 0 || N == 0) return MAPX_OK;
  MAPX_REQUIRE(A && B && C, "gemm_f32: null operand");
  MAPX_REQUIRE(!(a_kc == 0 && b_kc != 0), "gemm_f32: layout (A m-contiguous, B k-contiguous) unused");
  MAPX_REQUIRE(epi >= MAPX_EPI_NONE && epi <= MAPX_EPI_RELU_MASK_COLSUM, "gemm_f32: bad epilogue %d", epi);
  {   // the epilogue addresses every output / auxiliary operand with 32-bit element offsets
    const int64_t lim = (int64_t)1 << 31, rows = M > 0 ? M : 1;
    MAPX_REQUIRE(rows * ldc < lim && rows * ld1 < lim && rows * ld2 < lim && rows * ldo2 < lim,
                 "gemm_f32: an output or auxiliary operand spans 2^31 elements or more");
  }
  if (epi >= MAPX_EPI_BIAS && epi <= MAPX_EPI_BIAS_CROSS) MAPX_REQUIRE(bias, "gemm_f32: bias missing");
  if (epi == MAPX_EPI_BIAS_CROSS) MAPX_REQUIRE(aux1 && aux2 && out2, "gemm_f32: cross operands missing");
  if (epi == MAPX_EPI_ADD || epi == MAPX_EPI_RELU_MASK || epi == MAPX_EPI_RELU_MASK_COLSUM)
    MAPX_REQUIRE(aux1, "gemm_f32: aux missing");
  MAPX_REQUIRE(epi >= MAPX_EPI_NONE && epi <= MAPX_EPI_RELU_MASK_COLSUM, "gemm_f32: unknown epilogue %d", epi);
  if (epi == MAPX_EPI_RELU_MASK_COLSUM && gemm_mode() != 1) {
    set_error("gemm_f32: EPI_RELU_MASK_COLSUM exists in the MAPX_GEMM=x3 family only");
    return MAPX_EINVAL;
  }
  if (nsplit < 1) nsplit = 1;
  MAPX_REQUIRE(nsplit == 1 || epi == MAPX_EPI_NONE, "gemm_f32: split-K needs EPI_NONE");
  if (gemm_mode() == 1)
    return gemm_f32x3_launch(a_kc, b_kc, M, N, K, A, lda, B, ldb, C, ldc, epi, bias, aux1, ld1, aux2, ld2, out2, ldo2,
                             nsplit, tile_hint, ws, ws_bytes, nsplit_deferred, stream);

  GemmArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias;
  g.aux1 = aux1; g.ld1 = ld1; g.aux2 = aux2; g.ld2 = ld2; g.out2 = out2; g.ldo2 = ldo2;
  g.k_chunk = K > 0 ? K : kBK; g.slab_stride = 0;
  if (nsplit > 1) {
    const size_t need = mapx_gemm_splitk_workspace_bytes(M, N, nsplit);
    if (!ws || ws_bytes < need) {
      set_error("gemm_f32: split-K workspace %zu < %zu", ws_bytes, need);
      return MAPX_EWORKSPACE;
    }
    int kc = (int)ceil_div(ceil_div(K, nsplit), 64) * 64;
    g.k_chunk = kc;
    nsplit = (int)ceil_div(K, kc);
    g.C = static_cast<float*>(ws);
    g.ldc = N;
    g.slab_stride = (int64_t)M * N;
  }
  // float4 loads need every float4 to be wholly inside or outside the matrix
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) &&
                   ((uintptr_t)B % 16 == 0) && (g.k_chunk % 4 == 0) &&
                   (a_kc ? (K % 4 == 0) : (M % 4 == 0)) && (b_kc ? (K % 4 == 0) : (N % 4 == 0));
  // tile choice (measured on MI355X, tools/gemm_bench.py): 128x128 tiles only when they fill
  // the 256 CUs evenly; otherwise 64x64 tiles (4 blocks per CU hide the per-tile
  // prologue/epilogue).  128x64 never won by more than 2 %.
  auto blocks = [&](int bm, int bn) { return ceil_div(M, bm) * ceil_div(N, bn) * nsplit; };
  const int64_t big = blocks(128, 128);
  int tile = (big >= 240 && (big % 256 == 0 || big % 256 >= 224 || big >= 1024)) ? 2 : 0;
  g.dbg = tile_hint >= 0 ? (tile_hint >> 8) : 0;
  if (tile_hint >= 0) tile_hint &= 255;
  if (tile_hint >= 0 && tile_hint <= 4) tile = tile_hint;   // 2: 128x128, 1: 128x64, 0: 64x64; 3/4: 128x128 / 64x64 with BK = 64
  // weight gradients whose 64x64 tiles (x splits) make one round of blocks: the deep-K kernel
  const bool deep = !a_kc && !b_kc && epi == MAPX_EPI_NONE && vec && tn_deep_enabled() && tile_hint < 0 &&
                    M >= 64 && N >= 64 && K % kDeepBK == 0 && g.k_chunk % kDeepBK == 0 &&
                    g.k_chunk >= 2 * kDeepBK && blocks(64, 64) <= 256;
  if (deep) {
    MAPX_HIP(deep_raise_lds());
    g.tiles_m = (M + 63) / 64; g.tiles_n = (N + 63) / 64;
    hipLaunchKernelGGL(gemm_tn_deep_kernel, dim3(g.tiles_m * g.tiles_n, nsplit), dim3(256), kDeepLds, stream, g);
  } else if (a_kc && b_kc) launch_layout<true, true>(g, vec, tile, nsplit, stream);
  else if (a_kc) launch_layout<true, false>(g, vec, tile, nsplit, stream);
  else launch_layout<false, false>(g, vec, tile, nsplit, stream);
  if (nsplit_deferred) *nsplit_deferred = nsplit > 1 ? nsplit : 0;   // caller sums the slabs later
  if (nsplit > 1 && !nsplit_deferred) {
    // slabs are dense [M,N]; combine into the caller's C (ldc must equal N for split-K)
    MAPX_REQUIRE(ldc == N, "gemm_f32: split-K output must be dense (ldc == N)");
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                       static_cast<const float*>(ws), g.slab_stride, nsplit, n, C);
  }
  return check_launch("gemm_f32");
}

extern "C" int mapx_enc_group_layout(const int64_t* masked_index, int T, int L, int F, int cap_slots,
                                     int32_t* rowmap, int32_t* hpos, int32_t* tile_group,
                                     int32_t* group_start, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(masked_index && rowmap && hpos && tile_group && group_start, "enc_group_layout: null pointer");
  MAPX_REQUIRE(F >= 1 && F <= 64 && L >= 1 && T >= 0 && cap_slots % 128 == 0 && cap_slots >= T + 127 * F,
               "enc_group_layout: bad sizes (F <= 64, cap_slots multiple of 128 and >= T + 127 F)");
  MAPX_REQUIRE(T <= kLayoutMaxT, "enc_group_layout: at most %d targets per step", kLayoutMaxT);
  const size_t dyn = ((size_t)T + 15) & ~(size_t)15;
  static bool raised = false;
  if (!raised) {       // above the 64 KB default of dynamic LDS
    MAPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_group_layout_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kLayoutMaxT));
    raised = true;
  }
  static const bool one_block = [] { const char* e = getenv("MAPX_LAYOUT_ONE_BLOCK"); return e && atoi(e) != 0; }();
  if (one_block) {
    hipLaunchKernelGGL(enc_group_layout_kernel, dim3(1), dim3(kLayoutThreads), dyn, stream, masked_index, T, L, F,
                       cap_slots, rowmap, hpos, tile_group, group_start);
  } else {
    static bool raised_mw = false;
    if (!raised_mw) {
      MAPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_group_layout_mw_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLayoutMaxT));
      raised_mw = true;
    }
    hipLaunchKernelGGL(enc_group_layout_mw_kernel, dim3(F), dim3(kLayoutMwThreads), dyn, stream, masked_index, T, L, F,
                       cap_slots, rowmap, hpos, tile_group, group_start);
  }
  return check_launch("enc_group_layout");
}

extern "C" int mapx_enc_grouped_fwd(const float* final_act, int64_t ld_final, int nrows, int K, const float* W,
                                    int64_t ldw, const float* bias, const int32_t* rowmap,
                                    const int32_t* tile_group, const int32_t* group_start_opt, int F,
                                    int cap_slots, float* h_slots, float* zero_slots_opt, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(final_act && W && bias && rowmap && tile_group && h_slots, "enc_grouped_fwd: null pointer");
  MAPX_REQUIRE(K % 4 == 0 && ld_final % 4 == 0 && ldw % 4 == 0 && cap_slots % 128 == 0,
               "enc_grouped_fwd: K, leading dimensions %% 4 and cap_slots %% 128 must be 0");
  GroupedArgs g{};
  g.A = final_act; g.lda = ld_final; g.B = W; g.ldb = ldw; g.C = h_slots; g.ldc = 32; g.bias = bias;
  g.rowmap = rowmap; g.tile_group = tile_group; g.K = K; g.N = 32; g.nrows = nrows;
  g.group_start = group_start_opt; g.F = F;
  g.zero_out = zero_slots_opt;
  // bf16-matrix-core form (gemm_x3.hip) whenever the dense GEMMs use it and chunks of 8 floats line up
  if (gemm_mode() == 1 && K % 8 == 0 && (uintptr_t)final_act % 16 == 0 && (uintptr_t)W % 16 == 0) {
    MAPX_HIP(enc_grouped_fwd_x3_launch(g, cap_slots, stream));
    return check_launch("enc_grouped_fwd (3 x bf16)");
  }
  hipLaunchKernelGGL(gemm_grouped_kernel<false>, dim3(cap_slots / 128), dim3(256), 0, stream, g);
  return check_launch("enc_grouped_fwd");
}

extern "C" int mapx_enc_grouped_dw(const float* dh_slots, const float* final_act, int64_t ld_final, int nrows,
                                   int N, const int32_t* rowmap, const int32_t* group_start, int F,
                                   const float* gscale_opt, float* dW, int64_t ldw, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dh_slots && final_act && rowmap && group_start && dW, "enc_grouped_dw: null pointer");
  MAPX_REQUIRE(N % 4 == 0 && ld_final % 4 == 0 && F >= 1, "enc_grouped_dw: N, ld %% 4 must be 0");
  GroupedArgs g{};
  g.A = dh_slots; g.lda = 32; g.B = final_act; g.ldb = ld_final; g.C = dW; g.ldc = ldw;
  g.rowmap = rowmap; g.group_start = group_start; g.F = F; g.K = 0; g.N = N; g.nrows = nrows;
  g.gscale = gscale_opt;
  if (gemm_mode() == 1 && N % 8 == 0 && (uintptr_t)final_act % 16 == 0 && (uintptr_t)dh_slots % 16 == 0) {
    MAPX_HIP(enc_grouped_dw_x3_launch(g, F, stream));
    return check_launch("enc_grouped_dw (3 x bf16)");
  }
  hipLaunchKernelGGL(gemm_grouped_kernel<true>, dim3((N + 127) / 128, F), dim3(256), 0, stream, g);
  return check_launch("enc_grouped_dw");
}

extern "C" size_t mapx_colsum_workspace_bytes(int N) {
  return (size_t)mapx::kColChunks * N * sizeof(float);
}

extern "C" int mapx_sum_tasks(const mapx_sum_task* tasks_host, int ntasks, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(ntasks >= 0 && ntasks <= kMaxSumTasks, "sum_tasks: at most %d tasks per call", kMaxSumTasks);
  if (ntasks == 0) return MAPX_OK;
  MAPX_REQUIRE(tasks_host, "sum_tasks: null task list");
  SumTasks t;
  memset(&t, 0, sizeof(t));
  for (int i = 0; i < ntasks; ++i) {
    MAPX_REQUIRE(tasks_host[i].dst && tasks_host[i].src && tasks_host[i].nsplit >= 1 && tasks_host[i].n >= 0,
                 "sum_tasks: bad task %d", i);
    t.t[i] = tasks_host[i];
  }
  hipLaunchKernelGGL(sum_tasks_kernel, dim3(96, ntasks), dim3(256), 0, stream, t);
  return check_launch("sum_tasks");
}

extern "C" int mapx_colsum_chunks(void) { return mapx::kColChunks; }

extern "C" int mapx_colsum(const float* x, int64_t ld, int M, int N, float* out, void* ws,
                           size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(x && M >= 0 && N > 0, "colsum: bad arguments");
  if (!ws || ws_bytes < mapx_colsum_workspace_bytes(N)) {
    set_error("colsum: workspace too small");
    return MAPX_EWORKSPACE;
  }
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((N + 63) / 64, kColChunks), dim3(256), 0, stream, x,
                     ld, M, N, part);
  if (out)   // out == NULL: the caller sums the kColChunks partial rows later (mapx_sum_tasks)
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, out);
  return check_launch("colsum");
}

extern "C" int mapx_relu_mask_colsum(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int M, int N,
                                     float* dz, float* db,
                                      void* ws, size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dy && y && dz && M >= 0 && N > 0, "relu_mask_colsum: bad arguments");
  MAPX_REQUIRE(N % 4 == 0 && ld_dy % 4 == 0 && ld_dy >= N && (uintptr_t)dy % 16 == 0 && ld_y % 4 == 0 &&
                   ld_y >= N && (uintptr_t)y % 16 == 0,
               "relu_mask_colsum: N, ld_dy, ld_y %% 4 != 0 or dy / y not 16-byte aligned");
  if (!ws || ws_bytes < mapx_colsum_workspace_bytes(N)) {
    set_error("relu_mask_colsum: workspace too small");
    return MAPX_EWORKSPACE;
  }
  float* part = static_cast<float*>(ws);
  if (ew_narrow_lanes(N))
    hipLaunchKernelGGL((ew_colsum_kernel<0, 32>), dim3((N + 127) / 128, kColChunks), dim3(256), 0, stream, dy, ld_dy, y,
                       ld_y, (const float*)nullptr, M, N, dz, (float*)nullptr, 0, part);
  else
    hipLaunchKernelGGL((ew_colsum_kernel<0, 64>), dim3((N + 255) / 256, kColChunks), dim3(256), 0, stream, dy, ld_dy, y,
                       ld_y, (const float*)nullptr, M, N, dz, (float*)nullptr, 0, part);
  if (db) hipLaunchKernelGGL(colsum_stage2_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, db);
  return check_launch("relu_mask_colsum");
}

extern "C" int mapx_cross_bwd_pre_colsum(const float* g, int64_t ld_g, const float* x0, const float* u, int M, int N,
                                         float* t, float* dx0, int accumulate, float* db, void* ws,
                                         size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(g && x0 && u && t && dx0 && M >= 0 && N > 0, "cross_bwd_pre_colsum: bad arguments");
  MAPX_REQUIRE(N % 4 == 0 && ld_g % 4 == 0 && ld_g >= N && (uintptr_t)g % 16 == 0,
               "cross_bwd_pre_colsum: N, ld_g %% 4 != 0 or g not 16-byte aligned");
  if (!ws || ws_bytes < mapx_colsum_workspace_bytes(N)) {
    set_error("cross_bwd_pre_colsum: workspace too small");
    return MAPX_EWORKSPACE;
  }
  float* part = static_cast<float*>(ws);
  if (ew_narrow_lanes(N))
    hipLaunchKernelGGL((ew_colsum_kernel<1, 32>), dim3((N + 127) / 128, kColChunks), dim3(256), 0, stream, g, ld_g, x0,
                       (int64_t)N, u, M, N, t, dx0, accumulate, part);
  else
    hipLaunchKernelGGL((ew_colsum_kernel<1, 64>), dim3((N + 255) / 256, kColChunks), dim3(256), 0, stream, g, ld_g, x0,
                       (int64_t)N, u, M, N, t, dx0, accumulate, part);
  if (db) hipLaunchKernelGGL(colsum_stage2_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, db);
  return check_launch("cross_bwd_pre_colsum");
}

extern "C" int mapx_cross_bwd_pre(const float* g, const float* x0, const float* u, int64_t n,
                                  float* t, float* dx0, int accumulate, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(g && x0 && u && t && dx0 && n >= 0 && n % 4 == 0, "cross_bwd_pre: bad arguments");
  if (n == 0) return MAPX_OK;
  hipLaunchKernelGGL(cross_bwd_pre_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, stream, g, x0, u,
                     n / 4, t, dx0, accumulate);
  return check_launch("cross_bwd_pre");
}
