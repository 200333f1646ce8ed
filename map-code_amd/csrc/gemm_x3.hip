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
// the LDS store, in the shadow of the MFMAs).  Measured error against fp64: see tests (same bound as
// the fp32-MFMA kernel of gemm.hip, which stays selectable: MAPX_GEMM=mfma32).
//
// Same interface as gemm.hip: operand storage flags (A_KC / B_KC), epilogues, split-K slabs.
// Tiling: 256 threads = 2x2 waves, block tile (64 WMT) x (64 WNT), BK = 32; per operand three bf16
// planes in LDS, laid out like gemm_bf16.hip's single plane (k-contiguous: [row][32 + 8], one
// ds_read_b128 per fragment, conflict-free at the 80-byte row stride; k-strided: [k][rows + 32], two
// ds_read_b64_tr_b16 per fragment).  One barrier per K-step, two LDS buffers; the epilogue goes
// through LDS as fp32 rows and moves every operand with 16-byte accesses.
#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmX3Args {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int M, N, K;
  int epi;
  const float* bias;
  const float* aux1; int64_t ld1;
  const float* aux2; int64_t ld2;
  float* out2; int64_t ldo2;
  int k_chunk;
  int64_t slab_stride;
  int tiles_m, tiles_n;
  int dbg;      // timing experiments (tile_hint >> 8): 1 no global loads in the loop, 2 no cut / LDS stores, 4 no MFMAs
};

constexpr int kXBK = 32;

// Cut of 8 fp32 values into three planes of 8 bf16, each piece ROUNDED to nearest (v_cvt_pk_bf16_f32)
// and the residual taken exactly in fp32: a = hi + mid + lo to within 2^-25 |a|, with pieces of either
// sign, so that what the six-term product drops has no preferred sign (a truncating cut biased every
// product toward zero by 3/4 of an fp32 ulp — measured, tools/scratch/bias_probe.py).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk_bf16(float x0, float x1) {
  bf16x2 p;
  p[0] = (__bf16)x0;
  p[1] = (__bf16)x1;
  return __builtin_bit_cast(uint32_t, p);
}
__device__ inline void cut3(const float (&x)[8], uint4& hi, uint4& mid, uint4& lo) {
  uint32_t H[4], M[4], L[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x0 = x[2 * e], x1 = x[2 * e + 1];
    H[e] = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(H[e] << 16), r1 = x1 - __uint_as_float(H[e] & 0xffff0000u);   // exact
    M[e] = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(M[e] << 16), s1 = r1 - __uint_as_float(M[e] & 0xffff0000u);   // exact
    L[e] = pk_bf16(s0, s1);
  }
  hi = make_uint4(H[0], H[1], H[2], H[3]);
  mid = make_uint4(M[0], M[1], M[2], M[3]);
  lo = make_uint4(L[0], L[1], L[2], L[3]);
}

// One operand: global fp32 tile -> registers (chunks of 8 floats) -> three bf16 planes in LDS -> fragments.
// VEC: leading dimension % 4 == 0, 16-B aligned base, contiguous extent % 8 == 0 (a chunk is all-in or
// all-out); otherwise 8 scalar loads with per-element predicates.
template <int ROWS, int T, bool KC, bool VEC, int NT>
struct OperandX3 {
  static constexpr int LD = KC ? kXBK + 8 : ROWS + 32;            // bf16 elements per stored row
  static constexpr int PLANE = KC ? ROWS * LD : kXBK * LD;         // elements of one plane
  static constexpr int LDS_ELEMS = 3 * PLANE;
  static constexpr int CPR = KC ? kXBK / 8 : ROWS / 8;             // chunks per stored row
  static constexpr int NV = ROWS * kXBK / 8 / NT;                  // chunks per thread per tile (NT threads)
  static_assert(NV >= 1, "tile too small for the block");
  float4 r[NV][2];
  bool ok[NV];

  __device__ static inline void coords(int f, int& row, int& col) {
    row = f / CPR;
    col = (f % CPR) * 8;
  }

  __device__ inline void load(const float* __restrict__ g, int64_t ld, int row0, int nrows, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
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
  }

  // cut + store (zero fill applied here, so that no ALU touches a load's result before it is needed)
  template <bool MASK>
  __device__ inline void store(bf16_t* __restrict__ s) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tr, tc;
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

  // fragments of k16-step s2 (k = 16 s2 + 8 (lane >> 5) + j) of plane `pl` for this wave's T tiles
  __device__ static inline void frags(const bf16_t* __restrict__ s, int pl, int base, int lane, int s2,
                                      bf16x8 (&f)[T]) {
    const int l31 = lane & 31, kh = lane >> 5;
    const bf16_t* sp = s + pl * PLANE;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (KC) {
        f[t] = *reinterpret_cast<const bf16x8*>(sp + (base + 32 * t + l31) * LD + 16 * s2 + 8 * kh);
      } else {
        const int q = (lane >> 2) & 3, p = lane & 3, half = (lane >> 4) & 1;
        const bf16_t* a0 = sp + (16 * s2 + 8 * kh + q) * LD + base + 32 * t + 16 * half + 4 * p;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * LD));
        f[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  }
};

// Row-major second pass of the epilogue over the fp32 tile in LDS (see gemm_bf16.hip), all operands fp32.
template <int EPI, int BM, int BN, int NT>
__device__ inline void epilogue_rows_x3(const GemmX3Args& a, float* __restrict__ C, const float* __restrict__ tile,
                                        int m0, int n0, bool vio) {
  constexpr int LDT = BN + 4;
  constexpr bool kBias = EPI >= MAPX_EPI_BIAS && EPI <= MAPX_EPI_BIAS_CROSS;
  constexpr bool kAux1 = EPI == MAPX_EPI_BIAS_CROSS || EPI == MAPX_EPI_ADD || EPI == MAPX_EPI_RELU_MASK;
  for (int idx = threadIdx.x; idx < BM * BN / 4; idx += NT) {
    const int row = idx / (BN / 4), c0 = (idx % (BN / 4)) * 4;
    const int m = m0 + row, n = n0 + c0;
    if (m >= a.M || n >= a.N) continue;
    const float4 t0 = *reinterpret_cast<const float4*>(tile + row * LDT + c0);
    float v[4] = {t0.x, t0.y, t0.z, t0.w}, x1[4], x2[4], u[4];
    const int64_t oc = (int64_t)m * a.ldc + n, o1 = (int64_t)m * a.ld1 + n, o2 = (int64_t)m * a.ld2 + n,
                  oo = (int64_t)m * a.ldo2 + n;
    if (vio) {
      if (kBias) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + n);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
      }
      if (kAux1) {
        const float4 p0 = *reinterpret_cast<const float4*>(a.aux1 + o1);
        x1[0] = p0.x; x1[1] = p0.y; x1[2] = p0.z; x1[3] = p0.w;
      }
      if (EPI == MAPX_EPI_BIAS_CROSS) {
        const float4 p0 = *reinterpret_cast<const float4*>(a.aux2 + o2);
        x2[0] = p0.x; x2[1] = p0.y; x2[2] = p0.z; x2[3] = p0.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = n + e < a.N;
        if (kBias) v[e] += in ? a.bias[n + e] : 0.f;
        if (kAux1) x1[e] = in ? a.aux1[o1 + e] : 0.f;
        if (EPI == MAPX_EPI_BIAS_CROSS) x2[e] = in ? a.aux2[o2 + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (EPI == MAPX_EPI_BIAS_RELU) v[e] = fmaxf(v[e], 0.f);
      u[e] = v[e];
      if (EPI == MAPX_EPI_BIAS_CROSS) v[e] = x1[e] + x2[e] * v[e];
      if (EPI == MAPX_EPI_ADD) v[e] += x1[e];
      if (EPI == MAPX_EPI_RELU_MASK) v[e] = x1[e] > 0.f ? v[e] : 0.f;
    }
    if (vio) {
      *reinterpret_cast<float4*>(C + oc) = make_float4(v[0], v[1], v[2], v[3]);
      if (EPI == MAPX_EPI_BIAS_CROSS) *reinterpret_cast<float4*>(a.out2 + oo) = make_float4(u[0], u[1], u[2], u[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < a.N) {
          C[oc + e] = v[e];
          if (EPI == MAPX_EPI_BIAS_CROSS) a.out2[oo + e] = u[e];
        }
      }
    }
  }
}

// WR x WC waves, each owning WMT x WNT MFMA tiles of 32 x 32: block tile (32 WMT WR) x (32 WNT WC).
// 8 waves (two per SIMD) on a 128 x 128 tile: while one wave of a SIMD cuts and stores its share of
// the next tile (VALU + LDS), its partner's MFMAs keep the matrix pipe busy — the overlap that one
// wave per SIMD only gets from a perfect instruction interleave.
template <int WR, int WC, int WMT, int WNT, bool A_KC, bool B_KC, bool VEC>
__global__ void __launch_bounds__(64 * WR * WC) gemm_f32x3_kernel(GemmX3Args a) {
  constexpr int BM = 32 * WMT * WR, BN = 32 * WNT * WC, NT = 64 * WR * WC;
  using OpA = OperandX3<BM, WMT, A_KC, VEC, NT>;
  using OpB = OperandX3<BN, WNT, B_KC, VEC, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* const smem = reinterpret_cast<bf16_t*>(smem_raw);
  constexpr int kBuf = OpA::LDS_ELEMS + OpB::LDS_ELEMS;

  const int nb = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x;
  const int per = nb / 8;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order (see gemm.hip)
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.y * a.k_chunk;
  const int kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
  float* __restrict__ C = a.C + (int64_t)blockIdx.y * a.slab_stride;

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
  if (nk > 0) MAPX_X_LOAD(0, 0);
  if (nk > 1) MAPX_X_LOAD(1, 1);
  if (nk > 0) MAPX_X_STORE(0, 0, true);
  if (nk > 2) MAPX_X_LOAD(0, 2);
  __syncthreads();
  // K-step on LDS buffer CUR (= kt & 1, literal): at its start set CUR^1 holds tile kt+1 (landed),
  // set CUR holds tile kt+2 (in flight, issued one step ago)
  // Timing experiments (tools/gemm_f32_bench.py ablate): build with -DMAPX_X3_ABLATE to make the K loop's
  // phases switchable from tile_hint >> 8; a plain build keeps the loop free of those branches (they cost
  // about 10 % of a K-step even when never taken).
#ifdef MAPX_X3_ABLATE
  const int kDbg = a.dbg;
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
  MAPX_X_LOOPS(false);
#undef MAPX_X_LOOPS
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
  auto al16 = [](const void* p, int64_t ld) { return p == nullptr || ((uintptr_t)p % 16 == 0 && ld % 4 == 0); };
  const bool vio = a.N % 4 == 0 && al16(C, a.ldc) && al16(a.aux1, a.ld1) && al16(a.aux2, a.ld2) && al16(a.out2, a.ldo2) &&
                   al16(a.bias, 0);
  switch (a.epi) {
    case MAPX_EPI_BIAS: epilogue_rows_x3<MAPX_EPI_BIAS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_BIAS_RELU: epilogue_rows_x3<MAPX_EPI_BIAS_RELU, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_BIAS_CROSS: epilogue_rows_x3<MAPX_EPI_BIAS_CROSS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_ADD: epilogue_rows_x3<MAPX_EPI_ADD, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_RELU_MASK: epilogue_rows_x3<MAPX_EPI_RELU_MASK, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    default: epilogue_rows_x3<MAPX_EPI_NONE, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
  }
}

template <int WR, int WC, int WMT, int WNT, bool A_KC, bool B_KC, bool VEC>
static hipError_t launch_one_x3(const GemmX3Args& a, int nsplit, hipStream_t stream) {
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
  hipLaunchKernelGGL(fn, dim3(a.tiles_m * a.tiles_n, nsplit), dim3(NT), lds, stream, a);
  return hipSuccess;
}

// tile 2 (and 3): 128 x 128 by 8 waves (2 x 4, wave tile 64 x 32; 4 waves of 64 x 64 would need 256
// accumulator registers for the two sets and measured 15 % slower with one); tile 1: 128 x 64 by 4
// waves; tile 0: 64 x 64 by 4 waves
template <bool A_KC, bool B_KC>
static hipError_t launch_layout_x3(GemmX3Args& a, bool vec, int tile, int nsplit, hipStream_t stream) {
#define MAPX_X3(WR_, WC_, WM, WN) (vec ? launch_one_x3<WR_, WC_, WM, WN, A_KC, B_KC, true>(a, nsplit, stream) \
                                       : launch_one_x3<WR_, WC_, WM, WN, A_KC, B_KC, false>(a, nsplit, stream))
  // (128x128 by 4 waves with 64x64 wave tiles — 2/3 of the LDS fragment reads per MFMA — measured 8-14 %
  // slower on every shape of the step: one wave per SIMD leaves the cut nothing to hide behind.)
  if (tile == 2 || tile == 3) {
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

// Called by mapx_gemm_f32 (gemm.hip) when the split-bf16 path is selected.  Same contract.
int gemm_f32x3_launch(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, int epi, const float* bias, const float* aux1, int64_t ld1,
                      const float* aux2, int64_t ld2, float* out2, int64_t ldo2, int nsplit, int tile_hint, void* ws,
                      size_t ws_bytes, int* nsplit_deferred, hipStream_t stream) {
  GemmX3Args g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias;
  g.aux1 = aux1; g.ld1 = ld1; g.aux2 = aux2; g.ld2 = ld2; g.out2 = out2; g.ldo2 = ldo2;
  g.k_chunk = K > 0 ? K : kXBK; g.slab_stride = 0;
  g.dbg = tile_hint >= 0 ? (tile_hint >> 8) : 0;
  if (nsplit > 1) {
    const size_t need = (size_t)nsplit * M * N * sizeof(float);
    if (!ws || ws_bytes < need) {
      set_error("gemm_f32: split-K workspace %zu < %zu", ws_bytes, need);
      return MAPX_EWORKSPACE;
    }
    const int kc = (int)ceil_div(ceil_div(K, nsplit), 64) * 64;
    g.k_chunk = kc;
    nsplit = (int)ceil_div(K, kc);
    g.C = static_cast<float*>(ws);
    g.ldc = N;
    g.slab_stride = (int64_t)M * N;
  }
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) &&
                   (g.k_chunk % 8 == 0) && (a_kc ? (K % 8 == 0) : (M % 8 == 0)) && (b_kc ? (K % 8 == 0) : (N % 8 == 0));
  auto blocks = [&](int bm, int bn) { return ceil_div(M, bm) * ceil_div(N, bn) * nsplit; };
  const int64_t big = blocks(128, 128);
  int tile = (big >= 160) ? 2 : 0;
  if (tile_hint >= 0 && (tile_hint & 255) <= 3) tile = tile_hint & 255;
  hipError_t e;
  if (a_kc && b_kc) e = launch_layout_x3<true, true>(g, vec, tile, nsplit, stream);
  else if (a_kc) e = launch_layout_x3<true, false>(g, vec, tile, nsplit, stream);
  else e = launch_layout_x3<false, false>(g, vec, tile, nsplit, stream);
  MAPX_HIP(e);
  if (nsplit_deferred) *nsplit_deferred = nsplit > 1 ? nsplit : 0;
  if (nsplit > 1 && !nsplit_deferred) {
    MAPX_REQUIRE(ldc == N, "gemm_f32: split-K output must be dense (ldc == N)");
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_x3_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                       static_cast<const float*>(ws), g.slab_stride, nsplit, n, C);
  }
  return check_launch("gemm_f32 (3 x bf16)");
}

}  // namespace mapx
