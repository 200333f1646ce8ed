// bf16 compute mode (BASELINE configs[2]: "8 x MI355X DP bf16", north star tolerance 1e-2):
// the dense part of DCNv2 — CrossNetV2 (code/layers.py:197-201), MLPBlock (layers.py:173-188),
// feat_encoder / pred_rfd / fc_out (models.py:74,119-124,304) and their backward products — on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  Operands are bf16 in HBM (activations are
// stored in bf16 by the producing kernel's epilogue; weights are bf16 shadows of the fp32 master
// weights, refreshed by the optimizer kernel), accumulation, bias / ReLU / cross epilogues and
// every weight gradient are fp32.
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(k,n) )
//
// Same operand description as gemm.hip (no transposed copy of anything exists):
//   A_KC : A(m,k) = A[m*lda + k]  (k contiguous)      else A(m,k) = A[k*lda + m]
//   B_KC : B(k,n) = B[n*ldb + k]  (k contiguous)      else B(k,n) = B[k*ldb + n]
//   forward  Y = X W^T : A_KC, B_KC       dX = dY W : A_KC, B k-strided       dW = dY^T X : both k-strided
//
// Tiling: 256 threads = 2x2 waves, wave tile (32 WMT) x (32 WNT), block tile (64 WMT) x (64 WNT),
// BK = 64 (one 128-B line of bf16 per row and K-step).  Each operand keeps its GLOBAL orientation
// in LDS, so global -> LDS is a straight 16-byte copy for every layout, and the MFMA fragment
// (lane l: 8 consecutive k for row / column l & 31, k-half l >> 5) is fetched with
//   k-contiguous operand: LDS [row][64 + 8]   one ds_read_b128 (row stride 36 dwords: the 16 rows
//                                             of a b128 lane group land on 16 distinct 4-bank slots);
//   k-strided operand:    LDS [k][rows + 32]  two ds_read_b64_tr_b16 — gfx950's transposing LDS read:
//                                             per 16 lanes a block of 4 k-rows x 16 columns comes
//                                             back column-major, i.e. as 4 consecutive k of one
//                                             column per lane (row stride = 64 B mod 256 B: the 4
//                                             rows of a block sit on the 4 quarters of the bank row).
// Pipeline: one barrier per K-step, two LDS buffers, two register sets — global loads are issued
// two K-steps ahead of the MFMAs that consume them (a K-step of bf16 MFMAs is shorter than an L2
// round trip), LDS stores one K-step ahead.
#include "../../include/mapx_hip.h"
#include "common.h"
#include <utility>

namespace mapx {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GemmHArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* B; int64_t ldb;
  void* C; int64_t ldc;              // bf16 or fp32 (c_f32)
  int M, N, K;
  int epi;
  const float* bias;                 // [N] fp32
  const void* aux1; int64_t ld1;     // CROSS: Xi (bf16)   ADD: aux (bf16 or fp32: aux1_f32)   RELU_MASK: y (bf16)
  const bf16_t* aux2; int64_t ld2;   // CROSS: X0
  bf16_t* out2; int64_t ldo2;        // CROSS: u = W Xi + b (kept for backward)
  int k_chunk;                       // split-K: K range per blockIdx.y (multiple of 64)
  int64_t slab_stride;               // split-K: fp32 C offset per split
  int tiles_m, tiles_n;
  int dbg;                           // timing experiments only (tile_hint >> 8): 1 = no global stores, 2 = no K loop
};

constexpr int kHBK = 64;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression
template <class F, int... Z>
__device__ __forceinline__ void unroll_seq_h(F&& f, std::integer_sequence<int, Z...>) {
  (f(std::integral_constant<int, Z>{}), ...);
}

// One operand's staging: global tile -> registers (16-byte chunks of 8 bf16) -> LDS, and LDS -> fragments.
// VEC requires: leading dimension % 8 == 0, 16-B aligned base, contiguous extent % 8 == 0: a chunk is
// all-in or all-out.  Otherwise 8 scalar loads with per-element predicates.
template <int ROWS, int T, bool KC, bool VEC>
struct OperandH {
  static constexpr int LD = KC ? kHBK + 8 : ROWS + 32;          // bf16 elements
  static constexpr int LDS_ELEMS = KC ? ROWS * LD : kHBK * LD;
  static constexpr int CPR = KC ? kHBK / 8 : ROWS / 8;            // 16-B chunks per stored row
  static constexpr int NV = ROWS * kHBK / 8 / 256;                // chunks per thread per tile
  uint4 r[NV];
  bool ok[NV];      // VEC: chunk i lies inside the matrix; applied when the registers are WRITTEN to LDS, so
                    // that no ALU instruction (and with it no s_waitcnt) touches a load's result early

  __device__ static inline void coords(int f, int& row, int& col) {
    row = f / CPR;
    col = (f % CPR) * 8;
  }

  // tile at (row0 of the non-k extent, k0); rows beyond nrows / k beyond kend read as zero
  __device__ inline void load(const bf16_t* __restrict__ g, int64_t ld, int row0, int nrows, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tr, tc;
      coords(threadIdx.x + i * 256, tr, tc);
      const int gr = (KC ? row0 : k0) + tr, gc = (KC ? k0 : row0) + tc;
      const int rlim = KC ? nrows : kend, clim = KC ? kend : nrows;
      const bool rok = gr < rlim;
      const bf16_t* p = g + (int64_t)(rok ? gr : 0) * ld;
      if (VEC) {
        ok[i] = rok && gc < clim;
        r[i] = *reinterpret_cast<const uint4*>(p + (ok[i] ? gc : 0));
      } else {
        ok[i] = true;
        unsigned short h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool oke = rok && gc + e < clim;
          const unsigned short x = reinterpret_cast<const unsigned short*>(p)[oke ? gc + e : 0];
          h[e] = oke ? x : (unsigned short)0;
        }
        r[i] = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16),
                          h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
      }
    }
  }

  // MASK = false: the block's tile lies wholly inside the matrix and K (no zero fill needed)
  template <bool MASK>
  __device__ inline void store(bf16_t* __restrict__ s) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tr, tc;
      coords(threadIdx.x + i * 256, tr, tc);
      *reinterpret_cast<uint4*>(s + tr * LD + tc) = (!MASK || ok[i]) ? r[i] : make_uint4(0u, 0u, 0u, 0u);
    }
  }

  // fragment of k16-step s4 (k = 16 s4 + 8 (lane >> 5) + j) for this wave's tile t
  __device__ static inline bf16x8 frag1(const bf16_t* __restrict__ s, int base, int lane, int s4, int t) {
    const int l31 = lane & 31, kh = lane >> 5;
    if (KC) return *reinterpret_cast<const bf16x8*>(s + (base + 32 * t + l31) * LD + 16 * s4 + 8 * kh);
    // transposing read: lane 4q+p of a 16-lane group addresses block row q, columns 4p..4p+3, and
    // receives column (lane & 15) of the block's 4 rows
    const int q = (lane >> 2) & 3, p = lane & 3, half = (lane >> 4) & 1;
    const bf16_t* a0 = s + (16 * s4 + 8 * kh + q) * LD + base + 32 * t + 16 * half + 4 * p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * LD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  __device__ static inline void frags(const bf16_t* __restrict__ s, int base, int lane, int s4, bf16x8 (&f)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) f[t] = frag1(s, base, lane, s4, t);
  }
};

template <bool F32>
__device__ inline float ld_as_f32(const void* __restrict__ p, uint32_t off) {
  if (F32) return reinterpret_cast<const float*>(p)[off];
  return (float)reinterpret_cast<const bf16_t*>(p)[off];
}
template <bool F32>
__device__ inline void st_from_f32(void* __restrict__ p, uint32_t off, float v) {
  if (F32) reinterpret_cast<float*>(p)[off] = v;
  else reinterpret_cast<bf16_t*>(p)[off] = (bf16_t)v;      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}

// Epilogue.  The MFMA leaves a lane with ONE column and 16 scattered rows of a 32x32 tile (C/D map,
// dtype independent: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); storing bf16 from there is
// 64 two-byte stores per lane in 64-byte row pieces, and the auxiliary operands would be gathered the
// same way — measured: 8 of the 10 us a K = 64 GEMM took.  So the accumulators go through LDS (free
// after the K loop) as fp32 [BM][BN + 4], and a second pass walks the tile row-major, 8 consecutive
// columns per thread: bias, auxiliary operands and results all move as 16-byte accesses of whole
// 128/256-byte row segments.  `vio`: every operand of this launch allows that (block-uniform); else
// the same pass runs element by element.
template <int EPI, bool C_F32, bool AUX_F32, int BM, int BN>
__device__ inline void epilogue_rows_h(const GemmHArgs& a, void* __restrict__ C, const float* __restrict__ tile,
                                       int m0, int n0, bool vio) {
  constexpr int LDT = BN + 4;
  constexpr bool kBias = EPI >= MAPX_EPI_BIAS && EPI <= MAPX_EPI_BIAS_CROSS;
  constexpr bool kAux1 = EPI == MAPX_EPI_BIAS_CROSS || EPI == MAPX_EPI_ADD || EPI == MAPX_EPI_RELU_MASK;
  for (int idx = threadIdx.x; idx < BM * BN / 8; idx += 256) {
    const int row = idx / (BN / 8), c0 = (idx % (BN / 8)) * 8;
    const int m = m0 + row, n = n0 + c0;
    if (m >= a.M || n >= a.N) continue;
    float v[8], x1[8], x2[8], u[8];
    const float4 t0 = *reinterpret_cast<const float4*>(tile + row * LDT + c0);
    const float4 t1 = *reinterpret_cast<const float4*>(tile + row * LDT + c0 + 4);
    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
    const int64_t oc = (int64_t)m * a.ldc + n, o1 = (int64_t)m * a.ld1 + n, o2 = (int64_t)m * a.ld2 + n,
                  oo = (int64_t)m * a.ldo2 + n;
    if (vio) {             // n + 8 <= N (N % 8 == 0), every address 16-byte aligned
      if (kBias) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + n), b1 = *reinterpret_cast<const float4*>(a.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (kAux1) {
        if (AUX_F32) {
          const float4 p0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.aux1) + o1);
          const float4 p1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.aux1) + o1 + 4);
          x1[0] = p0.x; x1[1] = p0.y; x1[2] = p0.z; x1[3] = p0.w; x1[4] = p1.x; x1[5] = p1.y; x1[6] = p1.z; x1[7] = p1.w;
        } else {
          const bf16x8 p = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(a.aux1) + o1);
#pragma unroll
          for (int e = 0; e < 8; ++e) x1[e] = (float)p[e];
        }
      }
      if (EPI == MAPX_EPI_BIAS_CROSS) {
        const bf16x8 p = *reinterpret_cast<const bf16x8*>(a.aux2 + o2);
#pragma unroll
        for (int e = 0; e < 8; ++e) x2[e] = (float)p[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool in = n + e < a.N;
        if (kBias) v[e] += in ? a.bias[n + e] : 0.f;
        if (kAux1) x1[e] = in ? ld_as_f32<AUX_F32>(a.aux1, (uint32_t)(o1 + e)) : 0.f;
        if (EPI == MAPX_EPI_BIAS_CROSS) x2[e] = in ? (float)a.aux2[o2 + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (EPI == MAPX_EPI_BIAS_RELU) v[e] = fmaxf(v[e], 0.f);
      u[e] = v[e];
      if (EPI == MAPX_EPI_BIAS_CROSS) v[e] = x1[e] + x2[e] * v[e];
      if (EPI == MAPX_EPI_ADD) v[e] += x1[e];
      if (EPI == MAPX_EPI_RELU_MASK) v[e] = x1[e] > 0.f ? v[e] : 0.f;
    }
    if (vio) {
      if (C_F32) {
        float* c = reinterpret_cast<float*>(C) + oc;
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];          // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(C) + oc) = o;
      }
      if (EPI == MAPX_EPI_BIAS_CROSS) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)u[e];
        *reinterpret_cast<bf16x8*>(a.out2 + oo) = o;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (n + e < a.N) {
          st_from_f32<C_F32>(C, (uint32_t)(oc + e), v[e]);
          if (EPI == MAPX_EPI_BIAS_CROSS) a.out2[oo + e] = (bf16_t)u[e];
        }
      }
    }
  }
}

// The same pass for the all-aligned case (vio), without control flow between a load and its use — see
// epilogue_rows_x3_vec (gemm_x3.hip): bias once per thread (its 8 columns never change), rows in batches
// of four with every auxiliary load of the batch requested first, rows past M re-read row m0 and are not stored.
template <int EPI, bool C_F32, bool AUX_F32, int BM, int BN>
__device__ inline void epilogue_rows_h_vec(const GemmHArgs& a, void* __restrict__ C, const float* __restrict__ tile,
                                           int m0, int n0) {
  constexpr int LDT = BN + 4, CPR = BN / 8, RPI = 256 / CPR, NIT = BM / RPI, U = NIT < 4 ? NIT : 4;
  static_assert(256 % CPR == 0 && BM % RPI == 0 && NIT % U == 0, "epilogue tiling");
  constexpr bool kBias = EPI >= MAPX_EPI_BIAS && EPI <= MAPX_EPI_BIAS_CROSS;
  constexpr bool kColsum = EPI == MAPX_EPI_RELU_MASK_COLSUM;
  constexpr bool kAux1 = EPI == MAPX_EPI_BIAS_CROSS || EPI == MAPX_EPI_ADD || EPI == MAPX_EPI_RELU_MASK || kColsum;
  constexpr bool kAux2 = EPI == MAPX_EPI_BIAS_CROSS;
  const int c0 = (threadIdx.x % CPR) * 8, r0 = threadIdx.x / CPR;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // kColsum: this thread's 8 columns over its rows
  const int n = n0 + c0;
  const bool ncol = n < a.N;
  const int ns = ncol ? n : 0;
  float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (kBias) {
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + ns), b1 = *reinterpret_cast<const float4*>(a.bias + ns + 4);
    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  }
#pragma unroll
  for (int it0 = 0; it0 < NIT; it0 += U) {
    float4 t0[U], t1[U], q0[U], q1[U];
    bf16x8 h1[U], h2[U];
    bool ok[U];
    int64_t mrow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = r0 + (it0 + u) * RPI, m = m0 + row;
      ok[u] = ncol && m < a.M;
      mrow[u] = m < a.M ? m : m0;
      t0[u] = *reinterpret_cast<const float4*>(tile + row * LDT + c0);
      t1[u] = *reinterpret_cast<const float4*>(tile + row * LDT + c0 + 4);
      if (kAux1) {
        if (AUX_F32) {
          const float* p = reinterpret_cast<const float*>(a.aux1) + mrow[u] * a.ld1 + ns;
          q0[u] = *reinterpret_cast<const float4*>(p);
          q1[u] = *reinterpret_cast<const float4*>(p + 4);
        } else {
          h1[u] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(a.aux1) + mrow[u] * a.ld1 + ns);
        }
      }
      if (kAux2) h2[u] = *reinterpret_cast<const bf16x8*>(a.aux2 + mrow[u] * a.ld2 + ns);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[8] = {t0[u].x, t0[u].y, t0[u].z, t0[u].w, t1[u].x, t1[u].y, t1[u].z, t1[u].w}, w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] += b[e];
        float x1 = 0.f, x2 = 0.f;
        if (kAux1) x1 = AUX_F32 ? (e < 4 ? (&q0[u].x)[e] : (&q1[u].x)[e - 4]) : (float)h1[u][e];
        if (kAux2) x2 = (float)h2[u][e];
        if (EPI == MAPX_EPI_BIAS_RELU) v[e] = fmaxf(v[e], 0.f);
        w[e] = v[e];
        if (EPI == MAPX_EPI_BIAS_CROSS) v[e] = x1 + x2 * v[e];
        if (EPI == MAPX_EPI_ADD) v[e] += x1;
        if (EPI == MAPX_EPI_RELU_MASK || kColsum) v[e] = x1 > 0.f ? v[e] : 0.f;
        if (kColsum && ok[u]) csum[e] += C_F32 ? v[e] : (float)(bf16_t)v[e];     // the column sum of what is stored
      }
      if (ok[u]) {
        if (C_F32) {
          float* c = reinterpret_cast<float*>(C) + mrow[u] * a.ldc + n;
          *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(C) + mrow[u] * a.ldc + n) = o;
        }
        if (kAux2) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)w[e];
          *reinterpret_cast<bf16x8*>(a.out2 + mrow[u] * a.ldo2 + n) = o;
        }
      }
    }
  }
  if (kColsum) {
    // column sums of the tile's masked rows, fixed order: lanes of a wave that share their columns (xor CPR,
    // 2 CPR, ...), then the four waves through LDS behind the fp32 tile; out2 holds FP32 partial rows here,
    // one per 128-row tile (ldo2 in floats)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = CPR; d < 64; d <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) csum[e] += __shfl_xor(csum[e], d);
    float* const red = const_cast<float*>(tile) + BM * LDT;       // [4 waves][CPR][8]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(wave * CPR + lane) * 8 + e] = csum[e];
    }
    __syncthreads();
    if (threadIdx.x < CPR && ncol) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (red[threadIdx.x * 8 + e] + red[(CPR + threadIdx.x) * 8 + e]) +
                                         (red[(2 * CPR + threadIdx.x) * 8 + e] + red[(3 * CPR + threadIdx.x) * 8 + e]);
      float* const dst = reinterpret_cast<float*>(a.out2) + (int64_t)(m0 / 128) * a.ldo2 + n;
      *reinterpret_cast<float4*>(dst) = make_float4(t[0], t[1], t[2], t[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(t[4], t[5], t[6], t[7]);
    }
  }
}

// WEAVE (needs VEC and >= 2 K-steps in every slab): the K loop as hand-ordered slots, see below.
template <int WMT, int WNT, bool A_KC, bool B_KC, bool VEC, bool WEAVE>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(GemmHArgs a, int c_f32, int aux1_f32) {
  constexpr int BM = 64 * WMT, BN = 64 * WNT;
  using OpA = OperandH<BM, WMT, A_KC, VEC>;
  using OpB = OperandH<BN, WNT, B_KC, VEC>;
  // one dynamic LDS object (72-80 KB: above the 64 KB of static LDS): buffer b holds A at
  // b * kBuf and B behind it; every row starts 16-B aligned (b128 / transposing reads)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* const smem = reinterpret_cast<bf16_t*>(smem_raw);
  constexpr int kBuf = OpA::LDS_ELEMS + OpB::LDS_ELEMS;

  // XCD-aware tile order (see gemm.hip): blocks b, b+8, ... share an XCD's L2
  const int nb = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x;
  const int per = nb / 8;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int kbeg = blockIdx.y * a.k_chunk;
  const int kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
  void* __restrict__ C = c_f32 ? (void*)(reinterpret_cast<float*>(a.C) + (int64_t)blockIdx.y * a.slab_stride) : a.C;

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

  // Two register sets per operand: set (t & 1) carries tile t from its global load (issued two
  // K-steps before the tile is computed on) to its LDS store (one K-step before).  A K-step of
  // bf16 MFMAs is only 128-512 cycles — less than an L2 round trip — so the loads of tile kt+3 are
  // issued at the END of step kt, right after the registers they land in were stored to LDS, and
  // have the whole of steps kt+1 and kt+2's MFMAs to arrive.
  OpA la[2];
  OpB lb[2];
  const int nk = (a.dbg & 2) ? 0 : (kend - kbeg + kHBK - 1) / kHBK;
#define MAPX_H_LOAD(SET, t)                                              \
  do {                                                                   \
    la[SET].load(a.A, a.lda, m0, a.M, kbeg + (t) * kHBK, kend);          \
    lb[SET].load(a.B, a.ldb, n0, a.N, kbeg + (t) * kHBK, kend);          \
  } while (0)
#define MAPX_H_STORE_I(SET, buf)                                         \
  do {                                                                   \
    la[SET].template store<false>(smem + (buf) * kBuf);                  \
    lb[SET].template store<false>(smem + (buf) * kBuf + OpA::LDS_ELEMS); \
  } while (0)
#define MAPX_H_STORE(SET, buf)                                           \
  do {                                                                   \
    la[SET].template store<true>(smem + (buf) * kBuf);                   \
    lb[SET].template store<true>(smem + (buf) * kBuf + OpA::LDS_ELEMS);   \
  } while (0)
  // block-uniform: every chunk of every FULL K-step of this tile is inside the matrix (a K tail
  // only touches the last tile, which the epilogue loop below stores with its predicates)
  const bool interior = VEC && (m0 + BM <= a.M) && (n0 + BN <= a.N);
  if constexpr (WEAVE) {
    // One wave per SIMD: only the order of the wave's own instructions can keep the matrix pipe fed, and
    // hipcc's order — all fragment reads, all MFMAs, then the LDS stores and global loads — leaves it idle
    // for a third of a K-step (0.84 us per K-step of 64 where 16 MFMAs take 0.35).  As in gemm_x3.hip the
    // K-step is a sequence of slots fenced by sched_barrier(0): slot z = MFMA z + a fragment of the k16 step
    // after the next + (even slots) the LDS store of one chunk of tile kt+1 / (odd slots) the global load that
    // refills a stored chunk with tile kt+3.  Every K-step is the same body: the K remainder goes first
    // (tile 0 is the partial one, handled by the bounds-checked loads and the masked store below), edge
    // tiles clamp their out-of-range rows / columns to the first one (they only feed outputs that are
    // not stored), and past the last tile the loop re-loads the last tile.
    constexpr int kNM = 4 * WMT * WNT, kNCH = OpA::NV + OpB::NV, kFR = WMT + WNT, kPre = 2;
    static_assert(kNCH >= kPre && kNCH - kPre <= kNM / 2, "one stored chunk per even slot");
    const int wk0 = kbeg + (kend - kbeg) - kHBK * (nk - 1);        // start of tile 1
    auto wload = [&](auto& oa, auto& ob, int t) __attribute__((always_inline)) {
      const int tc = t < nk - 1 ? t : nk - 1, k0 = tc == 0 ? kbeg : wk0 + kHBK * (tc - 1);
      oa.load(a.A, a.lda, m0, a.M, k0, tc == 0 ? wk0 : kend);
      ob.load(a.B, a.ldb, n0, a.N, k0, tc == 0 ? wk0 : kend);
    };
    wload(la[0], lb[0], 0);
    wload(la[1], lb[1], 1);
    MAPX_H_STORE(0, 0);
    wload(la[0], lb[0], 2);
    __syncthreads();
    int64_t goffA[OpA::NV], goffB[OpB::NV];      // chunk's element offset from the K-step's operand base (edges clamped)
    int soffA[OpA::NV], soffB[OpB::NV];          // chunk's element offset inside an LDS buffer
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      int tr, tc;
      OpA::coords(threadIdx.x + i * 256, tr, tc);
      const bool in = (A_KC ? m0 + tr : m0 + tc) < a.M;
      goffA[i] = A_KC ? (int64_t)(in ? m0 + tr : 0) * a.lda + tc : (int64_t)tr * a.lda + (in ? m0 + tc : 0);
      soffA[i] = tr * OpA::LD + tc;
    }
#pragma unroll
    for (int i = 0; i < OpB::NV; ++i) {
      int tr, tc;
      OpB::coords(threadIdx.x + i * 256, tr, tc);
      const bool in = (B_KC ? n0 + tr : n0 + tc) < a.N;
      goffB[i] = B_KC ? (int64_t)(in ? n0 + tr : 0) * a.ldb + tc : (int64_t)tr * a.ldb + (in ? n0 + tc : 0);
      soffB[i] = OpA::LDS_ELEMS + tr * OpB::LD + tc;
    }
    // the loop's chunk registers: an ext-vector array (an array of HIP's uint4 structs, indexed from inside
    // the slot lambdas, went to scratch — a scratch store behind every global load, each with vmcnt(0))
    u32x4 rr[2][kNCH];
#pragma unroll
    for (int set = 0; set < 2; ++set) {
#pragma unroll
      for (int i = 0; i < OpA::NV; ++i) rr[set][i] = u32x4{la[set].r[i].x, la[set].r[i].y, la[set].r[i].z, la[set].r[i].w};
#pragma unroll
      for (int i = 0; i < OpB::NV; ++i)
        rr[set][OpA::NV + i] = u32x4{lb[set].r[i].x, lb[set].r[i].y, lb[set].r[i].z, lb[set].r[i].w};
    }
#define MAPX_H_WSTORE(CUR, c)                                                                          \
  do {                                                                                                 \
    constexpr bool isA_ = (c) < OpA::NV;                                                               \
    constexpr int i_ = isA_ ? (c) : (c) - OpA::NV;                                                     \
    *reinterpret_cast<u32x4*>(smem + ((CUR) ^ 1) * kBuf + (isA_ ? soffA[i_] : soffB[i_])) = rr[(CUR) ^ 1][c]; \
  } while (0)
#define MAPX_H_WLOAD(CUR, c)                                                                           \
  do {                                                                                                 \
    constexpr bool isA_ = (c) < OpA::NV;                                                               \
    constexpr int i_ = isA_ ? (c) : (c) - OpA::NV;                                                     \
    rr[(CUR) ^ 1][c] = *reinterpret_cast<const u32x4*>(isA_ ? wA + goffA[i_] : wB + goffB[i_]);        \
  } while (0)
#define MAPX_H_KSTEP_WEAVE(CUR, kt)                                                                    \
  do {                                                                                                 \
    const bf16_t* const As_cur = smem + (CUR) * kBuf;                                                  \
    const bf16_t* const Bs_cur = As_cur + OpA::LDS_ELEMS;                                              \
    bf16x8 fa[kHBK / 16][WMT], fb[kHBK / 16][WNT];   /* [k16 step][tile]: read two k16 steps ahead of use */ \
    OpA::frags(As_cur, abase, lane, 0, fa[0]);                                                         \
    OpB::frags(Bs_cur, bbase, lane, 0, fb[0]);                                                         \
    OpA::frags(As_cur, abase, lane, 1, fa[1]);                                                         \
    OpB::frags(Bs_cur, bbase, lane, 1, fb[1]);                                                         \
    const int wk_ = wk0 + kHBK * (((kt) + 3 < nk - 1 ? (kt) + 3 : nk - 1) - 1);   /* tile min(kt+3, nk-1) */ \
    const bf16_t* const wA = a.A + (int64_t)wk_ * (A_KC ? 1 : a.lda);                                  \
    const bf16_t* const wB = a.B + (int64_t)wk_ * (B_KC ? 1 : a.ldb);                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    MAPX_H_WSTORE(CUR, 0); MAPX_H_WSTORE(CUR, 1);    /* under the latency of the first fragments */   \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    unroll_seq_h([&](auto zc) __attribute__((always_inline)) {                                         \
      constexpr int z = decltype(zc)::value, s4 = z / (WMT * WNT), t = z % (WMT * WNT), i = t / WNT, j = t % WNT; \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s4][i], fb[s4][j], acc[i][j], 0, 0, 0);     \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if constexpr (s4 + 2 < kHBK / 16) {          /* the fragments of k16 step s4 + 2, spread over this step's slots */ \
        constexpr int q0 = t * kFR / (WMT * WNT), q1 = (t + 1) * kFR / (WMT * WNT);                    \
        unroll_seq_h([&](auto qc) __attribute__((always_inline)) {                                     \
          constexpr int q = q0 + decltype(qc)::value;                                                  \
          if constexpr (q < WMT) fa[s4 + 2][q] = OpA::frag1(As_cur, abase, lane, s4 + 2, q);           \
          else fb[s4 + 2][q - WMT] = OpB::frag1(Bs_cur, bbase, lane, s4 + 2, q - WMT);                 \
        }, std::make_integer_sequence<int, q1 - q0>{});                                                \
      }                                                                                                \
      if constexpr (z % 2 == 0) {                                                                      \
        constexpr int c = kPre + z / 2;                                                                \
        if constexpr (c < kNCH) MAPX_H_WSTORE(CUR, c);                                                 \
      } else {                                                                                         \
        constexpr int c0 = (z / 2) * kNCH / (kNM / 2), c1 = (z / 2 + 1) * kNCH / (kNM / 2);            \
        unroll_seq_h([&](auto cc) __attribute__((always_inline)) {                                     \
          constexpr int c = c0 + decltype(cc)::value;                                                  \
          MAPX_H_WLOAD(CUR, c);                                                                        \
        }, std::make_integer_sequence<int, c1 - c0>{});                                                \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }, std::make_integer_sequence<int, kNM>{});                                                        \
    __syncthreads();                                                                                   \
  } while (0)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      MAPX_H_KSTEP_WEAVE(0, kt);
      MAPX_H_KSTEP_WEAVE(1, kt + 1);
    }
    if (kt < nk) MAPX_H_KSTEP_WEAVE(0, kt);
#undef MAPX_H_KSTEP_WEAVE
#undef MAPX_H_WLOAD
#undef MAPX_H_WSTORE
  } else {
    if (nk > 0) MAPX_H_LOAD(0, 0);
    if (nk > 1) MAPX_H_LOAD(1, 1);
    if (nk > 0) MAPX_H_STORE(0, 0);
    if (nk > 2) MAPX_H_LOAD(0, 2);
    __syncthreads();
    // one K-step on LDS buffer CUR (= kt & 1, a literal: the loop is unrolled by two so that the
    // register sets are indexed statically — a runtime index would send them to scratch)
  #define MAPX_H_KSTEP(CUR, kt, STEADY, MASK)                                                            \
    do {                                                                                                 \
      const bf16_t* const As_cur = smem + (CUR) * kBuf;                                                  \
      const bf16_t* const Bs_cur = As_cur + OpA::LDS_ELEMS;                                              \
      bf16x8 af[2][WMT], bf[2][WNT];                                                                     \
      OpA::frags(As_cur, abase, lane, 0, af[0]);                                                         \
      OpB::frags(Bs_cur, bbase, lane, 0, bf[0]);                                                         \
      _Pragma("unroll") for (int s4 = 0; s4 < kHBK / 16; ++s4) {                                         \
        const int c = s4 & 1;                                                                            \
        if (s4 + 1 < kHBK / 16) {                                                                        \
          OpA::frags(As_cur, abase, lane, s4 + 1, af[c ^ 1]);                                            \
          OpB::frags(Bs_cur, bbase, lane, s4 + 1, bf[c ^ 1]);                                            \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < WMT; ++i)                                                  \
          _Pragma("unroll") for (int j = 0; j < WNT; ++j)                                                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c][i], bf[c][j], acc[i][j], 0, 0, 0); \
      }                                                                                                  \
      if ((STEADY) && !(MASK)) MAPX_H_STORE_I((CUR) ^ 1, (CUR) ^ 1);  /* tile kt+1: loaded two steps ago */ \
      else if ((STEADY) || (kt) + 1 < nk) MAPX_H_STORE((CUR) ^ 1, (CUR) ^ 1);                            \
      if ((STEADY) || (kt) + 3 < nk) MAPX_H_LOAD((CUR) ^ 1, (kt) + 3);                                   \
      /* (a sched_group_barrier pattern spreading LDS reads / stores / global loads between the MFMAs */ \
      /*  was measured: 59.5 vs 55.2 us at K = 4096 — the compiler's own order is the better one)       */ \
      __syncthreads();                                                                                   \
    } while (0)
    // steady state (STEADY literal: tiles kt+1 .. kt+4 exist): a branch-free body, so that the
    // compiler counts the loads in flight exactly instead of draining them at every join
    int kt = 0;
    if (interior) {
      for (; kt + 4 < nk; kt += 2) {
        MAPX_H_KSTEP(0, kt, true, false);
        MAPX_H_KSTEP(1, kt + 1, true, false);
      }
    } else {
      for (; kt + 4 < nk; kt += 2) {
        MAPX_H_KSTEP(0, kt, true, true);
        MAPX_H_KSTEP(1, kt + 1, true, true);
      }
    }
    for (; kt < nk; kt += 2) {
      MAPX_H_KSTEP(0, kt, false, true);
      if (kt + 1 < nk) MAPX_H_KSTEP(1, kt + 1, false, true);
    }
  }
#undef MAPX_H_KSTEP
#undef MAPX_H_LOAD
#undef MAPX_H_STORE
#undef MAPX_H_STORE_I

  // accumulators -> LDS as fp32 [BM][BN + 4] (the K loop's last barrier has retired every read of the
  // operand buffers); a lane writes one column of 16 rows: per store instruction two runs of 32
  // consecutive floats — conflict-free
  float* const tile = reinterpret_cast<float*>(smem_raw);
  constexpr int LDT = BN + 4;
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(abase + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2)) * LDT + bbase + 32 * j + l31] = acc[i][j][r];
  __syncthreads();
  // 16-byte row accesses for every operand of this launch? (block-uniform)
  auto al = [](const void* p, int64_t ld, int elem) {
    return p == nullptr || ((uintptr_t)p % 16 == 0 && (ld * elem) % 16 == 0);
  };
  const bool vio = a.N % 8 == 0 && al(C, a.ldc, c_f32 ? 4 : 2) && al(a.aux1, a.ld1, aux1_f32 ? 4 : 2) &&
                   al(a.aux2, a.ld2, 2) && al(a.out2, a.ldo2, 2) && al(a.bias, 0, 4);
#define MAPX_EPI_CASE(E, CF, AF)                                                          \
  do {                                                                                   \
    if (vio) epilogue_rows_h_vec<E, CF, AF, BM, BN>(a, C, tile, m0, n0);                 \
    else epilogue_rows_h<E, CF, AF, BM, BN>(a, C, tile, m0, n0, vio);                    \
  } while (0)
  if (a.dbg & 1) return;
  if (c_f32) {
    switch (a.epi) {       // fp32 outputs: logits of the heads, weight-gradient slabs
      case MAPX_EPI_BIAS: MAPX_EPI_CASE(MAPX_EPI_BIAS, true, false); break;
      case MAPX_EPI_BIAS_RELU: MAPX_EPI_CASE(MAPX_EPI_BIAS_RELU, true, false); break;
      default: MAPX_EPI_CASE(MAPX_EPI_NONE, true, false); break;
    }
  } else {
    switch (a.epi) {
      case MAPX_EPI_BIAS: MAPX_EPI_CASE(MAPX_EPI_BIAS, false, false); break;
      case MAPX_EPI_BIAS_RELU: MAPX_EPI_CASE(MAPX_EPI_BIAS_RELU, false, false); break;
      case MAPX_EPI_BIAS_CROSS: MAPX_EPI_CASE(MAPX_EPI_BIAS_CROSS, false, false); break;
      case MAPX_EPI_ADD:
        if (aux1_f32) MAPX_EPI_CASE(MAPX_EPI_ADD, false, true);
        else MAPX_EPI_CASE(MAPX_EPI_ADD, false, false);
        break;
      case MAPX_EPI_RELU_MASK: MAPX_EPI_CASE(MAPX_EPI_RELU_MASK, false, false); break;
      case MAPX_EPI_RELU_MASK_COLSUM:       /* (launcher: 128-row tiles, aligned operands) */
        if constexpr (BM == 128) epilogue_rows_h_vec<MAPX_EPI_RELU_MASK_COLSUM, false, false, BM, BN>(a, C, tile, m0, n0);
        break;
      default: MAPX_EPI_CASE(MAPX_EPI_NONE, false, false); break;
    }
  }
#undef MAPX_EPI_CASE
}

// out[i] = sum_s slabs[s][i] in slab order (deterministic split-K combine of the weight gradients)
__global__ void __launch_bounds__(256) splitk_reduce_h_kernel(const float* __restrict__ slabs, int64_t slab_stride,
                                                              int nsplit, int64_t n4, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(slabs)[i];
    for (int s = 1; s < nsplit; ++s) {
      const float4 x = reinterpret_cast<const float4*>(slabs + s * slab_stride)[i];
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}
__global__ void __launch_bounds__(256) splitk_reduce_h1_kernel(const float* __restrict__ slabs, int64_t slab_stride,
                                                               int nsplit, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slabs[s * slab_stride + i];
    out[i] = v;
  }
}

template <int WMT, int WNT, bool A_KC, bool B_KC, bool VEC, bool WEAVE>
static hipError_t launch_one_h(const GemmHArgs& a, int nsplit, int c_f32, int aux1_f32, hipStream_t stream) {
  using OpA = OperandH<64 * WMT, WMT, A_KC, VEC>;
  using OpB = OperandH<64 * WNT, WNT, B_KC, VEC>;
  constexpr size_t lds = (size_t)2 * (OpA::LDS_ELEMS + OpB::LDS_ELEMS) * sizeof(bf16_t);
  static_assert(lds >= (size_t)(64 * WMT) * (64 * WNT + 4) * sizeof(float), "the epilogue's fp32 tile must fit the operand buffers");
  auto* fn = &gemm_bf16_kernel<WMT, WNT, A_KC, B_KC, VEC, WEAVE>;
  static hipError_t raised = lds > 65536
      ? hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipSuccess;
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(fn, dim3(a.tiles_m * a.tiles_n, nsplit), dim3(256), lds, stream, a, c_f32, aux1_f32);
  return hipSuccess;
}

template <int WMT, int WNT, bool A_KC, bool B_KC>
static hipError_t launch_tile_h(const GemmHArgs& a, bool vec, int nsplit, int c_f32, int aux1_f32, hipStream_t stream) {
  // the woven K loop: vector loads and at least two K-steps in every split-K slab
  static const bool weave_on = [] { const char* e = getenv("MAPX_BF16_WEAVE"); return !e || atoi(e) != 0; }();
  const bool weave = weave_on && vec && (int64_t)a.K - (int64_t)a.k_chunk * (nsplit - 1) > kHBK;
  if (weave) return launch_one_h<WMT, WNT, A_KC, B_KC, true, true>(a, nsplit, c_f32, aux1_f32, stream);
  return vec ? launch_one_h<WMT, WNT, A_KC, B_KC, true, false>(a, nsplit, c_f32, aux1_f32, stream)
             : launch_one_h<WMT, WNT, A_KC, B_KC, false, false>(a, nsplit, c_f32, aux1_f32, stream);
}

template <bool A_KC, bool B_KC>
static hipError_t launch_layout_h(GemmHArgs& a, bool vec, int tile, int nsplit, int c_f32, int aux1_f32,
                                  hipStream_t stream) {
  if (tile == 2) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
    return launch_tile_h<2, 2, A_KC, B_KC>(a, vec, nsplit, c_f32, aux1_f32, stream);
  }
  if (tile == 1) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 63) / 64;
    return launch_tile_h<2, 1, A_KC, B_KC>(a, vec, nsplit, c_f32, aux1_f32, stream);
  }
  a.tiles_m = (a.M + 63) / 64; a.tiles_n = (a.N + 63) / 64;
  return launch_tile_h<1, 1, A_KC, B_KC>(a, vec, nsplit, c_f32, aux1_f32, stream);
}

// ---------------------------------------------------------------------------------------------
// fp32 <-> bf16 conversion of a flat array (weights shadows outside the optimizer, head gradients)
__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* __restrict__ src, int64_t n,
                                                            bf16_t* __restrict__ dst) {
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    bf16x8 o;
    o[0] = (bf16_t)a.x; o[1] = (bf16_t)a.y; o[2] = (bf16_t)a.z; o[3] = (bf16_t)a.w;
    o[4] = (bf16_t)b.x; o[5] = (bf16_t)b.y; o[6] = (bf16_t)b.z; o[7] = (bf16_t)b.w;
    reinterpret_cast<bf16x8*>(dst)[i] = o;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) dst[i] = (bf16_t)src[i];
}
__global__ void __launch_bounds__(256) cast_bf16_f32_kernel(const bf16_t* __restrict__ src, int64_t n,
                                                            float* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (float)src[i];
}

// ---------------------------------------------------------------------------------------------
// Elementwise backward steps of the bf16 path fused with the bias-gradient column sum (the bf16
// counterparts of gemm.hip's ew_colsum_kernel; activations bf16, sums fp32):
//   OP 0 (ReLU layer):   dz = y > 0 ? dy : 0                      db = colsum(dz)
//   OP 1 (cross layer):  t = g * x0 ; dx0 (+)= g * u (+ g)        db = colsum(t)     (dx0 kept in fp32)
//   OP 2 (plain):        nothing written                          db = colsum(a)
// Block = 64 column lanes x 4 row lanes over one of kColChunksH row chunks; a second launch adds the
// chunks in order (deterministic).  Scalar columns: any N, any leading dimension.
constexpr int kColChunksH = 128;
template <int OP>
__global__ void __launch_bounds__(256) ew_colsum_h_kernel(const bf16_t* __restrict__ a, int64_t lda,
                                                          const bf16_t* __restrict__ b, int64_t ldb,
                                                          const bf16_t* __restrict__ c, int64_t ldc, int M, int N,
                                                          bf16_t* __restrict__ o1, float* __restrict__ o2,
                                                          int accumulate, float* __restrict__ part) {
  const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) * 2;      // 2 columns per lane (one dword of bf16)
  const int rl = threadIdx.x >> 6;
  const int rows_per = (M + kColChunksH - 1) / kColChunksH;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
  float v0 = 0.f, v1 = 0.f;
  const bool in0 = col < N, in1 = col + 1 < N;
  if (in0) {
    for (int r = r0 + rl; r < r1; r += 4) {
      const float a0 = (float)a[(int64_t)r * lda + col], a1 = in1 ? (float)a[(int64_t)r * lda + col + 1] : 0.f;
      if (OP == 2) {
        v0 += a0; v1 += a1;
      } else if (OP == 0) {
        const float y0 = (float)b[(int64_t)r * ldb + col], y1 = in1 ? (float)b[(int64_t)r * ldb + col + 1] : 0.f;
        const float d0 = y0 > 0.f ? a0 : 0.f, d1 = y1 > 0.f ? a1 : 0.f;
        o1[(int64_t)r * N + col] = (bf16_t)d0;
        if (in1) o1[(int64_t)r * N + col + 1] = (bf16_t)d1;
        v0 += d0; v1 += d1;
      } else {
        const float x0 = (float)b[(int64_t)r * ldb + col], x1 = in1 ? (float)b[(int64_t)r * ldb + col + 1] : 0.f;
        const float u0 = (float)c[(int64_t)r * ldc + col], u1 = in1 ? (float)c[(int64_t)r * ldc + col + 1] : 0.f;
        const bf16_t t0 = (bf16_t)(a0 * x0), t1 = (bf16_t)(a1 * x1);
        float d0 = a0 * u0, d1 = a1 * u1;
        if (accumulate & 1) { d0 += o2[(int64_t)r * N + col]; if (in1) d1 += o2[(int64_t)r * N + col + 1]; }
        if (accumulate & 2) { d0 += a0; d1 += a1; }
        o1[(int64_t)r * N + col] = t0;
        o2[(int64_t)r * N + col] = d0;
        if (in1) { o1[(int64_t)r * N + col + 1] = t1; o2[(int64_t)r * N + col + 1] = d1; }
        v0 += (float)t0; v1 += (float)t1;        // db is the column sum of what the dW GEMM reads
      }
    }
  }
  __shared__ float s[4][128];
  s[rl][2 * (threadIdx.x & 63)] = v0;
  s[rl][2 * (threadIdx.x & 63) + 1] = v1;
  __syncthreads();
  if (rl == 0 && in0) {
    const int k = 2 * threadIdx.x;
    part[(int64_t)blockIdx.y * N + col] = s[0][k] + s[1][k] + s[2][k] + s[3][k];
    if (in1) part[(int64_t)blockIdx.y * N + col + 1] = s[0][k + 1] + s[1][k + 1] + s[2][k + 1] + s[3][k + 1];
  }
}
// The same three operations, 8 columns (one 16-byte bf16 chunk) per lane: N % 8 == 0, leading
// dimensions % 8 == 0, 16-byte aligned bases (the trunk's widths 368 / 400 / 624 / 1000 / 736 all are).
template <int OP>
__global__ void __launch_bounds__(256) ew_colsum_h8_kernel(const bf16_t* __restrict__ a, int64_t lda,
                                                           const bf16_t* __restrict__ b, int64_t ldb,
                                                           const bf16_t* __restrict__ c, int64_t ldc, int M, int N,
                                                           bf16_t* __restrict__ o1, float* __restrict__ o2,
                                                           int accumulate, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 8;
  const int rows_per = (M + kColChunksH - 1) / kColChunksH;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (col < N) {
#pragma unroll 2
    for (int r = r0 + rl; r < r1; r += 4) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + (int64_t)r * lda + col);
      if (OP == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)av[e];
      } else if (OP == 0) {
        const bf16x8 yv = *reinterpret_cast<const bf16x8*>(b + (int64_t)r * ldb + col);
        bf16x8 d;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          d[e] = (float)yv[e] > 0.f ? av[e] : (bf16_t)0.f;
          v[e] += (float)d[e];
        }
        *reinterpret_cast<bf16x8*>(o1 + (int64_t)r * N + col) = d;
      } else {
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(b + (int64_t)r * ldb + col);
        const bf16x8 uv = *reinterpret_cast<const bf16x8*>(c + (int64_t)r * ldc + col);
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0;
        float* dp = o2 + (int64_t)r * N + col;
        if (accumulate & 1) {
          d0 = *reinterpret_cast<const float4*>(dp);
          d1 = *reinterpret_cast<const float4*>(dp + 4);
        }
        float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = (float)av[e];
          t[e] = (bf16_t)(g * (float)xv[e]);
          d[e] += g * (float)uv[e];
          if (accumulate & 2) d[e] += g;
          v[e] += (float)t[e];
        }
        *reinterpret_cast<bf16x8*>(o1 + (int64_t)r * N + col) = t;
        *reinterpret_cast<float4*>(dp) = make_float4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<float4*>(dp + 4) = make_float4(d[4], d[5], d[6], d[7]);
      }
    }
  }
  __shared__ float s[4][64][9];          // 9: the 8 values of a lane on distinct banks from its neighbours'
#pragma unroll
  for (int e = 0; e < 8; ++e) s[rl][lane][e] = v[e];
  __syncthreads();
  if (rl == 0 && col < N) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = s[0][lane][e] + s[1][lane][e] + s[2][lane][e] + s[3][lane][e];
    float* dst = part + (int64_t)blockIdx.y * N + col;
    *reinterpret_cast<float4*>(dst) = make_float4(t[0], t[1], t[2], t[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(t[4], t[5], t[6], t[7]);
  }
}

__global__ void __launch_bounds__(256) colsum_stage2_h_kernel(const float* __restrict__ part, int N,
                                                              float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float v = 0.f;
  for (int i = 0; i < kColChunksH; ++i) v += part[(int64_t)i * N + c];
  out[c] = v;
}

__global__ void __launch_bounds__(256) relu_mask_h_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                          int64_t n, bf16_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (float)y[i] > 0.f ? dy[i] : (bf16_t)0.f;
}

}  // namespace mapx

extern "C" int mapx_gemm_bf16(int a_kc, int b_kc, int M, int N, int K, const mapx_bf16* A, int64_t lda,
                              const mapx_bf16* B, int64_t ldb, void* C, int64_t ldc, int c_f32, int epi,
                              const float* bias, const void* aux1, int64_t ld1, int aux1_f32,
                              const mapx_bf16* aux2, int64_t ld2, mapx_bf16* out2, int64_t ldo2, int nsplit,
                              int tile_hint, void* ws, size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_bf16: negative size");
  if (M == 0 || N == 0) return MAPX_OK;
  MAPX_REQUIRE(A && B && C, "gemm_bf16: null operand");
  MAPX_REQUIRE(!(a_kc == 0 && b_kc != 0), "gemm_bf16: layout (A m-contiguous, B k-contiguous) unused");
  MAPX_REQUIRE(epi >= MAPX_EPI_NONE && epi <= MAPX_EPI_RELU_MASK_COLSUM, "gemm_bf16: bad epilogue %d", epi);
  {
    const int64_t lim = (int64_t)1 << 31, rows = M > 0 ? M : 1;
    MAPX_REQUIRE(rows * ldc < lim && rows * ld1 < lim && rows * ld2 < lim && rows * ldo2 < lim,
                 "gemm_bf16: an output or auxiliary operand spans 2^31 elements or more");
  }
  if (epi >= MAPX_EPI_BIAS && epi <= MAPX_EPI_BIAS_CROSS) MAPX_REQUIRE(bias, "gemm_bf16: bias missing");
  if (epi == MAPX_EPI_BIAS_CROSS) MAPX_REQUIRE(aux1 && aux2 && out2 && !c_f32, "gemm_bf16: cross operands missing / fp32 output");
  if (epi == MAPX_EPI_ADD || epi == MAPX_EPI_RELU_MASK || epi == MAPX_EPI_RELU_MASK_COLSUM)
    MAPX_REQUIRE(aux1 && !c_f32, "gemm_bf16: aux missing, or fp32 output with ADD / RELU_MASK");
  if (epi == MAPX_EPI_RELU_MASK_COLSUM)      // out2 = fp32 partial rows [ceil(M/128)][ldo2 floats]
    MAPX_REQUIRE(out2 && N % 8 == 0 && nsplit <= 1 && (uintptr_t)C % 16 == 0 && ldc % 8 == 0 && (uintptr_t)aux1 % 16 == 0 &&
                     ld1 % 8 == 0 && (uintptr_t)out2 % 16 == 0 && ldo2 % 4 == 0,
                 "gemm_bf16: EPI_RELU_MASK_COLSUM needs N %% 8 == 0, no split-K, 16-byte aligned C / aux1 / out2");
  MAPX_REQUIRE(!aux1_f32 || epi == MAPX_EPI_ADD, "gemm_bf16: an fp32 auxiliary operand is built for EPI_ADD only");
  if (nsplit < 1) nsplit = 1;
  MAPX_REQUIRE(nsplit == 1 || (epi == MAPX_EPI_NONE && c_f32), "gemm_bf16: split-K needs EPI_NONE and an fp32 output");

  GemmHArgs g;
  g.A = reinterpret_cast<const bf16_t*>(A); g.lda = lda;
  g.B = reinterpret_cast<const bf16_t*>(B); g.ldb = ldb;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias;
  g.aux1 = aux1; g.ld1 = ld1; g.aux2 = reinterpret_cast<const bf16_t*>(aux2); g.ld2 = ld2;
  g.out2 = reinterpret_cast<bf16_t*>(out2); g.ldo2 = ldo2;
  g.k_chunk = K > 0 ? K : kHBK; g.slab_stride = 0;
  if (nsplit > 1) {
    const size_t need = (size_t)nsplit * M * N * sizeof(float);
    if (!ws || ws_bytes < need) {
      set_error("gemm_bf16: split-K workspace %zu < %zu", ws_bytes, need);
      return MAPX_EWORKSPACE;
    }
    const int kc = (int)ceil_div(ceil_div(K, nsplit), kHBK) * kHBK;
    g.k_chunk = kc;
    nsplit = (int)ceil_div(K, kc);
    g.C = ws;
    g.ldc = N;
    g.slab_stride = (int64_t)M * N;
  }
  // 16-byte chunks of 8 bf16: every chunk wholly inside or outside the matrix
  const bool vec = (lda % 8 == 0) && (ldb % 8 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) &&
                   (a_kc ? (K % 8 == 0) : (M % 8 == 0)) && (b_kc ? (K % 8 == 0) : (N % 8 == 0));
  auto blocks = [&](int bm, int bn) { return ceil_div(M, bm) * ceil_div(N, bn) * nsplit; };
  const int64_t big = blocks(128, 128);
  int tile = (big >= 160) ? 2 : 0;        // 128x128 tiles once they (with the splits) cover most of the 256 CUs
  g.dbg = tile_hint >= 0 ? (tile_hint >> 8) : 0;
  if (tile_hint >= 0) tile_hint &= 255;
  if (tile_hint >= 0 && tile_hint <= 2) tile = tile_hint;
  if (epi == MAPX_EPI_RELU_MASK_COLSUM && tile == 0) tile = 1;       // one partial row per 128-row tile
  hipError_t lerr;
  if (a_kc && b_kc) lerr = launch_layout_h<true, true>(g, vec, tile, nsplit, c_f32, aux1_f32, stream);
  else if (a_kc) lerr = launch_layout_h<true, false>(g, vec, tile, nsplit, c_f32, aux1_f32, stream);
  else lerr = launch_layout_h<false, false>(g, vec, tile, nsplit, c_f32, aux1_f32, stream);
  MAPX_HIP(lerr);
  if (nsplit > 1) {
    MAPX_REQUIRE(ldc == N, "gemm_bf16: split-K output must be dense (ldc == N)");
    const int64_t n = (int64_t)M * N;
    if (n % 4 == 0 && (uintptr_t)C % 16 == 0)
      hipLaunchKernelGGL(splitk_reduce_h_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, stream,
                         static_cast<const float*>(ws), g.slab_stride, nsplit, n / 4, static_cast<float*>(C));
    else
      hipLaunchKernelGGL(splitk_reduce_h1_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                         static_cast<const float*>(ws), g.slab_stride, nsplit, n, static_cast<float*>(C));
  }
  return check_launch("gemm_bf16");
}

extern "C" int mapx_cast_f32_bf16(const float* src, int64_t n, mapx_bf16* dst, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(n >= 0, "cast_f32_bf16: negative size");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(src && dst && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0, "cast_f32_bf16: null or unaligned pointer");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(ceil_div(n, 8), 256)), dim3(256), 0, stream, src, n,
                     reinterpret_cast<bf16_t*>(dst));
  return check_launch("cast_f32_bf16");
}

extern "C" int mapx_cast_bf16_f32(const mapx_bf16* src, int64_t n, float* dst, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(n >= 0, "cast_bf16_f32: negative size");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(src && dst, "cast_bf16_f32: null pointer");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(src), n, dst);
  return check_launch("cast_bf16_f32");
}

extern "C" size_t mapx_colsum_bf16_workspace_bytes(int N) { return (size_t)mapx::kColChunksH * N * sizeof(float); }

// 8-column lanes when every operand allows 16-byte accesses, scalar column pairs otherwise
template <int OP>
static void launch_ew_colsum_h(const mapx::bf16_t* a, int64_t lda, const mapx::bf16_t* b, int64_t ldb,
                               const mapx::bf16_t* c, int64_t ldc, int M, int N, mapx::bf16_t* o1, float* o2,
                               int accumulate, float* part, hipStream_t stream) {
  using namespace mapx;
  auto al16 = [](const void* p) { return (uintptr_t)p % 16 == 0; };
  const bool vec = N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && al16(a) && al16(b) && al16(c) &&
                   al16(o1) && al16(o2) && al16(part);
  if (vec)
    hipLaunchKernelGGL(ew_colsum_h8_kernel<OP>, dim3((N + 511) / 512, kColChunksH), dim3(256), 0, stream, a, lda, b,
                       ldb, c, ldc, M, N, o1, o2, accumulate, part);
  else
    hipLaunchKernelGGL(ew_colsum_h_kernel<OP>, dim3((N + 127) / 128, kColChunksH), dim3(256), 0, stream, a, lda, b,
                       ldb, c, ldc, M, N, o1, o2, accumulate, part);
}

static int colsum_ws_ok(const char* what, void* ws, size_t ws_bytes, int N) {
  if (!ws || ws_bytes < mapx_colsum_bf16_workspace_bytes(N)) {
    mapx::set_error("%s: workspace too small", what);
    return 0;
  }
  return 1;
}

extern "C" int mapx_colsum_bf16(const mapx_bf16* x, int64_t ld, int M, int N, float* out, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(x && M >= 0 && N > 0 && ld >= N, "colsum_bf16: bad arguments");
  if (!colsum_ws_ok("colsum_bf16", ws, ws_bytes, N)) return MAPX_EWORKSPACE;
  float* part = static_cast<float*>(ws);
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
  launch_ew_colsum_h<2>(xb, ld, xb, ld, xb, ld, M, N, nullptr, nullptr, 0, part, stream);
  if (out)     // out == NULL: the caller adds the chunk rows later (mapx_sum_tasks), as with mapx_colsum
    hipLaunchKernelGGL(colsum_stage2_h_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, out);
  return check_launch("colsum_bf16");
}

extern "C" int mapx_relu_mask_colsum_bf16(const mapx_bf16* dy, int64_t ld_dy, const mapx_bf16* y, int64_t ld_y, int M,
                                          int N, mapx_bf16* dz, float* db, void* ws, size_t ws_bytes,
                                          hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dy && y && dz && M >= 0 && N > 0 && ld_dy >= N && ld_y >= N, "relu_mask_colsum_bf16: bad arguments");
  if (!colsum_ws_ok("relu_mask_colsum_bf16", ws, ws_bytes, N)) return MAPX_EWORKSPACE;
  float* part = static_cast<float*>(ws);
  launch_ew_colsum_h<0>(reinterpret_cast<const bf16_t*>(dy), ld_dy, reinterpret_cast<const bf16_t*>(y), ld_y,
                        reinterpret_cast<const bf16_t*>(y), ld_y, M, N, reinterpret_cast<bf16_t*>(dz), nullptr, 0, part,
                        stream);
  if (db) hipLaunchKernelGGL(colsum_stage2_h_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, db);
  return check_launch("relu_mask_colsum_bf16");
}

extern "C" int mapx_cross_bwd_pre_colsum_bf16(const mapx_bf16* g, int64_t ld_g, const mapx_bf16* x0, const mapx_bf16* u,
                                              int M, int N, mapx_bf16* t, float* dx0, int accumulate, float* db,
                                              void* ws, size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(g && x0 && u && t && dx0 && M >= 0 && N > 0 && ld_g >= N, "cross_bwd_pre_colsum_bf16: bad arguments");
  if (!colsum_ws_ok("cross_bwd_pre_colsum_bf16", ws, ws_bytes, N)) return MAPX_EWORKSPACE;
  float* part = static_cast<float*>(ws);
  launch_ew_colsum_h<1>(reinterpret_cast<const bf16_t*>(g), ld_g, reinterpret_cast<const bf16_t*>(x0), (int64_t)N,
                        reinterpret_cast<const bf16_t*>(u), (int64_t)N, M, N, reinterpret_cast<bf16_t*>(t), dx0,
                        accumulate, part, stream);
  if (db) hipLaunchKernelGGL(colsum_stage2_h_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, db);
  return check_launch("cross_bwd_pre_colsum_bf16");
}

extern "C" int mapx_relu_mask_bf16(const mapx_bf16* dy, const mapx_bf16* y, int64_t n, mapx_bf16* out,
                                   hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dy && y && out && n >= 0, "relu_mask_bf16: bad arguments");
  if (n == 0) return MAPX_OK;
  hipLaunchKernelGGL(relu_mask_h_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(dy), reinterpret_cast<const bf16_t*>(y), n,
                     reinterpret_cast<bf16_t*>(out));
  return check_launch("relu_mask_bf16");
}
