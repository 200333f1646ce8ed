// Shared pieces of the split-bf16 fp32 GEMM kernels (gemm_x3.hip; tools/experiments/gemm_ws/gemm_ws.hip): argument block, the cut of
// fp32 values into three bf16 planes, and the row-major epilogue pass over the fp32 tile in LDS.
#pragma once
#include "../../include/mapx_hip.h"
#include "amax.h"
#include "common.h"
#include <utility>

namespace mapx {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  // MAPX_EPI_BWD_FUSED (mapx_gemm_f32_bwd_fused): aux1 = add (optional), aux2 = x0, aux3 = u, mask = ReLU output,
  // out2 = column-sum partial rows, out3 = t, out4 = dx0;  c0 = first ReLU-masked column;  flags: 1 accumulate
  // into dx0, 2 add v itself to dx0
  const float* aux3; int64_t ld3;
  const float* mask; int64_t ldm;
  float* out3; int64_t ldo3;
  float* out4; int64_t ldo4;
  int c0, flags;
  // batched launch (gridDim.z problems of one shape: mapx_gemm_f32_batched): operands of problem z; split-K slabs
  // of problem z start z * batch_slabs floats into the workspace
  const float* Az[4];
  const float* Bz[4];
  float* Cz[4];
  int64_t batch_slabs;
  int xcd_slices;        // split-K blocks dealt so that an XCD works on ONE k-slice (see the kernel's tile order)
  // Per-tensor magnitudes (include/mapx_hip.h: mapx_gemm_scale).  amax_a / amax_b: max |x| of the operands, read by
  // the two-piece fp16 kernels (gemm_h2.hip) to place them in fp16's range; amax_c / amax_c2: where this launch
  // leaves max |.| of what it stores (C — in EPI_BWD_FUSED its columns >= c0 — and out3), for the product that
  // reads it next.  Batched launches: operands of problem z.
  const float* amax_a;
  const float* amax_b;
  unsigned long long* amax_c;
  unsigned long long* amax_c2;
  const float* amax_az[4];
  const float* amax_bz[4];
  const int32_t* epoch;  // tag of the amax records written (amax.h)
};

// what the plain mapx_gemm_f32 contract does not carry (fused backward epilogue, batched launch)
struct GemmX3Extra {
  const float* aux3; int64_t ld3;
  const float* mask; int64_t ldm;
  float* out3; int64_t ldo3;
  float* out4; int64_t ldo4;
  int c0, flags;
  int batch;
  const float* Az[4];
  const float* Bz[4];
  float* Cz[4];
  const float* amax_az[4];
  const float* amax_bz[4];
  const float* amax_a;            // mapx_gemm_scale of the (unbatched) product
  const float* amax_b;
  unsigned long long* amax_c;
  unsigned long long* amax_c2;
  const void* b_planes;
};

#define MAPX_EPI_BWD_FUSED 7        // internal to the library: reached through mapx_gemm_f32_bwd_fused only

constexpr int kXBK = 32;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression
template <class F, int... Z>
__device__ __forceinline__ void unroll_seq(F&& f, std::integer_sequence<int, Z...>) {
  (f(std::integral_constant<int, Z>{}), ...);
}

// order in which a k16 half's 3 (WMT + WNT) fragments are read = order in which the MFMAs first need them
// (tiles (0,0) (0,1) .. row-major; per tile lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi):
// what = 0: operand (0 A, 1 B), 1: plane (0 hi, 1 mid, 2 lo), 2: tile of the operand
__host__ __device__ constexpr int frag_order(int q, int what, int wnt) {
  constexpr int first[6][2] = {{0, 2}, {1, 0}, {0, 0}, {1, 2}, {0, 1}, {1, 1}};   // tile (0,0): {operand, plane}
  constexpr int later[3] = {0, 2, 1};                                            // a new row / column: planes in use order
  if (q < 6) return what == 0 ? first[q][0] : what == 1 ? first[q][1] : 0;
  const int r = (q - 6) / 3, pl = later[(q - 6) % 3];
  const bool isB = r < wnt - 1;                 // columns 1 .. WNT-1 of B come first (tiles (0, j)), then rows of A
  return what == 0 ? (isB ? 1 : 0) : what == 1 ? pl : (isB ? r + 1 : r - (wnt - 1) + 1);
}

// Cut of 8 fp32 values into three planes of 8 bf16, each piece ROUNDED to nearest (v_cvt_pk_bf16_f32)
// and the residual taken exactly in fp32: a = hi + mid + lo to within 2^-25 |a|, with pieces of either
// sign, so that what the six-term product drops has no preferred sign (a truncating cut biased every
// product toward zero by 3/4 of an fp32 ulp — measured, tools/scratch/bias_probe.py).
// Written as the instructions themselves, one asm block per piece: from `(__bf16)x` the compiler re-derives
// each residual's bf16 value with a conversion of its own (80 v_cvt_pk per 16 pairs instead of 48) and packs
// the two subtractions of a pair into v_pk_add_f32, which is slow beside MFMAs; and between two dependent
// asm statements it puts an s_nop (4 cycles of issue each).  11 VALU per pair of floats.
__device__ inline uint32_t pk_bf16(float x0, float x1) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
  return r;
}
// P = the pair rounded to bf16 (packed); r0, r1 = what it leaves of x0, x1 (exact).  5 VALU.
__device__ inline void piece(float x0, float x1, uint32_t& P, float& r0, float& r1) {
  uint32_t h0, h1;
  asm("v_cvt_pk_bf16_f32 %0, %5, %6\n\t"
      "v_lshlrev_b32 %1, 16, %0\n\t"
      "v_and_b32 %2, 0xffff0000, %0\n\t"
      "v_sub_f32 %3, %5, %1\n\t"
      "v_sub_f32 %4, %6, %2"
      : "=&v"(P), "=&v"(h0), "=&v"(h1), "=&v"(r0), "=&v"(r1)
      : "v"(x0), "v"(x1));
}
__device__ inline void cut3(const float (&x)[8], uint4& hi, uint4& mid, uint4& lo) {
  uint32_t H[4], M[4], L[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r0, r1, s0, s1;
    piece(x[2 * e], x[2 * e + 1], H[e], r0, r1);
    piece(r0, r1, M[e], s0, s1);
    L[e] = pk_bf16(s0, s1);
  }
  hi = make_uint4(H[0], H[1], H[2], H[3]);
  mid = make_uint4(M[0], M[1], M[2], M[3]);
  lo = make_uint4(L[0], L[1], L[2], L[3]);
}

// Row-major second pass of the epilogue over the fp32 tile in LDS (see gemm_bf16.hip), all operands fp32.
template <int EPI, int BM, int BN, int NT>
__device__ inline void epilogue_rows_x3(const GemmX3Args& a, float* __restrict__ C, const float* __restrict__ tile,
                                        int m0, int n0, bool vio) {
  constexpr int LDT = BN + 4;
  constexpr bool kBias = EPI >= MAPX_EPI_BIAS && EPI <= MAPX_EPI_BIAS_CROSS;
  constexpr bool kAux1 = EPI == MAPX_EPI_BIAS_CROSS || EPI == MAPX_EPI_ADD || EPI == MAPX_EPI_RELU_MASK;
  uint32_t amx = 0;                    // max |C| this thread stores (amax.h)
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
      amx = amax4(amx, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < a.N) {
          C[oc + e] = v[e];
          if (EPI == MAPX_EPI_BIAS_CROSS) a.out2[oo + e] = u[e];
          amx = max(amx, finite_abs_bits(v[e]));
        }
      }
    }
  }
  if (a.amax_c) amax_publish_block(a.amax_c, amx, a.epoch);
}

// Partial rows of an epilogue's column sums (RELU_MASK_COLSUM, BWD_FUSED): out2 has one row per 64 rows of C,
// ceil(M / 64) in all, and every one is written: a 64-row tile writes its own, a 128-row tile its sum into the first
// of its two and zeros into the second (when that row exists) — the sum over the partial rows does not depend on which
// kernel ran.
template <int BM>
__device__ inline void partial_row_store(const GemmX3Args& a, int m0, int n, float4 sum) {
  static_assert(BM == 64 || BM == 128, "tile height");
  float* const dst = a.out2 + (int64_t)(m0 / 64) * a.ldo2 + n;
  *reinterpret_cast<float4*>(dst) = sum;
  if (BM == 128 && m0 + 64 < a.M) *reinterpret_cast<float4*>(dst + a.ldo2) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// The same pass for 16-byte-aligned operands, without control flow between a load and its use: a thread's
// column never changes (NT is a multiple of BN / 4), so the bias is fetched once; rows go in batches of
// four whose auxiliary operands are all requested before the first of them is used; rows past M re-read
// row m0 and are simply not stored.  (With the loads inside `if (m < M)` blocks every iteration paid a
// full L2 round trip: 2.4 us per launch on the 128 x 128 tile with bias + ReLU.)
template <int EPI, int BM, int BN, int NT>
__device__ inline void epilogue_rows_x3_vec(const GemmX3Args& a, float* __restrict__ C, const float* __restrict__ tile,
                                            int m0, int n0) {
  constexpr int LDT = BN + 4, CPR = BN / 4, RPI = NT / CPR, NIT = BM / RPI, U = NIT < 4 ? NIT : 4;
  static_assert(NT % CPR == 0 && BM % RPI == 0 && NIT % U == 0, "epilogue tiling");
  constexpr bool kBias = EPI >= MAPX_EPI_BIAS && EPI <= MAPX_EPI_BIAS_CROSS;
  constexpr bool kColsum = EPI == MAPX_EPI_RELU_MASK_COLSUM;
  constexpr bool kAux1 = EPI == MAPX_EPI_BIAS_CROSS || EPI == MAPX_EPI_ADD || EPI == MAPX_EPI_RELU_MASK || kColsum;
  constexpr bool kAux2 = EPI == MAPX_EPI_BIAS_CROSS;
  const int c0 = (threadIdx.x % CPR) * 4, r0 = threadIdx.x / CPR;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);        // kColsum: this thread's 4 columns over its rows
  uint32_t amx = 0;                                     // max |C| this thread stores (amax.h)
  const int n = n0 + c0;
  const bool ncol = n < a.N;
  const int ns = ncol ? n : 0;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kBias) b0 = *reinterpret_cast<const float4*>(a.bias + ns);
#pragma unroll
  for (int it0 = 0; it0 < NIT; it0 += U) {
    float4 t[U], p1[U], p2[U];
    bool ok[U];
    int64_t mrow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = r0 + (it0 + u) * RPI, m = m0 + row;
      ok[u] = ncol && m < a.M;
      mrow[u] = m < a.M ? m : m0;
      t[u] = *reinterpret_cast<const float4*>(tile + row * LDT + c0);
      if (kAux1) p1[u] = *reinterpret_cast<const float4*>(a.aux1 + mrow[u] * a.ld1 + ns);
      if (kAux2) p2[u] = *reinterpret_cast<const float4*>(a.aux2 + mrow[u] * a.ld2 + ns);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[4] = {t[u].x + b0.x, t[u].y + b0.y, t[u].z + b0.z, t[u].w + b0.w};
      const float x1[4] = {p1[u].x, p1[u].y, p1[u].z, p1[u].w}, x2[4] = {p2[u].x, p2[u].y, p2[u].z, p2[u].w};
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (EPI == MAPX_EPI_BIAS_RELU) v[e] = fmaxf(v[e], 0.f);
        w[e] = v[e];
        if (EPI == MAPX_EPI_BIAS_CROSS) v[e] = x1[e] + x2[e] * v[e];
        if (EPI == MAPX_EPI_ADD) v[e] += x1[e];
        if (EPI == MAPX_EPI_RELU_MASK || kColsum) v[e] = x1[e] > 0.f ? v[e] : 0.f;
      }
      if (kColsum && ok[u]) { csum.x += v[0]; csum.y += v[1]; csum.z += v[2]; csum.w += v[3]; }
      if (ok[u]) {
        *reinterpret_cast<float4*>(C + mrow[u] * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (kAux2) *reinterpret_cast<float4*>(a.out2 + mrow[u] * a.ldo2 + n) = make_float4(w[0], w[1], w[2], w[3]);
        amx = amax4(amx, v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (a.amax_c) amax_publish_block(a.amax_c, amx, a.epoch);
  if (kColsum) {
    // column sums of the tile's (masked) rows: the RPI threads of a column group meet in LDS behind the
    // fp32 tile and are added in a fixed order; partial rows (partial_row_store), summed later
    float4* const red = reinterpret_cast<float4*>(const_cast<float*>(tile) + BM * LDT);
    red[r0 * CPR + threadIdx.x % CPR] = csum;
    __syncthreads();
    if (r0 == 0 && ncol) {
      float4 t = red[threadIdx.x % CPR];
#pragma unroll
      for (int k = 1; k < RPI; ++k) {
        const float4 q = red[k * CPR + threadIdx.x % CPR];
        t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
      }
      partial_row_store<BM>(a, m0, n, t);
    }
  }
}


// Epilogue of mapx_gemm_f32_bwd_fused (include/mapx_hip.h): the elementwise backward that follows a dX GEMM in
// DCNv2's backward pass, done on the tile while it is in LDS instead of by one more launch on the chain:
//   v = acc (+ add);   n >= c0:  v = mask > 0 ? v : 0 (ReLU backward);   n < c0:  t = v x0, dx0 (+)= v u (+ v)
//   C = v;  partial rows (partial_row_store) of the column sums of (n >= c0 ? v : t).
// Same row pass as epilogue_rows_x3_vec (a thread's four columns never change, rows in batches of four whose
// operands are all requested before the first is used); 16-byte aligned operands, N and c0 multiples of 4.
template <int BM, int BN, int NT>
__device__ inline void epilogue_bwd_fused(const GemmX3Args& a, float* __restrict__ C, const float* __restrict__ tile,
                                          int m0, int n0) {
  constexpr int LDT = BN + 4, CPR = BN / 4, RPI = NT / CPR, NIT = BM / RPI, U = NIT < 4 ? NIT : 4;
  static_assert(NT % CPR == 0 && BM % RPI == 0 && NIT % U == 0 && (BM == 128 || BM == 64), "epilogue tiling");
  const int c0 = (threadIdx.x % CPR) * 4, r0 = threadIdx.x / CPR;
  const int n = n0 + c0;
  const bool ncol = n < a.N;
  const int ns = ncol ? n : 0;
  const bool relu = ns >= a.c0;                       // this thread's four columns: ReLU-masked, or the cross layer's
  const bool has_add = a.aux1 != nullptr, accum = (a.flags & 1) != 0, plus_v = (a.flags & 2) != 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t amx_dz = 0, amx_t = 0;       // max |dz| (columns >= c0 of C) and max |t| this thread stores (amax.h)
#pragma unroll
  for (int it0 = 0; it0 < NIT; it0 += U) {
    float4 t[U], pa[U], p1[U], p2[U], p3[U];
    bool ok[U];
    int64_t mrow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = r0 + (it0 + u) * RPI, m = m0 + row;
      ok[u] = ncol && m < a.M;
      mrow[u] = m < a.M ? m : m0;
      t[u] = *reinterpret_cast<const float4*>(tile + row * LDT + c0);
      pa[u] = has_add ? *reinterpret_cast<const float4*>(a.aux1 + mrow[u] * a.ld1 + ns) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (relu) {
        p1[u] = *reinterpret_cast<const float4*>(a.mask + mrow[u] * a.ldm + ns);
      } else {
        p1[u] = *reinterpret_cast<const float4*>(a.aux2 + mrow[u] * a.ld2 + ns);                       // x0
        p2[u] = *reinterpret_cast<const float4*>(a.aux3 + mrow[u] * a.ld3 + ns);                       // u
        p3[u] = accum ? *reinterpret_cast<const float4*>(a.out4 + mrow[u] * a.ldo4 + ns) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[4] = {t[u].x + pa[u].x, t[u].y + pa[u].y, t[u].z + pa[u].z, t[u].w + pa[u].w};
      const float q1[4] = {p1[u].x, p1[u].y, p1[u].z, p1[u].w};
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = q1[e] > 0.f ? v[e] : 0.f;
        if (ok[u]) {
          csum.x += v[0]; csum.y += v[1]; csum.z += v[2]; csum.w += v[3];
          *reinterpret_cast<float4*>(C + mrow[u] * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
          amx_dz = amax4(amx_dz, v[0], v[1], v[2], v[3]);
        }
      } else {
        const float q2[4] = {p2[u].x, p2[u].y, p2[u].z, p2[u].w}, q3[4] = {p3[u].x, p3[u].y, p3[u].z, p3[u].w};
        float tt[4], d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          tt[e] = v[e] * q1[e];
          // the order of the unfused kernels (ew_colsum_kernel<1>): dx0 + g * u, then + g
          d[e] = accum ? q3[e] + v[e] * q2[e] : v[e] * q2[e];
          if (plus_v) d[e] += v[e];
        }
        if (ok[u]) {
          csum.x += tt[0]; csum.y += tt[1]; csum.z += tt[2]; csum.w += tt[3];
          *reinterpret_cast<float4*>(C + mrow[u] * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(a.out3 + mrow[u] * a.ldo3 + n) = make_float4(tt[0], tt[1], tt[2], tt[3]);
          *reinterpret_cast<float4*>(a.out4 + mrow[u] * a.ldo4 + n) = make_float4(d[0], d[1], d[2], d[3]);
          amx_t = amax4(amx_t, tt[0], tt[1], tt[2], tt[3]);
        }
      }
    }
  }
  if (a.amax_c) amax_publish_block(a.amax_c, amx_dz, a.epoch);
  if (a.amax_c2) amax_publish_block(a.amax_c2, amx_t, a.epoch);
  float4* const red = reinterpret_cast<float4*>(const_cast<float*>(tile) + BM * LDT);
  red[r0 * CPR + threadIdx.x % CPR] = csum;
  __syncthreads();
  if (r0 == 0 && ncol) {
    float4 s = red[threadIdx.x % CPR];
#pragma unroll
    for (int k = 1; k < RPI; ++k) {
      const float4 q = red[k * CPR + threadIdx.x % CPR];
      s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    partial_row_store<BM>(a, m0, n, s);
  }
}

// Second pass of every dense GEMM kernel (gemm_x3.hip, gemm_h2.hip): the fp32 tile in LDS -> C through the epilogue
template <int BM, int BN, int NT>
__device__ inline void epilogue_dispatch(const GemmX3Args& a, float* __restrict__ C, const float* __restrict__ tile,
                                         int m0, int n0) {
  auto al16 = [](const void* p, int64_t ld) { return p == nullptr || ((uintptr_t)p % 16 == 0 && ld % 4 == 0); };
  const bool vio = a.N % 4 == 0 && al16(C, a.ldc) && al16(a.aux1, a.ld1) && al16(a.aux2, a.ld2) && al16(a.out2, a.ldo2) &&
                   al16(a.bias, 0);
  if (vio) {
    switch (a.epi) {
      case MAPX_EPI_BIAS: epilogue_rows_x3_vec<MAPX_EPI_BIAS, BM, BN, NT>(a, C, tile, m0, n0); break;
      case MAPX_EPI_BIAS_RELU: epilogue_rows_x3_vec<MAPX_EPI_BIAS_RELU, BM, BN, NT>(a, C, tile, m0, n0); break;
      case MAPX_EPI_BIAS_CROSS: epilogue_rows_x3_vec<MAPX_EPI_BIAS_CROSS, BM, BN, NT>(a, C, tile, m0, n0); break;
      case MAPX_EPI_ADD: epilogue_rows_x3_vec<MAPX_EPI_ADD, BM, BN, NT>(a, C, tile, m0, n0); break;
      case MAPX_EPI_RELU_MASK: epilogue_rows_x3_vec<MAPX_EPI_RELU_MASK, BM, BN, NT>(a, C, tile, m0, n0); break;
      case MAPX_EPI_RELU_MASK_COLSUM:       // (launchers: tiles of 128 rows, or 64 x 256)
        if constexpr (BM == 128 || BN == 256) epilogue_rows_x3_vec<MAPX_EPI_RELU_MASK_COLSUM, BM, BN, NT>(a, C, tile, m0, n0);
        break;
      case MAPX_EPI_BWD_FUSED:
        if constexpr (BM == 128 || BN == 256) epilogue_bwd_fused<BM, BN, NT>(a, C, tile, m0, n0);
        break;
      default: epilogue_rows_x3_vec<MAPX_EPI_NONE, BM, BN, NT>(a, C, tile, m0, n0); break;
    }
  } else {
    switch (a.epi) {
      case MAPX_EPI_BIAS: epilogue_rows_x3<MAPX_EPI_BIAS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
      case MAPX_EPI_BIAS_RELU: epilogue_rows_x3<MAPX_EPI_BIAS_RELU, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
      case MAPX_EPI_BIAS_CROSS: epilogue_rows_x3<MAPX_EPI_BIAS_CROSS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
      case MAPX_EPI_ADD: epilogue_rows_x3<MAPX_EPI_ADD, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
      case MAPX_EPI_RELU_MASK: epilogue_rows_x3<MAPX_EPI_RELU_MASK, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
      default: epilogue_rows_x3<MAPX_EPI_NONE, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    }
  }
}

}  // namespace mapx
