// Shared pieces of the two-piece fp16 fp32-GEMM kernels (gemm_h2.hip, gemm_grouped_h2.hip): the cut of fp32 values into two
// fp16 planes and the operand class (global fp32 tile -> registers -> planes in LDS -> MFMA fragments).
#pragma once
#include "amax.h"
#include "gemm_x3_common.h"

namespace mapx {

typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// The cut of two pairs of floats (a "pair group": one float4), four asm blocks of four full-rate VALU instructions
// each (tools/micro/valu_beside_mfma.hip: v_mul_f32, v_cvt_pk_f16_f32 and v_fma_mix_f32 issue in 4 cycles beside
// MFMAs; v_fma_mixlo/hi_f16, which would scale, round and place a half in one instruction, in 8 — with them the
// K-step took the SUM of its MFMA and staging times, 1.09 us):
//   unit 0: sx = s x                                    (4 v_mul_f32; s a power of two: exact)
//   unit 1: H = { f16(sx0), f16(sx1) } for both pairs   (v_cvt_pk_f16_f32, round to nearest even)
//           r0 = sx0 - H.lo                             (v_fma_mix_f32 reading the fp16 half: exact)
//   unit 2: r1 = sx1 - H.hi;  r0 *= 2048
//   unit 3: r1 *= 2048;  L = { f16(r0), f16(r1) }
struct CutRegs {
  float sx[4], r[4];
};
__device__ __forceinline__ void h2_unit0(float x0a, float x1a, float x0b, float x1b, float s, CutRegs& c) {
  asm volatile("v_mul_f32 %0, %8, %4\n\t"
      "v_mul_f32 %1, %8, %5\n\t"
      "v_mul_f32 %2, %8, %6\n\t"
      "v_mul_f32 %3, %8, %7"
      : "=&v"(c.sx[0]), "=&v"(c.sx[1]), "=&v"(c.sx[2]), "=&v"(c.sx[3])
      : "v"(x0a), "v"(x1a), "v"(x0b), "v"(x1b), "s"(s));
}
__device__ __forceinline__ void h2_unit1(CutRegs& c, uint32_t& Ha, uint32_t& Hb) {
  asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
      "v_fma_mix_f32 %2, %4, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %3, %6, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
      : "=&v"(Ha), "=&v"(Hb), "=&v"(c.r[0]), "=&v"(c.r[2])
      : "v"(c.sx[0]), "v"(c.sx[1]), "v"(c.sx[2]), "v"(c.sx[3]));
}
__device__ __forceinline__ void h2_unit2(CutRegs& c, uint32_t Ha, uint32_t Hb, float k2048) {
  asm volatile("v_fma_mix_f32 %0, %4, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_mul_f32 %2, %8, %2\n\t"
      "v_mul_f32 %3, %8, %3"
      : "=&v"(c.r[1]), "=&v"(c.r[3]), "+v"(c.r[0]), "+v"(c.r[2])
      : "v"(c.sx[1]), "v"(c.sx[3]), "v"(Ha), "v"(Hb), "s"(k2048));
}
__device__ __forceinline__ void h2_unit3(CutRegs& c, float k2048, uint32_t& La, uint32_t& Lb) {
  asm volatile("v_mul_f32 %2, %4, %2\n\t"
      "v_mul_f32 %3, %4, %3\n\t"
      "v_cvt_pk_f16_f32 %0, %5, %2\n\t"
      "v_cvt_pk_f16_f32 %1, %6, %3"
      : "=&v"(La), "=&v"(Lb), "+v"(c.r[1]), "+v"(c.r[3])
      : "s"(k2048), "v"(c.r[0]), "v"(c.r[2]));
}
// the cut of a whole chunk (prologue tile, bounds-checked)
__device__ inline void cut2(const float (&x)[8], float s, uint4& hi, uint4& lo) {
  uint32_t H[4], L[4];
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    CutRegs c;
    h2_unit0(x[2 * e], x[2 * e + 1], x[2 * e + 2], x[2 * e + 3], s, c);
    h2_unit1(c, H[e], H[e + 1]);
    h2_unit2(c, H[e], H[e + 1], 2048.f);
    h2_unit3(c, 2048.f, L[e], L[e + 1]);
  }
  hi = make_uint4(H[0], H[1], H[2], H[3]);
  lo = make_uint4(L[0], L[1], L[2], L[3]);
}

// One operand: global fp32 tile -> registers (chunks of 8 floats) -> two fp16 planes in LDS -> fragments.  16-byte
// loads only: leading dimension % 4 == 0, aligned base, contiguous extent % 8 == 0 (a chunk is all-in or all-out).
// Plane layouts.  k-strided: [k][rows + 32] as in gemm_x3.hip (a chunk is 8 rows of one k: 16 lanes store one k's
// 256 contiguous bytes; two ds_read_b64_tr_b16 per fragment).  k-contiguous: [row][32] WITHOUT padding, the four
// 16-byte cells of a row XOR-swizzled by (row >> 2) & 3.  gemm_x3.hip's [row][32 + 8] rows serve the fragment reads
// without conflicts (16 rows at one k offset) but not the stores: four lanes store one row's 64 bytes, and four rows
// of 80 bytes wrap around the 256 bytes of the banks — two-way conflicts on every ds_write_b128, which measured as
// THE cost of this K-step (tools/h2_ablate.sh: 1.09 us with the stores, 0.50 without, MFMAs alone 0.49).  Swizzled,
// four rows x 64 bytes are 256 consecutive bytes for the stores, and 16 rows at one k offset fall into 16 different
// cells for the reads.
template <int ROWS, int T, bool KC, int NT>
struct OperandH2 {
  static constexpr int LD = KC ? kXBK : ROWS + 32;
  static constexpr int PLANE = KC ? ROWS * LD : kXBK * LD;
  static constexpr int LDS_ELEMS = 2 * PLANE;
  static constexpr int CPR = KC ? kXBK / 8 : ROWS / 8;
  static constexpr int TOTAL = ROWS * kXBK / 8;
  static constexpr int NV = (TOTAL + NT - 1) / NT;                 // chunks per thread per tile
  static constexpr bool PARTIAL = TOTAL % NT != 0;                 // the last round is for the first waves only
  static_assert(!PARTIAL || TOTAL % 64 == 0, "a partial round must end on a wave boundary");
  float4 r[NV][2];
  bool ok[NV];

  __device__ static inline void coords(int f, int& row, int& col) {
    row = f / CPR;
    col = (f % CPR) * 8;
  }
  // element offset of chunk (row, col) inside a plane
  __device__ static inline int lds_off(int row, int col) {
    return KC ? row * LD + (((col >> 3) ^ ((row >> 2) & 3)) << 3) : row * LD + col;
  }
  __device__ inline void load(const float* __restrict__ g, int64_t ld, int row0, int nrows, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tr, tc;
      if (PARTIAL && threadIdx.x + i * NT >= TOTAL) break;
      coords(threadIdx.x + i * NT, tr, tc);
      const int gr = (KC ? row0 : k0) + tr, gc = (KC ? k0 : row0) + tc;
      const int rlim = KC ? nrows : kend, clim = KC ? kend : nrows;
      const bool rok = gr < rlim;
      ok[i] = rok && gc < clim;
      const float* q = g + (int64_t)(rok ? gr : 0) * ld + (ok[i] ? gc : 0);
      r[i][0] = *reinterpret_cast<const float4*>(q);
      r[i][1] = *reinterpret_cast<const float4*>(q + 4);
    }
  }
  __device__ inline void store_masked(f16_t* __restrict__ s, float scale) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tr, tc;
      if (PARTIAL && threadIdx.x + i * NT >= TOTAL) break;
      coords(threadIdx.x + i * NT, tr, tc);
      const bool keep = ok[i];
      const float x[8] = {keep ? r[i][0].x : 0.f, keep ? r[i][0].y : 0.f, keep ? r[i][0].z : 0.f, keep ? r[i][0].w : 0.f,
                          keep ? r[i][1].x : 0.f, keep ? r[i][1].y : 0.f, keep ? r[i][1].z : 0.f, keep ? r[i][1].w : 0.f};
      uint4 hi, lo;
      cut2(x, scale, hi, lo);
      f16_t* d = s + lds_off(tr, tc);
      *reinterpret_cast<uint4*>(d) = hi;
      *reinterpret_cast<uint4*>(d + PLANE) = lo;
    }
  }
  // fragment of k16-step s2 (k = 16 s2 + 8 (lane >> 5) + j) of plane `pl` for the wave's tile t
  __device__ static inline f16x8 frag1(const f16_t* __restrict__ s, int pl, int base, int lane, int s2, int t) {
    const int l31 = lane & 31, kh = lane >> 5;
    const f16_t* sp = s + pl * PLANE;
    if (KC) return *reinterpret_cast<const f16x8*>(sp + lds_off(base + 32 * t + l31, 16 * s2 + 8 * kh));
    const int q = (lane >> 2) & 3, p = lane & 3, half = (lane >> 4) & 1;
    const f16_t* a0 = sp + (16 * s2 + 8 * kh + q) * LD + base + 32 * t + 16 * half + 4 * p;
    typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a0));
    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a0 + 4 * LD));
    return __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

}  // namespace mapx
