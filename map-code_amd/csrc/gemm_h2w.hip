// The two-piece fp16 product (gemm_h2.hip) for the case that matters most: operand B is a WEIGHT matrix.  A weight is
// cut once per optimizer step, not once per row tile of every product that reads it: mapx_h2_weight_planes leaves its
// two fp16 pieces in HBM in the order the matrix instruction wants its B fragments — for column tile t (32 columns) and
// k16 step s one 2-KB block [piece hi | lo][lane][8 halves], lane (r, h) holding B(k = 16 s + 8 h + j, n = 32 t + r) —
// zero-padded to whole tiles and to K % 32 == 0, with the scale exponent in a header.  The product kernel then
//   * reads its B fragments straight from global memory into registers, one fully coalesced 1-KB load per fragment
//     (rows of a row-major matrix fetched 32 at a time by one load instruction run at a third of that: round 3,
//     tools/experiments/grouped_ra), two K-steps ahead of their use;
//   * stages only operand A (activations / upstream gradients, fp32 in HBM, cut between the global load and the LDS
//     store as in gemm_h2.hip) through LDS: half the ds_write_b128 traffic, half the cut's VALU work, no LDS reads
//     for B.  What bounds gemm_h2.hip's K-step is exactly that (tools/h2_ablate.sh: MFMAs alone 0.46 us, staging
//     alone 0.55, together 1.05 — the LDS store path, 79 B/clk per CU, and VALU issue beside MFMAs do not overlap
//     the matrix pipe; with operand B not staged at all: 0.73);
//   * gives each of its 4 waves all 128 rows of A and 32 columns of B (wave tile 128 x 32: the B fragments of the
//     four waves are disjoint, nothing is fetched twice).
// Forward products Y = X W^T (B(k, n) = W[n][k]) and input gradients dX = dY W (B(k, n) = W[k][n]) take the planes
// of the matching orientation; weight gradients (both operands activations) stay on gemm_h2.hip.
// Reference sites: MLPBlock layers.py:173-188, CrossNetV2 layers.py:197-201, heads models.py:74,119-124.
#include "amax.h"
#include "gemm_x3_common.h"

namespace mapx {

typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPlaneHeader = 256;        // bytes: int32 scale exponent n (pieces are those of 2^n B), then padding

// gemm_h2.hip: the cut of two pairs of floats, four asm blocks of four full-rate VALU instructions
struct CutRegs {
  float sx[4], r[4];
};
__device__ __forceinline__ void w_unit0(float x0a, float x1a, float x0b, float x1b, float s, CutRegs& c) {
  asm volatile("v_mul_f32 %0, %8, %4\n\t"
               "v_mul_f32 %1, %8, %5\n\t"
               "v_mul_f32 %2, %8, %6\n\t"
               "v_mul_f32 %3, %8, %7"
               : "=&v"(c.sx[0]), "=&v"(c.sx[1]), "=&v"(c.sx[2]), "=&v"(c.sx[3])
               : "v"(x0a), "v"(x1a), "v"(x0b), "v"(x1b), "s"(s));
}
__device__ __forceinline__ void w_unit1(CutRegs& c, uint32_t& Ha, uint32_t& Hb) {
  asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
               "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
               "v_fma_mix_f32 %2, %4, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mix_f32 %3, %6, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
               : "=&v"(Ha), "=&v"(Hb), "=&v"(c.r[0]), "=&v"(c.r[2])
               : "v"(c.sx[0]), "v"(c.sx[1]), "v"(c.sx[2]), "v"(c.sx[3]));
}
__device__ __forceinline__ void w_unit2(CutRegs& c, uint32_t Ha, uint32_t Hb, float k2048) {
  asm volatile("v_fma_mix_f32 %0, %4, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mix_f32 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "v_mul_f32 %2, %8, %2\n\t"
               "v_mul_f32 %3, %8, %3"
               : "=&v"(c.r[1]), "=&v"(c.r[3]), "+v"(c.r[0]), "+v"(c.r[2])
               : "v"(c.sx[1]), "v"(c.sx[3]), "v"(Ha), "v"(Hb), "s"(k2048));
}
__device__ __forceinline__ void w_unit3(CutRegs& c, float k2048, uint32_t& La, uint32_t& Lb) {
  asm volatile("v_mul_f32 %2, %4, %2\n\t"
               "v_mul_f32 %3, %4, %3\n\t"
               "v_cvt_pk_f16_f32 %0, %5, %2\n\t"
               "v_cvt_pk_f16_f32 %1, %6, %3"
               : "=&v"(La), "=&v"(Lb), "+v"(c.r[1]), "+v"(c.r[3])
               : "s"(k2048), "v"(c.r[0]), "v"(c.r[2]));
}
__device__ inline void w_cut8(const float (&x)[8], float s, uint4& hi, uint4& lo) {
  uint32_t H[4], L[4];
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    CutRegs c;
    w_unit0(x[2 * e], x[2 * e + 1], x[2 * e + 2], x[2 * e + 3], s, c);
    w_unit1(c, H[e], H[e + 1]);
    w_unit2(c, H[e], H[e + 1], 2048.f);
    w_unit3(c, 2048.f, L[e], L[e + 1]);
  }
  hi = make_uint4(H[0], H[1], H[2], H[3]);
  lo = make_uint4(L[0], L[1], L[2], L[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// W -> planes.  One wave per (column tile t, k16 step s) block; K padded to whole K-steps of 32, N to tiles of 32.
__global__ void __launch_bounds__(256) h2_weight_planes_kernel(const float* __restrict__ W, int64_t ldw, int N, int K,
                                                               int b_kc, const float* __restrict__ amax,
                                                               unsigned char* __restrict__ planes) {
  const int KS = ((K + 31) / 32) * 2, NT32 = (N + 31) / 32;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int n_exp = h2_scale_exp(amax);
  const float s = pow2f(n_exp);
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<int32_t*>(planes) = n_exp;
  const int64_t blocks = (int64_t)NT32 * KS;
  for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < blocks; blk += (int64_t)gridDim.x * 4) {
    const int t = (int)(blk / KS), sidx = (int)(blk % KS);
    const int n = 32 * t + r, k0 = 16 * sidx + 8 * h;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      const bool in = n < N && k < K;
      x[j] = in ? (b_kc ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]) : 0.f;
    }
    uint4 hi, lo;
    w_cut8(x, s, hi, lo);
    unsigned char* dst = planes + kPlaneHeader + blk * 2048 + lane * 16;
    *reinterpret_cast<uint4*>(dst) = hi;
    *reinterpret_cast<uint4*>(dst + 1024) = lo;
  }
}

// The same for up to kMaxPlaneTasks matrices in ONE launch (the optimizer re-cuts every registered weight behind its
// update: six launches of 5 us each sat at the end of the step).
constexpr int kMaxPlaneTasks = 16;
struct PlaneTasks {
  mapx_plane_task t[kMaxPlaneTasks];
  int64_t first[kMaxPlaneTasks + 1];          // first block of task i in the launch's numbering
  int n;
};
__global__ void __launch_bounds__(256) h2_weight_planes_multi_kernel(PlaneTasks tasks) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int64_t total = tasks.first[tasks.n];
  for (int64_t gb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); gb < total; gb += (int64_t)gridDim.x * 4) {
    int ti = 0;
    while (ti + 1 < tasks.n && gb >= tasks.first[ti + 1]) ++ti;
    const mapx_plane_task tk = tasks.t[ti];
    const int64_t blk = gb - tasks.first[ti];
    const int KS = ((tk.K + 31) / 32) * 2;
    const int n_exp = h2_scale_exp(static_cast<const float*>(tk.amax_record));
    const float s = pow2f(n_exp);
    unsigned char* const planes = static_cast<unsigned char*>(tk.planes);
    if (blk == 0 && lane == 0) *reinterpret_cast<int32_t*>(planes) = n_exp;
    const int t = (int)(blk / KS), sidx = (int)(blk % KS);
    const int n = 32 * t + r, k0 = 16 * sidx + 8 * h;
    float x[8];
    if (tk.b_kc && n < tk.N && k0 + 8 <= tk.K && tk.ldw % 4 == 0 && ((uintptr_t)tk.W & 15) == 0) {
      // a whole chunk of a k-contiguous row: two 16-byte loads (the wave reads 32 rows x 64 B)
      const float4 a = *reinterpret_cast<const float4*>(tk.W + (int64_t)n * tk.ldw + k0);
      const float4 b = *reinterpret_cast<const float4*>(tk.W + (int64_t)n * tk.ldw + k0 + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const bool in = n < tk.N && k < tk.K;
        x[j] = in ? (tk.b_kc ? tk.W[(int64_t)n * tk.ldw + k] : tk.W[(int64_t)k * tk.ldw + n]) : 0.f;
      }
    }
    uint4 hi, lo;
    w_cut8(x, s, hi, lo);
    unsigned char* dst = planes + kPlaneHeader + blk * 2048 + lane * 16;
    *reinterpret_cast<uint4*>(dst) = hi;
    *reinterpret_cast<uint4*>(dst + 1024) = lo;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Operand A: as gemm_h2.hip's k-contiguous operand (XOR-swizzled [row][32] planes in LDS)
template <int ROWS, int NT>
struct OperandA {
  static constexpr int LD = kXBK, PLANE = ROWS * LD, LDS_ELEMS = 2 * PLANE, CPR = kXBK / 8;
  static constexpr int TOTAL = ROWS * kXBK / 8, NV = TOTAL / NT;
  static_assert(TOTAL % NT == 0, "whole rounds of chunks");
  float4 r[NV][2];
  __device__ static inline void coords(int f, int& row, int& col) {
    row = f / CPR;
    col = (f % CPR) * 8;
  }
  __device__ static inline int lds_off(int row, int col) {
    return row * LD + (((col >> 3) ^ ((row >> 2) & 3)) << 3);
  }
  __device__ static inline f16x8 frag1(const f16_t* __restrict__ s, int pl, int lane, int s2, int t) {
    const int l31 = lane & 31, kh = lane >> 5;
    return *reinterpret_cast<const f16x8*>(s + pl * PLANE + lds_off(32 * t + l31, 16 * s2 + 8 * kh));
  }
};

// 128 x 128 tile by 4 waves of 128 x 32 (WMT = 4), or — products too narrow for 128 of those — 128 x 64 by 2 x 2 waves of
// 64 x 32 (WMT = 2: the two waves of a column tile read the same B fragments, from L1).  A k-contiguous [M][K]
// (lda % 4 == 0, 16-byte aligned, K % 8 == 0), B = planes.
// DEEP: the LDS pipeline one tile deeper — four buffers, tile t in buffer t & 3: K-step kt multiplies tile kt (its first
// fragments already in registers), reads the first fragments of tile kt + 1 (stored during K-step kt - 1, visible since
// the barrier that ended it) and stores tile kt + 2, and B runs three K-steps ahead — so that the one wave of a SIMD
// does not wait out an LDS round trip behind every barrier.  Same sums in the same order.
template <int WMT, bool DEEP>
__global__ void __launch_bounds__(256) gemm_f32h2w_kernel(GemmX3Args a, const unsigned char* __restrict__ planes) {
  constexpr int BM = 128, WCOLS = WMT == 4 ? 4 : 2, BN = 32 * WCOLS, NT = 256;
  static_assert(WMT == 4 || WMT == 2, "wave tile 128 x 32 or 64 x 32");
  using OpA = OperandA<BM, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16_t* const smem = reinterpret_cast<f16_t*>(smem_raw);
  constexpr int kBuf = OpA::LDS_ELEMS;

  const int na = __builtin_amdgcn_readfirstlane(h2_scale_exp(a.amax_a));
  const int nb = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int32_t*>(planes));
  const float sA = pow2f(na), k2048 = 2048.f;

  const int nb_tiles = a.tiles_m * a.tiles_n, per = nb_tiles / 8;
  int lin = blockIdx.x;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int wr = wave / WCOLS, wc = wave % WCOLS, tbase = wr * WMT;      // this wave's A row tiles tbase .. tbase + WMT - 1

  f32x16 acc[WMT], cor[WMT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = cor[i][r] = 0.f;

  const int nk = (a.K + kXBK - 1) / kXBK;                  // K-steps; the planes are zero beyond K
  const int KS = nk * 2;
  // this wave's B blocks: column tile WCOLS tn + wc, k16 steps 2 kt, 2 kt + 1: 4 KB per K-step, contiguous
  const unsigned char* const bbase =
      planes + kPlaneHeader + ((int64_t)(tn * WCOLS + wc) * KS) * 2048 + lane * 16;
  u32x4 fb[4][2][2];                                       // [set = kt & 3][k16 half][piece hi / lo]
#define MAPX_W_BLOAD(SET, t)                                                                           \
  do {                                                                                                 \
    const int tc_ = (t) < nk ? (t) : nk - 1;                                                           \
    const unsigned char* const q_ = bbase + (int64_t)tc_ * 4096;                                       \
    fb[SET][0][0] = *reinterpret_cast<const u32x4*>(q_);                                               \
    fb[SET][0][1] = *reinterpret_cast<const u32x4*>(q_ + 1024);                                        \
    fb[SET][1][0] = *reinterpret_cast<const u32x4*>(q_ + 2048);                                        \
    fb[SET][1][1] = *reinterpret_cast<const u32x4*>(q_ + 3072);                                        \
  } while (0)

  // operand A: chunk offsets (rows past M clamped to row 0, chunks past K to column 0: what they contribute meets
  // zeros of B or lands in rows the epilogue does not store), LDS offsets
  OpA la[2];
  int64_t goffA[OpA::NV];
  int soffA[OpA::NV], kcolA[OpA::NV];
#pragma unroll
  for (int i = 0; i < OpA::NV; ++i) {
    int tr, tc;
    OpA::coords(threadIdx.x + i * NT, tr, tc);
    const bool in = m0 + tr < a.M;
    goffA[i] = (int64_t)(in ? m0 + tr : 0) * a.lda;
    kcolA[i] = tc;
    soffA[i] = OpA::lds_off(tr, tc);
  }
#define MAPX_W_ALOAD(SET, i, hf, t)                                                                    \
  do {                                                                                                 \
    const int tc_ = (t) < nk ? (t) : nk - 1;                                                           \
    const int k_ = tc_ * kXBK + kcolA[i];                                                              \
    la[SET].r[i][hf] = *reinterpret_cast<const float4*>(a.A + goffA[i] + (k_ < a.K ? k_ : 0) + 4 * (hf)); \
  } while (0)

  f16x8 fa0[2][WMT];                          // DEEP: [piece hi / lo][A row tile], first k16 half of the running tile
  if constexpr (!DEEP) {
    // prologue: A tiles 0, 1 (tile 0 cut + stored), B sets 0, 1
  #pragma unroll
    for (int i = 0; i < OpA::NV; ++i) { MAPX_W_ALOAD(0, i, 0, 0); MAPX_W_ALOAD(0, i, 1, 0); }
  #pragma unroll
    for (int i = 0; i < OpA::NV; ++i) { MAPX_W_ALOAD(1, i, 0, 1); MAPX_W_ALOAD(1, i, 1, 1); }
    MAPX_W_BLOAD(0, 0);
    MAPX_W_BLOAD(1, 1);
  #pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      const float x[8] = {la[0].r[i][0].x, la[0].r[i][0].y, la[0].r[i][0].z, la[0].r[i][0].w,
                          la[0].r[i][1].x, la[0].r[i][1].y, la[0].r[i][1].z, la[0].r[i][1].w};
      uint4 hi, lo;
      w_cut8(x, sA, hi, lo);
      *reinterpret_cast<uint4*>(smem + soffA[i]) = hi;
      *reinterpret_cast<uint4*>(smem + soffA[i] + OpA::PLANE) = lo;
    }
  #pragma unroll
    for (int i = 0; i < OpA::NV; ++i) { MAPX_W_ALOAD(0, i, 0, 2); MAPX_W_ALOAD(0, i, 1, 2); }
    __syncthreads();
  } else {
    // A tiles 0, 1 cut and stored; B sets 0, 1, 2 and A tiles 2, 3 on their way, requested in the K loop's own order
    // (per K-step: a B set, then A's chunks) so that the wait counts the compiler merges at the loop head are the
    // steady state's; then the first fragments of tile 0
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) { MAPX_W_ALOAD(0, i, 0, 0); MAPX_W_ALOAD(0, i, 1, 0); }
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) { MAPX_W_ALOAD(1, i, 0, 1); MAPX_W_ALOAD(1, i, 1, 1); }
    MAPX_W_BLOAD(0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < OpA::NV; ++i) {
        const float x[8] = {la[t].r[i][0].x, la[t].r[i][0].y, la[t].r[i][0].z, la[t].r[i][0].w,
                            la[t].r[i][1].x, la[t].r[i][1].y, la[t].r[i][1].z, la[t].r[i][1].w};
        uint4 hi, lo;
        w_cut8(x, sA, hi, lo);
        *reinterpret_cast<uint4*>(smem + t * kBuf + soffA[i]) = hi;
        *reinterpret_cast<uint4*>(smem + t * kBuf + soffA[i] + OpA::PLANE) = lo;
      }
    __builtin_amdgcn_sched_barrier(0);
    MAPX_W_BLOAD(1, 1);
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) { MAPX_W_ALOAD(0, i, 0, 2); MAPX_W_ALOAD(0, i, 1, 2); }
    MAPX_W_BLOAD(2, 2);
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) { MAPX_W_ALOAD(1, i, 0, 3); MAPX_W_ALOAD(1, i, 1, 3); }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
      fa0[1][i] = OpA::frag1(smem, 1, lane, 0, tbase + i);
      fa0[0][i] = OpA::frag1(smem, 0, lane, 0, tbase + i);
    }
  }

  // K-step kt on LDS buffer CUR = kt & 1, B set BS = kt & 3.  Slots: 6 WMT MFMAs (k16 half h, A tile i, term); the cut
  // of A's tile kt + 1 (2 chunks = 16 units of 4 VALU), kUPS units per slot from slot 0; the second half's 2 WMT A
  // fragments in the first slots; then the 4 loads of B's K-step kt + 2; a chunk's two LDS stores and the two loads of
  // A's tile kt + 3 behind the slot that ends its cut.  WMT = 4: 24 slots, one unit each; WMT = 2: 12 slots, two each.
  constexpr int kNM = 6 * WMT, kU = 8 * OpA::NV, kUPS = WMT == 4 ? 1 : 2, kFR = 2 * WMT;
  constexpr int kC0 = 8 / kUPS + (WMT == 4 ? 4 : 0), kC1 = 16 / kUPS + (WMT == 4 ? 0 : 0);    // first memory slot of chunk 0 / 1
  static_assert(OpA::NV == 2 && kC1 + 4 <= kNM && kC0 + 4 <= kC1 + 4, "two chunks of A per thread; slots");
#define MAPX_W_UNIT(SET, u)                                                                            \
  do {                                                                                                 \
    constexpr int c_ = (u) / 8, pg_ = ((u) % 8) / 4, st_ = (u) % 4;                                    \
    if (st_ == 0) {                                                                                    \
      const float4 v_ = la[SET].r[c_][pg_];                                                             \
      w_unit0(v_.x, v_.y, v_.z, v_.w, sA, cr);                                                         \
    }                                                                                                  \
    if (st_ == 1) w_unit1(cr, cH[c_][2 * pg_], cH[c_][2 * pg_ + 1]);                                   \
    if (st_ == 2) w_unit2(cr, cH[c_][2 * pg_], cH[c_][2 * pg_ + 1], k2048);                            \
    if (st_ == 3) w_unit3(cr, k2048, cL[c_][2 * pg_], cL[c_][2 * pg_ + 1]);                            \
  } while (0)
#define MAPX_W_KSTEP(CUR, BS, kt)                                                                      \
  do {                                                                                                 \
    const f16_t* const As_cur = smem + (CUR) * kBuf;                                                   \
    f16_t* const As_nxt = smem + ((CUR) ^ 1) * kBuf;                                                   \
    f16x8 fa[2][2][WMT];                          /* [k16 half][piece hi / lo][A row tile] */          \
    _Pragma("unroll") for (int i = 0; i < WMT; ++i) {                                                  \
      fa[0][1][i] = OpA::frag1(As_cur, 1, lane, 0, tbase + i);                                         \
      fa[0][0][i] = OpA::frag1(As_cur, 0, lane, 0, tbase + i);                                         \
    }                                                                                                  \
    uint32_t cH[2][4], cL[2][4];                                                                       \
    CutRegs cr;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    unroll_seq([&](auto zc) __attribute__((always_inline)) {                                           \
      constexpr int z = decltype(zc)::value;                                                           \
      constexpr int h = z / (3 * WMT), i = (z % (3 * WMT)) / 3, term = z % 3;                          \
      const f16x8 bh_ = __builtin_bit_cast(f16x8, fb[BS][h][0]), bl_ = __builtin_bit_cast(f16x8, fb[BS][h][1]); \
      if (term == 0) cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[h][1][i], bh_, cor[i], 0, 0, 0); \
      if (term == 1) cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[h][0][i], bl_, cor[i], 0, 0, 0); \
      if (term == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[h][0][i], bh_, acc[i], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if constexpr (z < kFR) {                    /* second half's A fragments */                      \
        constexpr int pl = 1 - (z & 1), t = z >> 1;                                                    \
        fa[1][pl][t] = OpA::frag1(As_cur, pl, lane, 1, tbase + t);                                     \
      }                                                                                                \
      if constexpr (z * kUPS < kU) {                                                                   \
        MAPX_W_UNIT((CUR) ^ 1, z * kUPS);                                                                  \
        if constexpr (kUPS == 2) MAPX_W_UNIT((CUR) ^ 1, (z * kUPS + 1 < kU ? z * kUPS + 1 : 0));             \
      }                                                                                                \
      if constexpr (z >= kFR && z < kFR + 4) {    /* B fragments of K-step kt + 2 */                   \
        constexpr int q = z - kFR;                                                                     \
        const int tc_ = (kt) + 2 < nk ? (kt) + 2 : nk - 1;                                             \
        fb[((BS) + 2) & 3][q >> 1][q & 1] =                                                            \
            *reinterpret_cast<const u32x4*>(bbase + (int64_t)tc_ * 4096 + q * 1024);                   \
      }                                                                                                \
      if constexpr (z >= kC0 && z < kC0 + 4) {    /* chunk 0's stores (its cut ended with unit 7), then its loads */ \
        constexpr int q = z - kC0;                                                                     \
        if (q < 2) *reinterpret_cast<uint4*>(As_nxt + soffA[0] + q * OpA::PLANE) =                     \
            q == 0 ? make_uint4(cH[0][0], cH[0][1], cH[0][2], cH[0][3]) : make_uint4(cL[0][0], cL[0][1], cL[0][2], cL[0][3]); \
        else MAPX_W_ALOAD((CUR) ^ 1, 0, (q >= 2 ? q - 2 : 0), (kt) + 3);                               \
      }                                                                                                \
      if constexpr (z >= kC1 && z < kC1 + 4) {    /* chunk 1's */                                      \
        constexpr int q = z - kC1;                                                                     \
        if (q < 2) *reinterpret_cast<uint4*>(As_nxt + soffA[1] + q * OpA::PLANE) =                     \
            q == 0 ? make_uint4(cH[1][0], cH[1][1], cH[1][2], cH[1][3]) : make_uint4(cL[1][0], cL[1][1], cL[1][2], cL[1][3]); \
        else MAPX_W_ALOAD((CUR) ^ 1, 1, (q >= 2 ? q - 2 : 0), (kt) + 3);                               \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }, std::make_integer_sequence<int, kNM>{});                                                        \
    __syncthreads();                                                                                   \
  } while (0)

  // DEEP K-step kt on LDS buffer CUR = kt & 3 = B set.  Slots as above, except: the cut is that of tile kt + 2 (register
  // set kt & 1, re-loaded with tile kt + 4), stored into buffer (CUR + 2) & 3; B's loads are K-step kt + 3's (into the set
  // K-step kt - 1 used); and behind the first half's MFMAs (slots 0 .. 3 WMT - 1) their fragment registers are re-read
  // with tile kt + 1's first half from buffer (CUR + 1) & 3.
  constexpr int kN0 = 3 * WMT;
#define MAPX_WD_KSTEP(CUR, kt)                                                                         \
  do {                                                                                                 \
    const f16_t* const As_cur = smem + (CUR) * kBuf;                                                   \
    const f16_t* const As_n1 = smem + (((CUR) + 1) & 3) * kBuf;                                        \
    f16_t* const As_n2 = smem + (((CUR) + 2) & 3) * kBuf;                                              \
    f16x8 fa1[2][WMT];                            /* [piece hi / lo][A row tile]: second k16 half */   \
    uint32_t cH[2][4], cL[2][4];                                                                       \
    CutRegs cr;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    unroll_seq([&](auto zc) __attribute__((always_inline)) {                                           \
      constexpr int z = decltype(zc)::value;                                                           \
      constexpr int h = z / (3 * WMT), i = (z % (3 * WMT)) / 3, term = z % 3;                          \
      const f16x8 bh_ = __builtin_bit_cast(f16x8, fb[CUR][h][0]), bl_ = __builtin_bit_cast(f16x8, fb[CUR][h][1]);\
      const f16x8 ah_ = h == 0 ? fa0[0][i] : fa1[0][i], al_ = h == 0 ? fa0[1][i] : fa1[1][i];          \
      if (term == 0) cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, bh_, cor[i], 0, 0, 0);       \
      if (term == 1) cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bl_, cor[i], 0, 0, 0);       \
      if (term == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bh_, acc[i], 0, 0, 0);       \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if constexpr (z < kFR) {                    /* second half's A fragments */                      \
        constexpr int pl = 1 - (z & 1), t = z >> 1;                                                    \
        fa1[pl][t] = OpA::frag1(As_cur, pl, lane, 1, tbase + t);                                       \
      }                                                                                                \
      if constexpr (z >= kN0 && z < kN0 + kFR) {  /* the next tile's first half */                     \
        constexpr int q = z - kN0, pl = 1 - (q & 1), t = q >> 1;                                       \
        fa0[pl][t] = OpA::frag1(As_n1, pl, lane, 0, tbase + t);                                        \
      }                                                                                                \
      if constexpr (z * kUPS < kU) {                                                                   \
        MAPX_W_UNIT((CUR) & 1, z * kUPS);                                                              \
        if constexpr (kUPS == 2) MAPX_W_UNIT((CUR) & 1, (z * kUPS + 1 < kU ? z * kUPS + 1 : 0));       \
      }                                                                                                \
      if constexpr (z >= kFR && z < kFR + 4) {    /* B fragments of K-step kt + 3 */                   \
        constexpr int q = z - kFR;                                                                     \
        const int tc_ = (kt) + 3 < nk ? (kt) + 3 : nk - 1;                                             \
        fb[((CUR) + 3) & 3][q >> 1][q & 1] =                                                           \
            *reinterpret_cast<const u32x4*>(bbase + (int64_t)tc_ * 4096 + q * 1024);                   \
      }                                                                                                \
      if constexpr (z >= kC0 && z < kC0 + 4) {                                                         \
        constexpr int q = z - kC0;                                                                     \
        if (q < 2) *reinterpret_cast<uint4*>(As_n2 + soffA[0] + q * OpA::PLANE) =                      \
            q == 0 ? make_uint4(cH[0][0], cH[0][1], cH[0][2], cH[0][3]) : make_uint4(cL[0][0], cL[0][1], cL[0][2], cL[0][3]);\
        else MAPX_W_ALOAD((CUR) & 1, 0, (q >= 2 ? q - 2 : 0), (kt) + 4);                               \
      }                                                                                                \
      if constexpr (z >= kC1 && z < kC1 + 4) {                                                         \
        constexpr int q = z - kC1;                                                                     \
        if (q < 2) *reinterpret_cast<uint4*>(As_n2 + soffA[1] + q * OpA::PLANE) =                      \
            q == 0 ? make_uint4(cH[1][0], cH[1][1], cH[1][2], cH[1][3]) : make_uint4(cL[1][0], cL[1][1], cL[1][2], cL[1][3]);\
        else MAPX_W_ALOAD((CUR) & 1, 1, (q >= 2 ? q - 2 : 0), (kt) + 4);                               \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }, std::make_integer_sequence<int, kNM>{});                                                        \
    __syncthreads();                                                                                   \
  } while (0)

  int kt = 0;
  if constexpr (!DEEP) {
    for (; kt + 3 < nk; kt += 4) {
      MAPX_W_KSTEP(0, 0, kt);
      MAPX_W_KSTEP(1, 1, kt + 1);
      MAPX_W_KSTEP(0, 2, kt + 2);
      MAPX_W_KSTEP(1, 3, kt + 3);
    }
    if (kt < nk) { MAPX_W_KSTEP(0, 0, kt); ++kt; }
    if (kt < nk) { MAPX_W_KSTEP(1, 1, kt); ++kt; }
    if (kt < nk) { MAPX_W_KSTEP(0, 2, kt); ++kt; }
  } else {
    for (; kt + 3 < nk; kt += 4) {
      MAPX_WD_KSTEP(0, kt);
      MAPX_WD_KSTEP(1, kt + 1);
      MAPX_WD_KSTEP(2, kt + 2);
      MAPX_WD_KSTEP(3, kt + 3);
    }
    if (kt < nk) { MAPX_WD_KSTEP(0, kt); ++kt; }
    if (kt < nk) { MAPX_WD_KSTEP(1, kt); ++kt; }
    if (kt < nk) { MAPX_WD_KSTEP(2, kt); ++kt; }
  }
#undef MAPX_WD_KSTEP
#undef MAPX_W_KSTEP
#undef MAPX_W_UNIT
#undef MAPX_W_ALOAD
#undef MAPX_W_BLOAD

  float* const tile = reinterpret_cast<float*>(smem_raw);
  constexpr int LDT = BN + 4;
  const int dn = -(na + nb);
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      tile[(32 * (tbase + i) + 4 * kh + (r & 3) + 8 * (r >> 2)) * LDT + 32 * wc + l31] =
          __builtin_ldexpf(__builtin_fmaf(cor[i][r], 0x1p-11f, acc[i][r]), dn);
  __syncthreads();
  epilogue_dispatch<BM, BN, NT>(a, a.C, tile, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// The same product with EIGHT waves on a 64 x 256 tile: every wave takes all 64 rows of A and its own 32 columns of B.
// Why: the 4-wave kernel above is bound by instruction issue, not by the matrix pipe — a slot of its K-step is one
// MFMA + one 4-instruction unit of A's cut at ~8 cycles each beside MFMAs + a memory instruction, and after every
// barrier its one wave per SIMD waits out the LDS latency of the first fragments (0.78 us per K-step measured, 0.46 for
// the MFMAs alone).  Here the A tile is half as tall and shared by twice as many waves: per SIMD and K-step the same
// 24 MFMAs, but 32 VALU instructions of cut instead of 64 and 8 KB of LDS stores instead of 16; waves 0-3 stage A (one
// chunk each), waves 4-7 only multiply; and the LDS pipeline is one tile deeper — four buffers, tile t in buffer t & 3:
// K-step kt multiplies tile kt, reads the first fragments of tile kt + 1 (stored during K-step kt - 1, visible since
// the barrier that ended it) and stores tile kt + 2 — so that no MFMA waits for LDS behind a barrier.  B fragments are
// read from the planes by twice as many workgroups.
#ifndef MAPX_W8_ABLATE
#define MAPX_W8_ABLATE 0      // tools/h2_ablate.sh: 1 no B loads, 2 no cut, 4 no MFMAs, 8 no LDS stores, 16 no A loads, 32 no fragment reads, 64 no barrier
#endif
__global__ void __launch_bounds__(512) gemm_f32h2w8_kernel(GemmX3Args a, const unsigned char* __restrict__ planes) {
  constexpr int BM = 64, BN = 256, NT = 512, WMT = 2;
  using OpA = OperandA<BM, 256>;                    // 256 chunks per K-step: one per thread of waves 0-3
  static_assert(OpA::NV == 1, "one chunk per staging thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16_t* const smem = reinterpret_cast<f16_t*>(smem_raw);
  constexpr int kBuf = OpA::LDS_ELEMS;

  const int na = __builtin_amdgcn_readfirstlane(h2_scale_exp(a.amax_a));
  const int nb = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int32_t*>(planes));
  const float sA = pow2f(na), k2048 = 2048.f;

  const int nb_tiles = a.tiles_m * a.tiles_n, per = nb_tiles / 8;
  int lin = blockIdx.x;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const bool stager = wave < 4;

  f32x16 acc[WMT], cor[WMT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = cor[i][r] = 0.f;

  const int nk = (a.K + kXBK - 1) / kXBK;
  const int KS = nk * 2;
  // (the planes hold whole 128-column tiles: a wave whose 32 columns lie past them reads the last tile's — its
  // results fall in columns >= N, which the epilogue does not store)
  const int ct_last = (a.N + 127) / 128 * 4 - 1, ct = tn * 8 + wave < ct_last ? tn * 8 + wave : ct_last;
  const unsigned char* const bbase = planes + kPlaneHeader + ((int64_t)ct * KS) * 2048 + lane * 16;
  u32x4 fb[4][2][2];                                       // [set = kt & 3][k16 half][piece hi / lo]
#define MAPX_W8_BLOAD1(SET, q, t)                                                                      \
  do {                                                                                                 \
    const int tc_ = (t) < nk ? (t) : nk - 1;                                                           \
    fb[SET][(q) >> 1][(q) & 1] = *reinterpret_cast<const u32x4*>(bbase + (int64_t)tc_ * 4096 + (q) * 1024); \
  } while (0)

  // operand A (waves 0-3): this thread's chunk
  float4 ra[2][2];                                          // [register set = tile & 1][half of the chunk]
  int tr, tc;
  OpA::coords(threadIdx.x & 255, tr, tc);
  const int64_t goffA = (int64_t)((m0 + tr < a.M) ? m0 + tr : 0) * a.lda;
  const int soffA = OpA::lds_off(tr, tc);
#define MAPX_W8_ALOAD(SET, hf, t)                                                                      \
  do {                                                                                                 \
    const int tc_ = (t) < nk ? (t) : nk - 1;                                                           \
    const int k_ = tc_ * kXBK + tc;                                                                    \
    ra[SET][hf] = *reinterpret_cast<const float4*>(a.A + goffA + (k_ < a.K ? k_ : 0) + 4 * (hf));      \
  } while (0)

  // prologue: B sets 0, 1, 2; A tiles 0, 1 cut and stored, tiles 2, 3 on their way; the first fragments of tile 0
#pragma unroll
  for (int q = 0; q < 4; ++q) { MAPX_W8_BLOAD1(0, q, 0); MAPX_W8_BLOAD1(1, q, 1); MAPX_W8_BLOAD1(2, q, 2); }
  if (stager) {
    MAPX_W8_ALOAD(0, 0, 0); MAPX_W8_ALOAD(0, 1, 0);
    MAPX_W8_ALOAD(1, 0, 1); MAPX_W8_ALOAD(1, 1, 1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float x[8] = {ra[t][0].x, ra[t][0].y, ra[t][0].z, ra[t][0].w, ra[t][1].x, ra[t][1].y, ra[t][1].z, ra[t][1].w};
      uint4 hi, lo;
      w_cut8(x, sA, hi, lo);
      *reinterpret_cast<uint4*>(smem + t * kBuf + soffA) = hi;
      *reinterpret_cast<uint4*>(smem + t * kBuf + soffA + OpA::PLANE) = lo;
    }
    MAPX_W8_ALOAD(0, 0, 2); MAPX_W8_ALOAD(0, 1, 2);
    MAPX_W8_ALOAD(1, 0, 3); MAPX_W8_ALOAD(1, 1, 3);
  }
  __syncthreads();
  f16x8 fa0[2][WMT];                                       // [piece hi / lo][A row tile]: first k16 half of a tile
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
    fa0[1][i] = OpA::frag1(smem, 1, lane, 0, i);
    fa0[0][i] = OpA::frag1(smem, 0, lane, 0, i);
  }

  // K-step kt on LDS buffer CUR = kt & 3 = B set: 12 slots (k16 half h, A tile i, term).  Every wave: tile kt's second
  // half fragments in slots 0..3, tile kt + 1's first half in slots 7..10 (over the registers of tile kt's, whose
  // MFMAs are slots 0..5), the 4 loads of B's K-step kt + 3 in slots 4..7 (into the set K-step kt - 1 used: three
  // K-steps of latency — the planes of a long K do not fit L2).  Staging waves also: the 8 units of the cut of their
  // chunk of tile kt + 2 in slots 0..7, its two LDS stores in slots 8, 9, its registers re-loaded with tile kt + 4 in
  // slots 1 and 10.
  constexpr int kNM = 6 * WMT;
#define MAPX_W8_UNIT(CUR, u)                                                                           \
  do {                                                                                                 \
    constexpr int pg_ = (u) / 4, st_ = (u) % 4;                                                        \
    if (st_ == 0) {                                                                                    \
      const float4 v_ = ra[(CUR) & 1][pg_];                                                            \
      w_unit0(v_.x, v_.y, v_.z, v_.w, sA, cr);                                                         \
    }                                                                                                  \
    if (st_ == 1) w_unit1(cr, cH[2 * pg_], cH[2 * pg_ + 1]);                                           \
    if (st_ == 2) w_unit2(cr, cH[2 * pg_], cH[2 * pg_ + 1], k2048);                                    \
    if (st_ == 3) w_unit3(cr, k2048, cL[2 * pg_], cL[2 * pg_ + 1]);                                    \
  } while (0)
#define MAPX_W8_KSTEP(CUR, kt, STG)                                                                    \
  do {                                                                                                 \
    const f16_t* const As_cur = smem + (CUR) * kBuf;                                                   \
    const f16_t* const As_n1 = smem + (((CUR) + 1) & 3) * kBuf;                                        \
    f16_t* const As_n2 = smem + (((CUR) + 2) & 3) * kBuf;                                              \
    f16x8 fa1[2][WMT];                                                                                 \
    uint32_t cH[4], cL[4];                                                                             \
    CutRegs cr;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    unroll_seq([&](auto zc) __attribute__((always_inline)) {                                           \
      constexpr int z = decltype(zc)::value;                                                           \
      constexpr int h = z / 6, i = (z % 6) / 3, term = z % 3;                                          \
      const f16x8 bh_ = __builtin_bit_cast(f16x8, fb[CUR][h][0]), bl_ = __builtin_bit_cast(f16x8, fb[CUR][h][1]); \
      const f16x8 ah_ = h == 0 ? fa0[0][i] : fa1[0][i], al_ = h == 0 ? fa0[1][i] : fa1[1][i];          \
      if (!(MAPX_W8_ABLATE & 4)) {                                                                     \
        if (term == 0) cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, bh_, cor[i], 0, 0, 0);     \
        if (term == 1) cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bl_, cor[i], 0, 0, 0);     \
        if (term == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, bh_, acc[i], 0, 0, 0);     \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if constexpr (z < 4 && !(MAPX_W8_ABLATE & 32)) {                                                 \
        constexpr int pl = 1 - (z & 1), t = z >> 1;                                                    \
        fa1[pl][t] = OpA::frag1(As_cur, pl, lane, 1, t);                                               \
      }                                                                                                \
      if constexpr (z >= 7 && z < 11 && !(MAPX_W8_ABLATE & 32)) {   /* (the first half's MFMAs are slots 0..5) */ \
        constexpr int q = z - 7, pl = 1 - (q & 1), t = q >> 1;                                         \
        fa0[pl][t] = OpA::frag1(As_n1, pl, lane, 0, t);                                                \
      }                                                                                                \
      if constexpr (z >= 4 && z < 8 && !(MAPX_W8_ABLATE & 1)) MAPX_W8_BLOAD1(((CUR) + 3) & 3, z - 4, (kt) + 3); \
      if constexpr (STG) {                                                                             \
        if constexpr (z < 8 && !(MAPX_W8_ABLATE & 2)) MAPX_W8_UNIT(CUR, z);                            \
        if constexpr (z == 8 && !(MAPX_W8_ABLATE & 8)) *reinterpret_cast<uint4*>(As_n2 + soffA) = make_uint4(cH[0], cH[1], cH[2], cH[3]); \
        if constexpr (z == 9 && !(MAPX_W8_ABLATE & 8)) *reinterpret_cast<uint4*>(As_n2 + soffA + OpA::PLANE) = make_uint4(cL[0], cL[1], cL[2], cL[3]); \
        if constexpr (z == 1 && !(MAPX_W8_ABLATE & 16)) MAPX_W8_ALOAD((CUR) & 1, 0, (kt) + 4);         \
        if constexpr (z == 10 && !(MAPX_W8_ABLATE & 16)) MAPX_W8_ALOAD((CUR) & 1, 1, (kt) + 4);        \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }, std::make_integer_sequence<int, kNM>{});                                                        \
    if (!(MAPX_W8_ABLATE & 64)) __syncthreads();                                                       \
  } while (0)
#define MAPX_W8_LOOP(STG)                                                                              \
  do {                                                                                                 \
    int kt = 0;                                                                                        \
    for (; kt + 3 < nk; kt += 4) {                                                                     \
      MAPX_W8_KSTEP(0, kt, STG);                                                                       \
      MAPX_W8_KSTEP(1, kt + 1, STG);                                                                   \
      MAPX_W8_KSTEP(2, kt + 2, STG);                                                                   \
      MAPX_W8_KSTEP(3, kt + 3, STG);                                                                   \
    }                                                                                                  \
    if (kt < nk) { MAPX_W8_KSTEP(0, kt, STG); ++kt; }                                                  \
    if (kt < nk) { MAPX_W8_KSTEP(1, kt, STG); ++kt; }                                                  \
    if (kt < nk) { MAPX_W8_KSTEP(2, kt, STG); ++kt; }                                                  \
  } while (0)
  if (stager) MAPX_W8_LOOP(true);
  else MAPX_W8_LOOP(false);
#undef MAPX_W8_LOOP
#undef MAPX_W8_KSTEP
#undef MAPX_W8_UNIT
#undef MAPX_W8_ALOAD
#undef MAPX_W8_BLOAD1
  if (MAPX_W8_ABLATE & 64) __syncthreads();

  float* const tile = reinterpret_cast<float*>(smem_raw);
  constexpr int LDT = BN + 4;
  const int dn = -(na + nb);
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      tile[(32 * i + 4 * kh + (r & 3) + 8 * (r >> 2)) * LDT + 32 * wave + l31] =
          __builtin_ldexpf(__builtin_fmaf(cor[i][r], 0x1p-11f, acc[i][r]), dn);
  __syncthreads();
  epilogue_dispatch<BM, BN, NT>(a, a.C, tile, m0, n0);
}

static hipError_t launch_h2w8(const GemmX3Args& g, const void* planes, hipStream_t stream) {
  constexpr size_t ops = (size_t)4 * OperandA<64, 256>::LDS_ELEMS * sizeof(f16_t);
  constexpr size_t epi = ((size_t)64 * (256 + 4) + 4 * 512) * sizeof(float);
  constexpr size_t lds = ops > epi ? ops : epi;
  static hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32h2w8_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(gemm_f32h2w8_kernel, dim3(g.tiles_m * g.tiles_n), dim3(512), lds, stream, g,
                     static_cast<const unsigned char*>(planes));
  return hipSuccess;
}

template <int WMT, bool DEEP>
static hipError_t launch_h2w_form(const GemmX3Args& g, const void* planes, hipStream_t stream) {
  constexpr int BN = WMT == 4 ? 128 : 64;
  constexpr size_t ops = (size_t)(DEEP ? 4 : 2) * OperandA<128, 256>::LDS_ELEMS * sizeof(f16_t);
  constexpr size_t epi = ((size_t)128 * (BN + 4) + 4 * 256) * sizeof(float);
  constexpr size_t lds = ops > epi ? ops : epi;
  static hipError_t raised = lds > 65536 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32h2w_kernel<WMT, DEEP>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                         : hipSuccess;
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL((gemm_f32h2w_kernel<WMT, DEEP>), dim3(g.tiles_m * g.tiles_n), dim3(256), lds, stream, g,
                     static_cast<const unsigned char*>(planes));
  return hipSuccess;
}
template <int WMT>
static hipError_t launch_h2w(const GemmX3Args& g, const void* planes, hipStream_t stream) {
  // Opt-in (read at every call: tests and A/B runs switch it): the deeper pipeline changes nothing on the step's shapes
  // (4096 x 1000 x 1000: 33.5 vs 33.6 us, 4096 x 368 x 368: 11.9 vs 12.0; K = 4096: 115.6 -> 110.3) and doubles the
  // operands' LDS (64 KB) — RESULTS.md section 5.
  const char* const e = getenv("MAPX_GEMM_H2W_DEEP");
  const bool deep = e && atoi(e) != 0;
  return deep ? launch_h2w_form<WMT, true>(g, planes, stream) : launch_h2w_form<WMT, false>(g, planes, stream);
}

// Called by gemm_f32x3_launch (gemm_x3.hip) when the caller handed the weight operand's planes.  false: not this
// kernel's case (the caller goes on with gemm_h2.hip / gemm_x3.hip).
bool gemm_f32h2w_try(GemmX3Args& g, int a_kc, bool vec, const void* planes, int nsplit, int batch, hipStream_t stream,
                     hipError_t* err) {
  static const bool on = [] { const char* e = getenv("MAPX_GEMM_H2W"); return !e || atoi(e) != 0; }();
  static const bool narrow = [] { const char* e = getenv("MAPX_GEMM_H2W_NARROW"); return !e || atoi(e) != 0; }();
  if (!on || !planes || !a_kc || !vec || nsplit != 1 || batch != 1 || !g.amax_a) return false;
  if (g.K < 2 * kXBK || g.K % 8 != 0 || (uintptr_t)planes % 16 != 0) return false;
  // Opt-in (MAPX_GEMM_H2W8=1, read at every call so that a test can switch it): alone and repeated, the 8-wave
  // kernel is 5-9 % faster than the 4-wave one on N >= 736 (4096 x 1000 x 1000: 33.0 -> 30.9 us, K = 4096: 110 -> 101);
  // inside the step, on the step's own operands, its classes measure 3-8 % SLOWER one kernel at a time and the step
  // does not move (tools/experiments/gemm_h2/RESULTS.md, section 5).
  const char* const e8 = getenv("MAPX_GEMM_H2W8");
  const bool eight = e8 && atoi(e8) != 0;
  // 64 x 256 tiles by 8 waves: where they cover the chip and N fills its 256-column tiles (N = 368 computes 512
  // columns: slower than the 128-wide tiles, measured)
  if (eight && ceil_div(g.M, 64) * ceil_div(g.N, 256) >= 128 && ceil_div(g.N, 256) * 256 * 100 <= (int64_t)g.N * 115) {
    g.tiles_m = (int)ceil_div(g.M, 64);
    g.tiles_n = (int)ceil_div(g.N, 256);
    *err = launch_h2w8(g, planes, stream);
    return true;
  }
  g.tiles_m = (int)ceil_div(g.M, 128);
  if (ceil_div(g.M, 128) * ceil_div(g.N, 128) >= 128) {
    g.tiles_n = (int)ceil_div(g.N, 128);
    *err = launch_h2w<4>(g, planes, stream);
    return true;
  }
  if (narrow && ceil_div(g.M, 128) * ceil_div(g.N, 64) >= 128) {      // narrow products (N = 368): 128 x 64 tiles
    g.tiles_n = (int)ceil_div(g.N, 64);
    *err = launch_h2w<2>(g, planes, stream);
    return true;
  }
  return false;
}

}  // namespace mapx

extern "C" size_t mapx_h2_weight_planes_bytes(int N, int K) {
  const size_t KS = (size_t)((K + 31) / 32) * 2, NT32 = (size_t)((N + 127) / 128) * 4;     // whole 128-column tiles
  return mapx::kPlaneHeader + NT32 * KS * 2048;
}

extern "C" int mapx_h2_weight_planes(const float* W, int64_t ldw, int N, int K, int b_kc, const void* amax_record,
                                     void* planes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(W && planes && amax_record && N > 0 && K > 0, "h2_weight_planes: bad arguments");
  MAPX_REQUIRE((uintptr_t)planes % 16 == 0, "h2_weight_planes: planes must be 16-byte aligned");
  const int Np = ((N + 127) / 128) * 128;                   // tiles of the padding columns are written too (zeros)
  const int64_t blocks = (int64_t)(Np / 32) * (((K + 31) / 32) * 2);
  hipLaunchKernelGGL(h2_weight_planes_kernel, dim3(grid_for(blocks, 4, 4096)), dim3(256), 0, stream, W, ldw, N, K, b_kc,
                     static_cast<const float*>(amax_record), static_cast<unsigned char*>(planes));
  return check_launch("h2_weight_planes");
}

extern "C" int mapx_h2_weight_planes_multi(const mapx_plane_task* tasks_host, int ntasks, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(ntasks >= 0 && ntasks <= kMaxPlaneTasks, "h2_weight_planes_multi: at most %d matrices per call", kMaxPlaneTasks);
  if (ntasks == 0) return MAPX_OK;
  MAPX_REQUIRE(tasks_host, "h2_weight_planes_multi: null task list");
  PlaneTasks pt;
  memset(&pt, 0, sizeof(pt));
  pt.n = ntasks;
  int64_t run = 0;
  for (int i = 0; i < ntasks; ++i) {
    const mapx_plane_task& t = tasks_host[i];
    MAPX_REQUIRE(t.W && t.planes && t.amax_record && t.N > 0 && t.K > 0 && (uintptr_t)t.planes % 16 == 0,
                 "h2_weight_planes_multi: bad task %d", i);
    pt.t[i] = t;
    pt.first[i] = run;
    run += (int64_t)(((t.N + 127) / 128) * 4) * (((t.K + 31) / 32) * 2);
  }
  pt.first[ntasks] = run;
  hipLaunchKernelGGL(h2_weight_planes_multi_kernel, dim3(grid_for(run, 4, 4096)), dim3(256), 0, stream, pt);
  return check_launch("h2_weight_planes_multi");
}
