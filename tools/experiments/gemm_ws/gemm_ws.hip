// EXPERIMENT (round 3; not part of libmapx_hip.so — built by tools/experiments/gemm_ws/Makefile, results in
// RESULTS.md, summary in DESIGN.md section 4.1): fp32 GEMM on the bf16 matrix cores, wave-specialised: the same six-product arithmetic as gemm_x3.hip
// (a = hi + mid + lo, three bf16 planes per operand; leading products and corrections in two fp32
// accumulators), with the work of a K-step split between two kinds of wave instead of woven into one
// instruction stream:
//   waves 0-3  (one per SIMD)  CONSUMERS: fragment reads + the 48 MFMAs of a 64 x 64 wave tile, nothing else;
//   waves 4-7  (one per SIMD)  LOADERS:   bring the next tiles into an LDS ring.
// Two operand sources:
//   MODE 0  operands are PLANES in global memory (bf16 [3][rows][ld], written by whoever produced the
//           tensor: GEMM epilogues, the gather, the optimizer): a loader wave only issues LDS-DMA
//           (global_load_lds_dwordx4, 1 KiB per instruction) — no VALU, no ds_write, no staging registers;
//           3 stages of 48 KB, two tiles in flight.
//   MODE 1  operands are fp32: the loader waves load, cut (11 VALU per pair, in the shadow of their SIMD
//           partner's MFMAs) and store the planes; 2 stages.
// LDS image of a stage (both modes, unpadded, 16-byte slots XOR-swizzled so that fragment reads are
// conflict-free and a DMA instruction's 64 lanes x 16 B land contiguously):
//   k-contiguous operand, per plane  [128 rows][4 slots]:  slot s of row r holds k-chunk s ^ ((r >> 2) & 3)
//   k-strided operand, per plane     [32 k][16 slots]:     slot s of k-row kk holds column chunk s ^ (4 (kk & 3))
// Reference sites of the products: CrossNetV2 layers.py:197-201, MLPBlock layers.py:173-188, feat_encoder /
// pred_rfd / fc_out models.py:74,119-124,304 and their backward.
#include "../../../map-code_amd/csrc/gemm_x3_common.h"

namespace mapx {

// The same cut as ONE instruction block, stage by stage across the four pairs (4 conversions, 8 shift / mask,
// 8 subtractions, ...: 44 VALU), so that no instruction waits for the one before it.  For a wave that does
// nothing but cut (the loader waves of gemm_ws.hip): cut3's per-pair blocks are dependent chains, which a wave
// interleaved with MFMAs hides and a VALU-only wave pays in full (measured: 2000 -> cycles per K-step of 4 chunks).
__device__ inline void cut3_wide(const float (&xin)[8], uint4& hi, uint4& mid, uint4& lo) {
  float x0 = xin[0], x1 = xin[1], x2 = xin[2], x3 = xin[3], x4 = xin[4], x5 = xin[5], x6 = xin[6], x7 = xin[7];
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
  asm("v_cvt_pk_bf16_f32 %0, %12, %13\n\t"
      "v_cvt_pk_bf16_f32 %1, %14, %15\n\t"
      "v_cvt_pk_bf16_f32 %2, %16, %17\n\t"
      "v_cvt_pk_bf16_f32 %3, %18, %19\n\t"
      "v_lshlrev_b32 %20, 16, %0\n\t"
      "v_and_b32 %21, 0xffff0000, %0\n\t"
      "v_lshlrev_b32 %22, 16, %1\n\t"
      "v_and_b32 %23, 0xffff0000, %1\n\t"
      "v_lshlrev_b32 %24, 16, %2\n\t"
      "v_and_b32 %25, 0xffff0000, %2\n\t"
      "v_lshlrev_b32 %26, 16, %3\n\t"
      "v_and_b32 %27, 0xffff0000, %3\n\t"
      "v_sub_f32 %12, %12, %20\n\t"
      "v_sub_f32 %13, %13, %21\n\t"
      "v_sub_f32 %14, %14, %22\n\t"
      "v_sub_f32 %15, %15, %23\n\t"
      "v_sub_f32 %16, %16, %24\n\t"
      "v_sub_f32 %17, %17, %25\n\t"
      "v_sub_f32 %18, %18, %26\n\t"
      "v_sub_f32 %19, %19, %27\n\t"
      "v_cvt_pk_bf16_f32 %4, %12, %13\n\t"
      "v_cvt_pk_bf16_f32 %5, %14, %15\n\t"
      "v_cvt_pk_bf16_f32 %6, %16, %17\n\t"
      "v_cvt_pk_bf16_f32 %7, %18, %19\n\t"
      "v_lshlrev_b32 %20, 16, %4\n\t"
      "v_and_b32 %21, 0xffff0000, %4\n\t"
      "v_lshlrev_b32 %22, 16, %5\n\t"
      "v_and_b32 %23, 0xffff0000, %5\n\t"
      "v_lshlrev_b32 %24, 16, %6\n\t"
      "v_and_b32 %25, 0xffff0000, %6\n\t"
      "v_lshlrev_b32 %26, 16, %7\n\t"
      "v_and_b32 %27, 0xffff0000, %7\n\t"
      "v_sub_f32 %12, %12, %20\n\t"
      "v_sub_f32 %13, %13, %21\n\t"
      "v_sub_f32 %14, %14, %22\n\t"
      "v_sub_f32 %15, %15, %23\n\t"
      "v_sub_f32 %16, %16, %24\n\t"
      "v_sub_f32 %17, %17, %25\n\t"
      "v_sub_f32 %18, %18, %26\n\t"
      "v_sub_f32 %19, %19, %27\n\t"
      "v_cvt_pk_bf16_f32 %8, %12, %13\n\t"
      "v_cvt_pk_bf16_f32 %9, %14, %15\n\t"
      "v_cvt_pk_bf16_f32 %10, %16, %17\n\t"
      "v_cvt_pk_bf16_f32 %11, %18, %19"
      : "=&v"(hi.x), "=&v"(hi.y), "=&v"(hi.z), "=&v"(hi.w), "=&v"(mid.x), "=&v"(mid.y), "=&v"(mid.z), "=&v"(mid.w),
        "=&v"(lo.x), "=&v"(lo.y), "=&v"(lo.z), "=&v"(lo.w), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4),
        "+v"(x5), "+v"(x6), "+v"(x7), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6),
        "=&v"(t7));
}


constexpr int kWsPlane = 128 * 32 * 2;        // bytes: one plane of one operand, one K-step
constexpr int kWsOp = 3 * kWsPlane;           // 24 KB
constexpr int kWsStage = 2 * kWsOp;           // 48 KB: A planes, then B planes

struct GemmWsArgs {
  GemmX3Args e;                               // sizes, fp32 operands (MODE 1), C, epilogue operands, split-K
  const bf16_t* Ap; int64_t ldap, pa;         // MODE 0: plane p of A at Ap + p * pa, leading dimension ldap (elements)
  const bf16_t* Bp; int64_t ldbp, pb;
  const bf16_t* zeros;                        // >= 16 bytes of zeros (K tail of the DMA path)
  unsigned long long* stamps;                 // diagnostic builds (-DMAPX_WS_STAMP): [block][16]: {cycle, realtime} x 4 points, then accumulated cycles
};

// Diagnostic builds only: -DMAPX_WS_STAMP records s_memtime / s_memrealtime of consumer wave 0 at four points
// (kernel entry, first tile ready, K loop done, epilogue done); -DMAPX_WS_ABLATE=bits compiles phases out of
// the K loop (1: no DMA / loads + cut + stores, 2: no MFMAs, 4: no fragment reads; results are then wrong).
#ifdef MAPX_WS_ABLATE
constexpr int kWsDbg = MAPX_WS_ABLATE;
#else
constexpr int kWsDbg = 0;
#endif
#ifdef MAPX_WS_STAMP
#define MAPX_WS_STAMP_AT(i)                                                                    \
  do {                                                                                         \
    if (g.stamps && threadIdx.x == 0) {                                                        \
      g.stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime();         \
      g.stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                          \
  } while (0)
// accumulated cycles of a code span: T0 before, ACC after (slot 8 + i of the block's 16 words)
#define MAPX_WS_T0() const unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define MAPX_WS_ACC(var) var += __builtin_amdgcn_s_memtime() - t0_
#define MAPX_WS_PUT(i, var)                                                                    \
  do {                                                                                         \
    if (g.stamps && lane == 0) g.stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + 8 + (i)] = (var); \
  } while (0)
#else
#define MAPX_WS_STAMP_AT(i) do {} while (0)
#define MAPX_WS_T0() do {} while (0)
#define MAPX_WS_ACC(var) do {} while (0)
#define MAPX_WS_PUT(i, var) do {} while (0)
#endif

__device__ inline void ws_dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// byte offset inside a plane of the 16-byte slot that holds chunk c of stored row r
template <bool KC>
__device__ inline int ws_slot(int r, int c) {
  return KC ? r * 64 + ((c ^ ((r >> 2) & 3)) << 4) : r * 256 + ((c ^ ((r & 3) << 2)) << 4);
}

template <bool A_KC, bool B_KC, int MODE, int PRIO>
__global__ void __launch_bounds__(512) gemm_ws_kernel(GemmWsArgs g) {
  const GemmX3Args& a = g.e;
  constexpr int BM = 128, BN = 128, NT = 512, S = MODE == 1 ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int nb = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x;
  const int per = nb / 8;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order (see gemm.hip)
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.y * a.k_chunk;
  const int kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
  const int nk = (kend - kbeg + kXBK - 1) / kXBK;
  // the K range's remainder goes first: tile 0 = [kbeg, wk0), tiles t >= 1 = [wk0 + 32 (t - 1), + 32)
  const int wk0 = kend - kXBK * (nk - 1);
  float* __restrict__ C = a.C + (int64_t)blockIdx.y * a.slab_stride;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, kh = lane >> 5;

  f32x16 acc[2][2], cor[2][2];
  int abase = 0, bbase = 0;
  MAPX_WS_STAMP_AT(0);

  if (wave >= 4) {
    // ------------------------------------------------------------------ loaders
    const int w = wave - 4;
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
    if constexpr (MODE == 0) {
      // instruction j of an (operand, plane) covers stored rows 16 j .. 16 j + 15 (k-contiguous) or k-rows
      // 4 j .. 4 j + 3 (k-strided); this wave issues j = w and j = w + 4 of all six (operand, plane) pairs.
      // src[x][jj]: element offset of this lane's 16 bytes inside plane 0 at K offset 0; for a k-strided
      // operand the K offset adds k * ld, for a k-contiguous one it adds k.
      int64_t srcA[2], srcB[2];
      int chA, chB;          // k-contiguous: this lane's k-chunk (0..3); k-strided: its k-row inside the instruction
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = w + 4 * jj;
        if (A_KC) {
          const int rin = lane >> 2, c = (lane & 3) ^ ((rin >> 2) & 3);
          int row = m0 + 16 * j + rin;
          row = row < a.M ? row : a.M - 1;
          srcA[jj] = (int64_t)row * g.ldap + 8 * c;
          chA = c;
        } else {
          const int kin = lane >> 4, c = (lane & 15) ^ (kin << 2);
          int col = m0 + 8 * c;
          col = col + 8 <= a.M ? col : 0;
          srcA[jj] = (int64_t)(4 * j + kin) * g.ldap + col;
          chA = kin;
        }
        if (B_KC) {
          const int rin = lane >> 2, c = (lane & 3) ^ ((rin >> 2) & 3);
          int row = n0 + 16 * j + rin;
          row = row < a.N ? row : a.N - 1;
          srcB[jj] = (int64_t)row * g.ldbp + 8 * c;
          chB = c;
        } else {
          const int kin = lane >> 4, c = (lane & 15) ^ (kin << 2);
          int col = n0 + 8 * c;
          col = col + 8 <= a.N ? col : 0;
          srcB[jj] = (int64_t)(4 * j + kin) * g.ldbp + col;
          chB = kin;
        }
      }
      // byte offsets (32-bit, from the operand's plane 0 at the K-step's first k) of this lane's 16 bytes for the
      // wave's 12 DMA instructions: the loop adds nothing per lane — the K offset advances a scalar base
      uint32_t offA[2][3], offB[2][3];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          offA[jj][p] = (uint32_t)((srcA[jj] + p * g.pa) * 2);
          offB[jj][p] = (uint32_t)((srcB[jj] + p * g.pb) * 2);
        }
      // tile 0 (the K remainder, klen valid k): lanes past the range fetch zeros
      auto issue_first = [&](int klen) __attribute__((always_inline)) {
        const char* const ba = reinterpret_cast<const char*>(g.Ap) + (A_KC ? (int64_t)kbeg : (int64_t)kbeg * g.ldap) * 2;
        const char* const bb = reinterpret_cast<const char*>(g.Bp) + (B_KC ? (int64_t)kbeg : (int64_t)kbeg * g.ldbp) * 2;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = w + 4 * jj;
          const bool okA = A_KC ? 8 * chA + 8 <= klen : 4 * j + chA < klen;
          const bool okB = B_KC ? 8 * chB + 8 <= klen : 4 * j + chB < klen;
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            ws_dma16(okA ? (const void*)(ba + offA[jj][p]) : (const void*)g.zeros, smem + p * kWsPlane + j * 1024);
            ws_dma16(okB ? (const void*)(bb + offB[jj][p]) : (const void*)g.zeros, smem + kWsOp + p * kWsPlane + j * 1024);
          }
        }
      };
      // a full tile t >= 1 into `stage`: 12 x {M0, DMA with a scalar base and a 32-bit lane offset}
      auto issue = [&](int stage, int t) __attribute__((always_inline)) {
        const int k0 = wk0 + kXBK * (t - 1);
        const char* const ba = reinterpret_cast<const char*>(g.Ap) + (A_KC ? (int64_t)k0 : (int64_t)k0 * g.ldap) * 2;
        const char* const bb = reinterpret_cast<const char*>(g.Bp) + (B_KC ? (int64_t)k0 : (int64_t)k0 * g.ldbp) * 2;
        unsigned char* const sA = smem + stage * kWsStage;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = w + 4 * jj;
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            ws_dma16(ba + offA[jj][p], sA + p * kWsPlane + j * 1024);
            ws_dma16(bb + offB[jj][p], sA + kWsOp + p * kWsPlane + j * 1024);
          }
        }
      };
      unsigned long long c_issue = 0, c_wait = 0, c_bar = 0;
      (void)c_issue; (void)c_wait; (void)c_bar;
      issue_first(wk0 - kbeg);
      if (nk > 1) {
        issue(1, 1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      for (int kt = 0; kt < nk; ++kt) {
        // tile kt + 2 into the stage the consumers left at the last barrier, then wait for tile kt + 1
        if (kt + 2 < nk) {
          {
            MAPX_WS_T0();
            if (!(kWsDbg & 1)) issue((kt + 2) % S, kt + 2);
            MAPX_WS_ACC(c_issue);
          }
          MAPX_WS_T0();
          if (!(kWsDbg & 1)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          MAPX_WS_ACC(c_wait);
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        {
          MAPX_WS_T0();
          __builtin_amdgcn_s_barrier();
          MAPX_WS_ACC(c_bar);
        }
      }
      if (w == 0) {
        MAPX_WS_PUT(0, c_issue);
        MAPX_WS_PUT(1, c_wait);
        MAPX_WS_PUT(2, c_bar);
      }
    } else if constexpr (MODE == 2) {
      // HYBRID (round 3, after the two pure forms lost): operand A fp32 — loaded, cut and stored by the loader waves,
      // two chunks of 8 floats per thread and tile — operand B as PLANES fetched by LDS-DMA (the weights of the
      // forward and input-gradient products: only the optimizer would have to write planes).  Half the DMA volume
      // of mode 0 (24 KB per K-step), half the cut work of mode 1.  Three stages; per K-step kt:
      //   DMA B(kt+2) -> stage (kt+2) % 3 | wait for the registers loaded two K-steps ago, cut + store A(kt+2) |
      //   load A(kt+4) into those registers | LDS stores done | barrier
      const int tid = threadIdx.x - 256;
      // ---- B: as mode 0
      int64_t srcB[2];
      int chB = 0;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = w + 4 * jj;
        if (B_KC) {
          const int rin = lane >> 2, c = (lane & 3) ^ ((rin >> 2) & 3);
          int row = n0 + 16 * j + rin;
          row = row < a.N ? row : a.N - 1;
          srcB[jj] = (int64_t)row * g.ldbp + 8 * c;
          chB = c;
        } else {
          const int kin = lane >> 4, c = (lane & 15) ^ (kin << 2);
          int col = n0 + 8 * c;
          col = col + 8 <= a.N ? col : 0;
          srcB[jj] = (int64_t)(4 * j + kin) * g.ldbp + col;
          chB = kin;
        }
      }
      uint32_t offB[2][3];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int p = 0; p < 3; ++p) offB[jj][p] = (uint32_t)((srcB[jj] + p * g.pb) * 2);
      auto dmaB_first = [&](int klen) __attribute__((always_inline)) {
        const char* const bb = reinterpret_cast<const char*>(g.Bp) + (B_KC ? (int64_t)kbeg : (int64_t)kbeg * g.ldbp) * 2;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = w + 4 * jj;
          const bool okB = B_KC ? 8 * chB + 8 <= klen : 4 * j + chB < klen;
#pragma unroll
          for (int p = 0; p < 3; ++p)
            ws_dma16(okB ? (const void*)(bb + offB[jj][p]) : (const void*)g.zeros, smem + kWsOp + p * kWsPlane + j * 1024);
        }
      };
      auto dmaB = [&](int stage, int t) __attribute__((always_inline)) {
        const int k0 = wk0 + kXBK * (t - 1);
        const char* const bb = reinterpret_cast<const char*>(g.Bp) + (B_KC ? (int64_t)k0 : (int64_t)k0 * g.ldbp) * 2;
        unsigned char* const sB = smem + stage * kWsStage + kWsOp;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = w + 4 * jj;
#pragma unroll
          for (int p = 0; p < 3; ++p) ws_dma16(bb + offB[jj][p], sB + p * kWsPlane + j * 1024);
        }
      };
      // ---- A: named registers ra_<set>_<chunk>_<half> (see mode 1 on why not an array)
      f32x4 ra_0_0_0 = {0.f, 0.f, 0.f, 0.f}, ra_0_0_1 = ra_0_0_0, ra_0_1_0 = ra_0_0_0, ra_0_1_1 = ra_0_0_0;
      f32x4 ra_1_0_0 = ra_0_0_0, ra_1_0_1 = ra_0_0_0, ra_1_1_0 = ra_0_0_0, ra_1_1_1 = ra_0_0_0;
      int64_t ga0, ga1;
      int sa0, sa1, ka0, ka1;
      auto setupA = [&](int f, int64_t& go, int& so, int& kq) __attribute__((always_inline)) {
        if (A_KC) {
          const int r = f >> 2, c = f & 3, row = m0 + r < a.M ? m0 + r : a.M - 1;
          go = (int64_t)row * a.lda + 8 * c;
          so = ws_slot<true>(r, c);
          kq = 8 * c + 8;
        } else {
          const int r = f >> 4, c = f & 15, col = m0 + 8 * c + 8 <= a.M ? m0 + 8 * c : 0;
          go = (int64_t)r * a.lda + col;
          so = ws_slot<false>(r, c);
          kq = r + 1;
        }
      };
      setupA(tid, ga0, sa0, ka0);
      setupA(tid + 256, ga1, sa1, ka1);
      auto koffA = [&](int k0) { return A_KC ? (int64_t)k0 : (int64_t)k0 * a.lda; };
      auto cut_store = [&](const f32x4& v0, const f32x4& v1, unsigned char* d, bool keep) __attribute__((always_inline)) {
        const float x[8] = {keep ? v0[0] : 0.f, keep ? v0[1] : 0.f, keep ? v0[2] : 0.f, keep ? v0[3] : 0.f,
                            keep ? v1[0] : 0.f, keep ? v1[1] : 0.f, keep ? v1[2] : 0.f, keep ? v1[3] : 0.f};
        uint4 hi, mid, lo;
        cut3_wide(x, hi, mid, lo);
        // stores by inline assembly: a ds_write the compiler can see makes it wait for EVERY LDS-DMA in flight
        // (it cannot tell the stages apart), which exposed the whole DMA latency in each K-step
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)d;
        const u32x4 h4 = {hi.x, hi.y, hi.z, hi.w}, m4 = {mid.x, mid.y, mid.z, mid.w}, l4 = {lo.x, lo.y, lo.z, lo.w};
        asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(h4) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(la), "v"(m4), "n"(kWsPlane) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(la), "v"(l4), "n"(2 * kWsPlane) : "memory");
      };
// The fp32 loads are inline assembly with hand-counted waits: with LDS-DMA in flight the compiler's own wait before
// the first use of a loaded register is vmcnt(0), which exposed one whole load latency per K-step.  MAPX_WA ties the
// wait to the registers so that nothing that reads them moves above it.
#define MAPX_LA(S_, C_, PTR)                                                                              \
  do {                                                                                                    \
    const float* p_ = (PTR);                                                                              \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"           \
                 : "=&v"(ra_##S_##_##C_##_0), "=&v"(ra_##S_##_##C_##_1) : "v"(p_) : "memory");            \
  } while (0)
#define MAPX_WA(S_, CNT)                                                                                  \
  asm volatile("s_waitcnt vmcnt(" #CNT ")"                                                                \
               : "+v"(ra_##S_##_0_0), "+v"(ra_##S_##_0_1), "+v"(ra_##S_##_1_0), "+v"(ra_##S_##_1_1)::"memory")
#define MAPX_SA(S_, C_, STAGE, KEEP) cut_store(ra_##S_##_##C_##_0, ra_##S_##_##C_##_1, smem + (STAGE) * kWsStage + sa##C_, KEEP)
      // full tile t >= 1 starts at k = wk0 + 32 (t - 1); loads past the last tile re-read it (never stored)
      auto ktile = [&](int t) { const int tc = t < nk - 1 ? t : nk - 1; return koffA(wk0 + kXBK * (tc - 1)); };
      // prologue: tile 0 (the K remainder) and tile 1 into stages 0 / 1; A of tiles 2 / 3 into the register sets
      {
        const int rem = wk0 - kbeg;
        const int64_t k0 = koffA(kbeg);
        const int64_t back0 = ka0 <= rem ? 0 : (A_KC ? (int64_t)(ka0 - 8) : (int64_t)(ka0 - 1) * a.lda);
        const int64_t back1 = ka1 <= rem ? 0 : (A_KC ? (int64_t)(ka1 - 8) : (int64_t)(ka1 - 1) * a.lda);
        MAPX_LA(0, 0, a.A + ga0 + k0 - back0);
        MAPX_LA(0, 1, a.A + ga1 + k0 - back1);
        if (nk > 1) {
          MAPX_LA(1, 0, a.A + ga0 + ktile(1));
          MAPX_LA(1, 1, a.A + ga1 + ktile(1));
        }
        dmaB_first(rem);
        if (nk > 1) dmaB(1, 1);
        MAPX_WA(0, 0);
        MAPX_WA(1, 0);
        MAPX_SA(0, 0, 0, ka0 <= rem);
        MAPX_SA(0, 1, 0, ka1 <= rem);
        if (nk > 1) {
          MAPX_SA(1, 0, 1, true);
          MAPX_SA(1, 1, 1, true);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nk > 2) {
          MAPX_LA(0, 0, a.A + ga0 + ktile(2));
          MAPX_LA(0, 1, a.A + ga1 + ktile(2));
          MAPX_LA(1, 0, a.A + ga0 + ktile(3));
          MAPX_LA(1, 1, a.A + ga1 + ktile(3));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      unsigned long long c_issue = 0, c_wait = 0, c_bar = 0;
      (void)c_issue; (void)c_wait; (void)c_bar;
      // iteration kt: tile kt + 2 (register set kt & 1) goes to stage (kt + 2) % 3, the set is refilled with tile
      // kt + 4.  VMEM order per iteration: 6 DMA B(kt+2) | 4 loads A(kt+4); the wait before the cut leaves the 4 loads
      // of A(kt+3) and the 6 DMA of B(kt+2) in flight (vmcnt 10): A(kt+2) is two K-steps old, B(kt+1) one.
#define MAPX_HSTEP(S_, KT)                                                        \
  do {                                                                            \
    const int t2_ = (KT) + 2;                                                     \
    if (t2_ < nk) {                                                               \
      const int st_ = t2_ % S;                                                    \
      MAPX_WS_T0();                                                               \
      dmaB(st_, t2_);                                                             \
      __builtin_amdgcn_sched_barrier(0);                                          \
      MAPX_WS_ACC(c_issue);                      /* stamp 0: DMA issue */         \
      const unsigned long long t1_ = __builtin_amdgcn_s_memtime();               \
      MAPX_WA(S_, 10);                                                            \
      MAPX_SA(S_, 0, st_, true);                                                  \
      MAPX_SA(S_, 1, st_, true);                                                  \
      __builtin_amdgcn_sched_barrier(0);                                          \
      MAPX_LA(S_, 0, a.A + ga0 + ktile((KT) + 4));                                \
      MAPX_LA(S_, 1, a.A + ga1 + ktile((KT) + 4));                                \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
      c_wait += __builtin_amdgcn_s_memtime() - t1_;   /* stamp 1: wait + cut + store + loads */ \
    } else {                                                                      \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            \
    }                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
    {                                                                             \
      MAPX_WS_T0();                                                               \
      __builtin_amdgcn_s_barrier();                                               \
      MAPX_WS_ACC(c_bar);                                                         \
    }                                                                             \
  } while (0)
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) {
        MAPX_HSTEP(0, kt);
        MAPX_HSTEP(1, kt + 1);
      }
      if (kt < nk) MAPX_HSTEP(0, kt);
      if (w == 0) {
        MAPX_WS_PUT(0, c_issue);
        MAPX_WS_PUT(1, c_wait);
        MAPX_WS_PUT(2, c_bar);
      }
#undef MAPX_HSTEP
#undef MAPX_SA
#undef MAPX_WA
#undef MAPX_LA
    } else {
      // fp32 operands: this thread's two chunks of 8 floats per operand per tile, two register sets.
      // A K-step of a loader wave is chunk by chunk  cut (44 VALU) -> 3 LDS stores -> the 2 global loads that
      // refill the chunk's registers with the tile two K-steps on:  a load is issued every ~200 cycles, so it
      // finds the CU's address path free (a burst of 8 loads per wave blocks the wave ~125 cycles per load
      // while the four loader waves queue for it: measured, tools/experiments/gemm_ws/ws_ablate.sh).
      const int tid = threadIdx.x - 256;
      // 16 named registers rg_<set>_<chunk>_<half> (chunk: 0 A0, 1 B0, 2 A1, 3 B1).  Not an array: indexed from
      // inside lambdas, part of an array of vectors stayed in scratch memory (6 of 16 vectors, a scratch load and
      // store per use: the loaders ran 3x slower than the kernel they replace).
#define MAPX_RG(S_, C_, H_) rg_##S_##_##C_##_##H_
#define MAPX_RG_DECL(S_, C_) f32x4 MAPX_RG(S_, C_, 0) = {0.f, 0.f, 0.f, 0.f}, MAPX_RG(S_, C_, 1) = {0.f, 0.f, 0.f, 0.f};
      MAPX_RG_DECL(0, 0) MAPX_RG_DECL(0, 1) MAPX_RG_DECL(0, 2) MAPX_RG_DECL(0, 3)
      MAPX_RG_DECL(1, 0) MAPX_RG_DECL(1, 1) MAPX_RG_DECL(1, 2) MAPX_RG_DECL(1, 3)
#undef MAPX_RG_DECL
      // per chunk: element offset in its operand at K offset 0 (rows clamped), byte offset of its LDS slot inside a
      // stage (plane 0), and the k's it needs (k-contiguous: 8 c + 8; k-strided: k-row + 1)
      int64_t go0, go1, go2, go3;
      int so0, so1, so2, so3, kq0, kq1, kq2, kq3;
      auto setup = [&](int ch, int64_t& go, int& so, int& kq) __attribute__((always_inline)) {
        const bool isA = (ch & 1) == 0, kc = isA ? A_KC : B_KC;
        const int f = tid + 256 * (ch >> 1), x0 = isA ? m0 : n0, X = isA ? a.M : a.N;
        const int64_t ld = isA ? a.lda : a.ldb;
        if (kc) {
          const int r = f >> 2, c = f & 3, row = x0 + r < X ? x0 + r : X - 1;
          go = (int64_t)row * ld + 8 * c;
          so = (isA ? 0 : kWsOp) + ws_slot<true>(r, c);
          kq = 8 * c + 8;
        } else {
          const int r = f >> 4, c = f & 15, col = x0 + 8 * c + 8 <= X ? x0 + 8 * c : 0;
          go = (int64_t)r * ld + col;
          so = (isA ? 0 : kWsOp) + ws_slot<false>(r, c);
          kq = r + 1;
        }
      };
      setup(0, go0, so0, kq0); setup(1, go1, so1, kq1); setup(2, go2, so2, kq2); setup(3, go3, so3, kq3);
      // K offset k0 -> element offset for operand A / B
      auto koffA = [&](int k0) { return A_KC ? (int64_t)k0 : (int64_t)k0 * a.lda; };
      auto koffB = [&](int k0) { return B_KC ? (int64_t)k0 : (int64_t)k0 * a.ldb; };
      auto cut_store = [&](const f32x4& v0, const f32x4& v1, unsigned char* d, bool keep) __attribute__((always_inline)) {
        const float x[8] = {keep ? v0[0] : 0.f, keep ? v0[1] : 0.f, keep ? v0[2] : 0.f, keep ? v0[3] : 0.f,
                            keep ? v1[0] : 0.f, keep ? v1[1] : 0.f, keep ? v1[2] : 0.f, keep ? v1[3] : 0.f};
        uint4 hi, mid, lo;
        cut3_wide(x, hi, mid, lo);
        *reinterpret_cast<uint4*>(d) = hi;
        *reinterpret_cast<uint4*>(d + kWsPlane) = mid;
        *reinterpret_cast<uint4*>(d + 2 * kWsPlane) = lo;
      };
#define MAPX_LD(S_, C_, PTR)                                                     \
  do {                                                                           \
    const float* p_ = (PTR);                                                     \
    MAPX_RG(S_, C_, 0) = *reinterpret_cast<const f32x4*>(p_);                     \
    MAPX_RG(S_, C_, 1) = *reinterpret_cast<const f32x4*>(p_ + 4);                 \
  } while (0)
#define MAPX_SRC(C_, KA, KB) (((C_) & 1) == 0 ? a.A + go##C_ + (KA) : a.B + go##C_ + (KB))
#define MAPX_ST(S_, C_, STAGE, KEEP) cut_store(MAPX_RG(S_, C_, 0), MAPX_RG(S_, C_, 1), smem + (STAGE) * kWsStage + so##C_, KEEP)
      // prologue: tile 0 (the K remainder: chunks past it re-read the tile's first k and become zeros), tile 1, tile 2
      {
        const int rem = wk0 - kbeg;
        const int64_t ka = koffA(kbeg), kb = koffB(kbeg);
        // a chunk past the remainder: back to the tile's first k (k-contiguous: chunk 0; k-strided: k-row 0)
#define MAPX_BACK(C_) ((kq##C_ <= rem) ? (int64_t)0 : (((C_) & 1) == 0 ? (A_KC ? (int64_t)(kq##C_ - 8) : (int64_t)(kq##C_ - 1) * a.lda) \
                                                                      : (B_KC ? (int64_t)(kq##C_ - 8) : (int64_t)(kq##C_ - 1) * a.ldb)))
        MAPX_LD(0, 0, MAPX_SRC(0, ka, kb) - MAPX_BACK(0));
        MAPX_LD(0, 1, MAPX_SRC(1, ka, kb) - MAPX_BACK(1));
        MAPX_LD(0, 2, MAPX_SRC(2, ka, kb) - MAPX_BACK(2));
        MAPX_LD(0, 3, MAPX_SRC(3, ka, kb) - MAPX_BACK(3));
#undef MAPX_BACK
        if (nk > 1) {
          const int64_t ka1 = koffA(wk0), kb1 = koffB(wk0);
          MAPX_LD(1, 0, MAPX_SRC(0, ka1, kb1));
          MAPX_LD(1, 1, MAPX_SRC(1, ka1, kb1));
          MAPX_LD(1, 2, MAPX_SRC(2, ka1, kb1));
          MAPX_LD(1, 3, MAPX_SRC(3, ka1, kb1));
        }
        MAPX_ST(0, 0, 0, kq0 <= rem);
        MAPX_ST(0, 1, 0, kq1 <= rem);
        MAPX_ST(0, 2, 0, kq2 <= rem);
        MAPX_ST(0, 3, 0, kq3 <= rem);
        if (nk > 2) {
          const int64_t ka2 = koffA(wk0 + kXBK), kb2 = koffB(wk0 + kXBK);
          MAPX_LD(0, 0, MAPX_SRC(0, ka2, kb2));
          MAPX_LD(0, 1, MAPX_SRC(1, ka2, kb2));
          MAPX_LD(0, 2, MAPX_SRC(2, ka2, kb2));
          MAPX_LD(0, 3, MAPX_SRC(3, ka2, kb2));
        }
      }
      __syncthreads();
      // K-step: chunk by chunk, the set's tile goes to LDS stage STAGE and the chunk is refilled with tile TNEXT
      // (clamped to the last tile)
#define MAPX_CHUNK(S_, C_, STAGE, KA, KB)                 \
  MAPX_ST(S_, C_, STAGE, true);                           \
  __builtin_amdgcn_sched_barrier(0);                      \
  MAPX_LD(S_, C_, MAPX_SRC(C_, KA, KB));                  \
  __builtin_amdgcn_sched_barrier(0);
#define MAPX_KSTEP(S_, STAGE, TNEXT)                                                     \
  do {                                                                                   \
    const int tn_ = (TNEXT) < nk - 1 ? (TNEXT) : nk - 1, k0_ = wk0 + kXBK * (tn_ - 1);   \
    const int64_t ka_ = koffA(k0_), kb_ = koffB(k0_);                                    \
    MAPX_CHUNK(S_, 0, STAGE, ka_, kb_)                                                   \
    MAPX_CHUNK(S_, 1, STAGE, ka_, kb_)                                                   \
    MAPX_CHUNK(S_, 2, STAGE, ka_, kb_)                                                   \
    MAPX_CHUNK(S_, 3, STAGE, ka_, kb_)                                                   \
  } while (0)
      unsigned long long c_issue = 0, c_bar = 0;
      (void)c_issue; (void)c_bar;
      // (branch-free loop body, the K range's tail handled behind it)
      int kt = 0;
      for (; kt + 2 < nk; kt += 2) {
        {
          MAPX_WS_T0();
          if (!(kWsDbg & 1)) MAPX_KSTEP(1, 1, kt + 3);
          MAPX_WS_ACC(c_issue);
        }
        {
          MAPX_WS_T0();
          __syncthreads();
          MAPX_WS_ACC(c_bar);
        }
        {
          MAPX_WS_T0();
          if (!(kWsDbg & 1)) MAPX_KSTEP(0, 0, kt + 4);
          MAPX_WS_ACC(c_issue);
        }
        MAPX_WS_T0();
        __syncthreads();
        MAPX_WS_ACC(c_bar);
      }
      if (kt + 1 < nk) {                          // two K-steps left: the last tile goes to stage 1
        MAPX_ST(1, 0, 1, true);
        MAPX_ST(1, 1, 1, true);
        MAPX_ST(1, 2, 1, true);
        MAPX_ST(1, 3, 1, true);
        __syncthreads();
      }
      __syncthreads();
#undef MAPX_KSTEP
#undef MAPX_CHUNK
#undef MAPX_ST
#undef MAPX_SRC
#undef MAPX_LD
#undef MAPX_RG
      if (w == 0) {
        MAPX_WS_PUT(0, c_issue);
        MAPX_WS_PUT(2, c_bar);
      }
    }
  } else {
    // ------------------------------------------------------------------ consumers: 2 x 2 waves of 64 x 64
    abase = (wave >> 1) * 64;
    bbase = (wave & 1) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = cor[i][j][r] = 0.f;
    // this lane's fragment offsets inside a plane, tile 0 / k16 half 0 (the others are XORs / constants away)
    const int q = (lane >> 2) & 3, p4 = lane & 3, hf = (lane >> 4) & 1;
    const int offA = A_KC ? (abase + l31) * 64 + ((kh ^ ((l31 >> 2) & 3)) << 4)
                          : (8 * kh + q) * 256 + ((((abase >> 3) + 2 * hf + (p4 >> 1)) ^ (q << 2)) << 4) + (p4 & 1) * 8;
    const int offB = B_KC ? (bbase + l31) * 64 + ((kh ^ ((l31 >> 2) & 3)) << 4)
                          : (8 * kh + q) * 256 + ((((bbase >> 3) + 2 * hf + (p4 >> 1)) ^ (q << 2)) << 4) + (p4 & 1) * 8;
    auto frag = [&](const unsigned char* sp, bool kc, int off, int s2, int t) __attribute__((always_inline)) -> bf16x8 {
      if (kc) return *reinterpret_cast<const bf16x8*>(sp + ((off + t * 32 * 64) ^ (s2 << 5)));
      const unsigned char* a0 = sp + ((off ^ (t << 6)) + s2 * 16 * 256);
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * 256));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // Software pipeline over k16 halves: while the 24 MFMAs of one half run, the 12 fragments of the next half
    // are read (one fragment behind each of the first 12 MFMAs), so that no MFMA waits for LDS; the barrier of a K-step falls
    // between its two halves' MFMAs:
    //   iteration kt:  [MFMAs half 0 of tile kt || reads half 1 of tile kt]  barrier(kt)
    //                  [MFMAs half 1 of tile kt || reads half 0 of tile kt+1]
    // (tile kt is only read between barrier(kt-1) and barrier(kt): the ring needs no extra stage for it).
    bf16x8 fa[2][3][2], fb[2][3][2];              // [register set = k16 half][plane hi/mid/lo][tile]
    auto read_frag = [&](auto setc, auto qc, const unsigned char* sA, const unsigned char* sB, int s2)
        __attribute__((always_inline)) {
      constexpr int Y = decltype(setc)::value, qq = decltype(qc)::value;
      constexpr int op = frag_order(qq, 0, 2), pl = frag_order(qq, 1, 2), t = frag_order(qq, 2, 2);
      if constexpr (op == 0) fa[Y][pl][t] = frag(sA + pl * kWsPlane, A_KC, offA, s2, t);
      else fb[Y][pl][t] = frag(sB + pl * kWsPlane, B_KC, offB, s2, t);
    };
    auto phase = [&](auto setc, const unsigned char* sA, const unsigned char* sB, int s2) __attribute__((always_inline)) {
      constexpr int X = decltype(setc)::value;    // the MFMAs consume set X, the reads fill set X ^ 1
      unroll_seq([&](auto zc) __attribute__((always_inline)) {
        constexpr int z = decltype(zc)::value, t4 = z / 6, i = t4 / 2, j = t4 % 2, term = z % 6;
        if (!(kWsDbg & 2)) {
        if (term == 0) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[X][2][i], fb[X][0][j], cor[i][j], 0, 0, 0);
        if (term == 1) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[X][0][i], fb[X][2][j], cor[i][j], 0, 0, 0);
        if (term == 2) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[X][1][i], fb[X][1][j], cor[i][j], 0, 0, 0);
        if (term == 3) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[X][1][i], fb[X][0][j], cor[i][j], 0, 0, 0);
        if (term == 4) cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[X][0][i], fb[X][1][j], cor[i][j], 0, 0, 0);
        if (term == 5) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[X][0][i], fb[X][0][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (z < 12 && !(kWsDbg & 4)) {                   // reads in the first half of the phase: done well before the barrier
          read_frag(std::integral_constant<int, X ^ 1>{}, std::integral_constant<int, z>{}, sA, sB, s2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }, std::make_integer_sequence<int, 24>{});
    };
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(2);
    if constexpr (MODE != 1) __builtin_amdgcn_s_barrier(); else __syncthreads();
    MAPX_WS_STAMP_AT(1);
    unroll_seq([&](auto qc) __attribute__((always_inline)) {
      read_frag(std::integral_constant<int, 0>{}, qc, smem, smem + kWsOp, 0);
    }, std::make_integer_sequence<int, 12>{});
    unsigned long long c_cbar = 0;
    (void)c_cbar;
    int st = 0;                                    // kt % S
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* const sA = smem + st * kWsStage;
      st = st + 1 == S ? 0 : st + 1;
      const unsigned char* const sN = smem + st * kWsStage;
      phase(std::integral_constant<int, 0>{}, sA, sA + kWsOp, 1);
      if constexpr (MODE != 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MAPX_WS_T0();
        __builtin_amdgcn_s_barrier();
        MAPX_WS_ACC(c_cbar);
      } else {
        MAPX_WS_T0();
        __syncthreads();
        MAPX_WS_ACC(c_cbar);
      }
      // (past the last tile the reads fetch a stage nobody writes any more; their values are not used)
      phase(std::integral_constant<int, 1>{}, sN, sN + kWsOp, 0);
    }
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    MAPX_WS_STAMP_AT(2);
    if (wave == 0) MAPX_WS_PUT(3, c_cbar);
  }

  // ---------------------------------------------------------------------- epilogue (all 8 waves)
  __syncthreads();                              // every DMA has landed (loaders drained), every fragment is read
  float* const tile = reinterpret_cast<float*>(smem);
  constexpr int LDT = BN + 4;
  if (wave < 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[(abase + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2)) * LDT + bbase + 32 * j + l31] = acc[i][j][r] + cor[i][j][r];
  }
  __syncthreads();
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
      case MAPX_EPI_RELU_MASK_COLSUM: epilogue_rows_x3_vec<MAPX_EPI_RELU_MASK_COLSUM, BM, BN, NT>(a, C, tile, m0, n0); break;
      default: epilogue_rows_x3_vec<MAPX_EPI_NONE, BM, BN, NT>(a, C, tile, m0, n0); break;
    }
    MAPX_WS_STAMP_AT(3);
    return;
  }
  switch (a.epi) {
    case MAPX_EPI_BIAS: epilogue_rows_x3<MAPX_EPI_BIAS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_BIAS_RELU: epilogue_rows_x3<MAPX_EPI_BIAS_RELU, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_BIAS_CROSS: epilogue_rows_x3<MAPX_EPI_BIAS_CROSS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_ADD: epilogue_rows_x3<MAPX_EPI_ADD, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_RELU_MASK: epilogue_rows_x3<MAPX_EPI_RELU_MASK, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    default: epilogue_rows_x3<MAPX_EPI_NONE, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Sixteen waves, fp32 operands: 8 consumers (2 x 4, wave tile 64 x 32: accumulators 64 registers) + 8 loaders,
// i.e. two consumers and two loaders on every SIMD at <= 128 registers each.  A loader wave's K-step is a serial
// chain of things that block it (global loads queue for the CU's address path, ds_write_b128 for the LDS store
// path, the cut's 44 VALU per chunk): ~2000 cycles for 16 floats per lane with ONE loader per SIMD (measured,
// tools/experiments/gemm_ws/ws_ablate.sh), more than the 1536 MFMA cycles of the K-step.  With two per SIMD each has half the work
// and the two overlap each other's waits.  The consumers keep one fragment buffer for A (two 32-row tiles, used
// one after the other) and two for B (one 32-column tile): a tile's fragments for the NEXT k16 half are read while
// the other tile's six MFMAs run, the barrier of a K-step sits between the two tiles of its second half.
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(1024) gemm_ws16_kernel(GemmWsArgs g) {
  const GemmX3Args& a = g.e;
  constexpr int BM = 128, BN = 128, NT = 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = a.tiles_m * a.tiles_n;
  int lin = blockIdx.x;
  const int per = nb / 8;
  if (lin < per * 8) lin = (lin % 8) * per + lin / 8;      // XCD-aware tile order (see gemm.hip)
  const int tm = lin / a.tiles_n, tn = lin % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.y * a.k_chunk;
  const int kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
  const int nk = (kend - kbeg + kXBK - 1) / kXBK;
  const int wk0 = kend - kXBK * (nk - 1);                  // tile 0 = [kbeg, wk0) (the remainder), then full tiles
  float* __restrict__ C = a.C + (int64_t)blockIdx.y * a.slab_stride;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  f32x16 acc[2], cor[2];
  int abase = 0, bbase = 0;
  MAPX_WS_STAMP_AT(0);

  if (wave >= 8) {
    // ------------------------------------------------------------------ loaders: one chunk of A and one of B per tile
    const int tid = threadIdx.x - 512;
#define MAPX_RG(S_, C_, H_) rg_##S_##_##C_##_##H_
#define MAPX_RG_DECL(S_, C_) f32x4 MAPX_RG(S_, C_, 0) = {0.f, 0.f, 0.f, 0.f}, MAPX_RG(S_, C_, 1) = {0.f, 0.f, 0.f, 0.f};
    MAPX_RG_DECL(0, 0) MAPX_RG_DECL(0, 1) MAPX_RG_DECL(1, 0) MAPX_RG_DECL(1, 1)
#undef MAPX_RG_DECL
    int64_t go0, go1;
    int so0, so1, kq0, kq1;
    auto setup = [&](int ch, int64_t& go, int& so, int& kq) __attribute__((always_inline)) {
      const bool isA = ch == 0, kc = isA ? A_KC : B_KC;
      const int f = tid, x0 = isA ? m0 : n0, X = isA ? a.M : a.N;
      const int64_t ld = isA ? a.lda : a.ldb;
      if (kc) {
        const int r = f >> 2, c = f & 3, row = x0 + r < X ? x0 + r : X - 1;
        go = (int64_t)row * ld + 8 * c;
        so = (isA ? 0 : kWsOp) + ws_slot<true>(r, c);
        kq = 8 * c + 8;
      } else {
        const int r = f >> 4, c = f & 15, col = x0 + 8 * c + 8 <= X ? x0 + 8 * c : 0;
        go = (int64_t)r * ld + col;
        so = (isA ? 0 : kWsOp) + ws_slot<false>(r, c);
        kq = r + 1;
      }
    };
    setup(0, go0, so0, kq0);
    setup(1, go1, so1, kq1);
    auto koffA = [&](int k0) { return A_KC ? (int64_t)k0 : (int64_t)k0 * a.lda; };
    auto koffB = [&](int k0) { return B_KC ? (int64_t)k0 : (int64_t)k0 * a.ldb; };
    auto cut_store = [&](const f32x4& v0, const f32x4& v1, unsigned char* d, bool keep) __attribute__((always_inline)) {
      const float x[8] = {keep ? v0[0] : 0.f, keep ? v0[1] : 0.f, keep ? v0[2] : 0.f, keep ? v0[3] : 0.f,
                          keep ? v1[0] : 0.f, keep ? v1[1] : 0.f, keep ? v1[2] : 0.f, keep ? v1[3] : 0.f};
      uint4 hi, mid, lo;
      cut3_wide(x, hi, mid, lo);
      *reinterpret_cast<uint4*>(d) = hi;
      *reinterpret_cast<uint4*>(d + kWsPlane) = mid;
      *reinterpret_cast<uint4*>(d + 2 * kWsPlane) = lo;
    };
#define MAPX_LD(S_, C_, PTR)                                                     \
  do {                                                                           \
    const float* p_ = (PTR);                                                     \
    MAPX_RG(S_, C_, 0) = *reinterpret_cast<const f32x4*>(p_);                     \
    MAPX_RG(S_, C_, 1) = *reinterpret_cast<const f32x4*>(p_ + 4);                 \
  } while (0)
#define MAPX_SRC(C_, KA, KB) ((C_) == 0 ? a.A + go##C_ + (KA) : a.B + go##C_ + (KB))
#define MAPX_ST(S_, C_, STAGE, KEEP) cut_store(MAPX_RG(S_, C_, 0), MAPX_RG(S_, C_, 1), smem + (STAGE) * kWsStage + so##C_, KEEP)
    {
      const int rem = wk0 - kbeg;
      const int64_t ka = koffA(kbeg), kb = koffB(kbeg);
      // a chunk past the remainder re-reads the tile's first k (its data become zeros)
      const int64_t back0 = kq0 <= rem ? (int64_t)0 : (A_KC ? (int64_t)(kq0 - 8) : (int64_t)(kq0 - 1) * a.lda);
      const int64_t back1 = kq1 <= rem ? (int64_t)0 : (B_KC ? (int64_t)(kq1 - 8) : (int64_t)(kq1 - 1) * a.ldb);
      MAPX_LD(0, 0, MAPX_SRC(0, ka, kb) - back0);
      MAPX_LD(0, 1, MAPX_SRC(1, ka, kb) - back1);
      if (nk > 1) {
        const int64_t ka1 = koffA(wk0), kb1 = koffB(wk0);
        MAPX_LD(1, 0, MAPX_SRC(0, ka1, kb1));
        MAPX_LD(1, 1, MAPX_SRC(1, ka1, kb1));
      }
      MAPX_ST(0, 0, 0, kq0 <= rem);
      MAPX_ST(0, 1, 0, kq1 <= rem);
      if (nk > 2) {
        const int64_t ka2 = koffA(wk0 + kXBK), kb2 = koffB(wk0 + kXBK);
        MAPX_LD(0, 0, MAPX_SRC(0, ka2, kb2));
        MAPX_LD(0, 1, MAPX_SRC(1, ka2, kb2));
      }
    }
    __syncthreads();
#define MAPX_KSTEP(S_, STAGE, TNEXT)                                                     \
  do {                                                                                   \
    const int tn_ = (TNEXT) < nk - 1 ? (TNEXT) : nk - 1, k0_ = wk0 + kXBK * (tn_ - 1);   \
    const int64_t ka_ = koffA(k0_), kb_ = koffB(k0_);                                    \
    MAPX_ST(S_, 0, STAGE, true);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    MAPX_LD(S_, 0, MAPX_SRC(0, ka_, kb_));                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    MAPX_ST(S_, 1, STAGE, true);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    MAPX_LD(S_, 1, MAPX_SRC(1, ka_, kb_));                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  } while (0)
    unsigned long long c_issue = 0, c_bar = 0;
    (void)c_issue; (void)c_bar;
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
      {
        MAPX_WS_T0();
        if (!(kWsDbg & 1)) MAPX_KSTEP(1, 1, kt + 3);
        MAPX_WS_ACC(c_issue);
      }
      {
        MAPX_WS_T0();
        __syncthreads();
        MAPX_WS_ACC(c_bar);
      }
      {
        MAPX_WS_T0();
        if (!(kWsDbg & 1)) MAPX_KSTEP(0, 0, kt + 4);
        MAPX_WS_ACC(c_issue);
      }
      MAPX_WS_T0();
      __syncthreads();
      MAPX_WS_ACC(c_bar);
    }
    if (kt + 1 < nk) {
      MAPX_ST(1, 0, 1, true);
      MAPX_ST(1, 1, 1, true);
      __syncthreads();
    }
    __syncthreads();
#undef MAPX_KSTEP
#undef MAPX_ST
#undef MAPX_SRC
#undef MAPX_LD
#undef MAPX_RG
    if (wave == 8) {
      MAPX_WS_PUT(0, c_issue);
      MAPX_WS_PUT(2, c_bar);
    }
  } else {
    // ------------------------------------------------------------------ consumers: 2 x 4 waves of 64 x 32
    abase = (wave >> 2) * 64;
    bbase = (wave & 3) * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = cor[i][r] = 0.f;
    const int q = (lane >> 2) & 3, p4 = lane & 3, hf = (lane >> 4) & 1;
    const int offA = A_KC ? (abase + l31) * 64 + ((kh ^ ((l31 >> 2) & 3)) << 4)
                          : (8 * kh + q) * 256 + ((((abase >> 3) + 2 * hf + (p4 >> 1)) ^ (q << 2)) << 4) + (p4 & 1) * 8;
    const int offB = B_KC ? (bbase + l31) * 64 + ((kh ^ ((l31 >> 2) & 3)) << 4)
                          : (8 * kh + q) * 256 + ((((bbase >> 3) + 2 * hf + (p4 >> 1)) ^ (q << 2)) << 4) + (p4 & 1) * 8;
    // fragment of plane `pl`, k16 half s2; A: 32-row tile t of the wave's 64 rows
    auto fragA = [&](const unsigned char* st, int pl, int s2, int t) __attribute__((always_inline)) -> bf16x8 {
      const unsigned char* sp = st + pl * kWsPlane;
      if (A_KC) return *reinterpret_cast<const bf16x8*>(sp + ((offA + t * 32 * 64) ^ (s2 << 5)));
      const unsigned char* a0 = sp + ((offA ^ (t << 6)) + s2 * 16 * 256);
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * 256));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto fragB = [&](const unsigned char* st, int pl, int s2) __attribute__((always_inline)) -> bf16x8 {
      const unsigned char* sp = st + kWsOp + pl * kWsPlane;
      if (B_KC) return *reinterpret_cast<const bf16x8*>(sp + (offB ^ (s2 << 5)));
      const unsigned char* a0 = sp + (offB + s2 * 16 * 256);
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * 256));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 fa0[3], fa1[3], fb0[3], fb1[3];          // [plane]: A tile 0 / tile 1, B buffer 0 / 1
#define MAPX_RD_A(dst, st, s2, t) do { dst[0] = fragA(st, 0, s2, t); dst[1] = fragA(st, 1, s2, t); dst[2] = fragA(st, 2, s2, t); } while (0)
#define MAPX_RD_B(dst, st, s2) do { dst[0] = fragB(st, 0, s2); dst[1] = fragB(st, 1, s2); dst[2] = fragB(st, 2, s2); } while (0)
#define MAPX_MM(i, fa, fb)                                                                           \
  do {                                                                                               \
    if (!(kWsDbg & 2)) {                                                                             \
      cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], cor[i], 0, 0, 0);               \
      cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], cor[i], 0, 0, 0);               \
      cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], cor[i], 0, 0, 0);               \
      cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], cor[i], 0, 0, 0);               \
      cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], cor[i], 0, 0, 0);               \
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[i], 0, 0, 0);               \
    }                                                                                                \
  } while (0)
    __syncthreads();
    MAPX_WS_STAMP_AT(1);
    MAPX_RD_A(fa0, smem, 0, 0);
    MAPX_RD_B(fb0, smem, 0);
    unsigned long long c_cbar = 0;
    (void)c_cbar;
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* const sC = smem + (kt & 1) * kWsStage;
      const unsigned char* const sN = smem + ((kt + 1) & 1) * kWsStage;
      // half 0
      MAPX_RD_A(fa1, sC, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      MAPX_MM(0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      MAPX_RD_A(fa0, sC, 1, 0);
      MAPX_RD_B(fb1, sC, 1);
      __builtin_amdgcn_sched_barrier(0);
      MAPX_MM(1, fa1, fb0);
      __builtin_amdgcn_sched_barrier(0);
      // half 1; the K-step's barrier between its two tiles
      MAPX_RD_A(fa1, sC, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      MAPX_MM(0, fa0, fb1);
      __builtin_amdgcn_sched_barrier(0);
      {
        MAPX_WS_T0();
        __syncthreads();
        MAPX_WS_ACC(c_cbar);
      }
      // (past the last tile these reads fetch a stage nobody writes any more; their values are not used)
      MAPX_RD_A(fa0, sN, 0, 0);
      MAPX_RD_B(fb0, sN, 0);
      __builtin_amdgcn_sched_barrier(0);
      MAPX_MM(1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef MAPX_MM
#undef MAPX_RD_B
#undef MAPX_RD_A
    MAPX_WS_STAMP_AT(2);
    if (wave == 0) MAPX_WS_PUT(3, c_cbar);
  }

  __syncthreads();
  float* const tile = reinterpret_cast<float*>(smem);
  constexpr int LDT = BN + 4;
  if (wave < 8) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(abase + 32 * i + 4 * kh + (r & 3) + 8 * (r >> 2)) * LDT + bbase + l31] = acc[i][r] + cor[i][r];
  }
  __syncthreads();
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
      case MAPX_EPI_RELU_MASK_COLSUM: epilogue_rows_x3_vec<MAPX_EPI_RELU_MASK_COLSUM, BM, BN, NT>(a, C, tile, m0, n0); break;
      default: epilogue_rows_x3_vec<MAPX_EPI_NONE, BM, BN, NT>(a, C, tile, m0, n0); break;
    }
    MAPX_WS_STAMP_AT(3);
    return;
  }
  switch (a.epi) {
    case MAPX_EPI_BIAS: epilogue_rows_x3<MAPX_EPI_BIAS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_BIAS_RELU: epilogue_rows_x3<MAPX_EPI_BIAS_RELU, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_BIAS_CROSS: epilogue_rows_x3<MAPX_EPI_BIAS_CROSS, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_ADD: epilogue_rows_x3<MAPX_EPI_ADD, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    case MAPX_EPI_RELU_MASK: epilogue_rows_x3<MAPX_EPI_RELU_MASK, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
    default: epilogue_rows_x3<MAPX_EPI_NONE, BM, BN, NT>(a, C, tile, m0, n0, vio); break;
  }
}

template <bool A_KC, bool B_KC>
static hipError_t launch_ws16(const GemmWsArgs& g, int nsplit, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * kWsStage;
  static_assert(lds >= (size_t)128 * 132 * 4 + 32 * 32 * 16, "epilogue tile + column-sum rows must fit");
  auto* fn = &gemm_ws16_kernel<A_KC, B_KC>;
  static hipError_t raised =
      hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(fn, dim3(g.e.tiles_m * g.e.tiles_n, nsplit), dim3(1024), lds, stream, g);
  return hipSuccess;
}

// fp32 [M, N] (leading dimension ld) -> three bf16 planes [3][M][ldp] (plane stride ps elements); N % 8 == 0
__global__ void __launch_bounds__(256) cut_planes_kernel(const float* __restrict__ x, int64_t ld, int M, int N,
                                                         bf16_t* __restrict__ planes, int64_t ldp, int64_t ps) {
  const int cpr = N / 8;
  const int64_t total = (int64_t)M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i % cpr) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(x + r * ld + c);
    const float4 v1 = *reinterpret_cast<const float4*>(x + r * ld + c + 4);
    const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint4 hi, mid, lo;
    cut3(xs, hi, mid, lo);
    bf16_t* d = planes + r * ldp + c;
    *reinterpret_cast<uint4*>(d) = hi;
    *reinterpret_cast<uint4*>(d + ps) = mid;
    *reinterpret_cast<uint4*>(d + 2 * ps) = lo;
  }
}

__global__ void __launch_bounds__(256) splitk_reduce_ws_kernel(const float* __restrict__ slabs, int64_t slab_stride,
                                                               int nsplit, int64_t n4, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(slabs)[i];
    v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f;
    for (int s = 1; s < nsplit; ++s) {
      const float4 x = reinterpret_cast<const float4*>(slabs + s * slab_stride)[i];
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

__device__ __attribute__((aligned(64))) unsigned char g_ws_zero_page[64];
static unsigned long long* g_ws_stamps = nullptr;

template <bool A_KC, bool B_KC, int MODE, int PRIO>
static hipError_t launch_ws(const GemmWsArgs& g, int nsplit, hipStream_t stream) {
  constexpr size_t lds = (size_t)(MODE == 1 ? 2 : 3) * kWsStage;
  static_assert(lds >= (size_t)128 * 132 * 4 + 16 * 32 * 16, "epilogue tile + column-sum rows must fit");
  auto* fn = &gemm_ws_kernel<A_KC, B_KC, MODE, PRIO>;
  static hipError_t raised =
      hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (raised != hipSuccess) return raised;
  hipLaunchKernelGGL(fn, dim3(g.e.tiles_m * g.e.tiles_n, nsplit), dim3(512), lds, stream, g);
  return hipSuccess;
}

}  // namespace mapx

#ifdef MAPX_WS_STAMP
extern "C" int mapx_gemm_ws_set_stamps(void* buf) {       // diagnostic builds only (tools/experiments/gemm_ws/ws_ablate.sh); not part of the ABI
  mapx::g_ws_stamps = static_cast<unsigned long long*>(buf);
  return MAPX_OK;
}
#endif

extern "C" int mapx_cut_planes(const float* x, int64_t ld, int M, int N, void* planes, int64_t ldp,
                               int64_t plane_stride, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(x && planes && M >= 0 && N >= 0, "cut_planes: bad arguments");
  MAPX_REQUIRE(N % 8 == 0 && ld % 4 == 0 && ldp % 8 == 0 && plane_stride % 8 == 0 && (uintptr_t)x % 16 == 0 &&
                   (uintptr_t)planes % 16 == 0,
               "cut_planes: N %% 8, ld %% 4, ldp %% 8, plane_stride %% 8 must be 0 and the buffers 16-byte aligned");
  if (M == 0 || N == 0) return MAPX_OK;
  hipLaunchKernelGGL(cut_planes_kernel, dim3(grid_for((int64_t)M * N / 8, 256)), dim3(256), 0, stream, x, ld, M, N,
                     static_cast<bf16_t*>(planes), ldp, plane_stride);
  return check_launch("cut_planes");
}

extern "C" int mapx_gemm_ws(int mode, int a_kc, int b_kc, int M, int N, int K, const void* A, int64_t lda,
                            int64_t pa, const void* B, int64_t ldb, int64_t pb, float* C, int64_t ldc, int epi,
                            const float* bias, const float* aux1, int64_t ld1, const float* aux2, int64_t ld2,
                            float* out2, int64_t ldo2, int nsplit, void* ws, size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "gemm_ws: bad arguments");
  const int prio = (mode & 4) ? 2 : (mode & 2) ? 1 : 0;    // experiment switches: 2 consumers / 4 loaders at raised priority
  const bool ws16 = (mode & 8) != 0;                        // 16 waves: 8 consumers of 64 x 32 + 8 loaders, fp32 operands
  const bool hybrid = (mode & 16) != 0;                     // A fp32 (loader waves cut), B planes (LDS-DMA)
  mode = ws16 ? 1 : hybrid ? 2 : (mode & 1);
  MAPX_REQUIRE(!(a_kc == 0 && b_kc != 0), "gemm_ws: layout (A m-contiguous, B k-contiguous) unused");
  MAPX_REQUIRE(epi >= MAPX_EPI_NONE && epi <= MAPX_EPI_RELU_MASK_COLSUM, "gemm_ws: bad epilogue %d", epi);
  const int64_t align = mode == 0 ? 8 : 4;
  MAPX_REQUIRE(lda % align == 0 && ldb % (mode == 2 ? 8 : align) == 0 && (uintptr_t)A % 16 == 0 && (uintptr_t)B % 16 == 0 &&
                   pa % 8 == 0 && pb % 8 == 0,
               "gemm_ws: operands must be 16-byte aligned with 16-byte rows");
  MAPX_REQUIRE(K % 8 == 0 && (a_kc || M % 8 == 0) && (b_kc || N % 8 == 0),
               "gemm_ws: contiguous extents must be multiples of 8 (a chunk is all-in or all-out)");
  if (nsplit < 1) nsplit = 1;
  MAPX_REQUIRE(nsplit == 1 || epi == MAPX_EPI_NONE, "gemm_ws: split-K needs EPI_NONE");
  GemmWsArgs g{};
  GemmX3Args& e = g.e;
  e.C = C; e.ldc = ldc; e.M = M; e.N = N; e.K = K; e.epi = epi; e.bias = bias;
  e.aux1 = aux1; e.ld1 = ld1; e.aux2 = aux2; e.ld2 = ld2; e.out2 = out2; e.ldo2 = ldo2;
  e.k_chunk = K; e.slab_stride = 0;
  if (mode == 0) {
    g.Ap = static_cast<const bf16_t*>(A); g.ldap = lda; g.pa = pa;
    g.Bp = static_cast<const bf16_t*>(B); g.ldbp = ldb; g.pb = pb;
    static void* z = nullptr;                 // (resolved once: no runtime call inside a stream capture)
    if (!z) MAPX_HIP(hipGetSymbolAddress(&z, HIP_SYMBOL(g_ws_zero_page)));
    g.zeros = static_cast<const bf16_t*>(z);
  } else if (mode == 2) {
    e.A = static_cast<const float*>(A); e.lda = lda;
    g.Bp = static_cast<const bf16_t*>(B); g.ldbp = ldb; g.pb = pb;
    static void* z2 = nullptr;
    if (!z2) MAPX_HIP(hipGetSymbolAddress(&z2, HIP_SYMBOL(g_ws_zero_page)));
    g.zeros = static_cast<const bf16_t*>(z2);
  } else {
    e.A = static_cast<const float*>(A); e.lda = lda;
    e.B = static_cast<const float*>(B); e.ldb = ldb;
  }
  g.stamps = g_ws_stamps;
  if (nsplit > 1) {
    const size_t need = (size_t)nsplit * M * N * sizeof(float);
    if (!ws || ws_bytes < need) {
      set_error("gemm_ws: split-K workspace %zu < %zu", ws_bytes, need);
      return MAPX_EWORKSPACE;
    }
    const int kc = (int)ceil_div(ceil_div(K, nsplit), 64) * 64;
    e.k_chunk = kc;
    nsplit = (int)ceil_div(K, kc);
    e.C = static_cast<float*>(ws);
    e.ldc = N;
    e.slab_stride = (int64_t)M * N;
    MAPX_REQUIRE(ldc == N && ((int64_t)M * N) % 4 == 0 && (uintptr_t)C % 16 == 0 && (uintptr_t)ws % 16 == 0,
                 "gemm_ws: split-K output must be dense and 16-byte aligned");
  }
  e.tiles_m = (M + 127) / 128; e.tiles_n = (N + 127) / 128;
  hipError_t err;
#define MAPX_WS_P(AK, BK_, MD)                                                                       \
  (prio == 2 ? launch_ws<AK, BK_, MD, 2>(g, nsplit, stream)                                          \
             : prio == 1 ? launch_ws<AK, BK_, MD, 1>(g, nsplit, stream) : launch_ws<AK, BK_, MD, 0>(g, nsplit, stream))
#define MAPX_WS(AK, BK_) (mode == 0 ? MAPX_WS_P(AK, BK_, 0) : mode == 2 ? MAPX_WS_P(AK, BK_, 2) : MAPX_WS_P(AK, BK_, 1))
  if (ws16) err = (a_kc && b_kc) ? launch_ws16<true, true>(g, nsplit, stream)
                  : a_kc ? launch_ws16<true, false>(g, nsplit, stream) : launch_ws16<false, false>(g, nsplit, stream);
  else if (a_kc && b_kc) err = MAPX_WS(true, true);
  else if (a_kc) err = MAPX_WS(true, false);
  else err = MAPX_WS(false, false);
#undef MAPX_WS
#undef MAPX_WS_P
  MAPX_HIP(err);
  if (nsplit > 1)
    hipLaunchKernelGGL(splitk_reduce_ws_kernel, dim3(grid_for((int64_t)M * N / 4, 256)), dim3(256), 0, stream,
                       static_cast<const float*>(ws), e.slab_stride, nsplit, (int64_t)M * N / 4, C);
  return check_launch("gemm_ws");
}
