// Dense-trunk entry points (mapx_gemm_f32) and the small kernels around the GEMMs: the grouped encoder's slot
// layout, deferred slab / partial sums, column sums and the elementwise backward pieces of the ReLU and
// cross layers.  The products themselves are csrc/gemm_x3.hip (fp32 operands as three bf16 pieces on the
// bf16 matrix cores) and csrc/gemm_bf16.hip (bf16 mode).  Round 1's fp32-MFMA family (v_mfma_f32_32x32x2_f32,
// incl. the LDS-DMA deep-K weight-gradient kernel) lived here until round 3; it was the slower of the two
// (92 vs 130 TF class average) and ran nowhere on the default path, so it was removed rather than kept untested.
//
//   C[m,n] = epilogue( sum_k A(m,k) * B(k,n) )
//   A_KC : A(m,k) = A[m*lda + k]  (k contiguous)      else A(m,k) = A[k*lda + m]
//   B_KC : B(k,n) = B[n*ldb + k]  (k contiguous)      else B(k,n) = B[k*ldb + n]
//   forward  Y = X W^T      : A_KC (X [B,in]),   B_KC (W [out,in])
//   dX = dY W               : A_KC (dY [B,out]), B k-strided (W [out,in] read as [K=out, N=in])
//   dW = dY^T X             : A k-strided (dY [B,out] read as [K=B, M=out]), B k-strided (X [B,in])
#include "../../include/mapx_hip.h"
#include "common.h"
#include "amax.h"
#include "gemm_grouped.h"
#include "gemm_x3_common.h"

namespace mapx {

// ------------------------------------------------------------------------------------------
// Grouped GEMMs of the MFP head's feat_encoder (models.py:74-75).  The reference computes all
// F*P encoder outputs per row and then gathers the L masked fields' P-blocks: 74 % of the
// forward GEMM and of its weight-gradient GEMM is never read.  Here targets (b, l) are sorted
// by field (groups padded to 128 slots; rowmap[slot] = batch row or -1), and
//   FWD  h[slot, :]      = final[rowmap[slot], :] . W[f*P:(f+1)*P, :]^T + bias[f*P:(f+1)*P]
//   DW   dW[f*P + p, n]  = sum_{slot in group f} dh[slot, p] * final[rowmap[slot], n]
// (kernels: gemm_x3.hip, gemm_grouped_x3_kernel).  P = 32 only (the reference default); other proj sizes and
// widths that are not a multiple of 8 use the dense path.

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
  if (c < N) {
#pragma unroll 8                 // eight loads in flight, added in the same order (a tall narrow matrix — xDeepFM's
    for (int r = r0 + rl; r < r1; r += 4) v += x[(int64_t)r * ld + c];   // [65536, 50] — is one block per chunk: 51 us)
  }
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
                                                        int accumulate, float* __restrict__ part,
                                                        amax_rec* __restrict__ amax_o1, const int32_t* __restrict__ epoch) {
  // CL lanes x float4 columns per block row; RL row lanes; N % 4 == 0, lda % 4 == 0 (host-checked)
  constexpr int RL = 256 / CL;
  const int cl = threadIdx.x % CL;
  const int col = (blockIdx.x * CL + cl) * 4;
  const int rl = threadIdx.x / CL;
  const int rows_per = (M + kColChunks - 1) / kColChunks;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t amx = 0;                     // max |o1| (dz or t) for the products that read it next (amax.h)
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
        amx = amax4(amx, dz.x, dz.y, dz.z, dz.w);
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
        amx = amax4(amx, t.x, t.y, t.z, t.w);
        reinterpret_cast<float4*>(o2)[i] = d;
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    }
  }
  if (amax_o1) amax_publish_block(amax_o1, amx, epoch);
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

// gemm_x3.hip: the same product on the bf16 matrix cores (three bf16 pieces per fp32 operand, six MFMAs)
int gemm_f32x3_launch(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, int epi, const float* bias, const float* aux1, int64_t ld1,
                      const float* aux2, int64_t ld2, float* out2, int64_t ldo2, int nsplit, int tile_hint, void* ws,
                      size_t ws_bytes, int* nsplit_deferred, hipStream_t stream, const GemmX3Extra* ex = nullptr);

}  // namespace mapx

extern "C" size_t mapx_gemm_splitk_workspace_bytes(int M, int N, int nsplit) {
  return nsplit > 1 ? (size_t)nsplit * M * N * sizeof(float) : 0;
}

extern "C" int mapx_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, int epi,
                             const float* bias, const float* aux1, int64_t ld1, const float* aux2,
                             int64_t ld2, float* out2, int64_t ldo2, int nsplit, int tile_hint,
                             void* ws, size_t ws_bytes, int* nsplit_deferred, const mapx_gemm_scale* scale_opt,
                             hipStream_t stream) {
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
  if (nsplit < 1) nsplit = 1;
  MAPX_REQUIRE(nsplit == 1 || epi == MAPX_EPI_NONE, "gemm_f32: split-K needs EPI_NONE");
  GemmX3Extra ex{};
  ex.batch = 1;
  if (scale_opt) {
    ex.amax_a = scale_opt->amax_a; ex.amax_b = scale_opt->amax_b;
    ex.amax_c = static_cast<unsigned long long*>(scale_opt->amax_c);
    ex.amax_c2 = static_cast<unsigned long long*>(scale_opt->amax_c2);
    ex.b_planes = scale_opt->b_planes;
  }
  return gemm_f32x3_launch(a_kc, b_kc, M, N, K, A, lda, B, ldb, C, ldc, epi, bias, aux1, ld1, aux2, ld2, out2, ldo2,
                           nsplit, tile_hint, ws, ws_bytes, nsplit_deferred, stream, scale_opt ? &ex : nullptr);
}

extern "C" int mapx_gemm_f32_bwd_fused(int M, int N, int K, const float* dY, int64_t lda, const float* W, int64_t ldw,
                                       float* C, int64_t ldc, const float* add_opt, int64_t ld_add,
                                       const float* mask_opt, int64_t ld_mask, int c0, const float* x0, int64_t ld_x0,
                                       const float* u, int64_t ld_u, float* t, int64_t ld_t, float* dx0,
                                       int64_t ld_dx0, int accumulate, int plus_v, float* part, int64_t ld_part,
                                       const mapx_gemm_scale* scale_opt, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(M >= 0 && N > 0 && K > 0 && dY && W && C && part, "gemm_f32_bwd_fused: bad arguments");
  if (M == 0) return MAPX_OK;
  MAPX_REQUIRE(c0 >= 0 && c0 <= N && c0 % 4 == 0 && N % 4 == 0, "gemm_f32_bwd_fused: N and c0 must be multiples of 4");
  MAPX_REQUIRE(c0 == N || mask_opt, "gemm_f32_bwd_fused: columns >= c0 need the ReLU output");
  MAPX_REQUIRE(c0 == 0 || (x0 && u && t && dx0), "gemm_f32_bwd_fused: columns < c0 need x0, u, t and dx0");
  auto al16 = [](const void* p, int64_t ld) { return p == nullptr || ((uintptr_t)p % 16 == 0 && ld % 4 == 0); };
  MAPX_REQUIRE(al16(C, ldc) && al16(add_opt, ld_add) && al16(mask_opt, ld_mask) && al16(x0, ld_x0) && al16(u, ld_u) &&
                   al16(t, ld_t) && al16(dx0, ld_dx0) && al16(part, ld_part) && ld_part >= N,
               "gemm_f32_bwd_fused: every operand must be 16-byte aligned with a leading dimension %% 4 == 0");
  {
    const int64_t lim = (int64_t)1 << 31;
    MAPX_REQUIRE((int64_t)M * ldc < lim && (int64_t)M * ld_add < lim && (int64_t)M * ld_mask < lim,
                 "gemm_f32_bwd_fused: an operand spans 2^31 elements or more");
  }
  GemmX3Extra ex{};
  ex.aux3 = u; ex.ld3 = ld_u; ex.mask = mask_opt; ex.ldm = ld_mask; ex.out3 = t; ex.ldo3 = ld_t; ex.out4 = dx0;
  ex.ldo4 = ld_dx0; ex.c0 = c0; ex.flags = (accumulate ? 1 : 0) | (plus_v ? 2 : 0); ex.batch = 1;
  if (scale_opt) {
    ex.amax_a = scale_opt->amax_a; ex.amax_b = scale_opt->amax_b;
    ex.amax_c = static_cast<unsigned long long*>(scale_opt->amax_c);
    ex.amax_c2 = static_cast<unsigned long long*>(scale_opt->amax_c2);
    ex.b_planes = scale_opt->b_planes;
  }
  return gemm_f32x3_launch(1, 0, M, N, K, dY, lda, W, ldw, C, ldc, MAPX_EPI_BWD_FUSED, nullptr, add_opt, ld_add, x0,
                           ld_x0, part, ld_part, 1, -1, nullptr, 0, nullptr, stream, &ex);
}

extern "C" int mapx_gemm_f32_batched(int count, int a_kc, int b_kc, int M, int N, int K, const float* const* A,
                                     int64_t lda, const float* const* B, int64_t ldb, float* const* C, int nsplit,
                                     void* ws, size_t ws_bytes, const mapx_gemm_scale* scales_opt,
                                     hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(count >= 1 && count <= 4 && A && B && C, "gemm_f32_batched: 1 to 4 problems");
  MAPX_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f32_batched: empty problem");
  MAPX_REQUIRE(!(a_kc == 0 && b_kc != 0), "gemm_f32_batched: layout (A m-contiguous, B k-contiguous) unused");
  if (nsplit < 1) nsplit = 1;
  GemmX3Extra ex{};
  ex.batch = count;
  for (int z = 0; z < count; ++z) {
    MAPX_REQUIRE(A[z] && B[z] && C[z], "gemm_f32_batched: null operand");
    ex.Az[z] = A[z]; ex.Bz[z] = B[z]; ex.Cz[z] = C[z];
    if (scales_opt) { ex.amax_az[z] = scales_opt[z].amax_a; ex.amax_bz[z] = scales_opt[z].amax_b; }
  }
  if (count == 1) {
    ex.amax_a = ex.amax_az[0]; ex.amax_b = ex.amax_bz[0];
    return gemm_f32x3_launch(a_kc, b_kc, M, N, K, A[0], lda, B[0], ldb, C[0], N, MAPX_EPI_NONE, nullptr, nullptr, 0,
                             nullptr, 0, nullptr, 0, nsplit, -1, ws, ws_bytes, nullptr, stream, &ex);
  }
  int got = 0;
  // every problem with the vector-load conditions of problem 0 (same shapes and leading dimensions; bases checked)
  for (int z = 1; z < count; ++z)
    MAPX_REQUIRE(((uintptr_t)A[z] % 16 == 0) == ((uintptr_t)A[0] % 16 == 0) &&
                     ((uintptr_t)B[z] % 16 == 0) == ((uintptr_t)B[0] % 16 == 0),
                 "gemm_f32_batched: operands of one launch must share their alignment");
  const int st = gemm_f32x3_launch(a_kc, b_kc, M, N, K, A[0], lda, B[0], ldb, C[0], N, MAPX_EPI_NONE, nullptr, nullptr, 0,
                                   nullptr, 0, nullptr, 0, nsplit, -1, ws, ws_bytes, &got, stream, &ex);
  if (st != MAPX_OK) return st;
  if (got > 1) {            // slabs of problem z: ws + z * nsplit * M * N, `got` of them in use
    SumTasks t;
    memset(&t, 0, sizeof(t));
    for (int z = 0; z < count; ++z) {
      t.t[z].dst = C[z];
      t.t[z].src = static_cast<const float*>(ws) + (int64_t)z * nsplit * M * N;
      t.t[z].stride = (int64_t)M * N; t.t[z].n = (int64_t)M * N; t.t[z].nsplit = got;
    }
    hipLaunchKernelGGL(sum_tasks_kernel, dim3(96, count), dim3(256), 0, stream, t);
  }
  return check_launch("gemm_f32_batched");
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
                                    int cap_slots, float* h_slots, float* zero_slots_opt,
                                    const mapx_gemm_scale* scale_opt, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(final_act && W && bias && rowmap && tile_group && h_slots, "enc_grouped_fwd: null pointer");
  MAPX_REQUIRE(K % 4 == 0 && ld_final % 4 == 0 && ldw % 4 == 0 && cap_slots % 128 == 0,
               "enc_grouped_fwd: K, leading dimensions %% 4 and cap_slots %% 128 must be 0");
  GroupedArgs g{};
  g.A = final_act; g.lda = ld_final; g.B = W; g.ldb = ldw; g.C = h_slots; g.ldc = 32; g.bias = bias;
  g.rowmap = rowmap; g.tile_group = tile_group; g.K = K; g.N = 32; g.nrows = nrows;
  g.group_start = group_start_opt; g.F = F;
  g.zero_out = zero_slots_opt;
  MAPX_REQUIRE(K % 8 == 0 && ld_final % 4 == 0 && ldw % 4 == 0 && (uintptr_t)final_act % 16 == 0 && (uintptr_t)W % 16 == 0,
               "enc_grouped_fwd: K %% 8 != 0 or operands not 16-byte aligned (use the dense encoder GEMM)");
  static const bool h2 = [] { const char* e = getenv("MAPX_GEMM_H2"); return !e || atoi(e) != 0; }();
  if (h2 && scale_opt && scale_opt->amax_a && scale_opt->amax_b) {      // both records: the two-piece fp16 arithmetic
    g.amax_a = scale_opt->amax_a; g.amax_b = scale_opt->amax_b;
    MAPX_HIP(enc_grouped_fwd_h2_launch(g, cap_slots, stream));
  } else {
    MAPX_HIP(enc_grouped_fwd_x3_launch(g, cap_slots, stream));
  }
  return check_launch("enc_grouped_fwd");
}

extern "C" int mapx_enc_grouped_dw(const float* dh_slots, const float* final_act, int64_t ld_final, int nrows,
                                   int N, const int32_t* rowmap, const int32_t* group_start, int F,
                                   const float* gscale_opt, float* dW, int64_t ldw, const mapx_gemm_scale* scale_opt,
                                   hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dh_slots && final_act && rowmap && group_start && dW, "enc_grouped_dw: null pointer");
  MAPX_REQUIRE(N % 4 == 0 && ld_final % 4 == 0 && F >= 1, "enc_grouped_dw: N, ld %% 4 must be 0");
  GroupedArgs g{};
  g.A = dh_slots; g.lda = 32; g.B = final_act; g.ldb = ld_final; g.C = dW; g.ldc = ldw;
  g.rowmap = rowmap; g.group_start = group_start; g.F = F; g.K = 0; g.N = N; g.nrows = nrows;
  g.gscale = gscale_opt;
  MAPX_REQUIRE(N % 8 == 0 && (uintptr_t)final_act % 16 == 0 && (uintptr_t)dh_slots % 16 == 0,
               "enc_grouped_dw: N %% 8 != 0 or operands not 16-byte aligned (use the dense encoder GEMM)");
  static const bool h2 = [] { const char* e = getenv("MAPX_GEMM_H2"); return !e || atoi(e) != 0; }();
  if (h2 && scale_opt && scale_opt->amax_a && scale_opt->amax_b) {
    g.amax_a = scale_opt->amax_a; g.amax_b = scale_opt->amax_b;
    MAPX_HIP(enc_grouped_dw_h2_launch(g, F, stream));
  } else {
    MAPX_HIP(enc_grouped_dw_x3_launch(g, F, stream));
  }
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
                                     float* dz, float* db, void* ws, size_t ws_bytes, void* amax_out_opt,
                                     hipStream_t stream) {
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
                       ld_y, (const float*)nullptr, M, N, dz, (float*)nullptr, 0, part, static_cast<amax_rec*>(amax_out_opt),
                       amax_epoch_ptr());
  else
    hipLaunchKernelGGL((ew_colsum_kernel<0, 64>), dim3((N + 255) / 256, kColChunks), dim3(256), 0, stream, dy, ld_dy, y,
                       ld_y, (const float*)nullptr, M, N, dz, (float*)nullptr, 0, part, static_cast<amax_rec*>(amax_out_opt),
                       amax_epoch_ptr());
  if (db) hipLaunchKernelGGL(colsum_stage2_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, db);
  return check_launch("relu_mask_colsum");
}

extern "C" int mapx_cross_bwd_pre_colsum(const float* g, int64_t ld_g, const float* x0, const float* u, int M, int N,
                                         float* t, float* dx0, int accumulate, float* db, void* ws,
                                         size_t ws_bytes, void* amax_out_opt, hipStream_t stream) {
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
                       (int64_t)N, u, M, N, t, dx0, accumulate, part, static_cast<amax_rec*>(amax_out_opt), amax_epoch_ptr());
  else
    hipLaunchKernelGGL((ew_colsum_kernel<1, 64>), dim3((N + 255) / 256, kColChunks), dim3(256), 0, stream, g, ld_g, x0,
                       (int64_t)N, u, M, N, t, dx0, accumulate, part, static_cast<amax_rec*>(amax_out_opt), amax_epoch_ptr());
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
