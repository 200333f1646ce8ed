// Linear layers with at most 32 outputs — the heads' last layers: RFD's Linear(736 -> F) (reference models.py:119-124),
// the finetune head Linear(D + H -> 1) (models.py:304, 319).  On the MFMA GEMM these cost a 64 x 64 or 128 x 128 tile
// per 23 (or 1) useful columns and the scalar (non-vectorised) operand path: 16-37 us per product for 0.01-0.14 GF.
// They are streaming problems — one pass over the [M, K] activations — so they run as fp32 FMA kernels here:
//   fwd  y[m, n]  = sum_k x[m, k] w[n, k] + b[n]          a wave per 4 rows, lanes over k, butterfly reduction
//   dW   dw[n, k] = sum_m dy[m, n] x[m, k]                threads over k, row chunks -> partial rows (summed by
//                                                         mapx_sum_tasks with the step's other partial sums)
//   dX   dx[m, k] = sum_n dy[m, n] w[n, k]                a thread per 4 rows x 4 columns
// K % 4 == 0, rows 16-byte aligned (host-checked).  Results are plain fp32 sums (not the six-product arithmetic of
// gemm_x3.hip): closer to the fp64 value, not bit-identical to the GEMM path.
#include "../../include/mapx_hip.h"
#include "amax.h"
#include "common.h"

namespace mapx {

constexpr int kSkinnyChunks = 128;       // row chunks of the weight gradient (= partial rows per output)

// Element access of the streaming kernels: fp32, or (bf16 compute mode: the finetune head reads the trunk's bf16
// activations and the weight's bf16 operand) bf16 widened to fp32 — products and sums are fp32 either way, like the
// bf16 MFMA's.
typedef __bf16 sk_bf16x4 __attribute__((ext_vector_type(4)));
__device__ inline float4 sk_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline float4 sk_ld4(const __bf16* p) {
  const sk_bf16x4 v = *reinterpret_cast<const sk_bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ inline float sk_ld1(const float* p) { return *p; }
__device__ inline float sk_ld1(const __bf16* p) { return (float)*p; }
__device__ inline void sk_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ inline void sk_st4(__bf16* p, float4 v) {
  sk_bf16x4 o;
  o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
  *reinterpret_cast<sk_bf16x4*>(p) = o;
}

template <int NT, class T = float>
__global__ void __launch_bounds__(256) skinny_fwd_kernel(const T* __restrict__ x, int64_t ldx,
                                                         const T* __restrict__ w, int64_t ldw,
                                                         const float* __restrict__ bias, int M, int N, int K, int relu,
                                                         float* __restrict__ y, int64_t ldy) {
  constexpr int R = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  if (r0 >= M) return;
  float acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[r][n] = 0.f;
  const T* xr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) xr[r] = x + (r0 + r < M ? r0 + r : (int64_t)M - 1) * ldx;
  for (int k = 4 * lane; k < K; k += 256) {
    float4 xv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xv[r] = sk_ld4(xr[r] + k);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (n < N) {
        const float4 wv = sk_ld4(w + (int64_t)n * ldw + k);
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc[r][n] += xv[r].x * wv.x + xv[r].y * wv.y + xv[r].z * wv.z + xv[r].w * wv.w;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float v = acc[r][n];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);       // fixed order: bit-reproducible
      acc[r][n] = v;
    }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r0 + r >= M) break;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n < N) {
          float v = acc[r][n] + (bias ? bias[n] : 0.f);
          if (relu) v = v > 0.f ? v : 0.f;
          y[(r0 + r) * ldy + n] = v;
        }
      }
    }
  }
}

template <int NT, class T = float>
__global__ void __launch_bounds__(256) skinny_dw_kernel(const T* __restrict__ dy, int64_t ldy,
                                                        const T* __restrict__ x, int64_t ldx, int M, int N, int K,
                                                        float* __restrict__ part) {
  const int k = 4 * (blockIdx.x * 256 + threadIdx.x);
  const int rows_per = (M + (int)gridDim.y - 1) / (int)gridDim.y;
  const int m0 = blockIdx.y * rows_per, m1 = (m0 + rows_per < M) ? m0 + rows_per : M;
  if (k >= K) return;
  float4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = m0; m < m1; ++m) {
    const float4 xv = sk_ld4(x + (int64_t)m * ldx + k);
    const T* __restrict__ d = dy + (int64_t)m * ldy;           // the same N values for every thread: one line
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (n < N) {
        const float g = sk_ld1(d + n);
        acc[n].x += g * xv.x; acc[n].y += g * xv.y; acc[n].z += g * xv.z; acc[n].w += g * xv.w;
      }
    }
  }
  float* __restrict__ p = part + (int64_t)blockIdx.y * N * K + k;
#pragma unroll
  for (int n = 0; n < NT; ++n)
    if (n < N) *reinterpret_cast<float4*>(p + (int64_t)n * K) = acc[n];
}

template <int NT, class T = float>
__global__ void __launch_bounds__(256) skinny_dx_kernel(const T* __restrict__ dy, int64_t ldy,
                                                        const T* __restrict__ w, int64_t ldw, int M, int N, int K,
                                                        T* __restrict__ dx, int64_t lddx) {
  constexpr int R = 4;
  const int kq = K / 4;                                        // float4 columns
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rg = t / kq;
  const int k = 4 * (int)(t - rg * kq);
  const int64_t r0 = rg * R;
  if (r0 >= M) return;
  float4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const T* dr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) dr[r] = dy + (r0 + r < M ? r0 + r : (int64_t)M - 1) * ldy;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (n < N) {
      const float4 wv = sk_ld4(w + (int64_t)n * ldw + k);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float g = sk_ld1(dr[r] + n);
        acc[r].x += g * wv.x; acc[r].y += g * wv.y; acc[r].z += g * wv.z; acc[r].w += g * wv.w;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r0 + r < M) sk_st4(dx + (r0 + r) * lddx + k, acc[r]);
}

// Wider layers (8 < N <= 32 outputs: RFD's Linear(736 -> 23)): the weight-gradient kernel above re-reads N gradients
// per row and thread (39 us at N = 23).  Tiled instead: a block takes one of kSkinnyChunks row chunks and 128 columns;
// the chunk's dy rows go to LDS, thread (k4 = t % 32, group q = t / 32) accumulates dw[q + 8 j][4 k4 ..] over the rows:
// 14.6 us against the GEMM path's 22.5 + slab sum.  (The forward of such a layer stays on the GEMM — 15.9 us; a
// 32-row LDS-tiled streaming forward measured 25.8, the butterfly one 27.9 — the host dispatch sends only N <= 8 here.)
constexpr int kTK = 128;                                   // columns per block

template <int NJ>                                          // outputs per thread: N <= 8 NJ
__global__ void __launch_bounds__(256) skinny_dw_tiled_kernel(const float* __restrict__ dy, int64_t ldy,
                                                              const float* __restrict__ x, int64_t ldx, int M, int N,
                                                              int K, float* __restrict__ part) {
  constexpr int NW = 8 * NJ, LDS_ = NW + 1;
  const int rows_per = (M + (int)gridDim.y - 1) / (int)gridDim.y;        // <= 64 (host-checked)
  __shared__ float ds[64 * LDS_];
  const int t = threadIdx.x, k4 = t & 31, q = t >> 5;
  const int m0 = blockIdx.y * rows_per, m1 = (m0 + rows_per < M) ? m0 + rows_per : M;
  const int k = blockIdx.x * kTK + 4 * k4;
  for (int e = t; e < (m1 - m0) * NW; e += 256) {
    const int row = e / NW, n = e % NW;
    ds[row * LDS_ + n] = n < N ? dy[(int64_t)(m0 + row) * ldy + n] : 0.f;
  }
  __syncthreads();
  if (k >= K) return;
  float4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = m0; m < m1; ++m) {
    const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + k);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float g = ds[(m - m0) * LDS_ + q + 8 * j];
      acc[j].x += g * xv.x; acc[j].y += g * xv.y; acc[j].z += g * xv.z; acc[j].w += g * xv.w;
    }
  }
  float* __restrict__ p = part + (int64_t)blockIdx.y * N * K + k;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (q + 8 * j < N) *reinterpret_cast<float4*>(p + (int64_t)(q + 8 * j) * K) = acc[j];
}

// Tall weight gradients with BOTH dimensions small (AutoInt's attention projections: dW [40, 16] or [40, 40] over
// B*F = 94 208 rows — 97 us each as a one-tile split-K GEMM).  N <= 64, K <= 64.  Thread (k4 = t % 16, ng = (t / 16) % 4,
// rl = t / 64): float4 column k4, outputs 16 ng .. 16 ng + 15, every fourth 64-row stage row; dy stages go through
// LDS, the four row lanes' sums meet in LDS in a fixed order.
__global__ void __launch_bounds__(256) skinny_dw_tall_kernel(const float* __restrict__ dy, int64_t ldy,
                                                             const float* __restrict__ x, int64_t ldx, int M, int N,
                                                             int K, float* __restrict__ part) {
  const int rows_per = (M + (int)gridDim.y - 1) / (int)gridDim.y;
  __shared__ __attribute__((aligned(16))) float ds[64 * 68];             // 64 rows x 64 outputs (+ 4: rows 4 banks apart)
  __shared__ __attribute__((aligned(16))) float xs[64 * 68];             // 64 rows x K <= 64 inputs
  __shared__ float4 red[3][64];
  const int t = threadIdx.x, k4 = t & 15, ng = (t >> 4) & 3, rl = t >> 6;
  const int m0 = blockIdx.y * rows_per, m1 = (m0 + rows_per < M) ? m0 + rows_per : M;
  const int k = 4 * k4;
  const bool live = k < K;
  float4 acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = m0; s0 < m1; s0 += 64) {
    const int rows = (m1 - s0 < 64) ? m1 - s0 : 64;
    __syncthreads();
    for (int e = t; e < rows * 64; e += 256) {              // both operands of the stage: coalesced, all in flight at once
      const int row = e >> 6, n = e & 63;
      ds[row * 68 + n] = n < N ? dy[(int64_t)(s0 + row) * ldy + n] : 0.f;
      xs[row * 68 + n] = n < K ? x[(int64_t)(s0 + row) * ldx + n] : 0.f;
    }
    __syncthreads();
    if (live) {
      for (int r = rl; r < rows; r += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + r * 68 + k);
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const float4 g = *reinterpret_cast<const float4*>(ds + r * 68 + 16 * ng + 4 * j4);
          const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // One v_fmac_f32 per term, by hand.  Left to the compiler this loop is 32 v_pk_fma_f32, the e == 1 ones
            // taking dy from the HIGH half of a ds_read_b128 result for their LOW lane (op_sel:[0,1,0]).  Inside a
            // captured step, beside the cross tower's MFMA kernels, the low halves of exactly those sums (outputs
            // n % 4 == 1 at columns k % 2 == 0, lanes 48-63) were off by about one row's term once in ~10 steps,
            // while eager mode and a graph of this kernel alone stayed bit-exact.
            // 45 vs 43.5 us for AutoInt's 94 208 x 40 x 16 projection.
            float4& a = acc[4 * j4 + e];
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a.x) : "v"(gg[e]), "v"(xv.x));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a.y) : "v"(gg[e]), "v"(xv.y));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a.z) : "v"(gg[e]), "v"(xv.z));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a.w) : "v"(gg[e]), "v"(xv.w));
          }
        }
      }
    }
  }
  // the four row lanes -> one sum, lane 0 last (fixed order), 16 outputs at a time through 3 KB of LDS
  float* __restrict__ p = part + (int64_t)blockIdx.y * N * K + k;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    __syncthreads();
    if (rl > 0) red[rl - 1][t & 63] = acc[j];
    __syncthreads();
    if (rl == 0) {
      float4 v = acc[j];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 o = red[q][t & 63];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      const int n = 16 * ng + j;
      if (live && n < N) *reinterpret_cast<float4*>(p + (int64_t)n * K) = v;
    }
  }
}

// The input gradient of a narrow head over DCNv2's two towers (finetune: Linear(D + H -> 1)), with both towers' first
// backward step in the same pass — what layers.join_bwd_input does with two MFMA products and their epilogues
// (gemm_x3.hip EPI_BWD_FUSED / EPI_RELU_MASK_COLSUM), here one streaming launch:  v = dz w  (N <= 8 terms), then
//   columns c <  D (cross tower):  g = v,  t = v x0,  dx0 = v u (+ v),          partial column sums of t
//   columns c >= D (deep tower):   dzr = final > 0 ? v : 0,                     partial column sums of dzr
// one partial row per 128-row tile (the layout the GEMM epilogues leave for mapx_sum_tasks).
// Block = 32 float4 columns x 8 row lanes over a 128-row tile (16 rows per thread, loads of four rows in flight); the
// row lanes' sums meet in LDS in a fixed order.  (64 columns x 4 row lanes, 192 blocks: 35 us inside the step.)
template <int NT>
__global__ void __launch_bounds__(256) skinny_join_bwd_kernel(
    const float* __restrict__ dz, int64_t lddz, const float* __restrict__ w, int64_t ldw, int M, int N, int D, int H,
    const float* __restrict__ fin, int64_t ldf, const float* __restrict__ x0, int64_t ldx0, const float* __restrict__ u,
    int64_t ldu, int plus_v, float* __restrict__ g, int64_t ldg, float* __restrict__ t, int64_t ldt,
    float* __restrict__ dx0, int64_t lddx0, float* __restrict__ dzr, int64_t lddzr, float* __restrict__ part_cross,
    float* __restrict__ part_deep, amax_rec* __restrict__ amax_t, amax_rec* __restrict__ amax_dzr,
    const int32_t* __restrict__ epoch) {
  constexpr int CL = 32, RL = 8;
  uint32_t amx_t = 0, amx_z = 0;        // max |t|, max |dzr|: what the towers' weight / input gradient products read
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c = 4 * (blockIdx.x * CL + cl);
  const int m0 = blockIdx.y * 128, m1 = (m0 + 128 < M) ? m0 + 128 : M;
  const bool live = c < D + H, cross = c < D;
  float4 wv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
    wv[n] = (live && n < N) ? *reinterpret_cast<const float4*>(w + (int64_t)n * ldw + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
#pragma unroll 4
    for (int m = m0 + rl; m < m1; m += RL) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n < N) {
          const float d = dz[(int64_t)m * lddz + n];
          v.x += d * wv[n].x; v.y += d * wv[n].y; v.z += d * wv[n].z; v.w += d * wv[n].w;
        }
      }
      if (cross) {
        const float4 a = *reinterpret_cast<const float4*>(x0 + (int64_t)m * ldx0 + c);
        const float4 b = *reinterpret_cast<const float4*>(u + (int64_t)m * ldu + c);
        const float4 tt = make_float4(v.x * a.x, v.y * a.y, v.z * a.z, v.w * a.w);
        float4 dd = make_float4(v.x * b.x, v.y * b.y, v.z * b.z, v.w * b.w);
        if (plus_v) { dd.x += v.x; dd.y += v.y; dd.z += v.z; dd.w += v.w; }
        *reinterpret_cast<float4*>(g + (int64_t)m * ldg + c) = v;
        *reinterpret_cast<float4*>(t + (int64_t)m * ldt + c) = tt;
        *reinterpret_cast<float4*>(dx0 + (int64_t)m * lddx0 + c) = dd;
        sum.x += tt.x; sum.y += tt.y; sum.z += tt.z; sum.w += tt.w;
        amx_t = amax4(amx_t, tt.x, tt.y, tt.z, tt.w);
      } else {
        const float4 f = *reinterpret_cast<const float4*>(fin + (int64_t)m * ldf + c);
        const float4 z = make_float4(f.x > 0.f ? v.x : 0.f, f.y > 0.f ? v.y : 0.f, f.z > 0.f ? v.z : 0.f,
                                     f.w > 0.f ? v.w : 0.f);
        *reinterpret_cast<float4*>(dzr + (int64_t)m * lddzr + (c - D)) = z;
        sum.x += z.x; sum.y += z.y; sum.z += z.z; sum.w += z.w;
        amx_z = amax4(amx_z, z.x, z.y, z.z, z.w);
      }
    }
  }
  if (amax_t) amax_publish_block(amax_t, amx_t, epoch);
  if (amax_dzr) amax_publish_block(amax_dzr, amx_z, epoch);
  __shared__ float4 red[RL][CL];
  red[rl][cl] = sum;
  __syncthreads();
  if (rl == 0 && live) {
    float4 s4 = red[0][cl];
#pragma unroll
    for (int k = 1; k < RL; ++k) {
      const float4 q = red[k][cl];
      s4.x += q.x; s4.y += q.y; s4.z += q.z; s4.w += q.w;
    }
    if (cross) *reinterpret_cast<float4*>(part_cross + (int64_t)blockIdx.y * D + c) = s4;
    else *reinterpret_cast<float4*>(part_deep + (int64_t)blockIdx.y * H + (c - D)) = s4;
  }
}

template <template <int> class Launch, class... A>
static bool skinny_dispatch(int N, A... a) {
  if (N <= 1) Launch<1>::go(a...);
  else if (N <= 4) Launch<4>::go(a...);
  else if (N <= 8) Launch<8>::go(a...);
  else if (N <= 16) Launch<16>::go(a...);
  else if (N <= 24) Launch<24>::go(a...);
  else if (N <= 32) Launch<32>::go(a...);
  else if (N <= 48) Launch<48>::go(a...);
  else if (N <= 64) Launch<64>::go(a...);
  else return false;
  return true;
}

template <int NT>
struct FwdLaunch {
  static void go(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N, int K,
                 int relu, float* y, int64_t ldy, hipStream_t stream) {
    hipLaunchKernelGGL(skinny_fwd_kernel<NT>, dim3(grid_for(M, 16)), dim3(256), 0, stream, x, ldx, w, ldw, bias, M, N,
                       K, relu, y, ldy);
  }
};
template <int NT>
struct DwLaunch {
  static void go(const float* dy, int64_t ldy, const float* x, int64_t ldx, int M, int N, int K, float* part,
                 int chunks, hipStream_t stream) {
    hipLaunchKernelGGL(skinny_dw_kernel<NT>, dim3(grid_for(K / 4, 256), chunks), dim3(256), 0, stream, dy, ldy,
                       x, ldx, M, N, K, part);
  }
};
template <int NT>
struct DxLaunch {
  static void go(const float* dy, int64_t ldy, const float* w, int64_t ldw, int M, int N, int K, float* dx,
                 int64_t lddx, hipStream_t stream) {
    const int64_t threads = (int64_t)((M + 3) / 4) * (K / 4);
    hipLaunchKernelGGL(skinny_dx_kernel<NT>, dim3(grid_for(threads, 256)), dim3(256), 0, stream, dy, ldy, w, ldw, M, N,
                       K, dx, lddx);
  }
};

static bool al16(const void* p, int64_t ld) { return ((uintptr_t)p % 16 == 0) && ld % 4 == 0; }

}  // namespace mapx

extern "C" int mapx_skinny_chunks(void) { return mapx::kSkinnyChunks; }

extern "C" int mapx_skinny_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias_opt,
                                      int M, int N, int K, int relu, float* y, int64_t ldy, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(x && w && y && M >= 0 && N >= 1 && N <= 32 && K >= 4 && K % 4 == 0, "skinny_linear_fwd: bad sizes");
  MAPX_REQUIRE(al16(x, ldx) && al16(w, ldw) && ldy >= N, "skinny_linear_fwd: rows of x and w must be 16-byte aligned");
  if (M == 0) return MAPX_OK;
  skinny_dispatch<FwdLaunch>(N, x, ldx, w, ldw, bias_opt, M, N, K, relu, y, ldy, stream);
  return check_launch("skinny_linear_fwd");
}

extern "C" int mapx_skinny_linear_dw(const float* dy, int64_t ldy, const float* x, int64_t ldx, int M, int N, int K,
                                     float* part, int chunks, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dy && x && part && M >= 1 && N >= 1 && N <= 64 && K >= 4 && K % 4 == 0 && ldy >= N && chunks >= 1 &&
                   chunks <= 65535, "skinny_linear_dw: bad sizes (N <= 64)");
  MAPX_REQUIRE(al16(x, ldx) && (uintptr_t)part % 16 == 0, "skinny_linear_dw: rows of x must be 16-byte aligned");
  const int rows_per = (M + chunks - 1) / chunks;
  if (K <= 64 && (N > 32 || (N > 8 && rows_per > 64)))
    hipLaunchKernelGGL(skinny_dw_tall_kernel, dim3(1, chunks), dim3(256), 0, stream, dy, ldy, x, ldx, M, N, K, part);
  else if (N > 32 && rows_per <= 64)
    hipLaunchKernelGGL(skinny_dw_tiled_kernel<8>, dim3(grid_for(K, kTK), chunks), dim3(256), 0, stream, dy, ldy, x, ldx,
                       M, N, K, part);
  else if (N > 8 && rows_per <= 64)
    hipLaunchKernelGGL(skinny_dw_tiled_kernel<4>, dim3(grid_for(K, kTK), chunks), dim3(256), 0, stream, dy, ldy, x, ldx,
                       M, N, K, part);
  else
    skinny_dispatch<DwLaunch>(N, dy, ldy, x, ldx, M, N, K, part, chunks, stream);
  return check_launch("skinny_linear_dw");
}

extern "C" int mapx_skinny_linear_dx(const float* dy, int64_t ldy, const float* w, int64_t ldw, int M, int N, int K,
                                     float* dx, int64_t lddx, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dy && w && dx && M >= 0 && N >= 1 && N <= 64 && K >= 4 && K % 4 == 0 && ldy >= N,
               "skinny_linear_dx: bad sizes");
  MAPX_REQUIRE(al16(w, ldw) && al16(dx, lddx), "skinny_linear_dx: rows of w and dx must be 16-byte aligned");
  if (M == 0) return MAPX_OK;
  skinny_dispatch<DxLaunch>(N, dy, ldy, w, ldw, M, N, K, dx, lddx, stream);
  return check_launch("skinny_linear_dx");
}

// bf16 compute mode: the same three products on bf16 activations / gradients and the weight's bf16 operand
// (N <= 8; products and sums fp32; y and the weight gradient's partial rows fp32, dx bf16).
static bool al8h(const void* p, int64_t ld) { return ((uintptr_t)p % 8 == 0) && ld % 4 == 0; }

extern "C" int mapx_skinny_linear_fwd_bf16(const mapx_bf16* x, int64_t ldx, const mapx_bf16* w, int64_t ldw,
                                           const float* bias_opt, int M, int N, int K, int relu, float* y, int64_t ldy,
                                           hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(x && w && y && M >= 0 && N >= 1 && N <= 8 && K >= 4 && K % 4 == 0, "skinny_linear_fwd_bf16: bad sizes (N <= 8)");
  MAPX_REQUIRE(al8h(x, ldx) && al8h(w, ldw) && ldy >= N, "skinny_linear_fwd_bf16: rows of x and w must be 8-byte aligned");
  if (M == 0) return MAPX_OK;
  const __bf16* xx = reinterpret_cast<const __bf16*>(x);
  const __bf16* ww = reinterpret_cast<const __bf16*>(w);
#define MAPX_SKH(NT) hipLaunchKernelGGL((skinny_fwd_kernel<NT, __bf16>), dim3(grid_for(M, 16)), dim3(256), 0, stream, xx, ldx, ww, ldw, bias_opt, M, N, K, relu, y, ldy)
  if (N == 1) MAPX_SKH(1); else if (N <= 4) MAPX_SKH(4); else MAPX_SKH(8);
#undef MAPX_SKH
  return check_launch("skinny_linear_fwd_bf16");
}

extern "C" int mapx_skinny_linear_dw_bf16(const mapx_bf16* dy, int64_t ldy, const mapx_bf16* x, int64_t ldx, int M, int N,
                                          int K, float* part, int chunks, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dy && x && part && M >= 1 && N >= 1 && N <= 8 && K >= 4 && K % 4 == 0 && ldy >= N && chunks >= 1 &&
                   chunks <= 65535, "skinny_linear_dw_bf16: bad sizes (N <= 8)");
  MAPX_REQUIRE(al8h(x, ldx) && (uintptr_t)part % 16 == 0, "skinny_linear_dw_bf16: rows of x must be 8-byte aligned");
  const __bf16* dd = reinterpret_cast<const __bf16*>(dy);
  const __bf16* xx = reinterpret_cast<const __bf16*>(x);
#define MAPX_SKH(NT) hipLaunchKernelGGL((skinny_dw_kernel<NT, __bf16>), dim3(grid_for(K / 4, 256), chunks), dim3(256), 0, stream, dd, ldy, xx, ldx, M, N, K, part)
  if (N == 1) MAPX_SKH(1); else if (N <= 4) MAPX_SKH(4); else MAPX_SKH(8);
#undef MAPX_SKH
  return check_launch("skinny_linear_dw_bf16");
}

extern "C" int mapx_skinny_linear_dx_bf16(const mapx_bf16* dy, int64_t ldy, const mapx_bf16* w, int64_t ldw, int M, int N,
                                          int K, mapx_bf16* dx, int64_t lddx, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dy && w && dx && M >= 0 && N >= 1 && N <= 8 && K >= 4 && K % 4 == 0 && ldy >= N,
               "skinny_linear_dx_bf16: bad sizes (N <= 8)");
  MAPX_REQUIRE(al8h(w, ldw) && al8h(dx, lddx), "skinny_linear_dx_bf16: rows of w and dx must be 8-byte aligned");
  if (M == 0) return MAPX_OK;
  const __bf16* dd = reinterpret_cast<const __bf16*>(dy);
  const __bf16* ww = reinterpret_cast<const __bf16*>(w);
  __bf16* out = reinterpret_cast<__bf16*>(dx);
  const int64_t threads = (int64_t)((M + 3) / 4) * (K / 4);
#define MAPX_SKH(NT) hipLaunchKernelGGL((skinny_dx_kernel<NT, __bf16>), dim3(grid_for(threads, 256)), dim3(256), 0, stream, dd, ldy, ww, ldw, M, N, K, out, lddx)
  if (N == 1) MAPX_SKH(1); else if (N <= 4) MAPX_SKH(4); else MAPX_SKH(8);
#undef MAPX_SKH
  return check_launch("skinny_linear_dx_bf16");
}

extern "C" int mapx_skinny_join_bwd(const float* dz, int64_t lddz, const float* w, int64_t ldw, int M, int N, int D,
                                    int H, const float* final_act, int64_t ldf, const float* x0, int64_t ldx0,
                                    const float* u, int64_t ldu, int plus_v, float* g, int64_t ldg, float* t,
                                    int64_t ldt, float* dx0, int64_t lddx0, float* dzr, int64_t lddzr,
                                    float* part_cross, float* part_deep, void* amax_t_opt, void* amax_dzr_opt,
                                    hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(dz && w && final_act && x0 && u && g && t && dx0 && dzr && part_cross && part_deep,
               "skinny_join_bwd: null pointer");
  MAPX_REQUIRE(M >= 1 && N >= 1 && N <= 8 && D >= 4 && H >= 4 && D % 4 == 0 && H % 4 == 0 && lddz >= N,
               "skinny_join_bwd: bad sizes");
  MAPX_REQUIRE(al16(w, ldw) && al16(final_act, ldf) && al16(x0, ldx0) && al16(u, ldu) && al16(g, ldg) && al16(t, ldt) &&
                   al16(dx0, lddx0) && al16(dzr, lddzr) && (uintptr_t)part_cross % 16 == 0 && (uintptr_t)part_deep % 16 == 0,
               "skinny_join_bwd: rows must be 16-byte aligned");
  const dim3 grid(grid_for((D + H) / 4, 32), (M + 127) / 128);
#define MAPX_SJ(NT)                                                                                                   \
  hipLaunchKernelGGL(skinny_join_bwd_kernel<NT>, grid, dim3(256), 0, stream, dz, lddz, w, ldw, M, N, D, H, final_act, \
                     ldf, x0, ldx0, u, ldu, plus_v, g, ldg, t, ldt, dx0, lddx0, dzr, lddzr, part_cross, part_deep,         \
                     static_cast<amax_rec*>(amax_t_opt), static_cast<amax_rec*>(amax_dzr_opt), amax_epoch_ptr())
  if (N == 1) MAPX_SJ(1);
  else if (N <= 4) MAPX_SJ(4);
  else MAPX_SJ(8);
#undef MAPX_SJ
  return check_launch("skinny_join_bwd");
}
