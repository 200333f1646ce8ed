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
// Tiling: 256 threads = 2x2 waves (one wave per SIMD); a wave owns WMT x WNT MFMA tiles of
// 32x32; block tile (64*WMT) x (64*WNT), BK = 16.  Operands are staged in LDS as
// [k][m|n] (+4 pad) so that the 32 lanes of an MFMA operand fetch read 32 consecutive
// floats (conflict-free ds_read_b32) whatever the global layout; k-contiguous global tiles
// are transposed on the way in (4 x ds_write_b32, 2-way conflict = free).  Global loads of
// tile t+1 are issued into registers before the MFMAs of tile t and written to the other
// LDS buffer after them: one barrier per K-step.  Block ids are remapped so that each
// XCD (private 4 MiB L2) works on consecutive tiles of the same A row-panel.
#include "../../include/mapx_hip.h"
#include "common.h"

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
};

constexpr int BK = 16;

template <int ROWS /*BM or BN*/, bool KC, bool VEC>
struct TileLoader {
  // ROWS*BK floats per tile, 256 threads -> ROWS/16 floats = ROWS/64 float4 per thread
  static constexpr int NV = ROWS / 64;
  float4 r[NV];

  __device__ inline void load(const float* __restrict__ g, int64_t ld, int row0, int nrows,
                              int k0, int kend) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = threadIdx.x + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        const int m = f >> 2, kc = (f & 3) << 2;
        const int gm = row0 + m, gk = k0 + kc;
        if (gm < nrows) {
          const float* p = g + (int64_t)gm * ld + gk;
          if (VEC && gk + 4 <= kend) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            if (gk + 0 < kend) v.x = p[0];
            if (gk + 1 < kend) v.y = p[1];
            if (gk + 2 < kend) v.z = p[2];
            if (gk + 3 < kend) v.w = p[3];
          }
        }
      } else {
        constexpr int PER_K = ROWS / 4;
        const int kk = f / PER_K, mc = (f % PER_K) << 2;
        const int gk = k0 + kk, gm = row0 + mc;
        if (gk < kend) {
          const float* p = g + (int64_t)gk * ld + gm;
          if (VEC && gm + 4 <= nrows) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            if (gm + 0 < nrows) v.x = p[0];
            if (gm + 1 < nrows) v.y = p[1];
            if (gm + 2 < nrows) v.z = p[2];
            if (gm + 3 < nrows) v.w = p[3];
          }
        }
      }
      r[i] = v;
    }
  }

  __device__ inline void store(float* __restrict__ s /* [BK][ROWS+4] */) const {
    constexpr int LD = ROWS + 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = threadIdx.x + i * 256;
      if (KC) {
        const int m = f >> 2, kc = (f & 3) << 2;
        s[(kc + 0) * LD + m] = r[i].x;
        s[(kc + 1) * LD + m] = r[i].y;
        s[(kc + 2) * LD + m] = r[i].z;
        s[(kc + 3) * LD + m] = r[i].w;
      } else {
        constexpr int PER_K = ROWS / 4;
        const int kk = f / PER_K, mc = (f % PER_K) << 2;
        *reinterpret_cast<float4*>(s + kk * LD + mc) = r[i];
      }
    }
  }
};

template <int WMT, int WNT, bool A_KC, bool B_KC, bool VEC>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs a) {
  constexpr int BM = 64 * WMT, BN = 64 * WNT;
  constexpr int LDA_S = BM + 4, LDB_S = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB_S];

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

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileLoader<BM, A_KC, VEC> la;
  TileLoader<BN, B_KC, VEC> lb;
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    la.load(a.A, a.lda, m0, a.M, kbeg, kend);
    lb.load(a.B, a.ldb, n0, a.N, kbeg, kend);
    la.store(As[0]);
    lb.store(Bs[0]);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      la.load(a.A, a.lda, m0, a.M, kbeg + (kt + 1) * BK, kend);
      lb.load(a.B, a.ldb, n0, a.N, kbeg + (kt + 1) * BK, kend);
    }
    const float* __restrict__ as = As[cur] + wr * 32 * WMT + l31;
    const float* __restrict__ bs = Bs[cur] + wc * 32 * WNT + l31;
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      const int k = 2 * kp + kh;
      float av[WMT], bv[WNT];
#pragma unroll
      for (int i = 0; i < WMT; ++i) av[i] = as[k * LDA_S + 32 * i];
#pragma unroll
      for (int j = 0; j < WNT; ++j) bv[j] = bs[k * LDB_S + 32 * j];
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      la.store(As[cur ^ 1]);
      lb.store(Bs[cur ^ 1]);
    }
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int n = n0 + wc * 32 * WNT + 32 * j + l31;
      if (n >= a.N) continue;
      const float bn = (a.epi >= MAPX_EPI_BIAS && a.epi <= MAPX_EPI_BIAS_CROSS) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 * WMT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m >= a.M) continue;
        float v = acc[i][j][r] + bn;
        switch (a.epi) {
          case MAPX_EPI_BIAS_RELU: v = fmaxf(v, 0.f); break;
          case MAPX_EPI_BIAS_CROSS: {
            a.out2[(int64_t)m * a.ldo2 + n] = v;
            v = a.aux1[(int64_t)m * a.ld1 + n] + a.aux2[(int64_t)m * a.ld2 + n] * v;
            break;
          }
          case MAPX_EPI_ADD: v += a.aux1[(int64_t)m * a.ld1 + n]; break;
          case MAPX_EPI_RELU_MASK: v = a.aux1[(int64_t)m * a.ld1 + n] > 0.f ? v : 0.f; break;
          default: break;
        }
        C[(int64_t)m * a.ldc + n] = v;
      }
    }
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

// column sums of X [M,N] (bias gradients): stage 1 = 32 row chunks -> partial[32][N]
constexpr int kColChunks = 32;
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

template <int WMT, int WNT, bool A_KC, bool B_KC>
static void launch_tile(const GemmArgs& a, bool vec, int nsplit, hipStream_t stream) {
  dim3 grid(a.tiles_m * a.tiles_n, nsplit);
  if (vec)
    hipLaunchKernelGGL((gemm_f32_kernel<WMT, WNT, A_KC, B_KC, true>), grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<WMT, WNT, A_KC, B_KC, false>), grid, dim3(256), 0, stream, a);
}

template <bool A_KC, bool B_KC>
static void launch_layout(GemmArgs& a, bool vec, int tile, int nsplit, hipStream_t stream) {
  if (tile == 2) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
    launch_tile<2, 2, A_KC, B_KC>(a, vec, nsplit, stream);
  } else if (tile == 1) {
    a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 63) / 64;
    launch_tile<2, 1, A_KC, B_KC>(a, vec, nsplit, stream);
  } else {
    a.tiles_m = (a.M + 63) / 64; a.tiles_n = (a.N + 63) / 64;
    launch_tile<1, 1, A_KC, B_KC>(a, vec, nsplit, stream);
  }
}

}  // namespace mapx

extern "C" size_t mapx_gemm_splitk_workspace_bytes(int M, int N, int nsplit) {
  return nsplit > 1 ? (size_t)nsplit * M * N * sizeof(float) : 0;
}

extern "C" int mapx_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, int epi,
                             const float* bias, const float* aux1, int64_t ld1, const float* aux2,
                             int64_t ld2, float* out2, int64_t ldo2, int nsplit, void* ws,
                             size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_f32: negative size");
  if (M == 0 || N == 0) return MAPX_OK;
  MAPX_REQUIRE(A && B && C, "gemm_f32: null operand");
  MAPX_REQUIRE(!(a_kc == 0 && b_kc != 0), "gemm_f32: layout (A m-contiguous, B k-contiguous) unused");
  MAPX_REQUIRE(epi >= MAPX_EPI_NONE && epi <= MAPX_EPI_RELU_MASK, "gemm_f32: bad epilogue %d", epi);
  if (epi >= MAPX_EPI_BIAS && epi <= MAPX_EPI_BIAS_CROSS) MAPX_REQUIRE(bias, "gemm_f32: bias missing");
  if (epi == MAPX_EPI_BIAS_CROSS) MAPX_REQUIRE(aux1 && aux2 && out2, "gemm_f32: cross operands missing");
  if (epi == MAPX_EPI_ADD || epi == MAPX_EPI_RELU_MASK) MAPX_REQUIRE(aux1, "gemm_f32: aux missing");
  if (nsplit < 1) nsplit = 1;
  MAPX_REQUIRE(nsplit == 1 || epi == MAPX_EPI_NONE, "gemm_f32: split-K needs EPI_NONE");

  GemmArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias;
  g.aux1 = aux1; g.ld1 = ld1; g.aux2 = aux2; g.ld2 = ld2; g.out2 = out2; g.ldo2 = ldo2;
  g.k_chunk = K > 0 ? K : BK; g.slab_stride = 0;
  if (nsplit > 1) {
    const size_t need = mapx_gemm_splitk_workspace_bytes(M, N, nsplit);
    if (!ws || ws_bytes < need) {
      set_error("gemm_f32: split-K workspace %zu < %zu", ws_bytes, need);
      return MAPX_EWORKSPACE;
    }
    int kc = (int)ceil_div(ceil_div(K, nsplit), BK) * BK;
    g.k_chunk = kc;
    nsplit = (int)ceil_div(K, kc);
    g.C = static_cast<float*>(ws);
    g.ldc = N;
    g.slab_stride = (int64_t)M * N;
  }
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) &&
                   ((uintptr_t)B % 16 == 0) && (g.k_chunk % 4 == 0);
  // tile choice: biggest tile that still gives every CU a block
  auto blocks = [&](int bm, int bn) { return ceil_div(M, bm) * ceil_div(N, bn) * nsplit; };
  int tile = 2;
  if (blocks(128, 128) < 224) tile = blocks(128, 64) >= 224 ? 1 : 0;
  if (a_kc && b_kc) launch_layout<true, true>(g, vec, tile, nsplit, stream);
  else if (a_kc) launch_layout<true, false>(g, vec, tile, nsplit, stream);
  else launch_layout<false, false>(g, vec, tile, nsplit, stream);
  if (nsplit > 1) {
    // slabs are dense [M,N]; combine into the caller's C (ldc must equal N for split-K)
    MAPX_REQUIRE(ldc == N, "gemm_f32: split-K output must be dense (ldc == N)");
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                       static_cast<const float*>(ws), g.slab_stride, nsplit, n, C);
  }
  return check_launch("gemm_f32");
}

extern "C" size_t mapx_colsum_workspace_bytes(int N) {
  return (size_t)mapx::kColChunks * N * sizeof(float);
}

extern "C" int mapx_colsum(const float* x, int64_t ld, int M, int N, float* out, void* ws,
                           size_t ws_bytes, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(x && out && M >= 0 && N > 0, "colsum: bad arguments");
  if (!ws || ws_bytes < mapx_colsum_workspace_bytes(N)) {
    set_error("colsum: workspace too small");
    return MAPX_EWORKSPACE;
  }
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((N + 63) / 64, kColChunks), dim3(256), 0, stream, x,
                     ld, M, N, part);
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part, N, out);
  return check_launch("colsum");
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
