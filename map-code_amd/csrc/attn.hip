// Self-attention core of the AutoInt backbone (reference code/layers.py:724-744 inside
// MultiHeadAttention.forward :878-909) for field-sized sequences: S = Q K^T (/ sqrt(A)),
// P = softmax(S), O = P V, on G = B*H independent groups of [F, A] (F <= 64 fields, A <= 64).
// The reference splits heads with a plain .view(B*H, -1, A) of the projected [B, F, H*A] tensor, so
// a "group" is simply the g-th run of F*A consecutive floats: no transposes anywhere.
// One wave per group: Q, K, V in LDS, lane i owns query row i.  The problem is tiny (F*F*A MACs)
// and HBM-bound (4 tensors of G*F*A floats); P [G,F,F] is kept for backward.
#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

constexpr int kAttnMaxF = 64, kAttnMaxA = 64;   // one lane per field; LDS budget

// GPW groups per wave: with F <= 32 fields a wave takes TWO groups, one per half (lane & 31 = the query row): the
// one-group form left 41 of 64 lanes idle at Avazu's 23 fields.
template <int GPW>
__global__ void __launch_bounds__(64) attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int64_t G, int F, int A,
                                                      float inv_scale, float* __restrict__ o, float* __restrict__ p) {
  extern __shared__ float sm[];                 // per group: Q, K, V [F][A+1]; P [F][F+1]
  constexpr int LW = 64 / GPW;                  // lanes per group
  const int LD = A + 1, LF = F + 1;
  const int half = threadIdx.x / LW, li = threadIdx.x % LW;
  float* Qs = sm + half * (3 * F * LD + F * LF);
  float* Ks = Qs + F * LD;
  float* Vs = Ks + F * LD;
  float* Ps = Vs + F * LD;
  const int64_t g = (int64_t)blockIdx.x * GPW + half;
  const bool have = g < G;
  const int64_t base = g * F * A;
  for (int t = li; have && t < F * A; t += LW) {
    const int r = t / A, c = t - r * A;
    Qs[r * LD + c] = q[base + t];
    Ks[r * LD + c] = k[base + t];
    Vs[r * LD + c] = v[base + t];
  }
  __syncthreads();
  const int i = li;
  if (have && i < F) {
    float mx = -3.4e38f;
    for (int j = 0; j < F; ++j) {
      float s = 0.f;
      for (int a = 0; a < A; ++a) s += Qs[i * LD + a] * Ks[j * LD + a];
      s *= inv_scale;
      Ps[i * LF + j] = s;
      mx = fmaxf(mx, s);
    }
    float den = 0.f;
    for (int j = 0; j < F; ++j) {
      const float e = expf(Ps[i * LF + j] - mx);
      Ps[i * LF + j] = e;
      den += e;
    }
    const float rden = 1.f / den;
    for (int j = 0; j < F; ++j) Ps[i * LF + j] *= rden;
    for (int a = 0; a < A; ++a) {
      float acc = 0.f;
      for (int j = 0; j < F; ++j) acc += Ps[i * LF + j] * Vs[j * LD + a];
      o[base + i * A + a] = acc;
    }
  }
  __syncthreads();
  for (int t = li; have && t < F * F; t += LW) {       // coalesced copy of the probabilities for backward
    const int r = t / F, c = t - r * F;
    p[g * F * F + t] = Ps[r * LF + c];
  }
}

// dV = P^T dO;  dP = dO V^T;  dS = P (dP - rowsum(P dP)) * inv_scale;  dQ = dS K;  dK = dS^T Q
template <int GPW>
__global__ void __launch_bounds__(64) attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, const float* __restrict__ p,
                                                      const float* __restrict__ d_o, int64_t G, int F, int A,
                                                      float inv_scale, float* __restrict__ dq,
                                                      float* __restrict__ dk, float* __restrict__ dv) {
  extern __shared__ float sm[];                 // per group: K, V, Q, dO [F][A+1] each; dS [F][F+1]; P [F][F+1]
  constexpr int LW = 64 / GPW;
  const int LD = A + 1, LF = F + 1;
  const int half = threadIdx.x / LW, li = threadIdx.x % LW;
  float* Ks = sm + half * (4 * F * LD + 2 * F * LF);
  float* Vs = Ks + F * LD;
  float* Qs = Vs + F * LD;
  float* Ds = Qs + F * LD;
  float* dS = Ds + F * LD;
  float* Ps = dS + F * LF;
  const int64_t g = (int64_t)blockIdx.x * GPW + half;
  const bool have = g < G;
  const int64_t base = g * F * A;
  for (int t = li; have && t < F * A; t += LW) {
    const int r = t / A, c = t - r * A;
    Ks[r * LD + c] = k[base + t];
    Vs[r * LD + c] = v[base + t];
    Qs[r * LD + c] = q[base + t];
    Ds[r * LD + c] = d_o[base + t];
  }
  for (int t = li; have && t < F * F; t += LW) {
    const int r = t / F, c = t - r * F;
    Ps[r * LF + c] = p[g * F * F + t];
  }
  __syncthreads();
  const int i = li;
  if (have && i < F) {
    float dot = 0.f;
    for (int j = 0; j < F; ++j) {
      float dp = 0.f;
      for (int a = 0; a < A; ++a) dp += Ds[i * LD + a] * Vs[j * LD + a];
      dS[i * LF + j] = dp;
      dot += Ps[i * LF + j] * dp;
    }
    for (int j = 0; j < F; ++j) dS[i * LF + j] = Ps[i * LF + j] * (dS[i * LF + j] - dot) * inv_scale;
    for (int a = 0; a < A; ++a) {
      float s = 0.f;
      for (int j = 0; j < F; ++j) s += dS[i * LF + j] * Ks[j * LD + a];
      dq[base + i * A + a] = s;
    }
  }
  __syncthreads();
  if (have && i < F) {                           // lane i now owns key / value row i: column sums over queries
    for (int a = 0; a < A; ++a) {
      float sk = 0.f, sv = 0.f;
      for (int r = 0; r < F; ++r) {
        sk += dS[r * LF + i] * Qs[r * LD + a];
        sv += Ps[r * LF + i] * Ds[r * LD + a];
      }
      dk[base + i * A + a] = sk;
      dv[base + i * A + a] = sv;
    }
  }
}

// F = A = 64 needs 100 KB of dynamic LDS in backward: above the 64 KB default, inside the CU's 160 KB
static hipError_t raise_lds_limit() {
  static hipError_t done = [] {
    for (const void* fn : {reinterpret_cast<const void*>(&attn_fwd_kernel<1>), reinterpret_cast<const void*>(&attn_fwd_kernel<2>),
                           reinterpret_cast<const void*>(&attn_bwd_kernel<1>), reinterpret_cast<const void*>(&attn_bwd_kernel<2>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }();
  return done;
}

}  // namespace mapx

extern "C" int mapx_attn_fwd(const float* q, const float* k, const float* v, int64_t G, int F, int A, int scaled,
                             float* o, float* p, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(G >= 0 && F >= 1 && F <= kAttnMaxF && A >= 1 && A <= kAttnMaxA,
               "attn_fwd: F <= %d fields, attention size <= %d", kAttnMaxF, kAttnMaxA);
  if (G == 0) return MAPX_OK;
  MAPX_REQUIRE(q && k && v && o && p, "attn_fwd: null pointer");
  const float inv_scale = scaled ? 1.0f / sqrtf((float)A) : 1.0f;
  const size_t lds = ((size_t)3 * F * (A + 1) + (size_t)F * (F + 1)) * sizeof(float);
  MAPX_HIP(raise_lds_limit());
  if (F <= 32)
    hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((unsigned)((G + 1) / 2)), dim3(64), 2 * lds, stream, q, k, v, G, F, A,
                       inv_scale, o, p);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3((unsigned)G), dim3(64), lds, stream, q, k, v, G, F, A, inv_scale, o, p);
  return check_launch("attn_fwd");
}

extern "C" int mapx_attn_bwd(const float* q, const float* k, const float* v, const float* p, const float* d_o,
                             int64_t G, int F, int A, int scaled, float* dq, float* dk, float* dv,
                             hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(G >= 0 && F >= 1 && F <= kAttnMaxF && A >= 1 && A <= kAttnMaxA,
               "attn_bwd: F <= %d fields, attention size <= %d", kAttnMaxF, kAttnMaxA);
  if (G == 0) return MAPX_OK;
  MAPX_REQUIRE(q && k && v && p && d_o && dq && dk && dv, "attn_bwd: null pointer");
  const float inv_scale = scaled ? 1.0f / sqrtf((float)A) : 1.0f;
  const size_t lds = ((size_t)4 * F * (A + 1) + (size_t)2 * F * (F + 1)) * sizeof(float);
  MAPX_HIP(raise_lds_limit());
  if (F <= 32)
    hipLaunchKernelGGL(attn_bwd_kernel<2>, dim3((unsigned)((G + 1) / 2)), dim3(64), 2 * lds, stream, q, k, v, p, d_o, G,
                       F, A, inv_scale, dq, dk, dv);
  else
    hipLaunchKernelGGL(attn_bwd_kernel<1>, dim3((unsigned)G), dim3(64), lds, stream, q, k, v, p, d_o, G, F, A,
                       inv_scale, dq, dk, dv);
  return check_launch("attn_bwd");
}
