// MFP head: alias-method negative sampler + fused NCE gather-dot-loss (+ its backward
// coefficients) + output-table gradient.  Reference: code/nce/alias_multinomial.py:81-97
// (draw), code/nce/nce_loss.py:79-144,158-173,201-230 (loss), code/nce/index_linear.py:68-106
// (the (K+1)-row gather per target), code/models.py:75-77 (field gather, accuracy).
//
// HBM-bound.  The reference gathers B*L*(K+1) rows of emb[V,P] into an intermediate, reads
// it again for the dot, and in backward scatters dense [V,P]+[V,1] gradients.  Here every
// table row is read ONCE per step: the forward kernel keeps the row in registers for the
// logit, the loss, and dL/dh; the table gradient is a reduce-by-key over (dlogit, h) that
// never materialises per-pair rows (segreduce.h: NceContrib).
#include "../../include/mapx_hip.h"
#include "amax.h"
#include "common.h"
#include "lazy_adam.h"
#include "segreduce.h"

namespace mapx {

struct AliasRec {  // packed Walker table: one 8-byte record per class
  float prob;
  int32_t alias;
};

__global__ void __launch_bounds__(256) alias_pack_kernel(const float* __restrict__ prob,
                                                         const int64_t* __restrict__ alias,
                                                         int64_t V, AliasRec* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = AliasRec{prob[i], (int32_t)alias[i]};
}

// idx[t, 0] = target id, idx[t, 1 + k] = k-th negative: kk ~ U{0..V-1}; keep kk with
// probability prob[kk], else alias[kk].  One Philox block serves two draws.
__global__ void __launch_bounds__(256) alias_draw_kernel(const AliasRec* __restrict__ table,
                                                         int64_t V, const int64_t* __restrict__ targets,
                                                         int64_t T, int K, uint64_t seed,
                                                         uint64_t offset,
                                                         const int32_t* __restrict__ offset_dev,
                                                         int32_t* __restrict__ idx) {
  // offset_dev: device-resident step counter added to the stream offset, so that a captured
  // hipGraph draws fresh negatives on every replay
  if (offset_dev) offset += (uint64_t)(uint32_t)*offset_dev;
  const int K1 = K + 1;
  const int64_t pairs_per_t = (K + 1) / 2;  // pairs of negatives per target (K odd -> exact)
  const int64_t total = T * pairs_per_t;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total;
       w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = w / pairs_per_t;
    const int k0 = (int)(w - t * pairs_per_t) * 2;
    const Philox4 r = philox4x32_10(seed, (uint64_t)w, offset);
    const uint32_t kk0 = bounded(r.x, (uint32_t)V);
    const AliasRec a0 = table[kk0];
    idx[t * K1 + 1 + k0] = (unit_float(r.y) < a0.prob) ? (int32_t)kk0 : a0.alias;
    if (k0 + 1 < K) {
      const uint32_t kk1 = bounded(r.z, (uint32_t)V);
      const AliasRec a1 = table[kk1];
      idx[t * K1 + 2 + k0] = (unit_float(r.w) < a1.prob) ? (int32_t)kk1 : a1.alias;
    }
    if (k0 == 0) idx[t * K1] = (int32_t)targets[t];
  }
}

// Injected-noise variant (parity tests): copy caller-provided int64 indices.
__global__ void __launch_bounds__(256) nce_pack_idx_kernel(const int64_t* __restrict__ targets,
                                                           const int64_t* __restrict__ noise,
                                                           int64_t T, int K, int64_t V,
                                                           int32_t* __restrict__ idx,
                                                           int* __restrict__ err) {
  const int K1 = K + 1;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < T * K1;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = e / K1;
    const int j = (int)(e - t * K1);
    const int64_t id = j == 0 ? targets[t] : noise[t * K + j - 1];
    const bool ok = (id >= 0) & (id < V);
    if (!ok && err) atomicOr(err, 1);
    idx[e] = ok ? (int32_t)id : 0;
  }
}

__device__ inline float softplus_f(float x) {
  return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
}

// LG = P/4 lanes per target.  Per target t = b*L + l:
//   h      = enc[b, masked_index[b,l]*P : +P]                       (models.py:75)
//   s_j    = <h, emb[idx_j]> + bias[idx_j] - lnV                    (index_linear.py:102, nce_loss.py:171)
//   lt_j   = s_j - logq[idx_j] - lnK                                (nce_loss.py:215)
//   loss_t = softplus(-lt_0) + sum_{j>=1} softplus(lt_j)            (nce_loss.py:217-229)
//   dlogit = (sigmoid(lt_j) - [j==0]) / T,  dh = sum_j dlogit_j * emb[idx_j]
template <int LG>
__global__ void __launch_bounds__(256) nce_fwd_kernel(
    const float* __restrict__ enc, int64_t enc_stride, const int64_t* __restrict__ masked_index,
    int L, const int32_t* __restrict__ idx, int64_t T, int K1, const float* __restrict__ emb,
    const float* __restrict__ bias, const float* __restrict__ logq, float lnV, float lnK,
    float invT, float* __restrict__ h_out, float* __restrict__ dlogit, float* __restrict__ dh,
    float* __restrict__ logits, float* __restrict__ loss_partial, int* __restrict__ acc_partial) {
  constexpr int P = LG * 4;
  constexpr int GPB = 256 / LG;  // targets per block pass
  const int sub = threadIdx.x % LG;
  const int gib = threadIdx.x / LG;
  float loss_acc = 0.f;
  int acc_acc = 0;
  for (int64_t t = (int64_t)blockIdx.x * GPB + gib; t < T; t += (int64_t)gridDim.x * GPB) {
    const int64_t b = t / L;
    const int64_t mi = masked_index[t];
    const float4 h4 =
        *reinterpret_cast<const float4*>(enc + b * enc_stride + mi * P + 4 * sub);
    *reinterpret_cast<float4*>(h_out + t * P + 4 * sub) = h4;
    float4 dh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float s0 = 0.f, smax = -INFINITY, loss_t = 0.f;
    const int32_t* ix = idx + t * K1;
#pragma unroll 2
    for (int j = 0; j < K1; ++j) {
      const int32_t id = ix[j];
      const float4 r4 = *reinterpret_cast<const float4*>(emb + (int64_t)id * P + 4 * sub);
      const float bq = bias[id];
      const float lq = logq[id];
      float part = h4.x * r4.x + h4.y * r4.y + h4.z * r4.z + h4.w * r4.w;
      part = group_sum<LG>(part);
      const float s = part + bq - lnV;
      const float lt = s - lq - lnK;
      const float sig = 1.f / (1.f + __expf(-lt));
      float d;
      if (j == 0) {
        s0 = s;
        loss_t += softplus_f(-lt);
        d = (sig - 1.f) * invT;
      } else {
        smax = fmaxf(smax, s);
        loss_t += softplus_f(lt);
        d = sig * invT;
      }
      dh4.x += d * r4.x; dh4.y += d * r4.y; dh4.z += d * r4.z; dh4.w += d * r4.w;
      if ((j % LG) == sub) {
        dlogit[t * K1 + j] = d;
        if (logits) logits[t * K1 + j] = s;
      }
    }
    *reinterpret_cast<float4*>(dh + t * P + 4 * sub) = dh4;
    if (sub == 0) {
      loss_acc += loss_t;
      acc_acc += (s0 >= smax) ? 1 : 0;  // argmax == 0, ties resolve to the first index
    }
  }
  // block reduction in a fixed order: lane-group leaders -> LDS -> thread 0
  __shared__ float sl[GPB];
  __shared__ int sa[GPB];
  if (sub == 0) { sl[gib] = loss_acc; sa[gib] = acc_acc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f;
    int a = 0;
    for (int i = 0; i < GPB; ++i) { l += sl[i]; a += sa[i]; }
    loss_partial[blockIdx.x] = l;
    acc_partial[blockIdx.x] = a;      // summed by the finalize kernel: no atomics, no zero-fill launch
  }
}

// P = 32 fast path (the reference default, proj_size=32): same arithmetic, organised for
// memory-level parallelism.  A lane group (8 lanes) handles one target; rows are processed in
// batches of 8: the 8 row loads of a batch are issued together (8 x 128 B in flight per group,
// 64 rows per wave), the 8 dot products are reduced with a transposing butterfly (7 shuffles
// for 8 rows) that leaves row j's score in lane j%8 — the lane that also loaded that row's
// bias and log q — so logit, loss term and gradient are computed once per row, not once
// per lane.
// LAZY: the table rows are read through their pending zero-gradient updates (lazy_adam.h: LazyRows) — last[id] beside
// the id's bias, the m | v record of a stale row beside the row, the gap replayed in registers; nothing is written.
template <bool LAZY>
__global__ void __launch_bounds__(256) nce_fwd_p32_kernel(
    const float* __restrict__ enc, int64_t enc_stride, const int64_t* __restrict__ masked_index,
    int L, const int32_t* __restrict__ idx, int64_t T, int K1, const float* __restrict__ emb,
    const float* __restrict__ bias, const float* __restrict__ logq, float lnV, float lnK,
    float invT, float* __restrict__ h_out, float* __restrict__ dlogit, float* __restrict__ dh,
    float* __restrict__ logits, float* __restrict__ loss_partial, int* __restrict__ acc_partial,
    const int32_t* __restrict__ hpos, float* __restrict__ dh_slots, amax_rec* __restrict__ amax_dh,
    const int32_t* __restrict__ epoch, LazyRows lz) {
  constexpr int LG = 8, P = 32, GPB = 256 / LG, MAXB = 4;   // up to 32 rows per target
  const int target = LAZY ? *lz.done : 0;
  uint32_t amx = 0;                                          // max |dh| (the grouped encoder's weight gradient reads it)
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7, gbase = lane & ~7;
  const int gib = threadIdx.x / LG;
  float loss_acc = 0.f;
  int acc_acc = 0;
  for (int64_t t = (int64_t)blockIdx.x * GPB + gib; t < T; t += (int64_t)gridDim.x * GPB) {
    // grouped encoder: `enc` holds only the needed P-blocks, one per slot; hpos[t] = slot of t
    const int64_t hoff = hpos ? (int64_t)hpos[t] * P : (t / L) * enc_stride + masked_index[t] * P;
    const float4 h4 = *reinterpret_cast<const float4*>(enc + hoff + 4 * sub);
    *reinterpret_cast<float4*>(h_out + t * P + 4 * sub) = h4;
    const int32_t* ix = idx + t * K1;
    int myid[MAXB];
    float mybq[MAXB], mylq[MAXB];
#pragma unroll
    for (int c = 0; c < MAXB; ++c) {
      const int j = 8 * c + sub;
      myid[c] = j < K1 ? ix[j] : 0;
    }
    int myfrom[MAXB];                                   // LAZY: updates already applied to row myid[c]
#pragma unroll
    for (int c = 0; c < MAXB; ++c) {
      mybq[c] = bias[myid[c]];
      mylq[c] = logq[myid[c]];
      myfrom[c] = LAZY ? lz.last[myid[c]] : 0;
    }
    if (LAZY && lz.m1) {
#pragma unroll
      for (int c = 0; c < MAXB; ++c)
        if (myfrom[c] >= 0 && myfrom[c] < target) {
          const float m1 = lz.m1[(int64_t)myid[c] * lz.ld_mv1], v1 = lz.v1[(int64_t)myid[c] * lz.ld_mv1];
          lazy_replay1(lz, mybq[c], m1, v1, myfrom[c], target);
        }
    }
    float4 dh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float loss_l = 0.f, smax_l = -INFINITY, s0 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXB; ++c) {
      if (8 * c >= K1) break;
      float4 r[8];
      int rid[8], fr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        rid[u] = __shfl(myid[c], gbase + u, kWave);
        fr[u] = LAZY ? __shfl(myfrom[c], gbase + u, kWave) : 0;
        r[u] = *reinterpret_cast<const float4*>(emb + (int64_t)rid[u] * P + 4 * sub);
      }
      if (LAZY) {
        // the stale rows' moments, all requested before the first replay
        float4 m[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (fr[u] >= 0 && fr[u] < target) {
            m[u] = *reinterpret_cast<const float4*>(lz.m0 + (int64_t)rid[u] * lz.ld_mv0 + 4 * sub);
            v[u] = *reinterpret_cast<const float4*>(lz.v0 + (int64_t)rid[u] * lz.ld_mv0 + 4 * sub);
          }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (fr[u] >= 0 && fr[u] < target) lazy_replay4(lz, r[u], m[u], v[u], fr[u], target);
      }
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)          // (spelled out: both instantiations of this kernel must round alike)
        v[u] = __builtin_fmaf(h4.w, r[u].w, __builtin_fmaf(h4.z, r[u].z, __builtin_fmaf(h4.y, r[u].y, __fmul_rn(h4.x, r[u].x))));
      // transposing butterfly: after 3 steps lane `sub` holds the full dot product of row `sub`
      float w4[4], w2[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = (sub & 4) ? v[i + 4] : v[i], send = (sub & 4) ? v[i] : v[i + 4];
        w4[i] = keep + __shfl_xor(send, 4, kWave);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = (sub & 2) ? w4[i + 2] : w4[i], send = (sub & 2) ? w4[i] : w4[i + 2];
        w2[i] = keep + __shfl_xor(send, 2, kWave);
      }
      const float keep = (sub & 1) ? w2[1] : w2[0], send = (sub & 1) ? w2[0] : w2[1];
      const float dot = keep + __shfl_xor(send, 1, kWave);
      const int j = 8 * c + sub;
      const bool live = j < K1;
      const float sc = dot + mybq[c] - lnV;
      const float lt = sc - mylq[c] - lnK;
      const float sig = 1.f / (1.f + __expf(-lt));
      float d = 0.f;
      if (live) {
        if (j == 0) {
          loss_l += softplus_f(-lt);
          d = (sig - 1.f) * invT;
        } else {
          smax_l = fmaxf(smax_l, sc);
          loss_l += softplus_f(lt);
          d = sig * invT;
        }
        dlogit[t * K1 + j] = d;
        if (logits) logits[t * K1 + j] = sc;
      }
      if (c == 0) s0 = __shfl(sc, gbase, kWave);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float du = __shfl(d, gbase + u, kWave);     // 0 for rows beyond K1
        dh4.x = __builtin_fmaf(du, r[u].x, dh4.x); dh4.y = __builtin_fmaf(du, r[u].y, dh4.y);
        dh4.z = __builtin_fmaf(du, r[u].z, dh4.z); dh4.w = __builtin_fmaf(du, r[u].w, dh4.w);
      }
    }
    *reinterpret_cast<float4*>(dh + t * P + 4 * sub) = dh4;
    if (dh_slots) *reinterpret_cast<float4*>(dh_slots + (int64_t)hpos[t] * P + 4 * sub) = dh4;
    amx = amax4(amx, dh4.x, dh4.y, dh4.z, dh4.w);
    const float loss_t = group_sum<LG>(loss_l);
    float smax = smax_l;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, kWave));
    if (sub == 0) {
      loss_acc += loss_t;
      acc_acc += (s0 >= smax) ? 1 : 0;
    }
  }
  __shared__ float sl[GPB];
  __shared__ int sa[GPB];
  if (sub == 0) { sl[gib] = loss_acc; sa[gib] = acc_acc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f;
    int a = 0;
    for (int i = 0; i < GPB; ++i) { l += sl[i]; a += sa[i]; }
    loss_partial[blockIdx.x] = l;
    acc_partial[blockIdx.x] = a;      // summed by the finalize kernel: no atomics, no zero-fill launch
  }
  if (amax_dh) amax_publish_block(amax_dh, amx, epoch);
}

__global__ void nce_loss_finalize_kernel(const float* __restrict__ partial, const int32_t* __restrict__ acc_partial,
                                         int n, float invT, float* __restrict__ loss, int32_t* __restrict__ acc) {
  // single wave, fixed order: lane i sums partial[i], partial[i+64], ...; then butterfly
  float v = 0.f;
  int a = 0;
  for (int i = threadIdx.x; i < n; i += kWave) {
    v += partial[i];
    a += acc_partial[i];
  }
  v = group_sum<kWave>(v);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if (threadIdx.x == 0) {
    loss[0] = v * invT;
    loss[1] = (float)a * invT;      // fraction of targets ranked first (the trainers log this)
    *acc = a;
  }
}

// d enc[b, f*P + p] = g * sum_{l : masked_index[b,l] == f} dh[b,l,p]   (backward of the
// field gather, models.py:75).  One thread per (b, p) walks l in order: duplicates in
// masked_index accumulate deterministically, untouched fields are written as zero.
// `partial` != null: this launch also sums the loss / accuracy partials that the forward kernel left (what
// nce_loss_finalize_kernel does, same order: lane i takes partial[i], partial[i + 64], ..., then the butterfly) —
// by the first wave of block 0, before its share of the scatter.  Inside a training step the head's backward follows
// the forward at once, and the one-wave finalize launch sat between them on the step's critical chain.
__global__ void __launch_bounds__(256) nce_scatter_dh_kernel(const float* __restrict__ dh,
                                                             const int64_t* __restrict__ masked_index,
                                                             const float* __restrict__ gscale,
                                                             int64_t B, int L, int F, int P,
                                                             float* __restrict__ denc,
                                                             const float* __restrict__ partial,
                                                             const int32_t* __restrict__ acc_partial, int n_partial,
                                                             float invT, float* __restrict__ loss,
                                                             int32_t* __restrict__ acc, amax_rec* __restrict__ amax_out,
                                                             const int32_t* __restrict__ epoch) {
  if (partial && blockIdx.x == 0 && threadIdx.x < kWave) {
    float v = 0.f;
    int a = 0;
    for (int i = threadIdx.x; i < n_partial; i += kWave) {
      v += partial[i];
      a += acc_partial[i];
    }
    v = group_sum<kWave>(v);
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (threadIdx.x == 0) {
      loss[0] = v * invT;
      loss[1] = (float)a * invT;
      *acc = a;
    }
  }
  const float g = gscale ? *gscale : 1.f;
  const int64_t total = B * P;
  uint32_t amx = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total;
       w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = w / P;
    const int p = (int)(w - b * P);
    float* row = denc + b * (int64_t)F * P;
    for (int f = 0; f < F; ++f) row[f * P + p] = 0.f;
    for (int l = 0; l < L; ++l) {
      const int64_t f = masked_index[b * L + l];
      row[f * P + p] += g * dh[(b * L + l) * P + p];
    }
    if (amax_out)
      for (int l = 0; l < L; ++l) amx = max(amx, finite_abs_bits(row[masked_index[b * L + l] * P + p]));
  }
  if (amax_out) amax_publish_block(amax_out, amx, epoch);      // max |denc| for the encoder's input-gradient product
}

// scale rows in place by a device scalar (upstream gradient of the loss)
__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ x, int64_t n,
                                                    const float* __restrict__ g) {
  const float s = *g;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    x[i] *= s;
}

constexpr int kNceBlocks = 1024;

}  // namespace mapx

extern "C" int mapx_alias_pack(const float* prob, const int64_t* alias, int64_t V, void* packed,
                               hipStream_t stream) {
  MAPX_REQUIRE(prob && alias && packed && V > 0 && V < (1LL << 31), "alias_pack: bad arguments");
  hipLaunchKernelGGL(mapx::alias_pack_kernel, dim3(mapx::grid_for(V, 256)), dim3(256), 0, stream,
                     prob, alias, V, static_cast<mapx::AliasRec*>(packed));
  return mapx::check_launch("alias_pack");
}

extern "C" int mapx_alias_draw(const void* packed, int64_t V, const int64_t* targets, int64_t T,
                               int K, uint64_t seed, uint64_t offset, const int32_t* offset_dev,
                               int32_t* idx, hipStream_t stream) {
  MAPX_REQUIRE(packed && targets && idx, "alias_draw: null pointer");
  MAPX_REQUIRE(V > 0 && V < (1LL << 31) && T >= 0 && K >= 1, "alias_draw: bad sizes");
  if (T == 0) return MAPX_OK;
  const int64_t work = T * ((K + 1) / 2);
  hipLaunchKernelGGL(mapx::alias_draw_kernel, dim3(mapx::grid_for(work, 256)), dim3(256), 0,
                     stream, static_cast<const mapx::AliasRec*>(packed), V, targets, T, K, seed,
                     offset, offset_dev, idx);
  return mapx::check_launch("alias_draw");
}

extern "C" int mapx_nce_pack_idx(const int64_t* targets, const int64_t* noise, int64_t T, int K,
                                 int64_t V, int32_t* idx, int* err_flag, hipStream_t stream) {
  MAPX_REQUIRE(targets && noise && idx && T >= 0 && K >= 1, "nce_pack_idx: bad arguments");
  if (T == 0) return MAPX_OK;
  hipLaunchKernelGGL(mapx::nce_pack_idx_kernel, dim3(mapx::grid_for(T * (K + 1), 256)), dim3(256),
                     0, stream, targets, noise, T, K, V, idx, err_flag);
  return mapx::check_launch("nce_pack_idx");
}

extern "C" size_t mapx_nce_fwd_workspace_bytes(void) { return mapx::kNceBlocks * (sizeof(float) + sizeof(int32_t)); }

extern "C" int mapx_nce_fwd(const float* enc, int64_t B, int L, int F, int P,
                            const int64_t* masked_index, const int32_t* idx, int K,
                            const float* emb, const float* bias, const float* logq, int64_t V,
                            float* h_out, float* dlogit, float* dh, float* logits_opt,
                            float* loss_out, int32_t* acc_out, void* ws, size_t ws_bytes,
                            const int32_t* hpos_opt, float* dh_slots_opt, int* partials_left_opt,
                            void* amax_dh_opt, const mapx_lazy_rows* lazy_opt, hipStream_t stream) {
  MAPX_REQUIRE(enc && masked_index && idx && emb && bias && logq && h_out && dlogit && dh &&
                   loss_out && acc_out && ws,
               "nce_fwd: null pointer");
  mapx::LazyRows lz{};
  if (lazy_opt) {
    MAPX_REQUIRE(P == 32 && K + 1 <= 32, "nce_fwd: rows are read through their pending updates for P = 32, K <= 31 only");
    if (!mapx::lazy_rows_from(lazy_opt, P, "nce_fwd", &lz)) return MAPX_EINVAL;
  }
  MAPX_REQUIRE(B >= 0 && L >= 1 && F >= 1 && K >= 1 && V > 0, "nce_fwd: bad sizes");
  MAPX_REQUIRE(P == 8 || P == 16 || P == 32 || P == 64 || P == 128,
               "nce_fwd: proj_size %d unsupported (8, 16, 32, 64, 128)", P);
  MAPX_REQUIRE(!hpos_opt || (P == 32 && K + 1 <= 32), "nce_fwd: slot-indexed hidden needs P = 32, K <= 31");
  MAPX_REQUIRE(!dh_slots_opt || hpos_opt, "nce_fwd: dh_slots needs hpos");
  if (ws_bytes < mapx_nce_fwd_workspace_bytes()) {
    mapx::set_error("nce_fwd: workspace too small");
    return MAPX_EWORKSPACE;
  }
  const int64_t T = B * L;
  if (T == 0) {
    MAPX_HIP(hipMemsetAsync(acc_out, 0, sizeof(int32_t), stream));
    MAPX_HIP(hipMemsetAsync(loss_out, 0, 2 * sizeof(float), stream));
    if (partials_left_opt) *partials_left_opt = 0;
    return MAPX_OK;
  }
  const int LG = P / 4, GPB = 256 / LG;
  int grid = (int)mapx::ceil_div(T, GPB);
  if (grid > mapx::kNceBlocks) grid = mapx::kNceBlocks;
  float* partial = static_cast<float*>(ws);
  int32_t* acc_partial = reinterpret_cast<int32_t*>(partial + mapx::kNceBlocks);
  const float lnV = (float)log((double)V), lnK = (float)log((double)K), invT = 1.0f / (float)T;
  const int64_t enc_stride = (int64_t)F * P;
#define MAPX_NCE(LG_)                                                                           \
  hipLaunchKernelGGL(mapx::nce_fwd_kernel<LG_>, dim3(grid), dim3(256), 0, stream, enc,          \
                     enc_stride, masked_index, L, idx, T, K + 1, emb, bias, logq, lnV, lnK,     \
                     invT, h_out, dlogit, dh, logits_opt, partial, acc_partial)
  switch (LG) {
    case 2: MAPX_NCE(2); break;
    case 4: MAPX_NCE(4); break;
    case 8:
      if (K + 1 <= 32 && lazy_opt)
        hipLaunchKernelGGL(mapx::nce_fwd_p32_kernel<true>, dim3(grid), dim3(256), 0, stream, enc, enc_stride,
                           masked_index, L, idx, T, K + 1, emb, bias, logq, lnV, lnK, invT, h_out, dlogit,
                           dh, logits_opt, partial, acc_partial, hpos_opt, dh_slots_opt,
                           static_cast<mapx::amax_rec*>(amax_dh_opt), mapx::amax_epoch_ptr(), lz);
      else if (K + 1 <= 32)
        hipLaunchKernelGGL(mapx::nce_fwd_p32_kernel<false>, dim3(grid), dim3(256), 0, stream, enc, enc_stride,
                           masked_index, L, idx, T, K + 1, emb, bias, logq, lnV, lnK, invT, h_out, dlogit,
                           dh, logits_opt, partial, acc_partial, hpos_opt, dh_slots_opt,
                           static_cast<mapx::amax_rec*>(amax_dh_opt), mapx::amax_epoch_ptr(), lz);
      else
        MAPX_NCE(8);
      break;
    case 16: MAPX_NCE(16); break;
    default: MAPX_NCE(32); break;
  }
#undef MAPX_NCE
  if (partials_left_opt)
    *partials_left_opt = grid;         // the caller's mapx_nce_scatter_dh sums them (ws must live until then)
  else
    hipLaunchKernelGGL(mapx::nce_loss_finalize_kernel, dim3(1), dim3(64), 0, stream, partial,
                       acc_partial, grid, invT, loss_out, acc_out);
  return mapx::check_launch("nce_fwd");
}

extern "C" int mapx_nce_scatter_dh(const float* dh, const int64_t* masked_index,
                                   const float* gscale_opt, int64_t B, int L, int F, int P,
                                   float* denc, const void* partials_ws_opt, int n_partials, float* loss_out_opt,
                                   int32_t* acc_out_opt, void* amax_out_opt, hipStream_t stream) {
  MAPX_REQUIRE(dh && masked_index && denc, "nce_scatter_dh: null pointer");
  MAPX_REQUIRE(!partials_ws_opt || (n_partials >= 1 && n_partials <= mapx::kNceBlocks && loss_out_opt && acc_out_opt),
               "nce_scatter_dh: loss totals need 1..%d partials and both outputs", mapx::kNceBlocks);
  if (B == 0) return MAPX_OK;
  const float* partial = static_cast<const float*>(partials_ws_opt);
  const int32_t* acc_partial = partial ? reinterpret_cast<const int32_t*>(partial + mapx::kNceBlocks) : nullptr;
  hipLaunchKernelGGL(mapx::nce_scatter_dh_kernel, dim3(mapx::grid_for(B * P, 256)), dim3(256), 0,
                     stream, dh, masked_index, gscale_opt, B, L, F, P, denc, partial, acc_partial, n_partials,
                     1.0f / (float)(B * L), loss_out_opt, acc_out_opt, static_cast<mapx::amax_rec*>(amax_out_opt),
                     mapx::amax_epoch_ptr());
  return mapx::check_launch("nce_scatter_dh");
}

extern "C" size_t mapx_nce_table_grad_workspace_bytes(int64_t n, int P) {
  return mapx::seg_reduce_partial_bytes(n, P, true) + 256;
}

// out_emb[u, :] / out_bias[u] = gradient rows of the unique table rows uniq[u] of the plan
// built over idx.flatten() (n = T*(K+1) keys).
extern "C" int mapx_nce_table_grad(int64_t n, const int32_t* perm, const int32_t* rank,
                                   const int32_t* seg_start, const float* dlogit, const float* h,
                                   int K, int P, const float* gscale_opt, float* out_emb, float* out_bias,
                                   void* ws, size_t ws_bytes, int32_t* zeroed_counter_opt, hipStream_t stream) {
  MAPX_REQUIRE(n >= 0, "nce_table_grad: n < 0");
  if (n == 0) return MAPX_OK;
  MAPX_REQUIRE(perm && rank && seg_start && dlogit && h && out_emb && out_bias,
               "nce_table_grad: null pointer");
  mapx::SegPlanView pl{n, perm, rank, seg_start};
  mapx::NceContrib c{dlogit, h, K + 1, P, gscale_opt, 1.f};
  return mapx::seg_reduce_launch<true>(pl, c, P, out_emb, out_bias, ws, ws_bytes, zeroed_counter_opt, stream,
                                       "nce_table_grad");
}

extern "C" int mapx_scale_inplace(float* x, int64_t n, const float* g, hipStream_t stream) {
  MAPX_REQUIRE(x && g && n >= 0, "scale_inplace: bad arguments");
  if (n == 0) return MAPX_OK;
  hipLaunchKernelGGL(mapx::scale_kernel, dim3(mapx::grid_for(n, 256)), dim3(256), 0, stream, x, n, g);
  return mapx::check_launch("scale_inplace");
}
