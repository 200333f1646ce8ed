// Small per-step kernels around the dense trunk:
//   * BCE-with-logits loss + gradient + statistics for the RFD head (code/models.py:80-85)
//     and the CTR head (models.py:88-93);
//   * dynamic_mask on device (code/trainer.py:217-240): MFP masking and RFD/Unigram
//     replacement, with Philox-generated or caller-injected field indices.
#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {

constexpr int kLossBlocks = 256;

// loss_i = max(x,0) - x*y + log1p(exp(-|x|));  dlogit_i = (sigmoid(x) - y) / n
// stats: [0] = # (sigmoid(x) > 0.5) == y, [1] = sum(y)  (both exact integers in float)
__global__ void __launch_bounds__(256) bce_fwd_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ y, int64_t n,
                                                      float inv_n, float* __restrict__ dx,
                                                      float* __restrict__ part /*[blocks][3]*/) {
  float loss = 0.f, hit = 0.f, pos = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = x[i], yv = y[i];
    loss += fmaxf(xv, 0.f) - xv * yv + log1pf(__expf(-fabsf(xv)));
    const float sig = 1.f / (1.f + __expf(-xv));
    if (dx) dx[i] = (sig - yv) * inv_n;
    hit += ((sig > 0.5f ? 1.f : 0.f) == yv) ? 1.f : 0.f;
    pos += yv;
  }
  __shared__ float s[3][256];
  s[0][threadIdx.x] = loss; s[1][threadIdx.x] = hit; s[2][threadIdx.x] = pos;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s[0][threadIdx.x] += s[0][threadIdx.x + o];
      s[1][threadIdx.x] += s[1][threadIdx.x + o];
      s[2][threadIdx.x] += s[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x * 3 + 0] = s[0][0];
    part[blockIdx.x * 3 + 1] = s[1][0];
    part[blockIdx.x * 3 + 2] = s[2][0];
  }
}

__global__ void bce_finalize_kernel(const float* __restrict__ part, int nblocks, float inv_n,
                                    float* __restrict__ out /*[3]: loss mean, acc, pos ratio*/) {
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += kWave) {
    a += part[i * 3]; b += part[i * 3 + 1]; c += part[i * 3 + 2];
  }
  a = group_sum<kWave>(a); b = group_sum<kWave>(b); c = group_sum<kWave>(c);
  if (threadIdx.x == 0) { out[0] = a * inv_n; out[1] = b * inv_n; out[2] = c * inv_n; }
}

// MFP branch of dynamic_mask (trainer.py:224-232).  A block owns kMaskRows rows: all threads
// copy them (coalesced), then one thread per row draws / applies its L masks.
//   masked_index[b,l] ~ U{0..F-1} with replacement (sampling_method == "randint")
//   labels[b,l] = ids[b, masked_index[b,l]];  ids_out = ids with those fields set to 3
constexpr int kMaskRows = 32;
__global__ void __launch_bounds__(256) mask_mfp_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                       int F, int L,
                                                       const int64_t* __restrict__ mi_in,
                                                       uint64_t seed, uint64_t offset,
                                                       const int32_t* __restrict__ offset_dev,
                                                       int64_t* __restrict__ ids_out,
                                                       int64_t* __restrict__ labels,
                                                       int64_t* __restrict__ mi_out,
                                                       int32_t* __restrict__ keys_out,
                                                       const int64_t* __restrict__ sel,
                                                       const int64_t* __restrict__ sel_cursor, int64_t nrows,
                                                       int64_t sel_len) {
  // sel (optional): batch row b is row sel[b] of `ids` — the batch is cut out of the HBM-resident
  // split here instead of by two index kernels and two copies in front of every step
  if (offset_dev) offset += (uint64_t)(uint32_t)*offset_dev;   // graph-replay safe stream offset
  // sel_cursor (optional): `sel` is a whole epoch's permutation and the batch starts at *sel_cursor — a
  // captured step then walks the epoch by itself, with no copy of row numbers in front of each replay
  const int64_t c0 = (sel && sel_cursor) ? *sel_cursor : 0;
  // (positions outside the selection and row numbers outside the split — a device cursor that walked past its
  // permutation or diverged from the host's bookkeeping, a corrupt selection — are clamped: the caller checks its
  // cursor on the host, GraphedStep.__call__, but a stray index must not become a page fault)
  auto row_of = [&](int64_t b) {
    int64_t i = c0 + b;
    i = i < 0 ? 0 : (i < sel_len ? i : sel_len - 1);
    const int64_t r = sel[i];
    return r < 0 ? (int64_t)0 : (r < nrows ? r : nrows - 1);
  };
  for (int64_t b0 = (int64_t)blockIdx.x * kMaskRows; b0 < B; b0 += (int64_t)gridDim.x * kMaskRows) {
    const int64_t rows = (B - b0) < kMaskRows ? (B - b0) : kMaskRows;
    for (int64_t i = threadIdx.x; i < rows * F; i += blockDim.x) {
      const int64_t b = b0 + i / F;
      ids_out[b0 * F + i] = sel ? ids[row_of(b) * F + i % F] : ids[b0 * F + i];
    }
    __threadfence_block();
    __syncthreads();
    for (int64_t w = threadIdx.x; w < rows * L; w += blockDim.x) {     // draws: one per thread
      const int64_t b = b0 + w / L, l = w % L;
      int64_t f;
      if (mi_in) {
        f = mi_in[b * L + l];
      } else {
        const Philox4 r = philox4x32_10(seed, (uint64_t)(b * L + l), offset);
        f = bounded(r.x, (uint32_t)F);
      }
      if (mi_out) mi_out[b * L + l] = f;
      labels[b * L + l] = ids[(sel ? row_of(b) : b) * F + f];
      ids_out[b * F + f] = 3;  // '<mask>' (duplicates of f write the same value)
    }
    __threadfence_block();
    __syncthreads();
    if (keys_out)
      for (int64_t i = threadIdx.x; i < rows * F; i += blockDim.x) keys_out[b0 * F + i] = (int32_t)ids_out[b0 * F + i];
  }
}

// RFD branch of dynamic_mask (trainer.py:233-262), all four replacement generators:
//   mode 0 Unigram        replacement = column f of a uniformly drawn training row (x_train [N,F]
//                         resident in HBM), f = the masked field                 (:234-240)
//   mode 1 Uniform        replacement ~ U{idx_low[f] .. idx_high[f]-1}            (:241-246)
//   mode 2 Whole-Uniform  replacement ~ U{10 .. V-1}                              (:247-252)
//   mode 3 Whole-Unigram  replacement = column f' ~ U{0..F-1} of a drawn row      (:253-260)
// or caller-injected replace_in [B,L].  Duplicate fields in masked_index: the LAST l wins (CPU
// scatter order).  labels[b,f] = (ids[b,f] != ids_out[b,f]).
// One thread per (row, field) ELEMENT (round 3; before: one thread per row walked its F ids and its L dependent
// draws alone — 34 us at the head of every RFD step): reads and writes are coalesced, every thread of a row
// recomputes the row's L draws (Philox is cheap) and only the thread whose field is hit fetches the replacement.
__global__ void __launch_bounds__(256) mask_rfd_kernel(const int64_t* __restrict__ ids, int64_t B,
                                                       int F, int L,
                                                       const int64_t* __restrict__ mi_in,
                                                       const int64_t* __restrict__ replace_in,
                                                       const int64_t* __restrict__ x_train, int64_t N,
                                                       int mode, const int64_t* __restrict__ idx_low,
                                                       const int64_t* __restrict__ idx_high, int64_t V,
                                                       uint64_t seed, uint64_t offset,
                                                       const int32_t* __restrict__ offset_dev,
                                                       int64_t* __restrict__ ids_out,
                                                       float* __restrict__ labels,
                                                       int64_t* __restrict__ mi_out) {
  if (offset_dev) offset += (uint64_t)(uint32_t)*offset_dev;
  const int64_t total = B * F;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / F;
    const int f = (int)(e - b * F);
    const int64_t orig = ids[e];
    int64_t out = orig;
    for (int l = 0; l < L; ++l) {
      const Philox4 r = philox4x32_10(seed, (uint64_t)(b * L + l), offset);
      const int64_t fl = mi_in ? mi_in[b * L + l] : (int64_t)bounded(r.x, (uint32_t)F);
      if (mi_out && f == 0) mi_out[b * L + l] = fl;
      if (fl != f) continue;
      if (replace_in) {
        out = replace_in[b * L + l];
      } else {
        // 64-bit draw from two words (N and V may exceed 2^32 in principle)
        const uint64_t w = ((uint64_t)r.y << 32) | r.z;
        if (mode == 0 || mode == 3) {
          const uint64_t hi = (uint64_t)(((unsigned __int128)w * (unsigned __int128)N) >> 64);
          const int64_t col = mode == 0 ? fl : (int64_t)bounded(r.w, (uint32_t)F);
          out = x_train[(int64_t)hi * F + col];
        } else {
          const int64_t lo = mode == 1 ? idx_low[fl] : 10;
          const int64_t span = (mode == 1 ? idx_high[fl] : V) - lo;
          out = lo + (int64_t)(((unsigned __int128)w * (unsigned __int128)span) >> 64);
        }
      }
    }
    ids_out[e] = out;
    labels[e] = (orig != out) ? 1.f : 0.f;
  }
}

// ReLU backward: out = y > 0 ? dy : 0  (y = the layer's activated output)
__global__ void __launch_bounds__(256) relu_mask_kernel(const float* __restrict__ dy,
                                                        const float* __restrict__ y, int64_t n,
                                                        float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

}  // namespace mapx

extern "C" int mapx_relu_mask(const float* dy, const float* y, int64_t n, float* out,
                              hipStream_t stream) {
  MAPX_REQUIRE(dy && y && out && n >= 0, "relu_mask: bad arguments");
  if (n == 0) return MAPX_OK;
  hipLaunchKernelGGL(mapx::relu_mask_kernel, dim3(mapx::grid_for(n, 256)), dim3(256), 0, stream, dy,
                     y, n, out);
  return mapx::check_launch("relu_mask");
}

extern "C" size_t mapx_bce_workspace_bytes(void) { return mapx::kLossBlocks * 3 * sizeof(float); }

extern "C" int mapx_bce_with_logits(const float* logits, const float* labels, int64_t n,
                                    float* dlogits_opt, float* out3, void* ws, size_t ws_bytes,
                                    hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(logits && labels && out3 && n > 0, "bce_with_logits: bad arguments");
  if (!ws || ws_bytes < mapx_bce_workspace_bytes()) {
    set_error("bce_with_logits: workspace too small");
    return MAPX_EWORKSPACE;
  }
  int grid = grid_for(n, 256, kLossBlocks);
  float* part = static_cast<float*>(ws);
  const float inv_n = (float)(1.0 / (double)n);
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(grid), dim3(256), 0, stream, logits, labels, n, inv_n,
                     dlogits_opt, part);
  hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(64), 0, stream, part, grid, inv_n, out3);
  return check_launch("bce_with_logits");
}

extern "C" int mapx_dynamic_mask_mfp(const int64_t* ids, int64_t B, int F, int L,
                                     const int64_t* masked_index_in, uint64_t seed, uint64_t offset,
                                     const int32_t* offset_dev, int64_t* ids_out, int64_t* labels,
                                     int64_t* masked_index_out, int32_t* keys_out_opt, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(ids && ids_out && labels && B >= 0 && F > 0 && L >= 0, "dynamic_mask_mfp: bad arguments");
  MAPX_REQUIRE(ids != ids_out, "dynamic_mask_mfp: in-place masking is not supported");
  if (B == 0) return MAPX_OK;
  hipLaunchKernelGGL(mask_mfp_kernel, dim3(grid_for(B, kMaskRows)), dim3(256), 0, stream, ids, B, F, L,
                     masked_index_in, seed, offset, offset_dev, ids_out, labels, masked_index_out, keys_out_opt,
                     (const int64_t*)nullptr, (const int64_t*)nullptr, (int64_t)0, (int64_t)0);
  return check_launch("dynamic_mask_mfp");
}

extern "C" int mapx_dynamic_mask_mfp_rows(const int64_t* split_ids, int64_t N, const int64_t* sel, int64_t sel_len,
                                          const int64_t* sel_cursor_dev_opt, int64_t B, int F,
                                          int L, const int64_t* masked_index_in, uint64_t seed, uint64_t offset,
                                          const int32_t* offset_dev, int64_t* ids_out, int64_t* labels,
                                          int64_t* masked_index_out, int32_t* keys_out_opt, hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(split_ids && sel && ids_out && labels && B >= 0 && N > 0 && F > 0 && L >= 0 && sel_len >= 1,
               "dynamic_mask_mfp_rows: bad arguments");
  if (B == 0) return MAPX_OK;
  hipLaunchKernelGGL(mask_mfp_kernel, dim3(grid_for(B, kMaskRows)), dim3(256), 0, stream, split_ids, B, F, L,
                     masked_index_in, seed, offset, offset_dev, ids_out, labels, masked_index_out, keys_out_opt, sel,
                     sel_cursor_dev_opt, N, sel_len);
  return check_launch("dynamic_mask_mfp_rows");
}

namespace mapx {
// out[b, :] = src[sel[cursor + b], :] — the batch of an RFD / finetune step cut from the resident split inside the
// step (the MFP mask kernel reads its rows through `sel` itself).  One thread per element.
__global__ void __launch_bounds__(256) take_rows_i64_kernel(const int64_t* __restrict__ src, int64_t N, int F,
                                                            const int64_t* __restrict__ sel, int64_t sel_len,
                                                            const int64_t* __restrict__ cursor, int64_t B,
                                                            int64_t* __restrict__ out, float* __restrict__ out_f32) {
  const int64_t c0 = cursor ? *cursor : 0;
  const int64_t total = B * F;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / F;
    int64_t i = c0 + b;                             // (caller-checked; a wrong cursor must not read out of bounds:
    i = i < 0 ? 0 : (i >= sel_len ? sel_len - 1 : i);   //  neither past the selection nor, below, past the split)
    int64_t r = sel[i];
    r = r < 0 ? 0 : (r >= N ? N - 1 : r);
    const int64_t v = src[r * F + (e - b * F)];
    if (out_f32) out_f32[e] = (float)v;
    else out[e] = v;
  }
}
}  // namespace mapx

extern "C" int mapx_take_rows_i64(const int64_t* src, int64_t N, int F, const int64_t* sel, int64_t sel_len,
                                  const int64_t* sel_cursor_dev_opt, int64_t B, int64_t* out, float* out_f32_opt,
                                  hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(src && sel && (out || out_f32_opt) && N > 0 && F > 0 && B >= 0 && sel_len >= 1, "take_rows_i64: bad arguments");
  if (B == 0) return MAPX_OK;
  hipLaunchKernelGGL(take_rows_i64_kernel, dim3(grid_for(B * F, 256)), dim3(256), 0, stream, src, N, F, sel, sel_len,
                     sel_cursor_dev_opt, B, out, out_f32_opt);
  return check_launch("take_rows_i64");
}

extern "C" int mapx_dynamic_mask_rfd(const int64_t* ids, int64_t B, int F, int L,
                                     const int64_t* masked_index_in, const int64_t* replace_in,
                                     const int64_t* x_train, int64_t N, int mode,
                                     const int64_t* idx_low, const int64_t* idx_high, int64_t V,
                                     uint64_t seed, uint64_t offset, const int32_t* offset_dev,
                                     int64_t* ids_out, float* labels, int64_t* masked_index_out,
                                     hipStream_t stream) {
  using namespace mapx;
  MAPX_REQUIRE(ids && ids_out && labels && B >= 0 && F > 0 && L >= 0, "dynamic_mask_rfd: bad arguments");
  MAPX_REQUIRE(mode >= 0 && mode <= 3, "dynamic_mask_rfd: mode %d", mode);
  if (!replace_in) {
    if (mode == 0 || mode == 3) MAPX_REQUIRE(x_train && N > 0, "dynamic_mask_rfd: Unigram modes need x_train");
    if (mode == 1) MAPX_REQUIRE(idx_low && idx_high, "dynamic_mask_rfd: Uniform needs idx_low/idx_high");
    if (mode == 2) MAPX_REQUIRE(V > 10, "dynamic_mask_rfd: Whole-Uniform needs the vocabulary size");
  }
  MAPX_REQUIRE(ids != ids_out, "dynamic_mask_rfd: in-place replacement is not supported");
  if (B == 0) return MAPX_OK;
  hipLaunchKernelGGL(mask_rfd_kernel, dim3(grid_for(B * F, 256)), dim3(256), 0, stream, ids, B, F, L,
                     masked_index_in, replace_in, x_train, N, mode, idx_low, idx_high, V, seed, offset,
                     offset_dev, ids_out, labels, masked_index_out);
  return check_launch("dynamic_mask_rfd");
}
