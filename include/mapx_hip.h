/* libmapx_hip.so — C ABI of the MI355X (gfx950) kernels behind the DCNv2 + MFP/RFD
 * pretraining hot path of CHIANGEL/MAP-CODE.
 *
 * The reference has no FFI of its own (it is pure PyTorch); each entry point below names
 * the reference Python site (file:line under /root/reference/code) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns all memory (outputs and workspaces are caller-allocated; the library
 *     never allocates or frees device memory and never synchronises the stream);
 *   - kernels are enqueued on `stream`; return value 0 = enqueued, <0 = error
 *     (MAPX_E*), message via mapx_last_error();
 *   - ids cross the boundary as int64 (the reference's dtype); inside, table row ids are
 *     int32 (V < 2^31);
 *   - row-major, fp32 everywhere unless stated.
 */
#ifndef MAPX_HIP_H_
#define MAPX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t; /* opaque outside hipcc */
#else
#include <hip/hip_runtime_api.h>
#endif

#define MAPX_ABI_VERSION 46

#define MAPX_OK 0
#define MAPX_EINVAL (-1)     /* bad argument (shape, null pointer, alignment) */
#define MAPX_EHIP (-2)       /* HIP runtime / launch error */
#define MAPX_EWORKSPACE (-3) /* caller workspace too small */

/* bf16 values cross the boundary as raw 16-bit words (the upper half of the IEEE fp32 pattern). */
typedef uint16_t mapx_bf16;
typedef struct mapx_lazy_rows mapx_lazy_rows;      /* defined with mapx_nce_fwd below */

const char* mapx_last_error(void);
int mapx_abi_version(void);

/* ------------------------------------------------------------------ embedding (a3)
 * layers.py:97-102 (nn.Embedding forward) + models.py:308 flatten: out[i,:] = table[ids[i],:].
 * ids [n] (= [B,F] flattened), table [V,E], out [n,E] (= [B, F*E]).  An out-of-range id
 * sets *err_flag (may be NULL) and yields a zero row (reference: IndexError).  amax_out_opt: the magnitude record
 * of `out` (mapx_gemm_scale below), raised with max |out|. */
int mapx_emb_gather_fwd(const int64_t* ids, int64_t n, const float* table, int64_t V, int E,
                        float* out, int* err_flag, void* amax_out_opt, const mapx_lazy_rows* lazy_opt,
                        hipStream_t stream);
/* lazy_opt (E % 4 == 0; see mapx_lazy_rows at mapx_nce_fwd): the table's lazy-AdamW state — rows are read through their
 * pending zero-gradient updates (bit-identical to a catch-up pass followed by the plain gather; nothing written). */

/* int64 ids -> int32 row keys, range-checked against V. */
int mapx_ids_to_i32(const int64_t* ids, int64_t n, int64_t V, int32_t* out, int* err_flag,
                    hipStream_t stream);

/* ------------------------------------------------------------------ sparse gradients (a3, a10)
 * Replaces aten::embedding_dense_backward / index_select backward (layers.py:86,
 * nce/index_linear.py:99-100: dense [V,E] zero-fill + index_add) by a deterministic
 * reduce-by-key.  mapx_seg_plan sorts n keys and describes the runs of equal keys:
 *   sorted_keys[n], perm[n] (sorted position -> original position), rank[n] (1-based run
 *   id), uniq[<=n] (key of each run), seg_start[<=n+1] (first sorted position of each run,
 *   closed by n), n_uniq[2] = {number of runs, 0}.  All int32 device arrays of capacity n
 *   (seg_start: n+1).  n_uniq[1] is a counter left at zero for the ONE segment reduction that
 *   consumes the plan (pass &n_uniq[1] as zeroed_counter_opt there and it needs no memset launch;
 *   pass NULL on any later reduction over the same plan). */
size_t mapx_seg_plan_workspace_bytes(int64_t n, int64_t V);
int mapx_seg_plan(const int32_t* keys, int64_t n, int64_t V, void* ws, size_t ws_bytes,
                  int32_t* sorted_keys, int32_t* perm, int32_t* rank, int32_t* uniq,
                  int32_t* seg_start, int32_t* n_uniq, hipStream_t stream);
/* The plans of `count` (<= 2) key lists — the step's tables: embedding ids and sampled NCE ids —
 * from ONE chain of launches (a sort is a dependent chain of small kernels; two sorts one after the
 * other cost two chains): block b of every launch works on the list it belongs to, so a 3-pass
 * sort of both lists is 8 launches (mapx_seg_plan: 8 per list).  All arguments are HOST arrays of `count` entries holding what
 * mapx_seg_plan takes per list; outputs are identical to `count` calls of mapx_seg_plan. */
size_t mapx_seg_plan_multi_workspace_bytes(int count, const int64_t* n, const int64_t* V);
int mapx_seg_plan_multi(int count, const int32_t* const* keys, const int64_t* n, const int64_t* V, void* ws,
                        size_t ws_bytes, int32_t* const* sorted_keys, int32_t* const* perm,
                        int32_t* const* rank, int32_t* const* uniq, int32_t* const* seg_start,
                        int32_t* const* n_uniq, hipStream_t stream);
/* The same plan for keys that are `lists` concatenated lists of `len` keys each, every list
 * ascending as UNSIGNED 32-bit values (so -1 padding sits at its end) and free of repeats except
 * the padding — the gathered per-rank messages of the data-parallel exchange.  One launch ranks
 * every key by binary search in the other lists (ties: lower list first, i.e. exactly the order
 * a stable sort of the concatenation gives) instead of 6 radix launches; outputs are identical to
 * mapx_seg_plan(keys, lists * len, ...).  Workspace: mapx_seg_plan_workspace_bytes(lists * len, 0). */
int mapx_seg_plan_merge(const int32_t* keys, int lists, int64_t len, void* ws, size_t ws_bytes,
                        int32_t* sorted_keys, int32_t* perm, int32_t* rank, int32_t* uniq,
                        int32_t* seg_start, int32_t* n_uniq, hipStream_t stream);

/* out[u,:] = sum_{j in run u} src[perm[j],:]  (W floats per row, W % 4 == 0).  Embedding
 * table gradient: src = dL/dX0 viewed [B*F, E], plan over input_ids.flatten().  src2_opt: a second
 * tensor of the same shape added to src element by element on the fly (DCNv2, models.py:308-317: X0 feeds
 * the cross tower and the deep tower, autograd adds their two dL/dX0 with an elementwise kernel). */
size_t mapx_seg_reduce_workspace_bytes(int64_t n, int W);
int mapx_seg_reduce_rows(int64_t n, const int32_t* perm, const int32_t* rank,
                         const int32_t* seg_start, const float* src, const float* src2_opt, int W, float* out,
                         void* ws, size_t ws_bytes, int32_t* zeroed_counter_opt, hipStream_t stream);

/* Same with one extra scalar per run: out_extra[u] = sum_{j in run u} extra[(perm[j] / group) *
 * extra_stride]; src rows have stride ld_src >= W.  DeepFM: the LR weight w[V,1]
 * (models.py:134) is read with the embedding's ids, so its gradient dL/dlr[b], shared by the F
 * positions of row b (group = F), is reduced with the embedding rows.  Data-parallel merge: rows
 * [n, W+4] whose column W carries the scalar (extra = src + W, extra_stride = ld_src, group = 1).
 * Workspace: mapx_seg_reduce_workspace_bytes(n, W). */
int mapx_seg_reduce_rows_extra(int64_t n, const int32_t* perm, const int32_t* rank,
                               const int32_t* seg_start, const float* src, int W, int64_t ld_src,
                               const float* extra, int group, int64_t extra_stride, float* out,
                               float* out_extra, void* ws, size_t ws_bytes, int32_t* zeroed_counter_opt,
                               hipStream_t stream);
/* Data-parallel exchange message of one table (one process per GPU; mapx/parallel.py): keys_out
 * [maxc], rows_out [maxc, W0 (+4 if rows1)] = the first *n_uniq pairs scaled by `scale`, then
 * `pad_id` with zero rows. */
int mapx_pack_sparse(const int32_t* uniq, const float* rows0, int W0, const float* rows1_opt,
                     const int32_t* n_uniq, int64_t cap, int64_t maxc, float scale, int32_t pad_id,
                     int32_t* keys_out, float* rows_out, hipStream_t stream);
/* Publishes n (<= 64) int32 device values to host-visible pinned memory from inside a stream /
 * captured graph, so that the host can read them before the rest of the stream has run (the
 * data-parallel step reads its segment counts while forward/backward still execute):
 * ++*stamp_dev; host_out[0..n) = src[0..n); host_out[n] = *stamp_dev; host_out[n+1] = sum of the
 * n+1 values before it (checked by the reader against torn reads). */
int mapx_publish_i32(const int32_t* src, int n, int32_t* stamp_dev, int32_t* host_out,
                     hipStream_t stream);
/* Host memory for mapx_publish_i32's `host_out`: fine-grained (coherent) pinned memory, so a
 * kernel's store is visible to the polling host at once — ordinary pinned memory is only
 * guaranteed to be coherent at stream synchronisation points, i.e. when the graph has ended. */
int mapx_host_alloc_coherent(size_t bytes, void** out);
int mapx_host_free(void* p);

/* ------------------------------------------------------------------ xDeepFM: CIN
 * layers.py:696-721: X_{i+1}[b,o,:] = bias[o] + sum_{h,m} W[o, h*H_i+m] X_0[b,h,:] * X_i[b,m,:], every
 * layer sum-pooled over the embedding axis.  With activations kept embedding-major
 * (Xt[b,d,m] = X[b,m,d]; rows r = (b,d), R = B*E) the layer is  Xt_{i+1} = had W^T + bias  with
 * had[r, h*H+m] = X0t[r,h] * Xt_i[r,m]  (mapx_gemm_f32 does the product); these are the pieces
 * around it: transpose out[b,c,r] = x[b,r,c]; the outer product and its backward
 * (dxi[r,m] = sum_h dhad[r,h,m] x0t[r,h]; dx0t[r,h] (+)= sum_m dhad[r,h,m] xi[r,m]); pooling
 * out[b*ld_out+o] = sum_d xt[b,d,o] and its backward dxt[b,d,o] (+)= g[b*ld_g+o]. */
int mapx_transpose_batched(const float* x, int64_t B, int R, int C, float* out, hipStream_t stream);
/* (ld_had / ld_dhad >= F*H: row stride of the Hadamard matrix; the forward zeroes columns F*H .. ld - 1, so that a stride
 * that is a multiple of 8 floats gives the 1x1 convolution's GEMM aligned rows and a vectorisable K) */
int mapx_cin_outer_fwd(const float* x0t, int F, const float* xi, int H, int64_t R, float* had, int64_t ld_had,
                       hipStream_t stream);
int mapx_cin_outer_bwd(const float* dhad, int64_t ld_dhad, const float* x0t, int F, const float* xi, int H, int64_t R,
                       float* dx0t, int accumulate_x0, float* dxi, hipStream_t stream);
int mapx_cin_pool_fwd(const float* xt, int64_t B, int E, int H, float* out, int64_t ld_out,
                      hipStream_t stream);
int mapx_cin_pool_bwd(const float* g, int64_t ld_g, int64_t B, int E, int H, float* dxt, int accumulate,
                      hipStream_t stream);

/* ------------------------------------------------------------------ DeepFM terms
 * LR  (models.py:129-143): out[b] = sum_f w[ids[b,f]]  (bias added by the caller).
 * FM  (layers.py:123-131, 'product_sum'): out[b] = 0.5 sum_e((sum_f x)^2 - sum_f x^2), s[b,e] = sum_f x;
 *     backward dx[b,f,e] = g[b] (s[b,e] - x[b,f,e]).  x [B,F,E] dense, E in {4,8,16,32,64}. */
int mapx_lr_sum_fwd(const int64_t* ids, int64_t B, int F, const float* w, int64_t V, float* out,
                    int* err_flag, hipStream_t stream);
int mapx_fm_fwd(const float* x, int64_t B, int F, int E, float* out, float* s, hipStream_t stream);
int mapx_fm_bwd(const float* g, const float* s, const float* x, int64_t B, int F, int E, float* dx,
                hipStream_t stream);

/* ------------------------------------------------------------------ AutoInt attention core
 * layers.py:724-744 on G = B*heads groups of [F, A] (the reference's .view(B*H, -1, A) head split:
 * group g = floats [g*F*A, (g+1)*F*A) of the projected tensors): P = softmax(Q K^T / (scaled ?
 * sqrt(A) : 1)) [G,F,F] (kept for backward), O = P V.  F <= 64, A <= 64. */
int mapx_attn_fwd(const float* q, const float* k, const float* v, int64_t G, int F, int A, int scaled,
                  float* o, float* p, hipStream_t stream);
int mapx_attn_bwd(const float* q, const float* k, const float* v, const float* p, const float* d_o,
                  int64_t G, int F, int A, int scaled, float* dq, float* dk, float* dv, hipStream_t stream);

/* ------------------------------------------------------------------ NCE sampler (a7, a8)
 * nce/alias_multinomial.py:39-72: Walker table from the renormalised noise probabilities,
 * same visiting order and float32 arithmetic as the reference (HOST memory, one-off). */
int mapx_alias_build_host(const float* probs_host, int64_t n, float* out_prob_host,
                          int64_t* out_alias_host);
/* (prob f32[V], alias i64[V]) -> packed 8-byte records {f32 prob, i32 alias} [V]. */
int mapx_alias_pack(const float* prob, const int64_t* alias, int64_t V, void* packed,
                    hipStream_t stream);
/* nce/alias_multinomial.py:81-97 + nce_loss.py:146-156: idx[t,0] = targets[t],
 * idx[t,1+k] = k-th negative (Philox4x32-10 keyed by (seed, offset + *offset_dev); offset_dev
 * may be NULL: a device step counter keeps captured hipGraph replays on fresh streams).
 * idx int32 [T, K+1]. */
int mapx_alias_draw(const void* packed, int64_t V, const int64_t* targets, int64_t T, int K,
                    uint64_t seed, uint64_t offset, const int32_t* offset_dev, int32_t* idx,
                    hipStream_t stream);
/* Same index matrix from caller-provided negatives noise[T,K] (parity tests). */
int mapx_nce_pack_idx(const int64_t* targets, const int64_t* noise, int64_t T, int K, int64_t V,
                      int32_t* idx, int* err_flag, hipStream_t stream);

/* ------------------------------------------------------------------ NCE loss (a6, a9, a10)
 * models.py:74-77 + nce_loss.py:79-144,201-230 + index_linear.py:68-106, fused.
 *   enc [B, F*P] (feat_encoder output), masked_index [B,L], idx [B*L, K+1],
 *   emb [V,P], bias [V], logq [V] (= logprob_noise).
 * Outputs: h_out [B*L,P] (gathered hidden), dlogit [B*L,K+1] = dLoss/dlogit (mean over
 * B*L folded in), dh [B*L,P] = dLoss/dh, logits_opt [B*L,K+1] or NULL (the reference's
 * `logits`, i.e. score - ln V), loss_out[2] = {mean loss, fraction of targets ranked first},
 * acc_out[1] (# targets ranked first).
 * hpos_opt != NULL (grouped encoder, P = 32): `enc` is h_slots [slots, P] and target t reads
 * slot hpos_opt[t]; dh_slots_opt then also receives dh at the slot (for mapx_enc_grouped_dw). */
/* The lazy-AdamW state of the rows `emb` / `bias` name (mapx_table_adam's arguments of the same names): with it the
 * forward (and mapx_emb_gather_fwd for the embedding table) reads every sampled row THROUGH its pending zero-gradient updates — last[row] beside the row and, when the
 * row is stale, its moments; the gap replayed in registers with the arithmetic mapx_table_adam's catch-up would have
 * used (bit-identical) — and writes nothing: no catch-up pass before the forward, and the gradient update that ends
 * the step is the row's only read-modify-write (reference: every row is updated every step, trainer.py:328 over
 * index_linear.py:99-102's table).  coef_opt: mapx_replay_coef_table's output for the same *done (required, with the
 * 17-row aux).  mapx_nce_fwd: P = 32 and K + 1 <= 32 only. */
struct mapx_lazy_rows {
  const float* m0; const float* v0; int64_t ld_mv0; float wd0;
  const float* m1; const float* v1; int64_t ld_mv1; float wd1;   /* the scalar table's (bias), or NULL */
  const int32_t* last;
  const float* sched; int sched_len;
  const int32_t* done;
  const double* aux; int aux_len; int aux_rows;
  double beta1, beta2, eps;
  const float* coef_opt;
};
size_t mapx_nce_fwd_workspace_bytes(void);
int mapx_nce_fwd(const float* enc, int64_t B, int L, int F, int P, const int64_t* masked_index,
                 const int32_t* idx, int K, const float* emb, const float* bias,
                 const float* logq, int64_t V, float* h_out, float* dlogit, float* dh,
                 float* logits_opt, float* loss_out, int32_t* acc_out, void* ws, size_t ws_bytes,
                 const int32_t* hpos_opt, float* dh_slots_opt, int* partials_left_opt, void* amax_dh_opt,
                 const mapx_lazy_rows* lazy_opt, hipStream_t stream);
/* amax_dh_opt: magnitude record of dh (and dh_slots: the same values), for mapx_enc_grouped_dw.
 * partials_left_opt != NULL: the loss / accuracy totals are NOT formed by this call (loss_out / acc_out stay
 * unwritten); *partials_left_opt receives the number of per-block partials left in `ws`, which the caller keeps alive
 * and hands to the mapx_nce_scatter_dh that follows (a training step: trainer.py:317-322 runs loss.backward() right
 * behind the forward pass) — one launch less between the loss and the head's backward. */
/* Backward of the field gather: denc[b, f*P+p] = g * sum_{l: mi[b,l]==f} dh[b,l,p]; denc
 * [B, F*P] fully written.  gscale_opt: device scalar (upstream dLoss) or NULL = 1.
 * partials_ws_opt (with n_partials, loss_out_opt [2], acc_out_opt [1]): the forward's workspace when its totals
 * were left to this launch, which then writes them (same summation order as the forward's own finalize step).
 * amax_out_opt: magnitude record of denc (mapx_gemm_scale). */
int mapx_nce_scatter_dh(const float* dh, const int64_t* masked_index, const float* gscale_opt,
                        int64_t B, int L, int F, int P, float* denc, const void* partials_ws_opt, int n_partials,
                        float* loss_out_opt, int32_t* acc_out_opt, void* amax_out_opt, hipStream_t stream);
/* Output-table gradient rows for the plan over idx.flatten() (n = B*L*(K+1)):
 * out_emb[u,:] = sum dlogit*h, out_bias[u] = sum dlogit over run u. */
size_t mapx_nce_table_grad_workspace_bytes(int64_t n, int P);
int mapx_nce_table_grad(int64_t n, const int32_t* perm, const int32_t* rank,
                        const int32_t* seg_start, const float* dlogit, const float* h, int K,
                        int P, const float* gscale_opt, float* out_emb, float* out_bias, void* ws,
                        size_t ws_bytes, int32_t* zeroed_counter_opt, hipStream_t stream);
int mapx_scale_inplace(float* x, int64_t n, const float* g, hipStream_t stream);

/* ------------------------------------------------------------------ dense trunk (a4, a5, a6, a11, a12)
 * fp32 MFMA GEMM  C[m,n] = epi( sum_k A(m,k) * B(k,n) ):
 *   a_kc != 0 : A(m,k) = A[m*lda + k]   else A(m,k) = A[k*lda + m]
 *   b_kc != 0 : B(k,n) = B[n*ldb + k]   else B(k,n) = B[k*ldb + n]
 * nn.Linear forward  Y = X W^T + b            : a_kc=1, b_kc=1  (layers.py:178, models.py:74,120-122,304)
 * dX = dY W                                   : a_kc=1, b_kc=0
 * dW = dY^T X                                 : a_kc=0, b_kc=0
 * Epilogues: NONE; BIAS (+bias[n]); BIAS_RELU; BIAS_CROSS: u = acc + bias, out2 = u,
 * C = aux1 + aux2 * u  (CrossNetV2 layer, layers.py:200: aux1 = Xi, aux2 = X0);
 * ADD: C = acc + aux1; RELU_MASK: C = aux1 > 0 ? acc : 0 (ReLU backward).
 * nsplit > 1: deterministic split-K through `ws` (EPI_NONE, ldc == N).
 * tile_hint: -1 = choose from the grid size; 2 / 1 / 0 = force 128x128 / 128x64 / 64x64. */
#define MAPX_EPI_NONE 0
#define MAPX_EPI_BIAS 1
#define MAPX_EPI_BIAS_RELU 2
#define MAPX_EPI_BIAS_CROSS 3
#define MAPX_EPI_ADD 4
#define MAPX_EPI_RELU_MASK 5
/* RELU_MASK plus the column sums of the masked result as partial rows (the upstream layer's bias gradient: add the
 * rows with mapx_sum_tasks).  mapx_gemm_f32: out2 [ceil(M/64)][ldo2] fp32, one row per 64 rows of C, every row
 * written (a kernel with 128-row tiles leaves the tile's sum in the first of its two rows and zeros in the second).
 * mapx_gemm_bf16: `out2` points to FP32 partial rows [ceil(M/128)][ldo2 floats], one per 128 rows, and the sums are
 * those of the bf16 values stored.  16-byte aligned operands, no split-K; else MAPX_EINVAL. */
#define MAPX_EPI_RELU_MASK_COLSUM 6
/* nsplit_deferred != NULL: the split-K slabs stay in `ws` ([nsplit][M*N], dense) and
 * *nsplit_deferred receives the slab count (0 = C already final): the caller sums them later
 * with mapx_sum_tasks, together with every other deferred sum of the backward pass. */
size_t mapx_gemm_splitk_workspace_bytes(int M, int N, int nsplit);
/* Magnitude records (MAPX_AMAX_RECORD_BYTES of device memory each, zero-initialised by the caller once: 64 slots of
 * {fp32 bit pattern of a maximum over finite elements, epoch tag}; only ever raised, by integer atomicMax, so
 * order-independent and never reset — a later epoch outranks an earlier one; the tensor's max |x| is the maximum
 * over the slots; csrc/amax.h).  A product whose two operands come with their records (amax_a, amax_b) is formed by the two-piece
 * fp16 arithmetic (csrc/gemm_h2.hip: operands scaled by a power of two into fp16's range, hi + 2^-11 lo, three
 * MFMAs per product, the error bound of the six-product arithmetic on tensors whose values lie within 2^29 of
 * their maximum); without them, or where that kernel family does not build the case, by the six-product bf16
 * arithmetic (csrc/gemm_x3.hip), which carries fp32's exponent range per element.  amax_c / amax_c2 (optional):
 * records the launch raises with max |C| (mapx_gemm_f32_bwd_fused: of C's columns >= c0) and max |t|, for the
 * product that reads those tensors next. */
#define MAPX_AMAX_RECORD_BYTES 512
typedef struct mapx_gemm_scale {
  const float* amax_a;
  const float* amax_b;
  void* amax_c;
  void* amax_c2;
  const void* b_planes;     /* operand B pre-cut by mapx_h2_weight_planes (same orientation b_kc, same N, K), or NULL */
} mapx_gemm_scale;
/* A weight matrix cut ONCE per optimizer step into the two fp16 pieces of the two-piece arithmetic, stored in the
 * order the matrix instruction reads its B fragments (csrc/gemm_h2w.hip: blocks of 32 columns x 16 k, zero-padded),
 * with the power-of-two scale taken from its magnitude record at the time of the call.  W as operand B of
 * mapx_gemm_f32: b_kc != 0: B(k,n) = W[n*ldw + k] (forward, W [N,K]); b_kc == 0: B(k,n) = W[k*ldw + n] (input
 * gradient, W [K,N]; a column slice of a wider matrix is fine).  A product that is handed the planes
 * (mapx_gemm_scale.b_planes, with amax_a) and is large enough (>= 128 tiles of 128 x 128, A k-contiguous, no
 * split-K) reads B from them instead of cutting it per row tile; any other product ignores them. */
size_t mapx_h2_weight_planes_bytes(int N, int K);
/* ... and up to 16 matrices in one launch (the task list is HOST memory, copied into the kernel arguments). */
typedef struct mapx_plane_task {
  const float* W;
  int64_t ldw;
  int32_t N, K, b_kc, pad_;
  const void* amax_record;
  void* planes;
} mapx_plane_task;
int mapx_h2_weight_planes_multi(const mapx_plane_task* tasks_host, int ntasks, hipStream_t stream);
int mapx_h2_weight_planes(const float* W, int64_t ldw, int N, int K, int b_kc, const void* amax_record, void* planes,
                          hipStream_t stream);
/* The device int32 whose current value tags the records written from now on; mapx_step_advance adds 1 to it, so a
 * captured step never needs to reset a record.  Process-wide (one process drives one GPU); NULL: tag 0. */
int mapx_amax_epoch_source(int32_t* device_word_opt);
/* record = max(record, max |x|) over x [rows, cols] (row stride ld); reset != 0: the record is zeroed first. */
int mapx_amax_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, void* record, int reset,
                  hipStream_t stream);
int mapx_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, int epi, const float* bias,
                  const float* aux1, int64_t ld1, const float* aux2, int64_t ld2, float* out2,
                  int64_t ldo2, int nsplit, int tile_hint, void* ws, size_t ws_bytes,
                  int* nsplit_deferred, const mapx_gemm_scale* scale_opt, hipStream_t stream);
/* The input-gradient GEMM  v = dY W (+ add)  (a_kc = 1, b_kc = 0: dY [M,K], W [K,N]) whose epilogue also does the
 * elementwise backward that follows it in DCNv2's backward pass, on the tile while it is still in LDS:
 *   columns n >= c0:  v = mask[m,n] > 0 ? v : 0        ReLU backward of the layer whose OUTPUT `mask` is
 *                                                       (MLPBlock, layers.py:173-188)
 *   columns n <  c0:  t[m,n] = v x0[m,n];  dx0[m,n] = (accumulate ? dx0[m,n] : 0) + v u[m,n] (+ v if plus_v)
 *                                                       the cross layer's backward (layers.py:200:
 *                                                       X_{i+1} = X_i + X0 * u, u = W X_i + b)
 *   C[m,n] = v;  part [ceil(M/64)][ld_part]: partial rows of the column sums of (n >= c0 ? v : t[m,n]), one per 64
 *   rows of C, every row written (see MAPX_EPI_RELU_MASK_COLSUM; bias gradients: add the rows with mapx_sum_tasks).
 * c0 = 0: a ReLU layer's dZ only; c0 = N: a cross layer only; 0 < c0 < N: the concatenated input of the heads
 * (models.py:316-318), cross tower left of c0, deep tower right of it.  N, c0 % 4 == 0, 16-byte aligned operands. */
int mapx_gemm_f32_bwd_fused(int M, int N, int K, const float* dY, int64_t lda, const float* W, int64_t ldw,
                            float* C, int64_t ldc, const float* add_opt, int64_t ld_add, const float* mask_opt,
                            int64_t ld_mask, int c0, const float* x0, int64_t ld_x0, const float* u, int64_t ld_u,
                            float* t, int64_t ld_t, float* dx0, int64_t ld_dx0, int accumulate, int plus_v,
                            float* part, int64_t ld_part, const mapx_gemm_scale* scale_opt, hipStream_t stream);
/* `count` (<= 4) products of ONE shape in one launch (the cross layers' weight gradients, layers.py:197-201:
 * three 368 x 368 x 4096 products fill the GPU together, none of them alone): C[z] [M,N] dense = A[z] . B[z],
 * operands described as in mapx_gemm_f32 (a_kc / b_kc, lda, ldb).  nsplit > 1: split-K through `ws`
 * (count * mapx_gemm_splitk_workspace_bytes(M, N, nsplit) bytes), the slabs of all problems summed by one launch.
 * The pointer arrays are HOST memory; scales_opt: `count` records (amax_a, amax_b of problem z), host memory. */
int mapx_gemm_f32_batched(int count, int a_kc, int b_kc, int M, int N, int K, const float* const* A,
                          int64_t lda, const float* const* B, int64_t ldb, float* const* C, int nsplit,
                          void* ws, size_t ws_bytes, const mapx_gemm_scale* scales_opt, hipStream_t stream);
/* dst[i] = sum_{s < nsplit} src[s*stride + i], i < n, for up to 32 tasks in ONE launch.  The
 * task list is HOST memory (copied into the kernel arguments). */
typedef struct mapx_sum_task {
  float* dst;
  const float* src;
  int64_t stride;
  int64_t n;
  int32_t nsplit;
  int32_t pad_;
} mapx_sum_task;
int mapx_sum_tasks(const mapx_sum_task* tasks_host, int ntasks, hipStream_t stream);

/* Linear layers with N <= 32 outputs as fp32 FMA streaming kernels (csrc/skinny.hip) — the heads' last layers:
 * RFD's Linear(F*P -> F) (reference models.py:119-124: pred_rfd[2]) and the finetune head Linear(D+H -> 1)
 * (models.py:304, 319: fc_out).  x [M,K], w [N,K], dy / y [M,N]; K % 4 == 0, rows of x / w / dx 16-byte aligned (fwd: N <= 32; dw, dx: N <= 64).
 *   fwd: y = x w^T + bias_opt (relu != 0: max(., 0));   dx = dy w;
 *   dw:  part [chunks][N*K] <- per-row-chunk partial sums of dy^T x (chunks: mapx_skinny_chunks() = 128 for a batch
 *        of a few thousand rows; more for taller problems: a chunk is one workgroup); the caller adds the chunks with
 *        mapx_sum_tasks (stride N*K, nsplit = chunks), alone or with the step's other deferred sums.  (33..64 outputs
 *        over more than 64 columns: at most 64 rows per chunk.  AutoInt's attention projections, layers.py:724-744, are
 *        the tall case: dW [40, 16 | 40] over B*F rows.)
 * Plain fp32 sums in a fixed order (bit-reproducible), not the six-product arithmetic of mapx_gemm_f32. */
int mapx_skinny_chunks(void);
int mapx_skinny_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias_opt, int M,
                           int N, int K, int relu, float* y, int64_t ldy, hipStream_t stream);
int mapx_skinny_linear_dw(const float* dy, int64_t ldy, const float* x, int64_t ldx, int M, int N, int K, float* part,
                          int chunks, hipStream_t stream);
int mapx_skinny_linear_dx(const float* dy, int64_t ldy, const float* w, int64_t ldw, int M, int N, int K, float* dx,
                          int64_t lddx, hipStream_t stream);
/* bf16 compute mode (configs[2]; the finetune head models.py:304,319 over the bf16 trunk): the same three products on
 * bf16 activations / gradients and the weight's bf16 operand, N <= 8 outputs; products and sums fp32, y and the weight
 * gradient's partial rows fp32, dx bf16.  Rows 8-byte aligned (K, ld % 4 == 0). */
int mapx_skinny_linear_fwd_bf16(const mapx_bf16* x, int64_t ldx, const mapx_bf16* w, int64_t ldw, const float* bias_opt,
                                int M, int N, int K, int relu, float* y, int64_t ldy, hipStream_t stream);
int mapx_skinny_linear_dw_bf16(const mapx_bf16* dy, int64_t ldy, const mapx_bf16* x, int64_t ldx, int M, int N, int K,
                               float* part, int chunks, hipStream_t stream);
int mapx_skinny_linear_dx_bf16(const mapx_bf16* dy, int64_t ldy, const mapx_bf16* w, int64_t ldw, int M, int N, int K,
                               mapx_bf16* dx, int64_t lddx, hipStream_t stream);
/* dL/d(final) of a head of N <= 8 outputs over DCNv2's two towers (the finetune head; reference models.py:304, 319
 * behind models.py:306-318's concat), with both towers' first backward step in the same pass (what
 * mapx_gemm_f32_bwd_fused and mapx_gemm_f32's EPI_RELU_MASK_COLSUM do for wide heads):  v = dz w [M, D+H];
 * columns < D: g = v, t = v x0, dx0 = v u (+ v when plus_v), part_cross [ceil(M/128)][D] = column sums of t per
 * 128-row tile; columns >= D: dzr = final > 0 ? v : 0, part_deep [ceil(M/128)][H] likewise (mapx_sum_tasks adds the
 * tiles into the bias gradients).  D % 4 == H % 4 == 0, rows 16-byte aligned.  amax_t_opt / amax_dzr_opt: magnitude
 * records of t and dzr (mapx_gemm_scale). */
int mapx_skinny_join_bwd(const float* dz, int64_t lddz, const float* w, int64_t ldw, int M, int N, int D, int H,
                         const float* final_act, int64_t ldf, const float* x0, int64_t ldx0, const float* u, int64_t ldu,
                         int plus_v, float* g, int64_t ldg, float* t, int64_t ldt, float* dx0, int64_t lddx0, float* dzr,
                         int64_t lddzr, float* part_cross, float* part_deep, void* amax_t_opt, void* amax_dzr_opt,
                         hipStream_t stream);
/* row chunks of the column-sum kernels: with out/db == NULL they leave `chunks` partial rows
 * [chunks][N] in `ws` for a later mapx_sum_tasks (stride N, nsplit = chunks). */
int mapx_colsum_chunks(void);
/* Grouped feat_encoder (models.py:74-75; proj_size P = 32): only the L masked fields' P-blocks of
 * the encoder output are ever read, so targets are sorted by field and only those are computed.
 * mapx_enc_group_layout: from masked_index.flatten() (T = B*L field ids in [0, F), F <= 64; ids
 *   outside are clamped) build, in one launch (stable counting sort in LDS), the padded slot
 *   layout: rowmap[cap_slots] (batch row or -1), hpos[T] (slot of target t),
 *   tile_group[cap_slots/128] (field of each 128-slot tile or -1), group_start[F+1].
 *   Slots are ordered by field, then by target index.  cap_slots = a multiple of 128 >= T + 127*F.
 * mapx_enc_grouped_fwd: h_slots[slot, 0:32] = final[rowmap[slot], :] . W[f*32:(f+1)*32, :]^T + bias;
 *   zero_slots_opt (may be NULL): a second [cap_slots, 32] buffer to clear in the same launch
 *   (the slot-ordered dL/dh that mapx_nce_fwd fills and mapx_enc_grouped_dw reads).  group_start_opt
 *   (the layout's [F+1] array, or NULL) only orders the tiles over the workgroups — tiles that hold the
 *   same batch rows go to the same XCD's L2 — and never changes a result.
 * mapx_enc_grouped_dw:  dW[f*32 + p, :] = sum_{slot in group f} dh_slots[slot, p] * final[rowmap[slot], :]
 *   (all F*32 rows written; dh_slots must be zero in unused slots), times *gscale_opt if given.
 * scale_opt (both): amax_a / amax_b = the magnitude records of the first / second operand (fwd: final_act, W; dw:
 *   dh_slots, final_act); both given: the two-piece fp16 arithmetic (csrc/gemm_grouped_h2.hip), else six products. */
int mapx_enc_group_layout(const int64_t* masked_index, int T, int L, int F, int cap_slots, int32_t* rowmap,
                          int32_t* hpos, int32_t* tile_group, int32_t* group_start, hipStream_t stream);
int mapx_enc_grouped_fwd(const float* final_act, int64_t ld_final, int nrows, int K, const float* W,
                         int64_t ldw, const float* bias, const int32_t* rowmap,
                         const int32_t* tile_group, const int32_t* group_start_opt, int F, int cap_slots,
                         float* h_slots, float* zero_slots_opt, const mapx_gemm_scale* scale_opt, hipStream_t stream);
int mapx_enc_grouped_dw(const float* dh_slots, const float* final_act, int64_t ld_final, int nrows, int N,
                        const int32_t* rowmap, const int32_t* group_start, int F, const float* gscale_opt,
                        float* dW, int64_t ldw, const mapx_gemm_scale* scale_opt, hipStream_t stream);
/* out[n] = sum_m x[m*ld + n]  (bias gradients), deterministic two-stage. */
size_t mapx_colsum_workspace_bytes(int N);
int mapx_colsum(const float* x, int64_t ld, int M, int N, float* out, void* ws, size_t ws_bytes,
                hipStream_t stream);
/* CrossNetV2 backward, elementwise part of one layer: t = g*x0; dx0 (+)= g*u  (n % 4 == 0). */
int mapx_cross_bwd_pre(const float* g, const float* x0, const float* u, int64_t n, float* t,
                       float* dx0, int accumulate, hipStream_t stream);

/* The two elementwise backward steps fused with the bias-gradient column sum (one pass):
 * dz = y > 0 ? dy : 0, db = colsum(dz)  /  t = g*x0, dx0 (+)= g*u, db = colsum(t).  Outputs and x0, u
 * are dense [M,N]; dy, y and g carry a leading dimension (column slices of the concatenated
 * trunk output and of its gradient are read in place).  cross: `accumulate` bit 0 adds to the dx0
 * already stored, bit 1 adds g as well (layer 0, whose Xi is X0). */
/* amax_out_opt: the magnitude record of dz / t (mapx_gemm_scale), raised with the maximum of what is stored. */
int mapx_relu_mask_colsum(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int M, int N, float* dz,
                          float* db, void* ws, size_t ws_bytes, void* amax_out_opt, hipStream_t stream);
int mapx_cross_bwd_pre_colsum(const float* g, int64_t ld_g, const float* x0, const float* u, int M, int N,
                              float* t, float* dx0, int accumulate, float* db, void* ws, size_t ws_bytes,
                              void* amax_out_opt, hipStream_t stream);
/* ReLU backward: out = y > 0 ? dy : 0 (y = activated output of layers.py:178-185). */
int mapx_relu_mask(const float* dy, const float* y, int64_t n, float* out, hipStream_t stream);

/* ------------------------------------------------------------------ heads + masking (a1, a2, a11, a12)
 * BCEWithLogits mean (models.py:81,91) + dlogits (= dLoss/dlogits, may be NULL) +
 * out3 = {loss, accuracy ((sigmoid>0.5)==y mean), mean(y)}. */
size_t mapx_bce_workspace_bytes(void);
int mapx_bce_with_logits(const float* logits, const float* labels, int64_t n, float* dlogits_opt,
                         float* out3, void* ws, size_t ws_bytes, hipStream_t stream);
/* Eval metrics on the device (replaces trainer.py:189-194: host lists + sklearn roc_auc_score /
 * log_loss): out6 (device, f64) = {ROC-AUC with ties on the fp32 sigmoid, log-loss with sklearn's
 * [eps, 1-eps] clip, mean logit, mean probability, #positives, #negatives}.  AUC is NaN when
 * only one class is present (sklearn raises ValueError; the host wrapper does too). */
size_t mapx_eval_metrics_workspace_bytes(int64_t n);
int mapx_eval_metrics(const float* logits, const float* labels, int64_t n, double* out6, void* ws,
                      size_t ws_bytes, hipStream_t stream);
/* trainer.py:217-232 (MFP, sampling_method="randint"): masked_index_in NULL -> Philox.
 * keys_out_opt [B*F] int32: the masked ids once more as the int32 row keys the embedding table's
 * catch-up and segment plan take (saves the mapx_ids_to_i32 launch at the head of the step). */
int mapx_dynamic_mask_mfp(const int64_t* ids, int64_t B, int F, int L,
                          const int64_t* masked_index_in, uint64_t seed, uint64_t offset,
                          const int32_t* offset_dev, int64_t* ids_out, int64_t* labels,
                          int64_t* masked_index_out, int32_t* keys_out_opt, hipStream_t stream);
/* mapx_dynamic_mask_mfp on batch rows that are still in the HBM-resident split: batch row b is row
 * sel[b] (0 <= sel[b] < N, caller-checked) of split_ids [N,F].  Replaces the DataLoader's collate of
 * trainer.py:51-58, 306-313 (and the row-gather + copy launches a resident split otherwise needs in
 * front of every step).  Same outputs as mapx_dynamic_mask_mfp(split_ids[sel], ...).
 * sel_cursor_dev_opt (device int64, may be NULL): the batch is sel[*cursor .. *cursor + B) — `sel` is then a
 * whole epoch's permutation and a captured step walks it by itself (the caller advances the cursor). */
int mapx_dynamic_mask_mfp_rows(const int64_t* split_ids, int64_t N, const int64_t* sel, int64_t sel_len,
                               const int64_t* sel_cursor_dev_opt, int64_t B, int F, int L,
                               const int64_t* masked_index_in, uint64_t seed, uint64_t offset,
                               const int32_t* offset_dev, int64_t* ids_out, int64_t* labels,
                               int64_t* masked_index_out, int32_t* keys_out_opt, hipStream_t stream);
/* out [B,F] = rows sel[c .. c + B) of src [N,F] (c = *sel_cursor_dev_opt, 0 when NULL): the batch of an RFD or
 * finetune step cut from the HBM-resident split inside the step — the DataLoader's collate of trainer.py:51-58,
 * 122-129, 431-438 — so that a captured step walks an epoch's permutation by itself (mapx_step_advance moves the
 * cursor).  Labels: F = 1; out_f32_opt (then `out` may be NULL): the values as fp32 — the finetune step's
 * labels.float() (models.py:91) without a launch of its own.  Row numbers outside [0, N) are clamped. */
int mapx_take_rows_i64(const int64_t* src, int64_t N, int F, const int64_t* sel, int64_t sel_len,
                       const int64_t* sel_cursor_dev_opt, int64_t B, int64_t* out, float* out_f32_opt,
                       hipStream_t stream);
/* trainer.py:233-262 (RFD).  mode = RFD_replace: 0 Unigram, 1 Uniform (idx_low/idx_high [F]),
 * 2 Whole-Uniform (ids 10..V-1), 3 Whole-Unigram; x_train [N,F] device-resident; labels f32 [B,F]. */
int mapx_dynamic_mask_rfd(const int64_t* ids, int64_t B, int F, int L,
                          const int64_t* masked_index_in, const int64_t* replace_in,
                          const int64_t* x_train, int64_t N, int mode, const int64_t* idx_low,
                          const int64_t* idx_high, int64_t V, uint64_t seed, uint64_t offset,
                          const int32_t* offset_dev, int64_t* ids_out, float* labels,
                          int64_t* masked_index_out, hipStream_t stream);

/* ------------------------------------------------------------------ optimizer (a13)
 * transformers-4.26 AdamW semantics (trainer.py:60-85).  sched [sched_len][2] f32 =
 * {lr_s*sqrt(1-b2^s)/(1-b1^s), lr_s} for update s = index+1; *done = updates applied. */
/* seg_off_opt [nseg + 1] (device; element offsets of the parameters inside the flat buffer, ascending, multiples of 8,
 * seg_off[nseg] >= n) with seg_amax_opt [nseg] consecutive magnitude records (mapx_gemm_scale): the launch raises parameter z's
 * record with max |p| of what it writes, tagged for the NEXT epoch — the weights are the next step's GEMM operands. */
int mapx_adamw_dense(float* p, const float* g, float* m, float* v, int64_t n, const float* sched,
                     int sched_len, const int32_t* done, double beta1, double beta2, double eps,
                     double weight_decay, const int64_t* seg_off_opt, int nseg, void* seg_amax_opt,
                     hipStream_t stream);
/* *done += 1 (scheduler.step(), trainer.py:141,329,453); cursor_opt: the device-side batch cursor of a
 * step that walks the epoch's permutation (the DataLoader's next batch, trainer.py:306), moved by
 * cursor_stride rows in the same launch. */
int mapx_step_advance(int32_t* done, int64_t* cursor_opt, int64_t cursor_stride, hipStream_t stream);
/* Lazy exact row-sparse AdamW on a table group {p0 [V,W0] (+ optional p1 [V])} sharing
 * last[V].  rows NULL: rows row_begin..row_begin+n_rows-1 (flush / sweep); else rows[i],
 * i < *n_rows_dev (or n_rows if NULL); negative entries of rows[] are skipped (padding).  grad0 NULL: catch-up to *done only; else catch-up
 * then update *done+1 with grad0 [*, W0] (grad1 [*]) and last = *done+1.
 * aux [aux_rows][aux_len] f64 (device), the host tables of the CLOSED-FORM replay of zero-gradient
 * steps (csrc/optim.hip: replay_coef): row 0 prefix products P[s] = prod_{i<s}(1 - lr_i*wd), row 1
 * beta1^n, row 2 beta2^n, rows 3..9 R_i[s] = a_s + q_i/(1 - lr_s*wd) R_i[s+1] (a_s = step size of
 * update s+1, q_i = beta1 * beta2^(-(i+1)/2), i = 0..6), rows 10..16 the same with wd = 0.
 * aux_rows = 17: a replay of any length is O(1) per element; aux_rows = 3: steps are replayed one
 * by one while the Adam term can still move p, the rest from rows 0-2.
 * rows_may_repeat (catch-up only): rows[] is the raw id list of the batch; one lane group per
 * stale row wins an atomicCAS on last[row], so no sort is needed before the forward pass.
 * ld_mv0 / ld_mv1: row strides (floats) of m0, v0 / m1, v1 — W0 / 1 for separate dense arrays; a row's two
 * moments side by side in one record: v0 = m0 + W0 with ld_mv0 = 2 W0, v1 = m1 + 1 with ld_mv1 = 2 (random rows
 * cost per access, not per byte: 2 random places per row instead of 3). */
/* The closed form's per-row coefficients for every gap that ends at *done, tabulated once per step for the readers
 * that replay rows in registers (mapx_lazy_rows.coef_opt): coef [2][aux_len][12] f32 (decayed | undecayed;
 * {P_e/P_s, beta1^n, beta2^n, T_0..T_6, 0, 0} at index `from`), entries from < *done written.  aux_rows = 17. */
size_t mapx_replay_coef_table_bytes(int aux_len);
int mapx_replay_coef_table(const double* aux, int aux_len, int aux_rows, double beta1, double beta2,
                           const int32_t* done, float* coef, hipStream_t stream);
int mapx_table_adam(float* p0, float* m0, float* v0, int64_t ld_mv0, int W0, float wd0, float* p1, float* m1,
                    float* v1, int64_t ld_mv1, float wd1, int32_t* last, const int32_t* rows, int64_t row_begin,
                    int64_t n_rows, const int32_t* n_rows_dev, const float* grad0,
                    const float* grad1, const float* sched, int sched_len, const int32_t* done,
                    const double* aux, int aux_len, int aux_rows, double beta1, double beta2, double eps,
                    int rows_may_repeat, hipStream_t stream);

/* ------------------------------------------------------------------ bf16 compute mode (g1)
 * BASELINE configs[2] ("8 x MI355X DP bf16"; north star tolerance 1e-2): the same dense products as
 * mapx_gemm_f32 — CrossNetV2 layers.py:197-201, MLPBlock layers.py:173-188, feat_encoder / pred_rfd /
 * fc_out models.py:74,119-124,304 and their backward — on v_mfma_f32_32x32x16_bf16.  A, B, aux2, out2
 * are bf16; accumulation, bias and every epilogue are fp32; C is bf16 or (c_f32) fp32 — the heads'
 * logits and every weight gradient stay fp32; aux1 is bf16 or (aux1_f32, EPI_ADD only) fp32.  Same
 * operand description (a_kc / b_kc), epilogues and split-K as mapx_gemm_f32; split-K needs
 * c_f32 and a workspace of nsplit * M * N floats (mapx_gemm_splitk_workspace_bytes). */
int mapx_gemm_bf16(int a_kc, int b_kc, int M, int N, int K, const mapx_bf16* A, int64_t lda,
                   const mapx_bf16* B, int64_t ldb, void* C, int64_t ldc, int c_f32, int epi, const float* bias,
                   const void* aux1, int64_t ld1, int aux1_f32, const mapx_bf16* aux2, int64_t ld2,
                   mapx_bf16* out2, int64_t ldo2, int nsplit, int tile_hint, void* ws, size_t ws_bytes,
                   hipStream_t stream);
/* fp32 <-> bf16 (round to nearest even) of a flat array: weight shadows outside the optimizer,
 * fp32 head gradients entering the bf16 trunk. */
int mapx_cast_f32_bf16(const float* src, int64_t n, mapx_bf16* dst, hipStream_t stream);
int mapx_cast_bf16_f32(const mapx_bf16* src, int64_t n, float* dst, hipStream_t stream);
/* layers.py:97-102 with bf16 output rows (the table stays fp32); E % 8 == 0. */
int mapx_emb_gather_fwd_bf16(const int64_t* ids, int64_t n, const float* table, int64_t V, int E,
                             mapx_bf16* out, int* err_flag, const mapx_lazy_rows* lazy_opt,
                             hipStream_t stream);
/* mapx_seg_reduce_rows over bf16 gradient rows (fp32 sums, fp32 output). */
int mapx_seg_reduce_rows_bf16(int64_t n, const int32_t* perm, const int32_t* rank, const int32_t* seg_start,
                              const mapx_bf16* src, const mapx_bf16* src2_opt, int W, float* out, void* ws,
                              size_t ws_bytes, int32_t* zeroed_counter_opt, hipStream_t stream);
/* bf16 counterparts of mapx_colsum / mapx_relu_mask_colsum / mapx_cross_bwd_pre_colsum /
 * mapx_relu_mask: activations and their gradients bf16, column sums (bias gradients) and the running
 * dL/dX0 of the cross tower fp32.  Any N and leading dimensions.  out / db == NULL: the chunk rows stay in `ws`
 * ([mapx_colsum_bf16_workspace_bytes(N) / (4 N)][N] fp32) for a later mapx_sum_tasks, as with mapx_colsum. */
size_t mapx_colsum_bf16_workspace_bytes(int N);
int mapx_colsum_bf16(const mapx_bf16* x, int64_t ld, int M, int N, float* out, void* ws, size_t ws_bytes,
                     hipStream_t stream);
int mapx_relu_mask_colsum_bf16(const mapx_bf16* dy, int64_t ld_dy, const mapx_bf16* y, int64_t ld_y, int M, int N,
                               mapx_bf16* dz, float* db, void* ws, size_t ws_bytes, hipStream_t stream);
int mapx_cross_bwd_pre_colsum_bf16(const mapx_bf16* g, int64_t ld_g, const mapx_bf16* x0, const mapx_bf16* u, int M,
                                   int N, mapx_bf16* t, float* dx0, int accumulate, float* db, void* ws,
                                   size_t ws_bytes, hipStream_t stream);
int mapx_relu_mask_bf16(const mapx_bf16* dy, const mapx_bf16* y, int64_t n, mapx_bf16* out, hipStream_t stream);
/* mapx_adamw_dense that also stores the updated parameters as bf16 in `shadow` [n] (the weight
 * operand of mapx_gemm_bf16): trainer.py:60-85 semantics on the fp32 master weights, unchanged. */
int mapx_adamw_dense_shadow(float* p, const float* g, float* m, float* v, int64_t n, const float* sched,
                            int sched_len, const int32_t* done, double beta1, double beta2, double eps,
                            double weight_decay, mapx_bf16* shadow, hipStream_t stream);

/* ------------------------------------------------------------------ options of the hot-path classes
 * nn.Dropout of layers.py:95 (Embeddings) and :183 (MLPBlock): out = keep ? x / (1 - p) : 0 with the
 * keep mask drawn from Philox(seed, offset + *offset_dev_opt, element) — never stored: backward calls
 * the same kernel with the same (seed, offset) on the incoming gradient. */
int mapx_dropout(const float* x, int64_t n, float p, uint64_t seed, uint64_t offset, const int32_t* offset_dev_opt,
                 float* out, hipStream_t stream);
/* nn.LayerNorm(embed_size, eps) of layers.py:92-94,99-100 over rows [R, E] (E % 4 == 0, E <= 64):
 * y = (x - mean) rstd w + b; stats [R,2] = {mean, rstd} kept for backward.  Backward: dx, and
 * dy_xhat [R,E] = dy * xhat whose column sums are dL/dw (dL/db = column sums of dy). */
int mapx_layernorm_fwd(const float* x, int64_t R, int E, const float* w, const float* b, float eps, float* y,
                       float* stats, hipStream_t stream);
int mapx_layernorm_bwd(const float* dy, const float* x, const float* w, const float* stats, int64_t R, int E,
                       float* dx, float* dy_xhat, hipStream_t stream);
/* `hidden_act` of MLPBlock other than relu (layers.py:55-80 get_act, :182): kind 1 tanh, 2 sigmoid, 3 none, 4 elu,
 * 5 leu (layers.py:13-27, alpha = 1), 6 gelu (erf form, :36-37), 7 gelu_new (tanh form, :41-42), 8 swish (:46-47),
 * 9 mish (:51-52).  y[m, 0:N] (row stride ldy) = f(z[m, :]) for contiguous z [M, N];  dz [M, N] = dy * f'(z) with
 * the pre-activation z saved by the caller (dy row stride ld_dy).  relu stays fused in mapx_gemm_f32's epilogue. */
int mapx_act_fwd(int kind, const float* z, int64_t M, int N, float* y, int64_t ldy, hipStream_t stream);
int mapx_act_bwd(int kind, const float* dy, int64_t ld_dy, const float* z, int64_t M, int N, float* dz,
                 hipStream_t stream);

/* ------------------------------------------------------------------ vocabulary builders (f4)
 * One categorical field of data_preprocess/proc_avazu.py:237-262 / proc_criteo.py:147-163:
 *   for k, v in Counter(feat).most_common(): if v >= n_core: feat_map[name-k] = len(feat_map)
 *   feat_map[name-<oov>] = len(feat_map);   feat_ids = [feat_map.get(name-f, oov) for f in feat]
 * keys [n] i64 = the column's raw values (any 64-bit pattern but INT64_MIN).  Steps, each one launch:
 *  table_init  an open-addressing table of `capacity` (power of two, >= 2 x distinct values) slots:
 *              keys INT64_MIN, counts 0, first positions INT32_MAX;
 *  count       every row finds / claims its value's slot: count += 1, first = min(first, row);
 *              slot_of_row [n].  *err_flag |= 1: a key equals INT64_MIN; |= 2: table full;
 *  compact     occupied slots -> entries (order unspecified): slot_entry [capacity], entry_keys /
 *              entry_count / entry_first [>= distinct values]; n_entries_maxc [2] = {#entries, largest count};
 *  (the caller sorts the entries by first position, then stably by the keys of rank_keys:
 *   mapx_seg_plan twice -> by_first [U], by_count [U]: rank r holds entry by_first[by_count[r]])
 *  rank_keys   keys2[r] = max_count - entry_count[by_first[r]];
 *  assign      rank_of_entry [U]; ranked_keys / ranked_counts [U] = the values and counts in id order;
 *              *n_kept = number of values with count >= n_core (they are ranks 0 .. n_kept-1);
 *  map         ids_out[i * ld_out] = base + (rank < n_kept ? rank : n_kept)   (base + n_kept = <oov>). */
int mapx_vocab_table_init(int64_t* table_keys, int32_t* table_count, int32_t* table_first, int64_t capacity,
                          hipStream_t stream);
int mapx_vocab_count(const int64_t* keys, int64_t n, int64_t* table_keys, int32_t* table_count,
                     int32_t* table_first, int64_t capacity, int32_t* slot_of_row, int* err_flag,
                     hipStream_t stream);
int mapx_vocab_compact(const int64_t* table_keys, const int32_t* table_count, const int32_t* table_first,
                       int64_t capacity, int32_t* slot_entry, int64_t* entry_keys, int32_t* entry_count,
                       int32_t* entry_first, int32_t* n_entries_maxc, hipStream_t stream);
int mapx_vocab_rank_keys(const int32_t* entry_count, const int32_t* by_first, int64_t n_entries, int32_t max_count,
                         int32_t* keys2, hipStream_t stream);
int mapx_vocab_assign(const int32_t* by_first, const int32_t* by_count, const int32_t* entry_count,
                      const int64_t* entry_keys, int64_t n_entries, int32_t n_core, int32_t* rank_of_entry,
                      int64_t* ranked_keys, int32_t* ranked_counts, int32_t* n_kept, hipStream_t stream);
int mapx_vocab_map(const int32_t* slot_of_row, const int32_t* slot_entry, const int32_t* rank_of_entry,
                   const int32_t* n_kept, int64_t n, int64_t base, int64_t* ids_out, int64_t ld_out,
                   hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPX_HIP_H_ */
