// The arithmetic of the lazy row-sparse AdamW (optim.hip's header comment): one update, one zero-gradient update, and the
// CLOSED-FORM replay of a gap of zero-gradient updates — shared by the optimizer's kernels (optim.hip) and by the
// forward kernels that read stale rows WITHOUT writing them back (nce.hip: LazyRows below).
#pragma once
#include "common.h"

namespace mapx {

struct AdamHyper {
  float beta1, beta2, eps;
  float one_m_b1, one_m_b2;
};

__device__ inline void adam_elem(float& p, float& m, float& v, float g, float step, float decay,
                                 const AdamHyper& h) {
  m = m * h.beta1 + g * h.one_m_b1;
  v = v * h.beta2 + (h.one_m_b2 * g) * g;
  const float denom = sqrtf(v) + h.eps;
  p = p + ((-step) * m) / denom;   // ATen addcdiv: self + (value * t1) / t2
  if (decay != 0.f) p = p + (-decay) * p;
}

// Zero-gradient update (the replayed steps of the lazy tables).  The Adam term uses the
// hardware sqrt / reciprocal (1 ulp each) instead of the correctly-rounded sequences: the
// term is <= lr in magnitude, so the deviation from the reference's arithmetic is <= ~1e-7*lr
// per replayed step, far inside the fp32 parity budget, at a third of the instruction count.
// Returns true when the Adam term was too small to change p (|term| < 2^-26 |p|: the fp32
// add is then a no-op, and stays one for all later zero-gradient steps because m shrinks by
// beta1 per step while sqrt(v) shrinks only by sqrt(beta2)).
__device__ inline bool adam_elem_zero_grad(float& p, float& m, float& v, float step, float decay,
                                           const AdamHyper& h) {
  m = m * h.beta1;
  v = v * h.beta2;
  const float denom = __builtin_amdgcn_sqrtf(v) + h.eps;
  const float term = ((-step) * m) * __builtin_amdgcn_rcpf(denom);
  const bool dead = fabsf(term) < 1.4901161e-8f * fabsf(p);
  p = p + term;
  if (decay != 0.f) p = p + (-decay) * p;
  return dead;
}

// Prefix tables for the closed-form tail of a replay (host fp64):
//   aux[0*len + s] = prod_{i<s} (1 - lr_i * wd)   (wd = the optimizer's weight decay)
//   aux[1*len + n] = beta1^n,  aux[2*len + n] = beta2^n
//   aux[(3 + i) * len + s]          = R_i[s] with the optimizer's weight decay, i = 0..kJ
//   aux[(3 + kJ + 1 + i) * len + s] = R_i[s] without decay                      (see replay_coef)
struct ReplayAux {
  const double* t;
  int len;
  int rows;          // 3: prefix tables only (iterative replay + closed-form tail); 3 + 2(kJ+1): full closed form
  double rho;        // beta1 / sqrt(beta2)
  double inv_beta;   // 1 / sqrt(beta2)
};

// ---------------------------------------------------------------------------------------------
// Closed form of n zero-gradient AdamW steps (updates s+1 .. e, n = e - s) on one element:
//     m_k = b1^k m,  v_k = b2^k v,   p_k = (p_{k-1} - a_u m_k / (sqrt(v_k) + eps)) (1 - d_u),  u = s+k-1
// with a_u the step size and d_u = lr_u * wd of update u+1.  With r = sqrt(v), beta = sqrt(b2),
// y = eps / (r + eps) and delta_k = beta^-k - 1 (<= 0.07 while rho^k matters):
//     1 / (r beta^k + eps) = beta^-k / ((r + eps) (1 + y delta_k)) = beta^-k/(r+eps) * sum_j (-y delta_k)^j
// so the total Adam displacement is   m / (r + eps) * sum_j (-y)^j T_j,   with per-ROW coefficients
//     T_j = sum_k a_u rho^k delta_k^j D_k = sum_i C(j,i) (-1)^(j-i) Q_i,      rho = b1 / beta,
//     Q_i = sum_k a_u (rho beta^-i)^k D_k = q_i [ (P_e / P_s) R_i[s] - q_i^n R_i[e] ],   q_i = rho beta^-i,
// D_k = prod_{j >= u}^{e-1} (1 - d_j), P = prefix product of (1 - d), and the host fp64 table
//     R_i[s] = a_s + q_i / (1 - d_s) * R_i[s+1]        (backward recurrence: every quantity is O(a), no
// underflow and no cancellation however long the gap or the schedule).  kJ + 1 = 7 terms leave a
// relative error < 0.07^7 = 8e-9 of the displacement for ANY eps/r; against step-by-step replay
// in fp64 the closed form agrees to 2e-14 (tests: lazy == dense reference AdamW), which is 8
// orders closer than the reference's own fp32 stepwise rounding.  Cost: O(1) per element instead
// of up to ~150 replayed steps — the catch-up kernels become HBM-bound.
constexpr int kJ = 6;
struct ReplayCoef {
  float fp, fm, fv;
  float T[kJ + 1];
};

__device__ inline void replay_coef(int s, int e, const ReplayAux& ax, bool decayed, ReplayCoef& c) {
  const int n = e - s;
  const double* __restrict__ P = ax.t;
  const double* __restrict__ R = ax.t + (size_t)(3 + (decayed ? 0 : kJ + 1)) * ax.len;
  const double pr = decayed ? P[e] / P[s] : 1.0;
  const double b1n = ax.t[ax.len + n], b2n = ax.t[2 * ax.len + n];
  const double binv_n = 1.0 / sqrt(b2n);                 // beta^-n
  double Q[kJ + 1];
  double qi = ax.rho, qn = b1n * binv_n;                 // q_i, q_i^n
#pragma unroll
  for (int i = 0; i <= kJ; ++i) {
    Q[i] = qi * (pr * R[(size_t)i * ax.len + s] - qn * R[(size_t)i * ax.len + e]);
    qi *= ax.inv_beta;
    qn *= binv_n;
  }
  // T_j = j-th forward difference of Q at 0 (in place)
#pragma unroll
  for (int j = 1; j <= kJ; ++j)
#pragma unroll
    for (int i = kJ; i >= j; --i) Q[i] -= Q[i - 1];
#pragma unroll
  for (int j = 0; j <= kJ; ++j) c.T[j] = (float)Q[j];
  c.fp = (float)pr;
  c.fm = (float)b1n;
  c.fv = (float)b2n;
}

// (Every multiply-add is spelled out: this function is compiled into the catch-up, the update AND the forward kernels
// that read rows through their pending updates, and all of them must produce the same bits — left to the compiler,
// `p * fp - q * poly` was contracted one way in one kernel and the other way in another.)
__device__ inline void replay_elem_closed(float& p, float& m, float& v, const ReplayCoef& c, float eps) {
  const float den = sqrtf(v) + eps;
  const float y = eps / den;
  float poly = c.T[kJ];
#pragma unroll
  for (int j = kJ - 1; j >= 0; --j) poly = __builtin_fmaf(-y, poly, c.T[j]);
  const float disp = __fmul_rn(m / den, poly);
  p = __builtin_fmaf(p, c.fp, -disp);
  m = __fmul_rn(m, c.fm);
  v = __fmul_rn(v, c.fv);
}

__device__ inline void closed_form_tail(int s, int to, const ReplayAux& ax, bool decayed,
                                        float& fp, float& fm, float& fv) {
  const int n = to - s;
  const int a = s < ax.len ? s : ax.len - 1, b = to < ax.len ? to : ax.len - 1;
  const int nn = n < ax.len ? n : ax.len - 1;
  fp = decayed ? (float)(ax.t[b] / ax.t[a]) : 1.f;
  fm = (float)ax.t[ax.len + nn];
  fv = (float)ax.t[2 * ax.len + nn];
}


// Replay zero-gradient updates (from+1 .. to) on one float4 of a row.
__device__ inline void replay4(float4& p, float4& m, float4& v, int from, int to,
                               const float2* __restrict__ sched, int sched_len, float wd,
                               const AdamHyper& h, const ReplayAux& ax) {
  if (ax.rows > 3 && to < ax.len) {       // O(1): the whole gap in closed form
    if (to <= from) return;
    ReplayCoef c;
    replay_coef(from, to, ax, wd != 0.f, c);
    replay_elem_closed(p.x, m.x, v.x, c, h.eps);
    replay_elem_closed(p.y, m.y, v.y, c, h.eps);
    replay_elem_closed(p.z, m.z, v.z, c, h.eps);
    replay_elem_closed(p.w, m.w, v.w, c, h.eps);
    return;
  }
  int s = from;
  for (; s < to; ++s) {  // update s+1 uses sched[s]
    const float2 sc = sched[s < sched_len ? s : sched_len - 1];
    const float decay = sc.y * wd;
    bool dead = adam_elem_zero_grad(p.x, m.x, v.x, sc.x, decay, h);
    dead &= adam_elem_zero_grad(p.y, m.y, v.y, sc.x, decay, h);
    dead &= adam_elem_zero_grad(p.z, m.z, v.z, sc.x, decay, h);
    dead &= adam_elem_zero_grad(p.w, m.w, v.w, sc.x, decay, h);
    if (dead) { ++s; break; }
  }
  if (s < to) {
    float fp, fm, fv;
    closed_form_tail(s, to, ax, wd != 0.f, fp, fm, fv);
    p.x *= fp; p.y *= fp; p.z *= fp; p.w *= fp;
    m.x *= fm; m.y *= fm; m.z *= fm; m.w *= fm;
    v.x *= fv; v.y *= fv; v.z *= fv; v.w *= fv;
  }
}


// betas arrive as doubles: the reference computes (1 - beta) in Python double precision and
// only then rounds to fp32 (1 - 0.999 != 1 - float(0.999) at the 1e-5 level).
inline AdamHyper make_hyper(double b1, double b2, double eps) {
  AdamHyper h;
  h.beta1 = (float)b1; h.beta2 = (float)b2; h.eps = (float)eps;
  h.one_m_b1 = (float)(1.0 - b1);
  h.one_m_b2 = (float)(1.0 - b2);
  return h;
}

// ---------------------------------------------------------------------------------------------------------------
// Rows read through their pending replay.  A forward kernel that reads table rows named by the batch (the NCE head's
// sampled rows) used to need a catch-up pass first: every stale row read-modify-written (p, m | v) so that the forward
// could read p, and read-modify-written AGAIN by the gradient update at the end of the step.  With a LazyRows the
// forward kernel reads last[row] beside the row and, when the row is stale, its m | v record, and replays the gap in
// registers — the same closed form, hence the same bits, as the catch-up pass would have stored — and nothing is written:
// the gradient update (table_adam_row<LATE_FROM>), which already replays whatever gap it finds before it applies the
// gradient, becomes the row's one read-modify-write of the step.
// `coef` (mapx_replay_coef_table): the per-row coefficients of the closed form depend on (from, *done) only, so they
// are tabulated once per step — coef[(decayed ? 0 : len) + from] = 12 floats {fp, fm, fv, T0..T6, -, -} — instead of
// being re-derived from the fp64 schedule tables (16 loads + ~100 fp64 operations) at every stale access.
constexpr int kCoefFloats = 12;
struct LazyRows {
  const float* m0; const float* v0; int64_t ld_mv0; float wd0;      // moments of the main rows [V, W0]
  const float* m1; const float* v1; int64_t ld_mv1; float wd1;      // ... of the scalar table, or null
  const int32_t* last;
  const float2* sched; int sched_len;
  const int32_t* done;
  AdamHyper h;
  ReplayAux ax;
  const float* coef;               // [2][ax.len][kCoefFloats] for *done (mapx_replay_coef_table)
};

__device__ inline void lazy_coef(const LazyRows& lz, int from, int to, bool decayed, ReplayCoef& c) {
  const float4* q = reinterpret_cast<const float4*>(lz.coef + ((size_t)(decayed ? 0 : lz.ax.len) + from) * kCoefFloats);
  const float4 a = q[0], b = q[1], d = q[2];
  c.fp = a.x; c.fm = a.y; c.fv = a.z;
  c.T[0] = a.w; c.T[1] = b.x; c.T[2] = b.y; c.T[3] = b.z; c.T[4] = b.w; c.T[5] = d.x; c.T[6] = d.y;
}
// (gaps that end past the schedule tables — steps beyond the planned training — are replayed step by step: kept out of
// line so that the forward kernels do not carry that loop's registers)
__device__ inline void lazy_replay4_steps(const LazyRows& lz, float4& p, float4 m, float4 v,
                                                                    int from, int to) {
  replay4(p, m, v, from, to, lz.sched, lz.sched_len, lz.wd0, lz.h, lz.ax);
}
__device__ inline void lazy_replay1_steps(const LazyRows& lz, float& p, float m, float v,
                                                                    int from, int to) {
  int s = from;
  for (; s < to; ++s) {
    const float2 sc = lz.sched[s < lz.sched_len ? s : lz.sched_len - 1];
    if (adam_elem_zero_grad(p, m, v, sc.x, sc.y * lz.wd1, lz.h)) { ++s; break; }
  }
  if (s < to) {
    float fp, fm, fv;
    closed_form_tail(s, to, lz.ax, lz.wd1 != 0.f, fp, fm, fv);
    p *= fp;
  }
}
// replay4 / the scalar table's replay of table_adam_row with the coefficients from the table (same bits)
__device__ inline void lazy_replay4(const LazyRows& lz, float4& p, float4 m, float4 v, int from, int to) {
  if (lz.ax.rows > 3 && to < lz.ax.len) {
    if (to <= from) return;
    ReplayCoef c;
    lazy_coef(lz, from, to, lz.wd0 != 0.f, c);
    replay_elem_closed(p.x, m.x, v.x, c, lz.h.eps);
    replay_elem_closed(p.y, m.y, v.y, c, lz.h.eps);
    replay_elem_closed(p.z, m.z, v.z, c, lz.h.eps);
    replay_elem_closed(p.w, m.w, v.w, c, lz.h.eps);
    return;
  }
  lazy_replay4_steps(lz, p, m, v, from, to);
}
__device__ inline void lazy_replay1(const LazyRows& lz, float& p, float m, float v, int from, int to) {
  if (lz.ax.rows > 3 && to < lz.ax.len) {
    if (to > from) {
      ReplayCoef c;
      lazy_coef(lz, from, to, lz.wd1 != 0.f, c);
      replay_elem_closed(p, m, v, c, lz.h.eps);
    }
    return;
  }
  lazy_replay1_steps(lz, p, m, v, from, to);
}

}  // namespace mapx

struct mapx_lazy_rows;
namespace mapx {
bool lazy_rows_from(const mapx_lazy_rows* q, int W, const char* what, LazyRows* lz);      // gather.hip
}  // namespace mapx

