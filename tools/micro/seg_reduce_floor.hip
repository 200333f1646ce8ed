// The floor under seg_reduce_pass_a<8, true, NceContrib> (the NCE table's gradient rows, csrc/segreduce.h + nce.hip;
// VERDICT r3 item 4b): the kernel is a chain  sorted position j -> perm[j] -> (dlogit[p], h[p / K1, :]) -> accumulate ->
// store at the end of a run.  This program times that chain stripped of everything that is the kernel's own
// doing — no run detection, no conditional stores, no head / tail partials, no second pass — on the step's sizes
// (T = 24 576 targets, K + 1 = 26, P = 32: n = 638 976 entries, keys skewed like a unigram draw and sorted):
//   A  "stream":   perm[j] = j: the same loads in order (what the bytes cost without the permutation)
//   B  "gather":   perm = the sort's permutation: each lane group of 8 walks 32 consecutive sorted entries, 8 row loads
//                  in flight, accumulates dlogit * h in registers, stores ONE row per 32 entries
//   D  C + the shipped kernel's owner list: one returning atomicAdd on ONE counter per chunk that owns a spanning run
//   C  "gather, one store per run":  B + a store whenever the key changes (the real kernel's output traffic), still
//                  without partial rows and without the second pass
// The shipped kernel pair (pass A + pass B) measures 30.0 + 8.4 us serial on this workload (profiles/r04_bench_kernel_
// stats_serial.csv); whatever B / C take is not the kernel's to win.
// (C's stores race between lane groups that share a key: the values are garbage, the traffic is what is timed.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/seg_reduce_floor.hip -o /tmp/srf && /tmp/srf [skew [plan file [T [K + 1]]]]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LG = 8, CH = 32, BATCH = 8, P = 32;

template <int MODE>   // 0: one store per chunk; 1: a store at every change of key; 2: 1 + the shipped kernel's owner list
__global__ void __launch_bounds__(256) walk_kernel(const int* __restrict__ perm, const int* __restrict__ key,
                                                   const float* __restrict__ dlogit, const float* __restrict__ h,
                                                   int K1, long n, float* __restrict__ out, float* __restrict__ outx,
                                                   int* __restrict__ owners, int* __restrict__ n_owners) {
  const int lane = threadIdx.x & 63, lig = lane % LG, gbase = lane - lig;
  const long group = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LG;
  const long j0 = group * CH;
  if (j0 >= n) return;
  int myperm[CH / LG], mykey[CH / LG];
#pragma unroll
  for (int i = 0; i < CH / LG; ++i) {
    const long j = j0 + i * LG + lig;
    myperm[i] = j < n ? perm[j] : 0;
    mykey[i] = j < n ? key[j] : -1;
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float accx = 0.f;
  int cur = __shfl(mykey[0], gbase, 64);
#pragma unroll
  for (int e0 = 0; e0 < CH; e0 += BATCH) {
    float4 v[BATCH];
    float d[BATCH];
    int rk[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = e0 + u;
      const int p = __shfl(myperm[e / LG], gbase + (e % LG), 64);
      rk[u] = __shfl(mykey[e / LG], gbase + (e % LG), 64);
      d[u] = dlogit[p];
      v[u] = *reinterpret_cast<const float4*>(h + (long)(p / K1) * P + 4 * lig);
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      if (MODE >= 1 && rk[u] != cur && rk[u] >= 0) {
        *reinterpret_cast<float4*>(out + (long)cur * P + 4 * lig) = acc;
        if (lig == 0) outx[cur] = accx;
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        accx = 0.f;
        cur = rk[u];
      }
      acc.x += d[u] * v[u].x; acc.y += d[u] * v[u].y; acc.z += d[u] * v[u].z; acc.w += d[u] * v[u].w;
      accx += d[u];
    }
  }
  float* dst = MODE >= 1 ? out + (long)cur * P : out + group * P;
  *reinterpret_cast<float4*>(dst + 4 * lig) = acc;
  if (lig == 0) outx[MODE >= 1 ? cur : group] = accx;
  if (MODE == 2) {      // a chunk whose last run starts inside it and continues into the next chunk lists itself
    const long j1 = j0 + CH < n ? j0 + CH : n;
    const bool started_inside = j0 == 0 || key[j0 - 1] != cur, ends = j1 == n || key[j1] != cur;
    if (started_inside && !ends && lig == 0) owners[atomicAdd(n_owners, 1)] = (int)group;
  }
}

int main(int argc, char** argv) {
  const double skew = argc > 1 ? atof(argv[1]) : 24.0;      // 24: ~86 k distinct keys, what one step's samples name
  const int T = argc > 3 ? atoi(argv[3]) : 24576, K1 = argc > 4 ? atoi(argv[4]) : 26;
  const long n = (long)T * K1, V = 9449445;
  std::vector<int> ids(n), perm(n), rank(n);
  const char* plan_file = argc > 2 ? argv[2] : nullptr;      // perm[n] then rank[n], int32 (tools/micro/seg_reduce_probe.py)
  srand(7);
  for (long i = 0; i < n; ++i) {                         // skewed like a unigram draw: half of the mass on a few thousand ids
    const double u = (double)rand() / RAND_MAX;
    ids[i] = (int)std::min<double>(V - 1, V * std::pow(u, skew));
  }
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return ids[a] < ids[b]; });
  long uniq = 0;
  for (long j = 0; j < n; ++j) {
    if (j == 0 || ids[perm[j]] != ids[perm[j - 1]]) ++uniq;
    rank[j] = (int)(uniq - 1);
  }
  if (plan_file) {                                           // one real step's plan instead of the synthetic one
    FILE* f = fopen(plan_file, "rb");
    if (!f || fread(perm.data(), 4, n, f) != (size_t)n || fread(rank.data(), 4, n, f) != (size_t)n) { printf("cannot read %s\n", plan_file); return 1; }
    fclose(f);
    uniq = 0;
    for (long j = 0; j < n; ++j) {
      if (rank[j] > 0) rank[j] -= 1;                         // (the plan's ranks are 1-based)
      uniq = std::max<long>(uniq, rank[j] + 1);
    }
  }
  std::vector<int> ident(n);
  std::iota(ident.begin(), ident.end(), 0);
  int *d_perm, *d_ident, *d_rank;
  float *d_dl, *d_h, *d_out, *d_outx;
  int *d_owners, *d_nown;
  CK(hipMalloc(&d_owners, n * 4)); CK(hipMalloc(&d_nown, 4));
  CK(hipMalloc(&d_perm, n * 4)); CK(hipMalloc(&d_ident, n * 4)); CK(hipMalloc(&d_rank, n * 4));
  CK(hipMalloc(&d_dl, n * 4)); CK(hipMalloc(&d_h, (size_t)T * P * 4));
  CK(hipMalloc(&d_out, (size_t)n * P * 4 / 8 + uniq * P * 4)); CK(hipMalloc(&d_outx, n * 4));
  CK(hipMemcpy(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ident, ident.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rank, rank.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_dl, 0, n * 4)); CK(hipMemset(d_h, 0, (size_t)T * P * 4));
  const long groups = (n + CH - 1) / CH;
  const int grid = (int)((groups * LG + 255) / 256);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* what, auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / 50, bytes = (double)n * (P * 4 + 4 + 8);
    printf("%-44s %7.1f us   %6.0f GB/s of (row + dlogit + perm + key) per entry\n", what, us, bytes / us / 1e3);
  };
  printf("n = %ld entries, %ld distinct keys, %d workgroups\n", n, uniq, grid);
  run("A  in order, one store per 32 entries", [&] { hipLaunchKernelGGL(walk_kernel<0>, dim3(grid), dim3(256), 0, 0, d_ident, d_rank, d_dl, d_h, K1, n, d_out, d_outx, d_owners, d_nown); });
  run("B  permuted, one store per 32 entries", [&] { hipLaunchKernelGGL(walk_kernel<0>, dim3(grid), dim3(256), 0, 0, d_perm, d_rank, d_dl, d_h, K1, n, d_out, d_outx, d_owners, d_nown); });
  run("C  permuted, one store per run of a key", [&] { hipLaunchKernelGGL(walk_kernel<1>, dim3(grid), dim3(256), 0, 0, d_perm, d_rank, d_dl, d_h, K1, n, d_out, d_outx, d_owners, d_nown); });
  run("D  C + the owner list (one returning atomic per spanning run)", [&] { CK(hipMemsetAsync(d_nown, 0, 4)); hipLaunchKernelGGL(walk_kernel<2>, dim3(grid), dim3(256), 0, 0, d_perm, d_rank, d_dl, d_h, K1, n, d_out, d_outx, d_owners, d_nown); });
  int nown = 0;
  CK(hipMemcpy(&nown, d_nown, 4, hipMemcpyDeviceToHost));
  printf("   (%d chunks own a spanning run)\n", nown);
  return 0;
}
