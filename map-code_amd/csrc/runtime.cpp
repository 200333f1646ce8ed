// Library-wide state of libmapx_hip.so: the thread-local error string behind
// mapx_last_error(), the ABI version, and the host-side Walker alias table builder.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "../../include/mapx_hip.h"
#include "common.h"

namespace mapx {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mapx

namespace mapx {
// amax.h: the device word whose value tags the magnitude records written from now on (one process = one GPU)
static int32_t* g_amax_epoch = nullptr;
const int32_t* amax_epoch_ptr() { return g_amax_epoch; }
}  // namespace mapx

extern "C" int mapx_amax_epoch_source(int32_t* device_word_opt) {
  mapx::g_amax_epoch = device_word_opt;
  return MAPX_OK;
}

extern "C" const char* mapx_last_error(void) { return mapx::g_err; }

extern "C" int mapx_abi_version(void) { return MAPX_ABI_VERSION; }

extern "C" int mapx_host_alloc_coherent(size_t bytes, void** out) {
  MAPX_REQUIRE(out && bytes > 0, "host_alloc_coherent: bad arguments");
  MAPX_HIP(hipHostMalloc(out, bytes, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable));
  memset(*out, 0, bytes);
  return MAPX_OK;
}

extern "C" int mapx_host_free(void* p) {
  if (p) MAPX_HIP(hipHostFree(p));
  return MAPX_OK;
}

// Walker alias table for the NCE noise distribution.  Replaces the pure-Python O(V) loop
// of reference code/nce/alias_multinomial.py:39-72 (minutes at V = 9.4 M) with the same
// visiting order and the same float32 arithmetic, so the table is bit-identical to the
// one the reference caches in data_dir/alias_self_{prob,alias}.h5.  Host memory in and out.
extern "C" int mapx_alias_build_host(const float* probs, int64_t n, float* out_prob,
                                     int64_t* out_alias) {
  MAPX_REQUIRE(probs && out_prob && out_alias && n > 0, "alias_build: null pointer or n <= 0");
  std::vector<int64_t> small, large;
  small.reserve(n);
  large.reserve(n);
  const float kf = (float)n;
  for (int64_t i = 0; i < n; ++i) {
    out_alias[i] = 0;
    float q = kf * probs[i];
    out_prob[i] = q;
    (q < 1.0f ? small : large).push_back(i);
  }
  while (!small.empty() && !large.empty()) {
    int64_t s = small.back(), l = large.back();
    small.pop_back();
    large.pop_back();
    out_alias[s] = l;
    float q = (out_prob[l] - 1.0f) + out_prob[s];
    out_prob[l] = q;
    (q < 1.0f ? small : large).push_back(l);
  }
  for (int64_t i : small) out_prob[i] = 1.0f;
  for (int64_t i : large) out_prob[i] = 1.0f;
  return MAPX_OK;
}
