// tau_common.hip — error text, device probe, version.
#include "../../include/taueng.h"
#include "tau_common.h"
#include <vector>

namespace tau {
char *err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return 1;
}
static std::vector<int> guided_chunk_starts(int H, int nstrips, int slots_per_xcd, int lmin, int lmax) {
  const int hb = (H + 7) / 8;
  std::vector<int> pat;                                   // chunk lengths of one band, descending
  for (int R = hb; R > 0;) {
    int len = (int)((double)R * nstrips / (double)slots_per_xcd + 0.5);
    len = len < lmin ? lmin : (len > lmax ? lmax : len);
    if (R - len < (lmin + 1) / 2) len = R;
    pat.push_back(len); R -= len;
  }
  std::vector<int> start;                                 // every band gets the same number of chunks, clipped to the band
  for (int b = 0; b < 8; b++) {
    const int lo = (int)((long)H * b / 8), hi = (int)((long)H * (b + 1) / 8);
    int r = lo;
    for (size_t c = 0; c < pat.size(); c++) { start.push_back(r < hi ? r : hi); r += pat[c]; }
  }
  start.push_back(H);
  return start;
}
int guided_chunks(int H, int nstrips, int slots_per_xcd, int lmin, int lmax, int **table_dev, int *nchunks) {
  const std::vector<int> start = guided_chunk_starts(H, nstrips, slots_per_xcd, lmin, lmax);
  *table_dev = nullptr;
  TAU_HIP(hipMalloc(table_dev, start.size() * sizeof(int)));
  TAU_HIP(hipMemcpy(*table_dev, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
  *nchunks = (int)start.size() - 1;
  return 0;
}
} // namespace tau

extern "C" const char *tau_last_error(void) { return tau::err_buf(); }
extern "C" int tau_version(void) { return 100; }
extern "C" int tau_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
extern "C" int tau_guided_chunks(int H, int nstrips, int slots_per_xcd, int lmin, int lmax, int *starts, int cap, int *nchunks) {
  if (!nchunks || H < 1 || nstrips < 1 || slots_per_xcd < 1 || lmin < 1 || lmax < lmin) return tau::fail("tau_guided_chunks: bad argument");
  const std::vector<int> start = tau::guided_chunk_starts(H, nstrips, slots_per_xcd, lmin, lmax);
  *nchunks = (int)start.size() - 1;
  if (starts) {
    if (cap < (int)start.size()) return tau::fail("tau_guided_chunks: %d entries do not fit %d", (int)start.size(), cap);
    for (size_t i = 0; i < start.size(); i++) starts[i] = start[i];
  }
  return 0;
}
extern "C" int tau_device_count(int *n) {
  if (!n) return tau::fail("tau_device_count: null argument");
  *n = 0;
  TAU_HIP(hipGetDeviceCount(n));
  return 0;
}
