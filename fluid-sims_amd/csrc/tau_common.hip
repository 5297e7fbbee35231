// tau_common.hip — error text, device probe, version.
#include "../../include/taueng.h"
#include "tau_common.h"

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
extern "C" int tau_device_count(int *n) {
  if (!n) return tau::fail("tau_device_count: null argument");
  *n = 0;
  TAU_HIP(hipGetDeviceCount(n));
  return 0;
}
