// tau_common.h — error plumbing shared by the engine translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace tau {

// thread-local last-error text, read through tau_last_error()
char *err_buf();
int fail(const char *fmt, ...);

#define TAU_HIP(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return ::tau::fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define TAU_LAUNCH_CHECK(name)                                                            \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess) return ::tau::fail("%s launch: %s", name, hipGetErrorString(e_)); \
  } while (0)

// Linear block id -> work item, so that the 8 XCDs (block b runs on XCD b % 8) each walk a
// contiguous range of tiles: neighbouring tiles share halo lines through the same L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned nblocks) {
  const unsigned per = nblocks >> 3;         // full groups of 8
  const unsigned body = per << 3;
  if (b >= body) return b;                   // ragged tail stays where it is
  return (b & 7u) * per + (b >> 3);
}

} // namespace tau
