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
    if (e_ != hipSuccess) {                                                               \
      (void)hipGetLastError(); /* the runtime's last-error is sticky: do not leave it for the next launch check */ \
      return ::tau::fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                     \
  } while (0)

#define TAU_LAUNCH_CHECK(name)                                                            \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess) return ::tau::fail("%s launch: %s", name, hipGetErrorString(e_)); \
  } while (0)

// Owns a half-built handle inside a *_create function: TAU_HIP returns early on the first failing HIP call,
// and this hands what was already allocated to the handle's own destroy function.
template <class H>
struct HandleGuard {
  H *h;
  void (*destroy)(H *);
  ~HandleGuard() { if (h) destroy(h); }
  H *release() { H *t = h; h = nullptr; return t; }
};

// Linear block id -> work item, so that the 8 XCDs (block b runs on XCD b % 8) each walk a
// contiguous range of tiles: neighbouring tiles share halo lines through the same L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned nblocks) {
  const unsigned per = nblocks >> 3;         // full groups of 8
  const unsigned body = per << 3;
  if (b >= body) return b;                   // ragged tail stays where it is
  return (b & 7u) * per + (b >> 3);
}

// Monotone max of positive floats kept as their bit pattern.  Every workgroup ends with one of these; a
// same-address atomic costs ~12 ns at the L2 (MI355X_MICROARCH.md, row "fanin"), which at 10^5 workgroups
// per launch is milliseconds — so first look (relaxed, agent scope: served by L2, never a stale L1 line)
// and only issue the atomic when this value would raise the maximum.  Skipping on an observed value that
// is already >= ours is exact because the word only ever grows within a launch.
__device__ __forceinline__ void atomic_max_float_bits(unsigned *word, float v) {
  if (!(v > 0.f)) return;
  const unsigned bits = __float_as_uint(v);
  if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= bits) return;
  atomicMax(word, bits);
}

// Guided chunk schedule of a row march (host side; tau_common.hip).  A march's work list is chunk-major (work item = chunk *
// nstrips + strip) and, through xcd_swizzle, each XCD walks a contiguous eighth of it in dispatch order: a band of H / 8 rows on
// `slots_per_xcd` resident waves.  Chunks of one length leave the chip draining for a chunk's duration at the end of the
// launch; here a chunk is as long as the band's remaining rows x strips shared out over the slots (clamped to [lmin, lmax]), so
// the first waves are long — few warm-up rows — and the last ones short.  Returns a device table of nchunks + 1 row starts
// (chunk c = rows [t[c], t[c + 1]), possibly empty where a band is a row shorter than the pattern); the caller frees it.
int guided_chunks(int H, int nstrips, int slots_per_xcd, int lmin, int lmax, int **table_dev, int *nchunks);

// one Burgers viscosity pass through the row-marching kernel (stencil2d.hip) with the time step read
// from a device word: nu * (*dt_dev) * frac  (used by flow2d.hip for visc_substeps > 1)
int st2_burgers_pass(const float *a, const float *b, float *oa, float *ob, int nx, int ny, float dx, float dy, float nu,
                     float u0, int oneD, const float *dt_dev, float frac, hipStream_t stream);

// Addressing of the marching / plane kernels: every global access is  <scalar base pointer> + <one 32-bit byte offset per
// lane>, the form global_load / global_store take directly (saddr + voffset).  64-bit per-lane addresses cost a half-rate
// v_lshl_add_u64 per access and a VGPR pair each.  hipcc would rather add the lane offset to a group's base once and then
// the field stride per access in 64-bit VALU ops: the empty asm pins each base in an SGPR pair.  Instruction selection
// works one basic block at a time and folds the zero-extension of the lane offset into the access only if it sees it
// there, so each block takes its own copy (lane_off) of the offset.
typedef __attribute__((address_space(1))) char GChar;     // global address space spelled out: the asm below would otherwise
typedef __attribute__((address_space(1))) float GFloat;   // hide the pointer's provenance and turn the access into flat_load / flat_store
__device__ __forceinline__ float gld(const GChar *sbase, unsigned voff) {
  asm volatile("" : "+s"(sbase));
  return *(const GFloat *)(sbase + voff);
}
__device__ __forceinline__ void gst(GChar *sbase, unsigned voff, float v) {
  asm volatile("" : "+s"(sbase));
  *(GFloat *)(sbase + voff) = v;
}
__device__ __forceinline__ unsigned max3u(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }   // v_max3_u32
__device__ __forceinline__ unsigned lane_off(unsigned v) { asm volatile("" : "+v"(v)); return v; }

} // namespace tau
