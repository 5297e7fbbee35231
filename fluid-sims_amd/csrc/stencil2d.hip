// stencil2d.hip — periodic 5-point-Laplacian steps for gfx950 (MI355X):
//   Gray-Scott reaction-diffusion        (tau_gray_scott.cu:141-171)
//   Burgers viscosity pass               (tau_burgers.cu:490-525)   — race-free ping-pong form
//   shallow-water viscosity pass         (tau_shallow_water.cu:516-547) — race-free ping-pong form
//
// All three are two fp32 fields in, two out, 16 B/cell of compulsory HBM traffic — HBM bound.
// CDNA4 layout: a wave64 owns a 256-column strip (one float4 per lane = 1 KiB per load
// instruction) and MARCHES down a chunk of rows keeping the three live rows in VGPRs, so every
// row is fetched once per strip and chunk (chunk-edge rows a second time, from L2); left/right neighbours come from the adjacent lane
// (wave shuffle), only the two edge lanes of a strip issue an extra scalar load.  No LDS, no
// integer modulo per cell (the reference wraps every index with %; here only strip edges and
// chunk edges wrap, on scalar registers).
//
// This file is compiled with -ffp-contract=off and IEEE division so that Gray-Scott and the
// shallow-water pass are BIT-EXACT against the oracle's arithmetic order.

#include "../../include/taueng.h"
#include "tau_common.h"
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

namespace st2 {

constexpr int ROWS = 8;         // rows marched by one wave, Burgers (VALU-bound: every fetched row is decoded through sinh, so short
                                // chunks would decode (r + 2) / r times as much)
constexpr int ROWS_HBM = 2;     // ... Gray-Scott and the shallow-water pass (HBM-bound).  Round 4, 8192^2, one step per launch, streaming
                                // stores: 1 row 0.211 ms, 2: 0.187, 3: 0.190, 4: 0.195, 8: 0.203, 16: 0.213, 32: 0.205, 64: 0.212 — 131 k short
                                // waves keep more loads in flight than 32 k longer ones, and the two rows a chunk shares with its
                                // neighbours come from the L2 of the same XCD (xcd_swizzle keeps neighbouring chunks together):
                                // 5.74 TB/s algorithmic = 71.7 % of the 8 TB/s roofline (torch's own copy_ on the same box: 5.2)
constexpr int WAVES = 4;        // waves per workgroup
enum { K_GS = 0, K_BURGERS = 1, K_SW = 2 };

struct Args {
  const float *a, *b;
  float *oa, *ob;
  int nx, ny;
  int nstrips, nchunks, rows, nt;
  // Gray-Scott
  float dx2, dt, Du, Dv, feed, kill;
  float inv_dx2;         // exact only when dx2 is a power of two
  int dx2_pow2;
  // Laplacian passes
  float invdx2, invdy2, nudt, u0, inv_u0;
  const float *dt_dev;   // when set: nudt = nudt * (*dt_dev)  (device-resident time step)
  int burgers_fast;      // fused Burgers passes keep u decoded between their levels (TAU_ST2_BURGERS_FAST=1; default 0)
};

// The per-cell coefficients as VGPR values: a VALU instruction with an SGPR operand issues at half rate on gfx950
// (profiles/r02/valu_calib.txt); the viscosity cells have four to six such operands among their ~20 instructions.  Same arithmetic.
#if !defined(TAU_EXPERIMENT) && (defined(TAU_ST2_VREG))
#error "tuning overrides need -DTAU_EXPERIMENT (scripts/variant_build_file.sh sets it)"
#endif
#ifndef TAU_ST2_VREG
#define TAU_ST2_VREG 1
#endif
__device__ __forceinline__ float vreg(float s) {
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
template <int KIND>
__device__ __forceinline__ Args coeffs_in_vgprs(const Args &A0) {
  Args A = A0;
  if (KIND == K_GS) return A;   // Gray-Scott is bandwidth bound: no gain, and the single-step kernel would drop from 6 to 5 waves per SIMD
#if TAU_ST2_VREG
  A.dt = vreg(A0.dt); A.Du = vreg(A0.Du); A.Dv = vreg(A0.Dv); A.feed = vreg(A0.feed); A.kill = vreg(A0.kill);
  A.inv_dx2 = vreg(A0.inv_dx2); A.dx2 = vreg(A0.dx2);
  A.invdx2 = vreg(A0.invdx2); A.invdy2 = vreg(A0.invdy2); A.u0 = vreg(A0.u0); A.inv_u0 = vreg(A0.inv_u0);
  A.nudt = vreg(A0.dt_dev ? A0.nudt * (*A0.dt_dev) : A0.nudt);
  A.dt_dev = nullptr;
#endif
  return A;
}

// -------- transcendental pair for the Burgers encoding, accurate to ~1e-7 relative
__device__ __forceinline__ float fsinh(float x) {
  float ax = fabsf(x);
  float x2 = x * x;
  float series = x * (1.f + x2 * (1.f / 6.f) * (1.f + x2 * (1.f / 20.f) * (1.f + x2 * (1.f / 42.f))));
  float e = __builtin_amdgcn_exp2f(ax * 1.44269504088896341f);
  float big = copysignf(0.5f * (e - __builtin_amdgcn_rcpf(e)), x);
  return (ax < 0.5f) ? series : big;
}
__device__ __forceinline__ float fasinh(float x) {
  float ax = fabsf(x);
  float x2 = x * x;
  // x - x^3/6 + 3x^5/40 - 15x^7/336 + 105x^9/3456
  float series = ax * (1.f + x2 * (-1.f / 6.f + x2 * (3.f / 40.f + x2 * (-15.f / 336.f + x2 * (105.f / 3456.f)))));
  float big = __builtin_amdgcn_logf(ax + __builtin_amdgcn_sqrtf(x2 + 1.0f)) * 0.69314718055994531f;
  return copysignf((ax < 0.125f) ? series : big, x);
}

struct Row {
  float4 a, b;     // the lane's four cells of both fields
  float al, ar;    // strip-edge neighbours (valid on the edge lanes only)
  float bl, br;
};

template <int KIND>
__device__ __forceinline__ void load_row(const Args &A, int j, int x4, int xl, int xr, bool act, bool first, bool last,
                                         Row &r) {
  if (act) {
    const size_t base = (size_t)j * A.nx;
    if (A.nt & 2) { // streaming loads (TAU_ST2_NT bit 1; off: measured 4.4-4.5 TB/s against 5.0-5.3 — the two rows a
                    // chunk shares with its neighbours are fetched twice, and a streamed line is gone by the second time)
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f va = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(A.a + base + x4));
      const v4f vb = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(A.b + base + x4));
      r.a = make_float4(va.x, va.y, va.z, va.w);
      r.b = make_float4(vb.x, vb.y, vb.z, vb.w);
    } else {
      r.a = *reinterpret_cast<const float4 *>(A.a + base + x4);
      r.b = *reinterpret_cast<const float4 *>(A.b + base + x4);
    }
    r.al = first ? A.a[base + xl] : 0.f;
    r.bl = first ? A.b[base + xl] : 0.f;
    r.ar = last ? A.a[base + xr] : 0.f;
    r.br = last ? A.b[base + xr] : 0.f;
    if (KIND == K_BURGERS) { // decode once per fetched value: u = u0*sinh(phi), :503-506
      r.a.x = A.u0 * fsinh(r.a.x); r.a.y = A.u0 * fsinh(r.a.y); r.a.z = A.u0 * fsinh(r.a.z); r.a.w = A.u0 * fsinh(r.a.w);
      r.b.x = A.u0 * fsinh(r.b.x); r.b.y = A.u0 * fsinh(r.b.y); r.b.z = A.u0 * fsinh(r.b.z); r.b.w = A.u0 * fsinh(r.b.w);
      r.al = A.u0 * fsinh(r.al); r.ar = A.u0 * fsinh(r.ar); r.bl = A.u0 * fsinh(r.bl); r.br = A.u0 * fsinh(r.br);
    }
  }
}

template <int KIND>
__device__ __forceinline__ void cell(const Args &A, float uc, float ul, float ur, float uu, float ud, float vc, float vl,
                                     float vr, float vu, float vd, float &uo, float &vo) {
  if (KIND == K_GS) { // tau_gray_scott.cu:155-170, same association order
    // x / dx^2 as the reference writes it.  When dx^2 is a power of two (the default dx = 1) the product with the
    // exact reciprocal is the same correctly rounded result, and an IEEE divide is ~10 VALU ops — two of them were
    // 40 % of this function (wave-uniform branch, not a select: a select would still execute the divides)
    float lap_u = ur + ul + ud + uu - 4.0f * uc, lap_v = vr + vl + vd + vu - 4.0f * vc;
    if (A.dx2_pow2) { lap_u *= A.inv_dx2; lap_v *= A.inv_dx2; }
    else { lap_u /= A.dx2; lap_v /= A.dx2; }
    float uvv = uc * vc * vc;
    float du = A.Du * lap_u - uvv + A.feed * (1.0f - uc);
    float dv = A.Dv * lap_v + uvv - (A.feed + A.kill) * vc;
    uo = uc + A.dt * du;
    vo = vc + A.dt * dv;
  } else { // tau_shallow_water.cu:531-546 / tau_burgers.cu:513-521
    float du = (ur - 2.0f * uc + ul) * A.invdx2 + (ud - 2.0f * uc + uu) * A.invdy2;
    float dv = (vr - 2.0f * vc + vl) * A.invdx2 + (vd - 2.0f * vc + vu) * A.invdy2;
    const float nudt = A.dt_dev ? A.nudt * (*A.dt_dev) : A.nudt;
    uo = uc + nudt * du;
    vo = vc + nudt * dv;
    if (KIND == K_BURGERS) { uo = fasinh(uo * A.inv_u0); vo = fasinh(vo * A.inv_u0); }
  }
}

// "ud" above is the row j+1 (the reference's jp), "uu" the row j-1 (jm).
template <int KIND>
__global__ __launch_bounds__(64 * WAVES) void k_march(const Args A0) {
  const Args A = coeffs_in_vgprs<KIND>(A0);
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(A.nstrips * A.nchunks);
  unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * WAVES + (threadIdx.x >> 6);
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)A.nstrips);
  const int chunk = (int)(wid / (unsigned)A.nstrips);
  const int x4 = (strip * 64 + lane) * 4;
  const bool act = x4 < A.nx;
  const int nl = min(64, (A.nx - strip * 256) >> 2); // active lanes in this strip
  const bool first = lane == 0, last = lane == nl - 1;
  const int xl = (strip * 256 - 1 + A.nx) % A.nx;     // scalar (wave-uniform)
  const int xr = (strip * 256 + nl * 4) % A.nx;
  const int j0 = chunk * A.rows;
  const int j1 = min(j0 + A.rows, A.ny);

  Row up, cur, dn;
  load_row<KIND>(A, (j0 - 1 + A.ny) % A.ny, x4, xl, xr, act, first, last, up);
  load_row<KIND>(A, j0, x4, xl, xr, act, first, last, cur);
  load_row<KIND>(A, (j0 + 1) % A.ny, x4, xl, xr, act, first, last, dn);

  for (int j = j0; j < j1; j++) {
    Row nx2;
    const int jn = (j + 2 < A.ny) ? j + 2 : j + 2 - A.ny;
    if (j + 1 < j1) load_row<KIND>(A, jn, x4, xl, xr, act, first, last, nx2); // prefetch row j+2

    // neighbours across lanes (all lanes execute the shuffles)
    float al = __shfl_up(cur.a.w, 1, 64), ar = __shfl_down(cur.a.x, 1, 64);
    float bl = __shfl_up(cur.b.w, 1, 64), br = __shfl_down(cur.b.x, 1, 64);
    if (first) { al = cur.al; bl = cur.bl; }
    if (last) { ar = cur.ar; br = cur.br; }

    if (act) {
      float4 oa, ob;
      cell<KIND>(A, cur.a.x, al, cur.a.y, up.a.x, dn.a.x, cur.b.x, bl, cur.b.y, up.b.x, dn.b.x, oa.x, ob.x);
      cell<KIND>(A, cur.a.y, cur.a.x, cur.a.z, up.a.y, dn.a.y, cur.b.y, cur.b.x, cur.b.z, up.b.y, dn.b.y, oa.y, ob.y);
      cell<KIND>(A, cur.a.z, cur.a.y, cur.a.w, up.a.z, dn.a.z, cur.b.z, cur.b.y, cur.b.w, up.b.z, dn.b.z, oa.z, ob.z);
      cell<KIND>(A, cur.a.w, cur.a.z, ar, up.a.w, dn.a.w, cur.b.w, cur.b.z, br, up.b.w, dn.b.w, oa.w, ob.w);
      const size_t o = (size_t)j * A.nx + x4;
      if (A.nt & 1) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4f{oa.x, oa.y, oa.z, oa.w}, reinterpret_cast<v4f *>(A.oa + o));
        __builtin_nontemporal_store(v4f{ob.x, ob.y, ob.z, ob.w}, reinterpret_cast<v4f *>(A.ob + o));
      } else {
        *reinterpret_cast<float4 *>(A.oa + o) = oa;
        *reinterpret_cast<float4 *>(A.ob + o) = ob;
      }
    }
    up = cur; cur = dn; dn = nx2;
  }
}

// -------------------------------------------------------------------------------------------------
// Several steps per pass (temporal fusion) for the HBM-bound kinds (Gray-Scott, the shallow-water viscosity pass).
// One step is HBM bound at 16 B per cell update; the arithmetic leaves half the VALU idle.  Here a wave carries
// several time levels down its strip — for two levels: it loads row r of level t, forms
// row r-1 of level t+1 from the three t-rows it holds, then row r-2 of level t+2 from the three (t+1)-rows it holds,
// and only level t+2 is written — 8.6 B per cell update instead of 16 (reads x1.16 for the halos, one write per
// two updates).  Every level-(t+1) value is produced by the same cell<> arithmetic as a stand-alone step, so the
// result is bit-identical to two single steps.
//   x: a wave loads 64 lanes x 4 cells = 256 columns but owns only the inner 248 (lanes 1..62): the outer lanes
//      are the 2-cell halo of the two steps (their level-t+1 edge cells are garbage that never reaches an owned
//      cell), so there is no cross-wave exchange and every load / store is an aligned float4;
//   y: a chunk of R output rows reads R + 4 rows and forms R + 2 intermediate rows (R = 32: +12 % / +6 %).
constexpr int GS2_COLS = 248;   // owned columns per wave
struct Lvl { float4 a, b; };

template <int KIND>
__device__ __forceinline__ void row_step(const Args &A, const Lvl &up, const Lvl &cu, const Lvl &dn, Lvl &o) {
  // neighbours across lanes; the wave's outermost cells get their own value (never consumed by an owned cell)
  float al = __shfl_up(cu.a.w, 1, 64), ar = __shfl_down(cu.a.x, 1, 64);
  float bl = __shfl_up(cu.b.w, 1, 64), br = __shfl_down(cu.b.x, 1, 64);
  cell<KIND>(A, cu.a.x, al, cu.a.y, up.a.x, dn.a.x, cu.b.x, bl, cu.b.y, up.b.x, dn.b.x, o.a.x, o.b.x);
  cell<KIND>(A, cu.a.y, cu.a.x, cu.a.z, up.a.y, dn.a.y, cu.b.y, cu.b.x, cu.b.z, up.b.y, dn.b.y, o.a.y, o.b.y);
  cell<KIND>(A, cu.a.z, cu.a.y, cu.a.w, up.a.z, dn.a.z, cu.b.z, cu.b.y, cu.b.w, up.b.z, dn.b.z, o.a.z, o.b.z);
  cell<KIND>(A, cu.a.w, cu.a.z, ar, up.a.w, dn.a.w, cu.b.w, cu.b.z, br, up.b.w, dn.b.w, o.a.w, o.b.w);
}

// K time levels per pass.  Stage s holds three consecutive rows of level t+s; every trip of the march loads one
// row of level t and lets each stage produce one row of the next level from the three rows of the stage below.
// The march starts K rows early with zeroed stages: whatever the upper stages compute before real data reaches
// them is overwritten 2 trips later and never stored.
template <int KIND, int K>
__global__ __launch_bounds__(64 * WAVES) void k_fused(const Args A0) {
  const Args A = coeffs_in_vgprs<KIND>(A0);
  static_assert(K >= 2 && K <= 4, "a 4-cell halo lane covers at most 4 levels");
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(A.nstrips * A.nchunks);
  const unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * WAVES + (threadIdx.x >> 6);
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)A.nstrips);
  const int chunk = (int)(wid / (unsigned)A.nstrips);
  const int xc = strip * GS2_COLS - 4 + 4 * lane;                 // first of this lane's four columns (may lie outside)
  int xw = xc;                                                     // periodic wrap; nx % 4 == 0 keeps float4 alignment
  if (xw < 0) xw += A.nx;
  while (xw < 0) xw += A.nx;                                       // small grids: a wave reaches 252 columns past nx,
  while (xw >= A.nx) xw -= A.nx;                                   // lanes beyond it are duplicates that own nothing
  const bool owner = lane >= 1 && lane <= 62 && xc < A.nx && xc < (strip + 1) * GS2_COLS;
  const int j0 = chunk * A.rows, j1 = min(j0 + A.rows, A.ny);

  auto load = [&](int j, Lvl &r) {
    int jw = j;
    while (jw < 0) jw += A.ny;                                     // (one trip on any grid taller than K rows)
    while (jw >= A.ny) jw -= A.ny;
    const size_t base = (size_t)jw * A.nx + xw;
    r.a = *reinterpret_cast<const float4 *>(A.a + base);
    r.b = *reinterpret_cast<const float4 *>(A.b + base);
    if (KIND == K_BURGERS) { // decode once per fetched value, u = u0*sinh(phi) (:503-506); the levels in between stay decoded
      r.a.x = A.u0 * fsinh(r.a.x); r.a.y = A.u0 * fsinh(r.a.y); r.a.z = A.u0 * fsinh(r.a.z); r.a.w = A.u0 * fsinh(r.a.w);
      r.b.x = A.u0 * fsinh(r.b.x); r.b.y = A.u0 * fsinh(r.b.y); r.b.z = A.u0 * fsinh(r.b.z); r.b.w = A.u0 * fsinh(r.b.w);
    }
  };
  // Burgers: the reference re-encodes phi = asinh(u/u0) after every pass and decodes it again in the next.  The fused
  // pass does the same round trip in registers between its levels (u0 sinh(asinh(u / u0)), the very two functions the
  // single pass stores and loads through), so a K-level pass is BIT-IDENTICAL to K single passes.  TAU_ST2_BURGERS_FAST=1
  // skips the round trip (u stays decoded between levels: a deviation of ~1e-7 relative, 600 instead of ~370 Gcell/s).
  constexpr int ARITH = (KIND == K_BURGERS) ? K_SW : KIND;

  const Lvl zero{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  Lvl st[K][3];            // st[s][0..2] = rows (r-2-s, r-1-s, r-s) of level t+s, r = newest level-t row
#pragma unroll
  for (int s = 0; s < K; s++) st[s][0] = st[s][1] = st[s][2] = zero;
  Lvl nxt;
  load(j0 - K, st[0][1]);
  load(j0 - K + 1, st[0][2]);
  load(j0 - K + 2, nxt);
  // trip r: newest level-t row is r; stage s then holds level t+s rows up to r-s; the output row is r-K
  for (int r = j0 - K + 2; r < j1 + K; r++) {
    st[0][0] = st[0][1]; st[0][1] = st[0][2]; st[0][2] = nxt;
    if (r + 1 < j1 + K) load(r + 1, nxt);                          // prefetch
    Lvl out;
#pragma unroll
    for (int s = 0; s < K; s++) {
      Lvl o;
      row_step<ARITH>(A, st[s][0], st[s][1], st[s][2], o);         // level t+s+1, row r-1-s
      if (s + 1 < K) {
        if (KIND == K_BURGERS && !A.burgers_fast) {                  // what the next single pass would load: decode(encode(u))
          o.a.x = A.u0 * fsinh(fasinh(o.a.x * A.inv_u0)); o.a.y = A.u0 * fsinh(fasinh(o.a.y * A.inv_u0));
          o.a.z = A.u0 * fsinh(fasinh(o.a.z * A.inv_u0)); o.a.w = A.u0 * fsinh(fasinh(o.a.w * A.inv_u0));
          o.b.x = A.u0 * fsinh(fasinh(o.b.x * A.inv_u0)); o.b.y = A.u0 * fsinh(fasinh(o.b.y * A.inv_u0));
          o.b.z = A.u0 * fsinh(fasinh(o.b.z * A.inv_u0)); o.b.w = A.u0 * fsinh(fasinh(o.b.w * A.inv_u0));
        }
        st[s + 1][0] = st[s + 1][1]; st[s + 1][1] = st[s + 1][2]; st[s + 1][2] = o;
      } else out = o;
    }
    const int j = r - K;                                           // row of level t+K just produced
    if (owner && j >= j0) {
      if (KIND == K_BURGERS) { // :521
        out.a.x = fasinh(out.a.x * A.inv_u0); out.a.y = fasinh(out.a.y * A.inv_u0); out.a.z = fasinh(out.a.z * A.inv_u0); out.a.w = fasinh(out.a.w * A.inv_u0);
        out.b.x = fasinh(out.b.x * A.inv_u0); out.b.y = fasinh(out.b.y * A.inv_u0); out.b.z = fasinh(out.b.z * A.inv_u0); out.b.w = fasinh(out.b.w * A.inv_u0);
      }
      const size_t off = (size_t)j * A.nx + xc;
      typedef float v4f __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(v4f{out.a.x, out.a.y, out.a.z, out.a.w}, reinterpret_cast<v4f *>(A.oa + off));
      __builtin_nontemporal_store(v4f{out.b.x, out.b.y, out.b.z, out.b.w}, reinterpret_cast<v4f *>(A.ob + off));
    }
  }
}

// any nx (not a multiple of 4): one thread per cell, wraps on index compare (no %)
template <int KIND>
__global__ __launch_bounds__(256) void k_simple(const Args A) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= A.nx || j >= A.ny) return;
  const int ip = (i + 1 == A.nx) ? 0 : i + 1, im = (i == 0) ? A.nx - 1 : i - 1;
  const int jp = (j + 1 == A.ny) ? 0 : j + 1, jm = (j == 0) ? A.ny - 1 : j - 1;
  auto ld = [&](const float *f, int jj, int ii) {
    float v = f[(size_t)jj * A.nx + ii];
    return (KIND == K_BURGERS) ? A.u0 * fsinh(v) : v;
  };
  float uo, vo;
  cell<KIND>(A, ld(A.a, j, i), ld(A.a, j, im), ld(A.a, j, ip), ld(A.a, jm, i), ld(A.a, jp, i), ld(A.b, j, i),
             ld(A.b, j, im), ld(A.b, j, ip), ld(A.b, jm, i), ld(A.b, jp, i), uo, vo);
  A.oa[(size_t)j * A.nx + i] = uo;
  A.ob[(size_t)j * A.nx + i] = vo;
}

template <int KIND>
static int launch(const Args &Ain, hipStream_t s) {
  Args A = Ain;
  if ((A.nx & 3) == 0) {
    static const int env_rows = getenv("TAU_ST2_ROWS") ? atoi(getenv("TAU_ST2_ROWS")) : 0;
    static const int env_nt = getenv("TAU_ST2_NT") ? atoi(getenv("TAU_ST2_NT")) : 1;   // streaming (nt) stores: +2-4 %
    A.rows = env_rows > 0 ? env_rows : (KIND == K_BURGERS ? ROWS : ROWS_HBM);
    A.nt = env_nt;
    A.nstrips = (A.nx + 255) / 256;
    A.nchunks = (A.ny + A.rows - 1) / A.rows;
    unsigned nwork = (unsigned)(A.nstrips * A.nchunks);
    unsigned nb = (nwork + WAVES - 1) / WAVES;
    hipLaunchKernelGGL(k_march<KIND>, dim3(nb), dim3(64 * WAVES), 0, s, A);
  } else {
    hipLaunchKernelGGL(k_simple<KIND>, dim3((A.nx + 63) / 64, (A.ny + 3) / 4), dim3(256), 0, s, A);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tau::fail("stencil2d launch: %s", hipGetErrorString(e));
  return 0;
}

struct Pair {
  int device;
  hipStream_t stream;
  bool own_stream;
  int nx, ny;
  float *buf[2][2];
  int cur;
  int levels = 0;   // time levels per pass: 0 = the default (TAU_ST2_LEVELS, 4), 1 = one launch per step, 2..4
};

static int pair_create(Pair *h, int nx, int ny, int device, void *stream) {
  if (nx < 2 || ny < 2) return tau::fail("stencil2d: grid must be at least 2x2");
  TAU_HIP(hipSetDevice(device));
  h->device = device; h->nx = nx; h->ny = ny; h->cur = 0;
  h->own_stream = (stream == nullptr);
  if (h->own_stream) TAU_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  else h->stream = (hipStream_t)stream;
  size_t bytes = (size_t)nx * ny * sizeof(float);
  for (int s = 0; s < 2; s++)
    for (int f = 0; f < 2; f++) TAU_HIP(hipMalloc(&h->buf[s][f], bytes));
  return 0;
}
static void pair_destroy(Pair *h) {
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  for (int s = 0; s < 2; s++)
    for (int f = 0; f < 2; f++) hipFree(h->buf[s][f]);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
}
static int pair_upload(Pair *h, const float *a, const float *b) {
  TAU_HIP(hipSetDevice(h->device));
  size_t bytes = (size_t)h->nx * h->ny * sizeof(float);
  TAU_HIP(hipMemcpyAsync(h->buf[h->cur][0], a, bytes, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipMemcpyAsync(h->buf[h->cur][1], b, bytes, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
static int pair_download(Pair *h, float *a, float *b) {
  TAU_HIP(hipSetDevice(h->device));
  size_t bytes = (size_t)h->nx * h->ny * sizeof(float);
  TAU_HIP(hipMemcpyAsync(a, h->buf[h->cur][0], bytes, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipMemcpyAsync(b, h->buf[h->cur][1], bytes, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// n steps of an HBM-bound kind: passes of up to `kmax` fused time levels, the remainder as single steps.
// TAU_ST2_FUSE=0 disables, TAU_ST2_LEVELS=2..4 (default 4), TAU_ST2_FROWS = output rows per wave (default 32).
template <int KIND>
static int run_steps(Pair *pr, Args A, int nsteps) {
  static const bool fuse_env = !(getenv("TAU_ST2_FUSE") && atoi(getenv("TAU_ST2_FUSE")) == 0);
  static const int frows = getenv("TAU_ST2_FROWS") ? atoi(getenv("TAU_ST2_FROWS")) : 32;
  static const int kmax_env = getenv("TAU_ST2_LEVELS") ? atoi(getenv("TAU_ST2_LEVELS")) : 4;
  // Burgers: three levels per pass (91 VGPRs, five waves per SIMD; the sinh / asinh round trip between levels is VALU work that
  // four levels at 115 VGPRs / four waves hide less well: 302 against 289 Gcell/s at 8192^2), Gray-Scott / shallow water: four
  const int kmax = pr->levels > 0 ? pr->levels : (getenv("TAU_ST2_LEVELS") ? kmax_env : (KIND == K_BURGERS ? 3 : 4));
  const bool fuse = fuse_env && kmax >= 2 && kmax <= 4 && (A.nx & 3) == 0 && A.nx >= 8 && A.ny >= 2;
  static const int bfast = getenv("TAU_ST2_BURGERS_FAST") ? atoi(getenv("TAU_ST2_BURGERS_FAST")) : 0;
  A.burgers_fast = bfast;
  int s = 0;
  while (s < nsteps) {
    A.a = pr->buf[pr->cur][0]; A.b = pr->buf[pr->cur][1];
    A.oa = pr->buf[pr->cur ^ 1][0]; A.ob = pr->buf[pr->cur ^ 1][1];
    const int left = nsteps - s;
    const int K = !fuse ? 1 : (left >= kmax ? kmax : (left >= 2 ? left : 1));
    if (K >= 2) {
      Args B = A;
      B.nstrips = (B.nx + GS2_COLS - 1) / GS2_COLS;
      // 32 output rows per wave on big grids (the 2K halo rows are re-read and re-computed); small grids are latency
      // bound and want many short waves instead: at least ~2 k waves, down to 4 rows each
      B.rows = frows > 0 ? frows : 32;
      if (frows <= 0 || !getenv("TAU_ST2_FROWS")) {
        long r = (long)B.ny * B.nstrips / 2048;
        B.rows = r >= 32 ? 32 : (r < 4 ? 4 : (int)r);
      }
      B.nchunks = (B.ny + B.rows - 1) / B.rows;
      const unsigned nwork = (unsigned)(B.nstrips * B.nchunks), nb = (nwork + WAVES - 1) / WAVES;
      if (K == 2) hipLaunchKernelGGL((k_fused<KIND, 2>), dim3(nb), dim3(64 * WAVES), 0, pr->stream, B);
      else if (K == 3) hipLaunchKernelGGL((k_fused<KIND, 3>), dim3(nb), dim3(64 * WAVES), 0, pr->stream, B);
      else hipLaunchKernelGGL((k_fused<KIND, 4>), dim3(nb), dim3(64 * WAVES), 0, pr->stream, B);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return tau::fail("st2::k_fused launch: %s", hipGetErrorString(e));
      s += K;
    } else {
      if (launch<KIND>(A, pr->stream)) return 1;
      s += 1;
    }
    pr->cur ^= 1; // one pointer swap per pass: a fused pass leaves level t+K in the other buffer
  }
  return 0;
}

} // namespace st2

int tau::st2_burgers_pass(const float *a, const float *b, float *oa, float *ob, int nx, int ny, float dx, float dy, float nu,
                          float u0, int oneD, const float *dt_dev, float frac, hipStream_t stream) {
  st2::Args A{};
  A.a = a; A.b = b; A.oa = oa; A.ob = ob; A.nx = nx; A.ny = ny;
  A.invdx2 = 1.0f / (dx * dx); A.invdy2 = oneD ? 0.0f : 1.0f / (dy * dy);
  A.nudt = nu * frac; A.u0 = u0; A.inv_u0 = 1.0f / u0;
  A.dt_dev = dt_dev;   // fl2::DevState::dt_last of the step in flight
  return st2::launch<st2::K_BURGERS>(A, stream);
}

// =====================================================================================
// Gray-Scott C-ABI
// =====================================================================================
struct taugs {
  st2::Pair pr;
  taugs_params p;
};

extern "C" void taugs_params_default(taugs_params *P, int nx, int ny) { // tau_gray_scott.cu:43-61
  P->nx = nx; P->ny = ny; P->dx = 1.0f; P->dt = 1.0f;
  P->Du = 0.2f; P->Dv = 0.1f; P->feed = 0.03f; P->kill = 0.06f;
}
extern "C" int taugs_create(taugs_t **out, const taugs_params *p, int device, void *stream) {
  if (!out || !p) return tau::fail("taugs_create: null argument");
  taugs *h = new (std::nothrow) taugs();
  if (!h) return tau::fail("taugs_create: out of host memory");
  h->p = *p;
  if (st2::pair_create(&h->pr, p->nx, p->ny, device, stream)) { taugs_destroy(h); return 1; }
  *out = h;
  return 0;
}
extern "C" void taugs_destroy(taugs_t *h) { if (h) { st2::pair_destroy(&h->pr); delete h; } }

/* init_pattern (:173-204) into host arrays of nx * ny floats: what a row-slab rank cuts its rows out of */
extern "C" int taugs_pattern_host(int nx, int ny, uint32_t seed, float *u, float *v) {
  if (nx < 1 || ny < 1 || !u || !v) return tau::fail("taugs_pattern_host: bad argument");
  for (size_t i = 0; i < (size_t)nx * ny; i++) { u[i] = 1.0f; v[i] = 0.0f; }
  const int cx = nx / 2, cy = ny / 2, r = (nx < ny ? nx : ny) / 12;
  for (int j = -r; j <= r; ++j)
    for (int i = -r; i <= r; ++i) {
      int x = (cx + i + nx) % nx, y = (cy + j + ny) % ny;
      u[(size_t)y * nx + x] = 0.50f;
      v[(size_t)y * nx + x] = 0.25f;
    }
  uint32_t state = seed ? seed : 1u;
  auto rng = [&]() { state ^= state << 13; state ^= state >> 17; state ^= state << 5; return state; };
  for (int n = 0; n < 64; ++n) {
    int x = (int)(rng() % (uint32_t)nx);
    int y = (int)(rng() % (uint32_t)ny);
    u[(size_t)y * nx + x] = 0.35f;
    v[(size_t)y * nx + x] = 0.65f;
  }
  return 0;
}
extern "C" int taugs_init_pattern(taugs_t *h, uint32_t seed) { // init_pattern, :173-204 (host) + H2D :308-309
  const int nx = h->p.nx, ny = h->p.ny;
  std::vector<float> u((size_t)nx * ny), v((size_t)nx * ny);
  if (taugs_pattern_host(nx, ny, seed, u.data(), v.data())) return 1;
  return st2::pair_upload(&h->pr, u.data(), v.data());
}
extern "C" int taugs_info(taugs_t *h, int *nx, int *ny, int *device, void **stream) {
  if (!h) return tau::fail("taugs_info: null handle");
  if (nx) *nx = h->pr.nx;
  if (ny) *ny = h->pr.ny;
  if (device) *device = h->pr.device;
  if (stream) *stream = (void *)h->pr.stream;
  return 0;
}
extern "C" int taugs_upload(taugs_t *h, const float *u, const float *v) { return st2::pair_upload(&h->pr, u, v); }
extern "C" int taugs_download(taugs_t *h, float *u, float *v) { return st2::pair_download(&h->pr, u, v); }
extern "C" int taugs_state_ptrs(taugs_t *h, float **u, float **v) {
  *u = h->pr.buf[h->pr.cur][0]; *v = h->pr.buf[h->pr.cur][1];
  return 0;
}
extern "C" int taugs_step_async(taugs_t *h, int nsteps) {
  TAU_HIP(hipSetDevice(h->pr.device));
  st2::Args A{};
  A.nx = h->p.nx; A.ny = h->p.ny;
  A.dx2 = h->p.dx * h->p.dx; A.dt = h->p.dt; A.Du = h->p.Du; A.Dv = h->p.Dv; A.feed = h->p.feed; A.kill = h->p.kill;
  {
    int e = 0;
    const float m = frexpf(A.dx2, &e);
    A.dx2_pow2 = (m == 0.5f && e > -100 && e < 100) ? 1 : 0;   // 2^(e-1): its reciprocal is exact and normal
    A.inv_dx2 = 1.0f / A.dx2;
  }
  return st2::run_steps<st2::K_GS>(&h->pr, A, nsteps);   // std::swap per pass, :327-328
}
extern "C" int taugs_sync(taugs_t *h) {
  TAU_HIP(hipSetDevice(h->pr.device));
  TAU_HIP(hipStreamSynchronize(h->pr.stream));
  return 0;
}
extern "C" int taugs_set_levels(taugs_t *h, int levels) {
  if (!h || levels < 0 || levels > 4) return tau::fail("taugs_set_levels: levels must be 0 (default), 1 (one launch per step) or 2..4");
  h->pr.levels = levels;
  return 0;
}
extern "C" int taugs_step(taugs_t *h, int nsteps) {
  if (taugs_step_async(h, nsteps)) return 1;
  return taugs_sync(h);
}

// =====================================================================================
// Laplacian viscosity passes C-ABI
// =====================================================================================
struct taulap {
  st2::Pair pr;
  taulap_params p;
  int kind, oneD;
};

extern "C" int taulap_create(taulap_t **out, const taulap_params *p, int kind, int oneD, int device, void *stream) {
  if (!out || !p) return tau::fail("taulap_create: null argument");
  if (kind != 0 && kind != 1) return tau::fail("taulap_create: kind must be 0 (Burgers) or 1 (shallow water)");
  taulap *h = new (std::nothrow) taulap();
  if (!h) return tau::fail("taulap_create: out of host memory");
  h->p = *p; h->kind = kind; h->oneD = oneD;
  if (st2::pair_create(&h->pr, p->nx, p->ny, device, stream)) { taulap_destroy(h); return 1; }
  *out = h;
  return 0;
}
extern "C" void taulap_destroy(taulap_t *h) { if (h) { st2::pair_destroy(&h->pr); delete h; } }
extern "C" int taulap_info(taulap_t *h, int *nx, int *ny, int *device, void **stream) {
  if (!h) return tau::fail("taulap_info: null handle");
  if (nx) *nx = h->pr.nx;
  if (ny) *ny = h->pr.ny;
  if (device) *device = h->pr.device;
  if (stream) *stream = (void *)h->pr.stream;
  return 0;
}
extern "C" int taulap_upload(taulap_t *h, const float *a, const float *b) { return st2::pair_upload(&h->pr, a, b); }
extern "C" int taulap_download(taulap_t *h, float *a, float *b) { return st2::pair_download(&h->pr, a, b); }
extern "C" int taulap_state_ptrs(taulap_t *h, float **a, float **b) {
  *a = h->pr.buf[h->pr.cur][0]; *b = h->pr.buf[h->pr.cur][1];
  return 0;
}
extern "C" int taulap_set_dt(taulap_t *h, float dt) { h->p.dt = dt; return 0; }
extern "C" int taulap_step_async(taulap_t *h, int npasses) {
  TAU_HIP(hipSetDevice(h->pr.device));
  st2::Args A{};
  A.nx = h->p.nx; A.ny = h->p.ny;
  A.invdx2 = 1.0f / (h->p.dx * h->p.dx);
  A.invdy2 = (h->kind == 0 && h->oneD) ? 0.0f : 1.0f / (h->p.dy * h->p.dy);
  A.nudt = h->p.nu * h->p.dt;
  A.u0 = h->p.u0; A.inv_u0 = 1.0f / h->p.u0;
  if (h->kind != 0) return st2::run_steps<st2::K_SW>(&h->pr, A, npasses);
  return st2::run_steps<st2::K_BURGERS>(&h->pr, A, npasses);
}
extern "C" int taulap_sync(taulap_t *h) {
  TAU_HIP(hipSetDevice(h->pr.device));
  TAU_HIP(hipStreamSynchronize(h->pr.stream));
  return 0;
}
extern "C" int taulap_step(taulap_t *h, int npasses) {
  if (taulap_step_async(h, npasses)) return 1;
  return taulap_sync(h);
}
