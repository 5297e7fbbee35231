// h3d.hip — 3D two-temperature hypersonic Euler step for gfx950 (MI355X).
//
// What it computes: one k_step of the reference (tau_hypersonic_3d_cuda.cu:987-1359):
// decode the log/asinh-encoded state, WENO5 + blended HLLC/HLL face fluxes, conservative
// update, Landau-Teller relaxation, sponges, max wavespeed, re-encode.
//
// How it is laid out for CDNA4 (nothing here follows the reference's launch shape):
//   * one 256-thread workgroup owns a 32x8 (x,y) column and MARCHES along z through a chunk
//     of planes.  Each thread keeps its own column's 6-plane z-window of primitives in
//     VGPRs, so only the CURRENT plane (plus its 3-cell x/y halo) lives in LDS: 13 KB
//     instead of the reference's 49 KB 3-D tile, and every cell is decoded ~2x per step
//     instead of 7.7x.
//   * every face flux is computed ONCE (the reference computes each face from both
//     sides): a thread computes the low-x and low-y face of its cell and the high-z face;
//     the high-x / high-y fluxes come from the neighbour thread through a small LDS flux
//     tile; the low-z flux is carried in registers from the previous plane.  The 40 faces
//     on the far edges of the tile are done by one extra round of the last wave.
//   * lanes run along x (32 consecutive floats = one 128-B segment per row), a wave64
//     covers two rows.
//   * divisions are v_rcp_f32, exp/log are v_exp_f32/v_log_f32 (the reference uses
//     __expf/__logf too); sinh uses a series below |x|<0.5 so small transverse velocities
//     keep their relative accuracy.
//   * dt / inflow gain are read from a device-side clock block, the max wavespeed goes
//     back through one atomicMax per workgroup: the d_tau controller never leaves the GPU.
//
// Bound: this kernel is FP32-VALU bound (≈2.6 k VALU instr per cell), not HBM bound —
// algorithmic traffic is 49 B/cell (6 fp32 in + 1 u8 + 6 fp32 out).

#include "../../include/taueng.h"
#include "tau_common.h"
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <new>
#include <type_traits>

// Tuning overrides (tile shapes, wave caps, LDS strides) are for scripts/variant_build.sh, which defines TAU_EXPERIMENT: a product
// build line that carries one of them by accident stops here instead of shipping another kernel.  (No macro of this file changes
// WHAT is computed any more: the wrong-result timing variants of rounds 3-5 are gone, their A/Bs are under profiles/.)
#if !defined(TAU_EXPERIMENT) && (defined(TAU3D_STEP_WAVES) || defined(TAU3D_TY) || defined(TAU3D_XY_TY) || defined(TAU3D_XPXS) || \
    defined(TAU3D_XY_WAVES) || defined(TAU3D_Z_WAVES) || defined(TAU3D_FAST_ONLY))
#error "TAU3D_* tuning overrides need -DTAU_EXPERIMENT (scripts/variant_build.sh sets it)"
#endif

namespace h3d {

#ifndef TAU3D_STEP_WAVES
#define TAU3D_STEP_WAVES 2
#endif
constexpr int HALO = 3;            // WENO_HALO, tau_hypersonic_3d_cuda.cu:58
// TY = 12 (six waves: the 56 edge faces would be shared by six waves instead of 48 by four) was measured: a
// six-wave workgroup lands 2+2+1+1 on the four SIMDs and a second one no longer fits under the 3-waves-per-SIMD
// register limit — 9.2 Gcell/s against 16.0.
#ifndef TAU3D_TY
#define TAU3D_TY 8
#endif
constexpr int TX = 32, TY = TAU3D_TY; // workgroup tile in (x, y)
constexpr int NT = TX * TY;        // threads: TY / 2 waves
constexpr int NW = NT / 64;
static_assert(TY % 2 == 0 && 2 * TY + TX <= 64 && 2 * HALO * (TX + TY) <= NT, "edge round is one wave, halo decode one round");
constexpr int PX = TX + 2 * HALO;  // 38
constexpr int PY = TY + 2 * HALO;  // 14
constexpr int PXS = 40;            // padded LDS row stride (floats)
constexpr int PLANE = PY * PXS;    // floats per variable per plane

constexpr float RHO_P_FLOOR = 1e-30f;          // :52
constexpr float THERMAL_ENERGY_FLOOR = 1e-12f; // :53
constexpr float DENOM_EPS = 1e-12f;            // :54
constexpr float NEWTON_TEMP_FLOOR = 1e-6f;     // :55
constexpr float WENO_EPS = 1e-6f;              // :56
constexpr float TAU_VIB_MIN = 1e-9f;           // :57

enum { IR = 0, IU = 1, IV = 2, IW = 3, IP = 4, IE = 5 };

struct Prim { float q[6]; };   // r, u, v, w, p, ev
struct Cons { float c[6]; };   // r, mx, my, mz, Et, Ev

// device-side clock block (one per handle)
struct DevClock {
  float t, d_tau, dt, gain, maxs_last;
  int step;
  float cfl;
  unsigned pad_;        // keeps the two words below on a 32-byte boundary (they are handed to RCCL as one tensor)
  // ---- tau3d_set_clock writes the words above only
  unsigned maxs_bits;   // max wavespeed of the step in flight, as float bits (>0 floats order as uints)
  unsigned fmax_bits;   // largest |primitive| written by the step in flight (same encoding).  The word after
                        // maxs_bits: tau3d_max_ptr hands out both and a slab ring all-reduces them together
  float fmax_in;        // largest |primitive| in the state the next k_step reads (committed by clock_begin)
  unsigned form_flip;   // 1 when that commit moved fmax_in across the WENO weight-form limit: an x/y flux launch that ran AHEAD of
                        // the commit (the ring's speculative launch, tau3d_slab_xy_async before the clock) took the other form and
                        // is repeated (k_flux_xy_fix)
};

static_assert(offsetof(DevClock, maxs_bits) % 32 == 0 && offsetof(DevClock, fmax_bits) == offsetof(DevClock, maxs_bits) + 4,
              "tau3d_max_ptr hands out {maxs_bits, fmax_bits} as one aligned 2-word tensor");

// kernel arguments (by value -> SGPRs)
constexpr unsigned UF_ALL = 1u, UF_W = 2u, UF_E = 4u, UF_S = 8u, UF_N = 16u, UF_PRED = 32u;
constexpr int UREC = 24, USTRIP = 6;   // floats per tile record; width of an edge strip = the reach of two steps' stencils (2 x HALO)
struct Args {
  const float *in[6];
  float *out[6];
  const uint8_t *solid;      // halo layout, plane -3 first
  DevClock *clk;
  // split step (k_flux_xy + k_update_z): the x/y flux divergence of the local planes (no halo)
  float *dxy[6];
  float *send[2];            // Z-slab ring: packed send buffers the step writes its new boundary planes into (or null)
  const unsigned *xyflag;    // k_flux_xy: [local plane][tile row][tile column] != 0 where the tile + its x/y halo holds a solid cell (or null: assume so)
  unsigned *dzero;           // uniform-region exits (round 6): [local plane][tile row][tile column], written by k_flux_xy every step — 1: every cell of
                             // the tile and of its x/y halo holds the same state, the tile's x/y divergence is exactly zero and was NOT stored;
                             // read by k_update_z in place of the divergence.  null: exits off (TAU3D_UNIFORM_EXITS=0)
  int dz_ntx, dz_nty;        // its tile grid (k_flux_xy's tiles: XT x YT)
  // Predicted-uniform tiles (round 6): k_flux_xy records, per tile, whether it found tile + halo uniform and with which encoded state
  // (uflag_w / uref_w: this step's, written; [tile] and [tile][UREC]); after k_update_z, k_tile_predict reads them (uflag_r / uref_r) and
  // decides which tiles of the NEXT state can only be uniform again — every tile around them, seven planes deep, held the same one
  // state — sets their dzero flag itself and lists the others (ulist, ucount): the next k_flux_xy is launched over the LIST.
  // A flag word: UF_ALL — tile + halo hold ONE encoded state (record words 0-5: the tile's first cell); else UF_W / UF_S — its first six
  // columns / rows do, with their 3-cell halo rows / columns (same record), UF_E / UF_N — its last six columns / rows (record words
  // 8-13: the first row's last cell / 16-21: the last row's first cell); UF_PRED: flagged by k_tile_predict, not by a k_flux_xy.
  unsigned *uflag_w; float *uref_w;
  const unsigned *uflag_r; const float *uref_r;
  unsigned *pflag; float *pref;           // k_tile_predict: the NEXT step's flags / states, written where it predicts (pred_commit: and dzero)
  unsigned *ulist; unsigned *ucount;      // the list and its length (k_tile_predict appends; k_flux_xy_list reads)
  unsigned *ucount_other;                 // the length word of the step before: k_tile_predict zeroes it for the step after
  unsigned *ucount_host;                  // k_flux_xy_list: the length again, in mapped host memory (the host sizes later launches by it)
  unsigned list_grid; int list_fast;      // k_flux_xy_list_rest: the main launch's grid and weight form
  int z_fill;                             // k_update_z: 0, or k_fill_z follows and takes the fully predicted chunks (1: if the fast weight form is due, 2: the other)
  int z_pred;                             // k_update_z: bits 1 / 2 of a dzero word are this step's prediction (k_tile_predict ran before it) —
                                          // 0: no, 1: yes but every plane is marched all the same (TAU3D_Z_SKIP=0), 2: predicted planes store the record's state
  int pred_commit;                        // 0: the verifying mode — flags into a scratch pair, the next k_flux_xy still runs every tile
  // the same groups as one base + stride (field m at base + m * stride): what k_update_z addresses them through
  const float *in0;
  float *out0;
  const float *d0;
  unsigned fstride, dstride; // floats
  int nx, ny, nz;            // global
  int nzl, z0;               // local planes, global index of local plane 0
  int zl_lo, zl_hi;          // local plane range to update
  int zl_lo2, zl_hi2, nzc1;  // optional second range in the same launch (chunks >= nzc1 belong to it); nzc1 = nzc if unused
  int zchunk;                // planes marched by one workgroup
  int wrap_halo;             // k_update_z (single periodic domain): the first / last three new planes also go into the opposite z halo
  int ntx, nty, nzc;         // tiles in x, y; chunks in z
  float dx, dy, dz, inv_dx, inv_dy, inv_dz;
  float u_ref, inv_u_ref, R, gamma, gm1, inv_gm1, Twall, theta_v, Rtheta, inv_tau_vib;
  float sdf_cx, sdf_cy, sdf_cz, sdf_r;
  float in_r, in_u, in_v, in_w, in_p, in_ev;   // inflow_prim(), :611-622 (ev from host expf)
  float in_fmax;                               // largest |component| of it
  int sponge_n, sponge_out_n;
  float sponge_strength, sponge_out_strength;
};

// ---------------------------------------------------------------- fast math
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
__device__ __forceinline__ float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
__device__ __forceinline__ float denom_guard(float x) { return copysignf(fmaxf(fabsf(x), DENOM_EPS), x); } // :147-150

// Constants that a hot VALU expression multiplies by or masks with are kept out of SGPRs: an SGPR source halves the issue
// rate on gfx950 (profiles/r02/valu_calib.txt), and a VOP3 instruction (fma, bfi, anything with |x|) cannot carry a literal,
// so hipcc parks 1/6, 1/20, log2 e, 0x7fffffff ... in SGPRs.  VOP2 forms (mul / add with a 32-bit literal) and inline
// constants (1.0, 0.5, 2.0) issue at full rate.
// stops a mul from being contracted into an fma that would need the constant in an SGPR.  NOT volatile: a volatile asm keeps
// its place among the other volatile asms (tau::gld's base pins), and inside a decode that serialises the six loads of a
// cell — measured: k_flux_xy 3.96 -> 4.37 ms
__device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ float vlit(unsigned bits) {
  float v;
  asm("v_mov_b32 %0, %1" : "=v"(v) : "i"(bits));
  return v;
}
// magnitude of `mag`, sign of `sgn`; maskv = 0x7fffffff held in a VGPR (vlit)
__device__ __forceinline__ float copysign_v(float mag, float sgn, float maskv) {
  float r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(maskv), "v"(mag), "v"(sgn));
  return r;
}
// sinh with full relative accuracy for small arguments (the reference calls sinhf, :119): the odd series to x^7 below 0.5
// (truncation 2.7e-9 relative), (e^x - e^-x) / 2 above.  Signed exponential: no copysign; Horner on literal multiplies.
__device__ __forceinline__ float fsinh_series(float x) {
  const float x2 = x * x;
  const float a = opaque(x2 * (1.f / 42.f)) + 1.f;
  const float c = (x2 * (1.f / 20.f)) * a + 1.f;
  const float s = (x2 * (1.f / 6.f)) * c + 1.f;
  return x * s;
}
__device__ __forceinline__ float fsinh_big(float x) {
  const float e = fexp(x);
  return 0.5f * (e - rcp(e));
}
__device__ __forceinline__ float fsinh(float x) {
  const float series = fsinh_series(x), big = fsinh_big(x);
  return (fabsf(x) < 0.5f) ? series : big;
}
// The same value in every lane, but the wave evaluates only the forms its lanes take (round 5): fsinh computes both — 8 VALU for
// the series, 5 with two transcendentals (~30 cycles) for the exponential form — and selects; the lanes of a wave are 64
// neighbouring cells of a row, and a velocity component sits on one side of |u| = 0.52 u_ref for all of them almost everywhere
// (the transverse components below it, the streamwise one above it outside the wake's core).  Two scalar branches on ballots;
// a mixed wave falls back to fsinh's select.  Call it from code whose loads are already waited for: a branch pins the arithmetic
// where it is written (the z kernel decodes a plane when it enters the ring, not where it is fetched).
__device__ __forceinline__ float fsinh_wave(float x) {
  const bool small = fabsf(x) < 0.5f;
  const unsigned long long sm = __builtin_amdgcn_ballot_w64(small), act = __builtin_amdgcn_ballot_w64(true);
  float r;
  if (sm == act) r = fsinh_series(x);
  else if (sm == 0ull) r = fsinh_big(x);
  else r = small ? fsinh_series(x) : fsinh_big(x);
  return r;
}
// asinh as the reference writes it, :121-125.  maskv: 0x7fffffff in a VGPR (vlit), or use the two-argument form
__device__ __forceinline__ float fasinh(float x, float maskv) {
  float ax = fabsf(x);
  float t = flog(ax + fsqrt(ax * ax + 1.0f));
  return copysign_v(t, x, maskv);
}
__device__ __forceinline__ float fasinh(float x) {
  float ax = fabsf(x);
  float t = flog(ax + fsqrt(ax * ax + 1.0f));
  return copysignf(t, x);
}

__device__ __forceinline__ float evib_eq(const Args &A, float T) { // :206-211
  float a = A.theta_v * rcp(fmaxf(T, NEWTON_TEMP_FLOOR));
  float ea = fexp(a);
  float denom = fmaxf(ea - 1.f, NEWTON_TEMP_FLOOR);
  return A.Rtheta * rcp(denom);
}

__device__ __forceinline__ Prim decode(const Args &A, size_t gi) { // log_to_prim_fast, :213-225
  Prim q;
  q.q[IR] = fexp(A.in[0][gi]);
  q.q[IU] = A.u_ref * fsinh(A.in[1][gi]);
  q.q[IV] = A.u_ref * fsinh(A.in[2][gi]);
  q.q[IW] = A.u_ref * fsinh(A.in[3][gi]);
  q.q[IP] = fexp(A.in[4][gi]);
  q.q[IE] = fexp(A.in[5][gi]);
  return q;
}

__device__ __forceinline__ Prim inflow_prim(const Args &A) {
  Prim q;
  q.q[IR] = A.in_r; q.q[IU] = A.in_u; q.q[IV] = A.in_v; q.q[IW] = A.in_w; q.q[IP] = A.in_p; q.q[IE] = A.in_ev;
  return q;
}

__device__ __forceinline__ float soundspeed(const Args &A, float p, float r) { // :264-266
  return fsqrt(fmaxf(A.gamma * p * rcp(r), DENOM_EPS));
}

// ghost state right of the last interior cell, :691-722
__device__ __forceinline__ Prim outflow_prim(const Args &A, Prim qR) {
  Prim q = qR;
  float aR = soundspeed(A, qR.q[IP], qR.q[IR]);
  float un = qR.q[IU];
  if (un < aR) q.q[IP] = fmaxf(q.q[IP] + 0.05f * (A.in_p - q.q[IP]), RHO_P_FLOOR);
  q.q[IR] = fmaxf(q.q[IR], RHO_P_FLOOR);
  q.q[IP] = fmaxf(q.q[IP], RHO_P_FLOOR);
  q.q[IE] = fmaxf(q.q[IE], 0.f);
  if (un < 0.0f) q = inflow_prim(A);
  return q;
}

// Solid classification of a cell centre, :173-189.  This is an INTEGER result (the mask must be bit-identical to
// the oracle's), and cell centres do land exactly on the sphere (z = 37 of 50 planes: (37.5)/50 - 0.5 = r): there a
// fused multiply-add rounds the other way than mul-then-sub and flips the cell.  Contraction is therefore switched
// off for this function (HIP's __fmul_rn & co. are plain operators and do not prevent it).
__device__ __forceinline__ bool sdf_solid(const Args &A, int x, int y, int zg) {
#pragma clang fp contract(off)
  const float X = (x + 0.5f) * A.dx, Y = (y + 0.5f) * A.dy, Z = (zg + 0.5f) * A.dz;
  const float ddx = X - A.sdf_cx, ddy = Y - A.sdf_cy, ddz = Z - A.sdf_cz;
  return (sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) - A.sdf_r) < 0.f;
}

__device__ __forceinline__ int wrapi(int i, int n) { i %= n; return (i < 0) ? i + n : i; }
// the same for -n <= i < 2n without the integer division (~22 VALU instructions): k_flux_xy's rows, which overhang the
// grid by at most a tile + the halo; `nearby` is a kernel-uniform flag (the row count covers that overhang)
__device__ __forceinline__ int wrap_near(int i, int n, bool nearby) {
  if (!nearby) return wrapi(i, n);
  i = (i < 0) ? i + n : i;
  return (i >= n) ? i - n : i;
}

// ---------------------------------------------------------------- WENO5, :534-558
// The reference evaluates weno5_left(v0..v4) for the left state and weno5_left(v5..v1) for the
// right state of every face.  Two algebraically identical re-groupings are used here:
//  * weno_face: both states of ONE face from its six cells, sharing the two second differences
//    the two stencils have in common (x and y faces, whose cells come from LDS);
//  * weno_cell: both edge states of ONE cell (left state of its high face, right state of its low
//    face) from its five cells — the three smoothness indicators are the same for both
//    (b0' = b2, b1' = b1, b2' = b0), so the marching (z) axis pays for them once per cell.
// Both are written on first differences D_i = v_{i+1} - v_i: second differences, the one-sided
// slopes and the three candidate values are short combinations of the D_i, and the result is
//   v_c + sum_k a_k q_k / (6 sum_k a_k)   with a_k = c_k / (eps + b_k)^2
// (identical to the reference's w0*p0 + w1*p1 + w2*p2 because the weights sum to one).
// Weights.  The reference's a_k = c_k / t_k^2 costs three quarter-rate reciprocals per state.  The FAST form puts the
// three weights over their common denominator, a_k = c_k (t_i t_j)^2 with {i, j} the other two stencils: three
// multiplies instead.  That needs t^4 (and t^4 times a first difference) inside fp32:
//  * t is carried scaled by WLAM = 2^-10 (folded into the constants: exact).  The all-smooth stencil then sits at
//    t = 1e-9, t^4 = 9e-37: still normal, and its products with the slopes underflow gradually (fp32 denormals are
//    on in this code object, float_denorm_mode_32 = 3) — worst-case absolute error of a state 2e-9;
//  * with every primitive of every cell at most W_FLIM in magnitude: |D| <= 2F, t <= 33.4 WLAM F^2 = 1.2e8, and
//    the numerator <= 8 F t^4 = 9e37 < FLT_MAX.
// The step kernel therefore tracks the largest |primitive| it WRITES (one more word next to the max wavespeed,
// all-reduced with it across slabs) — that is exactly the largest value the next step reads — and the next launch
// takes the FAST body only when that and the inflow state are within W_FLIM; otherwise the reciprocal form.  The
// choice is one scalar branch at the top of the kernel, identical for every workgroup and every slab of a step.
// (Mach-100 runs stay below |primitive| 700 until the reference scheme itself runs away: FAST throughout.)
// Round 5: the kernel pair's weno_cell carries t unscaled (one multiply less per measure; derivation at weno_cell), which
// closes its fast window at 2 760 — W_FLIM is the one window every kernel uses, so the fused kernel's scaled forms, good to
// 6e4, switch at 2.5e3 as well.  Mach-100 runs stay below 700 until the reference scheme itself runs away.
constexpr float WLAM = 0x1p-10f;
constexpr float W_FLIM = 2.5e3f;
// Both operands are tested on their own: fmaxf(NaN, x) returns x, so a NaN range (nothing known about the input) would
// otherwise select the fast form — the one case where t^4 may overflow.  A NaN comparison is false: reciprocal form.
__host__ __device__ __forceinline__ bool fast_form(float fmax_in, float in_fmax) { return (fmax_in <= W_FLIM) && (in_fmax <= W_FLIM); }
__device__ __forceinline__ float inv_sq(float t) { return rcp(t * t); }
// t_k = eps + 13/12 d^2 + 1/4 e^2
template <bool FAST> __device__ __forceinline__ float smooth_t(float sd, float e) {
  return ((FAST ? 0.25f * WLAM : 0.25f) * e) * e + sd;
}
template <bool FAST> __device__ __forceinline__ float sd_term(float d) {
  return ((FAST ? (13.f / 12.f) * WLAM : (13.f / 12.f)) * d) * d + (FAST ? WENO_EPS * WLAM : WENO_EPS);
}
template <bool FAST>
__device__ __forceinline__ void weno_weights(float t0, float t1, float t2, float &a0, float &a1, float &a2) {
  if (FAST) {
    const float u0 = t1 * t2, u1 = t0 * t2, u2 = t0 * t1;
    a0 = (0.1f * u0) * u0; a1 = (0.6f * u1) * u1; a2 = (0.3f * u2) * u2;
  } else {
    a0 = 0.1f * inv_sq(t0); a1 = 0.6f * inv_sq(t1); a2 = 0.3f * inv_sq(t2);
  }
}

// x faces of the own cells share weights across lanes (weno_face_xshare_r01 below, the fused kernel) and the split step's
// k_flux_xy takes the left state of a face from the lane below: a DPP wave shift (k_flux_xy with ds_bpermute shuffles instead:
// 4.04 against 3.85 ms — its LDS pipe is busy; the VALU-bound 2D marches with an idle LDS pipe prefer the shuffle)
__device__ __forceinline__ float lane_below(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}

// cell-centred: m2,m1,c,p1,p2 around a cell -> Lhi = left state at its high face, Rlo = right state at its low face.
// Round 5: the weighted sum is taken around the CENTRAL candidate.  With i_k = 1 / t_k^2 (up to a common factor) and the
// candidates q_k (in sixths, relative to the cell): q0 - q1 = 2 (E0 - E1), q2 - q1 = E1 - E2, so
//     Lhi = c + [ (2 D2 + D1) + (2 P + 3 Q) / (i0 + 6 i1 + 3 i2) ] / 6,      P = i0 (E0 - E1),  Q = i2 (E1 - E2)
//     Rlo = c - [ (2 D1 + D2) + (3 P + 2 Q) / (3 i0 + 6 i1 + i2) ] / 6
// — the linear weights 0.1 / 0.6 / 0.3 have become the integers of the two sums (6 is the only one that is not an inline
// constant), P and Q serve both states, and the smoothness measures carry the common factor 12/13:
//     t_k = E_k^2 + (3/13) e_k^2 + (12/13) eps        (two fma and one multiply; the round-2 form took four instructions)
// 44 VALU instructions + 2 reciprocals per cell and variable (round 4: 53 + 2; the reference: ~150 + 6 divisions).
// Fast form: i_k = (t_i t_j)^2 — t is NOT scaled any more (the scale cost a multiply per measure): with every primitive
// <= F, |D| <= 2F, |E| <= 4F, |e| <= 8F, t <= 30.8 F^2, i <= 9.0e5 F^8, |2P + 3Q| <= 5 * 8F * i = 3.6e7 F^9 < FLT_MAX for
// F < 2760 (W_FLIM = 2.5e3); the all-smooth stencil sits at t = 9.2e-7, i = 7e-25: no denormals anywhere.
constexpr float WENO_TE = (12.f / 13.f) * WENO_EPS, WENO_TC = 3.f / 13.f;
__device__ __forceinline__ float smooth_m(float E, float e) { return __builtin_fmaf(WENO_TC * e, e, __builtin_fmaf(E, E, WENO_TE)); }
template <bool FAST>
__device__ __forceinline__ void weno_inv(float t0, float t1, float t2, float &i0, float &i1, float &i2) {
  if (FAST) {
    const float u0 = t1 * t2, u1 = t0 * t2, u2 = t0 * t1;
    i0 = u0 * u0; i1 = u1 * u1; i2 = u2 * u2;
  } else {
    i0 = rcp(t0 * t0); i1 = rcp(t1 * t1); i2 = rcp(t2 * t2);
  }
}
template <bool FAST>
__device__ __forceinline__ void weno_cell(float m2, float m1, float c0, float p1, float p2, float &Lhi, float &Rlo) {
  const float D0 = m1 - m2, D1 = c0 - m1, D2 = p1 - c0, D3 = p2 - p1;
  const float E0 = D1 - D0, E1 = D2 - D1, E2 = D3 - D2;
  // one-sided slopes 3 D1 - D0 = 2 D1 + E0 and 3 D2 - D3 = 2 D2 - E2: the factor 2 is an inline constant, 3.0 is not (a VOP3
  // v_fma cannot carry a literal on gfx950: hipcc parks it in an SGPR, and an SGPR source halves the issue rate)
  const float eL = __builtin_fmaf(2.f, D1, E0), eR = __builtin_fmaf(2.f, D2, -E2);
  float i0, i1, i2;
  weno_inv<FAST>(smooth_m(E0, eL), smooth_m(E1, D1 + D2), smooth_m(E2, eR), i0, i1, i2);
  const float P = i0 * (E0 - E1), Q = i2 * (E1 - E2);
  const float S = P + Q;
  const float M = __builtin_fmaf(i1, 6.f, i0 + i2);
  const float sumL = __builtin_fmaf(2.f, i2, M), sumR = __builtin_fmaf(2.f, i0, M);
  const float numL = __builtin_fmaf(2.f, S, Q), numR = __builtin_fmaf(2.f, S, P);
  Lhi = __builtin_fmaf(__builtin_fmaf(numL, rcp(sumL), __builtin_fmaf(2.f, D2, D1)), 1.f / 6.f, c0);
  Rlo = __builtin_fmaf(__builtin_fmaf(numR, rcp(sumR), __builtin_fmaf(2.f, D1, D2)), -1.f / 6.f, c0);
}
// Lhi alone: the same arithmetic, instruction for instruction (the ring cells around a k_flux_xy tile, whose other state
// nobody reads).  Rlo of a cell is Lhi of the mirrored stencil — every difference changes sign exactly, the measures and
// the sums are symmetric — so the ring's high side calls this with its five cells in reverse order and the face it feeds
// gets bit for bit the state the neighbouring tile's own cell computes with weno_cell.
template <bool FAST>
__device__ __forceinline__ float weno_cell_hi(float m2, float m1, float c0, float p1, float p2) {
  const float D0 = m1 - m2, D1 = c0 - m1, D2 = p1 - c0, D3 = p2 - p1;
  const float E0 = D1 - D0, E1 = D2 - D1, E2 = D3 - D2;
  const float eL = __builtin_fmaf(2.f, D1, E0), eR = __builtin_fmaf(2.f, D2, -E2);
  float i0, i1, i2;
  weno_inv<FAST>(smooth_m(E0, eL), smooth_m(E1, D1 + D2), smooth_m(E2, eR), i0, i1, i2);
  const float P = i0 * (E0 - E1), Q = i2 * (E1 - E2);
  const float S = P + Q;
  const float M = __builtin_fmaf(i1, 6.f, i0 + i2);
  const float sumL = __builtin_fmaf(2.f, i2, M);
  const float numL = __builtin_fmaf(2.f, S, Q);
  return __builtin_fmaf(__builtin_fmaf(numL, rcp(sumL), __builtin_fmaf(2.f, D2, D1)), 1.f / 6.f, c0);
}

// ---- the r01 forms of the three reconstructions (3.0 / 5.0 factors, SGPR-resident): the fused k_step keeps them — with
// the inline-constant forms it needs 174 VGPRs instead of 168 and drops from three waves per SIMD to two
template <bool FAST>
__device__ __forceinline__ void weno_face_r01(float v0, float v1, float v2, float v3, float v4, float v5, float &L,
                                          float &R) {
  const float D0 = v1 - v0, D1 = v2 - v1, D2 = v3 - v2, D3 = v4 - v3, D4 = v5 - v4;
  const float sA = sd_term<FAST>(D1 - D0), sB = sd_term<FAST>(D2 - D1), sC = sd_term<FAST>(D3 - D2),
              sD = sd_term<FAST>(D4 - D3);
  float sumL, sumR;
  // left state (centre cell v2): stencils {0,1,2} {1,2,3} {2,3,4}
  {
    float a0, a1, a2;
    weno_weights<FAST>(smooth_t<FAST>(sA, 3.f * D1 - D0), smooth_t<FAST>(sB, D1 + D2),
                       smooth_t<FAST>(sC, 3.f * D2 - D3), a0, a1, a2);
    float num = a0 * (5.f * D1 - 2.f * D0) + a1 * (2.f * D2 + D1) + a2 * (4.f * D2 - D3);
    sumL = a0 + a1 + a2;
    L = v2 + num * (rcp(sumL) * (1.f / 6.f));
  }
  // right state (centre cell v3): the mirror image, stencils {5,4,3} {4,3,2} {3,2,1}
  {
    float a0, a1, a2;
    weno_weights<FAST>(smooth_t<FAST>(sD, 3.f * D3 - D4), smooth_t<FAST>(sC, D3 + D2),
                       smooth_t<FAST>(sB, 3.f * D2 - D1), a0, a1, a2);
    float num = a0 * (2.f * D4 - 5.f * D3) - a1 * (2.f * D2 + D3) + a2 * (D1 - 4.f * D2);
    sumR = a0 + a1 + a2;
    R = v3 + num * (rcp(sumR) * (1.f / 6.f));
  }
}

template <bool FAST>
__device__ __forceinline__ void weno_face_xshare_r01(float v0, float v1, float v2, float v3, float v4, float v5, float &L,
                                                 float &R) {
  const float D0 = v1 - v0, D1 = v2 - v1, D2 = v3 - v2, D3 = v4 - v3, D4 = v5 - v4;
  const float sB = sd_term<FAST>(D2 - D1), sC = sd_term<FAST>(D3 - D2), sD = sd_term<FAST>(D4 - D3);
  // own cell (v3): stencils {5,4,3} {4,3,2} {3,2,1}
  const float t0 = smooth_t<FAST>(sD, 3.f * D3 - D4), t1 = smooth_t<FAST>(sC, D3 + D2), t2 = smooth_t<FAST>(sB, 3.f * D2 - D1);
  float w0, w1, w2;
  if (FAST) {
    const float u0 = t1 * t2, u1 = t0 * t2, u2 = t0 * t1;
    w0 = u0 * u0; w1 = u1 * u1; w2 = u2 * u2;
  } else {
    w0 = inv_sq(t0); w1 = inv_sq(t1); w2 = inv_sq(t2);
  }
  {
    const float a0 = 0.1f * w0, a1 = 0.6f * w1, a2 = 0.3f * w2;
    const float num = a0 * (2.f * D4 - 5.f * D3) - a1 * (2.f * D2 + D3) + a2 * (D1 - 4.f * D2);
    R = v3 + num * (rcp(a0 + a1 + a2) * (1.f / 6.f));
  }
  { // cell v2 = the own cell of the lane below: its stencils {0,1,2} {1,2,3} {2,3,4} are that lane's {3,2,1} {4,3,2} {5,4,3}
    const float a0 = 0.1f * lane_below(w2), a1 = 0.6f * lane_below(w1), a2 = 0.3f * lane_below(w0);
    const float num = a0 * (5.f * D1 - 2.f * D0) + a1 * (2.f * D2 + D1) + a2 * (4.f * D2 - D3);
    L = v2 + num * (rcp(a0 + a1 + a2) * (1.f / 6.f));
  }
}

template <bool FAST>
__device__ __forceinline__ void weno_cell_r01(float m2, float m1, float c0, float p1, float p2, float &Lhi, float &Rlo) {
  const float D0 = m1 - m2, D1 = c0 - m1, D2 = p1 - c0, D3 = p2 - p1;
  float w0, w1, w2; // 0.1 i0, 0.6 i1, 0.3 i2 with i_k = 1 / t_k^2 up to a common factor
  weno_weights<FAST>(smooth_t<FAST>(sd_term<FAST>(D1 - D0), 3.f * D1 - D0), smooth_t<FAST>(sd_term<FAST>(D2 - D1), D1 + D2),
                     smooth_t<FAST>(sd_term<FAST>(D3 - D2), 3.f * D2 - D3), w0, w1, w2);
  float sumL, sumR;
  {
    float num = w0 * (5.f * D1 - 2.f * D0) + w1 * (2.f * D2 + D1) + w2 * (4.f * D2 - D3);
    sumL = w0 + w1 + w2;
    Lhi = c0 + num * (rcp(sumL) * (1.f / 6.f));
  }
  { // mirrored roles: a0 = 0.1 i2, a1 = 0.6 i1, a2 = 0.3 i0
    float a0 = (1.f / 3.f) * w2, a2 = 3.f * w0;
    float num = a0 * (2.f * D3 - 5.f * D2) - w1 * (2.f * D1 + D2) + a2 * (D0 - 4.f * D1);
    sumR = a0 + w1 + a2;
    Rlo = c0 + num * (rcp(sumR) * (1.f / 6.f));
  }
}

// one edge state of a cell only (HI: left state at its high face, else right state at its low face): bit for bit what
// weno_cell returns for that side
template <bool FAST, bool HI>
__device__ __forceinline__ float weno_cell_side(float m2, float m1, float c0, float p1, float p2) {
  return HI ? weno_cell_hi<FAST>(m2, m1, c0, p1, p2) : weno_cell_hi<FAST>(p2, p1, c0, m1, m2);
}

__device__ __forceinline__ void prim_floor(Prim &q) { // :565-571
  q.q[IR] = fmaxf(q.q[IR], RHO_P_FLOOR);
  q.q[IP] = fmaxf(q.q[IP], RHO_P_FLOOR);
  q.q[IE] = fmaxf(q.q[IE], 0.f);
}

// ---------------------------------------------------------------- HLLC blended with HLL, :383-460
// axis may be a literal (specialised after inlining) or a lane-varying value in {0,1}.
// The three gas constants hllc needs.  From kernel arguments they are SGPRs, and a VALU instruction with an SGPR source
// issues at half rate on gfx950 (profiles/r02/valu_calib.txt); kernels with registers to spare copy them to VGPRs once.
struct Gas { float gamma, gm1, inv_gm1; };
__device__ __forceinline__ float vreg(float s) {
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
__device__ __forceinline__ Gas gas_sgpr(const Args &A) { return Gas{A.gamma, A.gm1, A.inv_gm1}; }
__device__ __forceinline__ Gas gas_vgpr(const Args &A) { return Gas{vreg(A.gamma), vreg(A.gm1), vreg(A.inv_gm1)}; }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ Cons hllc(const Gas &A, const Prim &L, const Prim &R, int axis) {
  const float rL = L.q[IR], rR = R.q[IR], pL = L.q[IP], pR = R.q[IP];
  const float irL = rcp(rL), irR = rcp(rR);
  // a = sqrt(max(gamma p / r, eps)) and 1 / a from ONE transcendental: a = x rsq(x).  x >= 1e-12, so a >= 1e-6 and the
  // reference's guards on aRef (max(aRef, 1e-12), 0.1 aRef > 1e-12; :366-374) never bind: 1 / aRef = min(1 / aL, 1 / aR).
  const float xL = fmaxf(A.gamma * pL * irL, DENOM_EPS), xR = fmaxf(A.gamma * pR * irR, DENOM_EPS);
  const float rsL = rsq(xL), rsR = rsq(xR);
  const float aL = xL * rsL, aR = xR * rsR;
  const float unL = (axis == 0) ? L.q[IU] : (axis == 1) ? L.q[IV] : L.q[IW];
  const float unR = (axis == 0) ? R.q[IU] : (axis == 1) ? R.q[IV] : R.q[IW];
  float sL = fminf(unL - aL, unR - aR);
  float sR = fmaxf(unL + aL, unR + aR);
  const float aRef = fmaxf(aL, aR);
  const float iaRef = fminf(rsL, rsR);
  { // entropy_fix_speed, :366-374  (1/max(0.1 a, eps) == 10/a)
    const float d = 0.1f * aRef;
    float asl = fabsf(sL), asr = fabsf(sR);
    // the fix touches a signal speed within a tenth of the sound speed of zero — the sonic lines; a wave none of whose 64 faces
    // is there skips its arithmetic and selects (round 5; same values: the selects below keep sL / sR wherever the test fails)
    if (__builtin_amdgcn_ballot_w64(!(asl >= d) || !(asr >= d)) != 0ull)
    {
      const float id = 10.f * iaRef;
      float fl = 0.5f * (asl * asl * id + d), fr = 0.5f * (asr * asr * id + d);
      sL = (asl >= d) ? sL : ((sL >= 0.f) ? fl : -fl);
      sR = (asr >= d) ? sR : ((sR >= 0.f) ? fr : -fr);
    }
  }
  // ---- Round 6: the flux as ONE linear combination of the two STATES.  The reference (:383-460) forms U_L, U_R, F_L, F_R, the
  // star state and the HLL average, then blends; every one of those is linear in the two conserved states, so with
  //     F_K = un_K U_K + p_K (e_n + un_K e_E),      U* - U_K = g U_K  (+ the two components below),      g = (s_M - un_K) / (s_K - s_M)
  // the blend  (1 - alpha) [F_K + s_K (U* - U_K)] + alpha [s_R F_L - s_L F_R + s_L s_R (U_R - U_L)] / (s_R - s_L)  is
  //     F = B_L (1, u, v, w, h, ev)_L + B_R (1, u, v, w, h, ev)_R + pressure terms + the two star corrections,
  //     B_L = r_L (c_FL un_L - c_U + [K = L] G),   B_R = r_R (c_FR un_R + c_U + [K = R] G),   G = (1 - alpha) s_K g
  // — a multiply and an fma per component where round 5 spent seven (F_L, F_R, U_R - U_L, four fma), no conserved vectors, no
  // physical fluxes, and the supersonic exits become a choice of COEFFICIENTS (alpha = 0, K = the upwind side, G = 0: then
  // c_FK = 1, everything else 0, and the sum is un_K U_K + p_K terms: F_K to rounding) instead of twelve selects of components.
  // Star corrections, the two components of U* - U_K that are not g U_K:
  //     normal momentum  (rho un)* - (rho un)_K = s_K (rho* - rho_K) = g (rho un)_K + g r_K (s_K - un_K)
  //     energy           E* - E_K = ((s_M - un_K) E_K - p_K un_K + p* s_M) / (s_K - s_M)   — carried as the difference the
  //                      reference's own formulation resolves, fl(E_K + d) - E_K (round 5's finding, kept: between two floored
  //                      densities s_M is noise and must vanish in E_K's rounding, likewise g in fl(1 + g) - 1)
  // Explicit fma chains throughout: the inlined copies of this function (chunk prologue / marching loop, slab / single domain)
  // must round alike.
  const float keL = 0.5f * (L.q[IU] * L.q[IU] + L.q[IV] * L.q[IV] + L.q[IW] * L.q[IW]);
  const float keR = 0.5f * (R.q[IU] * R.q[IU] + R.q[IV] * R.q[IV] + R.q[IW] * R.q[IW]);
  // e_th = p / max((gamma-1) r, floor) = p min(1 / ((gamma-1) r), 1 / floor): the floor only bites below r = 1e-29.  One v_min
  // where a compare and a select stood (both half rate on gfx950, like the v_min: profiles/r06/valu_calib_r06.txt)
  // (both products formed, the smaller taken: the same bits as  g >= floor ? p / r / (gamma-1) : p / floor  of rounds 2-5)
  const float ethL = fminf(pL * irL * A.inv_gm1, pL * (1.f / RHO_P_FLOOR));
  const float ethR = fminf(pR * irR * A.inv_gm1, pR * (1.f / RHO_P_FLOOR));
  const float hL = (keL + ethL) + L.q[IE];   // E / r, :234-245
  const float EL = rL * hL;
  // Every face of the wave supersonic to the right (:431 s_L >= 0 -> F_L) — the x faces of the free stream and of most of the
  // shock layer at Mach 100: F_L alone, before anything of the right state's energy or of the star region exists.  (Round 5 got
  // this skip from the compiler, which turned the early return into a divergent branch; the coefficient form below has no
  // branch to skip, so the wave-uniform exit is spelled out.  Not on the z axis: the marching kernel's flow is never supersonic
  // in z, and the second exit costs its allocation registers.)
  // The exit returns, bit for bit, what the coefficient form below gives a supersonic face (c_FL = 1, everything else 0: B_L = r_L
  // un_L, B_R = 0) — a face's flux must not depend on which other faces share its wave, or the result would depend on the tiling:
  //     normal momentum  fl(fl(un_L B_L) + p_L)   (two roundings, as the fma chain below leaves it),    energy  fl(un_L p_L + fl(h_L B_L))
  if (axis != 2 && __builtin_amdgcn_ballot_w64(!(sL >= 0.f)) == 0ull) {
    const float rm = rL * unL;
    Cons F;
    F.c[0] = rm;
#pragma unroll
    for (int k = 1; k <= 3; k++) F.c[k] = (axis == k - 1) ? __builtin_fmaf(1.f, pL, unL * rm) : L.q[k] * rm;
    F.c[4] = __builtin_fmaf(unL, pL, hL * rm);
    F.c[5] = L.q[IE] * rm;
    return F;
  }
  const float hR = (keR + ethR) + R.q[IE];
  const float ER = rR * hR;

  const float dsL = sL - unL, dsR = sR - unR;
  const float mdL = rL * dsL, mdR = rR * dsR;          // r_K (s_K - un_K)
  const float denom = denom_guard(mdL - mdR);
  const float sM = __builtin_fmaf(-mdR, unR, __builtin_fmaf(mdL, unL, pR - pL)) * rcp(denom);
  const float pStarL = __builtin_fmaf(mdL, sM - unL, pL);
  const float pStarR = __builtin_fmaf(mdR, sM - unR, pR);
  const float pStar = 0.5f * (pStarL + pStarR);

  float vc; // axis_crossflow_speed, :318-325
  if (axis == 0) vc = (fabsf(L.q[IV]) + fabsf(R.q[IV]) + fabsf(L.q[IW]) + fabsf(R.q[IW])) * 0.5f;
  else if (axis == 1) vc = (fabsf(L.q[IU]) + fabsf(R.q[IU]) + fabsf(L.q[IW]) + fabsf(R.q[IW])) * 0.5f;
  else vc = (fabsf(L.q[IU]) + fabsf(R.q[IU]) + fabsf(L.q[IV]) + fabsf(R.q[IV])) * 0.5f;
  const float align = clampf(1.f - vc * iaRef, 0.f, 1.f);
  float alpha;
  { // shock_sensor, :376-381 — both ratios over one reciprocal
    const float sp = fmaxf(pR + pL, DENOM_EPS), sr = fmaxf(rR + rL, DENOM_EPS);
    const float inv = rcp(sp * sr);
    const float dp = fabsf(pR - pL) * sr * inv;
    const float dr = fabsf(rR - rL) * sp * inv;
    alpha = clampf(5.f * (0.5f * (dp + dr)), 0.f, 1.f) * align;
  }
  // s_R - s_L >= 0 always (s_R is a max over un + a, s_L a min over un - a, rounding is monotonic, and the entropy fix keeps the
  // signs and moves magnitudes up): the sign-preserving guard (:147-150) is a plain floor here
  const float ihll = rcp(fmaxf(sR - sL, DENOM_EPS));

  // which side: the supersonic exits (:431-434: s_L >= 0 -> F_L, else s_R <= 0 -> F_R) are the star branch's own choice of K
  // with alpha = 0 and no star correction
  const bool supL = (sL >= 0.f), supR = (sR <= 0.f);
  const bool sup = supL || supR;
  const bool left = supL || (!supR && (sM >= 0.f));
  alpha = sup ? 0.f : alpha;
  const float sK = left ? sL : sR, unK = left ? unL : unR, pK = left ? pL : pR, EK = left ? EL : ER, mdK = left ? mdL : mdR;
  const float iden = rcp(denom_guard(sK - sM));
  const float dM = sM - unK;
  const float g = (__builtin_fmaf(dM, iden, 1.f)) - 1.f;
  const float X = __builtin_fmaf(pStar, sM, __builtin_fmaf(dM, EK, -(pK * unK)));
  const float dEq = __builtin_fmaf(X, iden, EK) - EK;

  const float wC = 1.f - alpha, wH = alpha * ihll;
  const float aS = wC * sK;
  const float G = sup ? 0.f : aS * g;            // (a select, not a product with zero: g of a face nobody asked for may be anything)
  const float aSdE = sup ? 0.f : aS * dEq;
  const float cU = wH * (sL * sR);
  const float lw = left ? wC : 0.f;
  const float cFL = __builtin_fmaf(wH, sR, lw), cFR = __builtin_fmaf(-wH, sL, wC - lw);
  const float fpL = cFL * unL, fpR = cFR * unR;  // what multiplies p_L, p_R in the energy flux
  const float GL = left ? G : 0.f, GR = G - GL;
  const float BL0 = rL * (fpL - cU), BR0 = rR * (fpR + cU);
  const float BL = __builtin_fmaf(rL, GL, BL0), BR = __builtin_fmaf(rR, GR, BR0);
  Cons F;
  F.c[0] = BL + BR;
  // normal momentum: un_L B_L + un_R B_R + G r_K (s_K - un_K) + c_FL p_L + c_FR p_R
  const float Fn = __builtin_fmaf(cFR, pR, __builtin_fmaf(cFL, pL, __builtin_fmaf(G, mdK, __builtin_fmaf(unR, BR, unL * BL))));
#pragma unroll
  for (int k = 1; k <= 3; k++) {
    const float Ft = __builtin_fmaf(R.q[k], BR, L.q[k] * BL);
    F.c[k] = (axis == k - 1) ? Fn : Ft;
  }
  F.c[4] = aSdE + __builtin_fmaf(fpR, pR, __builtin_fmaf(fpL, pL, __builtin_fmaf(hR, BR0, hL * BL0)));
  F.c[5] = __builtin_fmaf(R.q[IE], BR, L.q[IE] * BL);
  return F;
}
__device__ __forceinline__ Cons hllc(const Args &A, const Prim &L, const Prim &R, int axis) { return hllc(gas_sgpr(A), L, R, axis); }

// Solid handling of a face between cells `lo` | `hi` (branch structure of :1125-1143): s = solid
// bits of the six cells around the face (bit 2 = lo, bit 3 = hi).  Any solid in the stencil ->
// first order (L = lo, R = hi); one side solid -> the fluid side mirrored across the wall.
__device__ __forceinline__ void solid_override(Prim &L, Prim &R, const float (&lo)[6], const float (&hi)[6],
                                               unsigned s, int axis) {
  if (s != 0u) {
    const bool s2 = (s >> 2) & 1u, s3 = (s >> 3) & 1u;
#pragma unroll
    for (int m = 0; m < 6; m++) { L.q[m] = lo[m]; R.q[m] = hi[m]; }
    const int un = (axis == 0) ? IU : (axis == 1) ? IV : IW;
    if (s2 && !s3) { // low side solid: L = mirror(R), :772-781
#pragma unroll
      for (int m = 0; m < 6; m++) L.q[m] = (m == un) ? -R.q[m] : R.q[m];
    } else if (s3 && !s2) { // high side solid: R = mirror(L)
#pragma unroll
      for (int m = 0; m < 6; m++) R.q[m] = (m == un) ? -L.q[m] : L.q[m];
    }
  }
}
// same with a lane-varying axis in {0, 1}
__device__ __forceinline__ void solid_override_xy(Prim &L, Prim &R, const float (&lo)[6], const float (&hi)[6],
                                                  unsigned s, bool isx) {
  if (s != 0u) {
    const bool s2 = (s >> 2) & 1u, s3 = (s >> 3) & 1u;
#pragma unroll
    for (int m = 0; m < 6; m++) { L.q[m] = lo[m]; R.q[m] = hi[m]; }
    if (s2 && !s3) {
#pragma unroll
      for (int m = 0; m < 6; m++) L.q[m] = R.q[m];
      if (isx) L.q[IU] = -L.q[IU]; else L.q[IV] = -L.q[IV];
    } else if (s3 && !s2) {
#pragma unroll
      for (int m = 0; m < 6; m++) R.q[m] = L.q[m];
      if (isx) R.q[IU] = -R.q[IU]; else R.q[IV] = -R.q[IV];
    }
  }
}

// face between line cells v[2] | v[3]; six cells from LDS
template <bool FAST, bool XSHARE = false>
__device__ __forceinline__ Cons face_flux6(const Args &A, const float (&v)[6][6], unsigned s, int axis) {
  Prim L, R;
#pragma unroll
  for (int m = 0; m < 6; m++) {
    if (XSHARE) weno_face_xshare_r01<FAST>(v[0][m], v[1][m], v[2][m], v[3][m], v[4][m], v[5][m], L.q[m], R.q[m]);
    else weno_face_r01<FAST>(v[0][m], v[1][m], v[2][m], v[3][m], v[4][m], v[5][m], L.q[m], R.q[m]);
  }
  solid_override(L, R, v[2], v[3], s, axis);
  prim_floor(L);
  prim_floor(R);
  return hllc(A, L, R, axis);
}

// one cell of the grid as the kernel sees it: inflow ghost left of x = 0, transmissive ghost right
// of x = nx-1 (built from the last interior cell of the same row), y periodic; :1019-1056
__device__ __forceinline__ void fetch_cell(const Args &A, int gx, int gyw, int zh, int zg, float (&q)[6], bool &sol) {
  Prim p;
  if (gx < 0) {
    p = inflow_prim(A);
    sol = sdf_solid(A, gx, gyw, zg);
  } else if (gx >= A.nx) {
    size_t gi = ((size_t)zh * A.ny + gyw) * A.nx + (A.nx - 1);
    p = outflow_prim(A, decode(A, gi));
    sol = sdf_solid(A, gx, gyw, zg);
  } else {
    size_t gi = ((size_t)zh * A.ny + gyw) * A.nx + gx;
    p = decode(A, gi);
    sol = A.solid[gi] != 0;
  }
#pragma unroll
  for (int m = 0; m < 6; m++) q[m] = p.q[m];
}

// one field of decode(): the primitive the encoded value e of field m stands for
__device__ __forceinline__ float decode_field(float u_ref, int m, float e) {
  return (m >= 1 && m <= 3) ? u_ref * fsinh(e) : fexp(e);
}
// decode_field for the split step's kernels: fsinh_wave (bit-identical values)
__device__ __forceinline__ float decode_field_w(float u_ref, int m, float e) {
  return (m >= 1 && m <= 3) ? u_ref * fsinh_wave(e) : fexp(e);
}
#define ZDEC(u, m, e) decode_field_w(u, m, e)
using tau::GChar; using tau::GFloat; using tau::gld; using tau::gst; using tau::lane_off;   // tau_common.h: scalar base + 32-bit lane offset

// fetch_cell through a scalar plane base.  epl: field 0 of the encoded state at plane zh; fs4: bytes between fields; spl:
// the solid mask at plane zh
// test: eq = the six ENCODED values are eqref[0..5] bit for bit (an interior cell; ghost columns never are)
__device__ __forceinline__ void fetch_cell_e(const Args &A, float uref, const GChar *qpl, size_t fs4, const uint8_t *spl, int gx, int gyw,
                                             int zg, float (&q)[6], bool &sol, bool test, const float (&eqref)[6], bool strips,
                                             const float (&refE)[6], const float (&refN)[6], unsigned &eq) {
  Prim p;
  eq = 0u;   // bit 0: the cell is eqref, bit 1: it is refE, bit 2: refN (strips only)
  if (gx < 0) {
    p = inflow_prim(A);
    sol = sdf_solid(A, gx, gyw, zg);
  } else if (gx >= A.nx) {
    const unsigned vo = lane_off((unsigned)(gyw * A.nx + (A.nx - 1)) << 2);
    float e[6];
#pragma unroll
    for (int m = 0; m < 6; m++) e[m] = gld(qpl + m * fs4, vo);
#pragma unroll
    for (int m = 0; m < 6; m++) p.q[m] = decode_field_w(uref, m, e[m]);
    p = outflow_prim(A, p);
    sol = sdf_solid(A, gx, gyw, zg);
  } else {
    const unsigned vo = lane_off((unsigned)(gyw * A.nx + gx) << 2);
    float e[6];   // all six loads in flight before the first decode (fsinh_wave's branches pin the arithmetic behind its load)
#pragma unroll
    for (int m = 0; m < 6; m++) e[m] = gld(qpl + m * fs4, vo);
    sol = spl[vo >> 2] != 0;
    // BIT patterns (round 6): +0 and -0 compare equal as numbers, and a tile at the edge of the disturbance can hold both in v or w.
    // The exit itself survives that (a flagged tile's +0 divergence absorbs the sign), the predictions built on the flags do not:
    // k_update_z stores one new state for every cell of a run of predicted planes.  (Also cheaper: xor / or at full rate, one compare.)
    if (test) {
      auto x = [&](const float (&r)[6], int m) { return __float_as_uint(e[m]) ^ __float_as_uint(r[m]); };
      auto same = [&](const float (&r)[6]) { return ((x(r, 0) | x(r, 1) | x(r, 2)) | (x(r, 3) | x(r, 4) | x(r, 5))) == 0u; };
      eq = same(eqref) ? 1u : 0u;
      if (strips) eq |= (same(refE) ? 2u : 0u) | (same(refN) ? 4u : 0u);
    }
#pragma unroll
    for (int m = 0; m < 6; m++) p.q[m] = decode_field_w(uref, m, e[m]);
  }
#pragma unroll
  for (int m = 0; m < 6; m++) q[m] = p.q[m];
}

// ---------------------------------------------------------------- the step kernel
struct StepLds {
  float sP[6][PLANE];            // current plane, primitives, x/y halo 3
  uint8_t sS[PLANE];             // solid flags of the same cells
  float sFx[6][TY][TX + 1];      // low-x face flux of cell (y, x); column TX = far edge
  float sFy[6][TY + 1][TX];      // low-y face flux of cell (y, x); row TY = far edge
  float sRed[2][NT / 64];
};
// `b`: linear work item (tile x, tile y, z chunk) of this pass; dt, gain: the step's clock; maxw: the step's two max words
template <bool FAST> __device__ __forceinline__ void step_body(const Args &A, StepLds &S, unsigned b, const float dt, const float gain,
                                                               unsigned *maxw) {
  auto &sP = S.sP; auto &sS = S.sS; auto &sFx = S.sFx; auto &sFy = S.sFy; auto &sRed = S.sRed;

  const int tid = threadIdx.x;
  const int tx = tid & (TX - 1), ty = tid >> 5;
  const int lane = tid & 63, wave = tid >> 6;

  const int bx = (int)(b % (unsigned)A.ntx); b /= (unsigned)A.ntx;
  const int by = (int)(b % (unsigned)A.nty);
  const int bz = (int)(b / (unsigned)A.nty);
  const int bx0 = bx * TX, by0 = by * TY;
  const bool second = bz >= A.nzc1;                    // wave-uniform: two disjoint plane ranges can share a launch
  const int zc_lo = second ? A.zl_lo2 + (bz - A.nzc1) * A.zchunk : A.zl_lo + bz * A.zchunk;
  const int zc_hi = min(zc_lo + A.zchunk, second ? A.zl_hi2 : A.zl_hi);

  const int x = bx0 + tx, y = by0 + ty;
  const bool in_xy = (x < A.nx) && (y < A.ny);
  const int yw = (y >= A.ny) ? y - A.ny : y;   // partial tiles: own column is a wrapped / ghost column
  const size_t plane_n = (size_t)A.nx * A.ny;
  const size_t col = (size_t)yw * A.nx + min(x, A.nx - 1);

  // own column: planes z-1 .. z+3 around the current plane z (W[1] is the cell being updated)
  float W[5][6];
  unsigned ws = 0; // solid bits of planes z-2 .. z+3 (bit 2 = plane z, bit 3 = plane z+1)
  auto load_own = [&](int zl, float (&dst)[6]) -> unsigned {
    bool sol;
    fetch_cell(A, x, yw, zl + HALO, wrapi(A.z0 + zl, A.nz), dst, sol);
    return sol ? 1u : 0u;
  };

  // prologue: cells zc_lo-3 .. zc_lo+2 -> flux through the low-z face of plane zc_lo, and the
  // left state cell zc_lo contributes to its high face
  float Fz_lo[6];
  float Lz[6];          // carried: WENO left state at the face above the current cell
  {
    float T[6];         // plane zc_lo-3 (only needed here)
    ws |= load_own(zc_lo - 3, T) << 0;
#pragma unroll
    for (int k = 0; k < 5; k++) ws |= load_own(zc_lo - 2 + k, W[k]) << (k + 1);
    // W[0..4] = zc_lo-2 .. zc_lo+2 ; face zc_lo-1/2 lies between W[1] and W[2]
    Prim L, R;
#pragma unroll
    for (int m = 0; m < 6; m++) {
      float Rdummy;
      weno_cell_r01<FAST>(T[m], W[0][m], W[1][m], W[2][m], W[3][m], L.q[m], Rdummy);      // cell zc_lo-1 -> L at zc_lo-1/2
      weno_cell_r01<FAST>(W[0][m], W[1][m], W[2][m], W[3][m], W[4][m], Lz[m], R.q[m]);     // cell zc_lo -> R there, L above
    }
    solid_override(L, R, W[1], W[2], ws, 2);
    prim_floor(L);
    prim_floor(R);
    Cons F = hllc(A, L, R, 2);
#pragma unroll
    for (int m = 0; m < 6; m++) Fz_lo[m] = F.c[m];
    // shift so that W[0..3] = zc_lo-1 .. zc_lo+2 and the loop's first slide makes W = z-1 .. z+3
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int m = 0; m < 6; m++) W[k][m] = W[k + 1][m];
  }

  float smax = 0.f, fmx = 0.f;

  for (int z = zc_lo; z < zc_hi; z++) {
    // ---- bring plane z+3 into the window: W[0..4] = z-1 .. z+3
    ws = (ws >> 1) | (load_own(z + 3, W[4]) << 5);

    // ---- stage plane z into LDS: own cell from the window, the 240 halo cells decoded here
    {
      const int zh = z + HALO;
      const int zg = wrapi(A.z0 + z, A.nz);
      const int lc0 = (ty + HALO) * PXS + (tx + HALO);
#pragma unroll
      for (int m = 0; m < 6; m++) sP[m][lc0] = W[1][m];
      sS[lc0] = (uint8_t)((ws >> 2) & 1u);
      // halo of the plane: 3 rows above and below the tile (own columns only) + 3 cells left and right of every
      // own row.  The four 3x3 corners are never read — every stencil is axis-aligned — and leaving them out
      // brings the count to 240 <= 256: ONE decode round per plane instead of a second round for 20 lanes.
      constexpr int NROWS = 2 * HALO * TX;            // 192
      constexpr int NHALO = NROWS + TY * 2 * HALO;    // 240
      for (int p = tid; p < NHALO; p += NT) {
        int ly, lx;
        if (p < NROWS) {
          const int r = p / TX;
          ly = (r < HALO) ? r : r + TY;
          lx = HALO + (p - r * TX);
        } else {
          const int q = p - NROWS;
          const int r = q / (2 * HALO), c = q - r * (2 * HALO);
          ly = HALO + r;
          lx = (c < HALO) ? c : c + TX;
        }
        const int gx = bx0 + lx - HALO;
        const int gy = wrapi(by0 + ly - HALO, A.ny);
        float q[6];
        bool sol;
        fetch_cell(A, gx, gy, zh, zg, q, sol);
        const int li = ly * PXS + lx;
#pragma unroll
        for (int m = 0; m < 6; m++) sP[m][li] = q[m];
        sS[li] = sol ? 1 : 0;
      }
    }
    __syncthreads();

    // ---- face fluxes: low-x, low-y of the own cell; high-z of the own cell
    const int lc = (ty + HALO) * PXS + (tx + HALO);
    {
      float v[6][6];
      unsigned s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) {
#pragma unroll
        for (int m = 0; m < 6; m++) v[k][m] = sP[m][lc + (k - 3)];
        s |= (unsigned)sS[lc + (k - 3)] << k;
      }
      Cons F = face_flux6<FAST, true>(A, v, s, 0);
      if (tx != 0) {      // the left state of column 0 came from another row: that face belongs to the edge round
#pragma unroll
        for (int m = 0; m < 6; m++) sFx[m][ty][tx] = F.c[m];
      }
    }
    {
      float v[6][6];
      unsigned s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) {
#pragma unroll
        for (int m = 0; m < 6; m++) v[k][m] = sP[m][lc + (k - 3) * PXS];
        s |= (unsigned)sS[lc + (k - 3) * PXS] << k;
      }
      Cons F = face_flux6<FAST>(A, v, s, 1);
#pragma unroll
      for (int m = 0; m < 6; m++) sFy[m][ty][tx] = F.c[m];
    }
    // edge faces of the tile: 8 x-faces at column TX, 32 y-faces at row TY, 8 x-faces at column 0 — one extra
    // round of one wave, axis is lane-varying
    if (wave == (int)((unsigned)z % (unsigned)NW) && lane < 2 * TY + TX) { // the wave that takes the extra round rotates with z: SIMD balance
      const bool isx = lane < TY || lane >= TY + TX;
      const int ey = lane < TY ? lane : (lane < TY + TX ? TY : lane - (TY + TX));
      const int ex = lane < TY ? TX : (lane < TY + TX ? lane - TY : 0);
      const int st = isx ? 1 : PXS;
      const int c0 = (ey + HALO) * PXS + (ex + HALO);
      float v[6][6];
      unsigned s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) {
#pragma unroll
        for (int m = 0; m < 6; m++) v[k][m] = sP[m][c0 + (k - 3) * st];
        s |= (unsigned)sS[c0 + (k - 3) * st] << k;
      }
      Prim L, R;
#pragma unroll
      for (int m = 0; m < 6; m++) weno_face_r01<FAST>(v[0][m], v[1][m], v[2][m], v[3][m], v[4][m], v[5][m], L.q[m], R.q[m]);
      solid_override_xy(L, R, v[2], v[3], s, isx);
      prim_floor(L);
      prim_floor(R);
      Cons F = hllc(A, L, R, isx ? 0 : 1);
      if (isx) {
#pragma unroll
        for (int m = 0; m < 6; m++) sFx[m][ey][ex] = F.c[m];
      } else {
#pragma unroll
        for (int m = 0; m < 6; m++) sFy[m][TY][ex] = F.c[m];
      }
    }
    float Fz_hi[6];
    {
      // cell z+1 (W[2]) from its five cells W[0..4]: right state at face z+1/2, left state at z+3/2
      Prim L, R;
      float Lnext[6];
#pragma unroll
      for (int m = 0; m < 6; m++) L.q[m] = Lz[m];
#pragma unroll
      for (int m = 0; m < 6; m++) weno_cell_r01<FAST>(W[0][m], W[1][m], W[2][m], W[3][m], W[4][m], Lnext[m], R.q[m]);
      solid_override(L, R, W[1], W[2], ws, 2);
      prim_floor(L);
      prim_floor(R);
      Cons F = hllc(A, L, R, 2);
#pragma unroll
      for (int m = 0; m < 6; m++) { Fz_hi[m] = F.c[m]; Lz[m] = Lnext[m]; }
    }
    __syncthreads();

    // ---- conservative update of cell (x, y, z), :1266-1358
    const bool own_solid = (ws >> 2) & 1u;
    if (in_xy) {
      const size_t gi = (size_t)(z + HALO) * plane_n + col;
      if (own_solid) { // :1063-1072 copy-through
#pragma unroll
        for (int m = 0; m < 6; m++) A.out[m][gi] = A.in[m][gi];
      } else {
        const float r0 = W[1][IR], u0 = W[1][IU], v0 = W[1][IV], w0 = W[1][IW], p0 = W[1][IP], e0 = W[1][IE];
        float U0[6];
        U0[0] = r0; U0[1] = r0 * u0; U0[2] = r0 * v0; U0[3] = r0 * w0;
        {
          float ke = 0.5f * (u0 * u0 + v0 * v0 + w0 * w0);
          float eth = p0 * rcp(fmaxf(A.gm1 * r0, RHO_P_FLOOR));
          U0[4] = r0 * (ke + eth + e0);
          U0[5] = r0 * e0;
        }
        float U1[6];
#pragma unroll
        for (int m = 0; m < 6; m++) {
          float dU = -((sFx[m][ty][tx + 1] - sFx[m][ty][tx]) * A.inv_dx +
                       (sFy[m][ty + 1][tx] - sFy[m][ty][tx]) * A.inv_dy + (Fz_hi[m] - Fz_lo[m]) * A.inv_dz);
          U1[m] = U0[m] + dU * dt;
        }
        // cons_to_prim, :247-262
        float r1 = fmaxf(U1[0], RHO_P_FLOOR);
        float ir1 = rcp(r1);
        float u1 = U1[1] * ir1, v1 = U1[2] * ir1, w1 = U1[3] * ir1;
        float ke = 0.5f * (u1 * u1 + v1 * v1 + w1 * w1);
        float ev1 = fmaxf(U1[5] * ir1, 0.f);
        float e_th = fmaxf(U1[4] * ir1 - ke - ev1, THERMAL_ENERGY_FLOOR);
        float p1 = fmaxf(A.gm1 * r1 * e_th, RHO_P_FLOOR);
        const bool bad = !(__builtin_isfinite(r1) && __builtin_isfinite(p1) && __builtin_isfinite(u1) &&
                           __builtin_isfinite(v1) && __builtin_isfinite(w1) && __builtin_isfinite(ev1)) ||
                         r1 <= 0.f || p1 <= 0.f || ev1 < 0.f;
        if (bad) { r1 = A.in_r; u1 = A.in_u; v1 = A.in_v; w1 = A.in_w; p1 = A.in_p; ev1 = A.in_ev; } // :1284-1289
        float T1 = p1 * rcp(r1 * A.R);
        ev1 = fmaxf(ev1 + (evib_eq(A, T1) - ev1) * (dt * A.inv_tau_vib), 0.f); // :1290-1292

        if (A.sponge_n > 0 && x < A.sponge_n) { // :1295-1319
          float s = 1.0f - (float)x / (float)A.sponge_n;
          s = fminf(fmaxf(s, 0.0f), 1.0f);
          float k = A.sponge_strength * (s * s);
          r1 = fmaxf(r1 + k * (A.in_r - r1), RHO_P_FLOOR);
          p1 = fmaxf(p1 + k * (A.in_p - p1), RHO_P_FLOOR);
          u1 = u1 + k * (gain * A.in_u - u1);
          v1 = v1 + k * (gain * A.in_v - v1);
          w1 = w1 + k * (gain * A.in_w - w1);
          ev1 = fmaxf(ev1 + k * (A.in_ev - ev1), 0.f);
        }
        if (A.sponge_out_n > 0 && x >= (A.nx - A.sponge_out_n)) { // :1320-1344
          int xo2 = x - (A.nx - A.sponge_out_n);
          float s = (float)xo2 / (float)A.sponge_out_n;
          s = fminf(fmaxf(s, 0.0f), 1.0f);
          float k = A.sponge_out_strength * (s * s);
          r1 = fmaxf(r1 + k * (A.in_r - r1), RHO_P_FLOOR);
          p1 = fmaxf(p1 + k * (A.in_p - p1), RHO_P_FLOOR);
          u1 = u1 + k * (0.0f - u1);
          v1 = v1 + k * (0.0f - v1);
          w1 = w1 + k * (0.0f - w1);
          ev1 = fmaxf(ev1 + k * (A.in_ev - ev1), 0.f);
        }
        float a = soundspeed(A, p1, r1); // :1345-1351
        float ssum = (fabsf(u1) + a) * A.inv_dx + (fabsf(v1) + a) * A.inv_dy + (fabsf(w1) + a) * A.inv_dz;
        // :1338-1356 takes ssum into the maximum if it is finite and positive.  It is a sum of non-negative terms: as an unsigned integer
  // its bit pattern orders like the value, with +inf and the NaNs on top — (bits - 0x7f800000) >> 31 (arithmetic) is all ones exactly
  // for the finite ones; three full-rate integer instructions and one v_max_u32 for two compares, a class test and a select
  {
    const unsigned sb = __float_as_uint(ssum);
    const unsigned keep = (unsigned)((int)(sb - 0x7f800000u) >> 31) & ~(unsigned)((int)sb >> 31);   // finite, sign bit clear
    smax = __uint_as_float(max(__float_as_uint(smax), sb & keep));
  }
        fmx = fmaxf(fmaxf(fmx, r1), fabsf(u1));            // three v_max3_f32 (a balanced tree of fmaxf compiles to six v_max_f32)
        fmx = fmaxf(fmaxf(fmx, fabsf(v1)), fabsf(w1));
        fmx = fmaxf(fmaxf(fmx, p1), ev1);

        A.out[0][gi] = flog(fmaxf(r1, RHO_P_FLOOR)); // :1353-1358
        A.out[1][gi] = fasinh(u1 * A.inv_u_ref);
        A.out[2][gi] = fasinh(v1 * A.inv_u_ref);
        A.out[3][gi] = fasinh(w1 * A.inv_u_ref);
        A.out[4][gi] = flog(fmaxf(p1, RHO_P_FLOOR));
        A.out[5][gi] = flog(fmaxf(ev1, RHO_P_FLOOR));
      }
    }
    // ---- slide the window down one plane
#pragma unroll
    for (int m = 0; m < 6; m++) Fz_lo[m] = Fz_hi[m];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int m = 0; m < 6; m++) W[k][m] = W[k + 1][m];
  }

  // ---- max wavespeed and max |primitive|: wave64 butterflies, then one atomic each per workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    smax = fmaxf(smax, __shfl_xor(smax, o, 64));
    fmx = fmaxf(fmx, __shfl_xor(fmx, o, 64));
  }
  if (lane == 0) { sRed[0][wave] = smax; sRed[1][wave] = fmx; }
  __syncthreads();
  if (tid < 2) {
    float m = sRed[tid][0];
#pragma unroll
    for (int k = 1; k < NW; k++) m = fmaxf(m, sRed[tid][k]);
    tau::atomic_max_float_bits(maxw + tid, m);
  }
}

#ifndef TAU3D_SPLIT_TU   // (the split-step translation unit holds k_flux_xy and k_update_z only, see below)
__global__ __launch_bounds__(NT, TAU3D_STEP_WAVES) void k_step(const Args A) {
  __shared__ StepLds S;
  // linear block id -> work item, XCD-contiguous
  const unsigned b = tau::xcd_swizzle(blockIdx.x, (unsigned)(A.ntx * A.nty * A.nzc));
  const float dt = A.clk->dt, gain = A.clk->gain;
  // one scalar decision for the whole launch (see WLAM above)
  if (fast_form(A.clk->fmax_in, A.in_fmax)) step_body<true>(A, S, b, dt, gain, &A.clk->maxs_bits);
  else step_body<false>(A, S, b, dt, gain, &A.clk->maxs_bits);
}
#endif

// ---------------------------------------------------------------- the split step: k_flux_xy + k_update_z
// The fused kernel above carries a five-plane register window through a tile that also stages LDS planes: 168 VGPRs,
// three waves per SIMD, two barriers per plane — and a gfx950 wave issues at most one VALU instruction every ~6-7
// cycles (profiles/r02/valu_calib.txt), so three resident waves only just cover a ~2.3-cycle pipe and every stall of
// one of them is lost issue time (measured: 4.0 cycles per instruction against ~3.1 for the instruction mix).
// The split trades HBM traffic, of which this VALU-bound step uses a tenth, for occupancy:
//   k_flux_xy   one plane per workgroup, no z dependence at all (so it needs no z halo): stage the tile + x/y halo,
//               decoded on load, in LDS, every x / y face once, write the x/y flux divergence (6 floats per cell);
//   k_update_z  one column per lane, no barrier: the z window (decoded on load) in a private LDS ring, every z face
//               once, add the x/y divergence, update, re-encode, write the new state.
// Both decode the encoded state where they load it (1.6 + 1 decodes per cell and step; the fused kernel: 2.7).  A cache
// of the decoded primitives beside the state (written by k_update_z, read by both) was measured: it saves k_flux_xy
// 4 % but costs k_update_z, which is HBM-bound at ~5 TB/s, 24 of its 97 bytes per cell — 2.67 against 2.35 ms.
#ifndef TAU3D_XY_TY
#define TAU3D_XY_TY 16
#endif
constexpr int XT = 32, YT = TAU3D_XY_TY;   // k_flux_xy tile; XT * YT threads
constexpr int XNT = XT * YT, XNW = XNT / 64;
// row stride of the staged plane.  40 (the fused kernel's) puts a COLUMN of cells on four banks: the ring and far-face rounds,
// whose lanes sit one per row, then run four lanes per bank (SQ_LDS_BANK_CONFLICT 26 % of the LDS-active cycles, profiles/r02).
// An odd stride spreads a column over all 32 banks; rows stay conflict-free (32 consecutive words per half-wave).
#ifndef TAU3D_XPXS
#define TAU3D_XPXS 41
#endif
constexpr int XPXS = TAU3D_XPXS;
constexpr int XPY = YT + 2 * HALO, XPLANE = XPY * XPXS;
static_assert(YT % 2 == 0 && XT + YT <= 64 && 2 * HALO * (XT + YT) <= XNT, "ring rounds are one wave each, halo staging one round");
// Cell-centred reconstruction in x and y as well: a thread weights its own cell ONCE per axis (weno_cell: the three
// smoothness indicators serve the left state at the high face and the right state at the low face) and the two states
// of a face meet through LDS (y) or a lane shift (x).  Around the tile, one ring of cells on each side is evaluated
// for the one state the tile's outermost faces need: XT + YT lanes of one wave per side, and the XT + YT far faces in a
// third partial round; the three rounds go to three different waves, rotating with the plane.
struct XyLds {
  float sP[6][XPLANE];          // the plane, primitives, x/y halo 3
  uint8_t sS[XPLANE];
  float sLy[6][YT + 1][XT];     // [r][x]: left state of the face below row r (from the cell in row r-1; r = 0: ring)
  // (the flux through that face takes the SAME slot: a slot's left state is read by exactly one lane — the one that then
  //  writes the flux there — so no second [6][YT + 1][XT] array: 51 -> 38 KB per workgroup, four workgroups per CU)
  float sLx0[6][YT];            // left state of the tile's low-x faces (ring cell x = -1)
  float sLxT[6][YT];            // left state of the far x faces (own column XT-1)
  float sRxT[6][YT];            // right state of the far x faces (ring cell x = XT)
  float sRyT[6][XT];            // right state of the far y faces (ring cell y = YT)
  float sFxT[6][YT];            // flux through the far x faces
  unsigned wuni[XNW];           // uniform-region exit: per wave, 1 when every cell it staged equals the tile's first cell
};
__device__ __forceinline__ float lane_above(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
}

// what k_flux_xy knows about its cell when the x / y faces are done (core / store split: the core also served the
// small-grid experiment of DESIGN §4.1's tried-list)
struct XyCell { float d[6]; bool in_xy, own_solid; int x, yw, z, lc; };
// SOLID = false: the tile and its 3-cell x / y halo hold no solid cell (a flag per tile and plane, computed once from the
// static mask: tau3d_create / tau3d_init) — no solid bytes are staged or read and the faces take the WENO states as they are.
template <bool FAST, bool SOLID> __device__ __forceinline__ void flux_xy_core(const Args &A, XyLds &S, XyCell &C, int bx, int by, int z) {
  auto &sP = S.sP; auto &sS = S.sS;
  const int tid = threadIdx.x;
  const int tx = tid & (XT - 1), ty = tid >> 5;
  const int lane = tid & 63, wave = tid >> 6;
  const Gas G = gas_vgpr(A);
  const float uref = vreg(A.u_ref);   // three multiplies per decoded cell: an SGPR operand would halve their rate

  const int bx0 = bx * XT, by0 = by * YT;

  const int x = bx0 + tx, y = by0 + ty;
  const bool in_xy = (x < A.nx) && (y < A.ny);
  const bool ynear = A.ny >= YT + HALO;   // y in [-HALO, ny + YT + HALO): one conditional add / subtract wraps it
  const int yw = wrap_near(y, A.ny, ynear);
  const int zh = z + HALO;
  const int zg = wrapi(A.z0 + z, A.nz);
  const int lc = (ty + HALO) * XPXS + (tx + HALO);

  const size_t plane_n = (size_t)A.nx * A.ny;
  const GChar *const qpl = (const GChar *)(A.in0 + (size_t)zh * plane_n);
  const uint8_t *const spl = A.solid + (size_t)zh * plane_n;
  const size_t fs4 = (size_t)A.fstride << 2;
  bool own_solid = false;
  // Uniform-region exit (round 6).  Upstream of the bow shock a supersonic flow IS the inflow state, cell for cell and bit for bit
  // (the same arithmetic runs on the same operands in every such cell, every step), and so is everything the disturbance has not
  // reached yet.  A tile whose cells and x/y halo cells all hold one encoded state has one state at every face: the reconstruction
  // returns the cell value exactly (all differences zero), every face takes the same flux, and the divergence is (F - F) / dx +
  // (F - F) / dy = +0 in every cell — exactly what the code below would compute and store.  Such a tile writes one flag instead
  // (A.dzero, read by k_update_z in place of the six zeros per cell) and leaves.  The test: every staged cell's six ENCODED values
  // against the tile's first cell's (a scalar load); ghost columns never pass, tiles near the body are not asked (SOLID).
  const bool utest = !SOLID && A.dzero != nullptr;
  const size_t tile_i = ((size_t)z * A.nty + by) * A.ntx + bx;
  float uref6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // Edge strips (a handle that keeps the tile list, see k_tile_predict): a tile that is NOT uniform still says whether its first /
  // last six columns and rows are — against its own corner cells — because that is all its neighbours' predictions need of it.
  const bool strips = utest && A.uflag_w != nullptr;
  float urefE[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, urefN[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (utest) {
    const float *const rp = A.in0 + (size_t)zh * plane_n + (size_t)wrap_near(by0, A.ny, ynear) * A.nx + bx0;
#pragma unroll
    for (int m = 0; m < 6; m++) uref6[m] = rp[(size_t)m * A.fstride];
    if (strips) {   // (whole tiles: both cells are inside the grid)
      const float *const rn = A.in0 + (size_t)zh * plane_n + (size_t)wrap_near(by0 + YT - 1, A.ny, ynear) * A.nx + bx0;
#pragma unroll
      for (int m = 0; m < 6; m++) { urefE[m] = rp[(size_t)m * A.fstride + (XT - 1)]; urefN[m] = rn[(size_t)m * A.fstride]; }
    }
  }
  unsigned ufail = 0u;   // which claims this thread's cells refute: UF_ALL, UF_W, UF_E, UF_S, UF_N
  { // ---- stage the plane: own cell + the halo cells (3 rows above / below, 3 columns left / right; no corners)
    float q[6];
    bool osol;
    unsigned eqm;
    fetch_cell_e(A, uref, qpl, fs4, spl, x, yw, zg, q, osol, utest, uref6, strips, urefE, urefN, eqm);
    ufail = (eqm & 1u) ? 0u : UF_ALL;
    if (strips) {
      const bool f0 = !(eqm & 1u), fe = !(eqm & 2u), fn = !(eqm & 4u);
      ufail |= ((tx < USTRIP && f0) ? UF_W : 0u) | ((tx >= XT - USTRIP && fe) ? UF_E : 0u) | ((ty < USTRIP && f0) ? UF_S : 0u) |
               ((ty >= YT - USTRIP && fn) ? UF_N : 0u);
    }
#pragma unroll
    for (int m = 0; m < 6; m++) sP[m][lc] = q[m];
    if (SOLID) { own_solid = osol; sS[lc] = osol ? 1 : 0; }
    constexpr int NROWS = 2 * HALO * XT;
    constexpr int NHALO = NROWS + YT * 2 * HALO;
    if (tid < NHALO) {
      const int p = tid;
      int ly, lx;
      if (p < NROWS) {
        const int r = p / XT;
        ly = (r < HALO) ? r : r + YT;
        lx = HALO + (p - r * XT);
      } else {
        const int qq = p - NROWS;
        const int r = qq / (2 * HALO), c = qq - r * (2 * HALO);
        ly = HALO + r;
        lx = (c < HALO) ? c : c + XT;
      }
      const int gx = bx0 + lx - HALO;
      const int gy = wrap_near(by0 + ly - HALO, A.ny, ynear);
      bool sol;
      fetch_cell_e(A, uref, qpl, fs4, spl, gx, gy, zg, q, sol, utest, uref6, strips, urefE, urefN, eqm);
      ufail |= (eqm & 1u) ? 0u : UF_ALL;
      if (strips) {   // a halo row belongs to the column strips it prolongs, a halo column to the row strips
        const bool f0 = !(eqm & 1u), fe = !(eqm & 2u), fn = !(eqm & 4u);
        const int c = lx - HALO, r = ly - HALO;
        if (p < NROWS) ufail |= ((c < USTRIP && f0) ? UF_W : 0u) | ((c >= XT - USTRIP && fe) ? UF_E : 0u);
        else ufail |= ((r < USTRIP && f0) ? UF_S : 0u) | ((r >= YT - USTRIP && fn) ? UF_N : 0u);
      }
      const int li = ly * XPXS + lx;
#pragma unroll
      for (int m = 0; m < 6; m++) sP[m][li] = q[m];
      if (SOLID) sS[li] = sol ? 1 : 0;
    }
  }
  if (utest) {   // the claims no lane of this wave refutes
    unsigned pass = __builtin_amdgcn_ballot_w64((ufail & UF_ALL) != 0u) == 0ull ? UF_ALL : 0u;
    if (strips) {
      pass |= (__builtin_amdgcn_ballot_w64((ufail & UF_W) != 0u) == 0ull ? UF_W : 0u) | (__builtin_amdgcn_ballot_w64((ufail & UF_E) != 0u) == 0ull ? UF_E : 0u) |
              (__builtin_amdgcn_ballot_w64((ufail & UF_S) != 0u) == 0ull ? UF_S : 0u) | (__builtin_amdgcn_ballot_w64((ufail & UF_N) != 0u) == 0ull ? UF_N : 0u);
    }
    if (lane == 0) S.wuni[wave] = pass;
  }
  __syncthreads();
  unsigned all = 0u;
  if (utest) {
    all = ~0u;
#pragma unroll
    for (int w = 0; w < XNW; w++) all &= S.wuni[w];
    if (all & UF_ALL) {      // (the same word for every thread: the workgroup leaves together)
      if (tid == 0) {
        A.dzero[tile_i] = 1u;
        if (A.uflag_w != nullptr) {
          A.uflag_w[tile_i] = UF_ALL;
#pragma unroll
          for (int m = 0; m < 6; m++) A.uref_w[tile_i * UREC + m] = uref6[m];
        }
      }
      C.in_xy = false;
      return;
    }
  }
  if (A.dzero != nullptr && tid == 0) {
    A.dzero[tile_i] = 0u;
    if (A.uflag_w != nullptr) {
      const unsigned sf = strips ? (all & (UF_W | UF_E | UF_S | UF_N)) : 0u;
      A.uflag_w[tile_i] = sf;
      if (sf != 0u) {
#pragma unroll
        for (int m = 0; m < 6; m++) {
          A.uref_w[tile_i * UREC + m] = uref6[m]; A.uref_w[tile_i * UREC + 8 + m] = urefE[m]; A.uref_w[tile_i * UREC + 16 + m] = urefN[m];
        }
      }
    }
  }
  // ---- edge states of the own cell; ring cells
  // One variable at a time.  Left alone, hipcc runs the six variables breadth-first (all first differences, then all
  // smoothness indicators, ...) and needs ~140 VGPRs for it; occupancy is worth more than that ILP here.  The empty asm
  // ties the next variable's LDS address to this variable's results, which orders them.
  float Rx[6], Ry[6], Lxo[6];
  int lcs = lc;
#pragma unroll
  for (int m = 0; m < 6; m++) {
    const float *p = &sP[m][lcs];
    float Ly;
    weno_cell<FAST>(p[-2], p[-1], p[0], p[1], p[2], Lxo[m], Rx[m]);
    weno_cell<FAST>(p[-2 * XPXS], p[-XPXS], p[0], p[XPXS], p[2 * XPXS], Ly, Ry[m]);
    S.sLy[m][ty + 1][tx] = Ly;
    asm volatile("" : "+v"(lcs), "+v"(Lxo[m]), "+v"(Rx[m]), "+v"(Ry[m]));
  }
  if (tx == XT - 1) {
#pragma unroll
    for (int m = 0; m < 6; m++) S.sLxT[m][ty] = Lxo[m];
  }
  constexpr int TAU3D_XY_ROT = 0, TAU3D_XY_WC = 2;   // (rotation with the tile as well, and other wave offsets: measured in round 5, profiles/r05/ab_xy_rotation_waves.txt)
  // the waves that take the extra rounds rotate with the plane AND (TAU3D_XY_ROT) with the tile: the workgroups resident on a CU
  // at one time are tiles of one or two planes, and the extra rounds of all of them sat on the same two SIMDs
  const int wA = (int)((unsigned)(z + TAU3D_XY_ROT * (bx + 3 * by)) % (unsigned)XNW), wC = (wA + TAU3D_XY_WC) % XNW;
  // The ring: the cells at x = -1 and y = -1 give their LEFT states (of the tile's low faces), those at x = XT and y = YT
  // their RIGHT states (of the far faces).  One lane per (ring cell, variable) — 2 x 48 cells x 6 variables = 576 tasks,
  // one round of all eight waves and a 64-task remainder on one wave (rotating with the plane).  With one lane per ring
  // cell and the six variables in sequence, two waves ran ~270 instructions each while the other six sat at the barrier.
  constexpr int RING = XT + YT, RTASKS = 2 * 6 * RING;
  static_assert(RTASKS <= XNT + 64, "ring tasks: one round of the workgroup (+ one wave)");
  auto ring_task = [&](int t) {
    const int side = t >= 6 * RING ? 1 : 0;   // 0: low ring (left states), 1: high ring (right states)
    const int r = t - side * (6 * RING);
    const int m = r / RING, ln = r - m * RING;
    const bool isx = ln < YT;
    const int c0 = isx ? (ln + HALO) * XPXS + (side ? XT + HALO : HALO - 1) : (side ? YT + HALO : HALO - 1) * XPXS + (ln - YT + HALO);
    // one-sided (round 5): the high ring walks its five cells in reverse — Rlo of a cell is Lhi of the mirrored stencil, bit
    // for bit (weno_cell_hi) — so a ring task pays for the one state that is read
    const int st0 = isx ? 1 : XPXS, st = side ? -st0 : st0;
    const float *p = &sP[0][0] + m * XPLANE + c0;
    float *dst = side ? (isx ? &S.sRxT[0][0] + m * YT + ln : &S.sRyT[0][0] + m * XT + (ln - YT))
                      : (isx ? &S.sLx0[0][0] + m * YT + ln : &S.sLy[0][0][0] + m * ((YT + 1) * XT) + (ln - YT));
    *dst = weno_cell_hi<FAST>(p[-2 * st], p[-st], p[0], p[st], p[2 * st]);
  };
  if (RTASKS >= XNT || tid < RTASKS) ring_task(tid);
  if (RTASKS > XNT && wave == wA) ring_task(XNT + lane);
  __syncthreads();

  // ---- faces: low-x and low-y of the own cell; the tile's far faces
  float Fx[6], Fy[6];
  {
    Prim L, R;
#pragma unroll
    for (int m = 0; m < 6; m++) {
      const float l = lane_below(Lxo[m]);
      L.q[m] = (tx == 0) ? S.sLx0[m][ty] : l;
      R.q[m] = Rx[m];
    }
    if (SOLID) {
      float lo[6], hi[6];
      unsigned s = 0;
#pragma unroll
      for (int m = 0; m < 6; m++) { lo[m] = sP[m][lc - 1]; hi[m] = sP[m][lc]; }
#pragma unroll
      for (int k = 0; k < 6; k++) s |= (unsigned)sS[lc + (k - 3)] << k;
      solid_override(L, R, lo, hi, s, 0);
    }
    prim_floor(L);
    prim_floor(R);
    const Cons F = hllc(G, L, R, 0);
#pragma unroll
    for (int m = 0; m < 6; m++) Fx[m] = F.c[m];
  }
  {
    Prim L, R;
#pragma unroll
    for (int m = 0; m < 6; m++) {
      L.q[m] = S.sLy[m][ty][tx];
      R.q[m] = Ry[m];
    }
    if (SOLID) {
      float lo[6], hi[6];
      unsigned s = 0;
#pragma unroll
      for (int m = 0; m < 6; m++) { lo[m] = sP[m][lc - XPXS]; hi[m] = sP[m][lc]; }
#pragma unroll
      for (int k = 0; k < 6; k++) s |= (unsigned)sS[lc + (k - 3) * XPXS] << k;
      solid_override(L, R, lo, hi, s, 1);
    }
    prim_floor(L);
    prim_floor(R);
    const Cons F = hllc(G, L, R, 1);
#pragma unroll
    for (int m = 0; m < 6; m++) { Fy[m] = F.c[m]; S.sLy[m][ty][tx] = F.c[m]; }
  }
  if (wave == wC && lane < XT + YT) { // far faces: x faces at column XT (rows 0 .. YT-1), y faces at row YT
    const bool isx = lane < YT;
    const int c0 = isx ? (lane + HALO) * XPXS + (XT + HALO) : (YT + HALO) * XPXS + (lane - YT + HALO);   // the cell above the face
    const int st = isx ? 1 : XPXS;
    Prim L, R;
#pragma unroll
    for (int m = 0; m < 6; m++) {
      L.q[m] = isx ? S.sLxT[m][lane] : S.sLy[m][YT][lane - YT];
      R.q[m] = isx ? S.sRxT[m][lane] : S.sRyT[m][lane - YT];
    }
    if (SOLID) {
      float lo[6], hi[6];
      unsigned s = 0;
#pragma unroll
      for (int m = 0; m < 6; m++) { lo[m] = sP[m][c0 - st]; hi[m] = sP[m][c0]; }
#pragma unroll
      for (int k = 0; k < 6; k++) s |= (unsigned)sS[c0 + (k - 3) * st] << k;
      solid_override_xy(L, R, lo, hi, s, isx);
    }
    prim_floor(L);
    prim_floor(R);
    const Cons F = hllc(G, L, R, isx ? 0 : 1);
#pragma unroll
    for (int m = 0; m < 6; m++) {
      if (isx) S.sFxT[m][lane] = F.c[m]; else S.sLy[m][YT][lane - YT] = F.c[m];
    }
  }
  __syncthreads();

  // ---- x/y flux divergence of the own cell
  const float inv_dx = vreg(A.inv_dx), inv_dy = vreg(A.inv_dy);   // twelve uses: SGPR operands would make them half rate
#pragma unroll
  for (int m = 0; m < 6; m++) {
    const float up = lane_above(Fx[m]);
    const float fxh = (tx == XT - 1) ? S.sFxT[m][ty] : up;
    C.d[m] = (fxh - Fx[m]) * inv_dx + (S.sLy[m][ty + 1][tx] - Fy[m]) * inv_dy;
  }
  C.in_xy = in_xy; C.own_solid = own_solid; C.x = x; C.yw = yw; C.z = z; C.lc = lc;
}
template <bool FAST> __device__ __forceinline__ void flux_xy_tile(const Args &A, XyLds &S, unsigned b);
template <bool FAST> __device__ __forceinline__ void flux_xy_body(const Args &A, XyLds &S, unsigned bid) {
  flux_xy_tile<FAST>(A, S, tau::xcd_swizzle(bid, (unsigned)(A.ntx * A.nty * A.nzc)));
}
// list mode: work item i of *A.ucount is tile A.ulist[i] (a whole-slab launch: the tile index IS (plane, tile row, tile column))
template <bool FAST> __device__ __forceinline__ void flux_xy_listed(const Args &A, XyLds &S, unsigned i) {
  flux_xy_tile<FAST>(A, S, A.ulist[i]);
}
template <bool FAST> __device__ __forceinline__ void flux_xy_tile(const Args &A, XyLds &S, unsigned b) {
  XyCell C;
  const int bx = (int)(b % (unsigned)A.ntx); b /= (unsigned)A.ntx;
  const int by = (int)(b % (unsigned)A.nty);
  const int bz = (int)(b / (unsigned)A.nty);
  const int z = (bz >= A.nzc1) ? A.zl_lo2 + (bz - A.nzc1) : A.zl_lo + bz;   // here a "chunk" is one plane
  // one scalar load, one scalar branch: ~90 % of the tiles of the 512^3 sphere case hold no solid cell
  const unsigned tflag = A.xyflag == nullptr ? 1u : A.xyflag[((size_t)z * A.nty + by) * A.ntx + bx];
  if (tflag == 2u) {         // the tile is inside the body: no cell of it takes a divergence (4 % of the tiles of the 512^3 sphere case)
    if (A.dzero != nullptr && threadIdx.x == 0) {
      A.dzero[((size_t)z * A.nty + by) * A.ntx + bx] = 0u;
      if (A.uflag_w != nullptr) A.uflag_w[((size_t)z * A.nty + by) * A.ntx + bx] = 0u;
    }
    return;
  }
  const bool any_solid = tflag != 0u;
  if (any_solid) flux_xy_core<FAST, true>(A, S, C, bx, by, z);
  else flux_xy_core<FAST, false>(A, S, C, bx, by, z);
  if (C.in_xy && !C.own_solid) {
    GChar *const dpl = (GChar *)(A.d0 + (size_t)C.z * ((size_t)A.nx * A.ny));
    const size_t ds4 = (size_t)A.dstride << 2;
    const unsigned vo = lane_off((unsigned)(C.yw * A.nx + C.x) << 2);
#pragma unroll
    for (int m = 0; m < 6; m++) gst(dpl + m * ds4, vo, C.d[m]);
  }
}

// Round 6: 8 waves per SIMD = FOUR workgroups per CU (4 x 37.9 KB of LDS fits the 160 KB).  The kernel needed 67 VGPRs under the cap of
// 80 (six waves, three workgroups); asked for 64 it takes 63 without a spill, and the fourth resident workgroup is worth 4 % with the
// uniform-region exits on or off (same box, interleaved, twice: headline 33.5 -> 34.8-35.0, late state 29.5 -> 30.4-30.8, every face
// evaluated 25.0-25.7 -> 26.3-26.9 Gcell/s) — a short-lived workgroup spends most of its life waiting for its staging loads.
#ifndef TAU3D_XY_WAVES
#define TAU3D_XY_WAVES 8
#endif
// The two kernels of the split step are compiled in a translation unit of their own (this file again with -DTAU3D_SPLIT_TU,
// Makefile: build/h3d_split.o) under `-mllvm -amdgpu-sched-strategy=max-ilp`: the ILP-first list scheduler is worth 1.5-2 % on
// both (3.89 -> 3.82 and 2.29 -> 2.26 ms at 512^3, same box, interleaved) within their launch-bound register caps, while it takes
// the fused k_step — bounded at two waves per SIMD — from 152 to 252 VGPRs and 96^3 from 92 to 105 us.  The option is per module.
#ifdef TAU3D_SPLIT_TU
// ONE WENO weight form per kernel (round 4).  Which form a step takes is decided on the device (the field range lives in the
// clock block; the host never waits for it), and rounds 2-3 carried both bodies in one kernel behind a scalar branch.  Measured,
// same box, interleaved: the fast body alone in its kernel runs k_flux_xy 3.79 -> 3.67 ms and k_update_z 2.37 -> 2.32 (the
// kernel with both bodies is twice the code, and k_update_z's allocation is the maximum of the two: 96 VGPRs, 33 spilled SGPRs
// and 16 B of scratch against 94 / 0 / 0).  So each form is its own kernel, and a step launches BOTH for each piece: the one the
// host expects (its last look at the clock block: tau3d_get_clock, tau3d_field_range, tau3d_step) over the full grid, the other
// as a safety net — a small resident grid that strides over the tiles.  Each leaves at once when the device's range says the
// other form is due, so exactly one of the two does the work, whatever the host guessed: a wrong guess costs time (an empty full
// grid, then the strided kernel), never correctness.  STRIDE = false is the launch of rounds 2-3, one tile per workgroup.
using XyShared = XyLds;
constexpr int XYNT = XNT;
template <bool FAST> __device__ __forceinline__ void xy_body(const Args &A, XyShared &S, unsigned b) {
  flux_xy_body<FAST>(A, S, b);
}
// The launch over the list of tiles k_tile_predict could not clear.  The host sizes the grid from the length of an earlier list plus
// a margin (split_xy); one tile per workgroup and NO loop — a loop over tiles costs this kernel its allocation (hoisted index
// arithmetic: 64 VGPRs + 176 B of scratch against 63 + 0).  What a short grid leaves, and the whole list when the device's field
// range says the other weight form is due, is k_flux_xy_list_rest's: a small resident grid that strides, both bodies behind a
// branch, empty on every step of a sane run.
template <bool FAST> __global__ __launch_bounds__(XYNT, TAU3D_XY_WAVES) void k_flux_xy_list(const Args A) {
  __shared__ XyShared S;
  const unsigned n = *A.ucount;
  if (blockIdx.x == 0 && threadIdx.x == 0) *(volatile unsigned *)A.ucount_host = n;
  if (fast_form(A.clk->fmax_in, A.in_fmax) != FAST) return;
  // (k_tile_predict appends in roughly ascending tile order: an XCD takes a contiguous eighth of the list, neighbours share its L2)
  // (the swizzle over the part of the grid that has work: over a grid with a margin of idle workgroups it would hand the first XCDs
  // all of the list and the last ones none — measured with a grid of twice the list: 1.27 -> 2.36 ms)
  const unsigned i = tau::xcd_swizzle(blockIdx.x, n < gridDim.x ? n : gridDim.x);
  if (i < n) flux_xy_listed<FAST>(A, S, i);
}
__global__ __launch_bounds__(XYNT, TAU3D_XY_WAVES) void k_flux_xy_list_rest(const Args A) {
  __shared__ XyShared S;
  const unsigned n = *A.ucount;
  const bool fast = fast_form(A.clk->fmax_in, A.in_fmax);
  const unsigned first = (fast == (A.list_fast != 0)) ? A.list_grid : 0u;   // the main launch ran this form: it did [0, list_grid)
  for (unsigned i = first + blockIdx.x; i < n; i += gridDim.x) {
    if (fast) flux_xy_listed<true>(A, S, i); else flux_xy_listed<false>(A, S, i);
    __syncthreads();
  }
}
template <bool FAST, bool STRIDE> __global__ __launch_bounds__(XYNT, TAU3D_XY_WAVES) void k_flux_xy(const Args A) {
  __shared__ XyShared S;
  if (fast_form(A.clk->fmax_in, A.in_fmax) != FAST) return;
  if (!STRIDE) { xy_body<FAST>(A, S, blockIdx.x); return; }
  const unsigned nb = (unsigned)(A.ntx * A.nty * A.nzc);
  for (unsigned b = blockIdx.x; b < nb; b += gridDim.x) {
    xy_body<FAST>(A, S, b);
    __syncthreads();   // the next tile's staging overwrites what slow waves of this one still read
  }
}
// The repeat of a launch that ran ahead of the clock (Z-slab ring: the x/y fluxes of step n+1 start before the all-reduced
// field range of step n is in).  It read the range of the step BEFORE; the commit says whether that put it on the wrong weight
// form (DevClock::form_flip) — if not, which is every step of a sane run, each workgroup of this small resident grid leaves after
// one scalar load.  Both bodies behind a branch: the register allocation of the pair does not matter for a launch that never runs.
__global__ __launch_bounds__(XYNT, TAU3D_XY_WAVES) void k_flux_xy_fix(const Args A) {
  __shared__ XyShared S;
  if (A.clk->form_flip == 0u) return;
  const bool fast = fast_form(A.clk->fmax_in, A.in_fmax);
  const unsigned nb = (unsigned)(A.ntx * A.nty * A.nzc);
  for (unsigned b = blockIdx.x; b < nb; b += gridDim.x) {
    if (fast) xy_body<true>(A, S, b); else xy_body<false>(A, S, b);
    __syncthreads();
  }
}
void launch_flux_xy_fix(unsigned nwg, hipStream_t s, const Args &A) {
  hipLaunchKernelGGL(k_flux_xy_fix, dim3(nwg < 768u ? nwg : 768u), dim3(XYNT), 0, s, A);
}
void launch_flux_xy(unsigned nwg, hipStream_t s, const Args &A, bool expect_fast) {
  const unsigned net = nwg < 768u ? nwg : 768u;   // three workgroups per CU resident
#ifdef TAU3D_FAST_ONLY   // ISA statistics only (scripts/isa_kernel_mix.py)
  hipLaunchKernelGGL((k_flux_xy<true, false>), dim3(nwg), dim3(XYNT), 0, s, A);
#else
  if (expect_fast) {
    hipLaunchKernelGGL((k_flux_xy<true, false>), dim3(nwg), dim3(XYNT), 0, s, A);
    hipLaunchKernelGGL((k_flux_xy<false, true>), dim3(net), dim3(XYNT), 0, s, A);
  } else {
    hipLaunchKernelGGL((k_flux_xy<false, false>), dim3(nwg), dim3(XYNT), 0, s, A);
    hipLaunchKernelGGL((k_flux_xy<true, true>), dim3(net), dim3(XYNT), 0, s, A);
  }
#endif
}
void launch_flux_xy_list(unsigned nwg, hipStream_t s, const Args &A0, bool expect_fast) {
  Args A = A0;
  A.list_grid = nwg; A.list_fast = expect_fast ? 1 : 0;
  if (expect_fast) hipLaunchKernelGGL((k_flux_xy_list<true>), dim3(nwg), dim3(XYNT), 0, s, A);
  else hipLaunchKernelGGL((k_flux_xy_list<false>), dim3(nwg), dim3(XYNT), 0, s, A);
  hipLaunchKernelGGL(k_flux_xy_list_rest, dim3(768), dim3(XYNT), 0, s, A);
}
#else
void launch_flux_xy_list(unsigned nwg, hipStream_t s, const Args &A, bool expect_fast);   // the same over k_tile_predict's list
void launch_flux_xy(unsigned nwg, hipStream_t s, const Args &A, bool expect_fast);   // XNT threads per workgroup; both weight forms, see k_flux_xy
void launch_flux_xy_fix(unsigned nwg, hipStream_t s, const Args &A);                 // the repeat of a launch that ran ahead of the clock
#endif

// The update of one fluid cell, :1266-1358: conservative update from the x/y divergence D and the two z-face fluxes,
// repairs, Landau-Teller relaxation, sponges, the max-wavespeed / max-|primitive| contributions, re-encoding.  One
// function (k_update_z; a one-plane-per-workgroup kernel for small grids that shared it was measured and dropped).
// the constants update_cell uses three times per cell, in VGPRs (an SGPR operand halves the issue rate); gamma and gamma - 1
// come from the Gas copy the z face already holds.  k_update_z has three registers to spare under its five-wave limit (96):
// the constants with one use per cell stay in SGPRs.
struct UpdK { float gm1, inv_gm1, gamma, inv_u_ref, absmask; };
__device__ __forceinline__ UpdK updk_vgpr(const Args &A, const Gas &G) { return UpdK{G.gm1, G.inv_gm1, G.gamma, vreg(A.inv_u_ref), vlit(0x7fffffffu)}; }
__device__ __forceinline__ void update_cell(const Args &A, const UpdK &K, const float (&own)[6], const float (&D)[6], const float (&Fz_lo)[6],
                                            const float (&Fz_hi)[6], float dt, float inv_dz, float gain, int x, float (&E)[6],
                                            float &smax, float &fmx) {
  const float r0 = own[IR], u0 = own[IU], v0 = own[IV], w0 = own[IW], p0 = own[IP], e0 = own[IE];
  float U0[6];
  U0[0] = r0; U0[1] = r0 * u0; U0[2] = r0 * v0; U0[3] = r0 * w0;
  {
    float ke = 0.5f * (u0 * u0 + v0 * v0 + w0 * w0);
    // r e_th = r p / max((gamma - 1) r, floor) = p / (gamma - 1) wherever the floor does not bite (r >= 1e-29): no reciprocal
    // (hllc forms its conserved states the same way)
    const float reth = fminf(p0 * K.inv_gm1, (p0 * r0) * (1.f / RHO_P_FLOOR));   // (one v_min for a compare and a select; the same bits)
    U0[4] = r0 * (ke + e0) + reth;
    U0[5] = r0 * e0;
  }
  float U1[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    float dU = -(D[m] + (Fz_hi[m] - Fz_lo[m]) * inv_dz);
    U1[m] = U0[m] + dU * dt;
  }
  float r1 = fmaxf(U1[0], RHO_P_FLOOR);
  float ir1 = rcp(r1);
  float u1 = U1[1] * ir1, v1 = U1[2] * ir1, w1 = U1[3] * ir1;
  float ke = 0.5f * (u1 * u1 + v1 * v1 + w1 * w1);
  float ev1 = fmaxf(U1[5] * ir1, 0.f);
  float e_th = fmaxf(U1[4] * ir1 - ke - ev1, THERMAL_ENERGY_FLOOR);
  float p1 = fmaxf(K.gm1 * r1 * e_th, RHO_P_FLOOR);
  // :1300-1309 resets a cell whose new state holds a non-finite value or a non-positive density / pressure.  r1, p1, ev1 come out of
  // fmaxf against a positive floor / zero (never NaN, never below it), so the sign tests cannot fire and what is left is "all six
  // finite" — the largest |bit pattern| of the six below the exponent-all-ones patterns (infinities and NaNs sort above every finite
  // value as unsigned integers): two v_max3_u32 and one compare where six class tests and three compares stood.
  const unsigned amask = __float_as_uint(K.absmask);
  const unsigned hi6 = tau::max3u(tau::max3u(__float_as_uint(r1), __float_as_uint(p1), __float_as_uint(ev1)),
                                  __float_as_uint(u1) & amask, __float_as_uint(v1) & amask) ;
  const bool bad = max(hi6, __float_as_uint(w1) & amask) >= 0x7f800000u;
  float r1e = r1, p1e = p1;   // what the encode takes the logarithm of: floored already, except after the reset (inflow values as given)
  if (bad) {
    r1 = A.in_r; u1 = A.in_u; v1 = A.in_v; w1 = A.in_w; p1 = A.in_p; ev1 = A.in_ev; ir1 = rcp(r1);
    r1e = fmaxf(r1, RHO_P_FLOOR); p1e = fmaxf(p1, RHO_P_FLOOR);
  }
  {   // evib_eq(T), T = p / (r R), :206-211: theta_v / max(T, floor) = theta_v r R / max(p, floor r R) — one reciprocal for T and 1 / T
    const float rR = r1 * A.R;
    const float av = (A.theta_v * rR) * rcp(fmaxf(p1, NEWTON_TEMP_FLOOR * rR));
    const float eq = A.Rtheta * rcp(fmaxf(fexp(av) - 1.f, NEWTON_TEMP_FLOOR));
    ev1 = fmaxf(ev1 + (eq - ev1) * (dt * A.inv_tau_vib), 0.f);
  }

  if (A.sponge_n > 0 && x < A.sponge_n) {
    float s = 1.0f - (float)x / (float)A.sponge_n;
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    float k = A.sponge_strength * (s * s);
    r1 = fmaxf(r1 + k * (A.in_r - r1), RHO_P_FLOOR);
    p1 = fmaxf(p1 + k * (A.in_p - p1), RHO_P_FLOOR);
    u1 = u1 + k * (gain * A.in_u - u1);
    v1 = v1 + k * (gain * A.in_v - v1);
    w1 = w1 + k * (gain * A.in_w - w1);
    ev1 = fmaxf(ev1 + k * (A.in_ev - ev1), 0.f);
    ir1 = rcp(r1);
    r1e = r1; p1e = p1;
  }
  if (A.sponge_out_n > 0 && x >= (A.nx - A.sponge_out_n)) {
    int xo2 = x - (A.nx - A.sponge_out_n);
    float s = (float)xo2 / (float)A.sponge_out_n;
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    float k = A.sponge_out_strength * (s * s);
    r1 = fmaxf(r1 + k * (A.in_r - r1), RHO_P_FLOOR);
    p1 = fmaxf(p1 + k * (A.in_p - p1), RHO_P_FLOOR);
    u1 = u1 + k * (0.0f - u1);
    v1 = v1 + k * (0.0f - v1);
    w1 = w1 + k * (0.0f - w1);
    ev1 = fmaxf(ev1 + k * (A.in_ev - ev1), 0.f);
    ir1 = rcp(r1);
    r1e = r1; p1e = p1;
  }
  float a = fsqrt(fmaxf(K.gamma * p1 * ir1, DENOM_EPS));   // soundspeed, :264-266 (1 / r: the update's own, re-formed only where a sponge or the reset changed r)
  float ssum = (fabsf(u1) + a) * A.inv_dx + (fabsf(v1) + a) * A.inv_dy + (fabsf(w1) + a) * inv_dz;
  // :1338-1356 takes ssum into the maximum if it is finite and positive.  It is a sum of non-negative terms: as an unsigned integer
  // its bit pattern orders like the value, with +inf and the NaNs on top — (bits - 0x7f800000) >> 31 (arithmetic) is all ones exactly
  // for the finite ones; three full-rate integer instructions and one v_max_u32 for two compares, a class test and a select
  {
    const unsigned sb = __float_as_uint(ssum);
    const unsigned keep = (unsigned)((int)(sb - 0x7f800000u) >> 31) & ~(unsigned)((int)sb >> 31);   // finite, sign bit clear
    smax = __uint_as_float(max(__float_as_uint(smax), sb & keep));
  }
  fmx = fmaxf(fmaxf(fmx, r1), fabsf(u1));            // three v_max3_f32 (a balanced tree of fmaxf compiles to six v_max_f32)
  fmx = fmaxf(fmaxf(fmx, fabsf(v1)), fabsf(w1));
  fmx = fmaxf(fmaxf(fmx, p1), ev1);

  E[0] = flog(r1e);
  E[1] = fasinh(u1 * K.inv_u_ref, K.absmask);
  E[2] = fasinh(v1 * K.inv_u_ref, K.absmask);
  E[3] = fasinh(w1 * K.inv_u_ref, K.absmask);
  E[4] = flog(p1e);
  E[5] = flog(fmaxf(ev1, RHO_P_FLOOR));
}

// k_update_z: a wave owns 64 consecutive x of one row and marches a chunk of planes; ZT_Y rows per workgroup.
// The five-plane window of the column (planes z-1 .. z+3) lives in LDS, one private slot per thread and plane
// (ring of five, no barrier: a thread only ever reads what it wrote): in registers it costs 30 VGPRs and 24 moves
// per plane to slide, and pushed the kernel to 148 VGPRs / three waves.
constexpr int ZT_X = 64, ZT_Y = 4, ZNT = ZT_X * ZT_Y;
static_assert(ZT_X == 2 * XT && YT % ZT_Y == 0, "k_update_z's prediction mask: a wave spans two k_flux_xy tiles of one tile row");
typedef float ZRing[5][6][ZNT];
// PART 0: the whole job; 1: the march only — a chunk all of whose planes are predicted is left to k_fill_z; 2: those chunks only (k_fill_z)
template <bool FAST, int PART = 0> __device__ __forceinline__ void update_z_body(const Args &A, ZRing &ring, unsigned bid) {
  const int tid = threadIdx.x;
  const int lx = tid & (ZT_X - 1), ly = tid >> 6;
  const int nby = (A.ny + ZT_Y - 1) / ZT_Y;
  // Workgroup -> (column block, chunk): ROWS fastest, then chunks, then the 64-column blocks.  The columns share nothing (no x / y
  // halo here), so there is no locality to keep — what matters since the uniform-region exits and the predictions is where the
  // EXPENSIVE workgroups go.  The hardware deals consecutive workgroups round the XCDs and, within one, round its CUs: work whose cost
  // is periodic in the workgroup number with a period that divides those counts lands on a fraction of the chip.  With x fastest
  // (8 blocks of 64 columns at 512) the disturbed blocks were always the same residues mod 8: measured on a flow that is uniform but
  // for the sponge columns (2 of 8 blocks expensive), the kernel kept 15 % of its wave slots busy and took 1.31 ms for 0.3 ms of
  // work; with xcd_swizzle (round 5) an XCD owned one 64-plane layer.  Along y the cost varies slowly, so consecutive workgroups
  // cost about the same and the deal is even.
  unsigned b = bid;
  const int by = (int)(b % (unsigned)nby); b /= (unsigned)nby;
  const int bz = (int)(b % (unsigned)A.nzc);
  const int bx = (int)(b / (unsigned)A.nzc);
  const bool second = bz >= A.nzc1;
  const int zc_lo = second ? A.zl_lo2 + (bz - A.nzc1) * A.zchunk : A.zl_lo + bz * A.zchunk;
  const int zc_hi = min(zc_lo + A.zchunk, second ? A.zl_hi2 : A.zl_hi);

  const int x = bx * ZT_X + lx, y = by * ZT_Y + ly;
  const bool in_xy = (x < A.nx) && (y < A.ny);
  const size_t plane_n = (size_t)A.nx * A.ny;
  const size_t col = (size_t)min(y, A.ny - 1) * A.nx + min(x, A.nx - 1);

  const float dt = vreg(A.clk->dt);        // six uses per plane each: an SGPR operand makes a VALU op half rate
  const float inv_dz = vreg(A.inv_dz);
  const float gain = A.clk->gain;
  const Gas G = gas_vgpr(A);
  // (u_ref and 1 / u_ref: three multiplies per plane each, as SGPR operands since round 6 — the prediction's kept state (memoE) needs
  // their registers under the five-wave cap more than those six instructions need the full rate)
  const float uref = A.u_ref;
  const UpdK K = UpdK{G.gm1, G.inv_gm1, G.gamma, A.inv_u_ref, vlit(0x7fffffffu)};

  // Addressing (see gld / gst): scalar field bases at the chunk's first plane + one 32-bit byte offset per lane (with
  // 64-bit lane addresses: 61 half-rate adds per plane); the host keeps zchunk * plane bytes below 2^32.  The field
  // stride is made opaque once per plane so that the 24 field bases are recomputed by the scalar unit instead of living
  // in SGPRs across the loop — those spilled.
  const unsigned plane4 = (unsigned)plane_n << 2;
  const unsigned col4 = (unsigned)col << 2;
  const size_t fs4 = (size_t)A.fstride << 2, ds4 = (size_t)A.dstride << 2;
  const GChar *const qP = (const GChar *)(A.in0 + (size_t)zc_lo * plane_n);              // halo-layout plane zc_lo-3
  const uint8_t *const solP = A.solid + (size_t)zc_lo * plane_n;
  unsigned ws = 0;
  auto load_own = [&](int k, float (&dst)[6]) -> unsigned {   // plane zc_lo-3+k
    const unsigned vo = col4 + (unsigned)k * plane4;
    float e[6];   // the six loads first: fsinh_wave's branches pin a decode behind its own load
#pragma unroll
    for (int m = 0; m < 6; m++) e[m] = *(const GFloat *)(qP + m * fs4 + vo);
    const unsigned sol = solP[vo >> 2] != 0 ? 1u : 0u;
#pragma unroll
    for (int m = 0; m < 6; m++) dst[m] = ZDEC(uref, m, e[m]);
    return sol;
  };

  const bool uex = A.dzero != nullptr;
  const bool zsk = uex && A.z_pred == 2;
  // The chunk's prediction bits for the wave's two tiles (see "Predicted-uniform tiles" below): lane l reads the words of plane zc_lo + l.
  unsigned long long pmask = 0ull, smask = 0ull;
  if (zsk) {
    const int xa = bx * ZT_X, xb = xa + XT;              // the wave's two tiles start here (ZT_X = 2 XT)
    const int zl = zc_lo + lx;                           // this lane's plane of the chunk
    unsigned both = 0u;
    if (zl < zc_hi && xb + XT <= A.nx && y < A.ny) {
      const unsigned *const w = A.dzero + ((size_t)zl * A.dz_nty + (size_t)(y / YT)) * A.dz_ntx + (size_t)(xa / XT);
      both = w[0] & w[1];
    }
    pmask = __builtin_amdgcn_ballot_w64((both & 2u) != 0u);
    smask = __builtin_amdgcn_ballot_w64((both & 4u) != 0u);
  }
  // A chunk ALL of whose planes are predicted for both tiles of the wave (most chunks away from the disturbance): every cell of it
  // holds the state S of its tile's record, and the march below would compute update_cell(S, +0, F, F) at its first plane and store
  // that 64 times.  Done directly — one load of the record, one update, the stores — and in a kernel of its own (k_fill_z: no LDS
  // ring, so not five workgroups per CU), because workgroups are dispatched in order: marched, such a workgroup lives ~30 us on next
  // to no arithmetic (prologue loads, a full first trip, 63 trips of bookkeeping), and even this short form ~12 us (384 stores per
  // lane at 64 in flight) in one of the five slots the expensive workgroups behind it wait for — with most of the grid predicted
  // the chip idled (15 % of the wave slots busy).  In k_update_z such a chunk returns after its flag words.  The z flux difference
  // is F - F = +-0 either way and meets the flagged +0 divergence: the same -(+0) as below, the same bits.
  {
    const int nch = zc_hi - zc_lo;
    const unsigned long long full = nch >= 64 ? ~0ull : ((1ull << nch) - 1ull);
    const bool fullp = zsk && nch <= 64 && (pmask & full) == full && A.send[0] == nullptr;
    if (PART == 2 && !fullp) return;
    if (PART == 1 && fullp) return;
    if (fullp) {   // the new state: k_tile_predict computed it (one update per tile; the planes of a run share it)
      const size_t ti = ((size_t)zc_lo * A.dz_nty + (size_t)(y / YT)) * A.dz_ntx + (size_t)(x / XT);
      const float *const er = A.pref + ti * UREC;
      float E[6];
#pragma unroll
      for (int m = 0; m < 6; m++) E[m] = er[m];
      GChar *const outB = (GChar *)(A.out0 + (size_t)(zc_lo + HALO) * plane_n);
      unsigned vo = col4;
      for (int z = zc_lo; z < zc_hi; z++) {
        size_t f4 = fs4;
        asm volatile("" : "+s"(f4));
        if ((smask & 1ull) == 0ull) {   // (steady: the buffer holds these bits already)
          const unsigned vb = lane_off(vo);
#pragma unroll
          for (int m = 0; m < 6; m++) gst(outB + m * f4, vb, E[m]);
        }
        smask >>= 1;
        if (A.wrap_halo && (z < HALO || z >= A.nzl - HALO)) {
          GChar *const wB = (GChar *)(A.out0 + (size_t)(z < HALO ? z + A.nzl + HALO : z - A.nzl + HALO) * plane_n);
          const unsigned vb = lane_off(col4);
#pragma unroll
          for (int m = 0; m < 6; m++) gst(wB + m * f4, vb, E[m]);
        }
        vo += plane4;
      }
      return;
    }
  }

  // prologue: planes zc_lo-2 .. zc_lo+2 into slots 0 .. 4 (plane zc_lo-3 is only needed here); flux through the low
  // face of plane zc_lo and the left state cell zc_lo contributes to its high face
  float Fz_lo[6], Lz[6];
  {
    float T[6], P[6];
    ws |= load_own(0, T) << 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      ws |= load_own(1 + k, P) << (k + 1);
#pragma unroll
      for (int m = 0; m < 6; m++) ring[k][m][tid] = P[m];
    }
    Prim L, R;
    float lo[6], hi[6];
    int ts = tid;
#pragma unroll
    for (int m = 0; m < 6; m++) {
      const float w0 = ring[0][m][ts], w1 = ring[1][m][ts], w2 = ring[2][m][ts], w3 = ring[3][m][ts], w4 = ring[4][m][ts];
      L.q[m] = weno_cell_side<FAST, true>(T[m], w0, w1, w2, w3);
      weno_cell<FAST>(w0, w1, w2, w3, w4, Lz[m], R.q[m]);
      lo[m] = w1; hi[m] = w2;
      asm volatile("" : "+v"(ts), "+v"(L.q[m]), "+v"(Lz[m]), "+v"(R.q[m]));
    }
    solid_override(L, R, lo, hi, ws, 2);
    prim_floor(L);
    prim_floor(R);
    Cons F = hllc(G, L, R, 2);
#pragma unroll
    for (int m = 0; m < 6; m++) Fz_lo[m] = F.c[m];
  }

  // Uniform-region exits (round 6; see flux_xy_core).  urun = how many of the newest planes in the ring equal their predecessor
  // in EVERY lane of the wave (all six decoded values).  With the four newest comparisons true the five planes of the z stencil
  // hold one state per lane: the reconstruction returns that state exactly (all differences zero: weno_cell's weighted sum is
  // +0), so it is read instead of recomputed.  Two such windows in a row, no solid cell in the stencil and w = 0 exactly (the free
  // stream) make the face's two states equal with no normal velocity: the blended HLLC flux is then (0, 0, 0, p, 0, 0) to the bit
  // (s_M = 0, g = 0, alpha = 0: every coefficient of hllc()'s sum is 0 or 1) — written down instead of evaluated.
  int urun = 0;
  bool prev_wuni = false;
  if (uex) {   // planes zc_lo-1 .. zc_lo+2 against their predecessors (slots 0 .. 4 hold planes zc_lo-2 .. zc_lo+2)
#pragma unroll
    for (int k = 1; k < 5; k++) {
      bool eq = true;
#pragma unroll
      for (int m = 0; m < 6; m++) eq = eq && (__float_as_uint(ring[k][m][tid]) == __float_as_uint(ring[k - 1][m][tid]));
      urun = (__builtin_amdgcn_ballot_w64(!eq) == 0ull) ? urun + 1 : 0;
    }
  }

  float smax = 0.f, fmx = 0.f;
  // Plane z+3 of the first iteration takes over slot 0 (plane zc_lo-2 is done).  From then on every iteration issues the loads
  // of the plane the NEXT one needs right at its top and consumes them (writes them into the ring) at its very END, after its
  // own stores are issued: consumed at the top of the next trip, the wait across the back-edge was a vmcnt(0) that also covered
  // the twelve stores of the trip before (measured: no difference either way — the stores have long completed by then).
  float Nx[6];
  unsigned nsol = load_own(6, Nx);
  if (uex) {   // plane zc_lo+3 against plane zc_lo+2 (slot 4)
    bool eq = true;
#pragma unroll
    for (int m = 0; m < 6; m++) eq = eq && (__float_as_uint(Nx[m]) == __float_as_uint(ring[4][m][tid]));
    urun = (__builtin_amdgcn_ballot_w64(!eq) == 0ull) ? urun + 1 : 0;
  }
#pragma unroll
  for (int m = 0; m < 6; m++) ring[0][m][tid] = Nx[m];
  ws = (ws >> 1) | (nsol << 5);
  int s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 0;   // slots of planes z-1 .. z+3
  // scalar bases at the chunk's first plane; vo = (z - zc_lo) * plane bytes + column bytes
  const GChar *const qN = qP + 7 * (size_t)plane4;                                         // plane z+4
  const uint8_t *const solN = solP + 7 * plane_n;
  const GChar *const inB = (const GChar *)(A.in0 + (size_t)(zc_lo + HALO) * plane_n);      // plane z, halo layout
  GChar *const outB = (GChar *)(A.out0 + (size_t)(zc_lo + HALO) * plane_n);
  const GChar *const dB = (const GChar *)(A.d0 + (size_t)zc_lo * plane_n);                 // plane z, no halo
  unsigned vo = col4;
  // k_flux_xy's "divergence is zero, not stored" flag of this lane's tile (two tiles per wave at most), one word per plane
  // (bit 0: the divergence is zero; bit 1: k_tile_predict's prediction for this step, see below; a scalar base and a 32-bit lane offset)
  unsigned dzo = (unsigned)(((size_t)zc_lo * A.dz_nty + (size_t)(min(y, A.ny - 1) / YT)) * A.dz_ntx + (size_t)(min(x, A.nx - 1) / XT)) << 2;
  const unsigned dz_plane4 = (unsigned)(A.dz_nty * A.dz_ntx) << 2;
  const GChar *const dzB = (const GChar *)A.dzero;
  // Predicted-uniform tiles (k_tile_predict, which ran between k_flux_xy and this kernel): a plane of a predicted tile has one
  // state S in every cell its stencils reach and in its own tile three planes either way, so this trip's window, divergence, face
  // and update are the ones of the trip before if that was a predicted plane too: same operands, same bits.  The first such plane
  // of a run takes the full path and keeps its result; the following ones store that (zskip: wave-uniform — both tiles a wave
  // spans).  And where plane z+1 is predicted, plane z+4 — the one this trip would load, decode and put into the ring slot of plane
  // z-1 — holds S like plane z-1 itself: the slot is right as it is (skipl).  A trip in the middle of a run is six stores.  The
  // predictions of the chunk's planes for the wave's two tiles are read ONCE, lane l the words of plane zc_lo + l: the ballot is the
  // chunk's mask (a flag load per trip is a dependent memory round trip per trip: measured, the skipped trips then cost what the
  // full ones do).
  float memoE[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool memo = false;
  // Steady tiles (bit 2, see k_tile_predict): where the new state of a predicted plane is, bit for bit, the state the plane holds
  // now AND held a step ago, the output buffer — the input of the step before — holds it already: no store (smask: the planes where
  // both tiles of the wave are steady).

  for (int z = zc_lo; z < zc_hi; z++) {
    const bool more = z + 1 < zc_hi;
    size_t f4 = fs4, d4 = ds4;
    asm volatile("" : "+s"(f4), "+s"(d4));
    const bool wpred = (pmask & 1ull) != 0ull;          // this plane and (skipl) the next: predicted in both tiles of the wave
    const bool skipl = more && (pmask & 2ull) != 0ull;
    const bool cand = (smask & 1ull) != 0ull;
    pmask >>= 1; smask >>= 1;
    const bool zskip = wpred && memo;
    // (Every vector-memory result of a trip is consumed inside the trip, late, and a skipped trip issues no load at all: a value
    // loaded at the top and tested at once — the solid byte — or a register a skipped trip overwrites with a constant — the flag
    // word — put an s_waitcnt vmcnt(0) at the top of EVERY trip, and a skipped trip then waited for its own six stores to land.)
    // (Hence no initial values either: a constant written on the skipped path is a write to the register of a load that may be in flight.)
    unsigned dzf;        // this trip's flag word: full trips with the exits on only
    if (uex && !zskip) dzf = __float_as_uint(gld(dzB, dzo));
    unsigned nsb;        // plane z+4's solid byte, looked at where the window slides: trips that load the plane only
    if (more && !skipl) {
#pragma unroll
      for (int m = 0; m < 6; m++) Nx[m] = gld(qN + m * f4, vo);   // encoded: decoded where the plane enters the ring
      nsb = solN[vo >> 2];
    }
    const bool own_solid = (ws >> 2) & 1u;
    const bool wuni = uex && urun >= 4;   // (scalar) planes z-1 .. z+3 hold one state per lane

    float Fz_hi[6];
    float D[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const char *const rb = (const char *)&ring[0][0][0];   // byte offsets: what ds_read takes, the variable's part as its immediate
    auto rd = [&](int a, int m) { return *(const float *)(rb + a + m * (ZNT * 4)); };
    const int t4 = tid * 4;
    // A wave whose 64 columns are inside the body at planes z AND z+1 needs no z face here: both cells either side are copied
    // through (the reference's threads return at once there, :1063-1072), and the left state carried to the next face is
    // discarded there too — plane z+1 is in that face's stencil, so it takes the first-order / mirror form (solid_override).
    const bool face_dead = __builtin_amdgcn_ballot_w64(in_xy && ((ws >> 2) & 3u) != 3u) == 0ull;
    if (zskip) {   // plane z+4 into the ring, nothing else: Lz, Fz_lo and the new state are the last trip's
      if (skipl) urun++;
      else if (more) {
        bool eq = true;
#pragma unroll
        for (int m = 0; m < 6; m++) {
          const float nv = ZDEC(uref, m, Nx[m]);
          eq = eq && (__float_as_uint(nv) == __float_as_uint(rd(s4 * (24 * ZNT) + t4, m)));
          ring[s0][m][tid] = nv;
        }
        urun = (__builtin_amdgcn_ballot_w64(!eq) == 0ull) ? urun + 1 : 0;
      }
    } else
    if (face_dead) {
      if (more && !skipl) {
#pragma unroll
        for (int m = 0; m < 6; m++) ring[s0][m][tid] = ZDEC(uref, m, Nx[m]);
      }
      urun = 0;   // (plane z+4 entered the ring uncompared: the count of equal planes starts again — left alone it ran one plane ahead
                  //  of the ring when the wave came out of the body, and a window with ONE plane that differs passed for uniform)
#pragma unroll
      for (int m = 0; m < 6; m++) Fz_hi[m] = 0.f;
    } else {
      Prim L, R;
#pragma unroll
      for (int m = 0; m < 6; m++) L.q[m] = Lz[m];
      // one variable at a time (see k_flux_xy): the empty asm orders the next variable's LDS reads behind this one.  The
      // five slot addresses are formed once per plane and ride through the asm (tied to one lane index instead, they
      // were re-formed for every variable: 30 half-rate adds)
      int a0 = s0 * (24 * ZNT) + t4, a1 = s1 * (24 * ZNT) + t4, a2 = s2 * (24 * ZNT) + t4, a3 = s3 * (24 * ZNT) + t4,
          a4 = s4 * (24 * ZNT) + t4;
      if (wuni) {
#pragma unroll
        for (int m = 0; m < 6; m++) {   // weno_cell on five equal values, to the bit: every difference is +0, Lhi = fma(+0, 1/6, c), Rlo = fma(+0, -1/6, c)
          const float w2 = rd(a2, m);
          Lz[m] = __builtin_fmaf(0.f, 1.f / 6.f, w2); R.q[m] = w2;   // (c = -0 leaves as +0 on the left, as -0 on the right: the K-selects of hllc() look at signs)
        }
      } else {
#pragma unroll
        for (int m = 0; m < 6; m++) {
          const float w0 = rd(a0, m), w1 = rd(a1, m), w2 = rd(a2, m), w3 = rd(a3, m), w4 = rd(a4, m);
          weno_cell<FAST>(w0, w1, w2, w3, w4, Lz[m], R.q[m]);
          asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(Lz[m]), "+v"(R.q[m]));
        }
      }
      // Latencies behind work: the next plane (loads issued at the top of the trip) has had the reconstruction to arrive
      // and takes over slot s0, which nothing reads any more; its six registers then carry the x/y divergence of THIS
      // plane, fetched behind the z face.  (Loaded right where the update needs it, every trip stalled a full HBM round
      // trip: 3.9 cycles per instruction against 3.1 for the mix.)
      if (skipl) urun++;
      else if (more) {
        bool eq = true;
#pragma unroll
        for (int m = 0; m < 6; m++) {
          const float nv = ZDEC(uref, m, Nx[m]);
          if (uex) eq = eq && (__float_as_uint(nv) == __float_as_uint(rd(a4, m)));     // plane z+4 against plane z+3: BIT patterns
          ring[s0][m][tid] = nv;
        }
        if (uex) urun = (__builtin_amdgcn_ballot_w64(!eq) == 0ull) ? urun + 1 : 0;
      }
      if (in_xy && !own_solid && (!uex || (dzf & 1u) == 0u)) {   // (a flagged tile's divergence is +0 in every cell and was not stored)
        const unsigned vb = lane_off(vo);
#pragma unroll
        for (int m = 0; m < 6; m++) D[m] = gld(dB + m * d4, vb);
      }
      if (ws != 0u) {   // rare: the cells either side of the face, from the ring
        float lo[6], hi[6];
#pragma unroll
        for (int m = 0; m < 6; m++) { lo[m] = rd(s1 * (24 * ZNT) + t4, m); hi[m] = rd(s2 * (24 * ZNT) + t4, m); }
        solid_override(L, R, lo, hi, ws, 2);
      }
      prim_floor(L);
      prim_floor(R);
      // (this window and the one before uniform: L, the left state the last trip left behind, is plane z's value = plane z+1's = R)
      const bool zdirect = wuni && prev_wuni && __builtin_amdgcn_ballot_w64(ws != 0u || !(R.q[IW] == 0.f)) == 0ull;
      if (zdirect) {
#pragma unroll
        for (int m = 0; m < 6; m++) Fz_hi[m] = (m == 3) ? R.q[IP] : 0.f;
      } else {
        Cons F = hllc(G, L, R, 2);
#pragma unroll
        for (int m = 0; m < 6; m++) Fz_hi[m] = F.c[m];
      }
    }
    prev_wuni = wuni;

    if (in_xy) {
      float E[6];          // the cell's new encoded state
      if (zskip) {
#pragma unroll
        for (int m = 0; m < 6; m++) E[m] = memoE[m];
      } else {
        float own[6];        // the cell itself, back from the ring (carried from the reconstruction it cost six registers across the face)
#pragma unroll
        for (int m = 0; m < 6; m++) own[m] = rd(s1 * (24 * ZNT) + t4, m);
        if (own_solid) { // :1063-1072 copy-through
#pragma unroll
          for (int m = 0; m < 6; m++) E[m] = *(const GFloat *)(inB + m * fs4 + vo);   // (rare path: plain addressing)
        } else {
          update_cell(A, K, own, D, Fz_lo, Fz_hi, dt, inv_dz, gain, x, E, smax, fmx);
        }
        if (wpred) {
#pragma unroll
          for (int m = 0; m < 6; m++) memoE[m] = E[m];
        }
      }
      if (!(wpred && cand)) {   // (steady, k_tile_predict: the buffer holds these bits already)
        const unsigned vb = lane_off(vo);
#pragma unroll
        for (int m = 0; m < 6; m++) gst(outB + m * f4, vb, E[m]);
      }
      // Z-slab ring: the first / last three local planes of the NEW state go straight into the packed send buffers
      // (what k_halo_pack would copy afterwards): wave-uniform branch, one dispatch less per step
      auto send_plane = [&](float *buf, int zrel) {
        GChar *const sB = (GChar *)(buf + (size_t)zrel * plane_n);
        const size_t n34 = (size_t)HALO * plane4;
        const unsigned vb = lane_off(col4);
#pragma unroll
        for (int m = 0; m < 6; m++) gst(sB + m * n34, vb, E[m]);
      };
      if (A.send[0]) {
        if (z < HALO) send_plane(A.send[0], z);
        else if (z >= A.nzl - HALO) send_plane(A.send[1], z - (A.nzl - HALO));
      }
      // single periodic domain: the same planes are the NEW state's z halos (what k_halo_periodic would copy before the next
      // step: 37.7 MB and a dependent dispatch); halo layout, local plane z sits at z + HALO
      if (A.wrap_halo) {
        auto wrap_plane = [&](int zh) {
          GChar *const wB = (GChar *)(A.out0 + (size_t)zh * plane_n);
          const unsigned vb = lane_off(col4);
#pragma unroll
          for (int m = 0; m < 6; m++) gst(wB + m * f4, vb, E[m]);
        };
        if (z < HALO) wrap_plane(z + A.nzl + HALO);
        if (z >= A.nzl - HALO) wrap_plane(z - A.nzl + HALO);
      }
    }
    if (!zskip) {
#pragma unroll
      for (int m = 0; m < 6; m++) Fz_lo[m] = Fz_hi[m];
    }
    memo = wpred;
    dzo += dz_plane4;
    vo += plane4;
    if (more) {   // plane z+4 has replaced plane z-1; the window slides
      nsol = 0u;
      if (!skipl) { asm volatile("" : "+v"(nsb)); nsol = nsb != 0u ? 1u : 0u; }
      ws = (ws >> 1) | (nsol << 5);
      const int t = s0; s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = t;
    }
  }

#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    smax = fmaxf(smax, __shfl_xor(smax, o, 64));
    fmx = fmaxf(fmx, __shfl_xor(fmx, o, 64));
  }
  if (lx == 0) {
    tau::atomic_max_float_bits(&A.clk->maxs_bits, smax);
    tau::atomic_max_float_bits(&A.clk->fmax_bits, fmx);
  }
}

#ifdef TAU3D_SPLIT_TU
#ifndef TAU3D_Z_WAVES
#define TAU3D_Z_WAVES 5
#endif
// PART 1: launched with k_fill_z behind it, which takes the fully predicted chunks (a kernel of its own: one body per kernel)
template <bool FAST, bool STRIDE, int PART = 0> __global__ __launch_bounds__(ZNT, TAU3D_Z_WAVES) void k_update_z(const Args A) {   // 5 waves per SIMD: 5 x 30 KB of LDS ring per CU
  __shared__ ZRing ring;
  if (fast_form(A.clk->fmax_in, A.in_fmax) != FAST) return;   // (one weight form per kernel: comment at k_flux_xy)
  if (!STRIDE) { update_z_body<FAST, PART>(A, ring, blockIdx.x); return; }
  const int nbx = (A.nx + ZT_X - 1) / ZT_X, nby = (A.ny + ZT_Y - 1) / ZT_Y;
  const unsigned nb = (unsigned)(nbx * nby * A.nzc);
  for (unsigned b = blockIdx.x; b < nb; b += gridDim.x) update_z_body<FAST>(A, ring, b);   // (a thread reads only its own ring slots: no barrier)
}
// the fully predicted chunks of a step whose full-grid k_update_z ran the expected weight form (else the strided kernel of the other
// form has done the whole job, these chunks included); no reconstruction in here: one kernel for both forms, no LDS
__global__ __launch_bounds__(ZNT) void k_fill_z(const Args A) {
  __shared__ float none[1];
  if (fast_form(A.clk->fmax_in, A.in_fmax) != (A.z_fill == 1)) return;
  update_z_body<true, 2>(A, *reinterpret_cast<ZRing *>(none), blockIdx.x);
}
void launch_update_z(unsigned nwg, hipStream_t s, const Args &A0, bool expect_fast) {
  Args A = A0;
  A.z_fill = A.z_pred == 2 && A.send[0] == nullptr ? (expect_fast ? 1 : 2) : 0;
  const unsigned net = nwg < 1280u ? nwg : 1280u;   // five workgroups per CU resident
#ifdef TAU3D_FAST_ONLY
  hipLaunchKernelGGL((k_update_z<true, false>), dim3(nwg), dim3(ZNT), 0, s, A);
#else
  if (expect_fast) {
    if (A.z_fill) hipLaunchKernelGGL((k_update_z<true, false, 1>), dim3(nwg), dim3(ZNT), 0, s, A);
    else hipLaunchKernelGGL((k_update_z<true, false, 0>), dim3(nwg), dim3(ZNT), 0, s, A);
    hipLaunchKernelGGL((k_update_z<false, true>), dim3(net), dim3(ZNT), 0, s, A);
  } else {
    if (A.z_fill) hipLaunchKernelGGL((k_update_z<false, false, 1>), dim3(nwg), dim3(ZNT), 0, s, A);
    else hipLaunchKernelGGL((k_update_z<false, false, 0>), dim3(nwg), dim3(ZNT), 0, s, A);
    hipLaunchKernelGGL((k_update_z<true, true>), dim3(net), dim3(ZNT), 0, s, A);
  }
  // (behind the march on the same stream: beside it on a stream of its own — it needs 16 VGPRs and no LDS — the pair took as long,
  // 1.28 against 1.25 ms: together they move 6.3 GB, and the HBM is what both wait for)
  if (A.z_fill) hipLaunchKernelGGL(k_fill_z, dim3(nwg), dim3(ZNT), 0, s, A);
#endif
}
}  // namespace h3d — the split-step translation unit ends here
#else
void launch_update_z(unsigned nwg, hipStream_t s, const Args &A, bool expect_fast);   // ZNT threads per workgroup; + k_fill_z where k_tile_predict ran

// ---------------------------------------------------------------- small kernels
__global__ void k_build_solid(uint8_t *solid, Args A) { // :759-770, halo planes included
  size_t n = (size_t)A.nx * A.ny * (A.nzl + 2 * HALO);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int x = (int)(i % A.nx);
    size_t r = i / A.nx;
    int y = (int)(r % A.ny);
    int zh = (int)(r / A.ny);
    int zg = wrapi(A.z0 + zh - HALO, A.nz);
    solid[i] = sdf_solid(A, x, y, zg) ? 1 : 0;
  }
}

// geometry of k_flux_xy's work items and of its flags, as the host needs it (both translation units see the constants)
constexpr int XY_FX = XT, XY_FY = YT;      // flag tile
constexpr int XY_WX = XT, XY_WY = YT;      // work item: one tile
// k_flux_xy's tile flags: one word per (local plane, tile row, tile column), non-zero where the tile or its 3-cell x / y halo
// holds a solid cell — exactly the cells whose solid byte the kernel would look at (ghost columns left of x = 0 and right of
// x = nx-1 by the SDF, as fetch_cell_e classifies them).  One workgroup per flag; runs when the mask is built.
__global__ __launch_bounds__(256) void k_xy_flags(Args A, const uint8_t *solid, unsigned *flags) {

  const int ntx = (A.nx + XT - 1) / XT, nty = (A.ny + YT - 1) / YT;
  unsigned b = blockIdx.x;
  const int bx = (int)(b % (unsigned)ntx); b /= (unsigned)ntx;
  const int by = (int)(b % (unsigned)nty);
  const int z = (int)(b / (unsigned)nty);
  const int zh = z + HALO, zg = wrapi(A.z0 + z, A.nz);
  constexpr int RX = XT + 2 * HALO, RY = YT + 2 * HALO;
  int any = 0, fluid_own = 0;   // fluid_own: a cell of the tile itself (inside the grid) that is NOT solid
  for (int i = threadIdx.x; i < RX * RY; i += 256) {
    const int ix = (i % RX) - HALO, iy = (i / RX) - HALO;
    const int gx = bx * XT + ix, gyu = by * YT + iy;
    const int gy = wrapi(gyu, A.ny);
    bool sol;
    if (gx >= 0 && gx < A.nx) sol = solid[((size_t)zh * A.ny + gy) * A.nx + gx] != 0;
    else sol = sdf_solid(A, gx, gy, zg);
    any |= sol ? 1 : 0;
    if (ix >= 0 && ix < XT && iy >= 0 && iy < YT && gx < A.nx && gyu < A.ny) fluid_own |= sol ? 0 : 1;
  }
  any = __syncthreads_or(any);
  fluid_own = __syncthreads_or(fluid_own);
  // 0: no solid cell in reach; 1: some; 2: every cell of the tile is solid — solid cells are copied through by k_update_z and take
  // no divergence (the reference's threads return at once there, :1063-1072), so k_flux_xy has nothing to produce for the tile
  if (threadIdx.x == 0) flags[blockIdx.x] = !any ? 0u : (fluid_own ? 1u : 2u);
}

// Predicted-uniform tiles (round 6).  k_flux_xy's uniform-region exit makes a uniform tile cheap, not free: upstream of the shock
// ~2/3 of the 512^3 case's 262 144 tiles stage their cells, find them equal and leave, and that launch is bound by the rate
// workgroups are dispatched at (~1 ms for ~180 k of them).  Which tiles CAN only be uniform again is decidable from the flags of
// the step before.  A cell's new state is a function of its own state, the states of the three cells either side of it along each
// axis, the step's dt / inflow gain, and its position (sponge zones, ghost columns, solids).  If tile T's 3 x 3 neighbourhood in
// its plane and T itself in the three planes either side were all flagged uniform WITH THE SAME STATE S at step n (a flag covers
// the tile and its 3-cell x / y halo, and implies no solid cell in reach), then every cell of T and of its halo at step n + 1 had
// nothing but S in its stencil, took +0 as its x/y divergence (the flag) and F - F as its z difference: one arithmetic on one set
// of operands, so one encoded result — provided no such cell lies in a sponge zone (ghost columns and solids are excluded by the
// flags themselves).  k_flux_xy's test of T at step n + 1 would pass, with the new state of T's first cell as the reference: this
// kernel writes the flag the exit would (the "divergence is zero" word is 1 already and stays), k_update_z the state, and the next
// k_flux_xy never sees T: every other tile goes on the list k_flux_xy_list runs over.  It is launched BETWEEN this step's
// k_flux_xy and k_update_z, because the latter gains as much: in a predicted tile the new state is the same in every cell, so a
// lane marching through a run of predicted planes computes it once (update_z_body: zskip).  One thread per tile.
constexpr int PREDICT_NT = 1024;
__global__ __launch_bounds__(PREDICT_NT) void k_tile_predict(const Args A) {
  __shared__ unsigned s_cnt, s_base, s_max[2];
  const int ntx = A.dz_ntx, nty = A.dz_nty;
  const unsigned nt = (unsigned)(ntx * nty * A.nzl);
  const unsigned t = blockIdx.x * (unsigned)PREDICT_NT + threadIdx.x;
  if (threadIdx.x == 0) { s_cnt = 0u; s_max[0] = 0u; s_max[1] = 0u; }
  __syncthreads();
  if (t == 0u) *A.ucount_other = 0u;
  const bool valid = t < nt;
  bool ok = false;
  if (valid) {
    const int bx = (int)(t % (unsigned)ntx);
    const int by = (int)((t / (unsigned)ntx) % (unsigned)nty);
    const int z = (int)(t / (unsigned)(ntx * nty));
    const int x_lo = bx * XT - HALO, x_hi = bx * XT + XT + HALO;   // the cells the flag of T covers: [x_lo, x_hi)
    ok = bx >= 1 && bx <= ntx - 2 && A.xyflag != nullptr && A.xyflag[t] == 0u && (A.uflag_r[t] & UF_ALL) != 0u &&
         (A.sponge_n <= 0 || x_lo >= A.sponge_n) && (A.sponge_out_n <= 0 || x_hi <= A.nx - A.sponge_out_n);
    if (ok) {
      // What the cells of T and its halo read in two steps' reach: T + halo itself (its own flag), six columns of the tiles east and
      // west with their halo rows, six rows of the tiles north and south with their halo columns — a neighbour qualifies whole
      // (UF_ALL) or by the edge strip that faces T (k_flux_xy's UF_W / UF_E / UF_S / UF_N, each with the state of the strip's own
      // corner cell); the corners of the 3-cell box around T lie in those strips' halos, so no diagonal tile is asked.  A cell of
      // a strip of a tile that is NOT uniform took a computed x/y divergence, not the flag's: with one bit pattern in its whole
      // stencil that is (F - F) / dx + (F - F) / dy = +0 as well.
      const uint4 *const rp = reinterpret_cast<const uint4 *>(A.uref_r);
      constexpr int Q = UREC / 4;   // uint4 per record
      const uint4 c0 = rp[(size_t)t * Q], c1 = rp[(size_t)t * Q + 1];
      // The ten neighbours: four in the plane (with the strip that faces T and where its state sits in the record), T itself in
      // the six planes around.  All flags first, then all records, then the verdict: asked one after the other with a short
      // circuit they were twenty dependent loads.
      const int ym = by == 0 ? nty - 1 : by - 1, yp = by == nty - 1 ? 0 : by + 1;
      const unsigned pl = (unsigned)(z * nty) * (unsigned)ntx;
      unsigned nb[10], strip[10], fl[10];
      int w4[10];
      nb[0] = pl + (unsigned)(by * ntx + bx + 1); strip[0] = UF_W; w4[0] = 0;
      nb[1] = pl + (unsigned)(by * ntx + bx - 1); strip[1] = UF_E; w4[1] = 2;
      nb[2] = pl + (unsigned)(yp * ntx + bx);     strip[2] = UF_S; w4[2] = 0;
      nb[3] = pl + (unsigned)(ym * ntx + bx);     strip[3] = UF_N; w4[3] = 4;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int dz = k < HALO ? k - HALO : k - HALO + 1;
        int zz = z + dz;
        if (zz < 0 || zz >= A.nzl) {   // whole periodic domain in the handle: wrap; a slab: the neighbour rank's plane, flags unknown
          if (A.nzl != A.nz) { ok = false; zz = z; }
          else zz += zz < 0 ? A.nzl : -A.nzl;
        }
        nb[4 + k] = (unsigned)(zz * nty + by) * (unsigned)ntx + (unsigned)bx; strip[4 + k] = 0u; w4[4 + k] = 0;
      }
#pragma unroll
      for (int k = 0; k < 10; k++) fl[k] = A.uflag_r[nb[k]];
      uint4 a0[10], a1[10];
#pragma unroll
      for (int k = 0; k < 10; k++) {
        const int w = (fl[k] & UF_ALL) ? 0 : w4[k];   // (a record word that is not meaningful is compared all the same: the flag decides)
        a0[k] = rp[(size_t)nb[k] * Q + w]; a1[k] = rp[(size_t)nb[k] * Q + w + 1];
      }
      unsigned bad = 0u;
#pragma unroll
      for (int k = 0; k < 10; k++)
        bad |= ((fl[k] & (UF_ALL | strip[k])) == 0u ? 1u : 0u) | (a0[k].x ^ c0.x) | (a0[k].y ^ c0.y) | (a0[k].z ^ c0.z) | (a0[k].w ^ c0.w) |
               (a1[k].x ^ c1.x) | (a1[k].y ^ c1.y);
      ok = ok && bad == 0u;
    }
    // (bit 0 of the tile's "divergence is zero" word is set already — the tile was flagged this step — and stays: the next k_flux_xy
    //  does not come here.)
  }
  // The NEW state of a predicted tile, once per tile: update_cell on S with the flagged +0 divergence and a z flux difference of
  // F - F — what k_update_z computes in every cell of it (the difference is +-0 there and meets +0: the same -(+0)).  It goes into
  // the tile's record for the next step (what k_flux_xy's test would take as its reference) and k_update_z / k_fill_z store it in the
  // cells.  Steady tiles: predicted now AND by the step before (its flag carries UF_PRED: not a k_flux_xy's), the state then (the
  // record this one replaces) = the state now = the new state, bit for bit: the buffer k_update_z is about to write is the input of
  // step n - 1 and holds those bits in every cell of the tile — no store (bit 2 of the dzero word).
  {
    float smax = 0.f, fmx = 0.f;
    if (ok) {
      const float *const sr = A.uref_r + (size_t)t * UREC;
      float S6[6], own[6], E[6], Z6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 6; m++) S6[m] = sr[m];
      const Gas G = gas_vgpr(A);
      const UpdK K = UpdK{G.gm1, G.inv_gm1, G.gamma, A.inv_u_ref, vlit(0x7fffffffu)};
#pragma unroll
      for (int m = 0; m < 6; m++) own[m] = ZDEC(A.u_ref, m, S6[m]);
      const int xt = (int)(t % (unsigned)ntx) * XT;   // (any column of the tile: none of them is in a sponge zone)
      update_cell(A, K, own, Z6, Z6, Z6, A.clk->dt, A.inv_dz, A.clk->gain, xt, E, smax, fmx);
      bool steady = A.pred_commit && (A.uflag_r[t] & UF_PRED) != 0u;
      float *const pr = A.pref + (size_t)t * UREC;
#pragma unroll
      for (int m = 0; m < 6; m++) {
        steady = steady && __float_as_uint(pr[m]) == __float_as_uint(S6[m]) && __float_as_uint(E[m]) == __float_as_uint(S6[m]);
        pr[m] = E[m];
      }
      A.pflag[t] = UF_ALL | UF_PRED;
      A.dzero[t] = steady ? 7u : 3u;   // bit 1: k_update_z's copy of the prediction (a k_flux_xy that runs the tile writes 0 or 1)
    } else if (valid) {
      A.pflag[t] = 0u;
      A.dzero[t] = A.dzero[t] & 1u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      smax = fmaxf(smax, __shfl_xor(smax, o, 64));
      fmx = fmaxf(fmx, __shfl_xor(fmx, o, 64));
    }
    // (per workgroup first: the predicted tiles of a step mostly share ONE state, so every wave arrives with the same maximum at the
    // same moment and 4 096 of them raced to raise the two words — same-address atomics, ~12 ns each, 25 us of the kernel's 45)
    if (__lane_id() == 0) { atomicMax(&s_max[0], __float_as_uint(smax)); atomicMax(&s_max[1], __float_as_uint(fmx)); }   // (non-negative floats order like their bits)
  }
  // the others: one LDS atomic per wave, one global atomic per workgroup (same-address atomics cost ~12 ns each at the L2: a launch of
  // 4096 waves would spend 50 us on them); a wave's tiles in ascending order
  const unsigned long long need = __builtin_amdgcn_ballot_w64(valid && !ok);
  const unsigned lane = __lane_id();
  unsigned wbase = 0u;
  if (need != 0ull) {
    if (lane == (unsigned)__builtin_ctzll(need)) wbase = atomicAdd(&s_cnt, (unsigned)__builtin_popcountll(need));
    wbase = (unsigned)__builtin_amdgcn_readlane((int)wbase, __builtin_ctzll(need));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_base = s_cnt != 0u ? atomicAdd(A.ucount, s_cnt) : 0u;
    tau::atomic_max_float_bits(&A.clk->maxs_bits, __uint_as_float(s_max[0]));
    tau::atomic_max_float_bits(&A.clk->fmax_bits, __uint_as_float(s_max[1]));
  }
  __syncthreads();
  if (valid && !ok) A.ulist[s_base + wbase + (unsigned)__builtin_popcountll(need & ((1ull << lane) - 1ull))] = t;
}
// the verifying mode (TAU3D_TILE_LIST=2): the step after a prediction ran every tile all the same; each predicted tile must have
// come out flagged, with the predicted state.  Counts the ones that did not.
__global__ __launch_bounds__(256) void k_tile_predict_check(const unsigned *pflag, const float *pref, const unsigned *uflag, const float *uref,
                                                            unsigned nt, unsigned *bad, unsigned *npred) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nt || pflag[t] == 0u) return;
  atomicAdd(npred, 1u);
  bool same = (uflag[t] & UF_ALL) != 0u;
  for (int m = 0; m < 6 && same; m++) same = __float_as_uint(pref[(size_t)t * UREC + m]) == __float_as_uint(uref[(size_t)t * UREC + m]);
  if (!same) atomicAdd(bad, 1u);
}

struct InitVals { float f[6]; float s[6]; }; // encoded fluid / solid cell values (host-computed, libm)
__global__ void k_init(Args A, float *const st0, float *const st1, float *const st2, float *const st3,
                       float *const st4, float *const st5, const uint8_t *solid, InitVals iv) { // :939-985
  size_t n = (size_t)A.nx * A.ny * (A.nzl + 2 * HALO);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float *v = solid[i] ? iv.s : iv.f;
    st0[i] = v[0]; st1[i] = v[1]; st2[i] = v[2]; st3[i] = v[3]; st4[i] = v[4]; st5[i] = v[5];
  }
}

// periodic halo of a single domain: planes [nzl, nzl+3) -> low halo, planes [3, 6) -> high halo
// what the step about to run reads is what the last one wrote (or what init / upload measured, k_field_max)
__device__ __forceinline__ void field_max_commit(DevClock *c) {
  const float was = c->fmax_in, now = __uint_as_float(c->fmax_bits);
  c->form_flip = ((was <= W_FLIM) != (now <= W_FLIM)) ? 1u : 0u;   // (!(NaN <= x): a NaN range is "beyond the limit" on both sides)
  c->fmax_in = now;
  c->fmax_bits = 0u;
}
__device__ __forceinline__ void clock_begin(DevClock *c) { // log-time clock, :1680-1683 — before the step
  c->t *= expf(c->d_tau);
  c->dt = c->t * c->d_tau;
  float ramp = c->t / 0.02f;
  c->gain = fminf(fmaxf(ramp, 0.f), 1.f);
  c->maxs_bits = 0u;
  field_max_commit(c);
}
__device__ __forceinline__ void clock_end(DevClock *c) { // d_tau controller, :1697-1704 — after the step (and the max all-reduce)
  float maxs = __uint_as_float(c->maxs_bits);
  float dt_cfl = c->cfl / fmaxf(maxs, 1e-9f);
  if (c->dt > 1.10f * dt_cfl) c->d_tau *= 0.80f;
  else if (c->dt < 0.85f * dt_cfl) c->d_tau *= 1.10f;
  c->d_tau = fminf(fmaxf(c->d_tau, 1e-7f), 5e-2f);
  c->maxs_last = maxs;
  c->step += 1;
}
// The single-domain step loop folds the two 1-thread clock kernels into the halo copy that precedes every k_step
// (controller of the step before, then the clock of this one): two dependent dispatches per step instead of four,
// which is what a 64^3 run is made of (clk == nullptr: plain halo copy).
struct HaloArgs { float *f[6]; size_t plane_n; int nzl; DevClock *clk; int do_end; int copy; };
__global__ void k_halo_periodic(HaloArgs H) {
  if (H.clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (H.do_end) clock_end(H.clk);
    clock_begin(H.clk);
  }
  if (!H.copy) return;          // the halos are in place (k_update_z wrote them): the clock only
  const size_t n4 = (size_t)HALO * H.plane_n; // floats per halo block
  const int f = blockIdx.y >> 1, side = blockIdx.y & 1;
  float *base = H.f[f];
  const float *src = side == 0 ? base + (size_t)H.nzl * H.plane_n : base + (size_t)HALO * H.plane_n;
  float *dst = side == 0 ? base : base + (size_t)(H.nzl + HALO) * H.plane_n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

// packed halo exchange: boundary planes of 6 fields <-> one contiguous buffer per side
struct PackArgs { float *f[6]; float *buf[2]; size_t plane_n; int nzl; DevClock *clk; int do_end; };   // clk: also run the controller / clock (tau3d_slab_begin_async)
// dir 0: pack (send side s <- first / last 3 interior planes); dir 1: unpack (recv side s -> halo planes)
__global__ void k_halo_pack(PackArgs P, int dir) {
  if (P.clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // as k_halo_periodic does for the single domain
    if (P.do_end) clock_end(P.clk);
    clock_begin(P.clk);
  }
  const size_t n3 = (size_t)HALO * P.plane_n;
  const int f = blockIdx.y >> 1, side = blockIdx.y & 1;
  float *fld = P.f[f];
  float *b = P.buf[side] + (size_t)f * n3;
  float *planes = dir == 0 ? (side == 0 ? fld + (size_t)HALO * P.plane_n : fld + (size_t)P.nzl * P.plane_n)
                           : (side == 0 ? fld : fld + (size_t)(P.nzl + HALO) * P.plane_n);
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n3; i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n3) {
      if (dir == 0) *reinterpret_cast<float4 *>(b + i) = *reinterpret_cast<const float4 *>(planes + i);
      else *reinterpret_cast<float4 *>(planes + i) = *reinterpret_cast<const float4 *>(b + i);
    } else {
      for (size_t k = i; k < n3; k++) {
        if (dir == 0) b[k] = planes[k];
        else planes[k] = b[k];
      }
    }
  }
}

__global__ void k_clock_begin(DevClock *c) { clock_begin(c); }
__global__ void k_clock_turn(DevClock *c, int do_end) { if (do_end) clock_end(c); clock_begin(c); }   // what k_halo_pack's first thread does, alone
__global__ void k_clock_end(DevClock *c) { clock_end(c); }
__global__ void k_clock_set_explicit(DevClock *c, float dt, float gain) {
  c->dt = dt; c->gain = gain; c->maxs_bits = 0u;
  field_max_commit(c);
}
// ---- small grids.  A 64^3 step is two dependent dispatches (halo copy + clock, k_step): 38.7 us, of which a 32^3 step — almost
// no work — already costs 26.5.  Measured in round 3 and NOT kept: a hipGraph of the same launches replays at the
// same pace (37.7 us: the latency is on the device, between dependent dispatches and inside k_step's own chain of loads and
// barriers); the whole step loop as ONE resident cooperative grid (periodic z wrap instead of the halo copy, a clock replica
// per workgroup instead of the clock kernel, triple-buffered max words, one grid barrier per step) was bit-identical with
// cooperative_groups::grid.sync() but slower at every size — 29.9 / 61.1 / 86.4 / 170 us per step at 32^3 / 48^3 / 64^3 / 96^3
// against 26.5 / 37.0 / 38.6 / 93.4 — and no faster with a hand-written arrival-counter barrier (28.8 / 53.6 / 80.9 / 144).

// largest |primitive| of planes [zh_lo, zh_hi) of the halo layout, folded into fmax_bits (after init / upload)
__global__ __launch_bounds__(256) void k_field_max(Args A, int zh_lo, int zh_hi) {
  __shared__ float red[4];
  const size_t n0 = (size_t)A.nx * A.ny * zh_lo, n1 = (size_t)A.nx * A.ny * zh_hi;
  float m = 0.f;
  for (size_t i = n0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (size_t)gridDim.x * blockDim.x) {
    const Prim q = decode(A, i);
    float c = fmaxf(fmaxf(fmaxf(fabsf(q.q[0]), fabsf(q.q[1])), fabsf(q.q[2])),
                    fmaxf(fmaxf(fabsf(q.q[3]), fabsf(q.q[4])), fabsf(q.q[5])));
    if (!(c <= 3.0e38f)) c = __builtin_inff();   // NaN / inf cells: no claim about the range
    m = fmaxf(m, c);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    tau::atomic_max_float_bits(&A.clk->fmax_bits, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}


// ---------------------------------------------------------------- visualisation fields
// k_vis, tau_hypersonic_3d_cuda.cu:800-905: one scalar per cell from the primitives of the cell and its six
// neighbours (prim_at_xbc, :724-751: inflow ghost left, transmissive ghost right, y/z periodic, solid
// neighbours at the wall state).  One thread per cell, x fastest (a wave reads 64 consecutive floats of each
// field; the y/z neighbours are the same lines again from L2).  MODE is a template parameter so that each
// variant decodes only the fields it uses (the point modes read 1-3 arrays, the gradient modes all but Ev).
__device__ __forceinline__ void apply_wall(const Args &A, Prim &q) { // :511-521
  const float p_keep = fmaxf(q.q[IP], RHO_P_FLOOR);
  q.q[IU] = 0.f; q.q[IV] = 0.f; q.q[IW] = 0.f;
  q.q[IP] = p_keep;
  q.q[IR] = fmaxf(p_keep / (A.R * fmaxf(A.Twall, NEWTON_TEMP_FLOOR)), RHO_P_FLOOR);
  q.q[IE] = evib_eq(A, A.Twall);
}
__device__ __forceinline__ Prim prim_at_xbc(const Args &A, int x, int y, int zh, int zg) {
  y = wrapi(y, A.ny);
  Prim q;
  if (x < 0) {
    q = inflow_prim(A);
    if (sdf_solid(A, x, y, zg)) apply_wall(A, q);
    return q;
  }
  if (x >= A.nx) return outflow_prim(A, decode(A, ((size_t)zh * A.ny + y) * A.nx + (A.nx - 1)));
  const size_t gi = ((size_t)zh * A.ny + y) * A.nx + x;
  q = decode(A, gi);
  if (A.solid[gi]) apply_wall(A, q);
  return q;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_vis(const Args A, float *__restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int zl = blockIdx.z;
  if (x >= A.nx || y >= A.ny) return;
  const int zh = zl + HALO;
  const size_t oi = ((size_t)zl * A.ny + y) * A.nx + x;
  if (A.solid[((size_t)zh * A.ny + y) * A.nx + x]) { out[oi] = 0.f; return; }
  const int zg = wrapi(A.z0 + zl, A.nz);
  const Prim q0 = prim_at_xbc(A, x, y, zh, zg);
  if (MODE == 1) { out[oi] = logf(1.0f + fmaxf(q0.q[IR], 0.0f)); return; }
  if (MODE == 2) { out[oi] = logf(1.0f + fmaxf(q0.q[IP], 0.0f)); return; }
  const float sp = sqrtf(q0.q[IU] * q0.q[IU] + q0.q[IV] * q0.q[IV] + q0.q[IW] * q0.q[IW]);
  if (MODE == 3) { out[oi] = sp; return; }
  if (MODE == 4) { out[oi] = sp / fmaxf(sqrtf(fmaxf(A.gamma * q0.q[IP] / q0.q[IR], DENOM_EPS)), DENOM_EPS); return; }
  const Prim qxm = prim_at_xbc(A, x - 1, y, zh, zg), qxp = prim_at_xbc(A, x + 1, y, zh, zg);
  const Prim qym = prim_at_xbc(A, x, y - 1, zh, zg), qyp = prim_at_xbc(A, x, y + 1, zh, zg);
  const Prim qzm = prim_at_xbc(A, x, y, zh - 1, wrapi(A.z0 + zl - 1, A.nz));
  const Prim qzp = prim_at_xbc(A, x, y, zh + 1, wrapi(A.z0 + zl + 1, A.nz));
  const float inv2dx = 0.5f / A.dx, inv2dy = 0.5f / A.dy, inv2dz = 0.5f / A.dz;
  if (MODE == 0) {
    const float drdx = (qxp.q[IR] - qxm.q[IR]) * inv2dx, drdy = (qyp.q[IR] - qym.q[IR]) * inv2dy;
    const float drdz = (qzp.q[IR] - qzm.q[IR]) * inv2dz;
    out[oi] = sqrtf(drdx * drdx + drdy * drdy + drdz * drdz);
    return;
  }
  const float dudx = (qxp.q[IU] - qxm.q[IU]) * inv2dx, dudy = (qyp.q[IU] - qym.q[IU]) * inv2dy, dudz = (qzp.q[IU] - qzm.q[IU]) * inv2dz;
  const float dvdx = (qxp.q[IV] - qxm.q[IV]) * inv2dx, dvdy = (qyp.q[IV] - qym.q[IV]) * inv2dy, dvdz = (qzp.q[IV] - qzm.q[IV]) * inv2dz;
  const float dwdx = (qxp.q[IW] - qxm.q[IW]) * inv2dx, dwdy = (qyp.q[IW] - qym.q[IW]) * inv2dy, dwdz = (qzp.q[IW] - qzm.q[IW]) * inv2dz;
  if (MODE == 6) { out[oi] = dudx + dvdy + dwdz; return; }
  if (MODE == 5) {
    const float wx = dwdy - dvdz, wy = dudz - dwdx, wz = dvdx - dudy;
    out[oi] = sqrtf(wx * wx + wy * wy + wz * wz);
    return;
  }
  // Q = (||Omega||^2 - ||S||^2) / 2, :881-899
  const float O12 = 0.5f * (dudy - dvdx), O13 = 0.5f * (dudz - dwdx), O23 = 0.5f * (dvdz - dwdy);
  const float Om2 = 2.0f * (O12 * O12 + O13 * O13 + O23 * O23);
  const float S12 = 0.5f * (dudy + dvdx), S13 = 0.5f * (dudz + dwdx), S23 = 0.5f * (dvdz + dwdy);
  const float Sm2 = (dudx * dudx + dvdy * dvdy + dwdz * dwdz) + 2.0f * (S12 * S12 + S13 * S13 + S23 * S23);
  out[oi] = 0.5f * (Om2 - Sm2);
}

// slice_to_rgba, :1416-1442, as two passes over one plane of the vis volume: min/max (order-preserving
// integer keys so that atomicMin/Max work on signed floats), then the grey ramp with t^2 opacity.
__device__ __forceinline__ unsigned fkey(float v) { unsigned b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float funkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ float safe_log1pf(float x) { return logf(1.0f + fmaxf(x, 0.0f)); }

__global__ __launch_bounds__(256) void k_slice_minmax(const float *__restrict__ s, int n, int log_scale, unsigned *mm) {
  float mn = 1e30f, mx = -1e30f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float v = s[i];
    v = log_scale ? safe_log1pf(v) : v;
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], fkey(mn)); atomicMax(&mm[1], fkey(mx)); }
}
__global__ __launch_bounds__(256) void k_slice_rgba(const float *__restrict__ s, int n, int log_scale, float a_gain,
                                                    const unsigned *mm, uint32_t *__restrict__ dst) {
  const float mn = funkey(mm[0]), mx = funkey(mm[1]);
  const float inv = 1.0f / fmaxf(mx - mn, 1e-20f);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float v = s[i];
    v = log_scale ? safe_log1pf(v) : v;
    const float t = clampf((v - mn) * inv, 0.f, 1.f);
    const float a = clampf(a_gain * (t * t), 0.f, 1.f);
    const uint32_t c = (uint32_t)(unsigned char)(t * 255.0f), al = (uint32_t)(unsigned char)(a * 255.0f);
    dst[i] = (al << 24) | (c << 16) | (c << 8) | c;
  }
}

// th3cs.cu:1199-1222 — the export path of the headless program: a whole visualisation volume to 8-bit palette
// indices, (int)(pow((v - min) / max(max - min, 1e-12), gamma) * 255) clamped to 0..255.  The reference does the
// min / max and the map on the host over a downloaded float volume; here the volume never leaves the device and one
// byte per voxel comes back.
__global__ __launch_bounds__(256) void k_volume_minmax(const float *__restrict__ s, size_t n, unsigned *mm) {
  float mn = 1e30f, mx = -1e30f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = s[i];
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], fkey(mn)); atomicMax(&mm[1], fkey(mx)); }
}
__global__ __launch_bounds__(256) void k_palette_index(const float *__restrict__ s, size_t n, float gamma,
                                                       const unsigned *mm, uint8_t *__restrict__ dst) {
  const float mn = funkey(mm[0]), mx = funkey(mm[1]);
  const float range = fmaxf(mx - mn, 1e-12f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float norm = powf((s[i] - mn) / range, gamma);
    int p = (int)(norm * 255.0f);
    p = p < 0 ? 0 : (p > 255 ? 255 : p);
    dst[i] = (uint8_t)p;
  }
}

// k_outflow_reflection_metric, :1389-1408: max |p - p_inflow| over the last nprobe columns
__global__ __launch_bounds__(256) void k_outflow_reflection(const Args A, int x0, unsigned *out_bits) {
  const int ncol = A.nx - x0;
  const size_t n = (size_t)ncol * A.ny * A.nzl;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int x = x0 + (int)(i % ncol);
    const size_t r = i / ncol;
    const int y = (int)(r % A.ny), zl = (int)(r / A.ny);
    const float p = fexp(A.in[4][((size_t)(zl + HALO) * A.ny + y) * A.nx + x]);
    m = fmaxf(m, fabsf(p - A.in_p));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) tau::atomic_max_float_bits(out_bits, m);
}

} // namespace h3d

// =====================================================================================
// C-ABI
// =====================================================================================
struct tau3d {
  tau3d_params p;
  int z0, nzl, device;
  hipStream_t stream;
  bool own_stream;
  size_t plane_n, field_n;  // floats per plane, floats per field incl. halo
  size_t field_stride, dxy_stride;   // floats between consecutive fields of one allocation (state / cache; x/y divergence)
  float *buf[2][6];         // ping-pong, halo layout
  float *dxy[6];            // split step: x/y flux divergence of the local planes
  bool split;               // step = k_flux_xy + k_update_z (else the fused k_step)
  uint8_t *solid;
  unsigned *xyflag = nullptr;   // split step: k_flux_xy's solid-free tile flags (h3d::k_xy_flags)
  unsigned *dzero = nullptr;    // split step, uniform-region exits: k_flux_xy's "this tile's divergence is zero" flags (h3d::Args::dzero)
  bool debug_no_xy_fix = false; // TAU3D_DEBUG_NO_XY_FIX at tau3d_create (tests): tau3d_slab_xy_fix_async does nothing
  bool no_wrap = false;         // TAU3D_NO_WRAP at tau3d_create: k_update_z does not write the new state's periodic z halos (k_halo_periodic copies them)
  bool uniform_exits = true;    // TAU3D_UNIFORM_EXITS=0 (read at tau3d_create): every tile and every plane takes the full path
  // predicted-uniform tiles (h3d::k_tile_predict): TAU3D_TILE_LIST (read at tau3d_create) 0: off, 1 (default): k_flux_xy runs over the list
  // of the tiles that could not be predicted, 2: predictions are made and CHECKED against a k_flux_xy over every tile (tests)
  int tile_list = 1;
  bool z_skip = true;              // TAU3D_Z_SKIP=0 (read at tau3d_create): k_update_z does not use the predictions
  unsigned list_margin_div = 8;   // k_flux_xy_list's grid: the last length seen + 1 / this of it + 512
  unsigned *uflag[2] = {nullptr, nullptr};   // per tile: found (or predicted) uniform; [step parity]
  float *uref[2] = {nullptr, nullptr};       // ... and with which encoded states (h3d::UREC floats per tile: see h3d::Args::uflag_w)
  unsigned *vflag = nullptr; float *vref = nullptr;   // mode 2: the predictions
  unsigned *ulist = nullptr;                 // tile indices
  unsigned *ucount = nullptr;                // [0], [1]: list length by step parity; [2]: mode 2's mismatches; [3]: mode 2's predictions checked
  unsigned *ucount_host = nullptr;           // mapped host word: the length the last k_flux_xy_list to run saw
  unsigned *ucount_host_dev = nullptr;       // ... as the device addresses it
  hipEvent_t list_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // recorded after each whole-domain step of a handle that keeps the list (see split_xy)
  unsigned long list_step = 0;               // such steps issued
  int upar = 0;                              // this step's parity
  bool slab_can_list = false;                // tau3d_slab_xy_async -> tau3d_slab_z_async of the same step
  bool list_ok = false;                      // the list (mode 2: the prediction) made after the last step describes buf[list_cur] ...
  int list_cur = 0;                          // ... and nothing wrote the state since
  h3d::DevClock *clk;
  int cur;                  // which side holds the current state
  bool end_pending;         // the controller update of the last tau3d_step_async step has not run yet
  bool halo_fresh = false;  // the z halos of the current state are in place (written by the k_update_z of a tau3d_step_async step)
  bool wrap_now = false;    // the step being issued is such a step
  h3d::Args base;           // constants, pointers filled per launch
  int zchunk;
  float *xbuf[2][2];        // [kind: 0 send, 1 recv][side]: packed 6 x 3 planes
  bool expect_fast = true;  // which WENO weight form the host expects the next split step to take (its last look at the clock block;
                            // both forms are launched, the device decides: k_flux_xy)
  bool direct = false;      // Z-slab ring with direct halos (tau3d_set_halo_direct): neighbours write this slab's halo planes themselves
  uint8_t *pidx = nullptr;   // palette indices of the last tau3d_palette_indices
  float *vis;               // nx*ny*nzl scalar field of the last tau3d_vis (lazy)
  uint32_t *rgba;           // one slice of pixels (lazy)
  unsigned *scratch;        // 4 words: slice min/max keys, reflection metric bits
  // optional per-launch event timing
  bool timing;
  int n_ev;
  double ev_cells;
  hipEvent_t ev0[4096], ev1[4096], evm[4096];   // interval start / end, and (split step) the point between its two kernels
  bool evm_set[4096];
  bool ev_made;
};

static int flush_clock(tau3d *h);
static int split_buffers(tau3d *h);
// the solid mask of the slab (+ halo planes) and, for the split step, the per-tile flags derived from it
static int build_solid(tau3d *h) {
  hipLaunchKernelGGL(h3d::k_build_solid, dim3(1024), dim3(256), 0, h->stream, h->solid, h->base);
  TAU_LAUNCH_CHECK("k_build_solid");
  if (h->xyflag) {
    const int ntx = (h->p.nx + h3d::XY_FX - 1) / h3d::XY_FX, nty = (h->p.ny + h3d::XY_FY - 1) / h3d::XY_FY;
    hipLaunchKernelGGL(h3d::k_xy_flags, dim3((unsigned)(ntx * nty * h->nzl)), dim3(256), 0, h->stream, h->base, (const uint8_t *)h->solid, h->xyflag);
    TAU_LAUNCH_CHECK("k_xy_flags");
  }
  return 0;
}

static float host_evib_eq(const tau3d_params &P, float T) { // tau_hypersonic_3d_cuda.cu:206-211
  float a = P.theta_v / fmaxf(T, 1e-6f);
  float ea = expf(a);
  float denom = fmaxf(ea - 1.f, 1e-6f);
  return (P.R * P.theta_v) / denom;
}
static float host_asinh(float x) { // :121-125
  float ax = fabsf(x);
  float t = logf(ax + sqrtf(ax * ax + 1.0f));
  return copysignf(t, x);
}

extern "C" void tau3d_params_default(tau3d_params *hp, int nx, int ny, int nz) {
  hp->nx = nx; hp->ny = ny; hp->nz = nz;
  hp->dx = 1.f / nx; hp->dy = 1.f / ny; hp->dz = 1.f / nz;
  hp->cfl = 0.3333f; hp->u_ref = 10.f; hp->R = 10.f; hp->gamma_floor = 1.1f;
  hp->Twall = 0.02f; hp->tau_vib = 2e-4f; hp->theta_v = 0.2f;
  hp->sdf_cx = 0.5f; hp->sdf_cy = 0.5f; hp->sdf_cz = 0.5f; hp->sdf_r = 0.25f;
  hp->inflow_r = 0.02f; hp->inflow_p = 0.02f;
  hp->inflow_u = 100.0f; hp->inflow_v = 0.0f; hp->inflow_w = 0.0f;
  hp->sponge_n = 24; hp->sponge_strength = 0.05f;
  hp->sponge_out_n = 24; hp->sponge_out_strength = 0.05f;
}

static void fill_consts(tau3d *h) {
  const tau3d_params &P = h->p;
  h3d::Args &A = h->base;
  memset(&A, 0, sizeof(A));
  A.nx = P.nx; A.ny = P.ny; A.nz = P.nz; A.nzl = h->nzl; A.z0 = h->z0;
  A.dx = P.dx; A.dy = P.dy; A.dz = P.dz;
  A.inv_dx = 1.f / P.dx; A.inv_dy = 1.f / P.dy; A.inv_dz = 1.f / P.dz;
  A.u_ref = P.u_ref; A.inv_u_ref = 1.f / P.u_ref; A.R = P.R; A.gamma = P.gamma_floor;
  A.gm1 = P.gamma_floor - 1.f; A.inv_gm1 = 1.f / A.gm1; A.Twall = P.Twall; A.theta_v = P.theta_v; A.Rtheta = P.R * P.theta_v;
  A.inv_tau_vib = 1.f / fmaxf(P.tau_vib, 1e-9f);
  A.sdf_cx = P.sdf_cx; A.sdf_cy = P.sdf_cy; A.sdf_cz = P.sdf_cz; A.sdf_r = P.sdf_r;
  A.in_r = fmaxf(P.inflow_r, 1e-30f); A.in_p = fmaxf(P.inflow_p, 1e-30f);
  A.in_u = P.inflow_u; A.in_v = P.inflow_v; A.in_w = P.inflow_w;
  A.in_ev = host_evib_eq(P, A.in_p / (A.in_r * P.R));
  A.in_fmax = fmaxf(fmaxf(fmaxf(fabsf(A.in_r), fabsf(A.in_u)), fmaxf(fabsf(A.in_v), fabsf(A.in_w))), fmaxf(fabsf(A.in_p), fabsf(A.in_ev)));
  if (!(A.in_fmax <= 3.0e38f)) A.in_fmax = INFINITY;
  if (const char *e = getenv("TAU3D_WENO_RCP")) { if (atoi(e) != 0) A.in_fmax = INFINITY; }   // force the reciprocal form
  A.sponge_n = P.sponge_n > 0 ? P.sponge_n : 0; A.sponge_out_n = P.sponge_out_n > 0 ? P.sponge_out_n : 0;
  A.sponge_strength = P.sponge_strength; A.sponge_out_strength = P.sponge_out_strength;
  A.solid = h->solid; A.clk = h->clk;
  A.ntx = (P.nx + h3d::TX - 1) / h3d::TX; A.nty = (P.ny + h3d::TY - 1) / h3d::TY;
}

extern "C" int tau3d_create(tau3d_t **out, const tau3d_params *p, int z0, int nzl, int device, void *stream) {
  if (!out || !p) return tau::fail("tau3d_create: null argument");
  if (p->nx < 8 || p->ny < 8 || p->nz < 8) return tau::fail("tau3d_create: grid must be at least 8^3");
  if (nzl < 2 * h3d::HALO || z0 < 0 || z0 + nzl > p->nz)
    return tau::fail("tau3d_create: slab [%d,%d) invalid for nz=%d (need >= 6 planes)", z0, z0 + nzl, p->nz);
  TAU_HIP(hipSetDevice(device));
  tau3d *h = new (std::nothrow) tau3d();
  if (!h) return tau::fail("tau3d_create: out of host memory");
  tau::HandleGuard<tau3d> guard{h, tau3d_destroy};
  h->p = *p; h->z0 = z0; h->nzl = nzl; h->device = device; h->cur = 0;
  h->plane_n = (size_t)p->nx * p->ny;
  h->field_n = h->plane_n * (size_t)(nzl + 2 * h3d::HALO);
  h->own_stream = (stream == nullptr);
  if (h->own_stream) TAU_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  else h->stream = (hipStream_t)stream;
  // the split step pays off where the fused kernel is occupancy bound (large planes); small grids are bound by
  // the latency between dependent dispatches and keep the one-kernel step.  TAU3D_SPLIT=0/1 overrides.
  h->split = (long)p->nx * p->ny >= 128L * 128L;
  if (const char *e = getenv("TAU3D_SPLIT")) h->split = atoi(e) != 0;
  // The six fields of an array group are ONE allocation, field f at f * field_n: the per-field pointers the API hands out
  // are as before, but a kernel can address all six through one base pointer and a stride — k_update_z touches five such
  // groups, and thirty separate pointers (60 SGPRs) had it spilling scalars to VGPR lanes (478 v_readlane per plane).
  h->field_stride = (h->field_n + 63) & ~(size_t)63;          // keeps every field 256-byte aligned
  h->dxy_stride = (h->plane_n * (size_t)nzl + 63) & ~(size_t)63;
  for (int s = 0; s < 2; s++) {
    TAU_HIP(hipMalloc(&h->buf[s][0], 6 * h->field_stride * sizeof(float)));
    TAU_HIP(hipMemsetAsync(h->buf[s][0], 0, 6 * h->field_stride * sizeof(float), h->stream));
    for (int f = 1; f < 6; f++) h->buf[s][f] = h->buf[s][0] + f * h->field_stride;
  }
  h->debug_no_xy_fix = getenv("TAU3D_DEBUG_NO_XY_FIX") != nullptr;
  h->no_wrap = getenv("TAU3D_NO_WRAP") != nullptr;
  if (const char *e = getenv("TAU3D_UNIFORM_EXITS")) h->uniform_exits = atoi(e) != 0;   // 0: the full path everywhere (same bits; the A/B of the exits)
  if (const char *e = getenv("TAU3D_TILE_LIST")) { h->tile_list = atoi(e); if (h->tile_list < 0 || h->tile_list > 2) h->tile_list = 0; }
  if (const char *e = getenv("TAU3D_Z_SKIP")) h->z_skip = atoi(e) != 0;
  if (const char *e = getenv("TAU3D_TILE_LIST_MARGIN_DIV")) { const int v = atoi(e); if (v >= 1) h->list_margin_div = (unsigned)v; }
  if (h->split && split_buffers(h)) return 1;
  TAU_HIP(hipMalloc(&h->solid, h->field_n));
  TAU_HIP(hipMalloc(&h->clk, sizeof(h3d::DevClock)));
  for (int k = 0; k < 2; k++)
    for (int sd = 0; sd < 2; sd++) TAU_HIP(hipMalloc(&h->xbuf[k][sd], 6 * (size_t)h3d::HALO * h->plane_n * sizeof(float)));
  h->zchunk = 0; // 0 = pick per launch
  if (const char *e = getenv("TAU3D_ZCHUNK")) { int v = atoi(e); if (v >= 1) h->zchunk = v; }
  fill_consts(h);
  h->expect_fast = h->base.in_fmax <= 3.0e38f;   // (TAU3D_WENO_RCP=1 or an inflow state beyond the fast window: the reciprocal form for good)
  if (h->expect_fast) h->expect_fast = h3d::fast_form(0.f, h->base.in_fmax);
  // the solid mask depends on the parameters and the slab only: a caller that goes create -> upload -> step
  // (without tau3d_init) must find it built (and the state buffers defined: zeroed above)
  if (build_solid(h)) return 1;
  { // nothing is known about the state yet: the first k_step takes the reciprocal form unless init / upload measured it
    h3d::DevClock c0;
    memset(&c0, 0, sizeof(c0));
    c0.fmax_bits = 0x7F800000u;
    c0.fmax_in = INFINITY;
    TAU_HIP(hipMemcpyAsync(h->clk, &c0, sizeof(c0), hipMemcpyHostToDevice, h->stream));
    TAU_HIP(hipStreamSynchronize(h->stream));
  }
  tau3d_clock c = {1e-5f, 1e-3f, 0.f, 0.f, 0.f, 0};
  *out = guard.release();
  return tau3d_set_clock(h, &c);
}

extern "C" void tau3d_destroy(tau3d_t *h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  for (int s = 0; s < 2; s++)
    hipFree(h->buf[s][0]);
  hipFree(h->dxy[0]);
  hipFree(h->solid);
  hipFree(h->xyflag);
  hipFree(h->dzero);
  hipFree(h->uflag[0]); hipFree(h->uflag[1]); hipFree(h->uref[0]); hipFree(h->uref[1]); hipFree(h->vflag); hipFree(h->vref);
  hipFree(h->ulist); hipFree(h->ucount);
  if (h->ucount_host) hipHostFree(h->ucount_host);
  for (int i = 0; i < 4; i++) if (h->list_ev[i]) hipEventDestroy(h->list_ev[i]);
  hipFree(h->clk);
  hipFree(h->vis); hipFree(h->rgba); hipFree(h->scratch); hipFree(h->pidx);
  for (int k = 0; k < 2; k++)
    for (int sd = 0; sd < 2; sd++) hipFree(h->xbuf[k][sd]);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  if (h->ev_made) for (int i = 0; i < 4096; i++) { hipEventDestroy(h->ev0[i]); hipEventDestroy(h->ev1[i]); hipEventDestroy(h->evm[i]); }
  delete h;
}

extern "C" int tau3d_set_clock(tau3d_t *h, const tau3d_clock *in) {
  h->end_pending = false;   // whatever was pending is overwritten
  h3d::DevClock c;
  c.t = in->t; c.d_tau = in->d_tau; c.dt = in->dt; c.gain = in->gain; c.maxs_last = in->maxs;
  c.step = in->step; c.maxs_bits = 0u; c.cfl = h->p.cfl; c.pad_ = 0u;
  TAU_HIP(hipMemcpyAsync(h->clk, &c, offsetof(h3d::DevClock, fmax_bits), hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int tau3d_get_clock(tau3d_t *h, tau3d_clock *out) {
  if (flush_clock(h)) return 1;
  h3d::DevClock c;
  TAU_HIP(hipMemcpyAsync(&c, h->clk, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  out->t = c.t; out->d_tau = c.d_tau; out->dt = c.dt; out->gain = c.gain; out->maxs = c.maxs_last; out->step = c.step;
  // the host's expectation of the next step's WENO form (k_flux_xy): the range the last step read and the one it wrote
  h->expect_fast = h3d::fast_form(fmaxf(c.fmax_in, __builtin_bit_cast(float, c.fmax_bits)), h->base.in_fmax);
  return 0;
}

// The state was written from outside the step kernel: fold its largest |primitive| into fmax_bits (`fresh`: the
// whole local state was replaced, forget what was there).  The next clock_begin commits it.
static int measure_field(tau3d_t *h, int zl_lo, int zl_hi, bool fresh) {
  if (fresh) TAU_HIP(hipMemsetAsync(&h->clk->fmax_bits, 0, sizeof(unsigned), h->stream));
  h3d::Args a = h->base;
  for (int f = 0; f < 6; f++) a.in[f] = h->buf[h->cur][f];
  hipLaunchKernelGGL(h3d::k_field_max, dim3(1024), dim3(256), 0, h->stream, a, zl_lo + h3d::HALO, zl_hi + h3d::HALO);
  TAU_LAUNCH_CHECK("k_field_max");
  return 0;
}

extern "C" int tau3d_init(tau3d_t *h, int mode) {
  h->halo_fresh = false; h->list_ok = false;
  TAU_HIP(hipSetDevice(h->device));
  const tau3d_params &P = h->p;
  if (build_solid(h)) return 1;
  // encoded cell values with host libm (identical to what the reference's k_init encodes up to
  // __logf vs logf): fluid = inflow rho,p at rest (mode 0) or full inflow state (mode 1)
  h3d::InitVals iv;
  float r = fmaxf(P.inflow_r, 1e-30f), p = fmaxf(P.inflow_p, 1e-30f);
  float ev = host_evib_eq(P, p / (r * P.R));
  float u = mode ? P.inflow_u : 0.f, v = mode ? P.inflow_v : 0.f, w = mode ? P.inflow_w : 0.f;
  iv.f[0] = logf(fmaxf(r, 1e-30f)); iv.f[1] = host_asinh(u / P.u_ref); iv.f[2] = host_asinh(v / P.u_ref);
  iv.f[3] = host_asinh(w / P.u_ref); iv.f[4] = logf(fmaxf(p, 1e-30f)); iv.f[5] = logf(fmaxf(ev, 1e-30f));
  float rs = fmaxf(p / (P.R * fmaxf(P.Twall, 1e-6f)), 1e-30f);
  float evs = host_evib_eq(P, P.Twall);
  iv.s[0] = logf(fmaxf(rs, 1e-30f)); iv.s[1] = host_asinh(0.f); iv.s[2] = host_asinh(0.f); iv.s[3] = host_asinh(0.f);
  iv.s[4] = logf(fmaxf(p, 1e-30f)); iv.s[5] = logf(fmaxf(evs, 1e-30f));
  float **b = h->buf[h->cur];
  hipLaunchKernelGGL(h3d::k_init, dim3(2048), dim3(256), 0, h->stream, h->base, b[0], b[1], b[2], b[3], b[4], b[5],
                     (const uint8_t *)h->solid, iv);
  TAU_LAUNCH_CHECK("k_init");
  if (measure_field(h, -h3d::HALO, h->nzl + h3d::HALO, true)) return 1;
  tau3d_clock c = {1e-5f, 1e-3f, 0.f, 0.f, 0.f, 0};
  return tau3d_set_clock(h, &c);
}

extern "C" int tau3d_upload_state(tau3d_t *h, const float *const host[6]) {
  h->halo_fresh = false; h->list_ok = false;
  TAU_HIP(hipSetDevice(h->device));
  size_t n = h->plane_n * (size_t)h->nzl;
  for (int f = 0; f < 6; f++)
    TAU_HIP(hipMemcpyAsync(h->buf[h->cur][f] + h3d::HALO * h->plane_n, host[f], n * sizeof(float),
                           hipMemcpyHostToDevice, h->stream));
  // The local planes were replaced: the range is re-measured over them AND the halo planes (defined since tau3d_create
  // zeroed them; an earlier tau3d_upload_planes may have put values there that the next step reads before any exchange).
  // A ring all-reduces the range when it primes.
  if (measure_field(h, -h3d::HALO, h->nzl + h3d::HALO, true)) return 1;
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tau3d_download_state(tau3d_t *h, float *const host[6]) {
  TAU_HIP(hipSetDevice(h->device));
  size_t n = h->plane_n * (size_t)h->nzl;
  for (int f = 0; f < 6; f++)
    TAU_HIP(hipMemcpyAsync(host[f], h->buf[h->cur][f] + h3d::HALO * h->plane_n, n * sizeof(float),
                           hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tau3d_download_solid(tau3d_t *h, uint8_t *host) {
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipMemcpyAsync(host, h->solid + h3d::HALO * h->plane_n, h->plane_n * (size_t)h->nzl,
                         hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
static int planes_copy(tau3d_t *h, int zl_lo, int zl_hi, const float *const up[6], float *const down[6]) {
  h->halo_fresh = false; h->list_ok = false;
  if (zl_lo < -h3d::HALO || zl_hi > h->nzl + h3d::HALO || zl_lo >= zl_hi)
    return tau::fail("tau3d planes: range [%d,%d) outside [-3,%d)", zl_lo, zl_hi, h->nzl + 3);
  TAU_HIP(hipSetDevice(h->device));
  size_t off = (size_t)(zl_lo + h3d::HALO) * h->plane_n, n = (size_t)(zl_hi - zl_lo) * h->plane_n * sizeof(float);
  for (int f = 0; f < 6; f++) {
    if (up) TAU_HIP(hipMemcpyAsync(h->buf[h->cur][f] + off, up[f], n, hipMemcpyHostToDevice, h->stream));
    else TAU_HIP(hipMemcpyAsync(down[f], h->buf[h->cur][f] + off, n, hipMemcpyDeviceToHost, h->stream));
  }
  if (up && measure_field(h, zl_lo, zl_hi, false)) return 1;
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tau3d_upload_planes(tau3d_t *h, int zl_lo, int zl_hi, const float *const host[6]) {
  return planes_copy(h, zl_lo, zl_hi, host, nullptr);
}
extern "C" int tau3d_download_planes(tau3d_t *h, int zl_lo, int zl_hi, float *const host[6]) {
  return planes_copy(h, zl_lo, zl_hi, nullptr, host);
}
extern "C" int tau3d_state_ptrs(tau3d_t *h, float *dptr[6], uint8_t **solid) {
  for (int f = 0; f < 6; f++) dptr[f] = h->buf[h->cur][f] + h3d::HALO * h->plane_n;
  if (solid) *solid = h->solid + h3d::HALO * h->plane_n;
  return 0;
}

static int flush_clock(tau3d_t *h) { // the deferred k_clock_end of tau3d_step_async
  if (!h->end_pending) return 0;
  hipLaunchKernelGGL(h3d::k_clock_end, dim3(1), dim3(1), 0, h->stream, h->clk);
  TAU_LAUNCH_CHECK("k_clock_end");
  h->end_pending = false;
  return 0;
}
static int fill_halo(tau3d_t *h, bool with_clock) {
  h3d::HaloArgs H;
  H.clk = with_clock ? h->clk : nullptr; H.do_end = h->end_pending ? 1 : 0;
  for (int f = 0; f < 6; f++) H.f[f] = h->buf[h->cur][f];
  H.plane_n = h->plane_n; H.nzl = h->nzl;
  H.copy = (with_clock && h->halo_fresh) ? 0 : 1;
  if (!H.copy) hipLaunchKernelGGL(h3d::k_halo_periodic, dim3(1, 1), dim3(64), 0, h->stream, H);
  else
  hipLaunchKernelGGL(h3d::k_halo_periodic, dim3(64, 12), dim3(256), 0, h->stream, H);
  TAU_LAUNCH_CHECK("k_halo_periodic");
  if (with_clock) h->end_pending = false;
  return 0;
}
extern "C" int tau3d_fill_halo_periodic_async(tau3d_t *h) { return fill_halo(h, false); }

// ---- the two launches of the split step over planes [lo, hi) (+ [lo2, hi2) in the same launch)
static void split_args(tau3d_t *h, h3d::Args &A, int lo, int hi, int lo2, int hi2) {
  A = h->base;
  for (int f = 0; f < 6; f++) {
    A.in[f] = h->buf[h->cur][f]; A.out[f] = h->buf[h->cur ^ 1][f]; A.dxy[f] = h->dxy[f];
  }
  A.zl_lo = lo; A.zl_hi = hi; A.zl_lo2 = lo2; A.zl_hi2 = hi2;
  A.send[0] = A.send[1] = nullptr;
  A.in0 = h->buf[h->cur][0]; A.out0 = h->buf[h->cur ^ 1][0];
  A.d0 = h->dxy[0]; A.fstride = (unsigned)h->field_stride; A.dstride = (unsigned)h->dxy_stride;
  A.xyflag = h->xyflag;
  A.dzero = h->uniform_exits ? h->dzero : nullptr;
  A.dz_ntx = (A.nx + h3d::XY_FX - 1) / h3d::XY_FX; A.dz_nty = (A.ny + h3d::XY_FY - 1) / h3d::XY_FY;
  A.uflag_w = A.dzero ? h->uflag[h->upar] : nullptr; A.uref_w = h->uref[h->upar];   // (null unless the handle keeps a tile list)
  A.uflag_r = h->uflag[h->upar]; A.uref_r = h->uref[h->upar];
  const bool commit = h->tile_list == 1;
  A.pflag = commit ? h->uflag[h->upar ^ 1] : h->vflag; A.pref = commit ? h->uref[h->upar ^ 1] : h->vref;
  A.pred_commit = commit ? 1 : 0;
  A.ulist = h->ulist;
  A.ucount = nullptr; A.ucount_other = nullptr; A.z_pred = 0;   // (set by the launch that uses them)
  A.ucount_host = h->ucount_host_dev;
  A.wrap_halo = (h->wrap_now && lo == 0 && hi == h->nzl && lo2 >= hi2) ? 1 : 0;
}
static int split_xy(tau3d_t *h, int lo, int hi, int lo2, int hi2, hipStream_t s, bool fix = false, bool listed = false) {   // x/y faces: one plane per workgroup
  h3d::Args X;
  split_args(h, X, lo, hi, lo2, hi2);
  const int n1 = hi - lo, n2 = lo2 < hi2 ? hi2 - lo2 : 0;
  X.zchunk = 1; X.nzc1 = n1; X.nzc = n1 + n2;
  X.ntx = (X.nx + h3d::XY_WX - 1) / h3d::XY_WX; X.nty = (X.ny + h3d::XY_WY - 1) / h3d::XY_WY;
  if (listed) {
    // The grid: the length of the list as an earlier k_flux_xy_list reported it (a mapped host word), plus a margin — the
    // disturbed region gains a few tiles a step.  A short grid costs time (k_flux_xy_list_rest takes what is left), never tiles; a
    // long one costs the dispatch of empty workgroups.  For the length to be a recent one the host may not run far ahead of the
    // device: a step waits for the step three before it (an event per step), which leaves the device two steps of queued work —
    // the call stays asynchronous with a queue four steps deep.  No length seen yet: one workgroup per tile.
    if (h->list_step >= 3) TAU_HIP(hipEventSynchronize(h->list_ev[(h->list_step - 3) & 3]));
    const unsigned nt = (unsigned)(X.ntx * X.nty * X.nzc);
    const unsigned seen = *(volatile unsigned *)h->ucount_host;
    unsigned g = seen == 0xFFFFFFFFu ? nt : seen + seen / h->list_margin_div + 512u;
    g = g < nt ? g : nt;
    X.ucount = h->ucount + h->upar;
    h3d::launch_flux_xy_list(g, s, X, h->expect_fast);
  } else
  if (fix) h3d::launch_flux_xy_fix((unsigned)(X.ntx * X.nty * X.nzc), s, X);
  else h3d::launch_flux_xy((unsigned)(X.ntx * X.nty * X.nzc), s, X, h->expect_fast);
  TAU_LAUNCH_CHECK("k_flux_xy");
  return 0;
}
// z faces + update: a wave marches a chunk of planes; ~4k workgroups (four rounds of the ~1k resident ones).
// `pack`: the new boundary planes also go into the packed send buffers (Z-slab ring)
static int split_z(tau3d_t *h, int lo, int hi, int lo2, int hi2, bool pack, hipStream_t s, bool predicted = false) {
  h3d::Args Z;
  split_args(h, Z, lo, hi, lo2, hi2);
  if (predicted) Z.z_pred = h->z_skip ? 2 : 1;   // k_tile_predict ran for this step
  const int n1 = hi - lo, n2 = lo2 < hi2 ? hi2 - lo2 : 0;
  const long tz = (long)((Z.nx + h3d::ZT_X - 1) / h3d::ZT_X) * ((Z.ny + h3d::ZT_Y - 1) / h3d::ZT_Y);
  // chunk length: ~4 k workgroups (16 k waves: three rounds of the ~5 k this kernel keeps resident, for load balance),
  // but at least 16 planes where the range has them — a chunk pays about one plane-iteration of warm-up (two
  // reconstructions and a face).  Fewer, longer chunks were measured: 256^3 in 4 layers of 64 planes 1117 us against 997.
  int zc = h->zchunk;
  if (zc <= 0) { zc = (int)((long)(n1 + n2) * tz / 4096); zc = zc < 16 ? 16 : (zc > 64 ? 64 : zc); }
  {   // the kernel addresses a chunk with 32-bit byte offsets from its first plane (tau3d_create checked that 8 planes fit)
    const long zmax = (long)(0xFFFFFFFFull / (h->plane_n * sizeof(float)));
    if (zc > zmax) zc = (int)zmax;
  }
  Z.zchunk = zc < n1 ? zc : n1;
  Z.nzc1 = (n1 + Z.zchunk - 1) / Z.zchunk;
  Z.nzc = Z.nzc1 + (n2 ? (n2 + Z.zchunk - 1) / Z.zchunk : 0);
  if (pack) { Z.send[0] = h->xbuf[0][0]; Z.send[1] = h->xbuf[0][1]; }
  h3d::launch_update_z((unsigned)(tz * Z.nzc), s, Z, h->expect_fast);
  TAU_LAUNCH_CHECK("k_update_z");
  return 0;
}

// ---- predicted-uniform tiles: the three places a step over EVERY local plane touches the list
// x/y fluxes: over the list if the step before left one that describes this step's input, else over every tile (*can_list: the
// handle keeps a list and this step will make the next one)
static int list_xy(tau3d_t *h, hipStream_t s, bool *can_list) {
  *can_list = h->uflag[0] != nullptr && h->uniform_exits && h->tile_list != 0;
  const bool have_pred = *can_list && h->list_ok && h->list_cur == h->cur;   // the last step's prediction describes this step's input
  const bool use_list = have_pred && h->tile_list == 1;
  h->list_ok = false;
  if (*can_list && !use_list) TAU_HIP(hipMemsetAsync(h->ucount, 0, 2 * sizeof(unsigned), s));
  if (split_xy(h, 0, h->nzl, 0, 0, s, false, use_list)) return 1;
  if (have_pred && h->tile_list == 2) {
    const unsigned nt = (unsigned)(((h->p.nx + h3d::XY_FX - 1) / h3d::XY_FX) * ((h->p.ny + h3d::XY_FY - 1) / h3d::XY_FY)) * (unsigned)h->nzl;
    hipLaunchKernelGGL(h3d::k_tile_predict_check, dim3((nt + 255u) / 256u), dim3(256), 0, s, (const unsigned *)h->vflag, (const float *)h->vref,
                       (const unsigned *)h->uflag[h->upar], (const float *)h->uref[h->upar], nt, h->ucount + 2, h->ucount + 3);
    TAU_LAUNCH_CHECK("k_tile_predict_check");
  }
  return 0;
}
// between the step's k_flux_xy (and, on a slab, the clock of the step: the prediction computes the new state with its dt) and its k_update_z
static int list_predict(tau3d_t *h, hipStream_t s) {
  h3d::Args P;
  split_args(h, P, 0, h->nzl, 0, 0);
  P.ucount = h->ucount + (h->upar ^ 1); P.ucount_other = h->ucount + h->upar;
  const unsigned nt = (unsigned)(P.dz_ntx * P.dz_nty) * (unsigned)h->nzl;
  hipLaunchKernelGGL(h3d::k_tile_predict, dim3((nt + h3d::PREDICT_NT - 1) / h3d::PREDICT_NT), dim3(h3d::PREDICT_NT), 0, s, P);
  TAU_LAUNCH_CHECK("k_tile_predict");
  return 0;
}
// after the step's k_update_z: the list describes the buffer that becomes current with the swap
static int list_commit(tau3d_t *h, hipStream_t s) {
  h->upar ^= 1; h->list_ok = true; h->list_cur = h->cur ^ 1;
  TAU_HIP(hipEventRecord(h->list_ev[h->list_step & 3], s));
  h->list_step++;
  return 0;
}

// steps planes [zl_lo, zl_hi) and, if zl_lo2 < zl_hi2, also [zl_lo2, zl_hi2) in the SAME launch
static int step_ranges(tau3d_t *h, int zl_lo, int zl_hi, int zl_lo2, int zl_hi2, void *stream) {
  if (zl_lo < 0 || zl_hi > h->nzl || zl_lo >= zl_hi) return tau::fail("tau3d_step_range: bad plane range [%d,%d)", zl_lo, zl_hi);
  const bool two = zl_lo2 < zl_hi2;
  if (two && (zl_lo2 < zl_hi || zl_hi2 > h->nzl)) return tau::fail("tau3d_step_edges: bad second range [%d,%d)", zl_lo2, zl_hi2);
  TAU_HIP(hipSetDevice(h->device));
  h->halo_fresh = false;   // (tau3d_step_async sets it again after a step whose k_update_z wrote the halos)
  h3d::Args A = h->base;
  for (int f = 0; f < 6; f++) { A.in[f] = h->buf[h->cur][f]; A.out[f] = h->buf[h->cur ^ 1][f]; }
  A.zl_lo = zl_lo; A.zl_hi = zl_hi;
  int nplanes = zl_hi - zl_lo;           // chunking is chosen for the first range; a second range has the same length
  if (h->split) {
    const bool tm = h->timing && h->n_ev < 4096;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (tm) { TAU_HIP(hipEventRecord(h->ev0[h->n_ev], s)); h->evm_set[h->n_ev] = false; }
    // predicted-uniform tiles: a step over every local plane of a handle that keeps the list (list_xy / list_predict / list_commit)
    const bool whole = zl_lo == 0 && zl_hi == h->nzl && !two;
    bool can_list = false;
    if (whole) { if (list_xy(h, s, &can_list)) return 1; }
    else { h->list_ok = false; if (split_xy(h, zl_lo, zl_hi, zl_lo2, zl_hi2, s)) return 1; }
    if (tm) { TAU_HIP(hipEventRecord(h->evm[h->n_ev], s)); h->evm_set[h->n_ev] = true; }
    if (can_list && list_predict(h, s)) return 1;
    if (split_z(h, zl_lo, zl_hi, zl_lo2, zl_hi2, false, s, can_list)) return 1;
    if (can_list && list_commit(h, s)) return 1;
    if (tm) {
      TAU_HIP(hipEventRecord(h->ev1[h->n_ev], s));
      h->n_ev++;
      h->ev_cells += (double)(nplanes + (two ? zl_hi2 - zl_lo2 : 0)) * (double)h->plane_n;
    }
    return 0;
  }
  // Planes marched by one workgroup.  Large grids: enough workgroups (~32k) to load-balance 256 CUs x 3 resident
  // groups, chunks of 8..32 planes (a chunk re-decodes 4 warm-up planes, so longer is cheaper).  Small grids are
  // LATENCY bound instead — a 64^3 launch is 128 workgroups of 13 serial plane iterations with chunks of 8 — so
  // below that the chunk shrinks (down to 2) until there are ~2k workgroups: 64^3 2.8 -> 6.0 Gcell/s.
  int zc = h->zchunk;
  if (zc <= 0) {
    const long tiles = (long)A.ntx * A.nty;
    zc = (int)((long)nplanes * tiles / 32768);
    if (zc < 8) {
      if (tiles >= 768) {                              // wide planes fill the chip by themselves (a Z-slab of a big grid):
        const long nzc = (nplanes * tiles + 16384) / 32768;   // as few chunks as keep ~32k workgroups' worth of work
        zc = (int)((nplanes + (nzc > 0 ? nzc : 1) - 1) / (nzc > 0 ? nzc : 1));
        zc = zc < 8 ? 8 : zc;
      }
      else { zc = (int)((long)nplanes * tiles / 2048); zc = zc < 2 ? 2 : (zc > 8 ? 8 : zc); }
    }
    zc = zc > 32 ? 32 : zc;
  }
  A.zchunk = zc < nplanes ? zc : nplanes;
  A.nzc1 = (nplanes + A.zchunk - 1) / A.zchunk;
  A.zl_lo2 = zl_lo2; A.zl_hi2 = zl_hi2;
  A.nzc = A.nzc1 + (two ? (zl_hi2 - zl_lo2 + A.zchunk - 1) / A.zchunk : 0);
  unsigned nb = (unsigned)(A.ntx * A.nty * A.nzc);
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const bool tm = h->timing && h->n_ev < 4096;
  if (tm) { TAU_HIP(hipEventRecord(h->ev0[h->n_ev], s)); h->evm_set[h->n_ev] = false; }
  hipLaunchKernelGGL(h3d::k_step, dim3(nb), dim3(h3d::NT), 0, s, A);
  TAU_LAUNCH_CHECK("k_step");
  if (tm) {
    TAU_HIP(hipEventRecord(h->ev1[h->n_ev], s));
    h->n_ev++;
    h->ev_cells += (double)(nplanes + (two ? zl_hi2 - zl_lo2 : 0)) * (double)h->plane_n;
  }
  return 0;
}
extern "C" int tau3d_step_range_async(tau3d_t *h, int zl_lo, int zl_hi, void *stream) {
  return step_ranges(h, zl_lo, zl_hi, 0, 0, stream);
}
extern "C" int tau3d_step_edges_async(tau3d_t *h, int depth, void *stream) {
  if (depth < h3d::HALO) return tau::fail("tau3d_step_edges: depth %d is less than the %d halo planes", depth, h3d::HALO);
  if (2 * depth >= h->nzl) return step_ranges(h, 0, h->nzl, 0, 0, stream);
  return step_ranges(h, 0, depth, h->nzl - depth, h->nzl, stream);
}

// ---- Z-slab ring step in (at most) four dispatches: see include/taueng.h
static int halo_pack(tau3d_t *h, int which, int dir, bool with_clock);
extern "C" int tau3d_slab_begin_async(tau3d_t *h) {
  TAU_HIP(hipSetDevice(h->device));
  if (h->direct) {   // the halo planes were written in place by the neighbours: only the controller / clock part is left
    hipLaunchKernelGGL(h3d::k_clock_turn, dim3(1), dim3(1), 0, h->stream, h->clk, h->end_pending ? 1 : 0);
    TAU_LAUNCH_CHECK("k_clock_turn");
    h->end_pending = false;
    return 0;
  }
  return halo_pack(h, 0, 1, true);   // controller of the step before (if pending) + clock of this one + received halos
}
// the controller / clock part alone (the pipelined ring step of the packed transports unpacks the halos later, before its z kernel)
extern "C" int tau3d_slab_clock_async(tau3d_t *h) {
  if (!h) return tau::fail("tau3d_slab_clock: null handle");
  TAU_HIP(hipSetDevice(h->device));
  hipLaunchKernelGGL(h3d::k_clock_turn, dim3(1), dim3(1), 0, h->stream, h->clk, h->end_pending ? 1 : 0);
  TAU_LAUNCH_CHECK("k_clock_turn");
  h->end_pending = false;
  return 0;
}
extern "C" int tau3d_set_halo_direct(tau3d_t *h, int on) {
  if (!h) return tau::fail("tau3d_set_halo_direct: null handle");
  h->direct = on != 0;
  return 0;
}
extern "C" int tau3d_state_group(tau3d_t *h, int which, void **base, size_t *bytes, size_t *field_stride, int *index) {
  if (!h || (which | 1) != 1) return tau::fail("tau3d_state_group: bad argument");
  if (base) *base = h->buf[h->cur ^ which][0];
  if (bytes) *bytes = 6 * h->field_stride * sizeof(float);
  if (field_stride) *field_stride = h->field_stride;
  if (index) *index = h->cur ^ which;
  return 0;
}
// event timing of a slab piece (tau3d_timing_*): the interval covers every launch of the piece
static int slab_timed(tau3d_t *h, int planes, int (*body)(tau3d_t *, int), int depth) {
  const bool tm = h->timing && h->n_ev < 4096;
  if (tm) { TAU_HIP(hipEventRecord(h->ev0[h->n_ev], h->stream)); h->evm_set[h->n_ev] = false; }
  if (body(h, depth)) return 1;
  if (tm) {
    TAU_HIP(hipEventRecord(h->ev1[h->n_ev], h->stream));
    h->n_ev++;
    h->ev_cells += (double)planes * (double)h->plane_n;
  }
  return 0;
}
// Which planes get their x/y fluxes in the edges piece: all of them (four dispatches per step, but the exchange posted after
// the edges then only has the interior k_update_z to hide behind), or — the default — only the edge planes, the interior
// ones following in the interior piece (five dispatches; the exchange overlaps the interior's k_flux_xy AND k_update_z, ~0.75
// of the ~1.0 ms an 8-way 512^3 slab computes per step, against 18.9 MB per direction over one xGMI link).  TAU3D_SLAB_XY_FIRST=1
// selects the former.
static bool slab_xy_first() {
  static const bool v = [] { const char *e = getenv("TAU3D_SLAB_XY_FIRST"); return e && atoi(e) != 0; }();
  return v;
}
static int slab_edges_body(tau3d_t *h, int depth) {
  const int nzl = h->nzl;
  h->list_ok = false;   // (the edge / interior schedule steps in pieces: no tile list)
  const bool whole = 2 * depth >= nzl;
  if (h->split) {
    if (whole || slab_xy_first()) { if (split_xy(h, 0, nzl, 0, 0, h->stream)) return 1; }   // k_flux_xy needs no halo
    else if (split_xy(h, 0, depth, nzl - depth, nzl, h->stream)) return 1;
    const bool pack = !h->direct;   // direct halos: the ring copies the new boundary planes out of the state itself
    return whole ? split_z(h, 0, nzl, 0, 0, pack, h->stream) : split_z(h, 0, depth, nzl - depth, nzl, pack, h->stream);
  }
  const bool t = h->timing;
  h->timing = false;                                               // (step_ranges would open an interval of its own)
  const int rc = whole ? step_ranges(h, 0, nzl, 0, 0, nullptr) : step_ranges(h, 0, depth, nzl - depth, nzl, nullptr);
  h->timing = t;
  if (rc) return 1;
  return h->direct ? 0 : halo_pack(h, 1, 0, false);
}
static int slab_interior_body(tau3d_t *h, int depth) {
  const int nzl = h->nzl;
  if (h->split) {
    if (!slab_xy_first() && split_xy(h, depth, nzl - depth, 0, 0, h->stream)) return 1;
    return split_z(h, depth, nzl - depth, 0, 0, false, h->stream);
  }
  const bool t = h->timing;
  h->timing = false;
  const int rc = step_ranges(h, depth, nzl - depth, 0, 0, nullptr);
  h->timing = t;
  return rc;
}
extern "C" int tau3d_slab_edges_async(tau3d_t *h, int depth) {
  if (depth < h3d::HALO) return tau::fail("tau3d_slab_edges: depth %d is less than the %d halo planes", depth, h3d::HALO);
  TAU_HIP(hipSetDevice(h->device));
  return slab_timed(h, 2 * depth >= h->nzl ? h->nzl : 2 * depth, slab_edges_body, depth);
}
extern "C" int tau3d_slab_interior_async(tau3d_t *h, int depth) {
  if (2 * depth >= h->nzl) return 0;
  TAU_HIP(hipSetDevice(h->device));
  return slab_timed(h, h->nzl - 2 * depth, slab_interior_body, depth);
}
// The whole slab in two pieces for the direct-halo ring's pipelined step: the x/y fluxes of every plane read no halo plane, so
// they run while the halos of the step before are still arriving; the z kernel (fused kernel on small planes: the whole step)
// follows once those have landed.  One timing interval spans both.
extern "C" int tau3d_slab_xy_async(tau3d_t *h) {
  if (!h) return tau::fail("tau3d_slab_xy: null handle");
  TAU_HIP(hipSetDevice(h->device));
  h->halo_fresh = false;
  if (!h->split) { h->list_ok = false; return 0; }       // fused kernel: everything happens in tau3d_slab_z_async
  const bool tm = h->timing && h->n_ev < 4096;
  if (tm) { TAU_HIP(hipEventRecord(h->ev0[h->n_ev], h->stream)); h->evm_set[h->n_ev] = false; }
  // (a slab keeps the list too: its planes within three of an edge are never predicted — their z neighbours' flags are another rank's)
  if (list_xy(h, h->stream, &h->slab_can_list)) return 1;
  if (tm) { TAU_HIP(hipEventRecord(h->evm[h->n_ev], h->stream)); h->evm_set[h->n_ev] = true; }
  return 0;
}
// after a tau3d_slab_xy_async that was issued BEFORE this step's tau3d_slab_begin_async / tau3d_slab_clock_async (the ring's
// speculative launch): repeats the x/y fluxes if the clock's commit says they took the wrong WENO weight form — else nothing
extern "C" int tau3d_slab_xy_fix_async(tau3d_t *h) {
  if (!h) return tau::fail("tau3d_slab_xy_fix: null handle");
  TAU_HIP(hipSetDevice(h->device));
  if (!h->split) return 0;
  if (h->debug_no_xy_fix) return 0;   // tests only (TAU3D_DEBUG_NO_XY_FIX, read once at tau3d_create): the control run that shows a wrong-form launch is visible
  return split_xy(h, 0, h->nzl, 0, 0, h->stream, true);
}
/* test hook: the range word an x/y flux launch ahead of the clock reads (the next commit overwrites it) */
extern "C" int tau3d_debug_set_fmax_in(tau3d_t *h, float v) {
  if (!h) return tau::fail("tau3d_debug_set_fmax_in: null handle");
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipMemcpyAsync(&h->clk->fmax_in, &v, sizeof v, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tau3d_slab_z_async(tau3d_t *h) {
  if (!h) return tau::fail("tau3d_slab_z: null handle");
  TAU_HIP(hipSetDevice(h->device));
  if (!h->split) return slab_timed(h, h->nzl, slab_edges_body, h->nzl);
  const bool tm = h->timing && h->n_ev < 4096;
  const bool can_list = h->slab_can_list;   // (set by this step's tau3d_slab_xy_async; the clock of the step has run since)
  h->slab_can_list = false;
  if (can_list && list_predict(h, h->stream)) return 1;
  if (split_z(h, 0, h->nzl, 0, 0, !h->direct, h->stream, can_list)) return 1;
  if (can_list && list_commit(h, h->stream)) return 1;
  if (tm) {
    TAU_HIP(hipEventRecord(h->ev1[h->n_ev], h->stream));
    h->n_ev++;
    h->ev_cells += (double)h->nzl * (double)h->plane_n;
  }
  return 0;
}
extern "C" int tau3d_slab_end_async(tau3d_t *h) {
  h->halo_fresh = false;
  if (!(h->list_ok && h->list_cur == (h->cur ^ 1))) h->list_ok = false;   // (kept: made by this step's tau3d_slab_z_async for the buffer that becomes current)
  h->cur ^= 1;             // std::swap x6, :1706-1711
  h->end_pending = true;   // the controller update (after the caller's all-reduce) rides on the next tau3d_slab_begin_async
  return 0;
}

extern "C" int tau3d_clock_begin_async(tau3d_t *h) {
  if (flush_clock(h)) return 1;
  hipLaunchKernelGGL(h3d::k_clock_begin, dim3(1), dim3(1), 0, h->stream, h->clk);
  TAU_LAUNCH_CHECK("k_clock_begin");
  return 0;
}
extern "C" int tau3d_clock_end_async(tau3d_t *h) {
  h->halo_fresh = false; h->list_ok = false;
  hipLaunchKernelGGL(h3d::k_clock_end, dim3(1), dim3(1), 0, h->stream, h->clk);
  TAU_LAUNCH_CHECK("k_clock_end");
  h->cur ^= 1; // std::swap x6, :1706-1711
  return 0;
}

extern "C" int tau3d_step_async(tau3d_t *h, int nsteps) {
  if (h->nzl != h->p.nz) return tau::fail("tau3d_step: single-domain call on a slab handle (use the *_async pieces)");
  TAU_HIP(hipSetDevice(h->device));
  for (int s = 0; s < nsteps; s++) {
    if (fill_halo(h, true)) return 1;                        // + controller of the previous step + clock of this one
    // split step: k_update_z writes the new state's z halos itself (the copy above then only runs after init / upload / ...)
    h->wrap_now = h->split && h->nzl >= 2 * h3d::HALO && !h->no_wrap;
    const int rc = step_ranges(h, 0, h->nzl, 0, 0, nullptr);
    const bool wrapped = h->wrap_now;
    h->wrap_now = false;
    if (rc) return 1;
    h->halo_fresh = wrapped;
    h->cur ^= 1;                                             // std::swap x6, :1706-1711
    h->end_pending = true;                                   // its controller update rides on the next halo copy
  }
  return 0;
}
extern "C" int tau3d_step(tau3d_t *h, int nsteps, tau3d_clock *out) {
  if (tau3d_step_async(h, nsteps)) return 1;
  if (out) return tau3d_get_clock(h, out);
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int tau3d_step_explicit(tau3d_t *h, float dt, float inflow_gain, float *maxs) {
  if (h->nzl != h->p.nz) return tau::fail("tau3d_step_explicit: single-domain call on a slab handle");
  TAU_HIP(hipSetDevice(h->device));
  if (flush_clock(h)) return 1;
  hipLaunchKernelGGL(h3d::k_clock_set_explicit, dim3(1), dim3(1), 0, h->stream, h->clk, dt, inflow_gain);
  TAU_LAUNCH_CHECK("k_clock_set_explicit");
  if (tau3d_fill_halo_periodic_async(h)) return 1;
  if (tau3d_step_range_async(h, 0, h->nzl, nullptr)) return 1;
  h->cur ^= 1;
  h3d::DevClock c;
  TAU_HIP(hipMemcpyAsync(&c, h->clk, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  if (maxs) { memcpy(maxs, &c.maxs_bits, 4); }
  return 0;
}

extern "C" int tau3d_halo_send_ptr(tau3d_t *h, int which, int field, int side, float **p) {
  if (field < 0 || field > 5 || !p) return tau::fail("tau3d_halo_send_ptr: bad argument");
  float *b = h->buf[h->cur ^ (which & 1)][field];
  *p = side == 0 ? b + (size_t)h3d::HALO * h->plane_n : b + (size_t)h->nzl * h->plane_n;
  return 0;
}
extern "C" int tau3d_halo_recv_ptr(tau3d_t *h, int which, int field, int side, float **p) {
  if (field < 0 || field > 5 || !p) return tau::fail("tau3d_halo_recv_ptr: bad argument");
  float *b = h->buf[h->cur ^ (which & 1)][field];
  *p = side == 0 ? b : b + (size_t)(h->nzl + h3d::HALO) * h->plane_n;
  return 0;
}
static int halo_pack(tau3d_t *h, int which, int dir, bool with_clock) {
  h3d::PackArgs P;
  P.clk = with_clock ? h->clk : nullptr; P.do_end = h->end_pending ? 1 : 0;
  if (with_clock) h->end_pending = false;
  for (int f = 0; f < 6; f++) P.f[f] = h->buf[h->cur ^ (which & 1)][f];
  P.buf[0] = h->xbuf[dir][0]; P.buf[1] = h->xbuf[dir][1];
  P.plane_n = h->plane_n; P.nzl = h->nzl;
  hipLaunchKernelGGL(h3d::k_halo_pack, dim3(32, 12), dim3(256), 0, h->stream, P, dir);
  TAU_LAUNCH_CHECK("k_halo_pack");
  return 0;
}
extern "C" int tau3d_pack_halos_async(tau3d_t *h, int which) { return halo_pack(h, which, 0, false); }
extern "C" int tau3d_unpack_halos_async(tau3d_t *h, int which) { return halo_pack(h, which, 1, false); }
extern "C" int tau3d_halo_buf_ptr(tau3d_t *h, int kind, int side, float **p, size_t *nfloats) {
  if ((kind | 1) != 1 || (side | 1) != 1 || !p) return tau::fail("tau3d_halo_buf_ptr: bad argument");
  *p = h->xbuf[kind][side];
  if (nfloats) *nfloats = 6 * (size_t)h3d::HALO * h->plane_n;
  return 0;
}
extern "C" int tau3d_state_written(tau3d_t *h) {
  h->halo_fresh = false; h->list_ok = false;
  TAU_HIP(hipSetDevice(h->device));
  if (h->xyflag) {   // tau3d_state_ptrs also hands out the solid mask: k_flux_xy's static tile flags are derived from it
    const int ntx = (h->p.nx + h3d::XY_FX - 1) / h3d::XY_FX, nty = (h->p.ny + h3d::XY_FY - 1) / h3d::XY_FY;
    hipLaunchKernelGGL(h3d::k_xy_flags, dim3((unsigned)(ntx * nty * h->nzl)), dim3(256), 0, h->stream, h->base, (const uint8_t *)h->solid, h->xyflag);
    TAU_LAUNCH_CHECK("k_xy_flags");
  }
  return measure_field(h, -h3d::HALO, h->nzl + h3d::HALO, false);
}
extern "C" int tau3d_field_range(tau3d_t *h, float *read_max, float *written_max, int *fast_form) {
  TAU_HIP(hipSetDevice(h->device));
  h3d::DevClock c;
  TAU_HIP(hipMemcpyAsync(&c, h->clk, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  float w;
  memcpy(&w, &c.fmax_bits, 4);
  if (read_max) *read_max = c.fmax_in;
  if (written_max) *written_max = w;
  if (fast_form) *fast_form = h3d::fast_form(c.fmax_in, h->base.in_fmax) ? 1 : 0;
  h->expect_fast = h3d::fast_form(fmaxf(c.fmax_in, __builtin_bit_cast(float, c.fmax_bits)), h->base.in_fmax);
  return 0;
}
/* uniform-region exits of the split step: how many of k_flux_xy's tiles (32 x 16 cells, per local plane) the LAST step found to
 * hold one state — their x/y divergence is exactly zero and was neither computed nor stored; *enabled = 0 when the exits are off
 * (TAU3D_UNIFORM_EXITS=0) or the handle runs the fused kernel */
extern "C" int tau3d_uniform_tiles(tau3d_t *h, long *uniform, long *tiles, int *enabled) {
  if (!h) return tau::fail("tau3d_uniform_tiles: null handle");
  TAU_HIP(hipSetDevice(h->device));
  const bool on = h->split && h->uniform_exits && h->dzero;
  if (enabled) *enabled = on ? 1 : 0;
  const size_t n = (size_t)((h->p.nx + h3d::XY_FX - 1) / h3d::XY_FX) * ((h->p.ny + h3d::XY_FY - 1) / h3d::XY_FY) * (size_t)h->nzl;
  if (tiles) *tiles = (long)n;
  long u = 0;
  if (on) {
    unsigned *host = (unsigned *)malloc(n * sizeof(unsigned));
    if (!host) return tau::fail("tau3d_uniform_tiles: out of host memory");
    hipError_t e = hipMemcpyAsync(host, h->dzero, n * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { free(host); return tau::fail("tau3d_uniform_tiles: %s", hipGetErrorString(e)); }
    for (size_t i = 0; i < n; i++) u += host[i] & 1u;
    free(host);
  }
  if (uniform) *uniform = u;
  return 0;
}
// Predicted-uniform tiles: mode (0: the handle keeps no list — switched off, a slab, ragged tiles —, 1: k_flux_xy runs over the list,
// 2: verifying), the length of the list the last whole-domain step made (the tiles the NEXT k_flux_xy runs), the tile count, and
// in mode 2 the predictions checked so far and how many of them k_flux_xy did not confirm (must be 0).  Waits for the stream.
extern "C" int tau3d_tile_list_stats(tau3d_t *h, int *mode, long *listed, long *tiles, long *checked, long *mismatches) {
  if (!h) return tau::fail("tau3d_tile_list_stats: null handle");
  TAU_HIP(hipSetDevice(h->device));
  const bool on = h->split && h->uflag[0] != nullptr;
  if (mode) *mode = on ? h->tile_list : 0;
  if (tiles) *tiles = (long)((h->p.nx + h3d::XY_FX - 1) / h3d::XY_FX) * ((h->p.ny + h3d::XY_FY - 1) / h3d::XY_FY) * (long)h->nzl;
  unsigned c[4] = {0u, 0u, 0u, 0u};
  if (on) {
    TAU_HIP(hipMemcpyAsync(c, h->ucount, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    TAU_HIP(hipStreamSynchronize(h->stream));
  }
  if (listed) *listed = on && h->list_ok ? (long)c[h->upar] : -1;
  if (checked) *checked = (long)c[3];
  if (mismatches) *mismatches = (long)c[2];
  return 0;
}
extern "C" int tau3d_slab_info(tau3d_t *h, int *z0, int *nzl, int *nz, int *device, void **stream) {
  if (!h) return tau::fail("tau3d_slab_info: null handle");
  if (z0) *z0 = h->z0;
  if (nzl) *nzl = h->nzl;
  if (nz) *nz = h->p.nz;
  if (device) *device = h->device;
  if (stream) *stream = (void *)h->stream;
  return 0;
}
extern "C" int tau3d_is_split(tau3d_t *h) { return h && h->split ? 1 : 0; }
static int split_buffers(tau3d *h) {   // what the kernel pair needs beside the state: the x/y divergence and k_flux_xy's tile flags
  const tau3d_params *p = &h->p;
  if (h->plane_n * sizeof(float) * 8 > 0xFFFFFFFFull)   // k_update_z: 32-bit byte offsets within a chunk of planes
    return tau::fail("tau3d: %d x %d planes are beyond the split step's 32-bit in-chunk offsets", p->nx, p->ny);
  if (!h->dxy[0]) {
    TAU_HIP(hipMalloc(&h->dxy[0], 6 * h->dxy_stride * sizeof(float)));
    for (int f = 1; f < 6; f++) h->dxy[f] = h->dxy[0] + f * h->dxy_stride;
  }
  if (!h->dzero) {   // written by every k_flux_xy launch for the planes it covers before k_update_z reads them: no initial value needed — zeroed all the same
    const size_t ntiles = (size_t)((p->nx + h3d::XY_FX - 1) / h3d::XY_FX) * ((p->ny + h3d::XY_FY - 1) / h3d::XY_FY) * (size_t)h->nzl;
    TAU_HIP(hipMalloc(&h->dzero, ntiles * sizeof(unsigned)));
    TAU_HIP(hipMemsetAsync(h->dzero, 0, ntiles * sizeof(unsigned), h->stream));
  }
  // the tile list: whole tiles, room for a tile's neighbours and seven planes (a slab: no prediction within three planes of its edges)
  if (!h->uflag[0] && h->tile_list && h->uniform_exits && p->nx % h3d::XY_FX == 0 && p->ny % h3d::XY_FY == 0 &&
      p->nx / h3d::XY_FX >= 3 && p->ny / h3d::XY_FY >= 3 && h->nzl >= 2 * h3d::HALO + 1) {
    const size_t ntiles = (size_t)(p->nx / h3d::XY_FX) * (p->ny / h3d::XY_FY) * (size_t)h->nzl;
    for (int i = 0; i < 2; i++) {
      TAU_HIP(hipMalloc(&h->uflag[i], ntiles * sizeof(unsigned)));
      TAU_HIP(hipMalloc(&h->uref[i], ntiles * h3d::UREC * sizeof(float)));
      TAU_HIP(hipMemsetAsync(h->uflag[i], 0, ntiles * sizeof(unsigned), h->stream));
      TAU_HIP(hipMemsetAsync(h->uref[i], 0, ntiles * h3d::UREC * sizeof(float), h->stream));
    }
    if (h->tile_list == 2) {
      TAU_HIP(hipMalloc(&h->vflag, ntiles * sizeof(unsigned)));
      TAU_HIP(hipMalloc(&h->vref, ntiles * h3d::UREC * sizeof(float)));
    }
    TAU_HIP(hipMalloc(&h->ulist, ntiles * sizeof(unsigned)));
    TAU_HIP(hipMalloc(&h->ucount, 4 * sizeof(unsigned)));
    TAU_HIP(hipMemsetAsync(h->ucount, 0, 4 * sizeof(unsigned), h->stream));
    TAU_HIP(hipHostMalloc((void **)&h->ucount_host, 64, hipHostMallocMapped));
    TAU_HIP(hipHostGetDevicePointer((void **)&h->ucount_host_dev, h->ucount_host, 0));
    *h->ucount_host = 0xFFFFFFFFu;
    for (int i = 0; i < 4; i++) TAU_HIP(hipEventCreateWithFlags(&h->list_ev[i], hipEventDisableTiming));
  }
  if (!h->xyflag && !(getenv("TAU3D_XY_NOFLAGS") && atoi(getenv("TAU3D_XY_NOFLAGS")))) {
    const size_t ntiles = (size_t)((p->nx + h3d::XY_FX - 1) / h3d::XY_FX) * ((p->ny + h3d::XY_FY - 1) / h3d::XY_FY) * (size_t)h->nzl;
    TAU_HIP(hipMalloc(&h->xyflag, ntiles * sizeof(unsigned)));
  }
  return 0;
}
extern "C" int tau3d_set_split(tau3d_t *h, int on) {
  if (!h) return tau::fail("tau3d_set_split: null handle");
  TAU_HIP(hipSetDevice(h->device));
  if (on && !h->split) {
    if (split_buffers(h)) return 1;
    h->split = true;
    return build_solid(h);   // fills the tile flags (the mask itself is rebuilt identically)
  }
  h->split = on != 0;
  return 0;
}
extern "C" int tau3d_max_ptr(tau3d_t *h, float **p) {
  *p = reinterpret_cast<float *>(&h->clk->maxs_bits);
  return 0;
}
extern "C" int tau3d_timing_enable(tau3d_t *h, int on) {
  TAU_HIP(hipSetDevice(h->device));
  if (on && !h->ev_made) {
    for (int i = 0; i < 4096; i++) { TAU_HIP(hipEventCreate(&h->ev0[i])); TAU_HIP(hipEventCreate(&h->ev1[i])); TAU_HIP(hipEventCreate(&h->evm[i])); }
    h->ev_made = true;
  }
  h->timing = on != 0; h->n_ev = 0; h->ev_cells = 0.0;
  return 0;
}
extern "C" int tau3d_timing_read(tau3d_t *h, double *total_ms, int *launches, double *cells) {
  TAU_HIP(hipSetDevice(h->device));
  double tot = 0.0;
  for (int i = 0; i < h->n_ev; i++) {
    float ms = 0.f;
    TAU_HIP(hipEventSynchronize(h->ev1[i]));
    TAU_HIP(hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = h->n_ev;
  if (cells) *cells = h->ev_cells;
  return 0;
}
/* device time from the start of the first timed interval to the end of the last one: the whole timed region as the launch
 * stream saw it (kernels AND the gaps between them), without a host clock or a process barrier in it */
extern "C" int tau3d_timing_span(tau3d_t *h, double *span_ms) {
  if (!h || !span_ms) return tau::fail("tau3d_timing_span: null argument");
  TAU_HIP(hipSetDevice(h->device));
  *span_ms = 0.0;
  if (h->n_ev < 1) return 0;
  float ms = 0.f;
  TAU_HIP(hipEventSynchronize(h->ev1[h->n_ev - 1]));
  TAU_HIP(hipEventElapsedTime(&ms, h->ev0[0], h->ev1[h->n_ev - 1]));
  *span_ms = ms;
  return 0;
}
extern "C" int tau3d_timing_read_split(tau3d_t *h, double *xy_ms, double *z_ms, int *intervals) {
  TAU_HIP(hipSetDevice(h->device));
  double a = 0.0, b = 0.0;
  int n = 0;
  for (int i = 0; i < h->n_ev; i++) {
    if (!h->evm_set[i]) continue;
    float m0 = 0.f, m1 = 0.f;
    TAU_HIP(hipEventSynchronize(h->ev1[i]));
    TAU_HIP(hipEventElapsedTime(&m0, h->ev0[i], h->evm[i]));
    TAU_HIP(hipEventElapsedTime(&m1, h->evm[i], h->ev1[i]));
    a += m0; b += m1; n++;
  }
  if (xy_ms) *xy_ms = a;
  if (z_ms) *z_ms = b;
  if (intervals) *intervals = n;
  return 0;
}
// ---- visualisation (tau_hypersonic_3d_cuda.cu:1715-1739) ----
static int vis_buffers(tau3d *h) {
  if (!h->vis) TAU_HIP(hipMalloc(&h->vis, h->plane_n * (size_t)h->nzl * sizeof(float)));
  if (!h->rgba) TAU_HIP(hipMalloc(&h->rgba, h->plane_n * sizeof(uint32_t)));
  if (!h->scratch) TAU_HIP(hipMalloc(&h->scratch, 4 * sizeof(unsigned)));
  return 0;
}
extern "C" int tau3d_vis_async(tau3d_t *h, int mode, float *out_dev) {
  if (mode < 0 || mode > 7) return tau::fail("tau3d_vis: mode %d outside 0..7", mode);
  TAU_HIP(hipSetDevice(h->device));
  if (vis_buffers(h)) return 1;
  if (h->nzl == h->p.nz && tau3d_fill_halo_periodic_async(h)) return 1;   // slabs: the caller's exchange did this
  h3d::Args A = h->base;
  for (int f = 0; f < 6; f++) { A.in[f] = h->buf[h->cur][f]; A.out[f] = nullptr; }
  float *out = out_dev ? out_dev : h->vis;
  dim3 g((A.nx + 63) / 64, (A.ny + 3) / 4, h->nzl), b(256);
  switch (mode) {
#define TAU_VIS_CASE(M) case M: hipLaunchKernelGGL(h3d::k_vis<M>, g, b, 0, h->stream, A, out); break;
    TAU_VIS_CASE(0) TAU_VIS_CASE(1) TAU_VIS_CASE(2) TAU_VIS_CASE(3) TAU_VIS_CASE(4) TAU_VIS_CASE(5) TAU_VIS_CASE(6) TAU_VIS_CASE(7)
#undef TAU_VIS_CASE
  }
  TAU_LAUNCH_CHECK("k_vis");
  return 0;
}
extern "C" int tau3d_vis(tau3d_t *h, int mode, float *host_out) {
  if (tau3d_vis_async(h, mode, nullptr)) return 1;
  if (host_out)
    TAU_HIP(hipMemcpyAsync(host_out, h->vis, h->plane_n * (size_t)h->nzl * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tau3d_slice_rgba(tau3d_t *h, int zslice, int log_scale, float a_gain, uint32_t *host_rgba, float *mn,
                                float *mx) {
  TAU_HIP(hipSetDevice(h->device));
  if (!h->vis) return tau::fail("tau3d_slice_rgba: no visualisation field yet (call tau3d_vis first)");
  zslice = zslice < 0 ? 0 : (zslice >= h->nzl ? h->nzl - 1 : zslice);   // :1418
  const float *s = h->vis + (size_t)zslice * h->plane_n;
  const int n = (int)h->plane_n;
  const unsigned init[2] = {0xffffffffu, 0u};
  TAU_HIP(hipMemcpyAsync(h->scratch, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
  const int nb = n / 256 < 1 ? 1 : (n / 256 > 1024 ? 1024 : n / 256);
  hipLaunchKernelGGL(h3d::k_slice_minmax, dim3(nb), dim3(256), 0, h->stream, s, n, log_scale, h->scratch);
  TAU_LAUNCH_CHECK("k_slice_minmax");
  hipLaunchKernelGGL(h3d::k_slice_rgba, dim3(nb), dim3(256), 0, h->stream, s, n, log_scale, a_gain,
                     (const unsigned *)h->scratch, h->rgba);
  TAU_LAUNCH_CHECK("k_slice_rgba");
  unsigned keys[2];
  TAU_HIP(hipMemcpyAsync(keys, h->scratch, sizeof(keys), hipMemcpyDeviceToHost, h->stream));
  if (host_rgba) TAU_HIP(hipMemcpyAsync(host_rgba, h->rgba, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  auto unkey = [](unsigned k) { unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &b, 4); return f; };
  if (mn) *mn = unkey(keys[0]);
  if (mx) *mx = unkey(keys[1]);
  return 0;
}
extern "C" int tau3d_palette_indices(tau3d_t *h, float gamma, uint8_t *host_idx, float *mn, float *mx) {
  TAU_HIP(hipSetDevice(h->device));
  if (!h->vis) return tau::fail("tau3d_palette_indices: no visualisation field yet (call tau3d_vis first)");
  const size_t n = h->plane_n * (size_t)h->nzl;
  if (!h->pidx) TAU_HIP(hipMalloc(&h->pidx, n));
  const unsigned init[2] = {0xffffffffu, 0u};
  TAU_HIP(hipMemcpyAsync(h->scratch, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
  const unsigned nb = (unsigned)(n / 256 < 1 ? 1 : (n / 256 > 4096 ? 4096 : n / 256));
  hipLaunchKernelGGL(h3d::k_volume_minmax, dim3(nb), dim3(256), 0, h->stream, (const float *)h->vis, n, h->scratch);
  TAU_LAUNCH_CHECK("k_volume_minmax");
  hipLaunchKernelGGL(h3d::k_palette_index, dim3(nb), dim3(256), 0, h->stream, (const float *)h->vis, n, gamma,
                     (const unsigned *)h->scratch, h->pidx);
  TAU_LAUNCH_CHECK("k_palette_index");
  unsigned keys[2];
  TAU_HIP(hipMemcpyAsync(keys, h->scratch, sizeof(keys), hipMemcpyDeviceToHost, h->stream));
  if (host_idx) TAU_HIP(hipMemcpyAsync(host_idx, h->pidx, n, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  auto unkey = [](unsigned k) { unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &b, 4); return f; };
  if (mn) *mn = unkey(keys[0]);
  if (mx) *mx = unkey(keys[1]);
  return 0;
}
extern "C" int tau3d_outflow_reflection(tau3d_t *h, int nprobe, float *max_dp) {
  TAU_HIP(hipSetDevice(h->device));
  if (vis_buffers(h)) return 1;
  h3d::Args A = h->base;
  for (int f = 0; f < 6; f++) A.in[f] = h->buf[h->cur][f];
  int x0 = A.nx - (nprobe > 1 ? nprobe : 1);
  if (x0 < 0) x0 = 0;
  TAU_HIP(hipMemsetAsync(h->scratch + 2, 0, sizeof(unsigned), h->stream));
  hipLaunchKernelGGL(h3d::k_outflow_reflection, dim3(256), dim3(256), 0, h->stream, A, x0, h->scratch + 2);
  TAU_LAUNCH_CHECK("k_outflow_reflection");
  float v = 0.f;
  TAU_HIP(hipMemcpyAsync(&v, h->scratch + 2, 4, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  if (max_dp) *max_dp = v;
  return 0;
}

extern "C" int tau3d_sync(tau3d_t *h) {
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
#endif  // !TAU3D_SPLIT_TU
