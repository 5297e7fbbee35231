// h2d.hip — 2D compressible Euler step (GPU scheme of the reference) in fp32 for gfx950.
//
// Reference: tau_hypersonic_cuda.cu runs, per step and in fp64, six kernels that round-trip ~520 B
// per cell through HBM (max wavespeed x2, predict -> 16 face-state arrays, x/y flux arrays, update).
// Here the whole step is ONE kernel: a 256-thread workgroup owns a 32x8 tile, stages the
// conserved state with a 2-cell halo in LDS (coalesced 128-B rows), computes the MUSCL-Hancock
// predicted face states of the tile + 1-cell ring into LDS, then each face flux ONCE (low-x /
// low-y face per thread, the 40 far-edge faces in one extra round of the last wave), then the
// conservative update + 4th-order 5-tap diffusion, and finally the max wavespeed of the NEW state
// (one atomicMax per workgroup) so the next step's dt needs no separate reduction pass and no
// host round trip (the reference syncs the host every step, :1846-1850).
// Compulsory traffic: 4 fp32 + 1 u8 in, 4 fp32 out = 33 B/cell.
//
// Semantics kept: inflow column overwrite (k_apply_inflow_left, :772-784) is applied on load, so
// the state arrays at step boundaries equal the reference's; ghost rules of neighbor_or_wall /
// load_neighbor_or_wall_tiled (:266-290, 349-371): inflow left, copy-out right, y clamp, NO-SLIP
// mirror at a masked neighbour; HLLC with its HLLE fallbacks (:483-606); positivity contraction
// (:373-398); repairs (:1166-1173).  Predicted states stay primitive in LDS (the reference
// stores them conserved and converts back: an identity up to rounding that costs fp32 accuracy).

#include "../../include/taueng.h"
#include "tau_common.h"
#include <cmath>
#include <new>
#include <vector>

namespace h2d {

constexpr int TX = 32, TY = 8, NT = TX * TY;
constexpr int UW = TX + 4, UH = TY + 4;   // state tile, halo 2
constexpr int PW = TX + 2, PH = TY + 2;   // predicted-state tile, ring 1
constexpr float EPS_RHO = 1e-25f, EPS_P = 1e-25f; // :32-33

struct P4 { float r, u, v, p; };
struct C4 { float r, mx, my, E; };

struct DevState {
  // max wavespeed (float bits), three slots in rotation: step s reads slot s % 3 (its input state), reduces the
  // state it writes into slot (s+1) % 3 and clears slot (s+2) % 3 for the step after — no kernel between steps
  unsigned maxs_bits[3];
  unsigned pad_;
  double t;
  float dt_last;
  int step;
};

struct Args {
  const float *in[4];
  float *out[4];
  const uint8_t *mask;
  DevState *st;
  int W, H, ntx, nty;
  int slot;                // which maxs slot belongs to the input state
  float dt_explicit;       // > 0: use this dt instead of the CFL one
  float gamma, gm1, inv_gm1, cfl, dt_diff;
  float visc_nu, visc_rho, visc_e;
  float in_r, in_u, in_p;  // inflow_state(), :230-238
  C4 in_c;
  int uniform_exits;       // k_march_lds: trips whose five-row window holds one state skip the predictors and the faces (TAUH2_UNIFORM_EXITS=0: off)
};

__device__ __forceinline__ float vreg(float s) {   // a wave-uniform value moved into a VGPR
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp; the tolerance is 1e-5
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ P4 c2p(const Args &A, C4 c) { // cons_to_prim, :143-158
  P4 p;
  float rho = fmaxf(c.r, EPS_RHO);
  float inv = frcp(rho);
  float u = c.mx * inv, v = c.my * inv;
  float kin = 0.5f * rho * (u * u + v * v);
  p.r = rho; p.u = u; p.v = v;
  p.p = A.gm1 * fmaxf(c.E - kin, EPS_P);
  return p;
}
__device__ __forceinline__ C4 p2c(const Args &A, P4 p) { // prim_to_cons, :160-169
  C4 c;
  float rho = fmaxf(p.r, EPS_RHO), pr = fmaxf(p.p, EPS_P);
  c.r = rho; c.mx = rho * p.u; c.my = rho * p.v;
  c.E = pr * A.inv_gm1 + 0.5f * rho * (p.u * p.u + p.v * p.v);
  return c;
}
__device__ __forceinline__ float sound(const Args &A, P4 p) { // :171-173
  return fsqrt(A.gamma * fmaxf(p.p, EPS_P) * frcp(fmaxf(p.r, EPS_RHO)));
}
__device__ __forceinline__ C4 flux_p(const Args &A, P4 p, C4 c, int ax) { // flux_axis, :193-202
  C4 f;
  float un = ax ? p.v : p.u;
  f.r = ax ? c.my : c.mx;
  f.mx = ax ? (c.mx * un) : (c.mx * un + p.p);
  f.my = ax ? (c.my * un + p.p) : (c.my * un);
  f.E = (c.E + p.p) * un;
  return f;
}
__device__ __forceinline__ float minmod(float a, float b) { // :216-220
  return (a * b <= 0.0f) ? 0.0f : ((fabsf(a) < fabsf(b)) ? a : b);
}
// mc_limiter, :222-227: minmod(minmod(dl,dr), minmod(minmod(dc,2dl), minmod(dc,2dr))).  With dl, dr of one sign
// dc = (q+ - q-)/2 has that sign too and the nest collapses to sign * min(|dl|, |dr|, |dc|) (|dc| only matters
// when rounding puts it an ulp below both); with opposite signs or a zero it is 0.  Same values, 6 ops not 28.
// minmod(a, b) is the median of (a, b, 0) — the smaller magnitude where the signs agree, 0 where they do not or one is 0 — and
// v_med3_f32 is one instruction: two of them instead of min, min, mul, compare, select, sign copy (8 limiters per cell; +3 %).
// (Differs from the product test only where dl * dr underflows to 0, i.e. slopes below 1e-22.)
__device__ __forceinline__ float mc(float dl, float dc, float dr) {
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(dl, dr, 0.0f), dc, 0.0f);
}

// state tile in LDS
struct Tile {
  const float *r, *mx, *my, *E;
  const uint8_t *m;
  int x0, y0; // global coords of tile cell (0,0)  (= tile origin - 2)
  __device__ __forceinline__ int li(int x, int y) const { return (y - y0) * UW + (x - x0); }
  __device__ __forceinline__ C4 cons(int x, int y) const { int i = li(x, y); return C4{r[i], mx[i], my[i], E[i]}; }
  __device__ __forceinline__ bool masked(int x, int y) const { return m[li(x, y)] != 0; }
};

// neighbor_or_wall / load_neighbor_or_wall_tiled, :266-290, 349-371
__device__ __forceinline__ C4 neigh(const Args &A, const Tile &T, P4 center, int xn, int yn) {
  yn = max(0, min(yn, A.H - 1));
  if (xn < 0) return A.in_c;
  if (xn >= A.W) return T.cons(A.W - 1, yn);
  if (T.masked(xn, yn)) return p2c(A, P4{center.r, -center.u, -center.v, center.p}); // wall_ghost_prim: no-slip
  return T.cons(xn, yn);
}

// ONE staging rule for both step kernels (the tile kernel's LDS staging and the march's row loads) and for the
// known-answer check of tau_hypersonic_cuda_tests.cu:567-640 (k_unit_neighbors below):
struct MCell { C4 c; bool m, in; };   // staged conserved state, body mask, "a cell of the domain" (for has-state tests)

// Loads: scalar base of the field at row `row0` (wave-uniform, <= every row the caller asks for) + 32-bit lane offset
// (tau_common.h): a caller touches a band of at most ~60 rows, so the offsets fit whatever the grid (tauh2_create bounds
// W).  The inflow state is taken into registers FIRST: written as `q.c = loaded; if (..) q.c = A.in_c;` hipcc selected
// between the two ADDRESSES (global memory / a stack copy of the kernel argument) and issued flat_load through a generic
// pointer.
__device__ __forceinline__ MCell march_load(const Args &A, int gx, int row, int row0) {
  MCell q;
  float ir = A.in_c.r, imx = A.in_c.mx, imy = A.in_c.my, iE = A.in_c.E;
  asm volatile("" : "+s"(ir), "+s"(imx), "+s"(imy), "+s"(iE));
  const int sx = max(0, min(gx, A.W - 1)), sy = max(0, min(row, A.H - 1));
  const size_t rb = (size_t)row0 * A.W;
  const unsigned gi = (unsigned)((sy - row0) * A.W + sx);
  const unsigned g4 = tau::lane_off(gi << 2);
  const bool mk = (A.mask + rb)[gi] != 0;
  q.m = (gx < 0 || gx >= A.W) ? false : mk;
  const float c0 = tau::gld((const tau::GChar *)(A.in[0] + rb), g4), c1 = tau::gld((const tau::GChar *)(A.in[1] + rb), g4),
              c2 = tau::gld((const tau::GChar *)(A.in[2] + rb), g4), c3 = tau::gld((const tau::GChar *)(A.in[3] + rb), g4);
  const bool inflow = (sx == 0 && !mk) || gx < 0;
  q.c = C4{inflow ? ir : c0, inflow ? imx : c1, inflow ? imy : c2, inflow ? iE : c3};
  q.in = gx >= 0 && gx < A.W && row >= 0 && row < A.H;
  return q;
}

// The same function without branches, for the twelve calls per cell of the predictor and the diffusion stencil.
// Staging has already resolved the two x cases — halo columns left of x = 0 hold the inflow state, columns right of
// x = W-1 hold cell W-1, both with the mask cleared — and clamps y like the reference's tile load; the caller forms
// the wall ghost of its centre cell once.  What is left is one select per component (the branchy form above cost
// ~36 exec-mask branches per cell).
__device__ __forceinline__ C4 wall_ghost(const Args &A, P4 center) { return p2c(A, P4{center.r, -center.u, -center.v, center.p}); }
__device__ __forceinline__ C4 neigh_sel(const Args &A, const Tile &T, const C4 &wg, int xn, int yn) {
  yn = max(0, min(yn, A.H - 1));
  const C4 c = T.cons(xn, yn);
  const bool use_wall = T.masked(xn, yn);
  return C4{use_wall ? wg.r : c.r, use_wall ? wg.mx : c.mx, use_wall ? wg.my : c.my, use_wall ? wg.E : c.E};
}

__device__ __forceinline__ void enforce_positive(P4 &qm, const P4 &qc, P4 &qp) { // :373-398
  for (int it = 0; it < 8; it++) {
    bool bad = (qm.r <= EPS_RHO) || (qp.r <= EPS_RHO) || (qm.p <= EPS_P) || (qp.p <= EPS_P);
    if (!bad) return;
    qm.r = 0.5f * (qm.r + qc.r); qm.u = 0.5f * (qm.u + qc.u); qm.v = 0.5f * (qm.v + qc.v); qm.p = 0.5f * (qm.p + qc.p);
    qp.r = 0.5f * (qp.r + qc.r); qp.u = 0.5f * (qp.u + qc.u); qp.v = 0.5f * (qp.v + qc.v); qp.p = 0.5f * (qp.p + qc.p);
  }
  qm.r = fmaxf(qm.r, EPS_RHO); qp.r = fmaxf(qp.r, EPS_RHO);
  qm.p = fmaxf(qm.p, EPS_P);   qp.p = fmaxf(qp.p, EPS_P);
}

__device__ __forceinline__ P4 half_step(const Args &A, P4 q, C4 dF, float h) { // :442-455 (+ :931-934 floors)
  C4 c = p2c(A, q);
  c.r -= h * dF.r; c.mx -= h * dF.mx; c.my -= h * dF.my; c.E -= h * dF.E;
  P4 o = c2p(A, c);
  o.r = fmaxf(o.r, EPS_RHO);
  o.p = fmaxf(o.p, EPS_P);
  return o;
}

// MUSCL-Hancock predicted low / high face states of one cell along one axis, :911-961
__device__ __forceinline__ void predict_axis(const Args &A, const Tile &T, P4 qc, const C4 &wg, int x, int y, int ax, float half,
                                             P4 &lo, P4 &hi) {
  const int dx = ax ? 0 : 1, dy = ax ? 1 : 0;
  P4 qm = c2p(A, neigh_sel(A, T, wg, x - dx, y - dy));
  P4 qp = c2p(A, neigh_sel(A, T, wg, x + dx, y + dy));
  float s_r = mc(qc.r - qm.r, 0.5f * (qp.r - qm.r), qp.r - qc.r);
  float s_u = mc(qc.u - qm.u, 0.5f * (qp.u - qm.u), qp.u - qc.u);
  float s_v = mc(qc.v - qm.v, 0.5f * (qp.v - qm.v), qp.v - qc.v);
  float s_p = mc(qc.p - qm.p, 0.5f * (qp.p - qm.p), qp.p - qc.p);
  P4 L{qc.r - 0.5f * s_r, qc.u - 0.5f * s_u, qc.v - 0.5f * s_v, qc.p - 0.5f * s_p};
  P4 R{qc.r + 0.5f * s_r, qc.u + 0.5f * s_u, qc.v + 0.5f * s_v, qc.p + 0.5f * s_p};
  enforce_positive(L, qc, R);
  // flux_axis(prim) = flux of prim_to_cons(prim) whose own cons_to_prim floors rho and p
  P4 Lf{fmaxf(L.r, EPS_RHO), L.u, L.v, fmaxf(L.p, EPS_P)}, Rf{fmaxf(R.r, EPS_RHO), R.u, R.v, fmaxf(R.p, EPS_P)};
  C4 FL = flux_p(A, Lf, p2c(A, L), ax), FR = flux_p(A, Rf, p2c(A, R), ax);
  C4 dF{FR.r - FL.r, FR.mx - FL.mx, FR.my - FL.my, FR.E - FL.E};
  lo = half_step(A, L, dF, half);
  hi = half_step(A, R, dF, half);
}

__device__ __forceinline__ C4 hlle(const Args &A, P4 L, P4 R, C4 UL, C4 UR, C4 FL, C4 FR, float SL, float SR) { // :483-509
  float denom = SR - SL;
  if (fabsf(denom) < 1e-14f) return C4{0.5f * (FL.r + FR.r), 0.5f * (FL.mx + FR.mx), 0.5f * (FL.my + FR.my), 0.5f * (FL.E + FR.E)};
  float id = frcp(denom), s = SL * SR;
  return C4{id * (SR * FL.r - SL * FR.r + s * (UR.r - UL.r)), id * (SR * FL.mx - SL * FR.mx + s * (UR.mx - UL.mx)),
            id * (SR * FL.my - SL * FR.my + s * (UR.my - UL.my)), id * (SR * FL.E - SL * FR.E + s * (UR.E - UL.E))};
}

// HLLC with HLLE fallbacks on every degeneracy, :519-606.  L, R are primitive (floored).
__device__ __forceinline__ C4 hllc(const Args &A, P4 L, P4 R, int ax) {
  L.r = fmaxf(L.r, EPS_RHO); R.r = fmaxf(R.r, EPS_RHO);
  L.p = fmaxf(L.p, EPS_P);   R.p = fmaxf(R.p, EPS_P);
  float unL = ax ? L.v : L.u, unR = ax ? R.v : R.u, utL = ax ? L.u : L.v, utR = ax ? R.u : R.v;
  float aL = sound(A, L), aR = sound(A, R);
  float SL = fminf(unL - aL, unR - aR), SR = fmaxf(unL + aL, unR + aR);
  // (a wave-uniform supersonic exit before the right state's flux was measured in round 5 and not kept: 60.07 against 60.61 Gcell/s, profiles/r05/ab2d_euler_hllc.txt)
  C4 UL = p2c(A, L), UR = p2c(A, R);
  C4 FL = flux_p(A, L, UL, ax), FR = flux_p(A, R, UR, ax);
  // (round 5, measured and not kept: the ladder of returns below flattened into one straight-line star-state path with a single
  //  degeneracy flag and one HLLE evaluation per wave that needs it — 57.1 against 60.0 Gcell/s at 4096^2: the returns, divergent as
  //  they are, let the supersonic x faces skip the star state; profiles/r05/ab2d_euler_hllc.txt)
  if (SL >= 0.0f) return FL;
  if (SR <= 0.0f) return FR;
  float num = R.p - L.p + L.r * unL * (SL - unL) - R.r * unR * (SR - unR);
  float den = L.r * (SL - unL) - R.r * (SR - unR);
  if (fabsf(den) < 1e-14f || !isfinite(num) || !isfinite(den)) return hlle(A, L, R, UL, UR, FL, FR, SL, SR);
  float SM = num * frcp(den);
  if (!isfinite(SM)) return hlle(A, L, R, UL, UR, FL, FR, SL, SR);
  float pStar = fmaxf(L.p + L.r * (SL - unL) * (SM - unL), EPS_P);
  float dLS = SL - SM, dRS = SR - SM;
  if (fabsf(dLS) < 1e-14f || fabsf(dRS) < 1e-14f) return hlle(A, L, R, UL, UR, FL, FR, SL, SR);
  const float idL = frcp(dLS), idR = frcp(dRS);
  float rsL = L.r * (SL - unL) * idL, rsR = R.r * (SR - unR) * idR;
  if (!(rsL > 0.0f) || !(rsR > 0.0f) || !isfinite(rsL) || !isfinite(rsR)) return hlle(A, L, R, UL, UR, FL, FR, SL, SR);
  float EsL = ((SL - unL) * UL.E - L.p * unL + pStar * SM) * idL;
  float EsR = ((SR - unR) * UR.E - R.p * unR + pStar * SM) * idR;
  if (!isfinite(EsL) || !isfinite(EsR)) return hlle(A, L, R, UL, UR, FL, FR, SL, SR);
  const bool left = SM >= 0.0f;
  float rs = left ? rsL : rsR, ut = left ? utL : utR, Es = left ? EsL : EsR, S = left ? SL : SR;
  C4 UK = left ? UL : UR, FK = left ? FL : FR;
  float mN = rs * SM, mT = rs * ut;
  C4 US = ax ? C4{rs, mT, mN, Es} : C4{rs, mN, mT, Es};
  return C4{FK.r + S * (US.r - UK.r), FK.mx + S * (US.mx - UK.mx), FK.my + S * (US.my - UK.my), FK.E + S * (US.E - UK.E)};
}

struct PredTile { // predicted states, primitive, in LDS; index by global coords
  float *q;       // [16][PH*PW]: xlo(4) xhi(4) ylo(4) yhi(4)
  int x0, y0;     // global coords of pred cell (0,0) (= tile origin - 1)
  __device__ __forceinline__ int li(int x, int y) const { return (y - y0) * PW + (x - x0); }
  __device__ __forceinline__ P4 get(int slot, int x, int y) const {
    int i = li(x, y);
    const float *b = q + slot * 4 * (PH * PW);
    return P4{b[i], b[PH * PW + i], b[2 * PH * PW + i], b[3 * PH * PW + i]};
  }
  __device__ __forceinline__ void put(int slot, int x, int y, P4 p) {
    int i = li(x, y);
    float *b = q + slot * 4 * (PH * PW);
    b[i] = p.r; b[PH * PW + i] = p.u; b[2 * PH * PW + i] = p.v; b[3 * PH * PW + i] = p.p;
  }
};

// flux through the face between (xa,ya) [low side] and (xb,yb) [high side], :964-1030
__device__ __forceinline__ C4 face(const Args &A, const Tile &T, const PredTile &Q, int xa, int ya, int xb, int yb, int ax) {
  const bool hasL = (xa >= 0) && (ya >= 0) && !T.masked(xa, ya);
  const bool hasR = (xb < A.W) && (yb < A.H) && !T.masked(xb, yb);
  P4 L, R;
  if (hasL && hasR) {
    L = Q.get(1, xa, ya);
    R = Q.get(0, xb, yb);
  } else if (hasR) {
    L = c2p(A, neigh(A, T, c2p(A, T.cons(xb, yb)), xa, ya));
    R = Q.get(0, xb, yb);
  } else if (hasL) {
    L = Q.get(1, xa, ya);
    R = c2p(A, neigh(A, T, c2p(A, T.cons(xa, ya)), xb, yb));
  } else {
    return C4{0.f, 0.f, 0.f, 0.f};
  }
  return hllc(A, L, R, ax);
}

__device__ __forceinline__ float cell_speed(const Args &A, C4 c) { // k_max_wavespeed_blocks, :794-802
  P4 p = c2p(A, c);
  float a = sound(A, p);
  float v = fmaxf(fabsf(p.u) + a, fabsf(p.v) + a);
  return isfinite(v) ? v : 1e-12f;
}

// (NT, 7): 22.5 KB of LDS allow 7 workgroups per CU; the kernel needs 73 VGPRs, one more than 7 waves/SIMD
// allow — asking for 7 costs a single spilled register and buys the seventh wave (0.485 -> 0.474 ms)
__global__ __launch_bounds__(NT, 7) void k_step(const Args A) {
  __shared__ float sU[4][UH * UW];
  __shared__ uint8_t sM[UH * UW];
  __shared__ float sQ[8 * PH * PW];     // predicted low / high face states of ONE axis at a time
  __shared__ float sF[4][TY + 1][TX + 1];   // face fluxes of the axis in flight (x: [ty][tx], tx <= TX; y: [ty][tx], ty <= TY)
  __shared__ float sRed[NT / 64];

  const int tid = threadIdx.x, tx = tid & (TX - 1), ty = tid >> 5, lane = tid & 63, wave = tid >> 6;
  unsigned b = tau::xcd_swizzle(blockIdx.x, (unsigned)(A.ntx * A.nty));
  const int bx0 = (int)(b % (unsigned)A.ntx) * TX, by0 = (int)(b / (unsigned)A.ntx) * TY;
  const int x = bx0 + tx, y = by0 + ty;

  // dt: convective limit from the max wavespeed of the input state, capped by the diffusion limit (:1856-1865)
  float dt;
  if (A.dt_explicit > 0.f) dt = A.dt_explicit;
  else {
    float maxs = __uint_as_float(A.st->maxs_bits[A.slot]);
    if (!isfinite(maxs) || maxs < 1e-12f) maxs = 1e-12f;
    dt = fminf(A.cfl / maxs, A.dt_diff);
  }
  const float half = 0.5f * dt;
  if (blockIdx.x == 0 && tid == 0) { // the step's bookkeeping (was a 1-thread kernel of its own): sim_t += dt, :1888
    A.st->t += (double)dt;
    A.st->dt_last = dt;
    A.st->step += 1;
    A.st->maxs_bits[(A.slot + 2) % 3] = 0u;
  }

  // ---- A: stage the conserved state, halo 2, clamped like the reference's tile load (:877-901);
  //         the inflow column overwrite of k_apply_inflow_left happens here
  for (int t = tid; t < UH * UW; t += NT) {
    const int ly = t / UW, lx = t - ly * UW;
    // columns outside the domain: left = the inflow state, right = a copy of cell W-1; neither is ever "wall"
    // (neighbor_or_wall tests x before the mask, :266-290); rows clamp: march_load is that rule
    const MCell q = march_load(A, bx0 - 2 + lx, by0 - 2 + ly, max(by0 - 2, 0));
    sU[0][t] = q.c.r; sU[1][t] = q.c.mx; sU[2][t] = q.c.my; sU[3][t] = q.c.E;
    sM[t] = q.m ? 1 : 0;
  }
  __syncthreads();
  const Tile T{sU[0], sU[1], sU[2], sU[3], sM, bx0 - 2, by0 - 2};
  PredTile Q{sQ, bx0 - 1, by0 - 1};

  // ---- B/C, one axis at a time: predicted face states of tile + ring along the axis (8 floats per cell in LDS
  //      instead of 16 -> 27 KB per workgroup, 5 workgroups per CU instead of 4), then that axis' face fluxes
  const bool row_ok = y < A.H, col_ok = x < A.W;
  C4 dFx{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ax = 0; ax < 2; ax++) {
    for (int t = tid; t < PH * PW; t += NT) {
      const int py = t / PW, px = t - py * PW;
      const int cx = bx0 - 1 + px, cy = by0 - 1 + py;
      if (cx < 0 || cx >= A.W || cy < 0 || cy >= A.H) continue;
      if (T.masked(cx, cy)) continue;
      const P4 qc = c2p(A, T.cons(cx, cy));
      const C4 wg = wall_ghost(A, qc);
      P4 lo, hi;
      predict_axis(A, T, qc, wg, cx, cy, ax, half, lo, hi);
      Q.put(0, cx, cy, lo); Q.put(1, cx, cy, hi);
    }
    __syncthreads();
    if (ax == 0) {
      if (row_ok && x <= A.W) { // low-x face of (x,y): fx = x
        C4 F = face(A, T, Q, x - 1, y, x, y, 0);
        sF[0][ty][tx] = F.r; sF[1][ty][tx] = F.mx; sF[2][ty][tx] = F.my; sF[3][ty][tx] = F.E;
      }
      if (wave == NT / 64 - 1 && lane < TY) { // far edge: 8 x-faces at column TX
        const int gx = bx0 + TX, gy = by0 + lane;
        if (gy < A.H && gx <= A.W) {
          C4 F = face(A, T, Q, gx - 1, gy, gx, gy, 0);
          sF[0][lane][TX] = F.r; sF[1][lane][TX] = F.mx; sF[2][lane][TX] = F.my; sF[3][lane][TX] = F.E;
        }
      }
    } else {
      if (col_ok && y <= A.H) { // low-y face of (x,y): fy = y
        C4 F = face(A, T, Q, x, y - 1, x, y, 1);
        sF[0][ty][tx] = F.r; sF[1][ty][tx] = F.mx; sF[2][ty][tx] = F.my; sF[3][ty][tx] = F.E;
      }
      if (wave == NT / 64 - 1 && lane < TX) { // far edge: 32 y-faces at row TY
        const int gx = bx0 + lane, gy = by0 + TY;
        if (gx < A.W && gy <= A.H) {
          C4 F = face(A, T, Q, gx, gy - 1, gx, gy, 1);
          sF[0][TY][lane] = F.r; sF[1][TY][lane] = F.mx; sF[2][TY][lane] = F.my; sF[3][TY][lane] = F.E;
        }
      }
    }
    __syncthreads();
    if (ax == 0) { // the x flux difference goes to registers: the y faces reuse the same LDS (next barrier is in between)
      dFx = C4{sF[0][ty][tx + 1] - sF[0][ty][tx], sF[1][ty][tx + 1] - sF[1][ty][tx], sF[2][ty][tx + 1] - sF[2][ty][tx],
               sF[3][ty][tx + 1] - sF[3][ty][tx]};
    }
  }

  // ---- D: update + separable 4th-order diffusion + repairs, :1096-1175
  float smax = 0.f;
  if (row_ok && col_ok) {
    const size_t gi = (size_t)y * A.W + x;
    const C4 Uc = T.cons(x, y);
    C4 Un = Uc;
    if (!T.masked(x, y)) {
      Un.r -= dt * dFx.r; Un.mx -= dt * dFx.mx; Un.my -= dt * dFx.my; Un.E -= dt * dFx.E;
      Un.r -= dt * (sF[0][ty + 1][tx] - sF[0][ty][tx]); Un.mx -= dt * (sF[1][ty + 1][tx] - sF[1][ty][tx]);
      Un.my -= dt * (sF[2][ty + 1][tx] - sF[2][ty][tx]); Un.E -= dt * (sF[3][ty + 1][tx] - sF[3][ty][tx]);
      const C4 wg = wall_ghost(A, c2p(A, Uc));
      const C4 xm2 = neigh_sel(A, T, wg, x - 2, y), xm1 = neigh_sel(A, T, wg, x - 1, y), xp1 = neigh_sel(A, T, wg, x + 1, y), xp2 = neigh_sel(A, T, wg, x + 2, y);
      const C4 ym2 = neigh_sel(A, T, wg, x, y - 2), ym1 = neigh_sel(A, T, wg, x, y - 1), yp1 = neigh_sel(A, T, wg, x, y + 1), yp2 = neigh_sel(A, T, wg, x, y + 2);
      const float i12 = 1.0f / 12.0f;
#define D2(f) (((-xm2.f + 16.0f * xm1.f - 30.0f * Uc.f + 16.0f * xp1.f - xp2.f) * i12) + \
               ((-ym2.f + 16.0f * ym1.f - 30.0f * Uc.f + 16.0f * yp1.f - yp2.f) * i12))
      Un.r += (A.visc_rho * dt) * D2(r);
      Un.mx += (A.visc_nu * dt) * D2(mx);
      Un.my += (A.visc_nu * dt) * D2(my);
      Un.E += (A.visc_e * dt) * D2(E);
#undef D2
      Un.r = fmaxf(Un.r, EPS_RHO);
      P4 pp = c2p(A, Un);
      if (pp.p <= EPS_P || !isfinite(pp.p) || !isfinite(pp.r) || !isfinite(pp.u) || !isfinite(pp.v)) {
        pp.r = fmaxf(pp.r, EPS_RHO);
        pp.p = fmaxf(pp.p, EPS_P);
        Un = p2c(A, pp);
      }
      // wavespeed of the new state as the NEXT step will see it (its inflow column overwritten)
      smax = cell_speed(A, (x == 0) ? A.in_c : Un);
    }
    A.out[0][gi] = Un.r; A.out[1][gi] = Un.mx; A.out[2][gi] = Un.my; A.out[3][gi] = Un.E;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 64));
  if (lane == 0) sRed[wave] = smax;
  __syncthreads();
  if (tid == 0) {
    float m = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
    tau::atomic_max_float_bits(&A.st->maxs_bits[(A.slot + 1) % 3], m);
  }
}

// -------------------------------------------------------------------------------------------------
// The same step as a MARCH (large grids).  A wave owns a 60-column strip (64 lanes, two halo lanes a side: the
// diffusion stencil and "predicted state of the neighbour" both reach two cells) and walks down a chunk of rows
// holding a five-row window of the input state in registers.  Per trip it takes in row a, predicts row a-1 along
// both axes (MUSCL-Hancock, x neighbours by lane shift), forms the x-face fluxes of row a-1 and the y-face fluxes
// between rows a-2 and a-1, and completes row a-2: flux differences, 4th-order diffusion from the window, repairs,
// store, wavespeed.  Nothing goes through LDS memory and every face is evaluated once per wave; the tile kernel
// above stages 432 cells and predicts 340 to write 256.  The boundary rules are the ones the tile staging resolves
// (columns left of x = 0 hold the inflow state, columns right of W-1 a copy of cell W-1, rows are clamped, all
// with the mask cleared / taken from the clamped cell), applied where a row is loaded.
constexpr int MCOLS = 60;
#if !defined(TAU_EXPERIMENT) && (defined(TAU_H2_VREG_CONSTS) || defined(TAU_H2_LDS_WAVES))
#error "TAU_H2_* tuning overrides need -DTAU_EXPERIMENT (scripts/variant_build_file.sh sets it)"
#endif
#ifndef TAU_H2_VREG_CONSTS
#define TAU_H2_VREG_CONSTS 2   // gas constants (1) and dt (2) of the LDS-window march in VGPRs: +0.5 % (round 4; 0 = SGPR operands)
#endif
#ifndef TAU_H2_LDS_WAVES
#define TAU_H2_LDS_WAVES 4
#endif
__device__ __forceinline__ MCell lane_shift(const MCell &q, int d) {   // the cell d lanes away (own value at the wave's ends)
  MCell o;
  if (d > 0) {   // from the lane below
    o.c = C4{__shfl_up(q.c.r, d, 64), __shfl_up(q.c.mx, d, 64), __shfl_up(q.c.my, d, 64), __shfl_up(q.c.E, d, 64)};
    const int f = __shfl_up((int)q.m | ((int)q.in << 1), d, 64);
    o.m = f & 1; o.in = (f >> 1) & 1;
  } else {
    o.c = C4{__shfl_down(q.c.r, -d, 64), __shfl_down(q.c.mx, -d, 64), __shfl_down(q.c.my, -d, 64), __shfl_down(q.c.E, -d, 64)};
    const int f = __shfl_down((int)q.m | ((int)q.in << 1), -d, 64);
    o.m = f & 1; o.in = (f >> 1) & 1;
  }
  return o;
}
__device__ __forceinline__ C4 ghost_sel(const C4 &wg, const MCell &n) {   // neigh_sel on a staged cell
  return C4{n.m ? wg.r : n.c.r, n.m ? wg.mx : n.c.mx, n.m ? wg.my : n.c.my, n.m ? wg.E : n.c.E};
}
// predict_axis on explicit neighbours (already ghost-selected, primitive)
__device__ __forceinline__ void predict_from(const Args &A, P4 qc, const P4 &qm, const P4 &qp, int ax, float half, P4 &lo, P4 &hi) {
  float s_r = mc(qc.r - qm.r, 0.5f * (qp.r - qm.r), qp.r - qc.r);
  float s_u = mc(qc.u - qm.u, 0.5f * (qp.u - qm.u), qp.u - qc.u);
  float s_v = mc(qc.v - qm.v, 0.5f * (qp.v - qm.v), qp.v - qc.v);
  float s_p = mc(qc.p - qm.p, 0.5f * (qp.p - qm.p), qp.p - qc.p);
  P4 L{qc.r - 0.5f * s_r, qc.u - 0.5f * s_u, qc.v - 0.5f * s_v, qc.p - 0.5f * s_p};
  P4 R{qc.r + 0.5f * s_r, qc.u + 0.5f * s_u, qc.v + 0.5f * s_v, qc.p + 0.5f * s_p};
  enforce_positive(L, qc, R);
  P4 Lf{fmaxf(L.r, EPS_RHO), L.u, L.v, fmaxf(L.p, EPS_P)}, Rf{fmaxf(R.r, EPS_RHO), R.u, R.v, fmaxf(R.p, EPS_P)};
  C4 FL = flux_p(A, Lf, p2c(A, L), ax), FR = flux_p(A, Rf, p2c(A, R), ax);
  C4 dF{FR.r - FL.r, FR.mx - FL.mx, FR.my - FL.my, FR.E - FL.E};
  lo = half_step(A, L, dF, half);
  hi = half_step(A, R, dF, half);
}
// face() on explicit cells: a = low side (predicted high state ha), b = high side (predicted low state lb)
__device__ __forceinline__ C4 face_from(const Args &A, const MCell &a, const P4 &ha, const MCell &b, const P4 &lb, int ax) {
  const bool hasL = a.in && !a.m, hasR = b.in && !b.m;
  P4 L = ha, R = lb;
  // ghost states only where a side has no predicted state (body surface, domain edge): a real branch, skipped by
  // every wave away from those (as selects the two wall ghosts and four conversions ran for every face)
  if (__builtin_amdgcn_ballot_w64(!hasL || !hasR) != 0ull) {   // wave-uniform: some lane of the wave needs a ghost
    const P4 gl = c2p(A, ghost_sel(wall_ghost(A, c2p(A, b.c)), a)), gr = c2p(A, ghost_sel(wall_ghost(A, c2p(A, a.c)), b));
    if (!hasL) L = gl;
    if (!hasR) R = gr;
  }
  C4 F = hllc(A, L, R, ax);
  if (!hasL && !hasR) F = C4{0.f, 0.f, 0.f, 0.f};
  return F;
}

__global__ __launch_bounds__(256) void k_march(const Args A, int rows, int nstrips, int nchunks) {
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(nstrips * nchunks);
  const unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  float dt;
  if (A.dt_explicit > 0.f) dt = A.dt_explicit;
  else {
    float maxs = __uint_as_float(A.st->maxs_bits[A.slot]);
    if (!isfinite(maxs) || maxs < 1e-12f) maxs = 1e-12f;
    dt = fminf(A.cfl / maxs, A.dt_diff);
  }
  const float half = 0.5f * dt;
  if (blockIdx.x == 0 && threadIdx.x == 0) { // the step's bookkeeping: sim_t += dt, :1888
    A.st->t += (double)dt;
    A.st->dt_last = dt;
    A.st->step += 1;
    A.st->maxs_bits[(A.slot + 2) % 3] = 0u;
  }
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)nstrips), chunk = (int)(wid / (unsigned)nstrips);
  const int gx = strip * MCOLS + lane - 2;
  const bool own = lane >= 2 && lane < 2 + MCOLS && gx < A.W;
  const int j0 = chunk * rows, j1 = min(j0 + rows, A.H);

  MCell w0, w1, w2, w3, w4;                       // input rows a-4 .. a
  const int row0 = __builtin_amdgcn_readfirstlane(max(j0 - 2, 0));   // the band's first row: base of every load / store offset
  w2 = march_load(A, gx, j0 - 2, row0);           // (the loop's first slide makes these rows a-4, a-3 = j0-3?, see below)
  w3 = march_load(A, gx, j0 - 2, row0);
  w4 = march_load(A, gx, j0 - 1, row0);
  w0 = w2; w1 = w2;
  P4 yhi_prev{1.f, 0.f, 0.f, 1.f};                // predicted high-y state of row a-2
  C4 Gy_lo{0.f, 0.f, 0.f, 0.f};                   // y-face flux below row a-2 (between a-3 and a-2)
  C4 dFx{0.f, 0.f, 0.f, 0.f};                     // x flux difference of row a-2
  float smax = 0.f;
  const float in_sp = cell_speed(A, A.in_c);      // the inflow column's speed (its state is overwritten on load)
  MCell nxt = march_load(A, gx, j0, row0);
  P4 q3 = c2p(A, w3.c), q4 = c2p(A, w4.c), q2 = q3;   // primitives of rows a-2, a-1, a (after the slide): each row is converted once
  for (int a = j0; a <= j1 + 1; a++) {
    w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = nxt; // window = rows a-4 .. a
    q2 = q3; q3 = q4; q4 = c2p(A, w4.c);
    if (a < j1 + 1) nxt = march_load(A, gx, a + 1, row0);
    // ---- predict row p = a-1 (centre w3) along x and y
    const P4 qc = q3;
    const MCell l1 = lane_shift(w3, 1), r1 = lane_shift(w3, -1);
    // the x neighbours' primitives come by lane shift too (they are the neighbours' own qc), the y neighbours' are carried
    P4 pl{__shfl_up(qc.r, 1, 64), __shfl_up(qc.u, 1, 64), __shfl_up(qc.v, 1, 64), __shfl_up(qc.p, 1, 64)};
    P4 pr{__shfl_down(qc.r, 1, 64), __shfl_down(qc.u, 1, 64), __shfl_down(qc.v, 1, 64), __shfl_down(qc.p, 1, 64)};
    P4 pd = q2, pu = q4;
    if (__builtin_amdgcn_ballot_w64(l1.m | r1.m | w2.m | w4.m) != 0ull) {   // a masked neighbour is seen as the wall ghost of the centre (rare: wave-uniform branch)
      const C4 wg = wall_ghost(A, qc);
      if (l1.m) pl = c2p(A, wg);
      if (r1.m) pr = c2p(A, wg);
      if (w2.m) pd = c2p(A, wg);
      if (w4.m) pu = c2p(A, wg);
    }
    P4 xlo, xhi, ylo, yhi;
    predict_from(A, qc, pl, pr, 0, half, xlo, xhi);
    predict_from(A, qc, pd, pu, 1, half, ylo, yhi);
    // ---- x-face fluxes of row p: low face from the lane below, high face = the low face of the lane above
    P4 xhi_l;
    xhi_l.r = __shfl_up(xhi.r, 1, 64); xhi_l.u = __shfl_up(xhi.u, 1, 64); xhi_l.v = __shfl_up(xhi.v, 1, 64); xhi_l.p = __shfl_up(xhi.p, 1, 64);
    const C4 Fx = face_from(A, l1, xhi_l, w3, xlo, 0);
    const C4 dFx_p{__shfl_down(Fx.r, 1, 64) - Fx.r, __shfl_down(Fx.mx, 1, 64) - Fx.mx, __shfl_down(Fx.my, 1, 64) - Fx.my, __shfl_down(Fx.E, 1, 64) - Fx.E};
    // ---- y-face flux between rows a-2 (w2) and a-1 (w3)
    const C4 Gy = face_from(A, w2, yhi_prev, w3, ylo, 1);
    // ---- complete row j = a-2 (centre w2): update + separable 4th-order diffusion + repairs, :1096-1175
    const int j = a - 2;
    if (j >= j0 && j < j1) {   // wave-uniform
      const C4 Uc = w2.c;
      C4 Un = Uc;
      float sp = 0.f;
      const MCell xm1 = lane_shift(w2, 1), xm2 = lane_shift(w2, 2), xp1 = lane_shift(w2, -1), xp2 = lane_shift(w2, -2);
      if (!w2.m) {
        Un.r -= dt * dFx.r; Un.mx -= dt * dFx.mx; Un.my -= dt * dFx.my; Un.E -= dt * dFx.E;
        Un.r -= dt * (Gy.r - Gy_lo.r); Un.mx -= dt * (Gy.mx - Gy_lo.mx);
        Un.my -= dt * (Gy.my - Gy_lo.my); Un.E -= dt * (Gy.E - Gy_lo.E);
        C4 cxm2 = xm2.c, cxm1 = xm1.c, cxp1 = xp1.c, cxp2 = xp2.c, cym2 = w0.c, cym1 = w1.c, cyp1 = w3.c, cyp2 = w4.c;
        if (__builtin_amdgcn_ballot_w64(xm2.m | xm1.m | xp1.m | xp2.m | w0.m | w1.m | w3.m | w4.m) != 0ull) {
          const C4 wgc = wall_ghost(A, c2p(A, Uc));
          cxm2 = ghost_sel(wgc, xm2); cxm1 = ghost_sel(wgc, xm1); cxp1 = ghost_sel(wgc, xp1); cxp2 = ghost_sel(wgc, xp2);
          cym2 = ghost_sel(wgc, w0); cym1 = ghost_sel(wgc, w1); cyp1 = ghost_sel(wgc, w3); cyp2 = ghost_sel(wgc, w4);
        }
        const float i12 = 1.0f / 12.0f;
#define D2(f) (((-cxm2.f + 16.0f * cxm1.f - 30.0f * Uc.f + 16.0f * cxp1.f - cxp2.f) * i12) + \
               ((-cym2.f + 16.0f * cym1.f - 30.0f * Uc.f + 16.0f * cyp1.f - cyp2.f) * i12))
        Un.r += (A.visc_rho * dt) * D2(r);
        Un.mx += (A.visc_nu * dt) * D2(mx);
        Un.my += (A.visc_nu * dt) * D2(my);
        Un.E += (A.visc_e * dt) * D2(E);
#undef D2
        Un.r = fmaxf(Un.r, EPS_RHO);
        P4 pp = c2p(A, Un);
        if (pp.p <= EPS_P || !isfinite(pp.p) || !isfinite(pp.r) || !isfinite(pp.u) || !isfinite(pp.v)) {
          pp.r = fmaxf(pp.r, EPS_RHO);
          pp.p = fmaxf(pp.p, EPS_P);
          Un = p2c(A, pp);
        }
        // wavespeed of the new state as the next step will see it; pp IS cons_to_prim(Un) (re-floored where repaired)
        const float ca = sound(A, pp), cv = fmaxf(fabsf(pp.u) + ca, fabsf(pp.v) + ca);
        sp = (gx == 0) ? in_sp : (isfinite(cv) ? cv : 1e-12f);
      }
      if (own) {
        const size_t rb = (size_t)row0 * A.W;
        const unsigned g4 = tau::lane_off((unsigned)((j - row0) * A.W + gx) << 2);
        tau::gst((tau::GChar *)(A.out[0] + rb), g4, Un.r); tau::gst((tau::GChar *)(A.out[1] + rb), g4, Un.mx);
        tau::gst((tau::GChar *)(A.out[2] + rb), g4, Un.my); tau::gst((tau::GChar *)(A.out[3] + rb), g4, Un.E);
        smax = fmaxf(smax, sp);
      }
    }
    yhi_prev = yhi; Gy_lo = Gy; dFx = dFx_p;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 64));
  if (lane == 0) tau::atomic_max_float_bits(&A.st->maxs_bits[(A.slot + 1) % 3], smax);
}

// The same march with the five-row input window in LDS instead of registers (one private [5 rows][5 words][64 lanes] ring
// per wave, 6.4 KB: rows are written once and read where they are used, a neighbour d lanes away is an LDS read at
// lane + d instead of a shuffle, and nothing slides): the register window cost the kernel ~30 VGPRs and 25 moves per
// trip and held it at 164 VGPRs = three waves per SIMD, where a wave's ~6-cycle issue interval (profiles/r02/
// valu_calib.txt) leaves the VALU idle whenever one of the three stalls.
struct MRing {
  float (*w)[5][64];   // [slot][r, mx, my, E, flags][lane]
  int lane;
  __device__ __forceinline__ void put(int slot, const MCell &q) const {
    w[slot][0][lane] = q.c.r; w[slot][1][lane] = q.c.mx; w[slot][2][lane] = q.c.my; w[slot][3][lane] = q.c.E;
    w[slot][4][lane] = __int_as_float((int)q.m | ((int)q.in << 1));
    // get() / fld() of OTHER lanes read what this lane just wrote: the wave barrier (no instruction, a scheduling fence) keeps
    // the compiler from ever moving such a read above the stores, whatever its alias analysis concludes about constant slots
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ MCell get(int slot, int d) const {   // the cell d lanes below (d > 0) / above; own value at the wave's ends
    const int l = min(max(lane - d, 0), 63);
    MCell q;
    q.c = C4{w[slot][0][l], w[slot][1][l], w[slot][2][l], w[slot][3][l]};
    const int f = __float_as_int(w[slot][4][l]);
    q.m = f & 1; q.in = (f >> 1) & 1;
    return q;
  }
  __device__ __forceinline__ C4 cons(int slot) const { return C4{w[slot][0][lane], w[slot][1][lane], w[slot][2][lane], w[slot][3][lane]}; }
  __device__ __forceinline__ bool flag(int slot, int d) const { return __float_as_int(w[slot][4][min(max(lane - d, 0), 63)]) & 1; }
  __device__ __forceinline__ float fld(int slot, int f, int ln, int d) const { return w[slot][f][min(max(ln - d, 0), 63)]; }
};
// UEX: with the uniform-region exits (below).  Their bookkeeping costs the kernel six registers it does not have (122 -> 128 VGPRs and
// 44 B of scratch: 60.3 -> 57.4 Gcell/s where no trip is skipped), so the kernel without them is its own instantiation
// (TAUH2_UNIFORM_EXITS=0), untouched.
template <int WPB, bool UEX>   // WPB waves per workgroup: the waves of a workgroup share nothing, WPB only sets the granularity of dispatch
__global__ __launch_bounds__(64 * WPB, TAU_H2_LDS_WAVES) void k_march_lds(const Args A0, int rows, int nstrips, int nchunks, const int *__restrict__ crow) {
  __shared__ float sW[WPB][5][5][64];
#if TAU_H2_VREG_CONSTS >= 1
  Args A = A0;   // the gas constants in VGPRs: a VALU instruction with an SGPR operand issues at half rate (profiles/r02/valu_calib.txt)
  A.gamma = vreg(A0.gamma); A.gm1 = vreg(A0.gm1); A.inv_gm1 = vreg(A0.inv_gm1);
#else
  const Args &A = A0;
#endif
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(nstrips * nchunks);
  const unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * WPB + (threadIdx.x >> 6);
  float dt;
  if (A.dt_explicit > 0.f) dt = A.dt_explicit;
  else {
    float maxs = __uint_as_float(A.st->maxs_bits[A.slot]);
    if (!isfinite(maxs) || maxs < 1e-12f) maxs = 1e-12f;
    dt = fminf(A.cfl / maxs, A.dt_diff);
  }
#if TAU_H2_VREG_CONSTS >= 2
  dt = vreg(dt);
#endif
  const float half = 0.5f * dt;
  if (blockIdx.x == 0 && threadIdx.x == 0) { // the step's bookkeeping: sim_t += dt, :1888
    A.st->t += (double)dt;
    A.st->dt_last = dt;
    A.st->step += 1;
    A.st->maxs_bits[(A.slot + 2) % 3] = 0u;
  }
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)nstrips), chunk = (int)(wid / (unsigned)nstrips);
  const int gx = strip * MCOLS + lane - 2;
  const bool own = lane >= 2 && lane < 2 + MCOLS && gx < A.W;
  // chunk c = rows [crow[c], crow[c + 1]) where the launch hands a schedule of chunk lengths (h2_launch_step), else `rows` each
  const int j0 = crow ? crow[chunk] : chunk * rows, j1 = crow ? crow[chunk + 1] : min(j0 + rows, A.H);
  if (j0 >= j1) return;

  const int row0 = __builtin_amdgcn_readfirstlane(max(j0 - 2, 0));   // the band's first row: base of every load / store offset
  const MRing R{sW[threadIdx.x >> 6], lane};
  // slots: row a sits in slot s4, rows a-1 .. a-4 in s3 .. s0 (a ring of five, rotated once per trip).  Before the first
  // trip rows j0-2, j0-2, j0-1 stand in for a-3 .. a-1 as in k_march (the first two rows a trip completes are j0, j0+1).
  int s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4;
  {
    const MCell a2 = march_load(A, gx, j0 - 2, row0), a1 = march_load(A, gx, j0 - 1, row0);
    R.put(s1, a2); R.put(s2, a2); R.put(s3, a2); R.put(s4, a1);
  }
  P4 yhi_prev{1.f, 0.f, 0.f, 1.f};                // predicted high-y state of row a-2
  C4 Gy_lo{0.f, 0.f, 0.f, 0.f};                   // y-face flux below row a-2 (between a-3 and a-2)
  C4 dFx{0.f, 0.f, 0.f, 0.f};                     // x flux difference of row a-2
  float smax = 0.f;
  const float in_sp = cell_speed(A, A.in_c);      // the inflow column's speed (its state is overwritten on load)
  MCell nxt = march_load(A, gx, j0, row0);
  P4 q3 = c2p(A, R.cons(s3)), q4 = c2p(A, R.cons(s4)), q2 = q3;   // primitives of rows a-2, a-1, a (after the slide): each row is converted once
  // Uniform-region exits (round 6; the 3D step's are described in h3d.hip: flux_xy_core).  ucnt = how many of the newest rows of the
  // window are FLAT — the same conserved state in all 64 lanes, every lane a cell of the domain outside the body — and equal to the
  // row before them.  With five, the whole stencil of row a-2's update holds one state: the x fluxes of a row are one value (their
  // difference +0), the y fluxes below and above the row are one value (difference +0), and predictors and faces are not evaluated;
  // the update itself (diffusion stencil, repairs, wavespeed) runs as always — its operands are what the full path would hand it.
  // What the skipped trips would have left for the next one (the predicted high-y state of row a-2, the y flux below it) is
  // recomputed from that one state when the stretch ends (`stale`): the same calls on the same operands.
  int ucnt = 0;
  bool stale = false;
  float ur0 = 0.f, ur1 = 0.f, ur2 = 0.f, ur3 = 0.f;
  for (int a = j0; a <= j1 + 1; a++) {
    { const int t = s0; s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = t; }   // window = rows a-4 .. a in slots s0 .. s4
    R.put(s4, nxt);
    q2 = q3; q3 = q4; q4 = c2p(A, nxt.c);
    const bool m4 = nxt.m;
    if (UEX) {   // row a enters the window
      auto first = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };   // (the builtin takes an int: bits, not a value conversion)
      const float r0 = first(nxt.c.r), r1 = first(nxt.c.mx), r2 = first(nxt.c.my), r3 = first(nxt.c.E);
      // BIT patterns, in the lanes and against the row before (as in the 3D step since its sign-of-zero find: +0 == -0 as numbers, and
      // m_y = -0 beside +0 is a different operand for anything that looks at a sign)
      auto u = [](float v) { return __float_as_uint(v); };
      const unsigned dif = (u(nxt.c.r) ^ u(r0)) | (u(nxt.c.mx) ^ u(r1)) | (u(nxt.c.my) ^ u(r2)) | (u(nxt.c.E) ^ u(r3));
      const bool flat = __builtin_amdgcn_ballot_w64((dif != 0u) | nxt.m | !nxt.in) == 0ull;
      const bool same = ((u(r0) ^ u(ur0)) | (u(r1) ^ u(ur1)) | (u(r2) ^ u(ur2)) | (u(r3) ^ u(ur3))) == 0u;
      ucnt = flat ? ((same && ucnt > 0) ? ucnt + 1 : 1) : 0;
      ur0 = r0; ur1 = r1; ur2 = r2; ur3 = r3;
    }
    if (a < j1 + 1) nxt = march_load(A, gx, a + 1, row0);
    // ---- predict row p = a-1 (centre w3) along x and y
    const P4 qc = q3;
    const MCell w3 = R.get(s3, 0), l1 = R.get(s3, 1), r1 = R.get(s3, -1);
    const MCell w2 = R.get(s2, 0);
    // the x neighbours' primitives come by lane shift too (they are the neighbours' own qc), the y neighbours' are carried.
    // (ds_bpermute shuffles, not DPP moves: a DPP move is a half-rate VALU instruction and the kernel is short of VALU issue,
    // not of LDS issue — 50.0 against 49.4 Gcell/s, and the same sign in the Burgers / LBM marches)
    P4 pl{__shfl_up(qc.r, 1, 64), __shfl_up(qc.u, 1, 64), __shfl_up(qc.v, 1, 64), __shfl_up(qc.p, 1, 64)};
    P4 pr{__shfl_down(qc.r, 1, 64), __shfl_down(qc.u, 1, 64), __shfl_down(qc.v, 1, 64), __shfl_down(qc.p, 1, 64)};
    P4 pd = q2, pu = q4;
    if (__builtin_amdgcn_ballot_w64(l1.m | r1.m | w2.m | m4) != 0ull) {   // a masked neighbour is seen as the wall ghost of the centre (rare: wave-uniform branch)
      const C4 wg = wall_ghost(A, qc);
      if (l1.m) pl = c2p(A, wg);
      if (r1.m) pr = c2p(A, wg);
      if (w2.m) pd = c2p(A, wg);
      if (m4) pu = c2p(A, wg);
    }
    // The chunk's first trip predicts row j0-1 and its last one row j1: of those rows only the y states are read (by the
    // y faces below row j0 / above row j1-1) — their x predictor and x faces, and the y face below row j0-1, are skipped
    // (wave-uniform branches; with ~16-row chunks the two warm-up trips were 11 % of the kernel).
    const bool row_x = a > j0 && a <= j1;
    const int j = a - 2;
    // the whole window one state, and an interior trip of the chunk (rows a-4 .. a all loaded as themselves, row a-2 completed here)
    const bool uni = UEX && ucnt >= 5 && row_x && j >= j0 && j < j1;
    P4 xlo{1.f, 0.f, 0.f, 1.f}, xhi{1.f, 0.f, 0.f, 1.f}, ylo{1.f, 0.f, 0.f, 1.f}, yhi{1.f, 0.f, 0.f, 1.f};
    C4 dFx_p{0.f, 0.f, 0.f, 0.f};
    C4 Gy{0.f, 0.f, 0.f, 0.f};
    if (uni) {
      // nothing to evaluate: dFx of row a-2 is +0 (one x flux in every lane), Gy - Gy_lo is +0 (one y flux either side of row a-2)
      dFx = C4{0.f, 0.f, 0.f, 0.f}; Gy_lo = C4{0.f, 0.f, 0.f, 0.f};
      stale = true;
    } else {
    if (UEX && stale) {   // the stretch has ended: rows a-4 .. a-1 still hold its state; what the last (skipped) trip would have left behind
      P4 lo_u, hi_u;
      predict_from(A, q2, q2, q2, 1, half, lo_u, hi_u);                 // row a-2 along y, between two rows equal to itself
      yhi_prev = hi_u;
      Gy_lo = face_from(A, R.get(s1, 0), hi_u, w2, lo_u, 1);            // the y face between rows a-3 and a-2 (row a-3 predicts the same states)
      dFx = C4{0.f, 0.f, 0.f, 0.f};
      stale = false;
    }
    if (row_x) {
      predict_from(A, qc, pl, pr, 0, half, xlo, xhi);
    }
    predict_from(A, qc, pd, pu, 1, half, ylo, yhi);
    // ---- x-face fluxes of row p: low face from the lane below, high face = the low face of the lane above
    if (row_x) {
      P4 xhi_l;
      xhi_l.r = __shfl_up(xhi.r, 1, 64); xhi_l.u = __shfl_up(xhi.u, 1, 64); xhi_l.v = __shfl_up(xhi.v, 1, 64); xhi_l.p = __shfl_up(xhi.p, 1, 64);
      const C4 Fx = face_from(A, l1, xhi_l, w3, xlo, 0);
      dFx_p = C4{__shfl_down(Fx.r, 1, 64) - Fx.r, __shfl_down(Fx.mx, 1, 64) - Fx.mx, __shfl_down(Fx.my, 1, 64) - Fx.my, __shfl_down(Fx.E, 1, 64) - Fx.E};
    }
    // ---- y-face flux between rows a-2 (w2) and a-1 (w3)
    if (a > j0) Gy = face_from(A, w2, yhi_prev, w3, ylo, 1);
    }
    // ---- complete row j = a-2 (centre w2): update + separable 4th-order diffusion + repairs, :1096-1175
    if (j >= j0 && j < j1) {   // wave-uniform
      const C4 Uc = w2.c;
      C4 Un = Uc;
      float sp = 0.f;
      if (!w2.m) {
        Un.r -= dt * dFx.r; Un.mx -= dt * dFx.mx; Un.my -= dt * dFx.my; Un.E -= dt * dFx.E;
        Un.r -= dt * (Gy.r - Gy_lo.r); Un.mx -= dt * (Gy.mx - Gy_lo.mx);
        Un.my -= dt * (Gy.my - Gy_lo.my); Un.E -= dt * (Gy.E - Gy_lo.E);
        // 4th-order diffusion, ONE FIELD AT A TIME: the eight neighbours of a field are read from the ring where they are
        // used (all 32 values at once were the register peak of the kernel); a body neighbour is the wall ghost of the centre
        const bool bxm2 = R.flag(s2, 2), bxm1 = R.flag(s2, 1), bxp1 = R.flag(s2, -1), bxp2 = R.flag(s2, -2);
        const bool bym2 = R.flag(s0, 0), bym1 = R.flag(s1, 0), byp1 = w3.m, byp2 = m4;
        const bool anyb = __builtin_amdgcn_ballot_w64(bxm2 | bxm1 | bxp1 | bxp2 | bym2 | bym1 | byp1 | byp2) != 0ull;
        C4 wgc{0.f, 0.f, 0.f, 0.f};
        if (anyb) wgc = wall_ghost(A, c2p(A, Uc));
        const float i12 = 1.0f / 12.0f;
        int ls = lane;
        auto d2 = [&](int f, float uc, float wg) -> float {
          float xm2 = R.fld(s2, f, ls, 2), xm1 = R.fld(s2, f, ls, 1), xp1 = R.fld(s2, f, ls, -1), xp2 = R.fld(s2, f, ls, -2);
          float ym2 = R.fld(s0, f, ls, 0), ym1 = R.fld(s1, f, ls, 0), yp1 = R.fld(s3, f, ls, 0), yp2 = R.fld(s4, f, ls, 0);
          if (anyb) {
            xm2 = bxm2 ? wg : xm2; xm1 = bxm1 ? wg : xm1; xp1 = bxp1 ? wg : xp1; xp2 = bxp2 ? wg : xp2;
            ym2 = bym2 ? wg : ym2; ym1 = bym1 ? wg : ym1; yp1 = byp1 ? wg : yp1; yp2 = byp2 ? wg : yp2;
          }
          float r = ((-xm2 + 16.0f * xm1 - 30.0f * uc + 16.0f * xp1 - xp2) * i12) + ((-ym2 + 16.0f * ym1 - 30.0f * uc + 16.0f * yp1 - yp2) * i12);
          asm volatile("" : "+v"(r), "+v"(ls));
          return r;
        };
        Un.r += (A.visc_rho * dt) * d2(0, Uc.r, wgc.r);
        Un.mx += (A.visc_nu * dt) * d2(1, Uc.mx, wgc.mx);
        Un.my += (A.visc_nu * dt) * d2(2, Uc.my, wgc.my);
        Un.E += (A.visc_e * dt) * d2(3, Uc.E, wgc.E);
        Un.r = fmaxf(Un.r, EPS_RHO);
        P4 pp = c2p(A, Un);
        if (pp.p <= EPS_P || !isfinite(pp.p) || !isfinite(pp.r) || !isfinite(pp.u) || !isfinite(pp.v)) {
          pp.r = fmaxf(pp.r, EPS_RHO);
          pp.p = fmaxf(pp.p, EPS_P);
          Un = p2c(A, pp);
        }
        // wavespeed of the new state as the next step will see it; pp IS cons_to_prim(Un) (re-floored where repaired)
        const float ca = sound(A, pp), cv = fmaxf(fabsf(pp.u) + ca, fabsf(pp.v) + ca);
        sp = (gx == 0) ? in_sp : (isfinite(cv) ? cv : 1e-12f);
      }
      if (own) {
        const size_t rb = (size_t)row0 * A.W;
        const unsigned g4 = tau::lane_off((unsigned)((j - row0) * A.W + gx) << 2);
        tau::gst((tau::GChar *)(A.out[0] + rb), g4, Un.r); tau::gst((tau::GChar *)(A.out[1] + rb), g4, Un.mx);
        tau::gst((tau::GChar *)(A.out[2] + rb), g4, Un.my); tau::gst((tau::GChar *)(A.out[3] + rb), g4, Un.E);
        smax = fmaxf(smax, sp);
      }
    }
    yhi_prev = yhi; Gy_lo = Gy; dFx = dFx_p;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 64));
  if (lane == 0) tau::atomic_max_float_bits(&A.st->maxs_bits[(A.slot + 1) % 3], smax);
}

// max wavespeed of a freshly initialised / uploaded state (the reference's two reduction kernels)
__global__ __launch_bounds__(256) void k_maxspeed(const Args A) {
  __shared__ float sRed[4];
  const size_t n = (size_t)A.W * A.H;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (A.mask[i]) continue;
    C4 c{A.in[0][i], A.in[1][i], A.in[2][i], A.in[3][i]};
    if (i % A.W == 0) c = A.in_c;
    m = fmaxf(m, cell_speed(A, c));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
    tau::atomic_max_float_bits(&A.st->maxs_bits[A.slot], m);
  }
}

// ---- device self-test: the kernel's own helpers evaluated on the known answers of the
// reference's unit tests (tau_hypersonic_cuda_tests.cu:245-346), fp32
__global__ void k_unit(const Args A, float *out) {
  { P4 q = c2p(A, p2c(A, P4{1.4f, 2.2f, -0.7f, 3.6f})); out[0] = q.r; out[1] = q.u; out[2] = q.v; out[3] = q.p; }      // :245-253
  { C4 c = p2c(A, P4{-2.0f, 1.5f, -0.5f, -7.0f}); P4 q = c2p(A, C4{1.0f, 3.0f, 4.0f, 1e-20f});                          // :255-264
    out[4] = c.r; out[5] = c.E; out[6] = q.r; out[7] = q.p; }
  out[8] = minmod(1.0f, 2.0f); out[9] = minmod(-1.0f, 2.0f); out[10] = mc(1.0f, 1.2f, 1.5f); out[11] = mc(-1.0f, 0.2f, 1.0f); // :266-271
  { P4 p{2.0f, 3.0f, -4.0f, 5.0f}; C4 U = p2c(A, p); C4 Fx = flux_p(A, p, U, 0), Fy = flux_p(A, p, U, 1);               // :273-288
    out[12] = Fx.r; out[13] = Fx.mx; out[14] = Fx.my; out[15] = Fx.E; out[16] = Fy.r; out[17] = Fy.mx; out[18] = Fy.my;
    out[19] = Fy.E; out[20] = sound(A, p); }
  out[21] = A.in_r; out[22] = A.in_u; out[23] = 0.f; out[24] = A.in_p;                                                     // :290-296
  { P4 p = c2p(A, p2c(A, P4{1.4f, 2.2f, -0.7f, 3.6f})); C4 U = p2c(A, p);                                                  // :298-314
    for (int ax = 0; ax < 2; ax++) { C4 F = hllc(A, p, p, ax), G = flux_p(A, p, U, ax);
      out[25 + 4 * ax] = F.r - G.r; out[26 + 4 * ax] = F.mx - G.mx; out[27 + 4 * ax] = F.my - G.my; out[28 + 4 * ax] = F.E - G.E; } }
  { P4 qc{1.0f, 4.0f, -2.0f, 1.0f}, qm{-1.0f, 8.0f, -4.0f, -3.0f}, qp{-2.0f, -8.0f, 4.0f, -2.0f}; enforce_positive(qm, qc, qp); // :316-326
    out[33] = qm.r; out[34] = qm.p; out[35] = qp.r; out[36] = qp.p; }
  { P4 qc{1.0f, 2.0f, -1.0f, 1.0f}, qm{0.8f, 2.2f, -0.9f, 1.1f}, qp{1.2f, 1.8f, -1.2f, 0.9f}; enforce_positive(qm, qc, qp);   // :328-338
    out[37] = qm.r; out[38] = qm.p; out[39] = qp.r; out[40] = qp.p; }
  { C4 g = p2c(A, P4{1.0f, -3.0f, 0.5f, 1.0f}); C4 w = p2c(A, P4{1.0f, 3.0f, -0.5f, 1.0f}); out[41] = g.mx + w.mx; out[42] = g.my + w.my; } // no-slip ghost, :262-264
}

// the hand-built-field neighbour lookups of tau_hypersonic_cuda_tests.cu:348-371, 567-640 on the engine's own staging
// rule: a neighbour is march_load() (inflow left of x = 0, copy of cell W-1 right of it, clamped rows) and, where that
// is a body cell, the no-slip ghost of the centre — exactly what both step kernels evaluate.  The centre is read raw:
// the reference test does not run k_apply_inflow_left before its lookups, the step kernels apply it on load.
__global__ void k_unit_neighbors(const Args A, int x, int y, float *out) {
  const size_t ic = (size_t)y * A.W + x;
  const C4 wg = wall_ghost(A, c2p(A, C4{A.in[0][ic], A.in[1][ic], A.in[2][ic], A.in[3][ic]}));
  // row0 = the (clamped) row each lookup reads: march_load's 32-bit offsets are relative to it and stay small on any grid
  // (with row0 = 0 they span the whole grid and wrap from 2^30 cells on)
  auto rowc = [&](int r) { return max(0, min(r, A.H - 1)); };
  const C4 left = ghost_sel(wg, march_load(A, x - 1, y, rowc(y))), right = ghost_sel(wg, march_load(A, x + 1, y, rowc(y))),
           up = ghost_sel(wg, march_load(A, x, y + 1, rowc(y + 1))), top = ghost_sel(wg, march_load(A, x, A.H + 20, rowc(A.H + 20)));
  out[0] = left.r; out[1] = left.mx; out[2] = right.r; out[3] = right.mx; out[4] = up.mx;     // k_test_neighbors
  out[5] = left.r; out[6] = left.mx; out[7] = up.mx; out[8] = top.r;                          // k_test_neighbor_for_diff
}

// ---------------------------------------------------------------- rendering
// k_render_vals / k_reduce_minmax / k_compute_inv_range / k_render_pixels, tau_hypersonic_cuda.cu:1178-1334:
// a scalar per fluid cell (7 view modes), its min/max over the fluid, then the blue-green-red ramp.
// Here: one pass writes the scalar and folds the wave's min/max into two order-preserving integer keys
// (no block arrays, no second reduction kernel); one pass maps to pixels.
__device__ __forceinline__ unsigned fkey(float v) { unsigned b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float funkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ P4 sample_prim_bc(const Args &A, int xc, int yc, int x, int y) { // :706-727
  y = max(0, min(y, A.H - 1));
  if (x < 0) return P4{A.in_r, A.in_u, 0.f, A.in_p};
  auto ld = [&](int i) { return c2p(A, C4{A.in[0][i], A.in[1][i], A.in[2][i], A.in[3][i]}); };
  if (x >= A.W) return ld(y * A.W + (A.W - 1));
  const int i = y * A.W + x;
  if (A.mask[i]) { const P4 c = ld(yc * A.W + xc); return P4{c.r, -c.u, -c.v, c.p}; }   // wall_ghost_prim
  return ld(i);
}

__global__ __launch_bounds__(256) void k_render_vals(const Args A, int view_mode, float *__restrict__ val, unsigned *mm) {
  const int N = A.W * A.H;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float mn = 3.0e38f, mx = -3.0e38f;
  if (i < N) {
    float v = 0.f;
    if (!A.mask[i]) {
      const int x = i % A.W, y = i / A.W;
      const P4 p = c2p(A, C4{A.in[0][i], A.in[1][i], A.in[2][i], A.in[3][i]});
      if (view_mode == 0) v = logf(p.r);
      else if (view_mode == 1) v = logf(p.p);
      else if (view_mode == 2) v = sqrtf(p.u * p.u + p.v * p.v);
      else if (view_mode == 3) {
        const float gx = 0.5f * (sample_prim_bc(A, x, y, x + 1, y).r - sample_prim_bc(A, x, y, x - 1, y).r);
        const float gy = 0.5f * (sample_prim_bc(A, x, y, x, y + 1).r - sample_prim_bc(A, x, y, x, y - 1).r);
        v = logf(1e-12f + sqrtf(gx * gx + gy * gy));
      } else if (view_mode == 4) {
        const float dv_dx = 0.5f * (sample_prim_bc(A, x, y, x + 1, y).v - sample_prim_bc(A, x, y, x - 1, y).v);
        const float du_dy = 0.5f * (sample_prim_bc(A, x, y, x, y + 1).u - sample_prim_bc(A, x, y, x, y - 1).u);
        v = asinhf(dv_dx - du_dy);
      } else if (view_mode == 5) {
        v = sqrtf(p.u * p.u + p.v * p.v) / fmaxf(sqrtf(A.gamma * fmaxf(p.p, EPS_P) / fmaxf(p.r, EPS_RHO)), 1e-30f);
      } else {
        v = logf(fmaxf(p.p / fmaxf(p.r, EPS_RHO), 1e-30f));
      }
      if (!isfinite(v)) v = 0.f;
      mn = v; mx = v;
    }
    val[i] = v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0 && mn <= mx) { atomicMin(&mm[0], fkey(mn)); atomicMax(&mm[1], fkey(mx)); }
}

__global__ __launch_bounds__(256) void k_render_pixels(const uint8_t *__restrict__ mask, const float *__restrict__ val, int N,
                                                       const unsigned *mm, uint32_t *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  if (mask[i]) { out[i] = 0xff000000u | (110u << 16) | (110u << 8) | 110u; return; }   // pack_rgba(110,110,110)
  const float mn = funkey(mm[0]), mx = funkey(mm[1]);
  const float inv = 1.0f / fmaxf(mx - mn, 1e-30f);                                       // k_compute_inv_range
  float t = (val[i] - mn) * inv;
  t = fminf(fmaxf(t, 0.f), 1.f);                                                         // get_color, :692-704
  const float rr = 255.0f * fminf(1.0f, fmaxf(0.0f, 3.0f * t - 1.0f));
  const float gg = 255.0f * fminf(1.0f, fmaxf(0.0f, 2.0f - 4.0f * fabsf(t - 0.5f)));
  const float bb = 255.0f * fminf(1.0f, fmaxf(0.0f, 2.0f - 3.0f * t));
  out[i] = 0xff000000u | ((uint32_t)(uint8_t)bb << 16) | ((uint32_t)(uint8_t)gg << 8) | (uint32_t)(uint8_t)rr;
}

} // namespace h2d

// =====================================================================================
// C-ABI
// =====================================================================================
struct tauh2 {
  tauh2_params p;
  int device;
  hipStream_t stream;
  bool own_stream;
  float *buf[2][4];
  uint8_t *mask;
  h2d::DevState *st;
  int cur;
  int slot = 0;      // max slot of the current state (DevState::maxs_bits)
  bool maxs_valid;
  h2d::Args base;
  float *rval;              // render scalar per cell (lazy)
  uint32_t *rpix;           // render pixels (lazy)
  unsigned *rmm;            // min / max keys
  int *crow = nullptr;      // chunk schedule of the march (h2_schedule)
  int crow_n = 0;
  bool uniform_exits = true;   // TAUH2_UNIFORM_EXITS=0 at tauh2_create: every trip of the march evaluates its predictors and faces (same bits)
};

namespace {
// host fp64 geometry, expression for expression the reference's k_init (:625-686, 729-770), so the
// mask is bit-identical to the reference's
double sdSegment(double px, double py, double ax, double ay, double bx, double by) {
  double abx = bx - ax, aby = by - ay, apx = px - ax, apy = py - ay;
  double denom = abx * abx + aby * aby + 1e-30;
  double t = (apx * abx + apy * aby) / denom;
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  double qx = ax + t * abx, qy = ay + t * aby;
  return sqrt((px - qx) * (px - qx) + (py - qy) * (py - qy));
}
double sdBody(double x, double y, double Rb, double Rn, double theta) {
  double r = fabs(y);
  double st = sin(theta), ct = cos(theta), tt = tan(theta);
  double xt = Rn * (1.0 - st), rt = Rn * ct;
  double xb = xt + (Rb - rt) / fmax(tt, 1e-30);
  double rprof;
  if (x < 0.0) rprof = -1.0;
  else if (x <= xt) { double dx = x - Rn, in = Rn * Rn - dx * dx; rprof = (in > 0.0) ? sqrt(in) : 0.0; }
  else if (x <= xb) rprof = rt + (x - xt) * tt;
  else rprof = -1.0;
  int inside = (x >= 0.0 && x <= xb && r <= rprof);
  double d = fabs(sqrt((x - Rn) * (x - Rn) + r * r) - Rn);
  double d_cone = sdSegment(x, r, xt, rt, xb, Rb), d_base = sdSegment(x, y, xb, -Rb, xb, +Rb);
  double d_rim = sqrt((x - xb) * (x - xb) + (r - Rb) * (r - Rb));
  if (d_cone < d) d = d_cone;
  if (d_base < d) d = d_base;
  if (d_rim < d) d = d_rim;
  return inside ? -d : d;
}
} // namespace

extern "C" void tauh2_params_default(tauh2_params *c, int W, int H) { // default_config, :1394-1409
  c->W = W; c->H = H;
  c->gamma = 1.1; c->cfl = 0.25; c->visc_nu = 5e-2; c->visc_rho = 5e-2; c->visc_e = 2e-2;
  c->mach = 25.0; c->geom_x0 = 125.0; c->geom_cy = (double)H / 2.0;
  c->geom_rb = (double)H / 12.0; c->geom_rn = (double)H / 24.0; c->geom_theta = 3.14159265358979323846 / 4.0;
}

static void h2_consts(tauh2 *h) {
  const tauh2_params &P = h->p;
  h2d::Args &A = h->base;
  memset(&A, 0, sizeof(A));
  A.W = P.W; A.H = P.H;
  // the last x / y face of the domain (fx = W, fy = H) is the far edge of the last tile: plain ceil suffices
  A.ntx = (P.W + h2d::TX - 1) / h2d::TX; A.nty = (P.H + h2d::TY - 1) / h2d::TY;
  A.gamma = (float)P.gamma; A.gm1 = (float)(P.gamma - 1.0); A.inv_gm1 = (float)(1.0 / (P.gamma - 1.0));
  A.cfl = (float)P.cfl;
  double nu_max = fmax(P.visc_nu, fmax(P.visc_rho, P.visc_e));
  A.dt_diff = (std::isfinite(nu_max) && nu_max > 1e-12) ? (float)(0.25 / nu_max) : 3.0e38f;
  A.visc_nu = (float)P.visc_nu; A.visc_rho = (float)P.visc_rho; A.visc_e = (float)P.visc_e;
  double a = sqrt(P.gamma * 1.0 / 1.0), u = P.mach * a;
  A.in_r = 1.0f; A.in_u = (float)u; A.in_p = 1.0f;
  A.in_c.r = 1.0f; A.in_c.mx = (float)(1.0 * u); A.in_c.my = 0.0f;
  A.in_c.E = (float)(1.0 / (P.gamma - 1.0) + 0.5 * 1.0 * (u * u));
  A.mask = h->mask; A.st = h->st;
}

extern "C" int tauh2_create(tauh2_t **out, const tauh2_params *p, int device, void *stream) {
  if (!out || !p) return tau::fail("tauh2_create: null argument");
  if (p->W < 8 || p->H < 8) return tau::fail("tauh2_create: grid must be at least 8x8");
  if (p->W > (1 << 24))   // the kernels address their band of rows (<= 60) with 32-bit byte offsets
    return tau::fail("tauh2_create: W = %d is beyond the kernels' 32-bit in-band offsets (2^24 columns)", p->W);
  if (!(p->gamma > 1.0)) return tau::fail("tauh2_create: gamma must be > 1");
  TAU_HIP(hipSetDevice(device));
  tauh2 *h = new (std::nothrow) tauh2();
  if (!h) return tau::fail("tauh2_create: out of host memory");
  tau::HandleGuard<tauh2> guard{h, tauh2_destroy};
  h->p = *p; h->device = device; h->cur = 0; h->maxs_valid = false;
  if (const char *e = getenv("TAUH2_UNIFORM_EXITS")) h->uniform_exits = atoi(e) != 0;
  h->own_stream = (stream == nullptr);
  if (h->own_stream) TAU_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  else h->stream = (hipStream_t)stream;
  size_t n = (size_t)p->W * p->H;
  for (int s = 0; s < 2; s++)
    for (int f = 0; f < 4; f++) TAU_HIP(hipMalloc(&h->buf[s][f], n * sizeof(float)));
  TAU_HIP(hipMalloc(&h->mask, n));
  TAU_HIP(hipMalloc(&h->st, sizeof(h2d::DevState)));
  TAU_HIP(hipMemsetAsync(h->st, 0, sizeof(h2d::DevState), h->stream));
  TAU_HIP(hipMemsetAsync(h->mask, 0, n, h->stream));
  h2_consts(h);
  *out = guard.release();
  return 0;
}
extern "C" void tauh2_destroy(tauh2_t *h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  for (int s = 0; s < 2; s++)
    for (int f = 0; f < 4; f++) hipFree(h->buf[s][f]);
  hipFree(h->mask); hipFree(h->st);
  hipFree(h->rval); hipFree(h->rpix); hipFree(h->rmm); hipFree(h->crow);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int tauh2_upload(tauh2_t *h, const float *const host[4], const uint8_t *mask) {
  TAU_HIP(hipSetDevice(h->device));
  size_t n = (size_t)h->p.W * h->p.H;
  for (int f = 0; f < 4; f++)
    TAU_HIP(hipMemcpyAsync(h->buf[h->cur][f], host[f], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if (mask) TAU_HIP(hipMemcpyAsync(h->mask, mask, n, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  h->maxs_valid = false;
  return 0;
}
extern "C" int tauh2_download(tauh2_t *h, float *const host[4], uint8_t *mask) {
  TAU_HIP(hipSetDevice(h->device));
  size_t n = (size_t)h->p.W * h->p.H;
  for (int f = 0; f < 4; f++)
    TAU_HIP(hipMemcpyAsync(host[f], h->buf[h->cur][f], n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (mask) TAU_HIP(hipMemcpyAsync(mask, h->mask, n, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tauh2_state_ptrs(tauh2_t *h, float *dptr[4], uint8_t **mask) {
  for (int f = 0; f < 4; f++) dptr[f] = h->buf[h->cur][f];
  if (mask) *mask = h->mask;
  return 0;
}

extern "C" int tauh2_init(tauh2_t *h) { // k_init, :740-770 (geometry on the host in fp64, state constant per class)
  const tauh2_params &P = h->p;
  size_t n = (size_t)P.W * P.H;
  std::vector<uint8_t> m(n);
  std::vector<float> st[4];
  for (int f = 0; f < 4; f++) st[f].resize(n);
  double Rb = P.geom_rb, Rn = P.geom_rn, th = P.geom_theta;
  double xb = Rn * (1.0 - sin(th)) + (Rb - Rn * cos(th)) / fmax(tan(th), 1e-30);
  const h2d::C4 in = h->base.in_c;
  const float restE = (float)(1.0 / (P.gamma - 1.0));
  for (int y = 0; y < P.H; y++)
    for (int x = 0; x < P.W; x++) {
      size_t i = (size_t)y * P.W + x;
      double X = (double)x - P.geom_x0, Y = (double)y - P.geom_cy;
      double sd = sdBody(X, Y, Rb, Rn, th) - Rb;
      sd = fmax(sd, X - xb);
      m[i] = (sd < 0.0) ? 1 : 0;
      st[0][i] = 1.0f; st[1][i] = m[i] ? 0.0f : in.mx; st[2][i] = 0.0f; st[3][i] = m[i] ? restE : in.E;
    }
  const float *ptr[4] = {st[0].data(), st[1].data(), st[2].data(), st[3].data()};
  if (tauh2_upload(h, ptr, m.data())) return 1;
  h2d::DevState z{};
  TAU_HIP(hipMemcpyAsync(h->st, &z, sizeof(z), hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// Chunk schedule of the LDS-window march: tau::guided_chunks over the 512 wave slots of an XCD (120 VGPRs: four waves per
// SIMD), 6 to 56 rows (a chunk re-does 4 warm-up rows; the band's 32-bit row offsets).  Chunks of one length (4096^2 in 16-row
// chunks: 4.3 rounds of the resident waves, 3.2 of 4 waves resident on average) against this schedule, round 4: 4096^2 54.4 ->
// 58.8, 8192^2 61.6 -> 64.8, 8192x1024 51.4 -> 53.8, 2048^2 41.0 -> 45.7, 3000^2 49.9 -> 54.8 Gcell/s.  Results are unchanged
// (a row's arithmetic does not depend on the chunk it is in; the maximum is order-free).  TAU_H2_ROWS=n: uniform chunks.
static int h2_schedule(tauh2 *h, int *nchunks, const int **crow) {
  if (!h->crow && tau::guided_chunks(h->p.H, (h->p.W + h2d::MCOLS - 1) / h2d::MCOLS, 512, 6, 56, &h->crow, &h->crow_n)) return 1;
  *nchunks = h->crow_n; *crow = h->crow;
  return 0;
}
static int h2_launch_step(tauh2 *h, float dt_explicit) {
  h2d::Args A = h->base;
  for (int f = 0; f < 4; f++) { A.in[f] = h->buf[h->cur][f]; A.out[f] = h->buf[h->cur ^ 1][f]; }
  A.slot = h->slot; A.dt_explicit = dt_explicit;
  const bool uexits = h->uniform_exits;
  A.uniform_exits = uexits ? 1 : 0;
  if (!h->maxs_valid) { // first step after init / upload: one reduction pass
    TAU_HIP(hipMemsetAsync(h->st->maxs_bits, 0, sizeof(h->st->maxs_bits), h->stream));
    hipLaunchKernelGGL(h2d::k_maxspeed, dim3(2048), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("h2d::k_maxspeed");
    h->maxs_valid = true;
  }
  static const int use_march = [] { const char *e = getenv("TAU_H2_MARCH"); return e ? atoi(e) : 1; }();
  if (use_march && A.W >= 8 && A.H >= 4 && (use_march > 1 || (long)A.W * A.H >= (1L << 21))) {
    const int nstrips = (A.W + h2d::MCOLS - 1) / h2d::MCOLS;
    // uniform chunks (the register-window march, TAU_H2_ROWS; the LDS-window march follows h2_schedule): ~16 k waves (four
    // waves per SIMD are resident: 4096 at a time) — 4096^2 with the LDS window:
    // 16 rows 46.7, 20: 46.3, 24: 45.8, 32: 45.2, 48: 42.8, 64: 38.8 Gcell/s (shorter chunks re-do 4 warm-up rows more often)
    int rows = (int)((long)A.H * nstrips / 16384);
    rows = rows < 8 ? 8 : (rows > 32 ? 32 : rows);
    // a wave count just above a whole number of rounds of the 4096 resident waves leaves the chip nearly empty for a
    // chunk's duration: 17-row chunks at 4096^2 are 4.06 rounds (48.5 Gcell/s), 16-row chunks 4.31 (49.7)
    for (int k = 0; k < 2 && rows > 8; k++) {
      const double rounds = (double)nstrips * ((A.H + rows - 1) / rows) / 4096.0;
      if (rounds > 1.0 && rounds - (long)rounds < 0.2) rows--; else break;
    }
    static const int rows_env = [] { const char *e = getenv("TAU_H2_ROWS"); return e ? atoi(e) : 0; }();
    if (rows_env >= 1) rows = rows_env < 56 ? rows_env : 56;
    int nchunks = (A.H + rows - 1) / rows;
    static const int lds_win = [] { const char *e = getenv("TAU_H2_LDSWIN"); return e ? atoi(e) : 1; }();   // the window in LDS (default) or in registers
    static const int wpb = [] { const char *e = getenv("TAU_H2_WPB"); return e ? atoi(e) : 1; }();   // one wave per workgroup: 49.2 against 48.2 Gcell/s with four (4096^2)
    const int *crow = nullptr;
    if (lds_win && rows_env < 1 && h2_schedule(h, &nchunks, &crow)) return 1;
    const unsigned nwork = (unsigned)(nstrips * nchunks);
    if (lds_win && wpb == 1 && uexits) hipLaunchKernelGGL((h2d::k_march_lds<1, true>), dim3(nwork), dim3(64), 0, h->stream, A, rows, nstrips, nchunks, crow);
    else if (lds_win && wpb == 1) hipLaunchKernelGGL((h2d::k_march_lds<1, false>), dim3(nwork), dim3(64), 0, h->stream, A, rows, nstrips, nchunks, crow);
    else if (lds_win && wpb == 2) hipLaunchKernelGGL((h2d::k_march_lds<2, false>), dim3((nwork + 1) / 2), dim3(128), 0, h->stream, A, rows, nstrips, nchunks, crow);
    else if (lds_win) hipLaunchKernelGGL((h2d::k_march_lds<4, false>), dim3((nwork + 3) / 4), dim3(256), 0, h->stream, A, rows, nstrips, nchunks, crow);
    else hipLaunchKernelGGL(h2d::k_march, dim3((unsigned)((nstrips * nchunks + 3) / 4)), dim3(256), 0, h->stream, A, rows, nstrips, nchunks);
  } else {
    hipLaunchKernelGGL(h2d::k_step, dim3((unsigned)(A.ntx * A.nty)), dim3(h2d::NT), 0, h->stream, A);
  }
  TAU_LAUNCH_CHECK("h2d::k_step");
  h->cur ^= 1; // swap_Us, :1378-1392
  h->slot = (h->slot + 1) % 3;
  return 0;
}

extern "C" int tauh2_step_async(tauh2_t *h, int nsteps) {
  TAU_HIP(hipSetDevice(h->device));
  for (int s = 0; s < nsteps; s++)
    if (h2_launch_step(h, 0.f)) return 1;
  return 0;
}
extern "C" int tauh2_sync(tauh2_t *h) {
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tauh2_get_time(tauh2_t *h, double *t, double *dt_last, double *maxs, int *step) {
  TAU_HIP(hipSetDevice(h->device));
  h2d::DevState s;
  TAU_HIP(hipMemcpyAsync(&s, h->st, sizeof(s), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  if (t) *t = s.t;
  if (dt_last) *dt_last = s.dt_last;
  if (maxs) { float m; memcpy(&m, &s.maxs_bits[h->slot], 4); *maxs = m; }
  if (step) *step = s.step;
  return 0;
}
extern "C" int tauh2_step(tauh2_t *h, int nsteps, double *t_out) {
  if (tauh2_step_async(h, nsteps)) return 1;
  if (t_out) return tauh2_get_time(h, t_out, nullptr, nullptr, nullptr);
  return tauh2_sync(h);
}
extern "C" int tauh2_unit_eval(tauh2_t *h, float out[48]) {
  TAU_HIP(hipSetDevice(h->device));
  float *d = nullptr;
  TAU_HIP(hipMalloc(&d, 48 * sizeof(float)));
  hipError_t e = hipMemsetAsync(d, 0, 48 * sizeof(float), h->stream);   // from here on `d` is freed on every path
  if (e == hipSuccess) {
    hipLaunchKernelGGL(h2d::k_unit, dim3(1), dim3(1), 0, h->stream, h->base, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, 48 * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return tau::fail("tauh2_unit_eval: %s", hipGetErrorString(e));
  return 0;
}
extern "C" int tauh2_unit_neighbors(tauh2_t *h, int x, int y, float out[9]) {
  if (x < 0 || x >= h->p.W || y < 0 || y >= h->p.H) return tau::fail("tauh2_unit_neighbors: cell (%d,%d) outside the grid", x, y);
  TAU_HIP(hipSetDevice(h->device));
  float *d = nullptr;
  TAU_HIP(hipMalloc(&d, 9 * sizeof(float)));
  h2d::Args A = h->base;
  for (int f = 0; f < 4; f++) A.in[f] = h->buf[h->cur][f];
  hipLaunchKernelGGL(h2d::k_unit_neighbors, dim3(1), dim3(1), 0, h->stream, A, x, y, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, 9 * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return tau::fail("tauh2_unit_neighbors: %s", hipGetErrorString(e));
  return 0;
}
/* signed distance of the rounded sphere-cone body (host fp64, the expression k_init evaluates) */
extern "C" double tauh2_body_sdf(double x, double y, double Rb, double Rn, double theta) { return sdBody(x, y, Rb, Rn, theta); }

// ---- rendering (tau_hypersonic_cuda.cu:1871-1888: render_vals -> min/max -> inv range -> pixels) ----
extern "C" int tauh2_render(tauh2_t *h, int view_mode, uint32_t *host_rgba, float *host_vals, double *vmin, double *vmax) {
  if (view_mode < 0 || view_mode > 6) return tau::fail("tauh2_render: view mode %d outside 0..6", view_mode);
  TAU_HIP(hipSetDevice(h->device));
  const int N = h->p.W * h->p.H;
  if (!h->rval) TAU_HIP(hipMalloc(&h->rval, (size_t)N * sizeof(float)));
  if (!h->rpix) TAU_HIP(hipMalloc(&h->rpix, (size_t)N * sizeof(uint32_t)));
  if (!h->rmm) TAU_HIP(hipMalloc(&h->rmm, 2 * sizeof(unsigned)));
  h2d::Args A = h->base;
  for (int f = 0; f < 4; f++) A.in[f] = h->buf[h->cur][f];
  const unsigned init[2] = {0xffffffffu, 0u};
  TAU_HIP(hipMemcpyAsync(h->rmm, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(h2d::k_render_vals, dim3(nb), dim3(256), 0, h->stream, A, view_mode, h->rval, h->rmm);
  TAU_LAUNCH_CHECK("k_render_vals");
  hipLaunchKernelGGL(h2d::k_render_pixels, dim3(nb), dim3(256), 0, h->stream, (const uint8_t *)h->mask, (const float *)h->rval, N,
                     (const unsigned *)h->rmm, h->rpix);
  TAU_LAUNCH_CHECK("k_render_pixels");
  unsigned keys[2];
  TAU_HIP(hipMemcpyAsync(keys, h->rmm, sizeof(keys), hipMemcpyDeviceToHost, h->stream));
  if (host_rgba) TAU_HIP(hipMemcpyAsync(host_rgba, h->rpix, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
  if (host_vals) TAU_HIP(hipMemcpyAsync(host_vals, h->rval, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  auto unkey = [](unsigned k) { unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &b, 4); return (double)f; };
  if (vmin) *vmin = unkey(keys[0]);
  if (vmax) *vmax = unkey(keys[1]);
  return 0;
}

extern "C" int tauh2_step_explicit(tauh2_t *h, double dt) {
  if (!(dt > 0.0)) return tau::fail("tauh2_step_explicit: dt must be positive");
  TAU_HIP(hipSetDevice(h->device));
  if (h2_launch_step(h, (float)dt)) return 1;
  return tauh2_sync(h);
}
