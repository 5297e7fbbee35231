// flow2d.hip — full time step of the 2D viscous Burgers and shallow-water solvers for gfx950.
//
// Reference pipelines, per step:
//   tau_burgers.cu:677-718        wavespeed_block_max + HOST max -> flux_x_kernel, flux_y_kernel ->
//                                 update_convective -> K x viscosity_step          (6 arrays of scratch)
//   tau_shallow_water.cu:671-705  wavespeed_block_max + HOST max -> flux_x_kernel, flux_y_kernel ->
//                                 update_kernel -> viscosity_uv                     (6 arrays of scratch)
// Here ONE kernel per step: a 256-thread workgroup stages its 32x8 tile (+halo) in LDS, every cell of
// the tile + 1-cell ring forms its four face fluxes from LDS and takes the conservative update into
// LDS, the tile then applies the first viscosity pass to those updated values (the race-free form of
// the reference's in-place Laplacian), writes the new state and reduces the wavespeed of the NEW
// state (wave64 butterfly + one atomicMax per workgroup) for the next step's dt — no flux arrays,
// no separate reduction pass, no host round trip.  Compulsory traffic: Burgers 16 B/cell, shallow
// water 24 B/cell (the reference moves ~100-130 B/cell).
//
// Deviation (rounding level): the reference re-encodes after the convective update and the viscosity
// kernel decodes again (phi = asinh(u/u0); u = u0 sinh(phi)); the fused kernel keeps u between the two.

#include "../../include/taueng.h"
#include "tau_common.h"
#include <cmath>
#include <new>
#include <vector>

namespace fl2 {

constexpr int TX = 32, TY = 8, NT = TX * TY;
// tile halo: the ring needs its own faces (+1), MUSCL Burgers faces reach two cells further (+2); shallow water
// and plain Burgers have no reconstruction, so halo 2 is enough there (432 staged cells instead of 532, and the
// shallow-water LDS fits 8 workgroups per CU instead of 7).  MUSCL is a template parameter for that reason.
template <int KIND, bool MUSCL> struct TileDims {
  static constexpr int HB = (KIND == 0 && MUSCL) ? 3 : 2;
  static constexpr int UW = TX + 2 * HB, UH = TY + 2 * HB;
};
constexpr int RW = TX + 2, RH = TY + 2;             // updated values: tile + ring 1
enum { K_BURGERS = 0, K_SW = 1 };

struct DevState {
  // wavespeed metric (float bits), three slots in rotation: step s reads slot s % 3 (its input state), reduces the
  // state it writes into slot (s+1) % 3 and clears slot (s+2) % 3 for the step after — no kernel between steps
  unsigned maxbits[3];
  float dt_last;
};

struct Args {
  const float *in[3];
  float *out[3];
  DevState *st;
  int nx, ny, ntx, nty, slot;
  float dx, dy, invdx, invdy, invdx2, invdy2, nu, u0, inv_u0, g, CFL, dt_try, dt_explicit, cfl_len;
  int muscl, oneD, do_visc, reduce;
  float visc_frac;       // dt fraction of the fused viscosity pass (1/K)
};

// (codec constants out of SGPRs, as in h3d.hip: a VOP3 instruction — anything with |x|, bfi — cannot carry a literal, the constant
//  lands in an SGPR and an SGPR source halves the issue rate; so: a SIGNED exponential (no |x| multiply, no copysign) in sinh, and
//  the sign mask of asinh's copysign in a VGPR)
__device__ __forceinline__ float fsinh(float x) {
  float x2 = x * x;
  float series = x * (1.f + x2 * (1.f / 6.f) * (1.f + x2 * (1.f / 20.f) * (1.f + x2 * (1.f / 42.f))));
  float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
  float big = 0.5f * (e - __builtin_amdgcn_rcpf(e));
  return (fabsf(x) < 0.5f) ? series : big;
}
__device__ __forceinline__ float fasinh(float x) {
  float ax = fabsf(x), x2 = x * x;
  // x - x^3/6 + 3x^5/40 - 15x^7/336 + 105x^9/3456
  float series = ax * (1.f + x2 * (-1.f / 6.f + x2 * (3.f / 40.f + x2 * (-15.f / 336.f + x2 * (105.f / 3456.f)))));
  float big = __builtin_amdgcn_logf(ax + __builtin_amdgcn_sqrtf(x2 + 1.0f)) * 0.69314718055994531f;
  float mag = (ax < 0.125f) ? series : big, mask, r;
  asm("v_mov_b32 %0, 0x7fffffff" : "=v"(mask));
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(mag), "v"(x));
  return r;
}
__device__ __forceinline__ float minmodf(float a, float b) { // tau_burgers.cu:332-334
  return __builtin_amdgcn_fmed3f(a, b, 0.0f);   // the median of (a, b, 0): one v_med3_f32 (differs only where a * b underflows to 0)
}
__device__ __forceinline__ int wrapi(int i, int n) { i %= n; return i < 0 ? i + n : i; }
// periodic wrap of an index known to lie in [-n, 2n): two selects instead of an integer modulo
__device__ __forceinline__ int wrap1(int i, int n) { i = i < 0 ? i + n : i; return i >= n ? i - n : i; }

// Rusanov flux of the Burgers system through one face along `ax`, from the four phi values
// (m1, c | p1, p2) of each component around it; flux_x_kernel / flux_y_kernel, :364-455
template <bool MUSCL>
__device__ __forceinline__ void burgers_face(const Args &A, float um1, float uc, float up1, float up2, float vm1, float vc,
                                             float vp1, float vp2, int ax, float &Fu, float &Fv) {
  float pUL = uc, pUR = up1, pVL = vc, pVR = vp1;
  if (MUSCL) {
    pUL = uc + 0.5f * minmodf(uc - um1, up1 - uc);
    pUR = up1 - 0.5f * minmodf(up2 - up1, up1 - uc);
    pVL = vc + 0.5f * minmodf(vc - vm1, vp1 - vc);
    pVR = vp1 - 0.5f * minmodf(vp2 - vp1, vp1 - vc);
  }
  // without MUSCL the face states are the cell values, which the staging loop has already decoded
  const float uL = MUSCL ? A.u0 * fsinh(pUL) : pUL, vL = MUSCL ? A.u0 * fsinh(pVL) : pVL;
  const float uR = MUSCL ? A.u0 * fsinh(pUR) : pUR, vR = MUSCL ? A.u0 * fsinh(pVR) : pVR;
  if (ax == 0) {
    const float a = fmaxf(fabsf(uL), fabsf(uR));
    Fu = 0.5f * (0.5f * uL * uL + 0.5f * uR * uR) - 0.5f * a * (uR - uL);
    Fv = 0.5f * (uL * vL + uR * vR) - 0.5f * a * (vR - vL);
  } else {
    const float a = fmaxf(fabsf(vL), fabsf(vR));
    Fu = 0.5f * (uL * vL + uR * vR) - 0.5f * a * (uR - uL);
    Fv = 0.5f * (0.5f * vL * vL + 0.5f * vR * vR) - 0.5f * a * (vR - vL);
  }
}

// HLL flux of the shallow-water system (n = normal, t = tangential velocity), hll_x / hll_y, :327-390.
// cL, cR = sqrt(g h) come from LDS (one square root per staged cell instead of two per face evaluation);
// the two supersonic early-outs are selects.
__device__ __forceinline__ void sw_face(float g, float hL, float unL, float utL, float cL, float hR, float unR, float utR,
                                        float cR, float &Fh, float &Fn, float &Ft) {
  const float sL = fminf(unL - cL, unR - cR), sR = fmaxf(unL + cL, unR + cR);
  const float mL = hL * unL, mR = hR * unR, nL = hL * utL, nR = hR * utR;
  const float FLh = mL, FLn = mL * unL + 0.5f * g * hL * hL, FLt = mL * utL;
  const float FRh = mR, FRn = mR * unR + 0.5f * g * hR * hR, FRt = mR * utR;
  const float inv = __builtin_amdgcn_rcpf(sR - sL), ss = sR * sL;
  const float Hh = (sR * FLh - sL * FRh + ss * (hR - hL)) * inv;
  const float Hn = (sR * FLn - sL * FRn + ss * (mR - mL)) * inv;
  const float Ht = (sR * FLt - sL * FRt + ss * (nR - nL)) * inv;
  const bool left = sL >= 0.0f, right = sR <= 0.0f;
  Fh = left ? FLh : right ? FRh : Hh;
  Fn = left ? FLn : right ? FRn : Hn;
  Ft = left ? FLt : right ? FRt : Ht;
}

template <int KIND, bool MUSCL>
__global__ __launch_bounds__(NT) void k_step(const Args A) {
  constexpr int NF = (KIND == K_BURGERS) ? 2 : 3;
  constexpr int HB = TileDims<KIND, MUSCL>::HB, UW = TileDims<KIND, MUSCL>::UW, UH = TileDims<KIND, MUSCL>::UH;
  constexpr int NS = (KIND == K_BURGERS) ? 2 : 4;
  __shared__ float sU[NS][UH * UW];       // Burgers: phi_u, phi_v ; SW: h, u, v, sqrt(g h)
  __shared__ float sN[3][RH * RW];        // updated u, v (and h for shallow water) on tile + ring
  __shared__ float sFx[NF][RH * (RW + 1)]; // flux through the low-x face of ring cell (rx, ry), rx = 0..RW
  __shared__ float sFy[NF][(RH + 1) * RW]; // flux through the low-y face of ring cell (rx, ry), ry = 0..RH
  __shared__ float sRed[NT / 64];

  const int tid = threadIdx.x, tx = tid & (TX - 1), ty = tid >> 5, lane = tid & 63, wave = tid >> 6;
  unsigned b = tau::xcd_swizzle(blockIdx.x, (unsigned)(A.ntx * A.nty));
  const int bx0 = (int)(b % (unsigned)A.ntx) * TX, by0 = (int)(b / (unsigned)A.ntx) * TY;

  float dt; // dt_eff = min(t*dtau, CFL*len/max), tau_burgers.cu:693-694, tau_shallow_water.cu:689-690
  if (A.dt_explicit > 0.f) dt = A.dt_explicit;
  else {
    float m = __uint_as_float(A.st->maxbits[A.slot]);
    if (!(m >= 1e-12f)) m = 1e-12f;
    dt = fminf(A.dt_try, A.CFL * A.cfl_len / m);
  }
  if (blockIdx.x == 0 && tid == 0) { // the step's bookkeeping (was a 1-thread kernel of its own)
    A.st->dt_last = dt;
    A.st->maxbits[(A.slot + 2) % 3] = 0u;
  }

  for (int t = tid; t < UH * UW; t += NT) {
    const int ly = t / UW, lx = t - ly * UW;
    // tile coordinates overshoot the grid by at most TX + HB (ragged last tile): one conditional wrap is enough
    // unless the grid is smaller than that
    const int gy = by0 - HB + ly, gx = bx0 - HB + lx;
    const size_t gi = (size_t)(A.ny >= TY + HB ? wrap1(gy, A.ny) : wrapi(gy, A.ny)) * A.nx + (A.nx >= TX + HB ? wrap1(gx, A.nx) : wrapi(gx, A.nx));
    if (KIND == K_BURGERS) { // MUSCL limits the encoded phi, so phi is staged raw; otherwise decode once here
      const float a = A.in[0][gi], bb = A.in[1][gi];
      sU[0][t] = MUSCL ? a : A.u0 * fsinh(a);
      sU[1][t] = MUSCL ? bb : A.u0 * fsinh(bb);
    }
    else {
      const float hh = __builtin_amdgcn_exp2f(A.in[0][gi] * 1.44269504088896341f);   // v_exp_f32: 1 ulp, the tolerance is 1e-5
      sU[0][t] = hh; sU[1][t] = A.in[1][gi]; sU[2][t] = A.in[2][gi]; sU[NS - 1][t] = __builtin_amdgcn_sqrtf(A.g * hh);
    }
  }
  __syncthreads();

  // ---- every face of tile + ring ONCE (x faces first, then y faces; the axis varies per lane in the one
  // round that straddles the two lists).  A cell-owns-its-four-faces form evaluates 5.3 faces per tile cell,
  // this one 2.8 — and a MUSCL Burgers face costs four sinh.
  // (Plain Burgers keeps the cell-owns-its-faces form below: its faces are a dozen flops on already decoded
  // values, cheaper than a trip through LDS and a barrier — 92 vs 87 Gcell/s.)
  constexpr int NFX = RH * (RW + 1), NFY = (RH + 1) * RW;
  const bool own_faces = (KIND == K_BURGERS) && !MUSCL;
  for (int f = tid; f < (own_faces ? 0 : NFX + NFY); f += NT) {
    const bool isx = f < NFX;
    const int g = isx ? f : f - NFX;
    const int fy = isx ? g / (RW + 1) : g / RW;          // divisions by constants (a lane-varying divisor is ~30 instructions)
    const int fx = g - fy * (isx ? RW + 1 : RW);
    const int c = (fy + HB - 1) * UW + (fx + HB - 1);   // the cell on the high side of the face
    const int st = isx ? 1 : UW;                        // step across the face
    if (KIND == K_BURGERS) {
      const float *pu = sU[0], *pv = sU[1];
      float Fu = 0.f, Fv = 0.f;
      if (isx || !A.oneD)
        burgers_face<MUSCL>(A, pu[c - 2 * st], pu[c - st], pu[c], pu[c + st], pv[c - 2 * st], pv[c - st], pv[c], pv[c + st], isx ? 0 : 1,
                     Fu, Fv);
      (isx ? sFx[0] : sFy[0])[g] = Fu;
      (isx ? sFx[1] : sFy[1])[g] = Fv;
    } else {
      const float *ph = sU[0], *pn = isx ? sU[1] : sU[2], *pt = isx ? sU[2] : sU[1], *pc = sU[3];
      float Fh, Fn, Ft;
      sw_face(A.g, ph[c - st], pn[c - st], pt[c - st], pc[c - st], ph[c], pn[c], pt[c], pc[c], Fh, Fn, Ft);
      (isx ? sFx[0] : sFy[0])[g] = Fh;                  // components stored as (h, x-momentum, y-momentum)
      (isx ? sFx[1] : sFy[1])[g] = isx ? Fn : Ft;
      (isx ? sFx[NF - 1] : sFy[NF - 1])[g] = isx ? Ft : Fn;
    }
  }
  __syncthreads();

  // ---- conservative update of every cell of tile + ring from its four faces
  for (int t = tid; t < RH * RW; t += NT) {
    const int ry = t / RW, rx = t - ry * RW;
    const int c = (ry + HB - 1) * UW + (rx + HB - 1);
    const int fxi = ry * (RW + 1) + rx, fyi = ry * RW + rx;
    float un, vn;
    if (KIND == K_BURGERS) {
      const float *pu = sU[0], *pv = sU[1];
      const float invdy = A.oneD ? 0.0f : A.invdy;
      const float uc0 = MUSCL ? A.u0 * fsinh(pu[c]) : pu[c], vc0 = MUSCL ? A.u0 * fsinh(pv[c]) : pv[c];
      float Fu_lo, Fv_lo, Fu_hi, Fv_hi, Gu_lo = 0.f, Gv_lo = 0.f, Gu_hi = 0.f, Gv_hi = 0.f;
      if (own_faces) {
        burgers_face<MUSCL>(A, pu[c - 2], pu[c - 1], pu[c], pu[c + 1], pv[c - 2], pv[c - 1], pv[c], pv[c + 1], 0, Fu_lo, Fv_lo);
        burgers_face<MUSCL>(A, pu[c - 1], pu[c], pu[c + 1], pu[c + 2], pv[c - 1], pv[c], pv[c + 1], pv[c + 2], 0, Fu_hi, Fv_hi);
        if (!A.oneD) {
          burgers_face<MUSCL>(A, pu[c - 2 * UW], pu[c - UW], pu[c], pu[c + UW], pv[c - 2 * UW], pv[c - UW], pv[c], pv[c + UW], 1, Gu_lo, Gv_lo);
          burgers_face<MUSCL>(A, pu[c - UW], pu[c], pu[c + UW], pu[c + 2 * UW], pv[c - UW], pv[c], pv[c + UW], pv[c + 2 * UW], 1, Gu_hi, Gv_hi);
        }
      } else {
        Fu_lo = sFx[0][fxi]; Fu_hi = sFx[0][fxi + 1]; Fv_lo = sFx[1][fxi]; Fv_hi = sFx[1][fxi + 1];
        Gu_lo = sFy[0][fyi]; Gu_hi = sFy[0][fyi + RW]; Gv_lo = sFy[1][fyi]; Gv_hi = sFy[1][fyi + RW];
      }
      un = uc0 - dt * ((Fu_hi - Fu_lo) * A.invdx + (Gu_hi - Gu_lo) * invdy); // update_convective, :458-487
      vn = vc0 - dt * ((Fv_hi - Fv_lo) * A.invdx + (Gv_hi - Gv_lo) * invdy);
    } else {
      float h = sU[0][c], mx = h * sU[1][c], my = h * sU[2][c]; // update_kernel, :474-513
      h -= dt * ((sFx[0][fxi + 1] - sFx[0][fxi]) * A.invdx + (sFy[0][fyi + RW] - sFy[0][fyi]) * A.invdy);
      mx -= dt * ((sFx[1][fxi + 1] - sFx[1][fxi]) * A.invdx + (sFy[1][fyi + RW] - sFy[1][fyi]) * A.invdy);
      my -= dt * ((sFx[NF - 1][fxi + 1] - sFx[NF - 1][fxi]) * A.invdx + (sFy[NF - 1][fyi + RW] - sFy[NF - 1][fyi]) * A.invdy);
      h = fmaxf(h, 1e-6f);
      const float ih = __builtin_amdgcn_rcpf(h);
      un = mx * ih;
      vn = my * ih;
      sN[2][t] = h;
    }
    sN[0][t] = un;
    sN[1][t] = vn;
  }
  __syncthreads();

  // ---- first viscosity pass on the updated values (viscosity_step :490-525 / viscosity_uv :516-547), store
  const int x = bx0 + tx, y = by0 + ty;
  float red = 0.f;
  if (x < A.nx && y < A.ny) {
    const int r = (ty + 1) * RW + (tx + 1);
    float u = sN[0][r], v = sN[1][r];
    if (A.do_visc) {
      const float nudt = A.nu * (dt * A.visc_frac);
      const float invdy2 = (KIND == K_BURGERS && A.oneD) ? 0.0f : A.invdy2;
      const float lu = (sN[0][r + 1] - 2.0f * u + sN[0][r - 1]) * A.invdx2 + (sN[0][r + RW] - 2.0f * u + sN[0][r - RW]) * invdy2;
      const float lv = (sN[1][r + 1] - 2.0f * v + sN[1][r - 1]) * A.invdx2 + (sN[1][r + RW] - 2.0f * v + sN[1][r - RW]) * invdy2;
      u += nudt * lu;
      v += nudt * lv;
    }
    const size_t gi = (size_t)y * A.nx + x;
    if (KIND == K_BURGERS) {
      A.out[0][gi] = fasinh(u * A.inv_u0);
      A.out[1][gi] = fasinh(v * A.inv_u0);
      red = fabsf(u) * A.invdx + fabsf(v) * ((A.ny > 1) ? A.invdy : 0.0f); // wavespeed_block_max, :337-361
    } else {
      const float own_h = sN[2][r];
      A.out[0][gi] = __builtin_amdgcn_logf(own_h) * 0.69314718055994531f;          // v_log_f32
      A.out[1][gi] = u;
      A.out[NF - 1][gi] = v;
      const float c = __builtin_amdgcn_sqrtf(A.g * own_h);
      red = fmaxf(fabsf(u) + c, fabsf(v) + c); // :394-422
    }
  }
  if (A.reduce) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) red = fmaxf(red, __shfl_xor(red, o, 64));
    if (lane == 0) sRed[wave] = red;
    __syncthreads();
    if (tid == 0) {
      float m = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
      tau::atomic_max_float_bits(&A.st->maxbits[(A.slot + 1) % 3], m);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// The same step as a MARCH (plain Burgers and shallow water; MUSCL keeps the tile kernel above).  A wave owns 60
// columns (64 lanes, two halo lanes a side — the viscosity stencil of the updated values reaches two cells) and walks
// down a chunk of rows with everything in registers: per trip it takes in one input row, forms the y-face fluxes
// between the last two input rows and the x-face fluxes of the older one (left neighbour by lane shift), updates that
// row, applies the viscosity pass to the row before it from the three updated rows it holds, and stores it.  Every
// face is evaluated once per wave, nothing goes through LDS, and the arithmetic per cell is the tile kernel's, operand
// for operand — the results are bit-identical.  Redundancy: 64/60 lanes x (R + 4)/R rows instead of the tile's
// 432 staged and 340 updated cells per 256.
constexpr int MCOLS = 60;
struct MRow { float a, b, c, d; };   // Burgers: u, v ; shallow water: h, u, v, sqrt(g h)

template <int KIND>
__device__ __forceinline__ MRow march_load(const Args &A, int row, int col, bool ok) {
  MRow r{0.f, 0.f, 0.f, 0.f};
  const size_t gi = (size_t)row * A.nx + col;
  if (KIND == K_BURGERS) {
    r.a = A.u0 * fsinh(A.in[0][gi]);
    r.b = A.u0 * fsinh(A.in[1][gi]);
  } else {
    r.a = __builtin_amdgcn_exp2f(A.in[0][gi] * 1.44269504088896341f);
    r.b = A.in[1][gi]; r.c = A.in[2][gi];
    r.d = __builtin_amdgcn_sqrtf(A.g * r.a);
  }
  (void)ok;
  return r;
}
// flux through the face between cell L (low side) and cell R along axis ax: 2 (Burgers) or 3 components
template <int KIND>
__device__ __forceinline__ void march_face(const Args &A, const MRow &L, const MRow &R, int ax, float &f0, float &f1, float &f2) {
  if (KIND == K_BURGERS) {
    f2 = 0.f;
    burgers_face<false>(A, 0.f, L.a, R.a, 0.f, 0.f, L.b, R.b, 0.f, ax, f0, f1);
  } else {
    float Fh, Fn, Ft;
    if (ax == 0) sw_face(A.g, L.a, L.b, L.c, L.d, R.a, R.b, R.c, R.d, Fh, Fn, Ft);
    else sw_face(A.g, L.a, L.c, L.b, L.d, R.a, R.c, R.b, R.d, Fh, Fn, Ft);
    f0 = Fh; f1 = ax == 0 ? Fn : Ft; f2 = ax == 0 ? Ft : Fn;   // stored as (h, x-momentum, y-momentum)
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_march(const Args A, int rows, int nstrips, int nchunks, const int *__restrict__ crow) {
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(nstrips * nchunks);
  const unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);

  float dt; // dt_eff = min(t*dtau, CFL*len/max), tau_burgers.cu:693-694, tau_shallow_water.cu:689-690
  if (A.dt_explicit > 0.f) dt = A.dt_explicit;
  else {
    float m = __uint_as_float(A.st->maxbits[A.slot]);
    if (!(m >= 1e-12f)) m = 1e-12f;
    dt = fminf(A.dt_try, A.CFL * A.cfl_len / m);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { // the step's bookkeeping
    A.st->dt_last = dt;
    A.st->maxbits[(A.slot + 2) % 3] = 0u;
  }
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)nstrips), chunk = (int)(wid / (unsigned)nstrips);
  const int xo = strip * MCOLS + lane - 2;                       // column of this lane, before the periodic wrap
  const int col = ((xo % A.nx) + A.nx) % A.nx;
  const bool own = lane >= 2 && lane < 2 + MCOLS && xo < A.nx;   // the lanes that store
  const int j0 = crow ? crow[chunk] : chunk * rows, j1 = crow ? crow[chunk + 1] : min(j0 + rows, A.ny);   // (tau::guided_chunks, or `rows` each)
  if (j0 >= j1) return;
  const bool oneD = (KIND == K_BURGERS) && A.oneD;
  const float invdy = oneD ? 0.0f : A.invdy;
  const float invdy2 = oneD ? 0.0f : A.invdy2;
  const float nudt = A.nu * (dt * A.visc_frac);

  auto wrapy = [&](int r) { return ((r % A.ny) + A.ny) % A.ny; };
  MRow P{0.f, 0.f, 0.f, 0.f};       // input row r-1
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;   // y-face flux below row r-1 (between r-2 and r-1)
  float w0a = 0.f, w0b = 0.f, w1a = 0.f, w1b = 0.f, w1h = 0.f;   // updated (u, v) of rows r-3, r-2 (and h of r-2)
  float red = 0.f;
  MRow N = march_load<KIND>(A, wrapy(j0 - 2), col, true);
  for (int r = j0 - 2; r <= j1 + 1; r++) {
    MRow nxt = N;
    if (r < j1 + 1) nxt = march_load<KIND>(A, wrapy(r + 1), col, true);   // prefetch the next input row
    // y-face between rows r-1 (P) and r (N)
    float G0 = 0.f, G1 = 0.f, G2 = 0.f;
    if (!oneD) march_face<KIND>(A, P, N, 1, G0, G1, G2);
    // x-faces of row r-1: low face of this lane from the lane below, high face = the low face of the lane above
    MRow Pl;
    Pl.a = __shfl_up(P.a, 1, 64); Pl.b = __shfl_up(P.b, 1, 64);
    Pl.c = (KIND == K_SW) ? __shfl_up(P.c, 1, 64) : 0.f; Pl.d = (KIND == K_SW) ? __shfl_up(P.d, 1, 64) : 0.f;
    float F0, F1, F2;
    march_face<KIND>(A, Pl, P, 0, F0, F1, F2);
    const float F0h = __shfl_down(F0, 1, 64), F1h = __shfl_down(F1, 1, 64), F2h = (KIND == K_SW) ? __shfl_down(F2, 1, 64) : 0.f;
    // conservative update of row r-1 (update_convective :458-487 / update_kernel :474-513)
    float un, vn, hn = 0.f;
    if (KIND == K_BURGERS) {
      un = P.a - dt * ((F0h - F0) * A.invdx + (G0 - g0) * invdy);
      vn = P.b - dt * ((F1h - F1) * A.invdx + (G1 - g1) * invdy);
    } else {
      float h = P.a, mx = h * P.b, my = h * P.c;
      h -= dt * ((F0h - F0) * A.invdx + (G0 - g0) * A.invdy);
      mx -= dt * ((F1h - F1) * A.invdx + (G1 - g1) * A.invdy);
      my -= dt * ((F2h - F2) * A.invdx + (G2 - g2) * A.invdy);
      h = fmaxf(h, 1e-6f);
      const float ih = __builtin_amdgcn_rcpf(h);
      un = mx * ih; vn = my * ih; hn = h;
    }
    // viscosity pass on row r-2 from the updated rows r-3 (w0), r-2 (w1), r-1 (un, vn), then store it
    const int o = r - 2;
    {
      float u = w1a, v = w1b;
      if (A.do_visc) {
        const float ul = __shfl_up(w1a, 1, 64), ur = __shfl_down(w1a, 1, 64);
        const float vl = __shfl_up(w1b, 1, 64), vr = __shfl_down(w1b, 1, 64);
        const float lu = (ur - 2.0f * u + ul) * A.invdx2 + (un - 2.0f * u + w0a) * invdy2;
        const float lv = (vr - 2.0f * v + vl) * A.invdx2 + (vn - 2.0f * v + w0b) * invdy2;
        u += nudt * lu;
        v += nudt * lv;
      }
      if (own && o >= j0 && o < j1) {
        const size_t gi = (size_t)o * A.nx + xo;
        if (KIND == K_BURGERS) {
          A.out[0][gi] = fasinh(u * A.inv_u0);
          A.out[1][gi] = fasinh(v * A.inv_u0);
          red = fmaxf(red, fabsf(u) * A.invdx + fabsf(v) * ((A.ny > 1) ? A.invdy : 0.0f));
        } else {
          A.out[0][gi] = __builtin_amdgcn_logf(w1h) * 0.69314718055994531f;
          A.out[1][gi] = u;
          A.out[2][gi] = v;
          const float c = __builtin_amdgcn_sqrtf(A.g * w1h);
          red = fmaxf(red, fmaxf(fabsf(u) + c, fabsf(v) + c));
        }
      }
    }
    // slide
    w0a = w1a; w0b = w1b; w1a = un; w1b = vn; w1h = hn;
    g0 = G0; g1 = G1; g2 = G2;
    P = N; N = nxt;
  }
  if (A.reduce) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) red = fmaxf(red, __shfl_xor(red, o, 64));
    if (lane == 0) tau::atomic_max_float_bits(&A.st->maxbits[(A.slot + 1) % 3], red);
  }
}

// Two columns per lane (even nx): a wave covers 128 columns and owns the inner 124 — the two halo cells a side are ONE
// lane —, the face between a lane's two cells is local, and only the outer faces and the outer viscosity neighbours
// cross lanes: half the lane shifts per cell, two independent cells of work per lane, float2 loads and stores.
constexpr int MCOLS2 = 124;
template <int KIND>
__global__ __launch_bounds__(256) void k_march2(const Args A, int rows, int nstrips, int nchunks, const int *__restrict__ crow) {
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(nstrips * nchunks);
  const unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  float dt;
  if (A.dt_explicit > 0.f) dt = A.dt_explicit;
  else {
    float m = __uint_as_float(A.st->maxbits[A.slot]);
    if (!(m >= 1e-12f)) m = 1e-12f;
    dt = fminf(A.dt_try, A.CFL * A.cfl_len / m);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.st->dt_last = dt;
    A.st->maxbits[(A.slot + 2) % 3] = 0u;
  }
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)nstrips), chunk = (int)(wid / (unsigned)nstrips);
  const int xo = strip * MCOLS2 + 2 * lane - 2;                  // first of this lane's two columns (even)
  const int col = ((xo % A.nx) + A.nx) % A.nx;                   // nx is even: the pair never straddles the wrap
  const bool own = lane >= 1 && lane < 63 && xo < A.nx;
  const int j0 = crow ? crow[chunk] : chunk * rows, j1 = crow ? crow[chunk + 1] : min(j0 + rows, A.ny);   // (tau::guided_chunks, or `rows` each)
  if (j0 >= j1) return;
  const bool oneD = (KIND == K_BURGERS) && A.oneD;
  const float invdy = oneD ? 0.0f : A.invdy, invdy2 = oneD ? 0.0f : A.invdy2;
  const float nudt = A.nu * (dt * A.visc_frac);
  auto wrapy = [&](int r) { return ((r % A.ny) + A.ny) % A.ny; };
  auto load2 = [&](int r, MRow (&q)[2]) {
    const size_t gi = (size_t)wrapy(r) * A.nx + col;
    const float2 f0 = *reinterpret_cast<const float2 *>(A.in[0] + gi), f1 = *reinterpret_cast<const float2 *>(A.in[1] + gi);
    if (KIND == K_BURGERS) {
      q[0] = MRow{A.u0 * fsinh(f0.x), A.u0 * fsinh(f1.x), 0.f, 0.f};
      q[1] = MRow{A.u0 * fsinh(f0.y), A.u0 * fsinh(f1.y), 0.f, 0.f};
    } else {
      const float2 f2 = *reinterpret_cast<const float2 *>(A.in[2] + gi);
      const float h0 = __builtin_amdgcn_exp2f(f0.x * 1.44269504088896341f), h1 = __builtin_amdgcn_exp2f(f0.y * 1.44269504088896341f);
      q[0] = MRow{h0, f1.x, f2.x, __builtin_amdgcn_sqrtf(A.g * h0)};
      q[1] = MRow{h1, f1.y, f2.y, __builtin_amdgcn_sqrtf(A.g * h1)};
    }
  };
  MRow P[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, N[2], nxt[2];
  float g[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  float w0a[2] = {0.f, 0.f}, w0b[2] = {0.f, 0.f}, w1a[2] = {0.f, 0.f}, w1b[2] = {0.f, 0.f}, w1h[2] = {0.f, 0.f};
  float red = 0.f;
  load2(j0 - 2, N);
  for (int r = j0 - 2; r <= j1 + 1; r++) {
    nxt[0] = N[0]; nxt[1] = N[1];
    if (r < j1 + 1) load2(r + 1, nxt);
    float G[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (!oneD) {
      march_face<KIND>(A, P[0], N[0], 1, G[0][0], G[0][1], G[0][2]);
      march_face<KIND>(A, P[1], N[1], 1, G[1][0], G[1][1], G[1][2]);
    }
    // x faces of row r-1: left of cell 0 (from the lane below's cell 1), between the two cells, right of cell 1
    MRow Pl;
    Pl.a = __shfl_up(P[1].a, 1, 64); Pl.b = __shfl_up(P[1].b, 1, 64);
    Pl.c = (KIND == K_SW) ? __shfl_up(P[1].c, 1, 64) : 0.f; Pl.d = (KIND == K_SW) ? __shfl_up(P[1].d, 1, 64) : 0.f;
    float Fl[3], Fm[3], Fh[3];
    march_face<KIND>(A, Pl, P[0], 0, Fl[0], Fl[1], Fl[2]);
    march_face<KIND>(A, P[0], P[1], 0, Fm[0], Fm[1], Fm[2]);
    Fh[0] = __shfl_down(Fl[0], 1, 64); Fh[1] = __shfl_down(Fl[1], 1, 64); Fh[2] = (KIND == K_SW) ? __shfl_down(Fl[2], 1, 64) : 0.f;
    float un[2], vn[2], hn[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const float *lo = c ? Fm : Fl, *hi = c ? Fh : Fm;
      if (KIND == K_BURGERS) {
        un[c] = P[c].a - dt * ((hi[0] - lo[0]) * A.invdx + (G[c][0] - g[c][0]) * invdy);
        vn[c] = P[c].b - dt * ((hi[1] - lo[1]) * A.invdx + (G[c][1] - g[c][1]) * invdy);
      } else {
        float h = P[c].a, mx = h * P[c].b, my = h * P[c].c;
        h -= dt * ((hi[0] - lo[0]) * A.invdx + (G[c][0] - g[c][0]) * A.invdy);
        mx -= dt * ((hi[1] - lo[1]) * A.invdx + (G[c][1] - g[c][1]) * A.invdy);
        my -= dt * ((hi[2] - lo[2]) * A.invdx + (G[c][2] - g[c][2]) * A.invdy);
        h = fmaxf(h, 1e-6f);
        const float ih = __builtin_amdgcn_rcpf(h);
        un[c] = mx * ih; vn[c] = my * ih; hn[c] = h;
      }
    }
    // viscosity on row r-2, store
    const int o = r - 2;
    {
      float u[2] = {w1a[0], w1a[1]}, v[2] = {w1b[0], w1b[1]};
      if (A.do_visc) {
        const float ul = __shfl_up(w1a[1], 1, 64), ur = __shfl_down(w1a[0], 1, 64);
        const float vl = __shfl_up(w1b[1], 1, 64), vr = __shfl_down(w1b[0], 1, 64);
        const float lu0 = (w1a[1] - 2.0f * w1a[0] + ul) * A.invdx2 + (un[0] - 2.0f * w1a[0] + w0a[0]) * invdy2;
        const float lu1 = (ur - 2.0f * w1a[1] + w1a[0]) * A.invdx2 + (un[1] - 2.0f * w1a[1] + w0a[1]) * invdy2;
        const float lv0 = (w1b[1] - 2.0f * w1b[0] + vl) * A.invdx2 + (vn[0] - 2.0f * w1b[0] + w0b[0]) * invdy2;
        const float lv1 = (vr - 2.0f * w1b[1] + w1b[0]) * A.invdx2 + (vn[1] - 2.0f * w1b[1] + w0b[1]) * invdy2;
        u[0] += nudt * lu0; u[1] += nudt * lu1; v[0] += nudt * lv0; v[1] += nudt * lv1;
      }
      if (own && o >= j0 && o < j1) {
        const size_t gi = (size_t)o * A.nx + xo;
        if (KIND == K_BURGERS) {
          *reinterpret_cast<float2 *>(A.out[0] + gi) = make_float2(fasinh(u[0] * A.inv_u0), fasinh(u[1] * A.inv_u0));
          *reinterpret_cast<float2 *>(A.out[1] + gi) = make_float2(fasinh(v[0] * A.inv_u0), fasinh(v[1] * A.inv_u0));
          const float wy = (A.ny > 1) ? A.invdy : 0.0f;
          red = fmaxf(red, fmaxf(fabsf(u[0]) * A.invdx + fabsf(v[0]) * wy, fabsf(u[1]) * A.invdx + fabsf(v[1]) * wy));
        } else {
          *reinterpret_cast<float2 *>(A.out[0] + gi) = make_float2(__builtin_amdgcn_logf(w1h[0]) * 0.69314718055994531f,
                                                                  __builtin_amdgcn_logf(w1h[1]) * 0.69314718055994531f);
          *reinterpret_cast<float2 *>(A.out[1] + gi) = make_float2(u[0], u[1]);
          *reinterpret_cast<float2 *>(A.out[2] + gi) = make_float2(v[0], v[1]);
          const float c0 = __builtin_amdgcn_sqrtf(A.g * w1h[0]), c1 = __builtin_amdgcn_sqrtf(A.g * w1h[1]);
          red = fmaxf(red, fmaxf(fmaxf(fabsf(u[0]) + c0, fabsf(v[0]) + c0), fmaxf(fabsf(u[1]) + c1, fabsf(v[1]) + c1)));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      w0a[c] = w1a[c]; w0b[c] = w1b[c]; w1a[c] = un[c]; w1b[c] = vn[c]; w1h[c] = hn[c];
      g[c][0] = G[c][0]; g[c][1] = G[c][1]; g[c][2] = G[c][2];
      P[c] = N[c]; N[c] = nxt[c];
    }
  }
  if (A.reduce) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) red = fmaxf(red, __shfl_xor(red, o, 64));
    if (lane == 0) tau::atomic_max_float_bits(&A.st->maxbits[(A.slot + 1) % 3], red);
  }
}

// MUSCL Burgers as a march.  The limited reconstruction works on the ENCODED phi of four cells along the axis and
// decodes the two face states (four sinh per face), so phi is carried raw: a four-row window for the y faces
// (face a-2 | a-1 needs rows a-3 .. a), lanes l-2 .. l+1 for the x faces, three halo lanes a side (own 58 columns).
constexpr int MCOLS_M = 58;
__global__ __launch_bounds__(256) void k_march_muscl(const Args A, int rows, int nstrips, int nchunks, const int *__restrict__ crow) {
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(nstrips * nchunks);
  const unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  float dt;
  if (A.dt_explicit > 0.f) dt = A.dt_explicit;
  else {
    float m = __uint_as_float(A.st->maxbits[A.slot]);
    if (!(m >= 1e-12f)) m = 1e-12f;
    dt = fminf(A.dt_try, A.CFL * A.cfl_len / m);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.st->dt_last = dt;
    A.st->maxbits[(A.slot + 2) % 3] = 0u;
  }
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)nstrips), chunk = (int)(wid / (unsigned)nstrips);
  const int xo = strip * MCOLS_M + lane - 3;
  const int col = ((xo % A.nx) + A.nx) % A.nx;
  const bool own = lane >= 3 && lane < 3 + MCOLS_M && xo < A.nx;
  const int j0 = crow ? crow[chunk] : chunk * rows, j1 = crow ? crow[chunk + 1] : min(j0 + rows, A.ny);   // (tau::guided_chunks, or `rows` each)
  if (j0 >= j1) return;
  const bool oneD = A.oneD != 0;
  const float invdy = oneD ? 0.0f : A.invdy, invdy2 = oneD ? 0.0f : A.invdy2;
  const float nudt = A.nu * (dt * A.visc_frac);
  auto wrapy = [&](int r) { return ((r % A.ny) + A.ny) % A.ny; };
  auto load = [&](int r, float &pu, float &pv) { const size_t gi = (size_t)wrapy(r) * A.nx + col; pu = A.in[0][gi]; pv = A.in[1][gi]; };

  float u3, v3, u2, v2, u1, v1, un_, vn_;          // phi of rows a-3, a-2, a-1, a
  load(j0 - 3, u2, v2); load(j0 - 2, u1, v1); load(j0 - 1, un_, vn_);
  u3 = u2; v3 = v2;
  float nu_, nv_;
  load(j0, nu_, nv_);
  float g0 = 0.f, g1 = 0.f;                         // y-face flux below row a-2
  float w0a = 0.f, w0b = 0.f, w1a = 0.f, w1b = 0.f; // updated (u, v) of rows a-4, a-3
  float red = 0.f;
  for (int a = j0; a <= j1 + 2; a++) {
    u3 = u2; v3 = v2; u2 = u1; v2 = v1; u1 = un_; v1 = vn_; un_ = nu_; vn_ = nv_;   // window = rows a-3 .. a
    if (a < j1 + 2) load(a + 1, nu_, nv_);
    // y face between rows a-2 and a-1 (cells a-3, a-2 | a-1, a)
    float G0 = 0.f, G1 = 0.f;
    if (!oneD) burgers_face<true>(A, u3, u2, u1, un_, v3, v2, v1, vn_, 1, G0, G1);
    // x faces of row a-2: the face below this lane needs lanes l-2, l-1 | l, l+1
    const float ul2 = __shfl_up(u2, 2, 64), ul1 = __shfl_up(u2, 1, 64), ur1 = __shfl_down(u2, 1, 64);
    const float vl2 = __shfl_up(v2, 2, 64), vl1 = __shfl_up(v2, 1, 64), vr1 = __shfl_down(v2, 1, 64);
    float F0, F1;
    burgers_face<true>(A, ul2, ul1, u2, ur1, vl2, vl1, v2, vr1, 0, F0, F1);
    const float F0h = __shfl_down(F0, 1, 64), F1h = __shfl_down(F1, 1, 64);
    // conservative update of row a-2 (update_convective, :458-487)
    const float uc0 = A.u0 * fsinh(u2), vc0 = A.u0 * fsinh(v2);
    const float unew = uc0 - dt * ((F0h - F0) * A.invdx + (G0 - g0) * invdy);
    const float vnew = vc0 - dt * ((F1h - F1) * A.invdx + (G1 - g1) * invdy);
    // viscosity pass on row a-3 from the updated rows a-4 (w0), a-3 (w1), a-2 (new), then store it
    const int o = a - 3;
    {
      float u = w1a, v = w1b;
      if (A.do_visc) {
        const float ul = __shfl_up(w1a, 1, 64), ur = __shfl_down(w1a, 1, 64);
        const float vl = __shfl_up(w1b, 1, 64), vr = __shfl_down(w1b, 1, 64);
        const float lu = (ur - 2.0f * u + ul) * A.invdx2 + (unew - 2.0f * u + w0a) * invdy2;
        const float lv = (vr - 2.0f * v + vl) * A.invdx2 + (vnew - 2.0f * v + w0b) * invdy2;
        u += nudt * lu;
        v += nudt * lv;
      }
      if (own && o >= j0 && o < j1) {
        const size_t gi = (size_t)o * A.nx + xo;
        A.out[0][gi] = fasinh(u * A.inv_u0);
        A.out[1][gi] = fasinh(v * A.inv_u0);
        red = fmaxf(red, fabsf(u) * A.invdx + fabsf(v) * ((A.ny > 1) ? A.invdy : 0.0f));
      }
    }
    w0a = w1a; w0b = w1b; w1a = unew; w1b = vnew;
    g0 = G0; g1 = G1;
  }
  if (A.reduce) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) red = fmaxf(red, __shfl_xor(red, o, 64));
    if (lane == 0) tau::atomic_max_float_bits(&A.st->maxbits[(A.slot + 1) % 3], red);
  }
}

// wavespeed metric of a state (first step after init / upload, and after extra Burgers viscosity passes)
template <int KIND>
__global__ __launch_bounds__(256) void k_metric(const Args A, int slot) {
  __shared__ float sRed[4];
  const size_t n = (size_t)A.nx * A.ny;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (KIND == K_BURGERS) {
      float u = A.u0 * fsinh(A.in[0][i]), v = A.u0 * fsinh(A.in[1][i]);
      m = fmaxf(m, fabsf(u) * A.invdx + fabsf(v) * ((A.ny > 1) ? A.invdy : 0.0f));
    } else {
      float c = sqrtf(A.g * expf(A.in[0][i]));
      m = fmaxf(m, fmaxf(fabsf(A.in[1][i]) + c, fabsf(A.in[2][i]) + c));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
    tau::atomic_max_float_bits(&A.st->maxbits[slot], m);
  }
}

} // namespace fl2

// =====================================================================================
// C-ABI
// =====================================================================================
struct tauflow {
  tauflow_params p;
  int kind, nf, device;
  hipStream_t stream;
  bool own_stream;
  float *buf[2][3];
  fl2::DevState *st;
  int cur;
  bool max_valid;
  int slot;          // max slot of the current state (DevState::maxbits)
  float t, tau;
  long step;
  taulap_t *visc;   // Burgers: extra viscosity passes (K > 1) through the marching kernel
  int *crow = nullptr;   // chunk schedule of the march (flow_schedule)
  int crow_n = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // tauflow_timer_*: device time of a stretch of launches (lazy)
};

extern "C" void tauflow_params_default(tauflow_params *P, int kind, int nx, int ny) {
  memset(P, 0, sizeof(*P));
  P->nx = nx; P->ny = ny; P->dx = 1.0f; P->dy = 1.0f;
  if (kind == 0) { // tau_burgers.cu:56-91
    P->nu = 0.1f; P->u0 = 1.0f; P->CFL = 0.45f; P->tau0 = 0.0f; P->t0 = 1.0f; P->dtau = 1.0f;
    P->muscl = 0; P->visc_substeps = 1; P->oneD = 0;
    P->amp = 1.0f; P->bsig = 16.0f; P->swirl = 10.0f; P->rc = 40.0f; P->offx = 0.0f; P->offy = 0.0f; P->asym = 0.0f;
    P->ck = 4; P->ca = 0.5f;
  } else { // tau_shallow_water.cu:54-88
    P->g = 9.81f; P->nu = 0.001f; P->H0 = 1000.0f; P->amp = 1.0f; P->bsig = 1.0f; P->CFL = 0.5f;
    P->offx = 100.0f; P->offy = 100.0f; P->asym = 10.0f; P->swirl = 1.0f; P->rc = 100.0f;
    P->tau0 = 0.0f; P->t0 = 1.0f; P->dtau = 1.0f; P->u0 = 1.0f; P->visc_substeps = 1;
  }
}

extern "C" int tauflow_create(tauflow_t **out, const tauflow_params *P, int kind, int device, void *stream) {
  if (!out || !P) return tau::fail("tauflow_create: null argument");
  if (kind != 0 && kind != 1) return tau::fail("tauflow_create: kind must be 0 (Burgers) or 1 (shallow water)");
  if (P->nx < 1 || P->ny < 1) return tau::fail("tauflow_create: bad grid");
  TAU_HIP(hipSetDevice(device));
  tauflow *h = new (std::nothrow) tauflow();
  if (!h) return tau::fail("tauflow_create: out of host memory");
  tau::HandleGuard<tauflow> guard{h, tauflow_destroy};
  h->p = *P; h->kind = kind; h->nf = kind == 0 ? 2 : 3; h->device = device; h->cur = 0; h->slot = 0; h->max_valid = false;
  if (kind == 0 && h->p.oneD) h->p.ny = 1; // tau_burgers.cu:654-655
  h->own_stream = (stream == nullptr);
  if (h->own_stream) TAU_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  else h->stream = (hipStream_t)stream;
  size_t n = (size_t)h->p.nx * h->p.ny;
  for (int s = 0; s < 2; s++)
    for (int f = 0; f < h->nf; f++) TAU_HIP(hipMalloc(&h->buf[s][f], n * sizeof(float)));
  TAU_HIP(hipMalloc(&h->st, sizeof(fl2::DevState)));
  TAU_HIP(hipMemsetAsync(h->st, 0, sizeof(fl2::DevState), h->stream));
  h->tau = P->tau0; h->t = P->t0; h->step = 0; h->visc = nullptr;
  *out = guard.release();
  return 0;
}
extern "C" void tauflow_destroy(tauflow_t *h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  for (int s = 0; s < 2; s++)
    for (int f = 0; f < h->nf; f++) hipFree(h->buf[s][f]);
  hipFree(h->st); hipFree(h->crow);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
}
extern "C" int tauflow_upload(tauflow_t *h, const float *const f[3]) {
  TAU_HIP(hipSetDevice(h->device));
  size_t b = (size_t)h->p.nx * h->p.ny * sizeof(float);
  for (int k = 0; k < h->nf; k++) TAU_HIP(hipMemcpyAsync(h->buf[h->cur][k], f[k], b, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  h->max_valid = false;
  return 0;
}
extern "C" int tauflow_download(tauflow_t *h, float *const f[3]) {
  TAU_HIP(hipSetDevice(h->device));
  size_t b = (size_t)h->p.nx * h->p.ny * sizeof(float);
  for (int k = 0; k < h->nf; k++) TAU_HIP(hipMemcpyAsync(f[k], h->buf[h->cur][k], b, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tauflow_state_ptrs(tauflow_t *h, float *f[3]) {
  for (int k = 0; k < 3; k++) f[k] = k < h->nf ? h->buf[h->cur][k] : nullptr;
  return 0;
}

extern "C" int tauflow_init(tauflow_t *h) { // initialize_host: tau_burgers.cu:246-302 / tau_shallow_water.cu:238-276
  const tauflow_params &P = h->p;
  const int nx = P.nx, ny = P.ny;
  size_t n = (size_t)nx * ny;
  std::vector<float> a(n), b(n), c(n);
  if (h->kind == 0) {
    if (P.oneD) {
      float Lx = P.dx * nx, k = 2.0f * (float)M_PI * P.ck / Lx;
      for (int i = 0; i < nx; ++i) {
        float x = (i + 0.5f) * P.dx, denom = 1.0f + P.ca * cosf(k * x);
        float u = (denom != 0.0f) ? (2.0f * P.nu * P.ca * k * sinf(k * x) / denom) : 0.0f;
        float phi = asinhf(u / P.u0);
        for (int j = 0; j < ny; ++j) { a[(size_t)j * nx + i] = phi; b[(size_t)j * nx + i] = 0.0f; }
      }
    } else {
      float cx = 0.5f * nx + P.offx, cy = 0.5f * ny + P.offy, sig2 = P.bsig * P.bsig, rc = P.rc * fminf(P.dx, P.dy);
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
          float dx = i - cx, dy = j - cy;
          float r2 = (dx * dx + dy * dy) / fmaxf(sig2, 1e-6f);
          float theta = atan2f(dy, dx), mod = 1.0f + P.asym * cosf(theta);
          float rx = dx * P.dx, ry = dy * P.dy, r = sqrtf(rx * rx + ry * ry);
          float u_theta = (r > 0.0f) ? (P.swirl * r * expf(-0.5f * (r / rc) * (r / rc))) : 0.0f;
          float u = (r > 0.0f) ? (-u_theta * (ry / r)) : 0.0f, v = (r > 0.0f) ? (u_theta * (rx / r)) : 0.0f;
          float g = P.amp * mod * expf(-0.5f * r2);
          u += 0.5f * g; v += -0.5f * g;
          a[(size_t)j * nx + i] = asinhf(u / P.u0);
          b[(size_t)j * nx + i] = asinhf(v / P.u0);
        }
    }
  } else {
    float cx = 0.5f * nx + P.offx, cy = 0.5f * ny + P.offy, sig2 = P.bsig * P.bsig;
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        float dx = i - cx, dy = j - cy;
        float r2 = (dx * dx + dy * dy) / sig2;
        float theta = atan2f(dy, dx), mod = 1.0f + P.asym * cosf(theta);
        float hh = P.H0 + (P.amp * mod) * expf(-0.5f * r2);
        size_t id = (size_t)j * nx + i;
        a[id] = logf(fmaxf(hh, 1e-6f));
        float rx = dx * P.dx, ry = dy * P.dy, r = sqrtf(rx * rx + ry * ry), rc = P.rc * fminf(P.dx, P.dy);
        float u_theta = (r > 0.0f && P.swirl != 0.0f) ? (P.swirl * r * expf(-0.5f * (r / rc) * (r / rc))) : 0.0f;
        b[id] = (r > 0.0f) ? (-u_theta * (ry / r)) : 0.0f;
        c[id] = (r > 0.0f) ? (u_theta * (rx / r)) : 0.0f;
      }
  }
  const float *f[3] = {a.data(), b.data(), c.data()};
  h->tau = P.tau0; h->t = P.t0; h->step = 0;
  return tauflow_upload(h, f);
}

static void flow_args(tauflow *h, fl2::Args &A, float dt_explicit) {
  const tauflow_params &P = h->p;
  memset(&A, 0, sizeof(A));
  for (int f = 0; f < h->nf; f++) { A.in[f] = h->buf[h->cur][f]; A.out[f] = h->buf[h->cur ^ 1][f]; }
  A.st = h->st; A.nx = P.nx; A.ny = P.ny; A.ntx = (P.nx + fl2::TX - 1) / fl2::TX; A.nty = (P.ny + fl2::TY - 1) / fl2::TY;
  A.slot = h->slot; A.dx = P.dx; A.dy = P.dy; A.invdx = 1.0f / P.dx; A.invdy = 1.0f / P.dy;
  A.invdx2 = 1.0f / (P.dx * P.dx); A.invdy2 = 1.0f / (P.dy * P.dy);
  A.nu = P.nu; A.u0 = P.u0; A.inv_u0 = 1.0f / P.u0; A.g = P.g; A.CFL = P.CFL;
  A.dt_try = h->t * P.dtau; A.dt_explicit = dt_explicit;
  A.cfl_len = (h->kind == 0) ? 1.0f : fminf(P.dx, P.dy);
  A.muscl = P.muscl; A.oneD = (h->kind == 0) ? P.oneD : 0;
  const int K = (h->kind == 0) ? (P.visc_substeps > 0 ? P.visc_substeps : 1) : 1;
  A.do_visc = (h->kind == 0) ? 1 : (P.nu > 0.0f);
  A.visc_frac = 1.0f / (float)K;
  A.reduce = (K == 1);
}

// chunk schedule of the marches: tau::guided_chunks over the wave slots the kernel really gets, 12..64 rows.  Against chunks of
// one length (round 4, Gcell/s): 8192^2 shallow water 162 -> 178, Burgers 243 -> 260, --muscl 111 -> 113; 4096^2 / 2048^2 level
// (shorter minimum chunks lose there: 8 rows 141 -> 135 at 4096^2).  TAU_FLOW_GUIDED=0 / TAU_FLOW_ROWS=n: uniform chunks.
static int flow_schedule(tauflow *h, const void *fn, int nstrips, int *nchunks, const int **crow) {
  static const bool guided = !(getenv("TAU_FLOW_GUIDED") && atoi(getenv("TAU_FLOW_GUIDED")) == 0);
  if (!guided) return 0;
  if (!h->crow) {
    int per_cu = 0;
    TAU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0));
    static const int gmin = getenv("TAU_FLOW_GMIN") ? atoi(getenv("TAU_FLOW_GMIN")) : 12;
    static const int gmax = getenv("TAU_FLOW_GMAX") ? atoi(getenv("TAU_FLOW_GMAX")) : 64;
    if (tau::guided_chunks(h->p.ny, nstrips, (per_cu > 0 ? per_cu : 2) * 4 * 32, gmin, gmax, &h->crow, &h->crow_n)) return 1;
  }
  *nchunks = h->crow_n; *crow = h->crow;
  return 0;
}
static int flow_step_once(tauflow *h, float dt_explicit) {
  fl2::Args A;
  flow_args(h, A, dt_explicit);
  const tauflow_params &P = h->p;
  if (!h->max_valid) {
    TAU_HIP(hipMemsetAsync(h->st->maxbits, 0, sizeof(h->st->maxbits), h->stream));
    if (h->kind == 0) hipLaunchKernelGGL(fl2::k_metric<fl2::K_BURGERS>, dim3(1024), dim3(256), 0, h->stream, A, h->slot);
    else hipLaunchKernelGGL(fl2::k_metric<fl2::K_SW>, dim3(1024), dim3(256), 0, h->stream, A, h->slot);
    TAU_LAUNCH_CHECK("fl2::k_metric");
    h->max_valid = true;
  }
  const unsigned nb = (unsigned)(A.ntx * A.nty);
  static const int use_march = [] { const char *e = getenv("TAU_FLOW_MARCH"); return e ? atoi(e) : 1; }();
  // plain Burgers, shallow water: the marching kernel from ~2 M cells on (a wave walks its strip serially, so it needs
  // thousands of strips x chunks to fill the chip: 1024^2 24.6 vs 15.9 us per step, 1536^2 28.8 vs 29.4, 4096^2 103 vs 146);
  // TAU_FLOW_MARCH=2 forces it at any size
  if (use_march && !(h->kind == 0 && A.muscl) && P.nx >= 8 && P.ny >= 4 && (use_march > 1 || (long)P.nx * P.ny >= (1L << 21))) {
    static const int two = [] { const char *e = getenv("TAU_FLOW_COLS2"); return e ? atoi(e) : 1; }();
    const bool pair = two && (P.nx % 2 == 0) && P.nx >= 16;        // two columns per lane
    const int nstrips = pair ? (P.nx + fl2::MCOLS2 - 1) / fl2::MCOLS2 : (P.nx + fl2::MCOLS - 1) / fl2::MCOLS;
    int rows = (int)((long)P.ny * nstrips / 8192);                 // ~8k waves at least, chunks of 8..48 rows (8192^2: 32-48
    rows = rows < 8 ? 8 : (rows > 48 ? 48 : rows);                 // rows 184 Gcell/s, 16: 175, 64: 179, 128: 154)
    static const int rows_env = [] { const char *e = getenv("TAU_FLOW_ROWS"); return e ? atoi(e) : 0; }();
    if (rows_env >= 1) rows = rows_env;
    int nchunks = (P.ny + rows - 1) / rows;
    const int *crow = nullptr;
    const void *fn = pair ? (h->kind == 0 ? (const void *)fl2::k_march2<fl2::K_BURGERS> : (const void *)fl2::k_march2<fl2::K_SW>)
                          : (h->kind == 0 ? (const void *)fl2::k_march<fl2::K_BURGERS> : (const void *)fl2::k_march<fl2::K_SW>);
    if (rows_env < 1 && flow_schedule(h, fn, nstrips, &nchunks, &crow)) return 1;
    const unsigned nwg = (unsigned)((nstrips * nchunks + 3) / 4);
    if (pair && h->kind == 0) hipLaunchKernelGGL(fl2::k_march2<fl2::K_BURGERS>, dim3(nwg), dim3(256), 0, h->stream, A, rows, nstrips, nchunks, crow);
    else if (pair) hipLaunchKernelGGL(fl2::k_march2<fl2::K_SW>, dim3(nwg), dim3(256), 0, h->stream, A, rows, nstrips, nchunks, crow);
    else if (h->kind == 0) hipLaunchKernelGGL(fl2::k_march<fl2::K_BURGERS>, dim3(nwg), dim3(256), 0, h->stream, A, rows, nstrips, nchunks, crow);
    else hipLaunchKernelGGL(fl2::k_march<fl2::K_SW>, dim3(nwg), dim3(256), 0, h->stream, A, rows, nstrips, nchunks, crow);
  }
  else if (use_march && h->kind == 0 && A.muscl && P.nx >= 8 && P.ny >= 4 && (use_march > 1 || (long)P.nx * P.ny >= (1L << 21))) {
    const int nstrips = (P.nx + fl2::MCOLS_M - 1) / fl2::MCOLS_M;
    int rows = (int)((long)P.ny * nstrips / 8192);
    rows = rows < 8 ? 8 : (rows > 48 ? 48 : rows);
    static const int rows_env = [] { const char *e = getenv("TAU_FLOW_ROWS"); return e ? atoi(e) : 0; }();
    if (rows_env >= 1) rows = rows_env;
    int nchunks = (P.ny + rows - 1) / rows;
    const int *crow = nullptr;
    if (rows_env < 1 && flow_schedule(h, (const void *)fl2::k_march_muscl, nstrips, &nchunks, &crow)) return 1;
    hipLaunchKernelGGL(fl2::k_march_muscl, dim3((unsigned)((nstrips * nchunks + 3) / 4)), dim3(256), 0, h->stream, A, rows, nstrips, nchunks, crow);
  }
  else if (h->kind == 0 && A.muscl) hipLaunchKernelGGL((fl2::k_step<fl2::K_BURGERS, true>), dim3(nb), dim3(fl2::NT), 0, h->stream, A);
  else if (h->kind == 0) hipLaunchKernelGGL((fl2::k_step<fl2::K_BURGERS, false>), dim3(nb), dim3(fl2::NT), 0, h->stream, A);
  else hipLaunchKernelGGL((fl2::k_step<fl2::K_SW, false>), dim3(nb), dim3(fl2::NT), 0, h->stream, A);
  TAU_LAUNCH_CHECK("fl2::k_step");
  h->cur ^= 1;
  h->slot = (h->slot + 1) % 3;
  const int K = (h->kind == 0) ? (P.visc_substeps > 0 ? P.visc_substeps : 1) : 1;
  if (K > 1) { // remaining viscosity passes (marching kernel), then the metric of the final state
    for (int k = 1; k < K; k++) {
      if (tau::st2_burgers_pass(h->buf[h->cur][0], h->buf[h->cur][1], h->buf[h->cur ^ 1][0], h->buf[h->cur ^ 1][1], P.nx,
                                P.ny, P.dx, P.dy, P.nu, P.u0, P.oneD, &h->st->dt_last, 1.0f / (float)K, h->stream)) return 1;
      h->cur ^= 1;
    }
    fl2::Args B;
    flow_args(h, B, 0.f);
    TAU_HIP(hipMemsetAsync(&h->st->maxbits[h->slot], 0, sizeof(unsigned), h->stream));
    hipLaunchKernelGGL(fl2::k_metric<fl2::K_BURGERS>, dim3(1024), dim3(256), 0, h->stream, B, h->slot);
    TAU_LAUNCH_CHECK("fl2::k_metric");
  }
  return 0;
}

extern "C" int tauflow_step_async(tauflow_t *h, int nsteps) { // headless loop body, tau_burgers.cu:797-803 / tau_shallow_water.cu
  TAU_HIP(hipSetDevice(h->device));
  for (int s = 0; s < nsteps; s++) {
    if (flow_step_once(h, 0.f)) return 1;
    h->tau += h->p.dtau;
    h->t *= expf(h->p.dtau);
    h->step++;
  }
  return 0;
}
/* Device time of what is enqueued between the two calls, from events on the handle's stream — what the reference's headless
 * summaries print as "GPU" / "GPU only" (cudaEvent pairs, tau_burgers.cu:790-820, tau_shallow_water.cu:751-782).  stop
 * waits for the stream. */
extern "C" int tauflow_timer_start(tauflow_t *h) {
  if (!h) return tau::fail("tauflow_timer_start: null handle");
  TAU_HIP(hipSetDevice(h->device));
  if (!h->ev0) { TAU_HIP(hipEventCreate(&h->ev0)); TAU_HIP(hipEventCreate(&h->ev1)); }
  TAU_HIP(hipEventRecord(h->ev0, h->stream));
  return 0;
}
extern "C" int tauflow_timer_stop(tauflow_t *h, double *ms) {
  if (!h || !ms) return tau::fail("tauflow_timer_stop: null argument");
  if (!h->ev0) return tau::fail("tauflow_timer_stop: no tauflow_timer_start before it");
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipEventRecord(h->ev1, h->stream));
  TAU_HIP(hipEventSynchronize(h->ev1));
  float f = 0.f;
  TAU_HIP(hipEventElapsedTime(&f, h->ev0, h->ev1));
  *ms = (double)f;
  return 0;
}
extern "C" int tauflow_sync(tauflow_t *h) {
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tauflow_step(tauflow_t *h, int nsteps) {
  if (tauflow_step_async(h, nsteps)) return 1;
  return tauflow_sync(h);
}
extern "C" int tauflow_step_explicit(tauflow_t *h, float dt) {
  if (!(dt > 0.f)) return tau::fail("tauflow_step_explicit: dt must be positive");
  TAU_HIP(hipSetDevice(h->device));
  if (flow_step_once(h, dt)) return 1;
  return tauflow_sync(h);
}
extern "C" int tauflow_get_clock(tauflow_t *h, float *t, float *tau, float *dt_last, float *wavespeed, int64_t *step) {
  TAU_HIP(hipSetDevice(h->device));
  fl2::DevState s;
  TAU_HIP(hipMemcpyAsync(&s, h->st, sizeof(s), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  if (t) *t = h->t;
  if (tau) *tau = h->tau;
  if (dt_last) *dt_last = s.dt_last;
  if (wavespeed) memcpy(wavespeed, &s.maxbits[h->slot], 4);
  if (step) *step = h->step;
  return 0;
}
/* relative L2 error against the exact 1-D Cole-Hopf solution, tau_burgers.cu:720-736 */
extern "C" int tauflow_colehopf_relL2(tauflow_t *h, float t_now, double *rel) {
  if (h->kind != 0) return tau::fail("tauflow_colehopf_relL2: Burgers only");
  const tauflow_params &P = h->p;
  std::vector<float> a((size_t)P.nx * P.ny), b((size_t)P.nx * P.ny);
  float *f[3] = {a.data(), b.data(), nullptr};
  if (tauflow_download(h, f)) return 1;
  float Lx = P.dx * P.nx, k = 2.0f * (float)M_PI * P.ck / Lx, decay = expf(-P.nu * k * k * t_now);
  double num = 0.0, den = 0.0;
  for (int i = 0; i < P.nx; ++i) {
    float x = (i + 0.5f) * P.dx;
    float u_ex = (2.0f * P.nu * P.ca * k * decay * sinf(k * x)) / (1.0f + P.ca * decay * cosf(k * x));
    double diff = P.u0 * sinh((double)a[i]) - u_ex;
    num += diff * diff; den += u_ex * u_ex;
  }
  *rel = (den > 0.0) ? sqrt(num / den) : sqrt(num);
  return 0;
}
