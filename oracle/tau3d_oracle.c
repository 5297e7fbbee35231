/* tau3d_oracle.c — TEST INFRASTRUCTURE ONLY (CPU oracle, not a product path).
 *
 * Plain-C restatement of the 3D two-temperature hypersonic step of the
 * reference (tau_hypersonic_3d_cuda.cu), IEEE fp32 host semantics: expf/logf/
 * sinhf from libm, no FMA contraction (build with -ffp-contract=off), every
 * expression kept in the reference's association order so the result is the
 * reference's own arithmetic evaluated on the host.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 *
 * Parity pin: (GPU box) the reference's own k_build_solid_mask / k_init / k_step, built for gfx950 from th3cs.cu by
 * oracle/build_ref.sh and run on the MI355X: this file's step lands within 2e-6 (rho, m, E; lam / kappa 7e-7) of them
 * on developed states, its mask bit for bit (tests/test_gpu_ref3d.py).  (CPU) SURVEY.md §8(c) check-values
 * (outputs of the reference source, 32^3, 4 and 400 steps) — tests/golden/ref_checkvalues.json,
 * tests/test_oracle_pins.py.
 *
 * Layout (same as the engine): every field is a local Z-slab with a 3-plane
 * halo on both sides, index ((zl+3)*ny + y)*nx + x for zl in [-3, nzl+3).
 * Plane zl holds GLOBAL plane wrap(z0 + zl, nz) (z is periodic,
 * tau_hypersonic_3d_cuda.cu:729-730, 1029-1030).  With z0 = 0, nzl = nz and
 * o3_fill_halo_periodic() this is exactly the reference's single domain.
 */
#include "../include/tau_params.h"
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HALO 3                      /* WENO_HALO, :58 */
#define RHO_P_FLOOR 1e-30f          /* :52 */
#define THERMAL_ENERGY_FLOOR 1e-12f /* :53 */
#define DENOM_EPS 1e-12f            /* :54 */
#define NEWTON_TEMP_FLOOR 1e-6f     /* :55 */
#define WENO_EPS 1e-6f              /* :56 */
#define TAU_VIB_MIN 1e-9f           /* :57 */

typedef struct { float r, u, v, w, p, ev; } prim_t;       /* Prim :48-50 (T, Tv never read back) */
typedef struct { float r, mx, my, mz, Et, Ev; } cons_t;   /* Cons :44-46 */

static inline float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
static inline float denom_guard(float x) { return copysignf(fmaxf(fabsf(x), DENOM_EPS), x); } /* :147-150 */
static inline int wrapi(int i, int n) { i %= n; return (i < 0) ? i + n : i; }                  /* :156-159 */

/* :121-125, 131-133 */
static inline float asinh_enc(float x) {
  float ax = fabsf(x);
  float t = logf(ax + sqrtf(ax * ax + 1.0f));
  return copysignf(t, x);
}

/* :206-211 */
static inline float evib_eq(const tau3d_params *P, float T) {
  float a = P->theta_v / fmaxf(T, NEWTON_TEMP_FLOOR);
  float ea = expf(a);
  float denom = fmaxf(ea - 1.f, NEWTON_TEMP_FLOOR);
  return (P->R * P->theta_v) / denom;
}

/* :213-225 */
static inline prim_t decode(const tau3d_params *P, float xi, float phx, float phy, float phz,
                            float lam, float zet) {
  prim_t q;
  q.r = expf(xi);
  q.u = P->u_ref * sinhf(phx);
  q.v = P->u_ref * sinhf(phy);
  q.w = P->u_ref * sinhf(phz);
  q.p = expf(lam);
  q.ev = expf(zet);
  return q;
}

/* :511-521 */
static inline void apply_wall(const tau3d_params *P, prim_t *q) {
  float p_keep = fmaxf(q->p, RHO_P_FLOOR);
  q->u = 0.f; q->v = 0.f; q->w = 0.f;
  q->p = p_keep;
  q->r = fmaxf(q->p / (P->R * fmaxf(P->Twall, NEWTON_TEMP_FLOOR)), RHO_P_FLOOR);
  q->ev = evib_eq(P, P->Twall);
}

/* :611-622 */
static inline prim_t inflow_prim(const tau3d_params *P) {
  prim_t q;
  q.r = fmaxf(P->inflow_r, RHO_P_FLOOR);
  q.u = P->inflow_u; q.v = P->inflow_v; q.w = P->inflow_w;
  q.p = fmaxf(P->inflow_p, RHO_P_FLOOR);
  float T = q.p / (q.r * P->R);
  q.ev = evib_eq(P, T);
  return q;
}

/* :264-266 */
static inline float soundspeed(const tau3d_params *P, const prim_t *q) {
  return sqrtf(fmaxf(P->gamma_floor * q->p / q->r, DENOM_EPS));
}

/* :691-722 — ghost state right of x = nx-1, built from the last interior cell */
static inline prim_t outflow_prim(const tau3d_params *P, prim_t qR) {
  prim_t q = qR;
  float aR = soundspeed(P, &qR);
  float un = qR.u;
  if (un < 0.0f) return inflow_prim(P);
  if (un < aR) {
    float p_amb = fmaxf(P->inflow_p, RHO_P_FLOOR);
    float relax = 0.05f;
    q.p = fmaxf(q.p + relax * (p_amb - q.p), RHO_P_FLOOR);
  }
  q.r = fmaxf(q.r, RHO_P_FLOOR);
  q.p = fmaxf(q.p, RHO_P_FLOOR);
  q.ev = fmaxf(q.ev, 0.f);
  return q;
}

/* :234-245 */
static inline cons_t prim_to_cons(const tau3d_params *P, const prim_t *q) {
  cons_t U;
  U.r = q->r;
  U.mx = q->r * q->u;
  U.my = q->r * q->v;
  U.mz = q->r * q->w;
  float ke = 0.5f * (q->u * q->u + q->v * q->v + q->w * q->w);
  float e_th = q->p / fmaxf((P->gamma_floor - 1.f) * q->r, RHO_P_FLOOR);
  U.Ev = q->r * q->ev;
  U.Et = q->r * (ke + e_th + q->ev);
  return U;
}

/* :268-308 — physical flux along `axis` */
static inline cons_t axis_flux(const tau3d_params *P, const prim_t *q, int axis) {
  cons_t F;
  float un = (axis == 0) ? q->u : (axis == 1) ? q->v : q->w;
  float H = (q->p / q->r) + (0.5f * (q->u * q->u + q->v * q->v + q->w * q->w) + q->ev) +
            q->p / fmaxf((P->gamma_floor - 1.f) * q->r, RHO_P_FLOOR);
  F.r = q->r * un;
  F.mx = q->r * q->u * un;
  F.my = q->r * q->v * un;
  F.mz = q->r * q->w * un;
  if (axis == 0) F.mx = q->r * q->u * un + q->p;
  if (axis == 1) F.my = q->r * q->v * un + q->p;
  if (axis == 2) F.mz = q->r * q->w * un + q->p;
  F.Et = q->r * H * un;
  F.Ev = q->r * q->ev * un;
  return F;
}

/* :366-374 */
static inline float entropy_fix_speed(float s, float a_ref) {
  float d = 0.1f * a_ref;
  float as = fabsf(s);
  if (as >= d) return s;
  float sgn = (s >= 0.f) ? 1.f : -1.f;
  float sm = 0.5f * (as * as / fmaxf(d, DENOM_EPS) + d);
  return sgn * sm;
}

/* :376-381 */
static inline float shock_sensor(const prim_t *L, const prim_t *R) {
  float dp = fabsf(R->p - L->p) / fmaxf(R->p + L->p, DENOM_EPS);
  float dr = fabsf(R->r - L->r) / fmaxf(R->r + L->r, DENOM_EPS);
  float s = 0.5f * (dp + dr);
  return clampf(5.f * s, 0.f, 1.f);
}

#define C_ADD(a, b) ((cons_t){(a).r + (b).r, (a).mx + (b).mx, (a).my + (b).my, (a).mz + (b).mz, (a).Et + (b).Et, (a).Ev + (b).Ev})
#define C_SUB(a, b) ((cons_t){(a).r - (b).r, (a).mx - (b).mx, (a).my - (b).my, (a).mz - (b).mz, (a).Et - (b).Et, (a).Ev - (b).Ev})
#define C_MUL(a, s) ((cons_t){(a).r * (s), (a).mx * (s), (a).my * (s), (a).mz * (s), (a).Et * (s), (a).Ev * (s)})

/* :383-460 — HLLC blended towards HLL by (shock sensor x flow alignment) */
static cons_t hllc_flux(const tau3d_params *P, const prim_t *L, const prim_t *R, int axis) {
  float aL = soundspeed(P, L), aR = soundspeed(P, R);
  float unL = (axis == 0) ? L->u : (axis == 1) ? L->v : L->w;
  float unR = (axis == 0) ? R->u : (axis == 1) ? R->v : R->w;
  float sL = fminf(unL - aL, unR - aR);
  float sR = fmaxf(unL + aL, unR + aR);
  float aRef = fmaxf(aL, aR);
  sL = entropy_fix_speed(sL, aRef);
  sR = entropy_fix_speed(sR, aRef);

  cons_t UL = prim_to_cons(P, L), UR = prim_to_cons(P, R);
  cons_t FL = axis_flux(P, L, axis), FR = axis_flux(P, R, axis);
  if (sL >= 0.f) return FL;
  if (sR <= 0.f) return FR;

  float rL = L->r, rR = R->r, pL = L->p, pR = R->p;
  float denom = denom_guard(rL * (sL - unL) - rR * (sR - unR));
  float sM = (pR - pL + rL * unL * (sL - unL) - rR * unR * (sR - unR)) / denom;
  float pStarL = pL + rL * (sL - unL) * (sM - unL);
  float pStarR = pR + rR * (sR - unR) * (sM - unR);
  float pStar = 0.5f * (pStarL + pStarR);

  /* :318-325 cross-flow speed */
  float vc;
  if (axis == 0) vc = (fabsf(L->v) + fabsf(R->v) + fabsf(L->w) + fabsf(R->w)) * 0.5f;
  else if (axis == 1) vc = (fabsf(L->u) + fabsf(R->u) + fabsf(L->w) + fabsf(R->w)) * 0.5f;
  else vc = (fabsf(L->u) + fabsf(R->u) + fabsf(L->v) + fabsf(R->v)) * 0.5f;
  float align = clampf(1.f - vc / fmaxf(aRef, DENOM_EPS), 0.f, 1.f);
  float alpha = shock_sensor(L, R) * align;

  cons_t num = C_SUB(C_MUL(FL, sR), C_MUL(FR, sL));
  cons_t dUU = C_SUB(UR, UL);
  cons_t corr = C_MUL(dUU, sL * sR);
  cons_t sum = C_ADD(num, corr);
  cons_t FHLL = C_MUL(sum, 1.f / denom_guard(sR - sL));

  const prim_t *K = (sM >= 0.f) ? L : R;
  const cons_t *UK = (sM >= 0.f) ? &UL : &UR;
  const cons_t *FK = (sM >= 0.f) ? &FL : &FR;
  float sK = (sM >= 0.f) ? sL : sR;
  float unK = (sM >= 0.f) ? unL : unR;
  float rK = K->r, pK = K->p;

  float starDenom = denom_guard(sK - sM);
  float rStar = rK * (sK - unK) / starDenom;
  float EStar = ((sK - unK) * UK->Et - pK * unK + pStar * sM) / starDenom;
  float EvStar = UK->Ev * (sK - unK) / starDenom;
  cons_t US;
  US.r = rStar;
  US.mx = rStar * ((axis == 0) ? sM : K->u);   /* :335-350 */
  US.my = rStar * ((axis == 1) ? sM : K->v);
  US.mz = rStar * ((axis == 2) ? sM : K->w);
  US.Et = EStar;
  US.Ev = EvStar;
  cons_t dS = C_SUB(US, *UK);
  cons_t sdS = C_MUL(dS, sK);
  cons_t FHLLC = C_ADD(*FK, sdS);
  cons_t a = C_MUL(FHLLC, 1.f - alpha);
  cons_t b = C_MUL(FHLL, alpha);
  return C_ADD(a, b);
}

/* :534-558 */
static inline float weno5_left(float v0, float v1, float v2, float v3, float v4) {
  float p0 = (2.f * v0 - 7.f * v1 + 11.f * v2) * (1.f / 6.f);
  float p1 = (-1.f * v1 + 5.f * v2 + 2.f * v3) * (1.f / 6.f);
  float p2 = (2.f * v2 + 5.f * v3 - 1.f * v4) * (1.f / 6.f);
  float b0 = (13.f / 12.f) * (v0 - 2.f * v1 + v2) * (v0 - 2.f * v1 + v2) +
             0.25f * (v0 - 4.f * v1 + 3.f * v2) * (v0 - 4.f * v1 + 3.f * v2);
  float b1 = (13.f / 12.f) * (v1 - 2.f * v2 + v3) * (v1 - 2.f * v2 + v3) +
             0.25f * (v1 - v3) * (v1 - v3);
  float b2 = (13.f / 12.f) * (v2 - 2.f * v3 + v4) * (v2 - 2.f * v3 + v4) +
             0.25f * (3.f * v2 - 4.f * v3 + v4) * (3.f * v2 - 4.f * v3 + v4);
  float eps = WENO_EPS;
  float a0 = 0.1f / ((eps + b0) * (eps + b0));
  float a1 = 0.6f / ((eps + b1) * (eps + b1));
  float a2 = 0.3f / ((eps + b2) * (eps + b2));
  float s = a0 + a1 + a2;
  float w0 = a0 / s, w1 = a1 / s, w2 = a2 / s;
  return w0 * p0 + w1 * p1 + w2 * p2;
}

static inline void prim_floor(prim_t *q) { /* :565-571 */
  q->r = fmaxf(q->r, RHO_P_FLOOR);
  q->p = fmaxf(q->p, RHO_P_FLOOR);
  q->ev = fmaxf(q->ev, 0.f);
}

/* :578-598 — six cells q[0..5] around the face between q[2] and q[3] */
static inline void weno_face(const prim_t *q, prim_t *L, prim_t *R) {
#define WL(f) L->f = weno5_left(q[0].f, q[1].f, q[2].f, q[3].f, q[4].f)
#define WR(f) R->f = weno5_left(q[5].f, q[4].f, q[3].f, q[2].f, q[1].f)
  WL(r); WL(u); WL(v); WL(w); WL(p); WL(ev);
  WR(r); WR(u); WR(v); WR(w); WR(p); WR(ev);
#undef WL
#undef WR
  prim_floor(L);
  prim_floor(R);
}

static inline prim_t mirror(prim_t q, int axis) { /* :772-781 */
  if (axis == 0) q.u = -q.u;
  if (axis == 1) q.v = -q.v;
  if (axis == 2) q.w = -q.w;
  return q;
}

/* Flux through the lower (side = 0) or upper (side = 1) face of the fluid cell in
 * the middle of the 7-cell line c[0..6] (c[3] = the cell), :1115-1264. */
static cons_t face_flux(const tau3d_params *P, const prim_t *c, const uint8_t *s, int axis, int side) {
  if (side == 0) {
    int face_solid = s[2] || s[3];
    int stencil_solid = s[0] || s[1] || s[2] || s[3] || s[4] || s[5];
    if (face_solid) { prim_t R = c[3]; prim_t L = mirror(R, axis); return hllc_flux(P, &L, &R, axis); }
    if (stencil_solid) { prim_t L = c[2], R = c[3]; prim_floor(&L); prim_floor(&R); return hllc_flux(P, &L, &R, axis); }
    prim_t L, R; weno_face(c, &L, &R); return hllc_flux(P, &L, &R, axis);
  } else {
    int face_solid = s[3] || s[4];
    int stencil_solid = s[1] || s[2] || s[3] || s[4] || s[5] || s[6];
    if (face_solid) { prim_t L = c[3]; prim_t R = mirror(L, axis); return hllc_flux(P, &L, &R, axis); }
    if (stencil_solid) { prim_t L = c[3], R = c[4]; prim_floor(&L); prim_floor(&R); return hllc_flux(P, &L, &R, axis); }
    prim_t L, R; weno_face(c + 1, &L, &R); return hllc_flux(P, &L, &R, axis);
  }
}

static inline int sdf_solid(const tau3d_params *P, int x, int y, int z) { /* :173-189, 759-770 */
  float X = (x + 0.5f) * P->dx, Y = (y + 0.5f) * P->dy, Z = (z + 0.5f) * P->dz;
  float ddx = X - P->sdf_cx, ddy = Y - P->sdf_cy, ddz = Z - P->sdf_cz;
  return (sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) - P->sdf_r) < 0.f;
}

/* ------------------------------------------------------------------ public */

size_t o3_local_cells(const tau3d_params *P, int nzl) {
  return (size_t)P->nx * P->ny * (size_t)(nzl + 2 * HALO);
}

/* solid mask of the slab incl. halo planes, from the SDF at wrapped global z */
void o3_build_solid(const tau3d_params *P, int z0, int nzl, uint8_t *solid) {
  for (int zl = -HALO; zl < nzl + HALO; zl++) {
    int zg = wrapi(z0 + zl, P->nz);
    for (int y = 0; y < P->ny; y++)
      for (int x = 0; x < P->nx; x++)
        solid[((size_t)(zl + HALO) * P->ny + y) * P->nx + x] = (uint8_t)sdf_solid(P, x, y, zg);
  }
}

/* k_init, :939-985 — quiescent gas at inflow rho,p; solid cells at wall temperature */
void o3_init(const tau3d_params *P, int nzl, const uint8_t *solid, float *const st[6]) {
  size_t n = o3_local_cells(P, nzl);
  for (size_t i = 0; i < n; i++) {
    float r = fmaxf(P->inflow_r, RHO_P_FLOOR);
    float p = fmaxf(P->inflow_p, RHO_P_FLOOR);
    float T = p / (r * P->R);
    float ev = evib_eq(P, T);
    if (solid[i]) {
      T = P->Twall;
      r = fmaxf(p / (P->R * fmaxf(T, NEWTON_TEMP_FLOOR)), RHO_P_FLOOR);
      ev = evib_eq(P, T);
    }
    st[0][i] = logf(fmaxf(r, RHO_P_FLOOR));
    st[1][i] = asinh_enc(0.f / P->u_ref);
    st[2][i] = asinh_enc(0.f / P->u_ref);
    st[3][i] = asinh_enc(0.f / P->u_ref);
    st[4][i] = logf(fmaxf(p, RHO_P_FLOOR));
    st[5][i] = logf(fmaxf(ev, RHO_P_FLOOR));
  }
}

/* Synthetic developed-flow state (SURVEY §8d input (ii)): every fluid cell at the
 * full inflow state, solid cells at the wall state. */
void o3_init_impulsive(const tau3d_params *P, int nzl, const uint8_t *solid, float *const st[6]) {
  size_t n = o3_local_cells(P, nzl);
  for (size_t i = 0; i < n; i++) {
    prim_t q = inflow_prim(P);
    if (solid[i]) apply_wall(P, &q);
    st[0][i] = logf(fmaxf(q.r, RHO_P_FLOOR));
    st[1][i] = asinh_enc(q.u / P->u_ref);
    st[2][i] = asinh_enc(q.v / P->u_ref);
    st[3][i] = asinh_enc(q.w / P->u_ref);
    st[4][i] = logf(fmaxf(q.p, RHO_P_FLOOR));
    st[5][i] = logf(fmaxf(q.ev, RHO_P_FLOOR));
  }
}

/* single-domain periodic halo (z0 = 0, nzl = nz) */
void o3_fill_halo_periodic(const tau3d_params *P, int nzl, float *const st[6]) {
  size_t plane = (size_t)P->nx * P->ny;
  for (int f = 0; f < 6; f++) {
    memcpy(st[f], st[f] + (size_t)nzl * plane, HALO * plane * sizeof(float));
    memcpy(st[f] + (size_t)(nzl + HALO) * plane, st[f] + HALO * plane, HALO * plane * sizeof(float));
  }
}

/* One k_step (:987-1359) over interior planes [zl_lo, zl_hi) of the slab.
 * Returns the max over updated fluid cells of sum_axis (|u|+a)/dx (0 if none). */
float o3_step(const tau3d_params *P, int z0, int nzl, int zl_lo, int zl_hi, const float *const in[6],
              float *const out[6], const uint8_t *solid, float dt, float inflow_gain) {
  const int nx = P->nx, ny = P->ny;
  const int px = nx + 2 * HALO, py = ny + 2 * HALO;
  const int zp_lo = zl_lo, zp_hi = zl_hi + 2 * HALO; /* padded planes needed, in halo-layout index */
  const int pz = zp_hi - zp_lo;
  size_t pn = (size_t)px * py * pz;
  prim_t *pr = (prim_t *)malloc(pn * sizeof(prim_t));
  uint8_t *ps = (uint8_t *)malloc(pn);
  (void)nzl;

  /* stage A: decode every cell of the padded block once (the reference does this
   * per tile, :1019-1056; the function of a cell is the same wherever it is decoded) */
  for (int zz = 0; zz < pz; zz++) {
    int zh = zp_lo + zz;                          /* plane index in halo layout */
    int zg = wrapi(z0 + zh - HALO, P->nz);        /* wrapped global plane (ghost-x SDF only) */
    for (int yy = 0; yy < py; yy++) {
      int y = wrapi(yy - HALO, ny);
      for (int xx = 0; xx < px; xx++) {
        int x = xx - HALO;
        size_t pi = ((size_t)zz * py + yy) * px + xx;
        prim_t q;
        int sol;
        if (x < 0) {
          q = inflow_prim(P);
          sol = sdf_solid(P, x, y, zg); /* :180-189 — outside the array the mask is the SDF */
        } else if (x >= nx) {
          size_t gi = ((size_t)zh * ny + y) * nx + (nx - 1);
          prim_t qR = decode(P, in[0][gi], in[1][gi], in[2][gi], in[3][gi], in[4][gi], in[5][gi]);
          q = outflow_prim(P, qR);
          sol = sdf_solid(P, x, y, zg);
        } else {
          size_t gi = ((size_t)zh * ny + y) * nx + x;
          q = decode(P, in[0][gi], in[1][gi], in[2][gi], in[3][gi], in[4][gi], in[5][gi]);
          sol = solid[gi];
        }
        if (sol) apply_wall(P, &q);
        pr[pi] = q;
        ps[pi] = (uint8_t)sol;
      }
    }
  }

  float maxs = 0.f;
  for (int zl = zl_lo; zl < zl_hi; zl++) {
    int zz = zl - zl_lo + HALO;
    for (int y = 0; y < ny; y++) {
      for (int x = 0; x < nx; x++) {
        size_t gi = ((size_t)(zl + HALO) * ny + y) * nx + x;
        if (solid[gi]) { /* :1063-1072 copy-through */
          for (int f = 0; f < 6; f++) out[f][gi] = in[f][gi];
          continue;
        }
        size_t pc = ((size_t)zz * py + (y + HALO)) * px + (x + HALO);
        prim_t line[7];
        uint8_t sl[7];
        cons_t Fm[3], Fp[3];
        const ptrdiff_t stride[3] = {1, px, (ptrdiff_t)px * py};
        for (int ax = 0; ax < 3; ax++) {
          for (int k = -3; k <= 3; k++) {
            line[k + 3] = pr[pc + k * stride[ax]];
            sl[k + 3] = ps[pc + k * stride[ax]];
          }
          Fm[ax] = face_flux(P, line, sl, ax, 0);
          Fp[ax] = face_flux(P, line, sl, ax, 1);
        }
        prim_t q0 = pr[pc];
        cons_t U0 = prim_to_cons(P, &q0);
        cons_t dU; /* :1268-1280 */
        dU.r = -((Fp[0].r - Fm[0].r) / P->dx + (Fp[1].r - Fm[1].r) / P->dy + (Fp[2].r - Fm[2].r) / P->dz);
        dU.mx = -((Fp[0].mx - Fm[0].mx) / P->dx + (Fp[1].mx - Fm[1].mx) / P->dy + (Fp[2].mx - Fm[2].mx) / P->dz);
        dU.my = -((Fp[0].my - Fm[0].my) / P->dx + (Fp[1].my - Fm[1].my) / P->dy + (Fp[2].my - Fm[2].my) / P->dz);
        dU.mz = -((Fp[0].mz - Fm[0].mz) / P->dx + (Fp[1].mz - Fm[1].mz) / P->dy + (Fp[2].mz - Fm[2].mz) / P->dz);
        dU.Et = -((Fp[0].Et - Fm[0].Et) / P->dx + (Fp[1].Et - Fm[1].Et) / P->dy + (Fp[2].Et - Fm[2].Et) / P->dz);
        dU.Ev = -((Fp[0].Ev - Fm[0].Ev) / P->dx + (Fp[1].Ev - Fm[1].Ev) / P->dy + (Fp[2].Ev - Fm[2].Ev) / P->dz);
        cons_t ddt = C_MUL(dU, dt);
        cons_t U1 = C_ADD(U0, ddt);

        /* cons_to_prim :247-262 */
        prim_t q1;
        q1.r = fmaxf(U1.r, RHO_P_FLOOR);
        q1.u = U1.mx / q1.r;
        q1.v = U1.my / q1.r;
        q1.w = U1.mz / q1.r;
        float ke = 0.5f * (q1.u * q1.u + q1.v * q1.v + q1.w * q1.w);
        float ev = fmaxf(U1.Ev / q1.r, 0.f);
        float e_tot = U1.Et / q1.r;
        float e_th = fmaxf(e_tot - ke - ev, THERMAL_ENERGY_FLOOR);
        q1.p = fmaxf((P->gamma_floor - 1.f) * q1.r * e_th, RHO_P_FLOOR);
        q1.ev = ev;
        float T1 = q1.p / (q1.r * P->R);

        if (!isfinite(q1.r) || !isfinite(q1.p) || !isfinite(q1.u) || !isfinite(q1.v) ||
            !isfinite(q1.w) || !isfinite(q1.ev) || q1.r <= 0.f || q1.p <= 0.f || q1.ev < 0.f) {
          q1 = inflow_prim(P); /* :1284-1289 */
          T1 = q1.p / (q1.r * P->R);
        }
        /* Landau-Teller relaxation :1290-1292 */
        float eeq = evib_eq(P, T1);
        q1.ev = fmaxf(q1.ev + (eeq - q1.ev) * (dt / fmaxf(P->tau_vib, TAU_VIB_MIN)), 0.f);

        int nsp = (P->sponge_n > 0) ? P->sponge_n : 0; /* inflow sponge :1295-1319 */
        if (nsp > 0 && x < nsp) {
          float s = 1.0f - (float)x / (float)nsp;
          s = fminf(fmaxf(s, 0.0f), 1.0f);
          float k = P->sponge_strength * (s * s);
          float tr = fmaxf(P->inflow_r, RHO_P_FLOOR), tp = fmaxf(P->inflow_p, RHO_P_FLOOR);
          float tu = inflow_gain * P->inflow_u, tv = inflow_gain * P->inflow_v, tw = inflow_gain * P->inflow_w;
          float tT = tp / (tr * P->R);
          float tev = evib_eq(P, tT);
          q1.r = fmaxf(q1.r + k * (tr - q1.r), RHO_P_FLOOR);
          q1.p = fmaxf(q1.p + k * (tp - q1.p), RHO_P_FLOOR);
          q1.u = q1.u + k * (tu - q1.u);
          q1.v = q1.v + k * (tv - q1.v);
          q1.w = q1.w + k * (tw - q1.w);
          q1.ev = fmaxf(q1.ev + k * (tev - q1.ev), 0.f);
        }
        int nspo = (P->sponge_out_n > 0) ? P->sponge_out_n : 0; /* outflow sponge :1320-1344 */
        if (nspo > 0 && x >= (nx - nspo)) {
          int xo = x - (nx - nspo);
          float s = (float)xo / (float)nspo;
          s = fminf(fmaxf(s, 0.0f), 1.0f);
          float k = P->sponge_out_strength * (s * s);
          float tr = fmaxf(P->inflow_r, RHO_P_FLOOR), tp = fmaxf(P->inflow_p, RHO_P_FLOOR);
          float tT = tp / (tr * P->R);
          float tev = evib_eq(P, tT);
          q1.r = fmaxf(q1.r + k * (tr - q1.r), RHO_P_FLOOR);
          q1.p = fmaxf(q1.p + k * (tp - q1.p), RHO_P_FLOOR);
          q1.u = q1.u + k * (0.0f - q1.u);
          q1.v = q1.v + k * (0.0f - q1.v);
          q1.w = q1.w + k * (0.0f - q1.w);
          q1.ev = fmaxf(q1.ev + k * (tev - q1.ev), 0.f);
        }
        float a = soundspeed(P, &q1); /* :1345-1351 */
        float ssx = (fabsf(q1.u) + a) / P->dx;
        float ssy = (fabsf(q1.v) + a) / P->dy;
        float ssz = (fabsf(q1.w) + a) / P->dz;
        float ssum = ssx + ssy + ssz;
        if (isfinite(ssum) && ssum > 0.f && ssum > maxs) maxs = ssum;

        out[0][gi] = logf(fmaxf(q1.r, RHO_P_FLOOR)); /* :1353-1358 */
        out[1][gi] = asinh_enc(q1.u / P->u_ref);
        out[2][gi] = asinh_enc(q1.v / P->u_ref);
        out[3][gi] = asinh_enc(q1.w / P->u_ref);
        out[4][gi] = logf(fmaxf(q1.p, RHO_P_FLOOR));
        out[5][gi] = logf(fmaxf(q1.ev, RHO_P_FLOOR));
      }
    }
  }
  free(pr);
  free(ps);
  return maxs;
}

/* Host log-time controller, :1680-1683 (before the step) */
void o3_clock_begin(tau3d_clock *c) {
  c->t *= expf(c->d_tau);
  c->dt = c->t * c->d_tau;
  float ramp = c->t / 0.02f;
  c->gain = fminf(fmaxf(ramp, 0.f), 1.f);
}

/* :1697-1704 (after the step, with the step's global max wavespeed) */
void o3_clock_end(tau3d_clock *c, float cfl, float maxs) {
  float dt_cfl = cfl / fmaxf(maxs, 1e-9f);
  if (c->dt > 1.10f * dt_cfl) c->d_tau *= 0.80f;
  else if (c->dt < 0.85f * dt_cfl) c->d_tau *= 1.10f;
  c->d_tau = fminf(fmaxf(c->d_tau, 1e-7f), 5e-2f);
  c->maxs = maxs;
  c->step += 1;
}

void o3_clock_reset(tau3d_clock *c) { /* :1635-1636 */
  c->t = 1e-5f; c->d_tau = 1e-3f; c->dt = 0.f; c->gain = 0.f; c->maxs = 0.f; c->step = 0;
}

void o3_params_default(tau3d_params *hp, int nx, int ny, int nz) { /* :1531-1557 */
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

/* Single-domain convenience: run `nsteps` full steps (controller + k_step + swap)
 * on halo-layout arrays a[6] / b[6]; result ends in a[] if nsteps is even else b[].
 * Returns 0. */
int o3_run(const tau3d_params *P, float *const a[6], float *const b[6], const uint8_t *solid,
           tau3d_clock *clk, int nsteps) {
  float *cur[6], *nxt[6];
  for (int f = 0; f < 6; f++) { cur[f] = a[f]; nxt[f] = b[f]; }
  for (int s = 0; s < nsteps; s++) {
    o3_fill_halo_periodic(P, P->nz, cur);
    o3_clock_begin(clk);
    float m = o3_step(P, 0, P->nz, 0, P->nz, (const float *const *)cur, nxt, solid, clk->dt, clk->gain);
    o3_clock_end(clk, P->cfl, m);
    for (int f = 0; f < 6; f++) { float *t = cur[f]; cur[f] = nxt[f]; nxt[f] = t; }
  }
  return 0;
}

/* ---------------------------------------------------------------- visualisation */

/* prim_at_xbc, :724-751, on the slab layout (zh = plane index incl. halo, zg = wrapped global plane) */
static prim_t prim_at_xbc(const tau3d_params *P, const float *const st[6], const uint8_t *solid, int x, int y,
                          int zh, int zg) {
  const int nx = P->nx, ny = P->ny;
  y = wrapi(y, ny);
  prim_t q;
  if (x < 0) {
    q = inflow_prim(P);
    if (sdf_solid(P, x, y, zg)) apply_wall(P, &q);
    return q;
  }
  if (x >= nx) { /* outflow_prim_transmissive reads cell nx-1 of the same (y, z), :696-698 */
    size_t gi = ((size_t)zh * ny + y) * nx + (nx - 1);
    prim_t qR = decode(P, st[0][gi], st[1][gi], st[2][gi], st[3][gi], st[4][gi], st[5][gi]);
    return outflow_prim(P, qR);
  }
  size_t gi = ((size_t)zh * ny + y) * nx + x;
  q = decode(P, st[0][gi], st[1][gi], st[2][gi], st[3][gi], st[4][gi], st[5][gi]);
  if (solid[gi]) apply_wall(P, &q);
  return q;
}

static inline float safe_log1pf(float x) { return logf(1.0f + fmaxf(x, 0.0f)); } /* :796-798 */

/* k_vis, :800-905 — one scalar per cell of the slab's interior planes; out has nx*ny*nzl floats
 * (no halo).  The z neighbours come from the halo planes, which must be current. */
/* scale (may be NULL): per cell, the magnitude of the operands the value was formed from — |value| for the
 * point modes (1 + |value| for the two log(1 + x) modes: the operand 1 + x is rounded at eps(1), which is
 * an ABSOLUTE 6e-8 on a value that may itself be 1e-3), sum over the differenced pairs of (|q+| + |q-|) / (2 d) for the gradient modes (squared
 * for Q).  A second fp32 evaluation agrees with this one to a few eps of THAT, not of a difference that
 * may cancel to nothing in smooth flow; tests state their 1e-5 against it. */
void o3_vis(const tau3d_params *P, int z0, int nzl, const float *const st[6], const uint8_t *solid, int mode,
            float *out, float *scale) {
  const int nx = P->nx, ny = P->ny;
  for (int zl = 0; zl < nzl; zl++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) {
        const int zh = zl + HALO;
        size_t oi = ((size_t)zl * ny + y) * nx + x;
        size_t gi = ((size_t)zh * ny + y) * nx + x;
        if (scale) scale[oi] = 0.f;
        if (solid[gi]) { out[oi] = 0.f; continue; }
#define QAT(dx_, dy_, dz_) prim_at_xbc(P, st, solid, x + (dx_), y + (dy_), zh + (dz_), wrapi(z0 + zl + (dz_), P->nz))
        prim_t q0 = QAT(0, 0, 0);
        if (mode == 1) { out[oi] = safe_log1pf(q0.r); if (scale) scale[oi] = 1.f + fabsf(out[oi]); continue; }
        if (mode == 2) { out[oi] = safe_log1pf(q0.p); if (scale) scale[oi] = 1.f + fabsf(out[oi]); continue; }
        if (mode == 3) { out[oi] = sqrtf(q0.u * q0.u + q0.v * q0.v + q0.w * q0.w); if (scale) scale[oi] = out[oi]; continue; }
        if (mode == 4) {
          float a = soundspeed(P, &q0);
          float s = sqrtf(q0.u * q0.u + q0.v * q0.v + q0.w * q0.w);
          out[oi] = s / fmaxf(a, DENOM_EPS);
          if (scale) scale[oi] = out[oi];
          continue;
        }
        prim_t qxm = QAT(-1, 0, 0), qxp = QAT(1, 0, 0), qym = QAT(0, -1, 0), qyp = QAT(0, 1, 0);
        prim_t qzm = QAT(0, 0, -1), qzp = QAT(0, 0, 1);
#undef QAT
        float inv2dx = 0.5f / P->dx, inv2dy = 0.5f / P->dy, inv2dz = 0.5f / P->dz;
        float dudx = (qxp.u - qxm.u) * inv2dx, dudy = (qyp.u - qym.u) * inv2dy, dudz = (qzp.u - qzm.u) * inv2dz;
        float dvdx = (qxp.v - qxm.v) * inv2dx, dvdy = (qyp.v - qym.v) * inv2dy, dvdz = (qzp.v - qzm.v) * inv2dz;
        float dwdx = (qxp.w - qxm.w) * inv2dx, dwdy = (qyp.w - qym.w) * inv2dy, dwdz = (qzp.w - qzm.w) * inv2dz;
        if (scale) {
#define MAG(f_) ((fabsf(qxp.f_) + fabsf(qxm.f_)) * inv2dx + (fabsf(qyp.f_) + fabsf(qym.f_)) * inv2dy + \
                 (fabsf(qzp.f_) + fabsf(qzm.f_)) * inv2dz)
          float sv = MAG(u) + MAG(v) + MAG(w);
          scale[oi] = (mode == 0) ? MAG(r) : (mode == 7) ? sv * sv : sv;
#undef MAG
        }
        if (mode == 6) { out[oi] = dudx + dvdy + dwdz; continue; }
        float wx = dwdy - dvdz, wy = dudz - dwdx, wz = dvdx - dudy;
        if (mode == 5) { out[oi] = sqrtf(wx * wx + wy * wy + wz * wz); continue; }
        if (mode == 7) {
          float O12 = 0.5f * (dudy - dvdx), O13 = 0.5f * (dudz - dwdx), O23 = 0.5f * (dvdz - dwdy);
          float Om2 = 2.0f * (O12 * O12 + O13 * O13 + O23 * O23);
          float S12 = 0.5f * (dudy + dvdx), S13 = 0.5f * (dudz + dwdx), S23 = 0.5f * (dvdz + dwdy);
          float Sm2 = (dudx * dudx + dvdy * dvdy + dwdz * dwdz) + 2.0f * (S12 * S12 + S13 * S13 + S23 * S23);
          out[oi] = 0.5f * (Om2 - Sm2);
          continue;
        }
        float drdx = (qxp.r - qxm.r) * inv2dx, drdy = (qyp.r - qym.r) * inv2dy, drdz = (qzp.r - qzm.r) * inv2dz;
        out[oi] = sqrtf(drdx * drdx + drdy * drdy + drdz * drdz);
      }
}

/* slice_to_rgba, :1416-1442 — grey ramp with t^2 opacity, per-slice min/max normalisation */
void o3_slice_to_rgba(uint32_t *dst, const float *vol, int nx, int ny, int nz, int zslice, int log_scale,
                      float a_gain, float *mn_out, float *mx_out) {
  zslice = (zslice < 0) ? 0 : (zslice >= nz ? (nz - 1) : zslice);
  const float *s = vol + (size_t)zslice * (size_t)nx * (size_t)ny;
  float mn = 1e30f, mx = -1e30f;
  for (int i = 0; i < nx * ny; i++) {
    float v = s[i];
    v = log_scale ? safe_log1pf(v) : v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  float inv = 1.0f / fmaxf(mx - mn, 1e-20f);
  for (int i = 0; i < nx * ny; i++) {
    float v = s[i];
    v = log_scale ? safe_log1pf(v) : v;
    float t = clampf((v - mn) * inv, 0.f, 1.f);
    float a = clampf(a_gain * (t * t), 0.f, 1.f);
    unsigned char c = (unsigned char)(t * 255.0f);
    unsigned char A = (unsigned char)(a * 255.0f);
    dst[i] = ((uint32_t)A << 24) | ((uint32_t)c << 16) | ((uint32_t)c << 8) | (uint32_t)c;
  }
  if (mn_out) *mn_out = mn;
  if (mx_out) *mx_out = mx;
}

/* th3cs.cu:1199-1222 — the export map of the headless program: whole-volume min / max, then per voxel
 * pIdx = clamp((int)(powf((v - min) / max(max - min, 1e-12), gamma) * 255), 0, 255) (th3cs: gamma = 0.65).
 * The volume is th3cs's k_schlieren_export field (:641-673), which is o3_vis mode 0 (same prim_at_xbc, same
 * central differences of rho). */
void o3_palette_indices(const float *vol, size_t n, float gamma, uint8_t *out, float *mn_out, float *mx_out) {
  float min_val = 1e30f, max_val = -1e30f;
  for (size_t i = 0; i < n; i++) {
    min_val = fminf(min_val, vol[i]);
    max_val = fmaxf(max_val, vol[i]);
  }
  float range = fmaxf(max_val - min_val, 1e-12f);
  for (size_t i = 0; i < n; i++) {
    float norm = (vol[i] - min_val) / range;
    norm = powf(norm, gamma);
    int pIdx = (int)(norm * 255.0f);
    pIdx = pIdx < 0 ? 0 : (pIdx > 255 ? 255 : pIdx);
    out[i] = (uint8_t)pIdx;
  }
  if (mn_out) *mn_out = min_val;
  if (mx_out) *mx_out = max_val;
}

/* k_outflow_reflection_metric, :1389-1408 — max |p - p_inflow| over the last nprobe x-columns */
float o3_outflow_reflection(const tau3d_params *P, int nzl, const float *const st[6], int nprobe) {
  const int nx = P->nx, ny = P->ny;
  int x0 = nx - ((nprobe > 1) ? nprobe : 1);
  float p_ref = fmaxf(P->inflow_p, RHO_P_FLOOR), m = 0.f;
  for (int zl = 0; zl < nzl; zl++)
    for (int y = 0; y < ny; y++)
      for (int x = (x0 < 0 ? 0 : x0); x < nx; x++) {
        size_t gi = ((size_t)(zl + HALO) * ny + y) * nx + x;
        float d = fabsf(expf(st[4][gi]) - p_ref);
        if (d > m) m = d;
      }
  return m;
}
