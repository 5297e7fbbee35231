/* tau2d_oracle.c — TEST INFRASTRUCTURE ONLY (CPU oracle, not a product path).
 *
 * Plain-C fp64 restatement of the GPU 2D Euler scheme of the reference
 * (tau_hypersonic_cuda.cu): SoA state, per step
 *   k_apply_inflow_left (:772-784) -> max wavespeed (:786-847) -> host dt (:1852-1869) ->
 *   k_predict_face_states (:849-962) -> k_compute_x/yface_flux (:964-1030) -> k_step (:1032-1176)
 * with the grid size a run-time parameter (compile-time 8192 x 1024 in the reference, :28-29).
 * IEEE fp64, no FMA contraction (-ffp-contract=off), reference association order.
 *
 * Parity pin: SURVEY.md §8(c) — reference output at 512 x 256, 4 steps, default_config:
 *   t = 0.03654792676725048, fluid = 128770, sum rho = 128783.43433989958,
 *   sum mx = 3373957.4678593008, sum E = 45559661.995020151   (tests/test_oracle_pins.py)
 * plus the known answers of tau_hypersonic_cuda_tests.cu:245-346 exposed through o2h_unit_*.
 */
#include "../include/tau_params.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EPS_RHO 1e-25 /* :32 */
#define EPS_P 1e-25   /* :33 */

typedef struct { double rho, mx, my, E; } cons_t;
typedef struct { double rho, u, v, p; } prim_t;
typedef struct { prim_t L, R; } faceprim_t;

typedef struct {
  int W, H;
  tauh2_params c;
  const double *rho, *mx, *my, *E; /* current state (SoA) */
  const uint8_t *mask;
} ctx_t;

static inline double d_fmax(double a, double b) { return a > b ? a : b; }
static inline double d_fmin(double a, double b) { return a < b ? a : b; }
static inline double d_fabs(double a) { return a < 0 ? -a : a; }

static inline prim_t cons_to_prim(const ctx_t *X, cons_t c) { /* :143-158 */
  prim_t p;
  double rho = d_fmax(c.rho, EPS_RHO);
  double inv = 1.0 / rho;
  double u = c.mx * inv, v = c.my * inv;
  double kin = 0.5 * rho * (u * u + v * v);
  double eint = c.E - kin;
  p.rho = rho; p.u = u; p.v = v;
  p.p = (X->c.gamma - 1.0) * d_fmax(eint, EPS_P);
  return p;
}
static inline cons_t prim_to_cons(const ctx_t *X, prim_t p) { /* :160-169 */
  cons_t c;
  double rho = d_fmax(p.rho, EPS_RHO), pr = d_fmax(p.p, EPS_P);
  c.rho = rho; c.mx = rho * p.u; c.my = rho * p.v;
  c.E = pr / (X->c.gamma - 1.0) + 0.5 * rho * (p.u * p.u + p.v * p.v);
  return c;
}
static inline double sound_speed(const ctx_t *X, prim_t p) { /* :171-173 */
  return sqrt(X->c.gamma * d_fmax(p.p, EPS_P) / d_fmax(p.rho, EPS_RHO));
}
static inline cons_t flux_axis(const ctx_t *X, cons_t c, int ax) { /* :193-202 */
  prim_t p = cons_to_prim(X, c);
  cons_t f;
  double un = ax ? p.v : p.u;
  f.rho = ax ? c.my : c.mx;
  f.mx = ax ? (c.mx * un) : (c.mx * un + p.p);
  f.my = ax ? (c.my * un + p.p) : (c.my * un);
  f.E = (c.E + p.p) * un;
  return f;
}
static inline double minmod(double a, double b) { /* :216-220 */
  if (a * b <= 0.0) return 0.0;
  return (d_fabs(a) < d_fabs(b)) ? a : b;
}
static inline double mc_limiter(double dl, double dc, double dr) { /* :222-227 */
  double mm1 = minmod(dl, dr), mm2 = minmod(dc, 2.0 * dl), mm3 = minmod(dc, 2.0 * dr);
  return minmod(mm1, minmod(mm2, mm3));
}
static inline prim_t inflow_state(const ctx_t *X) { /* :230-238 */
  const double rho = 1.0, p = 1.0;
  double a = sqrt(X->c.gamma * p / rho);
  prim_t s = {rho, X->c.mach * a, 0, p};
  return s;
}
static inline cons_t load_cons(const ctx_t *X, int i) {
  cons_t c = {X->rho[i], X->mx[i], X->my[i], X->E[i]};
  return c;
}
static inline prim_t wall_ghost_prim(prim_t in) { /* :262-264: no-slip */
  prim_t g = {in.rho, -in.u, -in.v, in.p};
  return g;
}
/* neighbor_or_wall :266-290 / load_neighbor_or_wall_tiled :349-371 (same semantics) */
static inline cons_t neighbor(const ctx_t *X, prim_t center, int xn, int yn) {
  if (yn < 0) yn = 0;
  if (yn >= X->H) yn = X->H - 1;
  if (xn < 0) return prim_to_cons(X, inflow_state(X));
  if (xn >= X->W) return load_cons(X, yn * X->W + (X->W - 1));
  int j = yn * X->W + xn;
  if (X->mask[j]) return prim_to_cons(X, wall_ghost_prim(center));
  return load_cons(X, j);
}

static inline void enforce_positive_faces(prim_t *qm, prim_t qc, prim_t *qp) { /* :373-398 */
  for (int it = 0; it < 8; it++) {
    int bad = 0;
    if (qm->rho <= EPS_RHO || qp->rho <= EPS_RHO) bad = 1;
    if (qm->p <= EPS_P || qp->p <= EPS_P) bad = 1;
    if (!bad) return;
    qm->rho = 0.5 * (qm->rho + qc.rho); qm->u = 0.5 * (qm->u + qc.u);
    qm->v = 0.5 * (qm->v + qc.v);       qm->p = 0.5 * (qm->p + qc.p);
    qp->rho = 0.5 * (qp->rho + qc.rho); qp->u = 0.5 * (qp->u + qc.u);
    qp->v = 0.5 * (qp->v + qc.v);       qp->p = 0.5 * (qp->p + qc.p);
  }
  qm->rho = d_fmax(qm->rho, EPS_RHO); qp->rho = d_fmax(qp->rho, EPS_RHO);
  qm->p = d_fmax(qm->p, EPS_P);       qp->p = d_fmax(qp->p, EPS_P);
}
static inline faceprim_t reconstruct_limited_faces(prim_t qm, prim_t qc, prim_t qp) { /* :400-425 */
#define SLOPE(f) mc_limiter(qc.f - qm.f, 0.5 * (qp.f - qm.f), qp.f - qc.f)
  double s_rho = SLOPE(rho), s_u = SLOPE(u), s_v = SLOPE(v), s_p = SLOPE(p);
#undef SLOPE
  faceprim_t fp;
  fp.L = (prim_t){qc.rho - 0.5 * s_rho, qc.u - 0.5 * s_u, qc.v - 0.5 * s_v, qc.p - 0.5 * s_p};
  fp.R = (prim_t){qc.rho + 0.5 * s_rho, qc.u + 0.5 * s_u, qc.v + 0.5 * s_v, qc.p + 0.5 * s_p};
  enforce_positive_faces(&fp.L, qc, &fp.R);
  return fp;
}
static inline prim_t half_step_predict(const ctx_t *X, prim_t q, cons_t dF, double h) { /* :442-455 */
  cons_t c = prim_to_cons(X, q);
  c.rho -= h * dF.rho; c.mx -= h * dF.mx; c.my -= h * dF.my; c.E -= h * dF.E;
  prim_t out = cons_to_prim(X, c);
  out.rho = d_fmax(out.rho, EPS_RHO);
  out.p = d_fmax(out.p, EPS_P);
  return out;
}

#define C_SUB(a, b) ((cons_t){(a).rho - (b).rho, (a).mx - (b).mx, (a).my - (b).my, (a).E - (b).E})
#define C_ADD(a, b) ((cons_t){(a).rho + (b).rho, (a).mx + (b).mx, (a).my + (b).my, (a).E + (b).E})
#define C_MUL(s, a) ((cons_t){(s) * (a).rho, (s) * (a).mx, (s) * (a).my, (s) * (a).E})

static cons_t hlle_axis(const ctx_t *X, cons_t UL, cons_t UR, int ax) { /* :483-509 */
  prim_t L = cons_to_prim(X, UL), R = cons_to_prim(X, UR);
  double uL = ax ? L.v : L.u, uR = ax ? R.v : R.u;
  double aL = sound_speed(X, L), aR = sound_speed(X, R);
  double SL = d_fmin(uL - aL, uR - aR), SR = d_fmax(uL + aL, uR + aR);
  cons_t FL = flux_axis(X, UL, ax), FR = flux_axis(X, UR, ax);
  if (SL >= 0.0) return FL;
  if (SR <= 0.0) return FR;
  double denom = SR - SL;
  if (d_fabs(denom) < 1e-14) { cons_t s = C_ADD(FL, FR); return C_MUL(0.5, s); }
  cons_t t1 = C_MUL(SR, FL), t2 = C_MUL(-SL, FR), dU = C_SUB(UR, UL);
  cons_t t3 = C_MUL(SL * SR, dU);
  cons_t s12 = C_ADD(t1, t2), s = C_ADD(s12, t3);
  return C_MUL(1.0 / denom, s);
}

static cons_t hllc_axis(const ctx_t *X, cons_t UL, cons_t UR, int ax) { /* :519-606 */
  prim_t L = cons_to_prim(X, UL), R = cons_to_prim(X, UR);
  double unL = ax ? L.v : L.u, unR = ax ? R.v : R.u;
  double utL = ax ? L.u : L.v, utR = ax ? R.u : R.v;
  double aL = sound_speed(X, L), aR = sound_speed(X, R);
  double SL = d_fmin(unL - aL, unR - aR), SR = d_fmax(unL + aL, unR + aR);
  cons_t FL = flux_axis(X, UL, ax), FR = flux_axis(X, UR, ax);
  if (SL >= 0.0) return FL;
  if (SR <= 0.0) return FR;
  double rhoL = L.rho, rhoR = R.rho, pL = L.p, pR = R.p;
  double num = pR - pL + rhoL * unL * (SL - unL) - rhoR * unR * (SR - unR);
  double den = rhoL * (SL - unL) - rhoR * (SR - unR);
  if (d_fabs(den) < 1e-14 || !isfinite(num) || !isfinite(den)) return hlle_axis(X, UL, UR, ax);
  double SM = num / den;
  if (!isfinite(SM)) return hlle_axis(X, UL, UR, ax);
  double pStar = pL + rhoL * (SL - unL) * (SM - unL);
  pStar = d_fmax(pStar, EPS_P);
  double dLS = SL - SM, dRS = SR - SM;
  if (d_fabs(dLS) < 1e-14 || d_fabs(dRS) < 1e-14) return hlle_axis(X, UL, UR, ax);
  double rsL = rhoL * (SL - unL) / dLS, rsR = rhoR * (SR - unR) / dRS;
  if (!(rsL > 0.0) || !(rsR > 0.0) || !isfinite(rsL) || !isfinite(rsR)) return hlle_axis(X, UL, UR, ax);
  double EsL = ((SL - unL) * UL.E - pL * unL + pStar * SM) / dLS;
  if (!isfinite(EsL)) return hlle_axis(X, UL, UR, ax);
  double EsR = ((SR - unR) * UR.E - pR * unR + pStar * SM) / dRS;
  if (!isfinite(EsR)) return hlle_axis(X, UL, UR, ax);
  cons_t F;
  if (SM >= 0.0) {
    double mN = rsL * SM, mT = rsL * utL;
    cons_t US = ax ? (cons_t){rsL, mT, mN, EsL} : (cons_t){rsL, mN, mT, EsL};
    F.rho = FL.rho + SL * (US.rho - UL.rho); F.mx = FL.mx + SL * (US.mx - UL.mx);
    F.my = FL.my + SL * (US.my - UL.my);     F.E = FL.E + SL * (US.E - UL.E);
  } else {
    double mN = rsR * SM, mT = rsR * utR;
    cons_t US = ax ? (cons_t){rsR, mT, mN, EsR} : (cons_t){rsR, mN, mT, EsR};
    F.rho = FR.rho + SR * (US.rho - UR.rho); F.mx = FR.mx + SR * (US.mx - UR.mx);
    F.my = FR.my + SR * (US.my - UR.my);     F.E = FR.E + SR * (US.E - UR.E);
  }
  return F;
}

/* ---- body geometry, :625-686, 729-770 */
static inline double clamp01(double t) { return t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }
static inline double len2(double x, double y) { return sqrt(x * x + y * y); }
static inline double sdSegment(double px, double py, double ax, double ay, double bx, double by) {
  double abx = bx - ax, aby = by - ay, apx = px - ax, apy = py - ay;
  double denom = abx * abx + aby * aby + 1e-30;
  double t = clamp01((apx * abx + apy * aby) / denom);
  double qx = ax + t * abx, qy = ay + t * aby;
  return len2(px - qx, py - qy);
}
static double sdSphereConeCapsule(double x, double y, double Rb, double Rn, double theta) {
  double r = d_fabs(y);
  double st = sin(theta), ct = cos(theta), tt = tan(theta);
  double xt = Rn * (1.0 - st), rt = Rn * ct;
  double xb = xt + (Rb - rt) / d_fmax(tt, 1e-30);
  double rprof = 0.0;
  if (x < 0.0) rprof = -1.0;
  else if (x <= xt) { double dx = x - Rn; double in = Rn * Rn - dx * dx; rprof = (in > 0.0) ? sqrt(in) : 0.0; }
  else if (x <= xb) rprof = rt + (x - xt) * tt;
  else rprof = -1.0;
  int inside = (x >= 0.0 && x <= xb && r <= rprof);
  double d_sphere = d_fabs(len2(x - Rn, r) - Rn);
  double d_cone = sdSegment(x, r, xt, rt, xb, Rb);
  double d_base = sdSegment(x, y, xb, -Rb, xb, +Rb);
  double d_rim = len2(x - xb, r - Rb);
  double d = d_sphere;
  if (d_cone < d) d = d_cone;
  if (d_base < d) d = d_base;
  if (d_rim < d) d = d_rim;
  return inside ? -d : d;
}

/* ------------------------------------------------------------------ public */
void o2h_params_default(tauh2_params *c, int W, int H) { /* default_config, :1394-1409 */
  c->W = W; c->H = H;
  c->gamma = 1.1; c->cfl = 0.25; c->visc_nu = 5e-2; c->visc_rho = 5e-2; c->visc_e = 2e-2;
  c->mach = 25.0; c->geom_x0 = 125.0; c->geom_cy = (double)H / 2.0;
  c->geom_rb = (double)H / 12.0; c->geom_rn = (double)H / 24.0; c->geom_theta = 3.14159265358979323846 / 4.0;
}

/* k_init, :740-770 */
void o2h_init(const tauh2_params *c, double *rho, double *mx, double *my, double *E, uint8_t *mask) {
  ctx_t X = {c->W, c->H, *c, rho, mx, my, E, mask};
  double Rb = c->geom_rb, Rn = c->geom_rn, theta = c->geom_theta;
  double st = sin(theta), ct = cos(theta), tt = tan(theta);
  double xb = Rn * (1.0 - st) + (Rb - Rn * ct) / d_fmax(tt, 1e-30); /* spherecone_xb, :729-737 */
  prim_t inflow = inflow_state(&X);
  for (int y = 0; y < c->H; y++)
    for (int x = 0; x < c->W; x++) {
      int i = y * c->W + x;
      double Xc = (double)x - c->geom_x0, Yc = (double)y - c->geom_cy;
      double sd = sdSphereConeCapsule(Xc, Yc, Rb, Rn, theta) - Rb; /* rounded by k_round = Rb */
      sd = d_fmax(sd, Xc - xb);
      uint8_t m = (sd < 0.0) ? 1 : 0;
      mask[i] = m;
      prim_t s = m ? (prim_t){inflow.rho, 0.0, 0.0, inflow.p} : inflow;
      cons_t cc = prim_to_cons(&X, s);
      rho[i] = cc.rho; mx[i] = cc.mx; my[i] = cc.my; E[i] = cc.E;
    }
}

/* k_apply_inflow_left, :772-784 (in place) */
void o2h_apply_inflow(const tauh2_params *c, double *rho, double *mx, double *my, double *E, const uint8_t *mask) {
  ctx_t X = {c->W, c->H, *c, rho, mx, my, E, mask};
  cons_t in = prim_to_cons(&X, inflow_state(&X));
  for (int y = 0; y < c->H; y++) {
    int i0 = y * c->W;
    if (mask[i0]) continue;
    rho[i0] = in.rho; mx[i0] = in.mx; my[i0] = in.my; E[i0] = in.E;
  }
}

/* :786-847 + host clamp :1852-1854 */
double o2h_max_wavespeed(const tauh2_params *c, const double *rho, const double *mx, const double *my,
                         const double *E, const uint8_t *mask) {
  ctx_t X = {c->W, c->H, *c, rho, mx, my, E, mask};
  double m = 1e-12;
  for (int i = 0; i < c->W * c->H; i++) {
    if (mask[i]) continue;
    prim_t p = cons_to_prim(&X, load_cons(&X, i));
    double a = sound_speed(&X, p);
    double sx = d_fabs(p.u) + a, sy = d_fabs(p.v) + a;
    double v = (sx > sy) ? sx : sy;
    if (!isfinite(v)) v = 1e-12;
    if (v > m) m = v;
  }
  if (!isfinite(m) || m < 1e-12) m = 1e-12;
  return m;
}

double o2h_dt_from_maxs(const tauh2_params *c, double maxs) { /* :1856-1865 */
  double dt_conv = c->cfl * 1.0 / maxs;
  double nu_max = fmax(c->visc_nu, fmax(c->visc_rho, c->visc_e));
  double dt_diff = dt_conv;
  if (isfinite(nu_max) && nu_max > 1e-12) dt_diff = 0.25 / nu_max;
  return fmin(dt_conv, dt_diff);
}

/* One step with a given dt on a state that already had the inflow column applied:
 * predict -> face fluxes -> update + diffusion.  in[4] -> out[4] (rho, mx, my, E). */
void o2h_step_dt(const tauh2_params *c, const double *const in[4], double *const out[4], const uint8_t *mask,
                 double dt) {
  const int W = c->W, H = c->H, N = W * H;
  ctx_t Xs = {W, H, *c, in[0], in[1], in[2], in[3], mask};
  const ctx_t *X = &Xs;
  cons_t *xL = (cons_t *)malloc(sizeof(cons_t) * N), *xR = (cons_t *)malloc(sizeof(cons_t) * N);
  cons_t *yL = (cons_t *)malloc(sizeof(cons_t) * N), *yR = (cons_t *)malloc(sizeof(cons_t) * N);
  cons_t *xF = (cons_t *)malloc(sizeof(cons_t) * (W + 1) * H), *yF = (cons_t *)malloc(sizeof(cons_t) * W * (H + 1));
  double half = 0.5 * dt;

  for (int y = 0; y < H; y++) /* k_predict_face_states, :849-962 */
    for (int x = 0; x < W; x++) {
      int i = y * W + x;
      cons_t Uc = load_cons(X, i);
      if (mask[i]) { xL[i] = xR[i] = yL[i] = yR[i] = Uc; continue; }
      prim_t qc = cons_to_prim(X, Uc);
      for (int ax = 0; ax < 2; ax++) {
        int dx = ax ? 0 : 1, dy = ax ? 1 : 0;
        prim_t qm = cons_to_prim(X, neighbor(X, qc, x - dx, y - dy));
        prim_t qp = cons_to_prim(X, neighbor(X, qc, x + dx, y + dy));
        faceprim_t fp = reconstruct_limited_faces(qm, qc, qp);
        cons_t cL = prim_to_cons(X, fp.L), cR = prim_to_cons(X, fp.R);
        cons_t FL = flux_axis(X, cL, ax), FR = flux_axis(X, cR, ax);
        cons_t dF = {FR.rho - FL.rho, FR.mx - FL.mx, FR.my - FL.my, FR.E - FL.E};
        prim_t qL = half_step_predict(X, fp.L, dF, half), qR = half_step_predict(X, fp.R, dF, half);
        qL.rho = d_fmax(qL.rho, EPS_RHO); qL.p = d_fmax(qL.p, EPS_P);
        qR.rho = d_fmax(qR.rho, EPS_RHO); qR.p = d_fmax(qR.p, EPS_P);
        if (ax == 0) { xL[i] = prim_to_cons(X, qL); xR[i] = prim_to_cons(X, qR); }
        else { yL[i] = prim_to_cons(X, qL); yR[i] = prim_to_cons(X, qR); }
      }
    }

  for (int y = 0; y < H; y++) /* k_compute_xface_flux, :964-996 */
    for (int fx = 0; fx <= W; fx++) {
      int xl = fx - 1, xr = fx;
      int hasL = (xl >= 0) && !mask[y * W + xl], hasR = (xr < W) && !mask[y * W + xr];
      cons_t UL, UR, F = {0, 0, 0, 0};
      if (hasL && hasR) { UL = xR[y * W + xl]; UR = xL[y * W + xr]; F = hllc_axis(X, UL, UR, 0); }
      else if (hasR) { UL = neighbor(X, cons_to_prim(X, load_cons(X, y * W + xr)), xr - 1, y); UR = xL[y * W + xr]; F = hllc_axis(X, UL, UR, 0); }
      else if (hasL) { UL = xR[y * W + xl]; UR = neighbor(X, cons_to_prim(X, load_cons(X, y * W + xl)), xl + 1, y); F = hllc_axis(X, UL, UR, 0); }
      xF[y * (W + 1) + fx] = F;
    }
  for (int fy = 0; fy <= H; fy++) /* k_compute_yface_flux, :998-1030 */
    for (int x = 0; x < W; x++) {
      int yb = fy - 1, yt = fy;
      int hasB = (yb >= 0) && !mask[yb * W + x], hasT = (yt < H) && !mask[yt * W + x];
      cons_t UB, UT, F = {0, 0, 0, 0};
      if (hasB && hasT) { UB = yR[yb * W + x]; UT = yL[yt * W + x]; F = hllc_axis(X, UB, UT, 1); }
      else if (hasT) { UB = neighbor(X, cons_to_prim(X, load_cons(X, yt * W + x)), x, yt - 1); UT = yL[yt * W + x]; F = hllc_axis(X, UB, UT, 1); }
      else if (hasB) { UB = yR[yb * W + x]; UT = neighbor(X, cons_to_prim(X, load_cons(X, yb * W + x)), x, yb + 1); F = hllc_axis(X, UB, UT, 1); }
      yF[fy * W + x] = F;
    }

  const double inv12 = 1.0 / 12.0;
  for (int y = 0; y < H; y++) /* k_step, :1032-1176 */
    for (int x = 0; x < W; x++) {
      int i = y * W + x;
      cons_t Uc = load_cons(X, i);
      if (mask[i]) { out[0][i] = Uc.rho; out[1][i] = Uc.mx; out[2][i] = Uc.my; out[3][i] = Uc.E; continue; }
      cons_t FxL = xF[y * (W + 1) + x], FxR = xF[y * (W + 1) + x + 1], GyB = yF[y * W + x], GyT = yF[(y + 1) * W + x];
      prim_t cp = cons_to_prim(X, Uc);
      cons_t Un = Uc;
      Un.rho -= dt * (FxR.rho - FxL.rho); Un.mx -= dt * (FxR.mx - FxL.mx);
      Un.my -= dt * (FxR.my - FxL.my);    Un.E -= dt * (FxR.E - FxL.E);
      Un.rho -= dt * (GyT.rho - GyB.rho); Un.mx -= dt * (GyT.mx - GyB.mx);
      Un.my -= dt * (GyT.my - GyB.my);    Un.E -= dt * (GyT.E - GyB.E);
      cons_t xm2 = neighbor(X, cp, x - 2, y), xm1 = neighbor(X, cp, x - 1, y), xp1 = neighbor(X, cp, x + 1, y), xp2 = neighbor(X, cp, x + 2, y);
      cons_t ym2 = neighbor(X, cp, x, y - 2), ym1 = neighbor(X, cp, x, y - 1), yp1 = neighbor(X, cp, x, y + 1), yp2 = neighbor(X, cp, x, y + 2);
#define D2(f, m2, m1, p1, p2) ((-(m2).f + 16.0 * (m1).f - 30.0 * Uc.f + 16.0 * (p1).f - (p2).f) * inv12)
      double lap_rho = D2(rho, xm2, xm1, xp1, xp2) + D2(rho, ym2, ym1, yp1, yp2);
      double lap_mx = D2(mx, xm2, xm1, xp1, xp2) + D2(mx, ym2, ym1, yp1, yp2);
      double lap_my = D2(my, xm2, xm1, xp1, xp2) + D2(my, ym2, ym1, yp1, yp2);
      double lap_E = D2(E, xm2, xm1, xp1, xp2) + D2(E, ym2, ym1, yp1, yp2);
#undef D2
      Un.rho += (c->visc_rho * dt) * lap_rho; Un.mx += (c->visc_nu * dt) * lap_mx;
      Un.my += (c->visc_nu * dt) * lap_my;    Un.E += (c->visc_e * dt) * lap_E;
      Un.rho = d_fmax(Un.rho, EPS_RHO);
      prim_t pp = cons_to_prim(X, Un);
      if (pp.p <= EPS_P || !isfinite(pp.p) || !isfinite(pp.rho) || !isfinite(pp.u) || !isfinite(pp.v)) {
        pp.rho = d_fmax(pp.rho, EPS_RHO);
        pp.p = d_fmax(pp.p, EPS_P);
        Un = prim_to_cons(X, pp);
      }
      out[0][i] = Un.rho; out[1][i] = Un.mx; out[2][i] = Un.my; out[3][i] = Un.E;
    }
  free(xL); free(xR); free(yL); free(yR); free(xF); free(yF);
}

/* n full steps of the reference loop (:1833-1890); a[] holds the state on entry, the result is in
 * a[] if n is even else b[]; returns accumulated time through *t */
void o2h_run(const tauh2_params *c, double *const a[4], double *const b[4], const uint8_t *mask, int n, double *t) {
  double *cur[4], *nxt[4];
  for (int f = 0; f < 4; f++) { cur[f] = a[f]; nxt[f] = b[f]; }
  for (int s = 0; s < n; s++) {
    o2h_apply_inflow(c, cur[0], cur[1], cur[2], cur[3], mask);
    double maxs = o2h_max_wavespeed(c, cur[0], cur[1], cur[2], cur[3], mask);
    double dt = o2h_dt_from_maxs(c, maxs);
    o2h_step_dt(c, (const double *const *)cur, nxt, mask, dt);
    for (int f = 0; f < 4; f++) { double *tmp = cur[f]; cur[f] = nxt[f]; nxt[f] = tmp; }
    *t += dt;
  }
}

/* ---- known-answer hooks for tau_hypersonic_cuda_tests.cu:245-346 */
void o2h_unit_flux(double gamma, const double prim[4], int ax, double outF[4], double *a) {
  tauh2_params c; memset(&c, 0, sizeof(c)); c.gamma = gamma;
  ctx_t X = {1, 1, c, 0, 0, 0, 0, 0};
  prim_t p = {prim[0], prim[1], prim[2], prim[3]};
  cons_t F = flux_axis(&X, prim_to_cons(&X, p), ax);
  outF[0] = F.rho; outF[1] = F.mx; outF[2] = F.my; outF[3] = F.E;
  *a = sound_speed(&X, p);
}
void o2h_unit_roundtrip(double gamma, const double cons[4], double out[4]) {
  tauh2_params c; memset(&c, 0, sizeof(c)); c.gamma = gamma;
  ctx_t X = {1, 1, c, 0, 0, 0, 0, 0};
  cons_t cc = {cons[0], cons[1], cons[2], cons[3]};
  cons_t r = prim_to_cons(&X, cons_to_prim(&X, cc));
  out[0] = r.rho; out[1] = r.mx; out[2] = r.my; out[3] = r.E;
}
void o2h_unit_hllc(double gamma, const double cons[4], int ax, double outF[4], double refF[4]) {
  tauh2_params c; memset(&c, 0, sizeof(c)); c.gamma = gamma;
  ctx_t X = {1, 1, c, 0, 0, 0, 0, 0};
  cons_t cc = {cons[0], cons[1], cons[2], cons[3]};
  cons_t F = hllc_axis(&X, cc, cc, ax), G = flux_axis(&X, cc, ax);
  outF[0] = F.rho; outF[1] = F.mx; outF[2] = F.my; outF[3] = F.E;
  refF[0] = G.rho; refF[1] = G.mx; refF[2] = G.my; refF[3] = G.E;
}
double o2h_unit_minmod(double a, double b) { return minmod(a, b); }
double o2h_unit_mc(double dl, double dc, double dr) { return mc_limiter(dl, dc, dr); }
void o2h_unit_inflow(double gamma, double mach, double out[4]) {
  tauh2_params c; memset(&c, 0, sizeof(c)); c.gamma = gamma; c.mach = mach;
  ctx_t X = {1, 1, c, 0, 0, 0, 0, 0};
  prim_t p = inflow_state(&X);
  out[0] = p.rho; out[1] = p.u; out[2] = p.v; out[3] = p.p;
}

/* k_test_clamps, tau_hypersonic_cuda_tests.cu:255-264 (expectations :395-401) */
void o2h_unit_clamps(double gamma, double out[4], double eps[2]) {
  tauh2_params c; memset(&c, 0, sizeof(c)); c.gamma = gamma;
  ctx_t X = {1, 1, c, 0, 0, 0, 0, 0};
  prim_t badp = {-2.0, 1.5, -0.5, -7.0};
  cons_t cc = prim_to_cons(&X, badp);
  cons_t in = {1.0, 3.0, 4.0, 1e-20};
  prim_t q = cons_to_prim(&X, in);
  out[0] = cc.rho; out[1] = cc.E; out[2] = q.rho; out[3] = q.p;
  eps[0] = EPS_RHO; eps[1] = EPS_P;
}
/* k_test_enforce_positive / _no_change, :316-338 (expectations :455-478) */
void o2h_unit_enforce_positive(int valid, double out[4]) {
  prim_t qc, qm, qp;
  if (!valid) { qc = (prim_t){1.0, 4.0, -2.0, 1.0}; qm = (prim_t){-1.0, 8.0, -4.0, -3.0}; qp = (prim_t){-2.0, -8.0, 4.0, -2.0}; }
  else { qc = (prim_t){1.0, 2.0, -1.0, 1.0}; qm = (prim_t){0.8, 2.2, -0.9, 1.1}; qp = (prim_t){1.2, 1.8, -1.2, 0.9}; }
  enforce_positive_faces(&qm, qc, &qp);
  out[0] = qm.rho; out[1] = qm.p; out[2] = qp.rho; out[3] = qp.p;
}
/* k_test_sdf, :340-346 (expectations :480-484) */
void o2h_unit_sdf(double out[2]) {
  out[0] = sdSphereConeCapsule(1.0, 0.0, 5.0, 2.0, 0.6);
  out[1] = sdSphereConeCapsule(40.0, 0.0, 5.0, 2.0, 0.6);
}
/* k_test_neighbors + k_test_neighbor_for_diff on a caller-built field, :348-371 (expectations :613-631):
 * out = left.rho, left.mx, right.rho, right.mx, up.mx | left.rho, left.mx, wall(x, y+1).mx, top_clamped(x, H+20).rho */
void o2h_unit_neighbors(const tauh2_params *P, const double *rho, const double *mx, const double *my, const double *E,
                        const uint8_t *mask, int x, int y, double out[9]) {
  ctx_t X = {P->W, P->H, *P, rho, mx, my, E, mask};
  prim_t center = cons_to_prim(&X, load_cons(&X, y * P->W + x));
  cons_t left = neighbor(&X, center, x - 1, y), right = neighbor(&X, center, x + 1, y), up = neighbor(&X, center, x, y + 1);
  cons_t top = neighbor(&X, center, x, P->H + 20);
  out[0] = left.rho; out[1] = left.mx; out[2] = right.rho; out[3] = right.mx; out[4] = up.mx;
  out[5] = left.rho; out[6] = left.mx; out[7] = up.mx; out[8] = top.rho;
}

/* ---------------------------------------------------------------- rendering */

/* sample_prim_bc, :706-727 */
static prim_t sample_prim_bc(const ctx_t *X, int xc, int yc, int x, int y) {
  if (y < 0) y = 0;
  if (y >= X->H) y = X->H - 1;
  if (x < 0) return inflow_state(X);
  if (x >= X->W) return cons_to_prim(X, load_cons(X, y * X->W + (X->W - 1)));
  int i = y * X->W + x;
  if (X->mask[i]) return wall_ghost_prim(cons_to_prim(X, load_cons(X, yc * X->W + xc)));
  return cons_to_prim(X, load_cons(X, i));
}

/* k_render_vals (:1178-1255) + the two-level min/max reduction (:1276-1327) collapsed into one loop */
void o2h_render_vals(const tauh2_params *c, const double *rho, const double *mx, const double *my, const double *E,
                     const uint8_t *mask, int view_mode, double *val, double *vmin, double *vmax) {
  ctx_t X = {c->W, c->H, *c, rho, mx, my, E, mask};
  double mn = 1e300, mxv = -1e300;
  for (int i = 0; i < c->W * c->H; i++) {
    val[i] = 0.0;
    if (mask[i]) continue;
    int x = i % c->W, y = i / c->W;
    prim_t p = cons_to_prim(&X, load_cons(&X, i));
    double v;
    if (view_mode == 0) v = log(p.rho);
    else if (view_mode == 1) v = log(p.p);
    else if (view_mode == 2) v = sqrt(p.u * p.u + p.v * p.v);
    else if (view_mode == 3) {
      double gx = 0.5 * (sample_prim_bc(&X, x, y, x + 1, y).rho - sample_prim_bc(&X, x, y, x - 1, y).rho);
      double gy = 0.5 * (sample_prim_bc(&X, x, y, x, y + 1).rho - sample_prim_bc(&X, x, y, x, y - 1).rho);
      v = log(1e-12 + sqrt(gx * gx + gy * gy));
    } else if (view_mode == 4) {
      double dv_dx = 0.5 * (sample_prim_bc(&X, x, y, x + 1, y).v - sample_prim_bc(&X, x, y, x - 1, y).v);
      double du_dy = 0.5 * (sample_prim_bc(&X, x, y, x, y + 1).u - sample_prim_bc(&X, x, y, x, y - 1).u);
      v = asinh(dv_dx - du_dy);
    } else if (view_mode == 5) {
      v = sqrt(p.u * p.u + p.v * p.v) / d_fmax(sound_speed(&X, p), 1e-30);
    } else {
      v = log(d_fmax(p.p / d_fmax(p.rho, EPS_RHO), 1e-30));
    }
    if (!isfinite(v)) v = 0.0;
    val[i] = v;
    if (v < mn) mn = v;
    if (v > mxv) mxv = v;
  }
  *vmin = mn; *vmax = mxv;
}

/* k_compute_inv_range (:1329-1333) + k_render_pixels (:1257-1274) + get_color (:692-704) */
void o2h_render_pixels(int W, int H, const uint8_t *mask, const double *val, double vmin, double vmax, uint32_t *out) {
  double inv = 1.0 / d_fmax(vmax - vmin, 1e-30);
  for (int i = 0; i < W * H; i++) {
    if (mask[i]) { out[i] = 0xff000000u | (110u << 16) | (110u << 8) | 110u; continue; }
    double t = (val[i] - vmin) * inv;
    if (t < 0) t = 0;
    if (t > 1) t = 1;
    double rr = 255.0 * d_fmin(1.0, d_fmax(0.0, 3.0 * t - 1.0));
    double gg = 255.0 * d_fmin(1.0, d_fmax(0.0, 2.0 - 4.0 * d_fabs(t - 0.5)));
    double bb = 255.0 * d_fmin(1.0, d_fmax(0.0, 2.0 - 3.0 * t));
    out[i] = 0xff000000u | ((uint32_t)(uint8_t)bb << 16) | ((uint32_t)(uint8_t)gg << 8) | (uint32_t)(uint8_t)rr;
  }
}
