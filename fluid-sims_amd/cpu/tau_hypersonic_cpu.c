/* tau_hypersonic_cpu.c — the reference's CPU 2D Euler solver (BASELINE config 1), restated.
 *
 * tau_hypersonic.c / tau_hypersonic_simd.c are CPU programs in the reference too (fp64, AoS,
 * MUSCL-Hancock with the MC limiter + HLLC, slip-wall circle, Mach-15 inflow); there is no GPU
 * kernel for this scheme (the CUDA 2D solver is a different scheme).  This file is
 * that CPU path with the grid size made a run-time parameter (the reference fixes W = H = 300
 * at compile time, tau_hypersonic.c:12-13).  It backs the `tau_hypersonic` /
 * `tau_hypersonic_simd` drivers and the "2D CPU" baseline leg of bench.py.
 *
 * Arithmetic follows the reference expression by expression so that, built with the
 * reference's flags (gcc -O3, Makefile:57-58), it reproduces the reference's outputs to the
 * last bit (pinned in tests/test_cpu2d.py against SURVEY §8c check-values).  What is
 * restructured: the reference re-runs the MUSCL reconstruction four times per face (two of
 * them dead, tau_hypersonic.c:532-560); here every cell's predicted face states are computed
 * once per axis — same functions of the same inputs, hence the same bits.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#if defined(__AVX2__)
#include <immintrin.h>
#endif

#define GAMMA 1.4      /* tau_hypersonic.c:16 */
#define CFL 0.3        /* :17 */
#define EPS_RHO 1e-10  /* :21 */
#define EPS_P 1e-10    /* :22 */

typedef struct { double rho, mx, my, E; } cons_t;   /* :24-29 */
typedef struct { double rho, u, v, p; } prim_t;     /* :31-36 */

typedef struct th2_sim {
  int W, H;
  cons_t *U, *Unew;
  unsigned char *mask;
  prim_t *pL, *pR;    /* predicted low / high face state of every cell, current axis */
  double t;
  int simd_dt;        /* 1: compute_dt as tau_hypersonic_simd.c:556-637 (AVX2 association) */
} th2_sim;

static inline double minmod(double a, double b) { /* :49-53 */
  if (a * b <= 0.0) return 0.0;
  return (fabs(a) < fabs(b)) ? a : b;
}
static inline double mc_limiter(double dl, double dc, double dr) { /* :55-61 */
  double mm1 = minmod(dl, dr);
  double mm2 = minmod(dc, 2.0 * dl);
  double mm3 = minmod(dc, 2.0 * dr);
  return minmod(mm1, minmod(mm2, mm3));
}
static inline prim_t cons_to_prim(cons_t c) { /* :63-80 */
  prim_t p;
  double rho = fmax(c.rho, EPS_RHO);
  double inv = 1.0 / rho;
  double u = c.mx * inv, v = c.my * inv;
  double kin = 0.5 * rho * (u * u + v * v);
  double eint = c.E - kin;
  p.rho = rho; p.u = u; p.v = v;
  p.p = (GAMMA - 1.0) * fmax(eint, EPS_P);
  return p;
}
static inline cons_t prim_to_cons(prim_t p) { /* :82-92 */
  cons_t c;
  double rho = fmax(p.rho, EPS_RHO), pr = fmax(p.p, EPS_P);
  c.rho = rho; c.mx = rho * p.u; c.my = rho * p.v;
  c.E = pr / (GAMMA - 1.0) + 0.5 * rho * (p.u * p.u + p.v * p.v);
  return c;
}
static inline double sound_speed(prim_t p) { /* :94-96 */
  return sqrt(GAMMA * fmax(p.p, EPS_P) / fmax(p.rho, EPS_RHO));
}
static inline cons_t flux_axis(cons_t c, int ax) { /* flux_x / flux_y, :98-115 */
  prim_t p = cons_to_prim(c);
  cons_t f;
  if (ax == 0) {
    f.rho = c.mx; f.mx = c.mx * p.u + p.p; f.my = c.my * p.u; f.E = (c.E + p.p) * p.u;
  } else {
    f.rho = c.my; f.mx = c.mx * p.v; f.my = c.my * p.v + p.p; f.E = (c.E + p.p) * p.v;
  }
  return f;
}

/* HLLC, Davis wave speeds, no degeneracy guards, :117-243 */
static cons_t hllc_axis(cons_t UL, cons_t UR, int ax) {
  prim_t L = cons_to_prim(UL), R = cons_to_prim(UR);
  double aL = sound_speed(L), aR = sound_speed(R);
  double nL = ax ? L.v : L.u, nR = ax ? R.v : R.u;
  double SL = fmin(nL - aL, nR - aR), SR = fmax(nL + aL, nR + aR);
  cons_t FL = flux_axis(UL, ax), FR = flux_axis(UR, ax);
  if (SL >= 0.0) return FL;
  if (SR <= 0.0) return FR;
  double rhoL = L.rho, rhoR = R.rho, pL = L.p, pR = R.p;
  double num = pR - pL + rhoL * nL * (SL - nL) - rhoR * nR * (SR - nR);
  double den = rhoL * (SL - nL) - rhoR * (SR - nR);
  double SM = num / den;
  double pStar = pL + rhoL * (SL - nL) * (SM - nL);
  pStar = fmax(pStar, EPS_P);
  cons_t F;
  if (SM >= 0.0) {
    double rs = rhoL * (SL - nL) / (SL - SM);
    double mxs = ax ? rs * L.u : rs * SM, mys = ax ? rs * SM : rs * L.v;
    double Es = ((SL - nL) * UL.E - pL * nL + pStar * SM) / (SL - SM);
    F.rho = FL.rho + SL * (rs - UL.rho);
    F.mx = FL.mx + SL * (mxs - UL.mx);
    F.my = FL.my + SL * (mys - UL.my);
    F.E = FL.E + SL * (Es - UL.E);
  } else {
    double rs = rhoR * (SR - nR) / (SR - SM);
    double mxs = ax ? rs * R.u : rs * SM, mys = ax ? rs * SM : rs * R.v;
    double Es = ((SR - nR) * UR.E - pR * nR + pStar * SM) / (SR - SM);
    F.rho = FR.rho + SR * (rs - UR.rho);
    F.mx = FR.mx + SR * (mxs - UR.mx);
    F.my = FR.my + SR * (mys - UR.my);
    F.E = FR.E + SR * (Es - UR.E);
  }
  return F;
}

static inline prim_t inflow_state(void) { /* :245-254: Mach 15 */
  const double mach = 15.0, rho = 1.0, p = 1.0;
  double a = sqrt(GAMMA * p / rho);
  prim_t s = {rho, mach * a, 0.0, p};
  return s;
}

static inline cons_t reflect_slip(cons_t inside, double nx, double ny) { /* :279-293 */
  prim_t p = cons_to_prim(inside);
  double vn = p.u * nx + p.v * ny;
  double ut = -p.u * ny + p.v * nx;
  vn = -vn;
  double u = vn * nx - ut * ny, v = vn * ny + ut * nx;
  prim_t g = {p.rho, u, v, p.p};
  return prim_to_cons(g);
}

static inline cons_t neighbor_or_wall(const th2_sim *S, int x, int y, int dxc, int dyc, double nx, double ny) { /* :295-314 */
  int xn = x + dxc, yn = y + dyc;
  if (xn < 0) return prim_to_cons(inflow_state());
  if (xn >= S->W) return S->U[y * S->W + (S->W - 1)];
  if (yn < 0) yn = 0;
  if (yn >= S->H) yn = S->H - 1;
  int j = yn * S->W + xn;
  if (S->mask[j]) return reflect_slip(S->U[y * S->W + x], nx, ny);
  return S->U[j];
}

static inline void enforce_positive_faces(prim_t *qm, prim_t qc, prim_t *qp) { /* :320-346 */
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
  qm->rho = fmax(qm->rho, EPS_RHO); qp->rho = fmax(qp->rho, EPS_RHO);
  qm->p = fmax(qm->p, EPS_P);       qp->p = fmax(qp->p, EPS_P);
}

/* reconstruct_x / reconstruct_y, :348-420 */
static inline void reconstruct(const th2_sim *S, int x, int y, int ax, prim_t *qL, prim_t *qR) {
  cons_t Uc = S->U[y * S->W + x];
  cons_t Um = ax ? neighbor_or_wall(S, x, y, 0, -1, 0, 1) : neighbor_or_wall(S, x, y, -1, 0, 1, 0);
  cons_t Up = ax ? neighbor_or_wall(S, x, y, 0, +1, 0, 1) : neighbor_or_wall(S, x, y, +1, 0, 1, 0);
  prim_t qm = cons_to_prim(Um), qc = cons_to_prim(Uc), qp = cons_to_prim(Up);
#define SLOPE(f) mc_limiter(qc.f - qm.f, 0.5 * (qp.f - qm.f), qp.f - qc.f)
  double s_rho = SLOPE(rho), s_u = SLOPE(u), s_v = SLOPE(v), s_p = SLOPE(p);
#undef SLOPE
  prim_t l = {qc.rho - 0.5 * s_rho, qc.u - 0.5 * s_u, qc.v - 0.5 * s_v, qc.p - 0.5 * s_p};
  prim_t r = {qc.rho + 0.5 * s_rho, qc.u + 0.5 * s_u, qc.v + 0.5 * s_v, qc.p + 0.5 * s_p};
  enforce_positive_faces(&l, qc, &r);
  *qL = l; *qR = r;
}

static inline prim_t half_step_predict(prim_t q, cons_t dF, double h) { /* :422-448 */
  cons_t c = prim_to_cons(q);
  c.rho -= h * dF.rho; c.mx -= h * dF.mx; c.my -= h * dF.my; c.E -= h * dF.E;
  prim_t out = cons_to_prim(c);
  out.rho = fmax(out.rho, EPS_RHO);
  out.p = fmax(out.p, EPS_P);
  return out;
}

/* predicted face states of every fluid cell along one axis (the reference recomputes these per
 * face, :545-570 / :618-637) */
static void predict_axis(th2_sim *S, int ax, double half_dt) {
  for (int y = 0; y < S->H; y++)
    for (int x = 0; x < S->W; x++) {
      int i = y * S->W + x;
      if (S->mask[i]) continue;
      prim_t l, r;
      reconstruct(S, x, y, ax, &l, &r);
      cons_t Ff = flux_axis(prim_to_cons(r), ax), Fb = flux_axis(prim_to_cons(l), ax);
      cons_t dF = {Ff.rho - Fb.rho, Ff.mx - Fb.mx, Ff.my - Fb.my, Ff.E - Fb.E};
      S->pR[i] = half_step_predict(r, dF, half_dt);
      S->pL[i] = half_step_predict(l, dF, half_dt);
    }
}

th2_sim *th2_create(int W, int H, int simd_dt) {
  th2_sim *S = (th2_sim *)calloc(1, sizeof(*S));
  size_t n = (size_t)W * H;
  S->W = W; S->H = H; S->simd_dt = simd_dt;
  S->U = (cons_t *)malloc(n * sizeof(cons_t));
  S->Unew = (cons_t *)malloc(n * sizeof(cons_t));
  S->pL = (prim_t *)malloc(n * sizeof(prim_t));
  S->pR = (prim_t *)malloc(n * sizeof(prim_t));
  S->mask = (unsigned char *)malloc(n);
  return S;
}
void th2_destroy(th2_sim *S) {
  if (!S) return;
  free(S->U); free(S->Unew); free(S->pL); free(S->pR); free(S->mask); free(S);
}

void th2_init(th2_sim *S) { /* init_sim, :450-475 */
  S->t = 0.0;
  int cx = S->W / 3, cy = S->H / 2, r = S->H / 6;
  prim_t inflow = inflow_state();
  for (int y = 0; y < S->H; y++)
    for (int x = 0; x < S->W; x++) {
      int i = y * S->W + x, dx = x - cx, dy = y - cy;
      S->mask[i] = (dx * dx + dy * dy < r * r) ? 1 : 0;
      if (S->mask[i]) { prim_t s = {inflow.rho, 0.0, 0.0, inflow.p}; S->U[i] = prim_to_cons(s); }
      else S->U[i] = prim_to_cons(inflow);
    }
}

double th2_compute_dt(const th2_sim *S) { /* :477-498 ; SIMD variant tau_hypersonic_simd.c:556-637 */
  double maxs = 1e-12;
  int N = S->W * S->H, i = 0;
#if defined(__AVX2__)
  if (S->simd_dt) {
    const double *Uf = (const double *)(const void *)S->U;
    __m256d vmaxs = _mm256_set1_pd(maxs);
    for (; i + 4 <= N; i += 4) {
      if (S->mask[i] | S->mask[i + 1] | S->mask[i + 2] | S->mask[i + 3]) {
        for (int k = 0; k < 4; k++) {
          int j = i + k;
          if (S->mask[j]) continue;
          prim_t p = cons_to_prim(S->U[j]);
          double a = sound_speed(p), sx = fabs(p.u) + a, sy = fabs(p.v) + a;
          if (sx > maxs) maxs = sx;
          if (sy > maxs) maxs = sy;
        }
        /* the reference re-seeds the vector max from the scalar one here (:585) */
        vmaxs = _mm256_set1_pd(maxs);
        continue;
      }
      __m256d vrho = _mm256_set_pd(Uf[4 * (i + 3)], Uf[4 * (i + 2)], Uf[4 * (i + 1)], Uf[4 * i]);
      __m256d vmx = _mm256_set_pd(Uf[4 * (i + 3) + 1], Uf[4 * (i + 2) + 1], Uf[4 * (i + 1) + 1], Uf[4 * i + 1]);
      __m256d vmy = _mm256_set_pd(Uf[4 * (i + 3) + 2], Uf[4 * (i + 2) + 2], Uf[4 * (i + 1) + 2], Uf[4 * i + 2]);
      __m256d vE = _mm256_set_pd(Uf[4 * (i + 3) + 3], Uf[4 * (i + 2) + 3], Uf[4 * (i + 1) + 3], Uf[4 * i + 3]);
      vrho = _mm256_max_pd(vrho, _mm256_set1_pd(EPS_RHO));
      __m256d inv = _mm256_div_pd(_mm256_set1_pd(1.0), vrho);
      __m256d vu = _mm256_mul_pd(vmx, inv), vv = _mm256_mul_pd(vmy, inv);
      __m256d sum = _mm256_add_pd(_mm256_mul_pd(vu, vu), _mm256_mul_pd(vv, vv));
      __m256d kin = _mm256_mul_pd(_mm256_set1_pd(0.5), _mm256_mul_pd(vrho, sum)); /* 0.5*(rho*(uu+vv)), simd:524 */
      __m256d eint = _mm256_max_pd(_mm256_sub_pd(vE, kin), _mm256_set1_pd(EPS_P));
      __m256d vp = _mm256_mul_pd(_mm256_set1_pd(GAMMA - 1.0), eint);
      __m256d a = _mm256_sqrt_pd(_mm256_div_pd(_mm256_mul_pd(_mm256_set1_pd(GAMMA), vp), vrho));
      __m256d absu = _mm256_andnot_pd(_mm256_set1_pd(-0.0), vu), absv = _mm256_andnot_pd(_mm256_set1_pd(-0.0), vv);
      __m256d s = _mm256_max_pd(_mm256_add_pd(absu, a), _mm256_add_pd(absv, a));
      vmaxs = _mm256_max_pd(vmaxs, s);
    }
    double tmp[4];
    _mm256_storeu_pd(tmp, vmaxs);
    maxs = fmax(fmax(tmp[0], tmp[1]), fmax(tmp[2], tmp[3]));
  }
#endif
  for (; i < N; i++) {
    if (S->mask[i]) continue;
    prim_t p = cons_to_prim(S->U[i]);
    double a = sound_speed(p), sx = fabs(p.u) + a, sy = fabs(p.v) + a;
    if (sx > maxs) maxs = sx;
    if (sy > maxs) maxs = sy;
  }
  double dx = 1.0, dy = 1.0;
  return CFL * fmin(dx, dy) / maxs;
}

/* step_physics, :500-674 */
double th2_step(th2_sim *S) {
  const int W = S->W, H = S->H;
  double dx = 1.0, dy = 1.0;
  double dt = th2_compute_dt(S);
  double dt_dx = dt / dx, dt_dy = dt / dy;
  double half_dt_dx = 0.5 * dt_dx, half_dt_dy = 0.5 * dt_dy;

  cons_t inflowC = prim_to_cons(inflow_state()); /* :508-514 */
  for (int y = 0; y < H; y++)
    if (!S->mask[y * W]) S->U[y * W] = inflowC;
  memcpy(S->Unew, S->U, (size_t)W * H * sizeof(cons_t));

  /* x faces x = 1 .. W-1 (the boundary faces are not swept, :519) */
  predict_axis(S, 0, half_dt_dx);
  for (int y = 0; y < H; y++)
    for (int x = 1; x < W; x++) {
      int iL = y * W + x - 1, iR = y * W + x;
      if (S->mask[iL] && S->mask[iR]) continue;
      prim_t qL = S->mask[iL] ? cons_to_prim(reflect_slip(S->U[iR], 1, 0)) : S->pR[iL];
      prim_t qR = S->mask[iR] ? cons_to_prim(reflect_slip(S->U[iL], 1, 0)) : S->pL[iR];
      qL.rho = fmax(qL.rho, EPS_RHO); qL.p = fmax(qL.p, EPS_P);
      qR.rho = fmax(qR.rho, EPS_RHO); qR.p = fmax(qR.p, EPS_P);
      cons_t F = hllc_axis(prim_to_cons(qL), prim_to_cons(qR), 0);
      if (!S->mask[iL]) { S->Unew[iL].rho -= dt_dx * F.rho; S->Unew[iL].mx -= dt_dx * F.mx; S->Unew[iL].my -= dt_dx * F.my; S->Unew[iL].E -= dt_dx * F.E; }
      if (!S->mask[iR]) { S->Unew[iR].rho += dt_dx * F.rho; S->Unew[iR].mx += dt_dx * F.mx; S->Unew[iR].my += dt_dx * F.my; S->Unew[iR].E += dt_dx * F.E; }
    }
  /* y faces y = 1 .. H-1 */
  predict_axis(S, 1, half_dt_dy);
  for (int y = 1; y < H; y++)
    for (int x = 0; x < W; x++) {
      int iB = (y - 1) * W + x, iT = y * W + x;
      if (S->mask[iB] && S->mask[iT]) continue;
      prim_t qB = S->mask[iB] ? cons_to_prim(reflect_slip(S->U[iT], 0, 1)) : S->pR[iB];
      prim_t qT = S->mask[iT] ? cons_to_prim(reflect_slip(S->U[iB], 0, 1)) : S->pL[iT];
      qB.rho = fmax(qB.rho, EPS_RHO); qB.p = fmax(qB.p, EPS_P);
      qT.rho = fmax(qT.rho, EPS_RHO); qT.p = fmax(qT.p, EPS_P);
      cons_t F = hllc_axis(prim_to_cons(qB), prim_to_cons(qT), 1);
      if (!S->mask[iB]) { S->Unew[iB].rho -= dt_dy * F.rho; S->Unew[iB].mx -= dt_dy * F.mx; S->Unew[iB].my -= dt_dy * F.my; S->Unew[iB].E -= dt_dy * F.E; }
      if (!S->mask[iT]) { S->Unew[iT].rho += dt_dy * F.rho; S->Unew[iT].mx += dt_dy * F.mx; S->Unew[iT].my += dt_dy * F.my; S->Unew[iT].E += dt_dy * F.E; }
    }
  /* floors and copy back, :659-671 */
  for (int i = 0; i < W * H; i++) {
    if (S->mask[i]) continue;
    S->Unew[i].rho = fmax(S->Unew[i].rho, EPS_RHO);
    prim_t p = cons_to_prim(S->Unew[i]);
    if (p.p <= EPS_P) { p.p = EPS_P; S->Unew[i] = prim_to_cons(p); }
    S->U[i] = S->Unew[i];
  }
  S->t += dt;
  return dt;
}

double th2_time(const th2_sim *S) { return S->t; }
const double *th2_state(const th2_sim *S) { return (const double *)S->U; }   /* AoS rho,mx,my,E */
const unsigned char *th2_mask(const th2_sim *S) { return S->mask; }

/* fluid cell count and sums of rho, mx, my, E over fluid cells (the check-values of SURVEY §8c) */
void th2_sums(const th2_sim *S, long *fluid, double out[4]) {
  long n = 0;
  double s[4] = {0, 0, 0, 0};
  for (int i = 0; i < S->W * S->H; i++) {
    if (S->mask[i]) continue;
    n++;
    s[0] += S->U[i].rho; s[1] += S->U[i].mx; s[2] += S->U[i].my; s[3] += S->U[i].E;
  }
  *fluid = n;
  memcpy(out, s, sizeof(s));
}
