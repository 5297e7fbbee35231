/* stencil2d_oracle.c — TEST INFRASTRUCTURE ONLY (CPU oracle, not a product path).
 *
 * Plain-C restatement, IEEE fp32 host semantics (build with -ffp-contract=off),
 * of the periodic 5-point-Laplacian steps of the reference:
 *   - Gray-Scott reaction-diffusion     tau_gray_scott.cu:137-171 (+ init 173-204)
 *   - Burgers viscosity pass            tau_burgers.cu:490-525
 *   - shallow-water viscosity pass      tau_shallow_water.cu:516-547
 *
 * Parity pin: Gray-Scott is pinned by SURVEY.md §8(c) (reference output, 128^2,
 * 100 steps, defaults: sum u = 15681.3368, sum v = 247.689801).
 * The two viscosity passes have NO reference output anywhere (nothing
 * tests them) and the reference kernels update in place while neighbours read
 * (a data race) — "parity unpinned" against reference outputs; this
 * file implements the race-free all-reads-before-writes (Jacobi) semantics with the
 * reference's own arithmetic and is pinned analytically (Fourier-mode eigenvalue of
 * the 5-point Laplacian) in tests/test_oracle_pins.py.
 */
#include "../include/tau_params.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int wrap(int i, int n) { return (i % n + n) % n; } /* tau_gray_scott.cu:137-139 */

/* One forward-Euler Gray-Scott step, out-of-place (ping-pong), :141-171 */
void o2_gs_step(const taugs_params *P, const float *u, const float *v, float *un, float *vn) {
  const int nx = P->nx, ny = P->ny;
  const float dx = P->dx, dt = P->dt, Du = P->Du, Dv = P->Dv, feed = P->feed, kill = P->kill;
  for (int j = 0; j < ny; j++) {
    int jp = wrap(j + 1, ny), jm = wrap(j - 1, ny);
    for (int i = 0; i < nx; i++) {
      int ip = wrap(i + 1, nx), im = wrap(i - 1, nx);
      size_t idx = (size_t)j * nx + i;
      float uc = u[idx], vc = v[idx];
      float lap_u = (u[(size_t)j * nx + ip] + u[(size_t)j * nx + im] + u[(size_t)jp * nx + i] +
                     u[(size_t)jm * nx + i] - 4.0f * uc) / (dx * dx);
      float lap_v = (v[(size_t)j * nx + ip] + v[(size_t)j * nx + im] + v[(size_t)jp * nx + i] +
                     v[(size_t)jm * nx + i] - 4.0f * vc) / (dx * dx);
      float uvv = uc * vc * vc;
      float du = Du * lap_u - uvv + feed * (1.0f - uc);
      float dv = Dv * lap_v + uvv - (feed + kill) * vc;
      un[idx] = uc + dt * du;
      vn[idx] = vc + dt * dv;
    }
  }
}

/* init_pattern, :173-204 — centre square + 64 xorshift32 seeds */
void o2_gs_init(int nx, int ny, uint32_t seed, float *u, float *v) {
  for (size_t i = 0; i < (size_t)nx * ny; i++) { u[i] = 1.0f; v[i] = 0.0f; }
  int cx = nx / 2, cy = ny / 2;
  int r = (nx < ny ? nx : ny) / 12;
  for (int j = -r; j <= r; ++j)
    for (int i = -r; i <= r; ++i) {
      int x = (cx + i + nx) % nx, y = (cy + j + ny) % ny;
      u[(size_t)y * nx + x] = 0.50f;
      v[(size_t)y * nx + x] = 0.25f;
    }
  uint32_t state = seed ? seed : 1u;
  for (int n = 0; n < 64; ++n) {
    state ^= state << 13; state ^= state >> 17; state ^= state << 5;
    int x = (int)(state % (uint32_t)nx);
    state ^= state << 13; state ^= state >> 17; state ^= state << 5;
    int y = (int)(state % (uint32_t)ny);
    u[(size_t)y * nx + x] = 0.35f;
    v[(size_t)y * nx + x] = 0.65f;
  }
}

void o2_gs_params_default(taugs_params *P, int nx, int ny) { /* :43-61 */
  P->nx = nx; P->ny = ny; P->dx = 1.0f; P->dt = 1.0f;
  P->Du = 0.2f; P->Dv = 0.1f; P->feed = 0.03f; P->kill = 0.06f;
}

/* Burgers viscosity, Jacobi semantics (all reads from the input arrays), :490-525.
 * oneD mirrors the kernel's flag (invdy2 = 0). */
void o2_burgers_visc(const taulap_params *P, int oneD, const float *phu, const float *phv,
                     float *ou, float *ov) {
  const int nx = P->nx, ny = P->ny;
  const float u0 = P->u0, nu = P->nu, dt = P->dt;
  float invdx2 = 1.0f / (P->dx * P->dx);
  float invdy2 = oneD ? 0.0f : (1.0f / (P->dy * P->dy));
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      size_t c = (size_t)j * nx + i;
      size_t xp = (size_t)j * nx + wrap(i + 1, nx), xm = (size_t)j * nx + wrap(i - 1, nx);
      size_t yp = (size_t)wrap(j + 1, ny) * nx + i, ym = (size_t)wrap(j - 1, ny) * nx + i;
      const float *ph[2] = {phu, phv};
      float *o[2] = {ou, ov};
      for (int k = 0; k < 2; k++) {
        float cc = u0 * sinhf(ph[k][c]);
        float fxp = u0 * sinhf(ph[k][xp]), fxm = u0 * sinhf(ph[k][xm]);
        float fyp = u0 * sinhf(ph[k][yp]), fym = u0 * sinhf(ph[k][ym]);
        float lap = (fxp - 2.0f * cc + fxm) * invdx2 + (fyp - 2.0f * cc + fym) * invdy2;
        float val = cc + nu * dt * lap;
        o[k][c] = asinhf(val / u0);
      }
    }
}

/* shallow-water viscosity on u,v, Jacobi semantics, :516-547 */
void o2_sw_visc(const taulap_params *P, const float *u, const float *v, float *ou, float *ov) {
  const int nx = P->nx, ny = P->ny;
  const float nu = P->nu, dt = P->dt;
  float invdx2 = 1.0f / (P->dx * P->dx), invdy2 = 1.0f / (P->dy * P->dy);
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      size_t c = (size_t)j * nx + i;
      size_t xp = (size_t)j * nx + wrap(i + 1, nx), xm = (size_t)j * nx + wrap(i - 1, nx);
      size_t yp = (size_t)wrap(j + 1, ny) * nx + i, ym = (size_t)wrap(j - 1, ny) * nx + i;
      float du = (u[xp] - 2.0f * u[c] + u[xm]) * invdx2 + (u[yp] - 2.0f * u[c] + u[ym]) * invdy2;
      float dv = (v[xp] - 2.0f * v[c] + v[xm]) * invdx2 + (v[yp] - 2.0f * v[c] + v[ym]) * invdy2;
      float un = u[c], vn = v[c];
      un += nu * dt * du;
      vn += nu * dt * dv;
      ou[c] = un;
      ov[c] = vn;
    }
}

/* =====================================================================================
 * Full Burgers and shallow-water steps.  No reference output exists for
 * either program ("parity unpinned" against reference outputs); pinned analytically:
 * Burgers by the reference's own Cole-Hopf harness (tau_burgers.cu:256-273, 720-736), shallow
 * water by exact conservation of sum(h) and the lake-at-rest steady state (tests/).
 * The viscosity passes use the race-free Jacobi form above.
 * ===================================================================================== */
typedef struct o2_flow_params {
  int32_t nx, ny;
  float dx, dy;
  float nu;
  float u0;        /* Burgers: asinh scale */
  float g;         /* shallow water */
  float CFL;
  float dtau;
  int32_t muscl;   /* Burgers */
  int32_t visc_substeps;
  int32_t oneD;    /* Burgers Cole-Hopf mode (ny = 1) */
} o2_flow_params;

static inline int wrapi2(int i, int n) { i %= n; if (i < 0) i += n; return i; } /* tau_burgers.cu:94-99 */
static inline float minmodf(float a, float b) { /* :332-334 */
  return (a * b <= 0.0f) ? 0.0f : copysignf(fminf(fabsf(a), fabsf(b)), a);
}

/* max(|u|/dx + |v|/dy), tau_burgers.cu:337-361 + host max :684-692 */
float o2_burgers_smax(const o2_flow_params *P, const float *phu, const float *phv) {
  float invdx = 1.0f / P->dx, invdy = (P->ny > 1 ? 1.0f / P->dy : 0.0f);
  float smax = 1e-12f;
  for (size_t k = 0; k < (size_t)P->nx * P->ny; k++) {
    float u = P->u0 * sinhf(phu[k]), v = P->u0 * sinhf(phv[k]);
    smax = fmaxf(smax, fabsf(u) * invdx + fabsf(v) * invdy);
  }
  return smax;
}

/* one do_step (tau_burgers.cu:677-718) with a given dt_eff: Rusanov(+MUSCL) fluxes, convective
 * update, K viscosity passes.  in -> out */
void o2_burgers_step(const o2_flow_params *P, float dt_eff, const float *phu, const float *phv, float *ou, float *ov) {
  const int nx = P->nx, ny = P->ny;
  const size_t N = (size_t)nx * ny;
  float *Fu = (float *)malloc(N * 4), *Fv = (float *)malloc(N * 4), *Gu = (float *)calloc(N, 4), *Gv = (float *)calloc(N, 4);
  float *a = (float *)malloc(N * 4), *b = (float *)malloc(N * 4);
  const float u0 = P->u0;
#define PH(f, i, j) f[(size_t)wrapi2(j, ny) * nx + wrapi2(i, nx)]
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) { /* flux_x_kernel :364-408 */
      float pUL = PH(phu, i, j), pUR = PH(phu, i + 1, j), pVL = PH(phv, i, j), pVR = PH(phv, i + 1, j);
      if (P->muscl) {
        float pU_Lm = PH(phu, i - 1, j), pU_Rp = PH(phu, i + 2, j), pV_Lm = PH(phv, i - 1, j), pV_Rp = PH(phv, i + 2, j);
        float sUL = 0.5f * minmodf(pUL - pU_Lm, pUR - pUL), sUR = 0.5f * minmodf(pU_Rp - pUR, pUR - pUL);
        float sVL = 0.5f * minmodf(pVL - pV_Lm, pVR - pVL), sVR = 0.5f * minmodf(pV_Rp - pVR, pVR - pVL);
        pUL = pUL + sUL; pUR = pUR - sUR; pVL = pVL + sVL; pVR = pVR - sVR;
      }
      float uL = u0 * sinhf(pUL), vL = u0 * sinhf(pVL), uR = u0 * sinhf(pUR), vR = u0 * sinhf(pVR);
      float FL_u = 0.5f * uL * uL, FL_v = uL * vL, FR_u = 0.5f * uR * uR, FR_v = uR * vR;
      float aa = fmaxf(fabsf(uL), fabsf(uR));
      Fu[(size_t)j * nx + i] = 0.5f * (FL_u + FR_u) - 0.5f * aa * (uR - uL);
      Fv[(size_t)j * nx + i] = 0.5f * (FL_v + FR_v) - 0.5f * aa * (vR - vL);
    }
  if (!P->oneD)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) { /* flux_y_kernel :411-455 */
        float pUB = PH(phu, i, j), pUT = PH(phu, i, j + 1), pVB = PH(phv, i, j), pVT = PH(phv, i, j + 1);
        if (P->muscl) {
          float pU_Bm = PH(phu, i, j - 1), pU_Tp = PH(phu, i, j + 2), pV_Bm = PH(phv, i, j - 1), pV_Tp = PH(phv, i, j + 2);
          float sUB = 0.5f * minmodf(pUB - pU_Bm, pUT - pUB), sUT = 0.5f * minmodf(pU_Tp - pUT, pUT - pUB);
          float sVB = 0.5f * minmodf(pVB - pV_Bm, pVT - pVB), sVT = 0.5f * minmodf(pV_Tp - pVT, pVT - pVB);
          pUB = pUB + sUB; pUT = pUT - sUT; pVB = pVB + sVB; pVT = pVT - sVT;
        }
        float uB = u0 * sinhf(pUB), vB = u0 * sinhf(pVB), uT = u0 * sinhf(pUT), vT = u0 * sinhf(pVT);
        float GL_u = uB * vB, GL_v = 0.5f * vB * vB, GR_u = uT * vT, GR_v = 0.5f * vT * vT;
        float aa = fmaxf(fabsf(vB), fabsf(vT));
        Gu[(size_t)j * nx + i] = 0.5f * (GL_u + GR_u) - 0.5f * aa * (uT - uB);
        Gv[(size_t)j * nx + i] = 0.5f * (GL_v + GR_v) - 0.5f * aa * (vT - vB);
      }
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) { /* update_convective :458-487 */
      size_t id = (size_t)j * nx + i;
      float u = u0 * sinhf(phu[id]), v = u0 * sinhf(phv[id]);
      float invdx = 1.0f / P->dx, invdy = (P->oneD ? 0.0f : (1.0f / P->dy));
      float dFx_u = PH(Fu, i, j) - PH(Fu, i - 1, j), dFx_v = PH(Fv, i, j) - PH(Fv, i - 1, j);
      float dGy_u = P->oneD ? 0.0f : (PH(Gu, i, j) - PH(Gu, i, j - 1));
      float dGy_v = P->oneD ? 0.0f : (PH(Gv, i, j) - PH(Gv, i, j - 1));
      u -= dt_eff * (dFx_u * invdx + dGy_u * invdy);
      v -= dt_eff * (dFx_v * invdx + dGy_v * invdy);
      a[id] = asinhf(u / u0);
      b[id] = asinhf(v / u0);
    }
#undef PH
  int K = (P->visc_substeps > 0 ? P->visc_substeps : 1); /* :709-717 */
  taulap_params L = {nx, ny, P->dx, P->dy, P->nu, dt_eff / K, u0};
  for (int k = 0; k < K; k++) {
    o2_burgers_visc(&L, P->oneD, a, b, ou, ov);
    if (k + 1 < K) { memcpy(a, ou, N * 4); memcpy(b, ov, N * 4); }
  }
  free(Fu); free(Fv); free(Gu); free(Gv); free(a); free(b);
}

/* initialize_host, tau_burgers.cu:246-302 (mode 0 swirl + Gaussian, mode 1 Cole-Hopf 1-D) */
void o2_burgers_init(const o2_flow_params *P, int colehopf, int ck, float ca, float amp, float bsig, float swirl, float rc_cells,
                     float offx, float offy, float asym, float *phu, float *phv) {
  const int nx = P->nx, ny = P->ny;
  if (colehopf) {
    float Lx = P->dx * nx;
    float k = 2.0f * (float)M_PI * ck / Lx;
    for (int i = 0; i < nx; ++i) {
      float x = (i + 0.5f) * P->dx;
      float denom = 1.0f + ca * cosf(k * x);
      float u = (denom != 0.0f) ? (2.0f * P->nu * ca * k * sinf(k * x) / denom) : 0.0f;
      float phi = asinhf(u / P->u0);
      for (int j = 0; j < ny; ++j) { phu[(size_t)j * nx + i] = phi; phv[(size_t)j * nx + i] = 0.0f; }
    }
    return;
  }
  float cx = 0.5f * nx + offx, cy = 0.5f * ny + offy;
  float sig2 = bsig * bsig;
  float rc = rc_cells * fminf(P->dx, P->dy);
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      float dx = i - cx, dy = j - cy;
      float r2 = (dx * dx + dy * dy) / fmaxf(sig2, 1e-6f);
      float theta = atan2f(dy, dx);
      float mod = 1.0f + asym * cosf(theta);
      float rx = dx * P->dx, ry = dy * P->dy;
      float r = sqrtf(rx * rx + ry * ry);
      float u_theta = (r > 0.0f) ? (swirl * r * expf(-0.5f * (r / rc) * (r / rc))) : 0.0f;
      float u = (r > 0.0f) ? (-u_theta * (ry / r)) : 0.0f;
      float v = (r > 0.0f) ? (u_theta * (rx / r)) : 0.0f;
      float g = amp * mod * expf(-0.5f * r2);
      u += 0.5f * g;
      v += -0.5f * g;
      phu[(size_t)j * nx + i] = asinhf(u / P->u0);
      phv[(size_t)j * nx + i] = asinhf(v / P->u0);
    }
}

/* exact 1-D Cole-Hopf solution and relative L2 error, tau_burgers.cu:720-736 */
double o2_burgers_colehopf_relL2(const o2_flow_params *P, int ck, float ca, const float *phu, float t_now) {
  const int nx = P->nx;
  float Lx = P->dx * nx;
  float k = 2.0f * (float)M_PI * ck / Lx;
  float decay = expf(-P->nu * k * k * t_now);
  double num = 0.0, den = 0.0;
  for (int i = 0; i < nx; ++i) {
    float x = (i + 0.5f) * P->dx;
    float u_ex = (2.0f * P->nu * ca * k * decay * sinf(k * x)) / (1.0f + ca * decay * cosf(k * x));
    double u_num = P->u0 * sinh(phu[i]);
    double diff = u_num - u_ex;
    num += diff * diff;
    den += u_ex * u_ex;
  }
  return (den > 0.0) ? sqrt(num / den) : sqrt(num);
}

/* ---- shallow water */
static inline void hll_axis(float hL, float unL, float utL, float hR, float unR, float utR, float g, float *Fh, float *Fn,
                            float *Ft) { /* hll_x / hll_y, tau_shallow_water.cu:327-390 (n = normal, t = tangential momentum) */
  float cL = sqrtf(g * hL), cR = sqrtf(g * hR);
  float sL = fminf(unL - cL, unR - cR), sR = fmaxf(unL + cL, unR + cR);
  float mL = hL * unL, mR = hR * unR, nL = hL * utL, nR = hR * utR;
  float FL_h = mL, FL_n = mL * unL + 0.5f * g * hL * hL, FL_t = mL * utL;
  float FR_h = mR, FR_n = mR * unR + 0.5f * g * hR * hR, FR_t = mR * utR;
  if (sL >= 0.0f) { *Fh = FL_h; *Fn = FL_n; *Ft = FL_t; return; }
  if (sR <= 0.0f) { *Fh = FR_h; *Fn = FR_n; *Ft = FR_t; return; }
  float inv = 1.0f / (sR - sL);
  *Fh = (sR * FL_h - sL * FR_h + sR * sL * (hR - hL)) * inv;
  *Fn = (sR * FL_n - sL * FR_n + sR * sL * (mR - mL)) * inv;
  *Ft = (sR * FL_t - sL * FR_t + sR * sL * (nR - nL)) * inv;
}

float o2_sw_cmax(const o2_flow_params *P, const float *sig, const float *u, const float *v) { /* :394-422, 678-688 */
  float cmax = 0.0f;
  for (size_t k = 0; k < (size_t)P->nx * P->ny; k++) {
    float h = expf(sig[k]);
    float c = sqrtf(P->g * h);
    cmax = fmaxf(cmax, fmaxf(fabsf(u[k]) + c, fabsf(v[k]) + c));
  }
  if (cmax < 1e-12f) cmax = 1e-12f;
  return cmax;
}

/* one do_step (tau_shallow_water.cu:671-705) with a given dt_eff */
void o2_sw_step(const o2_flow_params *P, float dt_eff, const float *sig, const float *u, const float *v, float *osig,
                float *ou, float *ov) {
  const int nx = P->nx, ny = P->ny;
  const size_t N = (size_t)nx * ny;
  float *Fh = (float *)malloc(N * 4), *Fmx = (float *)malloc(N * 4), *Fmy = (float *)malloc(N * 4);
  float *Gh = (float *)malloc(N * 4), *Gmx = (float *)malloc(N * 4), *Gmy = (float *)malloc(N * 4);
  float *tu = (float *)malloc(N * 4), *tv = (float *)malloc(N * 4);
#define AT(f, i, j) f[(size_t)wrapi2(j, ny) * nx + wrapi2(i, nx)]
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      size_t id = (size_t)j * nx + i;
      float hL = expf(AT(sig, i, j)), hR = expf(AT(sig, i + 1, j)), hT = expf(AT(sig, i, j + 1));
      /* x: normal momentum is mx; y: normal momentum is my -> (Fh, Fmx, Fmy) and (Gh, Gmy, Gmx) */
      hll_axis(hL, AT(u, i, j), AT(v, i, j), hR, AT(u, i + 1, j), AT(v, i + 1, j), P->g, &Fh[id], &Fmx[id], &Fmy[id]);
      hll_axis(hL, AT(v, i, j), AT(u, i, j), hT, AT(v, i, j + 1), AT(u, i, j + 1), P->g, &Gh[id], &Gmy[id], &Gmx[id]);
    }
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) { /* update_kernel :474-513 */
      size_t id = (size_t)j * nx + i;
      float h = expf(sig[id]);
      float mx = h * u[id], my = h * v[id];
      float dFx_h = AT(Fh, i, j) - AT(Fh, i - 1, j), dFx_mx = AT(Fmx, i, j) - AT(Fmx, i - 1, j), dFx_my = AT(Fmy, i, j) - AT(Fmy, i - 1, j);
      float dGy_h = AT(Gh, i, j) - AT(Gh, i, j - 1), dGy_mx = AT(Gmx, i, j) - AT(Gmx, i, j - 1), dGy_my = AT(Gmy, i, j) - AT(Gmy, i, j - 1);
      float invdx = 1.0f / P->dx, invdy = 1.0f / P->dy;
      h -= dt_eff * (dFx_h * invdx + dGy_h * invdy);
      mx -= dt_eff * (dFx_mx * invdx + dGy_mx * invdy);
      my -= dt_eff * (dFx_my * invdx + dGy_my * invdy);
      h = fmaxf(h, 1e-6f);
      osig[id] = logf(h);
      tu[id] = mx / h;
      tv[id] = my / h;
    }
#undef AT
  if (P->nu > 0.0f) { /* :701-704 */
    taulap_params L = {nx, ny, P->dx, P->dy, P->nu, dt_eff, 1.0f};
    o2_sw_visc(&L, tu, tv, ou, ov);
  } else { memcpy(ou, tu, N * 4); memcpy(ov, tv, N * 4); }
  free(Fh); free(Fmx); free(Fmy); free(Gh); free(Gmx); free(Gmy); free(tu); free(tv);
}

/* initialize_host, tau_shallow_water.cu:238-276 */
void o2_sw_init(const o2_flow_params *P, float H0, float bumpAmp, float bumpSigma, float offx, float offy, float asym,
                float swirl, float swirlRc, float *sig, float *u, float *v) {
  const int nx = P->nx, ny = P->ny;
  float cx = 0.5f * nx + offx, cy = 0.5f * ny + offy;
  float sig2 = bumpSigma * bumpSigma;
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      float dx = i - cx, dy = j - cy;
      float r2 = (dx * dx + dy * dy) / sig2;
      float theta = atan2f(dy, dx);
      float mod = 1.0f + asym * cosf(theta);
      float h = H0 + (bumpAmp * mod) * expf(-0.5f * r2);
      size_t id = (size_t)j * nx + i;
      sig[id] = logf(fmaxf(h, 1e-6f));
      float rx = dx * P->dx, ry = dy * P->dy;
      float r = sqrtf(rx * rx + ry * ry);
      float rc = swirlRc * fminf(P->dx, P->dy);
      float u_theta = (r > 0.0f && swirl != 0.0f) ? (swirl * r * expf(-0.5f * (r / rc) * (r / rc))) : 0.0f;
      u[id] = (r > 0.0f) ? (-u_theta * (ry / r)) : 0.0f;
      v[id] = (r > 0.0f) ? (u_theta * (rx / r)) : 0.0f;
    }
}
