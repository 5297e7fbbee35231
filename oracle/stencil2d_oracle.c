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
