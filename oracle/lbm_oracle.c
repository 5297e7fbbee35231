/* lbm_oracle.c — TEST INFRASTRUCTURE ONLY (CPU oracle, not a product path).
 *
 * Plain-C restatement of the D2Q9 BGK lattice-Boltzmann program of the reference (tau_lbm.cu): init_kernel
 * (:72-92), collide_stream_kernel (:94-132) and render_kernel (:134-153), IEEE fp32 host semantics, no FMA
 * contraction (build with -ffp-contract=off), every expression in the reference's association order.  The
 * reference's push scheme writes every slot of fout exactly once (a fluid cell pushes post_q to its q-neighbour
 * unless that neighbour is solid or beyond the y walls, in which case it keeps it in its own opposite slot; a
 * solid cell reflects its nine values in place), so a sequential sweep gives the one result any GPU
 * schedule gives.  Only tests/ may load this file.
 *
 * Parity pin: the reference's own init_kernel / collide_stream_kernel built for gfx950 (oracle/_ref/tau_lbm.ieee.co,
 * oracle/build_ref.sh) and run on the MI355X — the engine, which is bit-identical to this file, reproduces them bit
 * for bit (tests/test_gpu_ref2d.py; the sheared start to 1 ulp: device sinf vs glibc sinf).  On the CPU, where no
 * reference output exists (SURVEY §8(c) recorded none for tau_lbm.cu), tests/test_vis_oracle.py pins it against
 * closed forms (mass conservation, the equilibrium fixed point, bounce-back symmetry).
 */
#include "../include/tau_params.h"
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static const int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};   /* :57-59 */
static const int EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
static const int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
static const float WQ[9] = {4.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f,
                            1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f};

static inline float feq(int q, float rho, float ux, float uy) { /* :67-71 */
  const float cu = 3.0f * (EX[q] * ux + EY[q] * uy);
  const float u2 = ux * ux + uy * uy;
  return WQ[q] * rho * (1.0f + cu + 0.5f * cu * cu - 1.5f * u2);
}

void olbm_params_default(taulbm_params *P) { /* :43-55 */
  P->nx = 512; P->ny = 256; P->obstacle = 1; P->tau = 0.56f; P->drive = 1.0e-6f; P->rho0 = 1.0f;
  P->obstacle_radius = 32.0f;
}

/* init_kernel, :72-92: channel walls at j = 0, ny-1, optional cylinder, sheared equilibrium */
void olbm_init(const taulbm_params *P, float *f, uint8_t *solid) {
  const int nx = P->nx, ny = P->ny;
  const size_t cells = (size_t)nx * ny;
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      const size_t p = (size_t)j * nx + i;
      const float cx = 0.28f * nx, cy = 0.5f * ny;
      const float dx = i - cx, dy = j - cy;
      const int wall = (j == 0 || j == ny - 1);
      const int cyl = P->obstacle && (dx * dx + dy * dy < P->obstacle_radius * P->obstacle_radius);
      solid[p] = (wall || cyl) ? 1 : 0;
      const float shear = 0.015f * sinf(2.0f * 3.14159265f * j / (ny > 1 ? ny - 1 : 1));
      for (int q = 0; q < 9; q++) f[q * cells + p] = feq(q, P->rho0, shear, 0.0f);
    }
}

/* collide_stream_kernel, :94-132 */
void olbm_step(const taulbm_params *P, const float *fin, float *fout, const uint8_t *solid) {
  const int nx = P->nx, ny = P->ny;
  const size_t cells = (size_t)nx * ny;
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      const size_t p = (size_t)j * nx + i;
      float local[9];
      for (int q = 0; q < 9; q++) local[q] = fin[q * cells + p];
      if (solid[p]) {
        for (int q = 0; q < 9; q++) fout[OPP[q] * cells + p] = local[q];
        continue;
      }
      float rho = 0.0f, ux = 0.0f, uy = 0.0f;
      for (int q = 0; q < 9; q++) {
        rho += local[q];
        ux += local[q] * EX[q];
        uy += local[q] * EY[q];
      }
      rho = fmaxf(rho, 1.0e-6f);
      ux = ux / rho + P->drive;
      uy /= rho;
      const float omega = 1.0f / P->tau;
      for (int q = 0; q < 9; q++) {
        const float post = local[q] - omega * (local[q] - feq(q, rho, ux, uy));
        const int ni = (i + EX[q] + nx) % nx;
        const int nj = j + EY[q];
        if (nj < 0 || nj >= ny || solid[(size_t)nj * nx + ni])
          fout[OPP[q] * cells + p] = post;
        else
          fout[q * cells + (size_t)nj * nx + ni] = post;
      }
    }
}

/* render_kernel, :134-153: |u| per fluid cell, -1 in solids */
void olbm_speed(const taulbm_params *P, const float *f, const uint8_t *solid, float *speed) {
  const size_t cells = (size_t)P->nx * P->ny;
  for (size_t p = 0; p < cells; p++) {
    if (solid[p]) { speed[p] = -1.0f; continue; }
    float rho = 0.0f, ux = 0.0f, uy = 0.0f;
    for (int q = 0; q < 9; q++) {
      const float fq = f[q * cells + p];
      rho += fq;
      ux += fq * EX[q];
      uy += fq * EY[q];
    }
    speed[p] = hypotf(ux / rho, uy / rho);
  }
}
