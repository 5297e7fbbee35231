/* sph_oracle.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle, not a product path).
 *
 * C++ restatement of the 2D WCSPH sub-step of the reference (tau_sph.cu), IEEE fp32 host semantics
 * (-ffp-contract=off, libm expf/logf/powf):
 *   k_clear_heads/k_build_cells (:159-176) -> k_density_pressure_cell (:178-213) ->
 *   k_forces_cell (:215-272) -> k_integrate (:324-355), host dt / log-time bookkeeping (:665-720).
 * The linked lists are built by visiting particles in ascending index, so every cell list is
 * traversed in DESCENDING particle index — one legal order of the reference's atomicExch build
 * (its real order is nondeterministic) and the order the survey's check-values
 * were produced with.  Rain / XSPH are off (out of scope).
 * C++ only because reset_particles (:493-510) draws from std::mt19937 +
 * std::uniform_real_distribution<float>, whose stream is libstdc++-specific.
 *
 * Parity pin: SURVEY.md §8(c): N = 4096, 3 steps, rain off: Gx = Gy = 16,
 * sum x = 2047.89578333, sum y = 1226.84338076, mean rho = 1.78871685.
 */
#include "../include/tau_params.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

namespace {
struct f2 { float x, y; };

inline float W_cubic(float r, float h) { /* :105-116 */
  float q = r / h;
  const float alpha = 10.0f / (7.0f * M_PI * h * h);
  if (q < 1.0f) { float q2 = q * q, q3 = q2 * q; return alpha * (1.f - 1.5f * q2 + 0.75f * q3); }
  else if (q < 2.0f) { float t = 2.f - q; return alpha * 0.25f * t * t * t; }
  return 0.f;
}
inline f2 gradW_cubic(f2 rij, float r, float h) { /* :118-133 */
  if (r <= 1e-8f || r >= 2.0f * h) return f2{0.f, 0.f};
  float q = r / h;
  const float alpha = 10.0f / (7.0f * M_PI * h * h);
  float dWdq;
  if (q < 1.0f) dWdq = alpha * (-3.0f * q + 2.25f * q * q);
  else { float t = 2.0f - q; dWdq = alpha * (-0.75f * t * t); }
  float invr = 1.0f / r;
  float dWdr = dWdq / h;
  return f2{dWdr * rij.x * invr, dWdr * rij.y * invr};
}
inline int grid_c(float x, float cell, int G) { /* grid_x / grid_y, :141-157 */
  int g = (int)floorf(x / cell);
  if (g < 0) g = 0;
  if (g >= G) g = G - 1;
  return g;
}
} // namespace

struct osph {
  tausph_params P;
  int Gx, Gy;
  float cell, h, mass;
  float tau, t;
  long step;
  std::vector<f2> pos, vel, acc;
  std::vector<float> accAbs; /* diagnostic: sum over pairs of |term| + pressure-scale term = conditioning scale of acc */
  std::vector<float> s, press;
  std::vector<int> head, next, cellOf;
  float rain_carry;
  long rain_spawned;
};

extern "C" {

void osph_params_default(tausph_params *P, int N) { /* :49-85 */
  P->N = N; P->boxX = 1.0f; P->boxY = 1.0f; P->dTau = 1.0f; P->t0 = 1.0f; P->CFL = 1.0f;
  P->rho0 = 1.0f; P->c0 = 1.0f; P->gammaEOS = 1.0f; P->hMul = 2.0f; P->viscAlpha = 0.25f; P->gravity = 9.81f;
  P->useVisc = 1; P->useGrav = 1; P->viscSub = 1; P->seed = 69420;
  P->useXSPH = 0; P->xsphEps = 0.25f; P->rain = 0;
}

/* reset_particles, :493-510 — jittered lattice in the lower 60 % of the box */
void osph_reset_particles(const tausph_params *P, float *pos_xy, float *vel_xy) {
  std::mt19937 rng(P->seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  int nSide = (int)sqrtf((float)P->N);
  int nx = nSide, ny = (P->N + nSide - 1) / nSide;
  float padX = 0.05f * P->boxX, padY = 0.05f * P->boxY;
  float width = P->boxX - 2 * padX, height = 0.6f * P->boxY - padY;
  for (int i = 0; i < P->N; ++i) {
    int ix = i % nx, iy = i / nx;
    float fx = (ix + 0.5f) / nx, fy = (iy + 0.5f) / ny;
    float x = padX + fx * width, y = padY + fy * height;
    x += (U(rng) - 0.5f) * 0.2f * width / nx;
    y += (U(rng) - 0.5f) * 0.2f * height / ny;
    pos_xy[2 * i] = x; pos_xy[2 * i + 1] = y;
    vel_xy[2 * i] = 0.f; vel_xy[2 * i + 1] = 0.f;
  }
}

osph *osph_create(const tausph_params *P) {
  osph *S = new osph();
  S->P = *P;
  const float area = P->boxX * P->boxY; /* :573-576 */
  S->mass = (P->rho0 * area) / P->N;
  const float spacing = sqrtf(area / P->N);
  S->h = P->hMul * spacing;
  S->cell = 2.0f * S->h; /* ensure_cell_buffers, :512-521 */
  S->Gx = (int)ceilf(P->boxX / S->cell); S->Gy = (int)ceilf(P->boxY / S->cell);
  if (S->Gx < 1) S->Gx = 1;
  if (S->Gy < 1) S->Gy = 1;
  S->tau = 0.f; S->t = P->t0 * expf(S->tau); S->step = 0;
  S->pos.resize(P->N); S->vel.resize(P->N); S->acc.assign(P->N, f2{0, 0}); S->accAbs.assign(P->N, 0.f);
  S->s.assign(P->N, 0.f); S->press.assign(P->N, 0.f);
  S->head.resize((size_t)S->Gx * S->Gy); S->next.resize(P->N); S->cellOf.resize(P->N);
  osph_reset_particles(P, (float *)S->pos.data(), (float *)S->vel.data());
  return S;
}
void osph_destroy(osph *S) { delete S; }
void osph_grid(const osph *S, int *Gx, int *Gy, float *cell, float *h, float *mass) {
  *Gx = S->Gx; *Gy = S->Gy; *cell = S->cell; *h = S->h; *mass = S->mass;
}
void osph_set_state(osph *S, const float *pos_xy, const float *vel_xy) {
  memcpy(S->pos.data(), pos_xy, sizeof(f2) * S->P.N);
  memcpy(S->vel.data(), vel_xy, sizeof(f2) * S->P.N);
}
void osph_get_state(const osph *S, float *pos_xy, float *vel_xy, float *acc_xy, float *s, float *press, int *cellOf) {
  if (pos_xy) memcpy(pos_xy, S->pos.data(), sizeof(f2) * S->P.N);
  if (vel_xy) memcpy(vel_xy, S->vel.data(), sizeof(f2) * S->P.N);
  if (acc_xy) memcpy(acc_xy, S->acc.data(), sizeof(f2) * S->P.N);
  if (s) memcpy(s, S->s.data(), sizeof(float) * S->P.N);
  if (press) memcpy(press, S->press.data(), sizeof(float) * S->P.N);
  if (cellOf) memcpy(cellOf, S->cellOf.data(), sizeof(int) * S->P.N);
}
void osph_get_accabs(const osph *S, float *out) { memcpy(out, S->accAbs.data(), sizeof(float) * S->P.N); }
void osph_get_clock(const osph *S, float *t, float *tau, long *step) { *t = S->t; *tau = S->tau; *step = S->step; }

/* host dt of one step, :666-669 */
float osph_dt(const osph *S) {
  float dt_try = S->t * S->P.dTau;
  float dt_cfl = S->P.CFL * S->h / (S->P.c0 * (1.0f + 2.0f * S->P.viscAlpha));
  return fminf(dt_try, dt_cfl);
}

/* one sub-step with time step dt: build cells, density/pressure, forces, integrate (:676-701) */
void osph_substep(osph *S, float dt) {
  const tausph_params &P = S->P;
  const int N = P.N, Gx = S->Gx, Gy = S->Gy;
  const float cell = S->cell, h = S->h, mass = S->mass;
  std::fill(S->head.begin(), S->head.end(), -1);
  /* k_build_cells.  Inserting in ascending i leaves each list in DESCENDING particle order — what the block emulator
   * produced for the reference.  -DTAU_SPH_ORACLE_ASCENDING inserts in descending i instead: lists in
   * ascending order, the other legal fp32 summation order (a GPU's atomicExch order is arbitrary). */
#ifdef TAU_SPH_ORACLE_ASCENDING
  for (int i = N - 1; i >= 0; i--) {
#else
  for (int i = 0; i < N; i++) {
#endif
    int gx = grid_c(S->pos[i].x, cell, Gx), gy = grid_c(S->pos[i].y, cell, Gy);
    int c = gy * Gx + gx;
    S->cellOf[i] = c;
    S->next[i] = S->head[c];
    S->head[c] = i;
  }
  const float twoh = 2.f * h, twoh2 = twoh * twoh;
  for (int i = 0; i < N; i++) { /* k_density_pressure_cell */
    f2 xi = S->pos[i];
    int gx = grid_c(xi.x, cell, Gx), gy = grid_c(xi.y, cell, Gy);
    float rho = 0.f;
    for (int oy = -1; oy <= 1; ++oy)
      for (int ox = -1; ox <= 1; ++ox) {
        int cx = gx + ox, cy = gy + oy;
        if ((unsigned)cx >= (unsigned)Gx || (unsigned)cy >= (unsigned)Gy) continue;
        for (int j = S->head[cy * Gx + cx]; j != -1; j = S->next[j]) {
          f2 rij{xi.x - S->pos[j].x, xi.y - S->pos[j].y};
          float r2 = rij.x * rij.x + rij.y * rij.y;
          if (r2 >= twoh2) continue;
          float r = sqrtf(r2);
          rho += mass * W_cubic(r, h);
        }
      }
    float si = logf(fmaxf(rho, 1e-6f));
    S->s[i] = si;
    rho = expf(si);
    float ratio = rho / P.rho0;
    float p = (P.c0 * P.c0) * P.rho0 * (powf(ratio, P.gammaEOS) - 1.0f) / P.gammaEOS;
    S->press[i] = fmaxf(p, 0.0f);
  }
  const float gxa = 0.f, gya = -(P.useGrav ? P.gravity : 0.f);
  for (int i = 0; i < N; i++) { /* k_forces_cell */
    f2 xi = S->pos[i], vi = S->vel[i];
    float rhoi = expf(S->s[i]), pi = S->press[i];
    f2 ai{0.f, 0.f};
    double aabs = 0.0;
    int gx_i = grid_c(xi.x, cell, Gx), gy_i = grid_c(xi.y, cell, Gy);
    for (int oy = -1; oy <= 1; ++oy)
      for (int ox = -1; ox <= 1; ++ox) {
        int cx = gx_i + ox, cy = gy_i + oy;
        if ((unsigned)cx >= (unsigned)Gx || (unsigned)cy >= (unsigned)Gy) continue;
        for (int j = S->head[cy * Gx + cx]; j != -1; j = S->next[j])
          if (j != i) {
            f2 rij{xi.x - S->pos[j].x, xi.y - S->pos[j].y};
            float r2 = rij.x * rij.x + rij.y * rij.y;
            if (r2 >= twoh2 || r2 <= 1e-16f) continue;
            float r = sqrtf(r2);
            f2 gW = gradW_cubic(rij, r, h);
            float rhoj = expf(S->s[j]), pj = S->press[j];
            float common = -mass * (pi / (rhoi * rhoi) + pj / (rhoj * rhoj));
            ai.x += common * gW.x;
            ai.y += common * gW.y;
            aabs += fabs((double)common) * sqrt((double)gW.x * gW.x + (double)gW.y * gW.y);
            /* what a 1e-5 relative error of rho does to this term through p = c0^2 rho0 ((rho/rho0)^g - 1)/g:
             * dp ~ c0^2 rho0 (rho/rho0)^g * 1e-5 — the pressure-scale force sum */
            aabs += (double)mass * (double)(P.c0 * P.c0 * P.rho0) *
                    (pow((double)rhoi / P.rho0, (double)P.gammaEOS) / ((double)rhoi * rhoi) +
                     pow((double)rhoj / P.rho0, (double)P.gammaEOS) / ((double)rhoj * rhoj)) *
                    sqrt((double)gW.x * gW.x + (double)gW.y * gW.y);
            if (P.useVisc) {
              f2 vj = S->vel[j];
              f2 vij{vi.x - vj.x, vi.y - vj.y};
              float dot = vij.x * rij.x + vij.y * rij.y;
              if (dot < 0.f) {
                float mu = (h * dot) / (r2 + 0.01f * h * h);
                float rhoBar = 0.5f * (rhoi + rhoj);
                float Pi_ij = (-P.viscAlpha * P.c0 * mu) / rhoBar;
                ai.x += -mass * Pi_ij * gW.x;
                ai.y += -mass * Pi_ij * gW.y;
                aabs += fabs((double)mass * Pi_ij) * sqrt((double)gW.x * gW.x + (double)gW.y * gW.y);
              }
            }
          }
      }
    if (P.useGrav) { ai.x += gxa; ai.y += gya; }
    S->acc[i] = ai;
    S->accAbs[i] = (float)(aabs + fabs((double)gya));
  }
  for (int i = 0; i < N; i++) { /* k_integrate, :324-355 */
    f2 v = S->vel[i], x = S->pos[i];
    v.x += S->acc[i].x * dt; v.y += S->acc[i].y * dt;
    x.x += v.x * dt; x.y += v.y * dt;
    const float e = 0.2f;
    if (x.x < 0.f) { x.x = 0.f; v.x = -e * v.x; }
    if (x.x > P.boxX) { x.x = P.boxX; v.x = -e * v.x; }
    if (x.y < 0.f) { x.y = 0.f; v.y = -e * v.y; }
    if (x.y > P.boxY) { x.y = P.boxY; v.y = -e * v.y; }
    S->pos[i] = x; S->vel[i] = v;
  }
  if (P.useXSPH && P.xsphEps > 0.f) { /* k_xsph_cell + k_apply_xsph, :274-322, launched :698-704: the lists are the
                                         ones built BEFORE the integrate, positions / velocities are the new ones */
    std::vector<f2> dvel(N);
    for (int i = 0; i < N; i++) {
      f2 xi = S->pos[i], vi = S->vel[i];
      float rhoi = expf(S->s[i]);
      f2 dv{0.f, 0.f};
      int gx_i = grid_c(xi.x, cell, Gx), gy_i = grid_c(xi.y, cell, Gy);
      for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox) {
          int cx = gx_i + ox, cy = gy_i + oy;
          if ((unsigned)cx >= (unsigned)Gx || (unsigned)cy >= (unsigned)Gy) continue;
          for (int j = S->head[cy * Gx + cx]; j != -1; j = S->next[j])
            if (j != i) {
              f2 rij{xi.x - S->pos[j].x, xi.y - S->pos[j].y};
              float r2 = rij.x * rij.x + rij.y * rij.y;
              if (r2 >= twoh2) continue;
              float r = sqrtf(r2);
              float w = W_cubic(r, h);
              float rhoj = expf(S->s[j]);
              float rhoBar = 0.5f * (rhoi + rhoj);
              f2 vij{S->vel[j].x - vi.x, S->vel[j].y - vi.y};
              dv.x += (mass / rhoBar) * vij.x * w;
              dv.y += (mass / rhoBar) * vij.y * w;
            }
        }
      dvel[i] = f2{P.xsphEps * dv.x, P.xsphEps * dv.y};
    }
    for (int i = 0; i < N; i++) { /* the reference parks dvel in acc (:699) */
      S->acc[i] = dvel[i];
      S->vel[i].x += dvel[i].x; S->vel[i].y += dvel[i].y;
    }
  }
  if (P.rain) { /* host bookkeeping :706-716 + k_rain :377-392.  Two drops may pick the same particle; the
                   reference leaves the winner to the hardware, here (and in the engine) the HIGHEST drop index
                   wins — what a launch that retires its threads in order would produce. */
    S->rain_carry += 0.02f * P.N * dt;
    int nspawn = (int)S->rain_carry;
    S->rain_carry -= nspawn;
    unsigned seed = (unsigned)(P.seed + (int)S->step);
    for (int k = 0; k < nspawn; k++) {
      unsigned s = seed ^ ((unsigned)k * 1664525u + 1013904223u);
      s = s * 1664525u + 1013904223u;
      float rx = (s & 0x00FFFFFF) / 16777216.f;
      s = s * 1664525u + 1013904223u;
      float x = rx * (P.boxX * 0.8f) + 0.1f * P.boxX;
      float ry = (s & 0x00FFFFFF) / 16777216.f;
      float y = P.boxY * (0.9f + 0.08f * ry);
      int i = (int)(s % (unsigned)N);
      S->pos[i] = f2{x, y};
      S->vel[i] = f2{0.f, -0.5f * P.c0};
    }
    S->rain_spawned += nspawn;
  }
}

/* k_rasterize, :363-374: particle counts on a W x 2H raster (y flipped), the ncurses view's input */
void osph_rasterize(const osph *S, int W, int H, int *grid2) {
  memset(grid2, 0, sizeof(int) * (size_t)W * 2 * H);
  for (int i = 0; i < S->P.N; i++) {
    f2 p = S->pos[i];
    int cx = (int)(p.x / S->P.boxX * (W - 1));
    int sy = (int)((S->P.boxY - p.y) / S->P.boxY * (2 * H - 1));
    if ((unsigned)cx < (unsigned)W && (unsigned)sy < (unsigned)(2 * H)) grid2[sy * W + cx] += 1;
  }
}
long osph_rain_spawned(const osph *S) { return S->rain_spawned; }

/* one step of the host loop (:665-721): K sub-steps + log-time bookkeeping */
void osph_step(osph *S, int nsteps) {
  for (int n = 0; n < nsteps; n++) {
    int K = (S->P.viscSub > 0 ? S->P.viscSub : 1);
    float dt_eff = osph_dt(S);
    float dt_sub = dt_eff / K;
    float dTau_accum = 0.f;
    for (int k = 0; k < K; ++k) {
      osph_substep(S, dt_sub);
      float dTau_actual = dt_sub / fmaxf(S->t, 1e-9f);
      dTau_accum += dTau_actual;
      S->t = S->P.t0 * expf(S->tau + dTau_accum);
    }
    S->tau += dTau_accum;
    S->step++;
  }
}

} // extern "C"
