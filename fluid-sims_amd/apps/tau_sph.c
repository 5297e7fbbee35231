/* tau_sph — headless driver of the 2D WCSPH solver.
 *
 * Stands where the reference's `tau_sph` target does (Makefile:93-94, tau_sph.cu): same flags
 * (:395-416, getopt_long short and long forms), same defaults (:49-85), same step loop (:665-721).
 * ncurses rendering is out of scope (always headless).  Deviations, both documented in DESIGN.md:
 * rain (:377-392) is parsed but not simulated (racy in the reference, out of the hot path), XSPH
 * likewise; and because the reference's headless mode never terminates (:621, 785-790) an additive
 * --steps N (default 1000) ends the run.  --dump PATH writes pos, vel.
 */
#include "tau_cli.h"
#include <getopt.h>

int main(int argc, char **argv) {
  tausph_params P;
  tausph_params_default(&P, 1 << 16);
  int steps = 1000, stride = 1, rain = 0, xsph = 0;
  const char *dump = NULL;
  static const struct option longopts[] = {
      {"n", required_argument, 0, 'n'}, {"box", required_argument, 0, 'b'}, {"dTau", required_argument, 0, 't'},
      {"rho0", required_argument, 0, 'r'}, {"c0", required_argument, 0, 'c'}, {"gamma", required_argument, 0, 'g'},
      {"CFL", required_argument, 0, 'f'}, {"hMul", required_argument, 0, 'h'}, {"visc", required_argument, 0, 'v'},
      {"gravity", required_argument, 0, 'y'}, {"fps", required_argument, 0, 'p'}, {"fpscap", required_argument, 0, 'F'},
      {"stride", required_argument, 0, 'S'}, {"seed", required_argument, 0, 's'}, {"rain", no_argument, 0, 'R'},
      {"headless", no_argument, 0, 'H'}, {"halfblocks", no_argument, 0, 'B'}, {"visc_substeps", required_argument, 0, 'k'},
      {"muscl", no_argument, 0, 'm'}, {"xsph_eps", required_argument, 0, 'x'},
      {"steps", required_argument, 0, 1000}, {"dump", required_argument, 0, 1001}, {0, 0, 0, 0}};
  int c;
  while ((c = getopt_long(argc, argv, "n:b:t:r:c:g:f:h:v:y:p:F:S:s:RHBk:mx:", longopts, NULL)) != -1) {
    switch (c) {
    case 'n': P.N = atoi(optarg); break;
    case 'b': sscanf(optarg, "%fx%f", &P.boxX, &P.boxY); break;
    case 't': P.dTau = (float)atof(optarg); break;
    case 'r': P.rho0 = (float)atof(optarg); break;
    case 'c': P.c0 = (float)atof(optarg); break;
    case 'g': P.gammaEOS = (float)atof(optarg); break;
    case 'f': P.CFL = (float)atof(optarg); break;
    case 'h': P.hMul = (float)atof(optarg); break;
    case 'v': P.viscAlpha = (float)atof(optarg); break;
    case 'y': P.gravity = (float)atof(optarg); P.useGrav = (P.gravity != 0.f); break;
    case 'p': case 'F': case 'H': case 'B': break;   /* display only */
    case 'S': stride = atoi(optarg); if (stride < 1) stride = 1; break;
    case 's': P.seed = atoi(optarg); break;
    case 'R': rain = 1; break;
    case 'k': P.viscSub = atoi(optarg); if (P.viscSub < 1) P.viscSub = 1; break;
    case 'm': xsph = 1; break;
    case 'x': xsph = xsph || atof(optarg) > 0; break;
    case 1000: steps = atoi(optarg); break;
    case 1001: dump = optarg; break;
    default: break;
    }
  }
  if (rain) fprintf(stderr, "note: --rain is not simulated by this engine (see DESIGN.md)\n");
  if (xsph) fprintf(stderr, "note: XSPH smoothing is not simulated by this engine (see DESIGN.md)\n");
  cli_need_gpu();
  tausph_t *h = NULL;
  TAU_CK(tausph_create(&h, &P, 0, NULL));
  TAU_CK(tausph_reset_particles(h));
  int Gx, Gy; float cell, hh, mass;
  TAU_CK(tausph_grid(h, &Gx, &Gy, &cell, &hh, &mass));
  printf("N=%d box=%gx%g h=%g cell=%g grid=%dx%d mass=%g\n", P.N, P.boxX, P.boxY, hh, cell, Gx, Gy, mass);
  double t0 = cli_now();
  for (int step = 0; step < steps; step++) {
    TAU_CK(tausph_step_async(h, 1));
    if (step % stride == 0 && step % (100 * stride) == 0) { /* :785-790 */
      float t, tau; int64_t s;
      TAU_CK(tausph_sync(h));
      TAU_CK(tausph_get_clock(h, &t, &tau, &s));
      printf("step %d  t=%.3g tau=%.3g\n", step, t, tau);
      fflush(stdout);
    }
  }
  TAU_CK(tausph_sync(h));
  double el = cli_now() - t0;
  printf("%d steps (x%d sub-steps) of %d particles in %.3f s: %.3f Mparticle-updates/s\n", steps, P.viscSub, P.N, el,
         (double)P.N * steps * P.viscSub / el / 1e6);
  if (dump) {
    size_t n = (size_t)P.N;
    float *pos = (float *)malloc(n * 8), *vel = (float *)malloc(n * 8);
    TAU_CK(tausph_download(h, pos, vel, NULL, NULL, NULL, NULL));
    char hdr[128];
    snprintf(hdr, sizeof hdr, "tau_sph f32 pos(x,y),vel(x,y) N=%d steps=%d", P.N, steps);
    const void *arrs[2] = {pos, vel};
    size_t by[2] = {n * 8, n * 8};
    if (!cli_dump(dump, hdr, arrs, by, 2)) return 1;
  }
  tausph_destroy(h);
  return 0;
}
