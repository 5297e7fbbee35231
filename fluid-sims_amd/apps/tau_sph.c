/* tau_sph — headless driver of the 2D WCSPH solver.
 *
 * Stands where the reference's `tau_sph` target does (Makefile:93-94, tau_sph.cu): same flags
 * (:395-416, getopt_long short and long forms), same defaults (:49-85), same step loop (:665-721).
 * Rain is ON by default exactly as in the reference (:76 — it has no flag to turn it off; the additive
 * --no-rain does), with collisions resolved deterministically; --muscl / --xsph_eps switch XSPH on (:480-485).
 * The ncurses view is replaced by --pgm PATH: the k_rasterize particle counts (:363-374) on a
 * --cols x 2*--rows raster (80 x 24 terminal by default) as a binary PGM.  Because the reference's headless
 * mode never terminates (:621, 785-790) an additive --steps N (default 1000) ends the run.  --dump PATH writes
 * pos, vel.
 */
#include "tau_cli.h"
#include <getopt.h>

int main(int argc, char **argv) {
  tausph_params P;
  tausph_params_default(&P, 1 << 16);
  int steps = 1000, stride = 1, cols = 80, rows = 24;
  const char *dump = NULL, *pgm = NULL;
  P.rain = 1; /* :76 */
  static const struct option longopts[] = {
      {"n", required_argument, 0, 'n'}, {"box", required_argument, 0, 'b'}, {"dTau", required_argument, 0, 't'},
      {"rho0", required_argument, 0, 'r'}, {"c0", required_argument, 0, 'c'}, {"gamma", required_argument, 0, 'g'},
      {"CFL", required_argument, 0, 'f'}, {"hMul", required_argument, 0, 'h'}, {"visc", required_argument, 0, 'v'},
      {"gravity", required_argument, 0, 'y'}, {"fps", required_argument, 0, 'p'}, {"fpscap", required_argument, 0, 'F'},
      {"stride", required_argument, 0, 'S'}, {"seed", required_argument, 0, 's'}, {"rain", no_argument, 0, 'R'},
      {"headless", no_argument, 0, 'H'}, {"halfblocks", no_argument, 0, 'B'}, {"visc_substeps", required_argument, 0, 'k'},
      {"muscl", no_argument, 0, 'm'}, {"xsph_eps", required_argument, 0, 'x'},
      {"steps", required_argument, 0, 1000}, {"dump", required_argument, 0, 1001}, {"no-rain", no_argument, 0, 1002},
      {"pgm", required_argument, 0, 1003}, {"cols", required_argument, 0, 1004}, {"rows", required_argument, 0, 1005},
      {0, 0, 0, 0}};
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
    case 'R': P.rain = 1; break;
    case 'k': P.viscSub = atoi(optarg); if (P.viscSub < 1) P.viscSub = 1; break;
    case 'm': P.useXSPH = 1; break;                                                         /* :480-482 */
    case 'x': P.xsphEps = (float)atof(optarg); P.useXSPH = (P.xsphEps > 0.f) || P.useXSPH; break; /* :483-486 */
    case 1000: steps = atoi(optarg); break;
    case 1001: dump = optarg; break;
    case 1002: P.rain = 0; break;
    case 1003: pgm = optarg; break;
    case 1004: cols = atoi(optarg); break;
    case 1005: rows = atoi(optarg); break;
    default: break;
    }
  }
  cli_need_gpu();
  tausph_t *h = NULL;
  TAU_CK(tausph_create(&h, &P, 0, NULL));
  TAU_CK(tausph_reset_particles(h));
  int Gx, Gy; float cell, hh, mass;
  TAU_CK(tausph_grid(h, &Gx, &Gy, &cell, &hh, &mass));
  printf("N=%d box=%gx%g h=%g cell=%g grid=%dx%d mass=%g rain=%s xsph=%s eps=%.2f\n", P.N, P.boxX, P.boxY, hh, cell, Gx, Gy,
         mass, P.rain ? "on" : "off", P.useXSPH ? "on" : "off", P.xsphEps);
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
  if (P.rain) printf("rain: %lld drops\n", (long long)tausph_rain_spawned(h));
  if (pgm) { /* the ncurses view's input, :357-374, 740-760 */
    if (cols < 1 || rows < 1) { fprintf(stderr, "Invalid --cols/--rows: %dx%d\n", cols, rows); return 1; }
    size_t n = (size_t)cols * 2 * rows;
    int32_t *g = (int32_t *)malloc(n * sizeof(int32_t));
    TAU_CK(tausph_rasterize(h, cols, rows, g));
    int32_t mx = 1;
    for (size_t i = 0; i < n; i++) if (g[i] > mx) mx = g[i];
    FILE *f = fopen(pgm, "wb");
    if (!f) { fprintf(stderr, "cannot open %s for writing\n", pgm); return 1; }
    fprintf(f, "P5\n%d %d\n255\n", cols, 2 * rows);
    for (size_t i = 0; i < n; i++) fputc((int)((long long)g[i] * 255 / mx), f);
    fclose(f);
    free(g);
    printf("raster %dx%d max count %d -> %s\n", cols, 2 * rows, (int)mx, pgm);
  }
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
