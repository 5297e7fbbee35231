/* tgs — headless driver of the Gray-Scott reaction-diffusion solver.
 *
 * Stands where the reference's `tgs` target does (Makefile:78-79, tau_gray_scott.cu): same flags
 * (:83-92), same defaults (:43-61), same loop (:321-329).  The ncurses renderer is out of scope, so
 * the program always runs headless (nx = ny = 128 when not given, :293-295); with --steps 0 the
 * reference runs forever — here 0 selects 1000 steps so the program terminates.
 */
#include "tau_cli.h"
#include <getopt.h>

static void usage(const char *prog) { /* :65-81 */
  printf("Usage: %s [options]\n", prog);
  puts("  --nx N        grid cells in x (128 when headless)");
  puts("  --ny N        grid cells in y (128 when headless)");
  puts("  --dx DX       cell size (1)");
  puts("  --dt DT       time step (1)");
  puts("  --Du D        diffusion coefficient for U (0.2)");
  puts("  --Dv D        diffusion coefficient for V (0.1)");
  puts("  --F F         feed rate (0.03)");
  puts("  --k K         kill rate (0.06)");
  puts("  --steps K     number of steps (0 = 1000 here; the reference runs forever)");
  puts("  --headless    accepted (this build is always headless)");
  puts("  --stride N    accepted, unused without a renderer");
  puts("  --fps N       accepted, unused");
  puts("  --seed S      RNG seed for initial pattern (1337)");
  puts("  --halfblocks  accepted, unused");
  puts("  --dump PATH   raw dump of u,v after the run (additive)");
  puts("  -h, --help    show this help message");
}

int main(int argc, char **argv) {
  taugs_params P;
  taugs_params_default(&P, 0, 0);
  int steps = 0;
  unsigned seed = 1337;
  const char *dump = NULL;
  /* getopt_long with the reference's own table (:84-104): `--nx 128`, `--nx=128` and unambiguous abbreviations all parse,
     unknown options get getopt's message and are skipped — as there.  --dump is additive. */
  static const struct option long_opts[] = {
      {"nx", required_argument, 0, 0},     {"ny", required_argument, 0, 0},    {"dx", required_argument, 0, 0},
      {"dt", required_argument, 0, 0},     {"Du", required_argument, 0, 0},    {"Dv", required_argument, 0, 0},
      {"F", required_argument, 0, 0},      {"k", required_argument, 0, 0},     {"steps", required_argument, 0, 0},
      {"headless", no_argument, 0, 0},     {"stride", required_argument, 0, 0}, {"fps", required_argument, 0, 0},
      {"seed", required_argument, 0, 0},   {"halfblocks", no_argument, 0, 0},  {"dump", required_argument, 0, 0},
      {"help", no_argument, 0, 'h'},       {0, 0, 0, 0}};
  for (;;) {
    int idx = 0;
    const int c = getopt_long(argc, argv, "h", long_opts, &idx);
    if (c == -1) break;
    if (c == 'h') { usage(argv[0]); return 0; }
    if (c) continue;
    const char *opt = long_opts[idx].name;    /* atoi / atof as the reference parses them, :106-127 */
    if (!strcmp(opt, "nx")) P.nx = atoi(optarg);
    else if (!strcmp(opt, "ny")) P.ny = atoi(optarg);
    else if (!strcmp(opt, "dx")) P.dx = (float)atof(optarg);
    else if (!strcmp(opt, "dt")) P.dt = (float)atof(optarg);
    else if (!strcmp(opt, "Du")) P.Du = (float)atof(optarg);
    else if (!strcmp(opt, "Dv")) P.Dv = (float)atof(optarg);
    else if (!strcmp(opt, "F")) P.feed = (float)atof(optarg);
    else if (!strcmp(opt, "k")) P.kill = (float)atof(optarg);
    else if (!strcmp(opt, "steps")) steps = atoi(optarg);
    else if (!strcmp(opt, "seed")) seed = (unsigned)strtoul(optarg, NULL, 10);
    else if (!strcmp(opt, "dump")) dump = optarg;
    /* headless, halfblocks, stride, fps: display only */
  }
  if (P.nx == 0) P.nx = 128;
  if (P.ny == 0) P.ny = 128;
  if (steps <= 0) steps = 1000;
  cli_need_gpu();
  taugs_t *h = NULL;
  TAU_CK(taugs_create(&h, &P, 0, NULL));
  TAU_CK(taugs_init_pattern(h, seed));
  double t0 = cli_now();
  TAU_CK(taugs_step(h, steps));
  double el = cli_now() - t0;
  size_t n = (size_t)P.nx * P.ny;
  float *u = (float *)malloc(n * 4), *v = (float *)malloc(n * 4);
  TAU_CK(taugs_download(h, u, v));
  double su = 0, sv = 0;
  for (size_t i = 0; i < n; i++) { su += u[i]; sv += v[i]; }
  printf("%d steps on %dx%d: %.3f ms/step, %.2f Gcell-updates/s, sum u = %.9g, sum v = %.9g\n", steps, P.nx, P.ny,
         el / steps * 1e3, (double)n * steps / el / 1e9, su, sv);
  if (dump) {
    char hdr[128];
    snprintf(hdr, sizeof hdr, "tgs f32 u,v nx=%d ny=%d steps=%d", P.nx, P.ny, steps);
    const void *arrs[2] = {u, v};
    size_t by[2] = {n * 4, n * 4};
    if (!cli_dump(dump, hdr, arrs, by, 2)) return 1;
  }
  free(u); free(v);
  taugs_destroy(h);
  return 0;
}
