/* tau_lbm — headless driver of the D2Q9 BGK lattice-Boltzmann solver.
 *
 * Stands where the reference's tau_lbm program does (tau_lbm.cu; built by the line in its header comment, it has
 * no Makefile target): same flags (:155-202), same clamps (:204-207), same loop (:262-265) and the same summary
 * line (:297-299) — through libtaueng's C-ABI.  The ncurses view is replaced by --pgm PATH (additive): the
 * render_kernel speed field (:134-153) with the view's own scale (35 |u| over six shades, :218-232; solids
 * black-on-white as '#').  The reference runs forever with --steps 0 (:261); here 0 means 1000.
 */
#include "tau_cli.h"
#include <getopt.h>

static void usage(const char *prog) { /* :155-170 */
  printf("Usage: %s [options]\n", prog);
  puts("  --nx N              grid x (512)");
  puts("  --ny N              grid y (256)");
  puts("  --tau T             BGK relaxation time > 0.5 (0.56)");
  puts("  --drive A           body-force-like x acceleration (1e-6)");
  puts("  --radius R          cylinder radius in cells (32)");
  puts("  --no-obstacle       disable cylinder; keep channel walls");
  puts("  --steps K           steps (0 = 1000 here; the reference runs forever)");
  puts("  --stride N          render every N steps (4)");
  puts("  --fps N             FPS cap (0 uncapped)");
  puts("  --headless          benchmark mode");
  puts("  --pgm PATH          write the final speed field as a PGM image");
  puts("  -h, --help");
}

int main(int argc, char **argv) {
  taulbm_params P;
  taulbm_params_default(&P);
  int steps = 0, stride = 4;
  const char *pgm = NULL;
  static const struct option opts[] = {{"nx", required_argument, 0, 0}, {"ny", required_argument, 0, 0},
                                       {"tau", required_argument, 0, 0}, {"drive", required_argument, 0, 0},
                                       {"radius", required_argument, 0, 0}, {"no-obstacle", no_argument, 0, 0},
                                       {"steps", required_argument, 0, 0}, {"stride", required_argument, 0, 0},
                                       {"fps", required_argument, 0, 0}, {"headless", no_argument, 0, 0},
                                       {"pgm", required_argument, 0, 0}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
  for (;;) {
    int idx = 0;
    const int c = getopt_long(argc, argv, "h", opts, &idx);
    if (c == -1) break;
    if (c == 'h') { usage(argv[0]); return 0; }
    if (c != 0) { usage(argv[0]); return 1; }
    const char *name = opts[idx].name;
    if (!strcmp(name, "nx")) P.nx = atoi(optarg);
    else if (!strcmp(name, "ny")) P.ny = atoi(optarg);
    else if (!strcmp(name, "tau")) P.tau = (float)atof(optarg);
    else if (!strcmp(name, "drive")) P.drive = (float)atof(optarg);
    else if (!strcmp(name, "radius")) P.obstacle_radius = (float)atof(optarg);
    else if (!strcmp(name, "no-obstacle")) P.obstacle = 0;
    else if (!strcmp(name, "steps")) steps = atoi(optarg);
    else if (!strcmp(name, "stride")) stride = atoi(optarg);
    else if (!strcmp(name, "pgm")) pgm = optarg;
    /* fps, headless: display only */
  }
  if (P.nx < 16) P.nx = 16;           /* :204-207 */
  if (P.ny < 16) P.ny = 16;
  if (P.tau < 0.501f) P.tau = 0.501f;
  if (stride < 1) stride = 1;
  if (steps <= 0) steps = 1000;
  cli_need_gpu();
  taulbm_t *h = NULL;
  TAU_CK(taulbm_create(&h, &P, 0, NULL));
  TAU_CK(taulbm_init(h));
  const size_t cells = (size_t)P.nx * P.ny;
  double t0 = cli_now();
  TAU_CK(taulbm_step(h, steps));
  double total = cli_now() - t0;
  printf("LBM D2Q9: %d steps, %zu cells, %.2f MLUPS\n", steps, cells, total > 0.0 ? (double)cells * steps / (total * 1.0e6) : 0.0);
  if (pgm) {
    float *sp = (float *)malloc(cells * sizeof(float));
    TAU_CK(taulbm_speed(h, sp));
    FILE *f = fopen(pgm, "wb");
    if (!f) { fprintf(stderr, "cannot open %s for writing\n", pgm); return 1; }
    fprintf(f, "P5\n%d %d\n255\n", P.nx, P.ny);
    float vmax = 0.f;
    for (size_t i = 0; i < cells; i++) {
      float v = sp[i];
      if (v > vmax) vmax = v;
      int k = v < 0.f ? 255 : (int)(v * 35.0f * 40.0f);   /* the view's scale: shade = 35 |u|, six shades */
      fputc(k > 200 ? (v < 0.f ? 255 : 200) : k, f);
    }
    fclose(f);
    printf("speed field: max |u| %.6g -> %s\n", vmax, pgm);
    free(sp);
  }
  taulbm_destroy(h);
  return 0;
}
