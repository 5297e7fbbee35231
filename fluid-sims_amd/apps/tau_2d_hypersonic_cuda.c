/* tau_2d_hypersonic_cuda — headless driver of the 2D Euler solver (GPU scheme).
 *
 * Stands where the reference's target of the same name does (Makefile:84-85,
 * tau_hypersonic_cuda.cu): same flags, same validation rules and messages (:1458-1639), same step
 * loop (:1833-1888) — one fused HIP kernel per step behind tauh2_step.  The raylib window is replaced by
 * a headless image: --ppm PATH writes the final frame in view mode --view 0..6 (the reference's keys 1-7,
 * :1896-1909; render kernels :1178-1334).  Additive: --W / --H (compile-time 8192 x 1024 in the reference),
 * --frames F (frames of --steps-per-frame steps, default 30), --dump PATH.
 * --tile-bx / --tile-by are parsed and validated for compatibility; the engine's tile is fixed.
 */
#include "tau_cli.h"

static const double kPi = 3.14159265358979323846;

static void print_usage(const char *argv0) { /* :1448-1456 */
  fprintf(stderr,
          "Usage: %s [--mach M] [--gamma G] [--cfl C] [--visc-nu NU]\n"
          "          [--visc-rho MU] [--visc-e K] [--steps-per-frame N]\n"
          "          [--geom-x0 X0] [--geom-cy CY] [--geom-rb RB]\n"
          "          [--geom-rn RN] [--geom-theta THETA]\n"
          "          [--tile-bx BX] [--tile-by BY]\n"
          "          [--W W] [--H H] [--frames F] [--dump PATH] [--ppm PATH] [--view 0..6]\n",
          argv0);
}

int main(int argc, char **argv) {
  int W = 8192, H = 1024, frames = 30, steps_per_frame = 2, tile_bx = -1, tile_by = -1, bx_set = 0, by_set = 0;
  const char *dump = NULL, *ppm = NULL;
  int view = 0;
  /* first pass: the grid (the geometry defaults depend on H, :1403-1406) */
  for (int i = 1; i + 1 < argc; i++) {
    if (!strcmp(argv[i], "--W")) { if (!cli_int("--W", argv[i + 1], &W)) return 1; }
    if (!strcmp(argv[i], "--H")) { if (!cli_int("--H", argv[i + 1], &H)) return 1; }
  }
  tauh2_params c;
  tauh2_params_default(&c, W, H);
  struct { const char *name; double *dst; } dflags[] = {
      {"--mach", &c.mach}, {"--gamma", &c.gamma}, {"--cfl", &c.cfl}, {"--visc-nu", &c.visc_nu},
      {"--visc-rho", &c.visc_rho}, {"--visc-e", &c.visc_e}, {"--geom-x0", &c.geom_x0}, {"--geom-cy", &c.geom_cy},
      {"--geom-rb", &c.geom_rb}, {"--geom-rn", &c.geom_rn}, {"--geom-theta", &c.geom_theta}};
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    int done = 0;
    if (i + 1 < argc) {
      for (size_t k = 0; k < sizeof dflags / sizeof dflags[0]; k++)
        if (!strcmp(a, dflags[k].name)) { if (!cli_double(a, argv[++i], dflags[k].dst)) { print_usage(argv[0]); return 1; } done = 1; break; }
      if (done) continue;
      if (!strcmp(a, "--steps-per-frame")) { if (!cli_int(a, argv[++i], &steps_per_frame)) { print_usage(argv[0]); return 1; } continue; }
      if (!strcmp(a, "--tile-bx")) { if (!cli_int(a, argv[++i], &tile_bx)) { print_usage(argv[0]); return 1; } bx_set = 1; continue; }
      if (!strcmp(a, "--tile-by")) { if (!cli_int(a, argv[++i], &tile_by)) { print_usage(argv[0]); return 1; } by_set = 1; continue; }
      if (!strcmp(a, "--W") || !strcmp(a, "--H")) { i++; continue; }
      if (!strcmp(a, "--frames")) { if (!cli_int(a, argv[++i], &frames)) return 1; continue; }
      if (!strcmp(a, "--dump")) { dump = argv[++i]; continue; }
      if (!strcmp(a, "--ppm")) { ppm = argv[++i]; continue; }
      if (!strcmp(a, "--view")) { if (!cli_int(a, argv[++i], &view)) return 1; continue; }
    }
    fprintf(stderr, "Unknown or incomplete argument: %s\n", a);
    print_usage(argv[0]);
    return 1;
  }
  /* validation, :1538-1636 — same order, same messages */
#define BAD(...) do { fprintf(stderr, __VA_ARGS__); print_usage(argv[0]); return 1; } while (0)
  if (c.gamma <= 1.0) BAD("Invalid --gamma: %.8g (must be > 1).\n", c.gamma);
  if (c.cfl <= 0.0) BAD("Invalid --cfl: %.8g (must be > 0).\n", c.cfl);
  if (c.visc_nu < 0.0) BAD("Invalid --visc-nu: %.8g (must be >= 0).\n", c.visc_nu);
  if (c.visc_rho < 0.0) BAD("Invalid --visc-rho: %.8g (must be >= 0).\n", c.visc_rho);
  if (c.visc_e < 0.0) BAD("Invalid --visc-e: %.8g (must be >= 0).\n", c.visc_e);
  if (c.mach <= 0.0) BAD("Invalid --mach: %.8g (must be > 0).\n", c.mach);
  if (steps_per_frame <= 0 || steps_per_frame > 1024) BAD("Invalid --steps-per-frame: %d (must be in [1, %d]).\n", steps_per_frame, 1024);
  if (!isfinite(c.geom_x0)) BAD("Invalid --geom-x0: %.8g (must be finite).\n", c.geom_x0);
  if (!isfinite(c.geom_cy)) BAD("Invalid --geom-cy: %.8g (must be finite).\n", c.geom_cy);
  if (c.geom_rb <= 0.0) BAD("Invalid --geom-rb: %.8g (must be > 0).\n", c.geom_rb);
  if (c.geom_rn <= 0.0) BAD("Invalid --geom-rn: %.8g (must be > 0).\n", c.geom_rn);
  if (c.geom_theta <= 0.0 || c.geom_theta >= 0.5 * kPi) BAD("Invalid --geom-theta: %.8g (must be in (0, pi/2)).\n", c.geom_theta);
  {
    const double st = sin(c.geom_theta), ct = cos(c.geom_theta), tt = tan(c.geom_theta);
    const double xt = c.geom_rn * (1.0 - st), rt = c.geom_rn * ct;
    if (c.geom_rb < rt)
      BAD("Invalid geometry: --geom-rb %.8g is smaller than the tangent radius %.8g implied by --geom-rn %.8g and "
          "--geom-theta %.8g. Require geom-rb >= geom-rn*cos(theta).\n", c.geom_rb, rt, c.geom_rn, c.geom_theta);
    if (!isfinite(tt) || tt <= 0.0)
      BAD("Invalid geometry: tan(theta)=%.8g for --geom-theta %.8g must be finite and positive.\n", tt, c.geom_theta);
    const double xb = xt + (c.geom_rb - rt) / tt;
    if (!isfinite(xb))
      BAD("Invalid geometry: computed xb is non-finite (xb=%.8g) from --geom-rb %.8g --geom-rn %.8g --geom-theta %.8g.\n",
          xb, c.geom_rb, c.geom_rn, c.geom_theta);
    if (xb < xt)
      BAD("Invalid geometry: computed xb %.8g is behind cone tangent point xt %.8g. Increase --geom-rb or reduce "
          "--geom-rn/--geom-theta.\n", xb, xt);
  }
  if ((bx_set && tile_bx <= 0) || (by_set && tile_by <= 0))
    BAD("Invalid tile dimensions: --tile-bx and --tile-by must be positive when provided.\n");
  if (W < 8 || H < 8) BAD("Invalid --W/--H: %dx%d (must be at least 8x8).\n", W, H);

  cli_need_gpu();
  /* print_config, :1687-1709 */
  printf("config: %dx%d mach=%.6g gamma=%.6g cfl=%.6g visc(nu,rho,e)=(%.4g,%.4g,%.4g) steps/frame=%d\n", W, H, c.mach,
         c.gamma, c.cfl, c.visc_nu, c.visc_rho, c.visc_e, steps_per_frame);
  printf("geometry: x0=%.6g cy=%.6g Rb=%.6g Rn=%.6g theta=%.6g\n", c.geom_x0, c.geom_cy, c.geom_rb, c.geom_rn, c.geom_theta);
  tauh2_t *h = NULL;
  TAU_CK(tauh2_create(&h, &c, 0, NULL));
  TAU_CK(tauh2_init(h));
  double t0 = cli_now(), t = 0;
  for (int f = 0; f < frames; f++) {
    if (!(f % 10 == 0 || f == frames - 1)) { TAU_CK(tauh2_step_async(h, steps_per_frame)); continue; }  /* dt stays on the device */
    TAU_CK(tauh2_step(h, steps_per_frame, &t));
    {
      double dt, maxs; int step;
      TAU_CK(tauh2_get_time(h, &t, &dt, &maxs, &step));
      printf("frame %d  step %d  t=%.6g  dt=%.4g  maxs=%.6g\n", f, step, t, dt, maxs);
    }
  }
  double el = cli_now() - t0;
  printf("%d steps on %dx%d in %.3f s: %.3f Gcell-updates/s\n", frames * steps_per_frame, W, H, el,
         (double)W * H * frames * steps_per_frame / el / 1e9);
  if (ppm) {
    uint32_t *px = (uint32_t *)malloc((size_t)W * H * sizeof(uint32_t));
    double vmin, vmax;
    TAU_CK(tauh2_render(h, view, px, NULL, &vmin, &vmax));
    if (!cli_write_ppm(ppm, W, H, px, 0)) return 1;
    printf("view mode %d: range [%.6g, %.6g] -> %s\n", view, vmin, vmax, ppm);
    free(px);
  }
  if (dump) {
    size_t n = (size_t)W * H;
    float *b[4];
    for (int k = 0; k < 4; k++) b[k] = (float *)malloc(n * 4);
    uint8_t *m = (uint8_t *)malloc(n);
    TAU_CK(tauh2_download(h, b, m));
    char hdr[160];
    snprintf(hdr, sizeof hdr, "tau2d f32 rho,mx,my,E + u8 mask W=%d H=%d t=%.9g", W, H, t);
    const void *arrs[5] = {b[0], b[1], b[2], b[3], m};
    size_t by[5] = {n * 4, n * 4, n * 4, n * 4, n};
    if (!cli_dump(dump, hdr, arrs, by, 5)) return 1;
  }
  tauh2_destroy(h);
  return 0;
}
