/* tau3d — headless driver of the 3D two-temperature hypersonic solver.
 *
 * Stands where the reference's `tau3d` target does (Makefile:81-82, tau_hypersonic_3d_cuda.cu main
 * :1530-1797): same parameters (:1531-1557), same step loop (:1678-1713: log-time clock, k_step,
 * d_tau controller, swap, two steps per frame) — through libtaueng's C-ABI instead of the CUDA
 * launches.  The raylib volume viewer is replaced by --ppm (below); the reference takes no flags, the ones
 * below are additive and default to the reference's behaviour (64^3, quiescent start).
 *   --n N | --nx/--ny/--nz   grid (64)         --frames F   frames of 2 steps (60)
 *   --start 0|1              0 = reference k_init, 1 = developed-flow start
 *   --dump PATH              raw dump of xi,phix,phiy,phiz,lam,zet after the run
 *   --ppm PATH               image of one z-slice of the visualisation field after the run (k_vis +
 *                            slice_to_rgba, :800-905, 1416-1442) — the headless stand-in for the viewer
 *   --vis 0..7               VisMode (:784-794; default 0 = |grad rho|, the reference's start-up mode)
 *   --slice Z  --log  --again G   slice (nz/2), log scaling (key L), opacity gain (keys +/-, 1.0)
 */
#include "tau_cli.h"

int main(int argc, char **argv) {
  int nx = 64, ny = 64, nz = 64, frames = 60, start = 0;
  const int steps_per_frame = 2; /* :1643 */
  const char *dump = NULL, *ppm = NULL;
  int vis = 0, slice = -1, logs = 0;
  double again = 1.0;
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    int v;
    if (!strcmp(a, "--n") && i + 1 < argc) { if (!cli_int(a, argv[++i], &v)) return 1; nx = ny = nz = v; }
    else if (!strcmp(a, "--nx") && i + 1 < argc) { if (!cli_int(a, argv[++i], &nx)) return 1; }
    else if (!strcmp(a, "--ny") && i + 1 < argc) { if (!cli_int(a, argv[++i], &ny)) return 1; }
    else if (!strcmp(a, "--nz") && i + 1 < argc) { if (!cli_int(a, argv[++i], &nz)) return 1; }
    else if (!strcmp(a, "--frames") && i + 1 < argc) { if (!cli_int(a, argv[++i], &frames)) return 1; }
    else if (!strcmp(a, "--start") && i + 1 < argc) { if (!cli_int(a, argv[++i], &start)) return 1; }
    else if (!strcmp(a, "--dump") && i + 1 < argc) dump = argv[++i];
    else if (!strcmp(a, "--ppm") && i + 1 < argc) ppm = argv[++i];
    else if (!strcmp(a, "--vis") && i + 1 < argc) { if (!cli_int(a, argv[++i], &vis)) return 1; }
    else if (!strcmp(a, "--slice") && i + 1 < argc) { if (!cli_int(a, argv[++i], &slice)) return 1; }
    else if (!strcmp(a, "--again") && i + 1 < argc) { if (!cli_double(a, argv[++i], &again)) return 1; }
    else if (!strcmp(a, "--log")) logs = 1;
    else { fprintf(stderr, "Unknown or incomplete argument: %s\n", a); return 1; }
  }
  cli_need_gpu();
  tau3d_params hp;
  tau3d_params_default(&hp, nx, ny, nz);
  tau3d_t *h = NULL;
  TAU_CK(tau3d_create(&h, &hp, 0, nz, 0, NULL));
  TAU_CK(tau3d_init(h, start));
  if (start) { tau3d_clock c = {0.02f, 1e-4f, 0.f, 0.f, 0.f, 0}; TAU_CK(tau3d_set_clock(h, &c)); }

  double t0 = cli_now();
  tau3d_clock c;
  TAU_CK(tau3d_get_clock(h, &c));   /* --frames 0 --dump prints it without ever entering the loop */
  for (int f = 0; f < frames; f++) {
    /* the clock lives on the device: only the frames that print it pay for the read-back (the reference copies
       maxs to the host every step, :1697) */
    const int show = (f % 10 == 0 || f == frames - 1);
    if (show) TAU_CK(tau3d_step(h, steps_per_frame, &c));
    else TAU_CK(tau3d_step_async(h, steps_per_frame));
    if (show) /* the reference's HUD line, :1762-1771 */
      printf("frame %d  step %d  t=%.6g  d_tau=%.4g  dt=%.4g  gain=%.3f  maxs=%.6g\n", f, c.step, c.t, c.d_tau, c.dt,
             c.gain, c.maxs);
  }
  double el = cli_now() - t0;
  double cells = (double)nx * ny * nz * (double)frames * steps_per_frame;
  printf("%d steps on %dx%dx%d in %.3f s: %.3f Gcell-updates/s\n", frames * steps_per_frame, nx, ny, nz, el, cells / el / 1e9);

  if (ppm) { /* the reference's per-frame tail, :1715-1739 */
    float refl = 0.f, mn = 0.f, mx = 0.f;
    uint32_t *px = (uint32_t *)malloc((size_t)nx * ny * sizeof(uint32_t));
    TAU_CK(tau3d_vis(h, vis, NULL));
    TAU_CK(tau3d_outflow_reflection(h, 6, &refl));
    TAU_CK(tau3d_slice_rgba(h, slice < 0 ? nz / 2 : slice, logs, (float)again, px, &mn, &mx));
    if (!cli_write_ppm(ppm, nx, ny, px, 1)) return 1;
    printf("vis mode %d slice %d: min %.6g max %.6g  outflow |dp|=%.6g -> %s\n", vis, slice < 0 ? nz / 2 : slice, mn, mx, refl, ppm);
    free(px);
  }
  if (dump) {
    size_t n = (size_t)nx * ny * nz;
    float *buf[6];
    for (int k = 0; k < 6; k++) buf[k] = (float *)malloc(n * sizeof(float));
    TAU_CK(tau3d_download_state(h, buf));
    char hdr[128];
    snprintf(hdr, sizeof hdr, "tau3d f32 xi,phix,phiy,phiz,lam,zet nx=%d ny=%d nz=%d steps=%d t=%.9g", nx, ny, nz, c.step, c.t);
    const void *arrs[6] = {buf[0], buf[1], buf[2], buf[3], buf[4], buf[5]};
    size_t by[6] = {n * 4, n * 4, n * 4, n * 4, n * 4, n * 4};
    if (!cli_dump(dump, hdr, arrs, by, 6)) return 1;
    for (int k = 0; k < 6; k++) free(buf[k]);
  }
  tau3d_destroy(h);
  return 0;
}
