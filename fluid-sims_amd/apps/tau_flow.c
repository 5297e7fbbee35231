/* tau_burgers / tau_sw — headless drivers of the 2D viscous Burgers and shallow-water solvers.
 *
 * Stand where the reference's targets do (Makefile:75-76, 90-91; tau_burgers.cu, tau_shallow_water.cu):
 * same long options (tau_burgers.cu:143-240; tau_shallow_water.cu:147-172 + its parser), same defaults,
 * same loop (do_step + tau += dtau, t *= exp(dtau)), same headless report (steps, frames, FPS — the
 * reference's only throughput print, tau_burgers.cu:790-820).  ncurses rendering is out of scope, so the
 * programs always run the headless branch; --steps 0 means 2000 steps as in that branch (:797).
 * Built twice: -DTAU_SW selects the shallow-water option table.  Additive: --dump PATH.
 */
#include "tau_cli.h"
#include <getopt.h>

#ifdef TAU_SW
#define KIND 1
#define PROG "tau_sw"
#else
#define KIND 0
#define PROG "tau_burgers"
#endif

/* the CLI surface of the two programs (tau_burgers.cu:110-141, tau_shallow_water.cu:118-144): option names, meanings, defaults */
static void usage(const char *prog) {
  printf("Usage: %s [options]\n", prog);
#ifdef TAU_SW
  static const char *const o[] = {
      "--nx N        grid cells in x (256)", "--ny N        grid cells in y (256)", "--dx M        cell size x meters (1000)",
      "--dy M        cell size y meters (1000)", "--g G         gravity (9.81)", "--f0 F        coriolis f (0 or 1e-4)",
      "--nu NU       eddy viscosity on u,v m^2/s (0)", "--H0 H        mean depth (1000)",
      "--amp A       initial Gaussian bump amplitude (1)", "--bsig S      bump sigma in cells (2)", "--CFL C       CFL number (0.45)",
      "--steps K     number of steps (0 = forever)", "--tau0 T      initial log-time (0)", "--t0 T0       physical seconds at tau0 (1)",
      "--dtau D      log-time step (1e-3)", "--headless    run without UI (benchmark)", "--stride N    render every Nth step (1)",
      "--fps N       limit FPS to N (0 = uncapped)", "--offx X      bump center x shift in cells (0)",
      "--offy Y      bump center y shift in cells (0)", "--asym A      small dipole modulation (0.0)",
      "--swirl O     angular speed [1/s] for initial vortex (0)", "--rc R        core radius in cells for vortex (5)",
      "--dump PATH   (this build) write the final sigma, u, v as raw fp32", "-h, --help    show this help"};
#else
  static const char *const o[] = {
      "--nx N           grid x (512)", "--ny N           grid y (512)", "--dx M           dx (1)", "--dy M           dy (1)",
      "--nu NU          viscosity (0.01)", "--u0 U0          scale for u = u0*sinh(phi) (1)", "--amp A          init amplitude (1)",
      "--bsig S         init sigma (cells) (16)", "--swirl O        init swirl rate (1)", "--rc R           core radius (cells) (40)",
      "--offx X         center x shift (0)", "--offy Y         center y shift (0)", "--asym A         dipole modulation (0)",
      "--CFL C          CFL (0.45)", "--steps K        steps (0 forever)", "--tau0 T         initial tau (0)", "--t0 T0          t at tau0 (1)",
      "--dtau D         log-time step (1e-3)", "--headless       benchmark mode", "--stride N       render every Nth step (5)",
      "--fps N          FPS cap (0 uncapped)", "--halfblocks     high-res terminal renderer", "--muscl          MUSCL/minmod reconstruction",
      "--visc_substeps K  viscosity sub-iterations (1)", "--colehopf       enable 1-D Cole-Hopf validation",
      "--ck M           Cole-Hopf mode number (4)", "--ca A           Cole-Hopf amplitude |A|<1 (0.5)",
      "--dump PATH      (this build) write the final phi_u, phi_v as raw fp32", "-h, --help"};
#endif
  for (size_t i = 0; i < sizeof o / sizeof o[0]; i++) printf("  %s\n", o[i]);
}

int main(int argc, char **argv) {
  tauflow_params P;
  tauflow_params_default(&P, KIND, 512, 512);
  int steps = 0, stride = 5, colehopf = 0;
  const char *dump = NULL;
  static const struct option lo[] = {
      {"nx", required_argument, 0, 0}, {"ny", required_argument, 0, 0}, {"dx", required_argument, 0, 0},
      {"dy", required_argument, 0, 0}, {"nu", required_argument, 0, 0}, {"CFL", required_argument, 0, 0},
      {"steps", required_argument, 0, 0}, {"tau0", required_argument, 0, 0}, {"t0", required_argument, 0, 0},
      {"dtau", required_argument, 0, 0}, {"offx", required_argument, 0, 0}, {"offy", required_argument, 0, 0},
      {"asym", required_argument, 0, 0}, {"swirl", required_argument, 0, 0}, {"headless", no_argument, 0, 'H'},
      {"stride", required_argument, 0, 'r'}, {"fps", required_argument, 0, 'f'}, {"halfblocks", no_argument, 0, 0},
      {"dump", required_argument, 0, 0}, {"help", no_argument, 0, 'h'},
#ifdef TAU_SW
      {"g", required_argument, 0, 0}, {"f0", required_argument, 0, 0}, {"H0", required_argument, 0, 0},
      {"bump", required_argument, 0, 0}, {"bsig", required_argument, 0, 0}, {"rc", required_argument, 0, 0},
#else
      {"u0", required_argument, 0, 0}, {"amp", required_argument, 0, 0}, {"bsig", required_argument, 0, 0},
      {"rc", required_argument, 0, 0}, {"muscl", no_argument, 0, 0}, {"visc_substeps", required_argument, 0, 0},
      {"colehopf", no_argument, 0, 0}, {"ck", required_argument, 0, 0}, {"ca", required_argument, 0, 0},
#endif
      {0, 0, 0, 0}};
  int idx = 0, c;
  while ((c = getopt_long(argc, argv, "Hr:f:h", lo, &idx)) != -1) {
    if (c == 'h') { usage(argv[0]); return 0; }
    if (c == 'H' || c == 'f') continue;
    if (c == 'r') { stride = atoi(optarg); continue; }
    if (c != 0) continue;   /* an unknown option: getopt has said so on stderr; the reference goes on (tau_burgers.cu:174-245) */
    const char *n = lo[idx].name;
    if (!strcmp(n, "nx")) P.nx = atoi(optarg);
    else if (!strcmp(n, "ny")) P.ny = atoi(optarg);
    else if (!strcmp(n, "dx")) P.dx = (float)atof(optarg);
    else if (!strcmp(n, "dy")) P.dy = (float)atof(optarg);
    else if (!strcmp(n, "nu")) P.nu = (float)atof(optarg);
    else if (!strcmp(n, "CFL")) P.CFL = (float)atof(optarg);
    else if (!strcmp(n, "steps")) steps = atoi(optarg);
    else if (!strcmp(n, "tau0")) P.tau0 = (float)atof(optarg);
    else if (!strcmp(n, "t0")) P.t0 = (float)atof(optarg);
    else if (!strcmp(n, "dtau")) P.dtau = (float)atof(optarg);
    else if (!strcmp(n, "offx")) P.offx = (float)atof(optarg);
    else if (!strcmp(n, "offy")) P.offy = (float)atof(optarg);
    else if (!strcmp(n, "asym")) P.asym = (float)atof(optarg);
    else if (!strcmp(n, "swirl")) P.swirl = (float)atof(optarg);
    else if (!strcmp(n, "bsig")) P.bsig = (float)atof(optarg);
    else if (!strcmp(n, "rc")) P.rc = (float)atof(optarg);
    else if (!strcmp(n, "dump")) dump = optarg;
    else if (!strcmp(n, "halfblocks")) { /* display only */ }
#ifdef TAU_SW
    else if (!strcmp(n, "g")) P.g = (float)atof(optarg);
    else if (!strcmp(n, "f0")) { /* parsed and shown by the reference, never used by a kernel */ }
    else if (!strcmp(n, "H0")) P.H0 = (float)atof(optarg);
    else if (!strcmp(n, "bump")) P.amp = (float)atof(optarg);
#else
    else if (!strcmp(n, "u0")) P.u0 = (float)atof(optarg);
    else if (!strcmp(n, "amp")) P.amp = (float)atof(optarg);
    else if (!strcmp(n, "muscl")) P.muscl = 1;
    else if (!strcmp(n, "visc_substeps")) P.visc_substeps = atoi(optarg);
    else if (!strcmp(n, "colehopf")) { colehopf = 1; P.oneD = 1; }
    else if (!strcmp(n, "ck")) P.ck = atoi(optarg);
    else if (!strcmp(n, "ca")) P.ca = (float)atof(optarg);
#endif
  }
  if (stride < 1) stride = 1;
  if (colehopf) P.ny = 1; /* tau_burgers.cu:654-655 */
  cli_need_gpu();
  tauflow_t *h = NULL;
  TAU_CK(tauflow_create(&h, &P, KIND, 0, NULL));
  TAU_CK(tauflow_init(h));
  const int nsteps = steps ? steps : 2000;
  int frames = 0;
  double elapsed = 0.0, t0 = cli_now(), gpu_ms = 0.0;
  TAU_CK(tauflow_timer_start(h));
  for (int s = 0; s < nsteps; s++) {
    TAU_CK(tauflow_step_async(h, 1));
    if (colehopf) { float dt; TAU_CK(tauflow_get_clock(h, NULL, NULL, &dt, NULL, NULL)); elapsed += dt; }
    if (s % stride == 0) frames++;
  }
  TAU_CK(tauflow_timer_stop(h, &gpu_ms));
  TAU_CK(tauflow_sync(h));
  double secs = cli_now() - t0, gsecs = gpu_ms * 1e-3;
  float t, tau, dt, w; int64_t st;
  TAU_CK(tauflow_get_clock(h, &t, &tau, &dt, &w, &st));
  /* the reference's headless summary, line for line (tau_burgers.cu:812-818, tau_shallow_water.cu:774-780); the GPU line is the
   * device time of the loop's launches from events on the handle's stream (tauflow_timer_*), as the reference's cudaEvent pair */
#ifdef TAU_SW
  printf("Headless benchmark (stride=%d):\n  Simulated steps: %d\n  Wall-clock: %d frames in %.3f s -> %.1f FPS\n  GPU only:   %d frames in %.3f s -> %.1f FPS\n",
         stride, nsteps, frames, secs, frames > 0 ? frames / secs : 0.0, frames, gsecs, frames > 0 && gsecs > 0 ? frames / gsecs : 0.0);
#else
  printf("Headless (stride=%d):\n  Steps: %d\n  Wall:  %d frames in %.3f s -> %.1f FPS\n  GPU:   %d frames in %.3f s -> %.1f FPS\n", stride, nsteps,
         frames, secs, frames > 0 ? frames / secs : 0.0, frames, gsecs, frames > 0 && gsecs > 0 ? frames / gsecs : 0.0);
#endif
  printf("  %s %dx%d: t=%.6g tau=%.6g dt=%.4g wavespeed=%.6g  %.3f Gcell-updates/s\n", PROG, P.nx, P.ny, t, tau, dt, w,
         (double)P.nx * P.ny * nsteps / secs / 1e9);
  if (colehopf) {
    double rel;
    TAU_CK(tauflow_colehopf_relL2(h, (float)elapsed, &rel));
    printf("  Cole-Hopf relative L2 error at elapsed t=%.6g: %.3e\n", elapsed, rel);
  }
  if (dump) {
    size_t n = (size_t)P.nx * P.ny;
    float *f[3] = {(float *)malloc(n * 4), (float *)malloc(n * 4), (float *)malloc(n * 4)};
    TAU_CK(tauflow_download(h, f));
    char hdr[128];
    snprintf(hdr, sizeof hdr, PROG " f32 %s nx=%d ny=%d steps=%d", KIND ? "sigma,u,v" : "phi_u,phi_v", P.nx, P.ny, nsteps);
    const void *arrs[3] = {f[0], f[1], f[2]};
    size_t by[3] = {n * 4, n * 4, n * 4};
    if (!cli_dump(dump, hdr, arrs, by, KIND ? 3 : 2)) return 1;
  }
  tauflow_destroy(h);
  return 0;
}
