/* tgs — headless driver of the Gray-Scott reaction-diffusion solver.
 *
 * Stands where the reference's `tgs` target does (Makefile:78-79, tau_gray_scott.cu): same flags
 * (:83-92), same defaults (:43-61), same loop (:321-329).  The ncurses renderer is out of scope, so
 * the program always runs headless (nx = ny = 128 when not given, :293-295); with --steps 0 the
 * reference runs forever — here 0 selects 1000 steps so the program terminates.
 */
#include "tau_cli.h"
#include <getopt.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

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
  puts("  --gpus N      row-slab ring over N GPUs of this node, one forked process per rank (additive; bit-identical to --gpus 1)");
  puts("  --transport rccl|host   how the halo rows travel (rccl: one device per rank; host: ranks may share a device)");
  puts("  --halo H      halo rows either side of a slab = steps between exchanges (4: one exchange per fused pass)");
  puts("  -h, --help    show this help message");
}

/* one rank of the row-slab ring: rows [y0, y0 + nyl) + `halo` rows either side, cut out of the global initial pattern */
static int run_rank(const taugs_params *G, int steps, unsigned seed, const char *dump, int rank, int world, int transport, int halo,
                    const char *rv, uint64_t key, const char *gather) {
  cli_need_gpu();
  int ndev = 1, y0 = 0, nyl = 0;
  if (tau_device_count(&ndev) || ndev < 1) { fprintf(stderr, "taueng error: %s\n", tau_last_error()); return 1; }
  if (transport == TAU3D_RING_RCCL && world > 1 && ndev < world) {
    fprintf(stderr, "tgs --gpus %d needs %d devices (one per rank over RCCL), this node shows %d; --transport host lets ranks share a device\n", world, world, ndev);
    return 1;
  }
  TAU_CK(taurow_bounds(G->ny, world, rank, &y0, &nyl));
  taugs_params P = *G;
  P.ny = nyl + 2 * halo;
  const size_t nx = (size_t)G->nx, n = nx * (size_t)G->ny, nl = nx * (size_t)P.ny;
  float *gu = (float *)malloc(n * 4), *gv = (float *)malloc(n * 4), *u = (float *)malloc(nl * 4), *v = (float *)malloc(nl * 4);
  TAU_CK(taugs_pattern_host(G->nx, G->ny, seed, gu, gv));
  for (int j = 0; j < P.ny; j++) {
    const int gy = ((y0 - halo + j) % G->ny + G->ny) % G->ny;
    memcpy(u + (size_t)j * nx, gu + (size_t)gy * nx, nx * 4);
    memcpy(v + (size_t)j * nx, gv + (size_t)gy * nx, nx * 4);
  }
  taugs_t *h = NULL;
  taurow_ring_t *r = NULL;
  TAU_CK(taugs_create(&h, &P, rank % ndev, NULL));
  TAU_CK(taugs_upload(h, u, v));
  TAU_CK(taugs_ring_create(&r, h, halo, rank, world, transport, rv, key));
  if (rank == 0) {
    int ver = 0, ranks = 0;
    TAU_CK(taurow_ring_info(r, NULL, NULL, NULL, &ver, &ranks));
    if (transport == TAU3D_RING_RCCL) printf("row ring: %d ranks, RCCL %d, communicator of %d, %d-row halos\n", world, ver, ranks, halo);
    else printf("row ring: %d ranks, host-staged transport, %d-row halos\n", world, halo);
  }
  TAU_CK(taugs_ring_finish(r));
  TAU_CK(taugs_ring_barrier(r));
  double t0 = cli_now();
  TAU_CK(taugs_ring_step_async(r, steps));
  TAU_CK(taugs_ring_finish(r));
  TAU_CK(taugs_ring_barrier(r));
  double el = cli_now() - t0;
  TAU_CK(taugs_download(h, u, v));
  /* every rank writes its owned rows into one file laid out as the single-domain arrays; rank 0 then reads the whole grid back and
     prints the reference's summary from it — the same numbers, summed in the same order, as --gpus 1 */
  FILE *f = fopen(gather, "r+b");
  if (!f) { fprintf(stderr, "cannot open %s\n", gather); return 1; }
  for (int k = 0; k < 2; k++) {
    const float *src = (k ? v : u) + (size_t)halo * nx;
    if (fseeko(f, (off_t)(((size_t)k * n + (size_t)y0 * nx) * 4), SEEK_SET) != 0 || fwrite(src, 4, nx * (size_t)nyl, f) != nx * (size_t)nyl) {
      fprintf(stderr, "short write to %s\n", gather); fclose(f); return 1;
    }
  }
  fclose(f);
  TAU_CK(taugs_ring_barrier(r));
  if (rank == 0) {
    f = fopen(gather, "rb");
    if (!f || fread(gu, 4, n, f) != n || fread(gv, 4, n, f) != n) { fprintf(stderr, "cannot read %s back\n", gather); return 1; }
    fclose(f);
    double su = 0, sv = 0;
    for (size_t i = 0; i < n; i++) { su += gu[i]; sv += gv[i]; }
    printf("%d steps on %dx%d: %.3f ms/step, %.2f Gcell-updates/s, sum u = %.9g, sum v = %.9g\n", steps, G->nx, G->ny,
           el / steps * 1e3, (double)n * steps / el / 1e9, su, sv);
    if (dump) {
      char hdr[128];
      snprintf(hdr, sizeof hdr, "tgs f32 u,v nx=%d ny=%d steps=%d", G->nx, G->ny, steps);
      const void *arrs[2] = {gu, gv};
      size_t by[2] = {n * 4, n * 4};
      if (!cli_dump(dump, hdr, arrs, by, 2)) return 1;
    }
  }
  free(gu); free(gv); free(u); free(v);
  taugs_ring_destroy(r);
  taugs_destroy(h);
  return 0;
}

int main(int argc, char **argv) {
  taugs_params P;
  taugs_params_default(&P, 0, 0);
  int steps = 0;
  unsigned seed = 1337;
  const char *dump = NULL;
  int gpus = 1, transport = TAU3D_RING_RCCL, halo = 4;
  /* getopt_long with the reference's own table (:84-104): `--nx 128`, `--nx=128` and unambiguous abbreviations all parse,
     unknown options get getopt's message and are skipped — as there.  --dump is additive. */
  static const struct option long_opts[] = {
      {"nx", required_argument, 0, 0},     {"ny", required_argument, 0, 0},    {"dx", required_argument, 0, 0},
      {"dt", required_argument, 0, 0},     {"Du", required_argument, 0, 0},    {"Dv", required_argument, 0, 0},
      {"F", required_argument, 0, 0},      {"k", required_argument, 0, 0},     {"steps", required_argument, 0, 0},
      {"headless", no_argument, 0, 0},     {"stride", required_argument, 0, 0}, {"fps", required_argument, 0, 0},
      {"seed", required_argument, 0, 0},   {"halfblocks", no_argument, 0, 0},  {"dump", required_argument, 0, 0},
      {"gpus", required_argument, 0, 0},   {"transport", required_argument, 0, 0}, {"halo", required_argument, 0, 0},
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
    else if (!strcmp(opt, "gpus")) gpus = atoi(optarg);
    else if (!strcmp(opt, "halo")) halo = atoi(optarg);
    else if (!strcmp(opt, "transport")) {
      if (!strcmp(optarg, "rccl")) transport = TAU3D_RING_RCCL;
      else if (!strcmp(optarg, "host")) transport = TAU3D_RING_HOST;
      else { fprintf(stderr, "Invalid value for --transport: %s (rccl | host)\n", optarg); return 1; }
    }
    /* headless, halfblocks, stride, fps: display only */
  }
  if (P.nx == 0) P.nx = 128;
  if (P.ny == 0) P.ny = 128;
  if (steps <= 0) steps = 1000;
  if (gpus < 1 || gpus > 64) { fprintf(stderr, "Invalid value for --gpus: %d\n", gpus); return 1; }
  if (halo < 1) { fprintf(stderr, "Invalid value for --halo: %d\n", halo); return 1; }
  if (gpus > 1) {   /* one process per rank, forked BEFORE anything touches the HIP runtime */
    char rv[128], gather[160];
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const uint64_t key = (((uint64_t)getpid() << 32) ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)ts.tv_sec << 20)) | 1u;
    snprintf(rv, sizeof rv, "/dev/shm/tgs_ring_%ld_%lx", (long)getpid(), (unsigned long)(key & 0xffffff));
    snprintf(gather, sizeof gather, "%s.grid", rv);
    FILE *g = fopen(gather, "wb");
    if (!g || ftruncate(fileno(g), (off_t)((size_t)P.nx * P.ny * 8)) != 0) { fprintf(stderr, "cannot create %s\n", gather); return 1; }
    fclose(g);
    fflush(NULL);
    pid_t kids[64];
    for (int r = 0; r < gpus; r++) {
      kids[r] = fork();
      if (kids[r] < 0) { perror("fork"); return 1; }
      if (kids[r] == 0) { int rc = run_rank(&P, steps, seed, dump, r, gpus, transport, halo, rv, key, gather); fflush(NULL); _exit(rc); }
    }
    int bad = 0;
    for (int r = 0; r < gpus; r++) {
      int st = 0;
      if (waitpid(kids[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
        fprintf(stderr, "tgs: rank %d %s %d\n", r, WIFSIGNALED(st) ? "killed by signal" : "exited with", WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st));
        bad = 1;
      }
    }
    unlink(rv); unlink(gather);
    return bad;
  }
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
