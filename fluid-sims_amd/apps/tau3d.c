/* tau3d — headless driver of the 3D two-temperature hypersonic solver.
 *
 * Stands where the reference's `tau3d` target does (Makefile:81-82, tau_hypersonic_3d_cuda.cu main
 * :1530-1797): same parameters (:1531-1557), same step loop (:1678-1713: log-time clock, k_step,
 * d_tau controller, swap, two steps per frame) — through libtaueng's C-ABI instead of the CUDA
 * launches.  The raylib volume viewer is replaced by --ppm (below); the reference takes no flags, the ones
 * below are additive and default to the reference's behaviour (64^3, quiescent start, one GPU).
 *   --n N | --nx/--ny/--nz   grid (64)         --frames F   frames of 2 steps (60)
 *   --start 0|1              0 = reference k_init, 1 = developed-flow start
 *   --dump PATH              raw dump of xi,phix,phiy,phiz,lam,zet after the run
 *   --ppm PATH               image of one z-slice of the visualisation field after the run (k_vis +
 *                            slice_to_rgba, :800-905, 1416-1442) — the headless stand-in for the viewer
 *   --vis 0..7               VisMode (:784-794; default 0 = |grad rho|, the reference's start-up mode)
 *   --slice Z  --log  --again G   slice (nz/2), log scaling (key L), opacity gain (keys +/-, 1.0)
 *   --gpus N                 Z-slab ring over N GPUs of this node: the program forks one process per device, each
 *                            owns nz/N planes and a tau3d_ring (halo exchange + max all-reduce over RCCL / xGMI,
 *                            issued by the library: include/taueng.h).  The result is bit-identical to --gpus 1.
 *   --transport rccl|ipc|host|ipc-host
 *                            ipc: neighbours' halo planes written directly (hipIpc-mapped, SDMA copies), RCCL for the 8-byte all-reduce;
 *                            ipc-host: the same copies with a host all-reduce (ranks may share a device);
 *                            rccl (default) needs N devices; host stages the halos through shared memory, ranks may
 *                            then share devices (rank r runs on device r mod the device count)
 *   --ring                   with --gpus 1: run the ring anyway (RCCL send / recv to itself)
 */
#include "tau_cli.h"
#include <sys/wait.h>
#include <unistd.h>

typedef struct {
  int nx, ny, nz, frames, start, vis, slice, logs, gpus, transport, ring;
  double again;
  const char *dump, *ppm;
} opts_t;

static int run_rank(const opts_t *o, int rank, const char *rv, uint64_t key) {
  const int steps_per_frame = 2; /* :1643 */
  const int world = o->gpus, use_ring = world > 1 || o->ring;
  cli_need_gpu();
  int ndev = 1;
  if (tau_device_count(&ndev) || ndev < 1) { fprintf(stderr, "taueng error: %s\n", tau_last_error()); return 1; }
  if (world > 1 && (o->transport == TAU3D_RING_RCCL || o->transport == TAU3D_RING_IPC) && ndev < world) {
    /* this launcher places rank r on device r % ndev itself; the library checks device IDENTITIES across ranks for any other launch */
    fprintf(stderr, "tau3d --gpus %d needs %d devices (one per rank over RCCL), this node shows %d; --transport host / ipc-host lets ranks share a device\n",
            world, world, ndev);
    return 1;
  }
  tau3d_params hp;
  tau3d_params_default(&hp, o->nx, o->ny, o->nz);
  int z0 = 0, nzl = o->nz;
  TAU_CK(tau3d_slab_bounds(o->nz, world, rank, &z0, &nzl));
  tau3d_t *h = NULL;
  tau3d_ring_t *r = NULL;
  TAU_CK(tau3d_create(&h, &hp, z0, nzl, rank % ndev, NULL));
  TAU_CK(tau3d_init(h, o->start));
  if (o->start) { tau3d_clock c = {0.02f, 1e-4f, 0.f, 0.f, 0.f, 0}; TAU_CK(tau3d_set_clock(h, &c)); }
  if (use_ring) {
    TAU_CK(tau3d_ring_create(&r, h, rank, world, o->transport, world > 1 ? rv : NULL, key));
    TAU_CK(tau3d_ring_prime(r));
    if (rank == 0) {
      int ver = 0, ranks = 0, edge = 0;
      char lib[256];
      TAU_CK(tau3d_ring_info(r, &ver, &ranks, &edge, lib, sizeof lib));
      if (o->transport == TAU3D_RING_RCCL) printf("ring: %d ranks, RCCL %d (%s), communicator of %d, %d-plane edges\n", world, ver, lib, ranks, edge);
      else if (o->transport == TAU3D_RING_IPC) printf("ring: %d ranks, direct halos (IPC-mapped neighbours, SDMA copies) + RCCL %d all-reduce, %d-plane edges\n", world, ver, edge);
      else if (o->transport == TAU3D_RING_IPC_HOSTMAX) printf("ring: %d ranks, direct halos (IPC-mapped neighbours), host all-reduce, %d-plane edges\n", world, edge);
      else printf("ring: %d ranks, host-staged transport, %d-plane edges\n", world, edge);
    }
    TAU_CK(tau3d_ring_finish(r));
    TAU_CK(tau3d_ring_barrier(r));
  }

  double t0 = cli_now();
  tau3d_clock c;
  if (use_ring) TAU_CK(tau3d_ring_get_clock(r, &c)); else TAU_CK(tau3d_get_clock(h, &c));   /* --frames 0 --dump prints it without ever entering the loop */
  for (int f = 0; f < o->frames; f++) {
    /* the clock lives on the device: only the frames that print it pay for the read-back (the reference copies
       maxs to the host every step, :1697) */
    const int show = (f % 10 == 0 || f == o->frames - 1);
    if (use_ring) {
      TAU_CK(tau3d_ring_step_async(r, steps_per_frame));
      if (show) TAU_CK(tau3d_ring_get_clock(r, &c));
    }
    else if (show) TAU_CK(tau3d_step(h, steps_per_frame, &c));
    else TAU_CK(tau3d_step_async(h, steps_per_frame));
    if (show && rank == 0) /* the reference's HUD line, :1762-1771 */
      printf("frame %d  step %d  t=%.9g  d_tau=%.9g  dt=%.9g  gain=%.3f  maxs=%.9g\n", f, c.step, c.t, c.d_tau, c.dt,
             c.gain, c.maxs);
  }
  if (use_ring) { TAU_CK(tau3d_ring_finish(r)); TAU_CK(tau3d_ring_barrier(r)); }
  else TAU_CK(tau3d_sync(h));
  double el = cli_now() - t0;
  double cells = (double)o->nx * o->ny * o->nz * (double)o->frames * steps_per_frame;
  if (rank == 0)
    printf("%d steps on %dx%dx%d in %.3f s: %.3f Gcell-updates/s%s\n", o->frames * steps_per_frame, o->nx, o->ny, o->nz, el,
           cells / el / 1e9, use_ring ? " (z-slab ring)" : "");

  if (o->ppm) { /* the reference's per-frame tail, :1715-1739; in a ring the slab that owns the slice renders it */
    const int zs = o->slice < 0 ? o->nz / 2 : (o->slice >= o->nz ? o->nz - 1 : o->slice);
    if (use_ring) { TAU_CK(tau3d_ring_prime(r)); TAU_CK(tau3d_ring_finish(r)); }   /* k_vis differentiates across the slab faces: current halos */
    if (zs >= z0 && zs < z0 + nzl) {
      float refl = 0.f, mn = 0.f, mx = 0.f;
      uint32_t *px = (uint32_t *)malloc((size_t)o->nx * o->ny * sizeof(uint32_t));
      TAU_CK(tau3d_vis(h, o->vis, NULL));
      TAU_CK(tau3d_outflow_reflection(h, 6, &refl));
      TAU_CK(tau3d_slice_rgba(h, zs - z0, o->logs, (float)o->again, px, &mn, &mx));
      if (!cli_write_ppm(o->ppm, o->nx, o->ny, px, 1)) return 1;
      printf("vis mode %d slice %d: min %.6g max %.6g  outflow |dp|=%.6g -> %s\n", o->vis, zs, mn, mx, refl, o->ppm);
      free(px);
    }
  }
  if (o->dump) { /* one file for the whole grid: rank 0 lays it out, every rank writes its planes in place */
    const size_t plane = (size_t)o->nx * o->ny, n = plane * (size_t)o->nz, nl = plane * (size_t)nzl;
    float *buf[6];
    for (int k = 0; k < 6; k++) buf[k] = (float *)malloc(nl * sizeof(float));
    TAU_CK(tau3d_download_state(h, buf));
    char hdr[160];
    int hl = snprintf(hdr, sizeof hdr, "tau3d f32 xi,phix,phiy,phiz,lam,zet nx=%d ny=%d nz=%d steps=%d t=%.9g\n", o->nx, o->ny, o->nz, c.step, c.t);
    if (rank == 0) {
      FILE *f = fopen(o->dump, "wb");
      if (!f) { fprintf(stderr, "cannot open %s for writing\n", o->dump); return 1; }
      fwrite(hdr, 1, (size_t)hl, f);
      fclose(f);
      if (truncate(o->dump, (off_t)((size_t)hl + 6 * n * sizeof(float))) != 0) { fprintf(stderr, "cannot size %s\n", o->dump); return 1; }
    }
    if (use_ring) TAU_CK(tau3d_ring_barrier(r));
    FILE *f = fopen(o->dump, "r+b");
    if (!f) { fprintf(stderr, "cannot open %s for writing\n", o->dump); return 1; }
    for (int k = 0; k < 6; k++) {
      if (fseeko(f, (off_t)((size_t)hl + ((size_t)k * n + (size_t)z0 * plane) * sizeof(float)), SEEK_SET) != 0 ||
          fwrite(buf[k], sizeof(float), nl, f) != nl) { fprintf(stderr, "short write to %s\n", o->dump); fclose(f); return 1; }
      free(buf[k]);
    }
    fclose(f);
    if (use_ring) TAU_CK(tau3d_ring_barrier(r));
  }
  tau3d_ring_destroy(r);
  tau3d_destroy(h);
  return 0;
}

int main(int argc, char **argv) {
  opts_t o = {64, 64, 64, 60, 0, 0, -1, 0, 1, TAU3D_RING_RCCL, 0, 1.0, NULL, NULL};
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    int v;
    if (!strcmp(a, "--n") && i + 1 < argc) { if (!cli_int(a, argv[++i], &v)) return 1; o.nx = o.ny = o.nz = v; }
    else if (!strcmp(a, "--nx") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.nx)) return 1; }
    else if (!strcmp(a, "--ny") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.ny)) return 1; }
    else if (!strcmp(a, "--nz") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.nz)) return 1; }
    else if (!strcmp(a, "--frames") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.frames)) return 1; }
    else if (!strcmp(a, "--start") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.start)) return 1; }
    else if (!strcmp(a, "--dump") && i + 1 < argc) o.dump = argv[++i];
    else if (!strcmp(a, "--ppm") && i + 1 < argc) o.ppm = argv[++i];
    else if (!strcmp(a, "--vis") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.vis)) return 1; }
    else if (!strcmp(a, "--slice") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.slice)) return 1; }
    else if (!strcmp(a, "--again") && i + 1 < argc) { if (!cli_double(a, argv[++i], &o.again)) return 1; }
    else if (!strcmp(a, "--log")) o.logs = 1;
    else if (!strcmp(a, "--gpus") && i + 1 < argc) { if (!cli_int(a, argv[++i], &o.gpus)) return 1; }
    else if (!strcmp(a, "--ring")) o.ring = 1;
    else if (!strcmp(a, "--transport") && i + 1 < argc) {
      const char *t = argv[++i];
      if (!strcmp(t, "rccl")) o.transport = TAU3D_RING_RCCL;
      else if (!strcmp(t, "host")) o.transport = TAU3D_RING_HOST;
      else if (!strcmp(t, "ipc")) o.transport = TAU3D_RING_IPC;
      else if (!strcmp(t, "ipc-host")) o.transport = TAU3D_RING_IPC_HOSTMAX;
      else { fprintf(stderr, "Invalid value for --transport: %s (rccl | ipc | host | ipc-host)\n", t); return 1; }
    }
    else { fprintf(stderr, "Unknown or incomplete argument: %s\n", a); return 1; }
  }
  if (o.gpus < 1 || o.gpus > 64) { fprintf(stderr, "Invalid value for --gpus: %d\n", o.gpus); return 1; }
  if (o.gpus == 1) return run_rank(&o, 0, NULL, 0);

  /* one process per rank, forked BEFORE anything touches the HIP runtime (a forked runtime is not usable) */
  char rv[128];
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  const uint64_t key = ((uint64_t)getpid() << 32) ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)ts.tv_sec << 20);
  snprintf(rv, sizeof rv, "/dev/shm/tau3d_ring_%ld_%lx", (long)getpid(), (unsigned long)(key & 0xffffff));
  fflush(NULL);
  pid_t kids[64];
  for (int r = 0; r < o.gpus; r++) {
    kids[r] = fork();
    if (kids[r] < 0) { perror("fork"); return 1; }
    if (kids[r] == 0) { int rc = run_rank(&o, r, rv, key); fflush(NULL); _exit(rc); }
  }
  int bad = 0;
  for (int r = 0; r < o.gpus; r++) {
    int st = 0;
    if (waitpid(kids[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
      fprintf(stderr, "tau3d: rank %d %s %d\n", r, WIFSIGNALED(st) ? "killed by signal" : "exited with", WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st));
      bad = 1;
    }
  }
  unlink(rv);
  return bad;
}
