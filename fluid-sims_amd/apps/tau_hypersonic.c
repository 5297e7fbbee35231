/* tau_hypersonic / tau_hypersonic_simd — headless drivers of the CPU 2D Euler solver (BASELINE
 * config 1).  Stand where the reference's targets do (Makefile:57-61); the reference programs take
 * no flags and open a raylib window (out of scope) — additive: --W / --H (300 x 300 compile-time in
 * the reference), --steps N (default 100), --dump PATH.  Built twice: plain (-O3) and with
 * -DTAU_SIMD -O3 -mavx2 -mfma (the AVX2 compute_dt of tau_hypersonic_simd.c:556-637).
 * This is a CPU program in the reference too: it never touches the GPU engine.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct th2_sim th2_sim;
th2_sim *th2_create(int W, int H, int simd_dt);
void th2_destroy(th2_sim *);
void th2_init(th2_sim *);
double th2_step(th2_sim *);
double th2_time(const th2_sim *);
const double *th2_state(const th2_sim *);
void th2_sums(const th2_sim *, long *fluid, double out[4]);

int main(int argc, char **argv) {
  int W = 300, H = 300, steps = 100;
  const char *dump = NULL;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--W") && i + 1 < argc) W = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--H") && i + 1 < argc) H = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
    else { fprintf(stderr, "Usage: %s [--W W] [--H H] [--steps N] [--dump PATH]\n", argv[0]); return 1; }
  }
  if (W < 8 || H < 8 || steps < 0) { fprintf(stderr, "bad size / step count\n"); return 1; }
#ifdef TAU_SIMD
  const int simd = 1;
#else
  const int simd = 0;
#endif
  th2_sim *S = th2_create(W, H, simd);
  th2_init(S);
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int s = 0; s < steps; s++) th2_step(S);
  clock_gettime(CLOCK_MONOTONIC, &b);
  double el = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
  long fluid; double sums[4];
  th2_sums(S, &fluid, sums);
  printf("%s %dx%d: %d steps, t=%.17g, fluid=%ld, sum_rho=%.17g, sum_mx=%.17g, sum_E=%.17g\n",
         simd ? "tau_hypersonic_simd" : "tau_hypersonic", W, H, steps, th2_time(S), fluid, sums[0], sums[1], sums[3]);
  printf("%.3f s: %.3f Mcell-updates/s on 1 thread\n", el, (double)W * H * steps / (el > 0 ? el : 1e-9) / 1e6);
  if (dump) {
    FILE *f = fopen(dump, "wb");
    if (!f) { fprintf(stderr, "cannot open %s\n", dump); return 1; }
    fprintf(f, "tau_hypersonic f64 AoS rho,mx,my,E W=%d H=%d t=%.17g\n", W, H, th2_time(S));
    fwrite(th2_state(S), sizeof(double), (size_t)W * H * 4, f);
    fclose(f);
  }
  th2_destroy(S);
  return 0;
}
