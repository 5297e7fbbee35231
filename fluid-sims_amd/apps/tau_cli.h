/* tau_cli.h — helpers shared by the plain-C drivers: strict number parsing, the reference's
 * print-and-exit error convention (`ck`, tau_hypersonic_3d_cuda.cu:62-67), raw dumps, timing. */
#ifndef TAU_CLI_H
#define TAU_CLI_H
#include "../../include/taueng.h"
#include <errno.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* engine call or die, like the reference's ck(): message to stderr, exit(1) */
#define TAU_CK(call)                                                        \
  do {                                                                      \
    if ((call) != 0) {                                                      \
      fprintf(stderr, "taueng error: %s: %s\n", #call, tau_last_error());   \
      exit(1);                                                              \
    }                                                                       \
  } while (0)

static inline int cli_double(const char *name, const char *value, double *out) {
  char *end = NULL;
  double v = strtod(value, &end);
  if (!end || *end != '\0' || !isfinite(v)) {
    fprintf(stderr, "Invalid value for %s: %s\n", name, value);
    return 0;
  }
  *out = v;
  return 1;
}
static inline int cli_int(const char *name, const char *value, int *out) {
  char *end = NULL;
  errno = 0;
  long v = strtol(value, &end, 10);
  if (!end || *end != '\0' || errno == ERANGE || v < INT_MIN || v > INT_MAX) {
    fprintf(stderr, "Invalid value for %s: %s\n", name, value);
    return 0;
  }
  *out = (int)v;
  return 1;
}
static inline double cli_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* raw dump: a small text header line, then the arrays back to back */
static inline int cli_dump(const char *path, const char *header, const void *const *arrs, const size_t *bytes, int n) {
  FILE *f = fopen(path, "wb");
  if (!f) { fprintf(stderr, "cannot open %s for writing\n", path); return 0; }
  fprintf(f, "%s\n", header);
  for (int i = 0; i < n; i++)
    if (fwrite(arrs[i], 1, bytes[i], f) != bytes[i]) { fclose(f); return 0; }
  fclose(f);
  return 1;
}
/* binary PPM (P6) from RGBA8 pixels (R in the lowest byte, as the reference uploads to raylib); flip = 1
 * writes the rows bottom-up (the reference draws its textures with y up) */
static inline int cli_write_ppm(const char *path, int w, int h, const uint32_t *rgba, int flip) {
  FILE *f = fopen(path, "wb");
  if (!f) { fprintf(stderr, "cannot open %s for writing\n", path); return 0; }
  fprintf(f, "P6\n%d %d\n255\n", w, h);
  unsigned char *row = (unsigned char *)malloc((size_t)w * 3);
  for (int y = 0; y < h; y++) {
    const uint32_t *src = rgba + (size_t)(flip ? h - 1 - y : y) * w;
    for (int x = 0; x < w; x++) {
      row[3 * x] = (unsigned char)(src[x] & 255u);
      row[3 * x + 1] = (unsigned char)((src[x] >> 8) & 255u);
      row[3 * x + 2] = (unsigned char)((src[x] >> 16) & 255u);
    }
    if (fwrite(row, 3, (size_t)w, f) != (size_t)w) { free(row); fclose(f); return 0; }
  }
  free(row);
  fclose(f);
  return 1;
}
static inline void cli_need_gpu(void) {
  if (!tau_device_available()) {
    fprintf(stderr, "no gfx950 (MI355X) device visible: this program has no CPU path\n");
    exit(1);
  }
}
#endif
