/* th3cs — headless 3D hypersonic run exported as a `.4spl` voxel video (reference: th3cs.cu:1062-1241).
 *
 * Same program: the quiescent start of k_init, 60 frames x 4 steps of the log-time step loop, per frame the
 * Schlieren volume |grad rho| mapped through gamma 0.65 to 256 palette indices, a black-red-yellow-white palette,
 * one file `tau_hypersonic.4spl`.  The step loop, the Schlieren kernel and the index map run in libtaueng
 * (tau3d_step / tau3d_vis mode 0 / tau3d_palette_indices); one byte per voxel per frame crosses PCIe instead of
 * the reference's four.  The reference takes no arguments; --n, --frames, --steps-per-frame and --out exist for
 * the tests. */
#include "tau_cli.h"
#include "tau_4splat.h"

int main(int argc, char **argv) {
  int n = 64, frames = 60, steps_per_frame = 4;
  const int pSize = 256;
  const char *out_path = "tau_hypersonic.4spl";
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    int has = i + 1 < argc;
    if (!strcmp(a, "--n") && has) { if (!cli_int(a, argv[++i], &n)) return 1; }
    else if (!strcmp(a, "--frames") && has) { if (!cli_int(a, argv[++i], &frames)) return 1; }
    else if (!strcmp(a, "--steps-per-frame") && has) { if (!cli_int(a, argv[++i], &steps_per_frame)) return 1; }
    else if (!strcmp(a, "--out") && has) out_path = argv[++i];
    else { fprintf(stderr, "usage: %s [--n N] [--frames F] [--steps-per-frame S] [--out PATH]\n", argv[0]); return 1; }
  }
  if (n < 8 || frames < 1 || steps_per_frame < 1) { fprintf(stderr, "need --n >= 8, --frames >= 1, --steps-per-frame >= 1\n"); return 1; }

  tau3d_params hp;
  tau3d_params_default(&hp, n, n, n);          /* the literals of th3cs.cu:1063-1087 */
  tau3d_t *h = NULL;
  TAU_CK(tau3d_create(&h, &hp, 0, hp.nz, 0, NULL));
  TAU_CK(tau3d_init(h, 0));                    /* k_build_solid_mask + k_init, t = 1e-5, d_tau = 1e-3 (:1128-1131, 1153-1154) */

  const size_t N = (size_t)hp.nx * hp.ny * hp.nz;
  uint8_t *indices = (uint8_t *)malloc(N * (size_t)frames);
  Splat4D *palette = (Splat4D *)malloc(sizeof(Splat4D) * (size_t)pSize);
  if (!indices || !palette) { fprintf(stderr, "out of host memory\n"); return 1; }
  for (int i = 0; i < pSize; i++) {            /* thermal map black -> red -> yellow -> white, :1144-1150 */
    float t_val = (float)i / (pSize - 1.0f);
    float r = fminf(1.0f, t_val * 2.5f);
    float g = fmaxf(0.0f, fminf(1.0f, t_val * 2.5f - 0.5f));
    float b = fmaxf(0.0f, fminf(1.0f, t_val * 2.5f - 1.5f));
    palette[i] = create_splat4D(0, 1, 0, 1, 0, 1, 0, 1, r, g, b, 1.0f);
  }

  printf("Running Hypersonic CFD for %d frames...\n", frames);
  for (int f = 0; f < frames; f++) {
    tau3d_clock c;
    TAU_CK(tau3d_step(h, steps_per_frame, &c));
    TAU_CK(tau3d_vis(h, 0, NULL));                                         /* k_schlieren_export, field stays on the device */
    TAU_CK(tau3d_palette_indices(h, 0.65f, indices + (size_t)f * N, NULL, NULL));
    printf("Frame %d/%d processed (t=%.6f)\n", f + 1, frames, c.t);
  }

  /* 0x0004: fp32 palette, 8-bit indices (:1226) */
  Splat4DHeader header = create_splat4DHeader((uint32_t)hp.nx, (uint32_t)hp.ny, (uint32_t)hp.nz, (uint32_t)frames, (uint32_t)pSize, 0x0004);
  printf("Writing simulation video to %s...\n", out_path);
  FILE *fp = fopen(out_path, "wb");
  int rc = 0;
  if (fp) {
    if (!write_splat4D_u8(fp, &header, palette, indices)) { fprintf(stderr, "Write failed!\n"); rc = 1; }
    fclose(fp);
    if (!rc) printf("Export Complete!\n");
  } else {
    fprintf(stderr, "Failed to open output file!\n");
    rc = 1;
  }
  free(indices);
  free(palette);
  tau3d_destroy(h);
  return rc;
}
