#!/bin/bash
# scripts/regen_checkvalues.sh — regenerates the CPU-solver check-values of tests/golden/ref_checkvalues.json from the
# reference's own source and compares them with the committed digits.
#
# BUILD CONTAINER ONLY: it needs /root/reference (absent on the GPU box; the path is in .gpurunignore) and works entirely in
# a temporary directory — no reference source and no stand-in build enters the repository or travels anywhere.  This is
# SURVEY.md §8(c) / Appendix A's recipe for the two CPU files: a declarations-only raylib.h (display types and no-op window
# calls: the solver never reads anything from them), `#define main ref_main`, `#include "<reference file>"`, then the
# file's own static init_sim() / step_physics() called directly.  It is NOT oracle/_ref (the task's rule: a reference that
# needs stand-in headers is unbuildable; the oracles are pinned to the recorded outputs) — it only makes the
# transcribed digits of the CPU entries reproducible.  The CUDA entries (2D / 3D / Gray-Scott / SPH) came from the survey's
# host-side block emulator (Appendix A), which is not reproduced here.
set -euo pipefail
REF=${REF:-/root/reference}
[ -f "$REF/tau_hypersonic.c" ] || { echo "regen_checkvalues: $REF/tau_hypersonic.c not found (build container only)"; exit 2; }
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
mkdir -p "$T/stub"
cat > "$T/stub/raylib.h" <<'H'
/* declarations only: what the two CPU files mention of raylib; nothing here computes */
#ifndef RAYLIB_H
#define RAYLIB_H
typedef struct { unsigned char r, g, b, a; } Color;
typedef struct { void *data; int width, height, mipmaps, format; } Image;
typedef struct { unsigned id; int width, height, mipmaps, format; } Texture2D;
typedef struct { float x, y, width, height; } Rectangle;
typedef struct { float x, y; } Vector2;
enum { KEY_R = 82, KEY_M = 77, KEY_SPACE = 32, PIXELFORMAT_UNCOMPRESSED_R8G8B8A8 = 7 };
#define WHITE ((Color){255, 255, 255, 255})
#define BLACK ((Color){0, 0, 0, 255})
#define GREEN ((Color){0, 228, 48, 255})
#define RAYWHITE WHITE
#define RED WHITE
#define YELLOW WHITE
#define GRAY WHITE
static inline void InitWindow(int w, int h, const char *t) { (void)w; (void)h; (void)t; }
static inline void SetTargetFPS(int f) { (void)f; }
static inline Texture2D LoadTextureFromImage(Image i) { Texture2D t = {0, i.width, i.height, 1, i.format}; return t; }
static inline int WindowShouldClose(void) { return 1; }
static inline int IsKeyPressed(int k) { (void)k; return 0; }
static inline int IsKeyDown(int k) { (void)k; return 0; }
static inline void UpdateTexture(Texture2D t, const void *p) { (void)t; (void)p; }
static inline void BeginDrawing(void) {}
static inline void EndDrawing(void) {}
static inline void ClearBackground(Color c) { (void)c; }
static inline void DrawTexturePro(Texture2D t, Rectangle a, Rectangle b, Vector2 o, float r, Color c) { (void)t; (void)a; (void)b; (void)o; (void)r; (void)c; }
static inline void DrawText(const char *s, int x, int y, int z, Color c) { (void)s; (void)x; (void)y; (void)z; (void)c; }
static inline const char *TextFormat(const char *f, ...) { return f; }
static inline void UnloadTexture(Texture2D t) { (void)t; }
static inline void CloseWindow(void) {}
static inline int GetFPS(void) { return 0; }
static inline void DrawFPS(int x, int y) { (void)x; (void)y; }
#endif
H
cat > "$T/harness.c" <<'C'
#define main ref_main
#include REF_FILE
#undef main
#include <stdio.h>
int main(int argc, char **argv) {
  int steps = argc > 1 ? atoi(argv[1]) : 1;
  init_sim();
  for (int s = 0; s < steps; s++) step_physics();
  double sr = 0, sm = 0, sE = 0; long fluid = 0;
  for (int i = 0; i < W * H; i++) if (!mask[i]) { fluid++; sr += U[i].rho; sm += U[i].mx; sE += U[i].E; }
  printf("{\"t\": %.17g, \"fluid\": %ld, \"sum_rho\": %.17g, \"sum_mx\": %.17g, \"sum_E\": %.17g}\n", sim_t, fluid, sr, sm, sE);
  if (argc > 2) { /* whole fields: W, H, steps, t, then rho / mx / my / E planes (fp64) and the mask */
    FILE *f = fopen(argv[2], "wb");
    int hdr[3] = {W, H, steps};
    fwrite(hdr, sizeof hdr, 1, f); fwrite(&sim_t, sizeof sim_t, 1, f);
    for (int k = 0; k < 4; k++) for (int i = 0; i < W * H; i++) { double v = k == 0 ? U[i].rho : k == 1 ? U[i].mx : k == 2 ? U[i].my : U[i].E; fwrite(&v, 8, 1, f); }
    fwrite(mask, 1, W * H, f);
    fclose(f);
  }
  return 0;
}
C
build() { # name, file, extra flags
  gcc -O3 $3 -I"$T/stub" -DREF_FILE="\"$2\"" "$T/harness.c" -lm -o "$T/$1" 2> "$T/$1.log" || { cat "$T/$1.log"; exit 1; }
}
build cpu300 "$REF/tau_hypersonic.c" ""
sed 's/^#define W 300/#define W 256/; s/^#define H 300/#define H 256/' "$REF/tau_hypersonic.c" > "$T/ref256.c"
build cpu256 "$T/ref256.c" ""
build simd300 "$REF/tau_hypersonic_simd.c" "-mavx2 -mfma"
{ echo "{"; echo "\"cpu300_1\": $("$T/cpu300" 1),"; echo "\"cpu300_24\": $("$T/cpu300" 24),"; echo "\"cpu256_8\": $("$T/cpu256" 8),"; echo "\"simd300_10\": $("$T/simd300" 10)"; echo "}"; } > "$T/out.json"
# field fixtures (data: the reference's own outputs; SURVEY §8c fixture (ii) for config C1): 96 x 64 after 12 and 13 steps
sed 's/^#define W 300/#define W 96/; s/^#define H 300/#define H 64/' "$REF/tau_hypersonic.c" > "$T/ref96.c"
build cpu96 "$T/ref96.c" ""
"$T/cpu96" 12 "$T/f12.bin" > /dev/null; "$T/cpu96" 13 "$T/f13.bin" > /dev/null
python3 - "$T/f12.bin" "$T/f13.bin" "$ROOT/tests/golden/cpu2d_ref_96x64_steps12_13.npz" <<'PY'
import sys, numpy as np
def rd(p):
    b = open(p, "rb").read()
    W, H, steps = np.frombuffer(b[:12], np.int32)
    t = np.frombuffer(b[12:20], np.float64)[0]
    f = np.frombuffer(b[20:20 + 32 * W * H], np.float64).reshape(4, H, W)
    m = np.frombuffer(b[20 + 32 * W * H:], np.uint8).reshape(H, W)
    return int(W), int(H), int(steps), t, f, m
W, H, s0, t0, f0, m0 = rd(sys.argv[1]); _, _, s1, t1, f1, m1 = rd(sys.argv[2])
new = dict(W=W, H=H, steps0=s0, steps1=s1, t0=t0, t1=t1, U0=f0, U1=f1, mask=m0)
try:
    old = np.load(sys.argv[3])
    same = all(np.array_equal(old[k], np.asarray(v)) for k, v in new.items())
    print(("ok   " if same else "DIFF ") + "96x64 field fixture (steps 12 -> 13) against the committed tests/golden file")
except FileNotFoundError:
    np.savez_compressed(sys.argv[3], **new); print("wrote", sys.argv[3])
PY
python3 - "$T/out.json" "$ROOT/tests/golden/ref_checkvalues.json" <<'PY'
import json, sys
new, gold = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
g3, g2 = gold["tau_hypersonic_cpu_300sq"], gold["tau_hypersonic_cpu_256sq_8steps"]
checks = [("300^2 t after 1 step", new["cpu300_1"]["t"], g3["t_1step"]), ("300^2 t after 24", new["cpu300_24"]["t"], g3["t_24steps"]),
          ("300^2 fluid", new["cpu300_24"]["fluid"], g3["fluid"]), ("300^2 sum rho", new["cpu300_24"]["sum_rho"], g3["sum_rho_24"]),
          ("300^2 sum mx", new["cpu300_24"]["sum_mx"], g3["sum_mx_24"]), ("300^2 sum E", new["cpu300_24"]["sum_E"], g3["sum_E_24"]),
          ("256^2 t after 8", new["cpu256_8"]["t"], g2["t"]), ("256^2 fluid", new["cpu256_8"]["fluid"], g2["fluid"]),
          ("256^2 sum rho", new["cpu256_8"]["sum_rho"], g2["sum_rho"]),
          ("SIMD file 300^2 sum rho after 10 (tests/test_drivers.py)", new["simd300_10"]["sum_rho"], 82947.469425548319)]
bad = 0
for name, a, b in checks:
    ok = a == b
    bad += not ok
    print(("ok   " if ok else "DIFF ") + f"{name}: regenerated {a!r} committed {b!r}")
sys.exit(1 if bad else 0)
PY
