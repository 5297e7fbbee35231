#!/bin/sh
# oracle/build_ref.sh — builds THE REFERENCE'S OWN SOURCES for gfx950.  TEST INFRASTRUCTURE, never part of the product.
#
# What it does: for each hot-path .cu file of the reference, where it lies under /root/reference,
#   hipify-perl (the ROCm image's own source-to-source tool)  ->  a temporary directory outside the repo
#   hipcc --offload-arch=gfx950                                 ->  oracle/_ref/<name>.co   the file's DEVICE code (raw code
#                                                                   object: every __global__ kernel and __constant__ block of
#                                                                   the reference, loadable with hipModuleLoad)
#                                                               ->  oracle/_ref/bin/<name>  the whole program, where the file
#                                                                   needs nothing the image lacks (ncursesw is here)
# Nothing is written for the reference: no stand-in header, library or source.  The hipified text only ever exists in the
# temporary directory; oracle/_ref/ holds binaries only (git-ignored, travels to the GPU box with the snapshot).
#
# Files that include raylib.h (which the image lacks) are built from a PURE LINE CUT of the reference's own text — the lines
# that hold the solver, none of the display code, nothing written in their place:
#   tau_hypersonic.c       lines 1-674  minus the `#include "raylib.h"` line   (everything raylib touches starts at :676 get_color)
#   tau_hypersonic_simd.c  lines 1-804  minus the include                      (:806 get_color)
#       -> oracle/_ref/libref_hyp_cpu{,_simd}_<W>x<H>.so (gcc -O3 / -O3 -mavx2 -mfma, the reference Makefile's flags :57-61);
#          W, H are compile-time `#define`s of the reference (:12-13 / :24-25): sizes other than 300x300 are the survey's `sed`
#          on exactly those two lines.  The cut is #included by a harness that only forwards to the file's own static
#          init_sim() / compute_dt() / step_physics() and copies U / mask / sim_t out (ref_init, ref_step, ref_state ...).
#   tau_hypersonic_3d_cuda.cu  lines 1-1409 minus the raylib includes (4-5) and the Vector3 helpers (69-101) — SURVEY 8c's cut
#       -> hipify-perl -> oracle/_ref/tau_hypersonic_3d_cuda{,.ieee}.co: k_step, k_init, k_build_solid_mask, k_vis,
#          k_maxwavespeed_pre, k_schlieren, k_outflow_reflection_metric of the file north_star names
#   slice_to_rgba (tau_hypersonic_3d_cuda.cu:1416-1442) and the palette map of th3cs.cu (:1199-1222) are host C++: cut by
#       line and compiled with g++ into oracle/_ref/libref_hostmaps.so
# Still absent: tau_hypersonic_cuda.cu's own main (raylib window loop) — the file builds through the reference's own seam
# (tau_hypersonic_cuda_tests.cu:6-8) — and th3cs.cu's main (needs 4splat.c, which the reference tree does not hold).
#
# Flags follow the reference Makefile (:69-94; nvcc -use_fast_math -> -ffast-math) plus a second object of each file without
# contraction and without fast-math (<name>.ieee.co: -ffp-contract=off) for the results this repo claims BIT-exact (masks, cell
# indices, Gray-Scott, shallow-water viscosity, LBM): a compiler that fuses a*b+c rounds once where the source rounds twice.
set -eu
REF=${TAU_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
ROCM=${ROCM:-/opt/rocm}
HIPCC=${HIPCC:-$ROCM/bin/hipcc}
HIPIFY=${HIPIFY:-$ROCM/bin/hipify-perl}
READELF=$ROCM/lib/llvm/bin/llvm-readelf
ARCH=${PYTORCH_ROCM_ARCH:-gfx950}

if [ ! -d "$REF" ]; then
    echo "build_ref: $REF not present (GPU box): using the prebuilt oracle/_ref as it travelled" >&2
    exit 0
fi
[ -x "$HIPIFY" ] || { echo "build_ref: no hipify-perl at $HIPIFY" >&2; exit 1; }

TMP=$(mktemp -d /tmp/tau_ref_build.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT/bin"

FILES="th3cs tau_hypersonic_cuda tau_hypersonic_cuda_tests tau_gray_scott tau_sph tau_lbm tau_burgers tau_shallow_water"
HP=""
for f in $FILES; do
    # device_launch_parameters.h has no HIP counterpart: hipify leaves an empty include behind, dropped here
    ( "$HIPIFY" "$REF/$f.cu" 2>/dev/null | sed '/^#include <>$/d' > "$TMP/$f.cu" ) & HP="$HP $!"
done
# the file north_star names: SURVEY 8c's line cut (1-1409 minus the raylib includes 4-5 and the Vector3 helpers 69-101)
( sed -n '1,1409p' "$REF/tau_hypersonic_3d_cuda.cu" | sed -e '4,5d' -e '69,101d' > "$TMP/tau_hypersonic_3d_cuda.cut.cu" &&
  "$HIPIFY" "$TMP/tau_hypersonic_3d_cuda.cut.cu" 2>/dev/null | sed '/^#include <>$/d' > "$TMP/tau_hypersonic_3d_cuda.cu" ) & HP="$HP $!"
for p in $HP; do wait "$p"; done
FILES="$FILES tau_hypersonic_3d_cuda"
for f in $FILES; do [ -s "$TMP/$f.cu" ] || { echo "build_ref: hipify-perl produced nothing for $f.cu" >&2; exit 1; }; done

devobj() {  # name out-suffix flags...
    n=$1; suf=$2; shift 2
    "$HIPCC" --offload-arch=$ARCH -std=c++17 "$@" -w --cuda-device-only --no-gpu-bundle-output -x hip -c "$TMP/$n.cu" -o "$OUT/$n$suf.co"
    # mangled <tab> demangled, kernels and device globals
    "$READELF" -s --wide "$OUT/$n$suf.co" | awk '($4=="FUNC"||$4=="OBJECT") && $5=="GLOBAL" && $7!="UND" {print $8}' | sort -u |
        grep -v '\.kd$' | grep -v '^__hip_cuid' | while read -r s; do printf '%s\t%s\n' "$s" "$(c++filt "$s")"; done > "$OUT/$n$suf.syms"
}
program() {  # name binary flags...
    n=$1; b=$2; shift 2
    "$HIPCC" --offload-arch=$ARCH -std=c++17 "$@" -w -x hip "$TMP/$n.cu" -o "$OUT/bin/$b" -lncursesw
}

PIDS=""
bg() { "$@" & PIDS="$PIDS $!"; }
bg devobj th3cs "" -O3
bg devobj th3cs .ieee -O3 -ffp-contract=off
bg devobj tau_hypersonic_3d_cuda "" -O3
bg devobj tau_hypersonic_3d_cuda .ieee -O3 -ffp-contract=off
bg devobj tau_hypersonic_cuda_tests "" -O2
bg devobj tau_gray_scott "" -O3 -ffast-math
bg devobj tau_gray_scott .ieee -O3 -ffp-contract=off
bg devobj tau_sph "" -O3 -ffast-math
bg devobj tau_sph .ieee -O3 -ffp-contract=off
bg devobj tau_lbm "" -O3
bg devobj tau_lbm .ieee -O3 -ffp-contract=off
bg devobj tau_burgers "" -O3 -ffast-math
bg devobj tau_burgers .ieee -O3 -ffp-contract=off
bg devobj tau_shallow_water "" -O3 -ffast-math
bg devobj tau_shallow_water .ieee -O3 -ffp-contract=off

bg program tau_hypersonic_cuda_tests tau_hypersonic_cuda_tests -O2
bg program tau_gray_scott tgs -O3 -ffast-math
bg program tau_sph tau_sph -O3 -ffast-math
bg program tau_lbm tau_lbm -O3
bg program tau_burgers tau_burgers -O3 -ffast-math
bg program tau_shallow_water tau_sw -O3 -ffast-math

# ---- the CPU solvers (BASELINE config C1): line cuts of the reference's own text, compiled by gcc --------------------------
sed -n '1,674p' "$REF/tau_hypersonic.c"      | sed '/#include "raylib.h"/d' > "$TMP/hyp_cpu.cut.c"
sed -n '1,804p' "$REF/tau_hypersonic_simd.c" | sed '/#include "raylib.h"/d' > "$TMP/hyp_cpu_simd.cut.c"
cat > "$TMP/hyp_harness.c" <<'C'
/* forwards to the cut's own static functions and copies its static arrays out; computes nothing */
#include REF_CUT
int ref_w(void) { return W; }
int ref_h(void) { return H; }
void ref_init(void) { init_sim(); }
double ref_compute_dt(void) { return compute_dt(); }
void ref_step(int n) { for (int s = 0; s < n; s++) step_physics(); }
double ref_time(void) { return sim_t; }
void ref_state(double *u4, unsigned char *m) { memcpy(u4, U, sizeof U); memcpy(m, mask, sizeof mask); }
void ref_set_state(const double *u4, double t) { memcpy(U, u4, sizeof U); sim_t = t; }
C
cpulib() {  # cut out-stem W H flags...
    cut=$1; stem=$2; w=$3; h=$4; shift 4
    # the grid size is a pair of #defines in the reference (tau_hypersonic.c:12-13, _simd.c:24-25): SURVEY 8c's sed
    sed -e "s/^#define W 300\$/#define W $w/" -e "s/^#define H 300\$/#define H $h/" "$TMP/$cut.cut.c" > "$TMP/${stem}_${w}x${h}.c"
    grep -q "^#define W $w\$" "$TMP/${stem}_${w}x${h}.c" || { echo "build_ref: W/H patch did not apply to $cut" >&2; exit 1; }
    gcc "$@" -w -fPIC -shared -DREF_CUT="\"$TMP/${stem}_${w}x${h}.c\"" "$TMP/hyp_harness.c" -o "$OUT/libref_${stem}_${w}x${h}.so" -lm
}
for wh in "300 300" "256 256" "96 64"; do
    set -- $wh
    bg cpulib hyp_cpu hyp_cpu "$1" "$2" -O3
    bg cpulib hyp_cpu_simd hyp_cpu_simd "$1" "$2" -O3 -mavx2 -mfma
done

# ---- the two host-side colour maps: plain C++ in the .cu files, cut by line, g++ ---------------------------------------------
sed -n '1410,1442p' "$REF/tau_hypersonic_3d_cuda.cu" > "$TMP/slice_to_rgba.cut.inc"      # clamp01, safe_log1p, slice_to_rgba
sed -n '1199,1222p' "$REF/th3cs.cu" > "$TMP/th3cs_palette.cut.inc"                        # min/max, gamma 0.65, index 0..255
cat > "$TMP/hostmaps.cpp" <<'C'
// forwards to the cut's slice_to_rgba; the th3cs palette lines sit in the middle of its main's frame loop, so the harness
// declares the names those lines read (h_sch, hp, f, N, h_indices) and pastes the lines between them
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "slice_to_rgba.cut.inc"
extern "C" void ref_slice_to_rgba(uint32_t *dst, const float *vol, int nx, int ny, int nz, int zslice, int log_scale, float a_gain) {
    slice_to_rgba(dst, vol, nx, ny, nz, zslice, log_scale != 0, a_gain);
}
extern "C" void ref_th3cs_palette(const float *sch, int nx, int ny, int nz, int frame, uint64_t *indices) {
    struct { int nx, ny, nz; } hp = {nx, ny, nz};
    size_t N = (size_t)nx * ny * nz;
    std::vector<float> h_sch(sch, sch + N);
    uint64_t *h_indices = indices;
    int f = frame;
#include "th3cs_palette.cut.inc"
}
C
bg g++ -std=c++17 -O2 -w -fPIC -shared -I"$TMP" "$TMP/hostmaps.cpp" -o "$OUT/libref_hostmaps.so"

for p in $PIDS; do wait "$p" || { echo "build_ref: a compile failed" >&2; exit 1; }; done

( cd "$OUT" && ls -1 *.co *.so bin/* | sort > MANIFEST )
echo "build_ref: $(wc -l < "$OUT/MANIFEST") reference objects in $OUT"
