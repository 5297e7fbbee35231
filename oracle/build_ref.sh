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
# Unbuildable here and therefore absent: tau_hypersonic_3d_cuda.cu, tau_hypersonic_cuda.cu's own main, tau_hypersonic.c,
# tau_hypersonic_simd.c (they include raylib.h, which the image lacks) and th3cs.cu's main (needs 4splat.c, which the reference
# tree does not hold).  th3cs.cu's device code — the reference author's headless copy of the 3D solver: the same k_step,
# k_init, k_build_solid_mask as tau_hypersonic_3d_cuda.cu:759-770, 939-985, 987-1359 without the dead Tv solves —
# builds, and tau_hypersonic_cuda.cu builds through the reference's own seam (tau_hypersonic_cuda_tests.cu:6-8).
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
for p in $HP; do wait "$p"; done
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
for p in $PIDS; do wait "$p" || { echo "build_ref: a compile failed" >&2; exit 1; }; done

( cd "$OUT" && ls -1 *.co bin/* | sort > MANIFEST )
echo "build_ref: $(wc -l < "$OUT/MANIFEST") reference objects in $OUT"
