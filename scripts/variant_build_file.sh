#!/bin/bash
# scripts/variant_build_file.sh NAME FILE "<extra hipcc flags>" — a tuning build of libtaueng.so with ONE engine source (h2d, sph,
# stencil2d, flow2d, lbm) recompiled under other flags / macros, the other objects from the in-tree build.
# Output: build_var/NAME/libtaueng.so (TAUENG_LIB; scripts/ab2d.py).
set -eu
NAME=$1; F=$2; DEFS="-DTAU_EXPERIMENT ${3:-}"   # (the sources refuse tuning overrides without it)
cd "$(dirname "$0")/../fluid-sims_amd"
make -s >/dev/null
OUT=../build_var/$NAME; mkdir -p "$OUT"
EXTRA=$(make -pn 2>/dev/null | grep "^EXTRA_$F " | sed 's/^[^=]*= *//' | sed 's/\$(H3D_DEFS)//')
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -Wno-unused-value $EXTRA $DEFS -c csrc/$F.hip -o "$OUT/$F.o"
OBJS=$(ls build/*.o | grep -v -E "/$F\.o$")
g++ -shared -fPIC -o "$OUT/libtaueng.so" $OBJS "$OUT/$F.o" -ldl
echo "$OUT/libtaueng.so"
