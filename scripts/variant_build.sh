#!/bin/bash
# scripts/variant_build.sh NAME "<-D flags / hipcc flags for h3d.hip>" [xy|z|both]
# A tuning build of libtaueng.so with other macro settings for the 3D step: only h3d.o / h3d_split.o are recompiled, the other
# objects come from the in-tree build.  Output: build_var/NAME/libtaueng.so (load it with TAUENG_LIB; scripts/ab3d.py).
set -eu
NAME=$1; DEFS="-DTAU_EXPERIMENT ${2:-}"   # (the sources refuse tuning overrides without it)
cd "$(dirname "$0")/../fluid-sims_amd"
make -s >/dev/null
OUT=../build_var/$NAME; mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -Wno-unused-function -Wno-sometimes-uninitialized -Wno-unused-value -Wno-unused-const-variable -ffp-contract=on"
$HIPCC $FLAGS $DEFS -c csrc/h3d.hip -o "$OUT/h3d.o" &
$HIPCC $FLAGS $DEFS -DTAU3D_SPLIT_TU -mllvm -amdgpu-sched-strategy=${SCHED:-max-ilp} -c csrc/h3d.hip -o "$OUT/h3d_split.o" &
wait
OBJS=$(ls build/*.o | grep -v -E '/h3d(_split)?\.o$')
g++ -shared -fPIC -o "$OUT/libtaueng.so" $OBJS "$OUT/h3d.o" "$OUT/h3d_split.o" -ldl
echo "$OUT/libtaueng.so"
