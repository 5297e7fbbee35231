#!/bin/bash
# scripts/valu_calib.sh [outfile] — build and run the VALU issue-ceiling calibration (needs an MI355X)
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Wno-unused-value scripts/valu_calib.hip -o scratch/valu_calib
if [ -n "$1" ]; then ./scratch/valu_calib | tee "$1"; else ./scratch/valu_calib; fi
