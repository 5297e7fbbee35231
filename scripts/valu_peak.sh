#!/bin/bash
# scripts/valu_peak.sh [outfile] — build and run the VALU issue-rate micro-benchmark (needs an MI355X)
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Wno-unused-value scripts/valu_peak.hip -o scratch/valu_peak
if [ -n "$1" ]; then ./scratch/valu_peak | tee "$1"; else ./scratch/valu_peak; fi
