#!/bin/bash
# scripts/profile_3d.sh <tag> [passes...] — the 3D step's counter passes only (profile_round.sh without the secondary kernels);
# TAUENG_LIB selects a variant build.  passes: stats sq lds fetch write grbm (default: stats sq lds)
set -u
TAG=${1:-x}; shift
PASSES=${*:-stats sq lds}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
CMD="python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-variants --no-configs"
run() { local name=$1; shift; rm -rf /tmp/rp_$name
  rocprofv3 "$@" -d /tmp/rp_$name -o x -- $CMD > /tmp/rp_$name.log 2>&1
  { echo "# rocprofv3 $* -- $CMD  (TAUENG_LIB=${TAUENG_LIB:-in-tree})"; grep -E '^\{"metric"' /tmp/rp_$name.log | cut -c1-300; python scripts/rocpd_summary.py /tmp/rp_$name/x_results.db; } > "$OUT/$name.txt"; }
for p in $PASSES; do case $p in
  stats) run kernel_stats --kernel-trace --stats;;
  sq) run pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY;;
  lds) run pmc_lds --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS;;
  fetch) run pmc_fetch --pmc FETCH_SIZE;;
  write) run pmc_write --pmc WRITE_SIZE;;
  grbm) run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES;;
esac; done
