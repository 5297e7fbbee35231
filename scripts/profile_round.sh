#!/bin/bash
# scripts/profile_round.sh <tag> — rocprofv3 evidence for bench.py's roofline figures.
# Run on the GPU box (gpurun); writes text summaries under gpurun_out/profiles_<tag>/ which are then
# copied into profiles/ and committed.  Counters are collected in their own passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; never combined with sys/hip traces).
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
CMD="python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-variants --no-configs"   # bench.py's default window
run() { # name, rocprof args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 "$@" -d /tmp/rp_$name -o x -- $CMD > /tmp/rp_$name.log 2>&1
  { echo "# rocprofv3 $* -- $CMD"; grep -E '^\{"metric"' /tmp/rp_$name.log | cut -c1-400; python scripts/rocpd_summary.py /tmp/rp_$name/x_results.db; } > "$OUT/$name.txt"
}
run kernel_stats --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES
run pmc_lds --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
# secondary kernels: Gray-Scott / Laplacians / 2D Euler / SPH through one script
CMD="python scripts/bench_secondary.py --quick"
run secondary_stats --kernel-trace --stats
run secondary_fetch --pmc FETCH_SIZE
run secondary_write --pmc WRITE_SIZE
# the SPH developed state by itself: the kernels carry the same names as on the lattice, so the summary takes the timed tail only
CMD="python scripts/bench_secondary.py --only sph_developed"
runlast() { # name, rocprof args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 "$@" -d /tmp/rp_$name -o x -- $CMD > /tmp/rp_$name.log 2>&1
  { echo "# rocprofv3 $* -- $CMD   (summary: the last 200 dispatches of every kernel = the timed sub-steps)"; grep -E '^\{"workload"' /tmp/rp_$name.log | cut -c1-400; python scripts/rocpd_summary.py /tmp/rp_$name/x_results.db --last 200; } > "$OUT/$name.txt"
}
runlast sphdev_stats --kernel-trace --stats
runlast sphdev_fetch --pmc FETCH_SIZE
runlast sphdev_write --pmc WRITE_SIZE
ls -la "$OUT"
