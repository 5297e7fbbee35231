cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r05d; mkdir -p $OUT
CMD="python scripts/bench_secondary.py --only sph_developed"
runlast() { local name=$1; shift; rm -rf /tmp/rp_$name; rocprofv3 "$@" -d /tmp/rp_$name -o x -- $CMD > /tmp/rp_$name.log 2>&1
  { echo "# rocprofv3 $* -- $CMD   (summary: the last 200 dispatches of every kernel = the timed sub-steps)"; grep -E '^\{"workload"' /tmp/rp_$name.log | cut -c1-400; python scripts/rocpd_summary.py /tmp/rp_$name/x_results.db --last 200; } > "$OUT/$name.txt"; }
runlast sphdev_stats --kernel-trace --stats
runlast sphdev_fetch --pmc FETCH_SIZE
runlast sphdev_write --pmc WRITE_SIZE
head -5 $OUT/sphdev_stats.txt | cut -c1-200
