#!/bin/bash
# scripts/profile_extra_r05.sh — round-5 evidence beside profile_round.sh: (a) rocprof timelines of the 64-plane rank's step under the
# round-5 and round-4 ring schedules, (b) SQ counters of the SPH lattice sub-step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r05x; mkdir -p $OUT
bash scripts/ring_step_timeline.sh plain ipc4 ipc > $OUT/ring_step_timeline.txt 2>&1
cat > /tmp/sphlat.py <<'P'
import sys
sys.path.insert(0, '.')
import fluid_sims_amd as f
N = 1 << 22
s = f.Sph2D(N); s.reset_particles(); s.step_async(20); s.sync(); s.step_async(50); s.sync(); s.close()
P
rm -rf /tmp/rpx; rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU -d /tmp/rpx -o x -- python /tmp/sphlat.py > /tmp/rpx.log 2>&1
{ echo "# rocprofv3 --pmc SQ_* -- SPH 4 M particles, lattice start, 50 sub-steps after 20 (the last 50 dispatches of every kernel)"; python scripts/rocpd_summary.py /tmp/rpx/x_results.db --last 50 | grep -E "k_density|k_forces|KERNEL|name"; } > $OUT/pmc_sq_sph_lattice.txt
# (c) was a TCC_HIT / TCC_MISS / TCC_REQ pass over the bench command: it did not finish within 20 minutes on the box (the per-channel
# TCC counters multiply the passes) and was dropped — k_update_z's HBM side is argued from FETCH_SIZE / WRITE_SIZE and its duration.
ls -la $OUT
