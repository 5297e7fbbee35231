#!/bin/sh
# Runs the reference's own programs (oracle/_ref/bin, built by oracle/build_ref.sh) headless on the GPU box and keeps what
# they print under gpurun_out/ref/ — evidence for profiles/, not part of any test.
set -u
O=gpurun_out/ref
mkdir -p $O
B=oracle/_ref/bin
export TERM=${TERM:-xterm}
( cd $O && timeout 600 ../../$B/tau_hypersonic_cuda_tests --steps 24 --write-baseline --baseline ref_baseline_8192x1024_24steps.txt ) > $O/h2_tests_write.txt 2>&1; echo "h2 tests write rc=$?" >> $O/h2_tests_write.txt
( cd $O && timeout 600 ../../$B/tau_hypersonic_cuda_tests --steps 24 --verify-baseline --baseline ref_baseline_8192x1024_24steps.txt ) > $O/h2_tests_verify.txt 2>&1; echo "rc=$?" >> $O/h2_tests_verify.txt
t0=$(date +%s%N); timeout 300 $B/tgs --headless --nx 8192 --ny 8192 --steps 1000 > $O/tgs_8192.txt 2>&1; rc=$?; t1=$(date +%s%N); echo "rc=$rc wall_ms=$(( (t1 - t0) / 1000000 )) (1000 steps of 8192^2, process start to exit)" >> $O/tgs_8192.txt
timeout 300 $B/tgs --headless --steps 100 > $O/tgs_default.txt 2>&1; echo "rc=$?" >> $O/tgs_default.txt
# tau_sph has no --steps: it runs until killed and prints its step count every 100*stride steps
timeout 40 $B/tau_sph --n 4194304 --headless --stride 1 > $O/sph_4m.txt 2>&1; echo "rc=$? (killed after 40 s)" >> $O/sph_4m.txt
timeout 300 $B/tau_lbm --headless --steps 1000 > $O/lbm.txt 2>&1; echo "rc=$?" >> $O/lbm.txt
timeout 300 $B/tau_burgers --headless --steps 1000 > $O/burgers.txt 2>&1; echo "rc=$?" >> $O/burgers.txt
timeout 300 $B/tau_sw --headless --steps 1000 > $O/sw.txt 2>&1; echo "rc=$?" >> $O/sw.txt
tail -n 5 $O/*.txt
