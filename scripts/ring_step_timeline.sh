#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# usage: ring_step_timeline.sh [modes...]   modes: plain | ipc0 (TAU3D_RING_PIPELINE=0) | ipc4 / rccl4 (TAU3D_RING_SPEC=0: the round-4 schedule) | ipc | rccl | local
# (TAU3D_RING_INJECT_AR_US in the environment is passed through)
MODES=${@:-plain ipc0 ipc}
for mode in $MODES; do
rm -rf /tmp/rt && mkdir -p /tmp/rt
export TAU3D_RING_PIPELINE=1 TAU3D_RING_SPEC=1; m=$mode
if [ $mode = ipc0 ]; then export TAU3D_RING_PIPELINE=0; m=ipc; fi
if [ $mode = ipc4 ]; then export TAU3D_RING_SPEC=0; m=ipc; fi
if [ $mode = rccl4 ]; then export TAU3D_RING_SPEC=0; m=rccl; fi
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/rt -o t -- python scripts/ring_step_timeline.py $m > /tmp/rt/log.txt 2>&1
python - $mode <<'PY'
import csv, sys, glob
mode = sys.argv[1]
ev = []
for fn in glob.glob("/tmp/rt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:34]))
for fn in glob.glob("/tmp/rt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20]))
ev.sort()
# last 2 steps: find the last 3 k_flux_xy<true,false> starts as step markers
marks = [i for i, e in enumerate(ev) if "k_clock_turn" in e[2] or "k_clock_begin" in e[2] or "k_halo_pack" in e[2] and False]
if not marks: marks = [i for i, e in enumerate(ev) if "k_flux_xy<true, false>" in e[2]]
# print the last ~2 steps
i0 = marks[-3] if len(marks) >= 3 else 0
t0 = ev[i0][0]
print("==", mode)
for s, e, nme in ev[i0:]:
    print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  {nme}")
PY
done
