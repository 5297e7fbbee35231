#!/usr/bin/env python
"""ab3d.py [--rounds R] [--steps K] [--n 512] name=path/libtaueng.so ... — interleaved A/B timing of 3D-step builds.

Each build is loaded in its OWN subprocess (TAUENG_LIB), R rounds over all builds in turn on the same box; prints per build the
k_flux_xy / k_update_z event times (ms) of every round and their medians — the protocol behind DESIGN §8's "interleaved" rows."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import fluid_sims_amd as f
n, steps, late = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
e = f.Tau3D(n)
if late:
    e.init(0); e.step_async(late)
else:
    e.init(1); e.set_clock(0.02, 1e-4); e.step_async(10)
e.sync()
e.timing_enable(True)
e.step_async(steps); e.sync()
xy, z, k = e.timing_read_split()
print(json.dumps({"xy": xy / k, "z": z / k}))
''' % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--late", type=int, default=0, help="ramped start + this many steps instead of the impulsive headline input")
    ap.add_argument("builds", nargs="+")
    a = ap.parse_args()
    builds = [b.split("=", 1) if "=" in b else (os.path.basename(os.path.dirname(b)), b) for b in a.builds]
    res = {k: {"xy": [], "z": []} for k, _ in builds}
    for r in range(a.rounds):
        for name, path in builds:
            env = dict(os.environ, TAUENG_LIB=os.path.abspath(path))
            out = subprocess.run([sys.executable, "-c", CHILD, str(a.n), str(a.steps), str(a.late)], env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception:
                print(name, "FAILED", out.stderr[-400:], flush=True)
                continue
            res[name]["xy"].append(d["xy"]); res[name]["z"].append(d["z"])
    for name, _ in builds:
        x, z = res[name]["xy"], res[name]["z"]
        if not x:
            continue
        mx, mz = statistics.median(x), statistics.median(z)
        print(f"{name:28s} xy {mx:.3f}  z {mz:.3f}  step {mx + mz:.3f} ms  {a.n ** 3 / (mx + mz) / 1e6:.2f} Gcell/s   "
              f"xy[{' '.join('%.3f' % v for v in x)}] z[{' '.join('%.3f' % v for v in z)}]", flush=True)


if __name__ == "__main__":
    main()
