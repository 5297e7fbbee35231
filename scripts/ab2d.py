#!/usr/bin/env python
"""ab2d.py [--rounds R] [--what euler|sph|gs] name=path/libtaueng.so ... — interleaved A/B timing of builds of the 2D configs
(2D Euler 4096^2, SPH 4 M lattice + developed, Gray-Scott 8192^2), each build in its own subprocess (TAUENG_LIB)."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys, time
sys.path.insert(0, %r)
import fluid_sims_amd as f
what = sys.argv[1]
def timed(step, sync, n):
    sync(); t0 = time.perf_counter(); step(n); sync(); return (time.perf_counter() - t0) / n * 1e3
out = {}
if what == "euler":
    for n in (4096, 8192):
        e = f.Hypersonic2D(n, n); e.init(); e.step_async(50)
        out["%%d^2 Gcell/s" %% n] = n * n / timed(e.step_async, e.sync, 200) / 1e6
        e.close()
elif what == "sph":
    N = 1 << 22
    s = f.Sph2D(N); s.reset_particles(); s.step_async(20)
    out["lattice Gp/s"] = N / timed(s.step_async, s.sync, 100) / 1e6
    s.step_async(1400)
    out["developed Gp/s"] = N / timed(s.step_async, s.sync, 100) / 1e6
    s.close()
elif what == "gs":
    n = 8192
    g = f.GrayScott(n, n); g.init_pattern(1337); g.step_async(40)
    out["fused Gcell/s"] = n * n / timed(g.step_async, g.sync, 400) / 1e6
    g.set_levels(1); g.step_async(8)
    out["single Gcell/s"] = n * n / timed(g.step_async, g.sync, 400) / 1e6
    g.close()
print(json.dumps(out))
''' % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--what", default="euler")
    ap.add_argument("builds", nargs="+")
    a = ap.parse_args()
    builds = [b.split("=", 1) for b in a.builds]
    res = {k: {} for k, _ in builds}
    for r in range(a.rounds):
        for name, path in builds:
            env = dict(os.environ, TAUENG_LIB=os.path.abspath(path))
            out = subprocess.run([sys.executable, "-c", CHILD, a.what], env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception:
                print(name, "FAILED", out.stderr[-400:], flush=True)
                continue
            for k, v in d.items():
                res[name].setdefault(k, []).append(v)
    for name, _ in builds:
        print(f"{name:24s} " + "   ".join(f"{k} {statistics.median(v):.3f} [{' '.join('%.3f' % x for x in v)}]" for k, v in res[name].items()), flush=True)


if __name__ == "__main__":
    main()
