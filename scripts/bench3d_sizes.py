#!/usr/bin/env python
"""3D hypersonic step time against the grid size, fused kernel (TAU3D_SPLIT=0) vs split step (=1): where the
cross-over between "dispatch-latency bound" and "occupancy bound" sits (tau3d_create's default rule)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_sims_amd as f  # noqa: E402


def run(n, split, steps):
    os.environ["TAU3D_SPLIT"] = str(split)
    e = f.Tau3D(n)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step_async(20)
    e.sync()
    t0 = time.perf_counter()
    e.step_async(steps)
    e.sync()
    dt = (time.perf_counter() - t0) / steps
    e.close()
    return dt


for n in [int(a) for a in sys.argv[1:]] or [64, 96, 128, 192, 256, 384]:
    steps = max(10, int(2e8 / n ** 3))
    r = {"n": n}
    for split in (0, 1):
        dt = run(n, split, steps)
        r["split" if split else "fused"] = {"us_per_step": round(dt * 1e6, 1), "Gcell_s": round(n ** 3 / dt / 1e9, 2)}
    print(json.dumps(r), flush=True)
