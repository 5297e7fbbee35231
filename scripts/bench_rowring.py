"""One-GPU timing of the row-slab ring (fluid-sims_amd/slab2d.py) for Gray-Scott: world = 1 (the halo refresh is a
device copy instead of two RCCL transfers), at the full 8192^2 grid and at the 8192 x 1024 slab one of 8 ranks would
own, for several halo depths H.  Prints one JSON line per case; the plain engine (no ring) is timed beside it."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_sims_amd as f
from importlib import import_module

slab2d = import_module("fluid_sims_amd.slab2d")


def timed(fn, sync, reps):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def main():
    nx = 8192
    steps = 64
    rng = np.random.default_rng(1)
    for ny in (8192, 1024):
        u = (1.0 - 0.5 * rng.random((ny, nx))).astype(np.float32)
        v = (0.25 * rng.random((ny, nx))).astype(np.float32)
        e = f.GrayScott(nx, ny)
        e.upload(u, v)
        t = timed(lambda: e.step_async(steps), e.sync, 5)
        print(json.dumps({"workload": f"gray-scott {nx}x{ny}", "ring": None, "Gcell/s": round(nx * ny * steps / t / 1e9, 1),
                          "us_per_step": round(t / steps * 1e6, 1)}), flush=True)
        e.close()
        for H in (4, 8, 16):
            be = slab2d.EngineRowBackend(lambda nyl, s: f.GrayScott(nx, nyl, stream=s), nx, ny, H, 0)
            be.upload(slab2d.local_rows(u, 0, ny, H), slab2d.local_rows(v, 0, ny, H))
            ring = slab2d.RowRing(be, 0, 1)
            t = timed(lambda: ring.step(steps), be.sync, 5)
            print(json.dumps({"workload": f"gray-scott {nx}x{ny}", "ring": {"world": 1, "H": H},
                              "Gcell/s": round(nx * ny * steps / t / 1e9, 1), "us_per_step": round(t / steps * 1e6, 1),
                              "exchanges_per_step": round(1.0 / H, 3)}), flush=True)
            del ring, be


if __name__ == "__main__":
    main()
