#!/usr/bin/env python
"""Throughput of the non-headline configs of BASELINE.json on one MI355X (inputs resident in HBM):
Gray-Scott 8192^2, Burgers / shallow-water viscosity passes and full steps 8192^2, 2D Euler 4096^2 fp32, SPH 4M
particles, D2Q9 LBM 8192^2.
Prints one JSON line per workload with the algorithmic-bytes roofline fraction."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import fluid_sims_amd as f  # noqa: E402

HBM = 8000.0


def timed(step_async, sync, units_per_step, steps, warm):
    step_async(warm)
    sync()
    t0 = time.perf_counter()
    step_async(steps)
    sync()
    el = time.perf_counter() - t0
    return units_per_step * steps / el, el / steps * 1e3


def line(name, unit, rate, ms, bytes_per_unit, bound, extra=None):
    gbs = rate * bytes_per_unit / 1e9
    d = {"workload": name, "value": round(rate / 1e9, 4), "unit": "G" + unit + "/s", "ms_per_step": round(ms, 4),
         "algorithmic_bytes_per_unit": bytes_per_unit, "achieved_GBps": round(gbs, 1), "hbm_peak_GBps": HBM,
         "frac_hbm": round(gbs / HBM, 4), "binding_bound": bound}
    d.update(extra or {})
    print(json.dumps(d), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default=None, help="sph_developed: the SPH developed state alone (1 720 sub-steps in, then the 200 bench.py times)")
    a = ap.parse_args()
    k = 0.25 if a.quick else 1.0

    if a.only == "sph_developed":
        N = 1 << 22
        s = f.Sph2D(N)
        s.reset_particles()
        s.step_async(1720)        # bench.py: 20 + 200 (its lattice window) + 1 500 sub-steps before the developed window
        s.sync()
        r, ms = timed(s.step_async, s.sync, N, 200, 0)      # the window bench.py times: sub-steps 1 720 .. 1 920
        line(f"tau_sph {N} particles, developed state (1 720 sub-steps after reset)", "particle-substeps", r, ms, 100, "pair-evaluation valu",
             {"grid": s.grid(), "timed_substeps": 200})
        s.close()
        return

    n = 8192
    g = f.GrayScott(n, n)
    g.init_pattern(1337)
    g.set_levels(1)          # the reference's structure, one launch per step: st2::k_march<0> (the 70 % HBM figure of bench.py's configs)
    r, ms = timed(g.step_async, g.sync, n * n, int(200 * k), 8)
    line(f"tau_gray_scott {n}^2, one step per launch", "cell-updates", r, ms, 16, "hbm")
    g.set_levels(4)
    r, ms = timed(g.step_async, g.sync, n * n, int(400 * k), 20)
    fused = {"levels_per_pass": 4, "hbm_bytes_per_update_moved": 4.6,
             "note": "4 time levels per pass (temporal fusion): 16 B is the single-step algorithmic figure, the pass moves ~4.6 B per update"}
    line(f"tau_gray_scott {n}^2", "cell-updates", r, ms, 16, "valu (hbm for a single step)", fused)
    g.close()
    rng = np.random.default_rng(1)
    fld = (rng.standard_normal((n, n)) * 0.5).astype(np.float32)
    for kind, bound in (("sw", "valu (hbm for a single pass)"), ("burgers", "valu (sinh/asinh)")):
        h = f.Laplacian2D(n, n, kind, nu=0.1, dt=0.2, u0=1.0)
        h.upload(fld, fld * 0.5)
        r, ms = timed(h.step_async, h.sync, n * n, int(400 * k), 10)
        line(f"{kind} viscosity pass {n}^2", "cell-updates", r, ms, 16, bound, fused)
        h.close()
    del fld

    n = 8192
    for kind, nbytes, kw in (("burgers", 16, {}), ("burgers", 16, {"muscl": 1}), ("sw", 24, {})):
        fl = f.Flow2D(kind, n, n, dtau=0.01, **kw)
        fl.init()
        r, ms = timed(fl.step_async, fl.sync, n * n, int(200 * k), 10)
        name = {"burgers": "tau_burgers", "sw": "tau_sw"}[kind] + (" --muscl" if kw else "")
        line(f"{name} full step {n}^2", "cell-updates", r, ms, nbytes, "valu (sinh/asinh, sqrt, divides)", {"clock": fl.clock()})
        fl.close()

    n = 4096
    e = f.Hypersonic2D(n, n)
    e.init()
    r, ms = timed(e.step_async, e.sync, n * n, int(400 * k), 100)
    line(f"tau_hypersonic_cuda {n}^2 fp32", "cell-updates", r, ms, 33, "valu", {"sim": e.time()})
    e.close()

    N = 1 << 22
    s = f.Sph2D(N)
    s.reset_particles()
    r, ms = timed(s.step_async, s.sync, N, int(100 * k), 20)
    line(f"tau_sph {N} particles", "particle-substeps", r, ms, 100, "pair-evaluation valu", {"grid": s.grid()})
    s.close()

    n = 8192
    lb = f.Lbm2D(n, n, obstacle_radius=n / 8)
    lb.init()
    r, ms = timed(lb.step_async, lb.sync, n * n, int(200 * k), 20)
    line(f"tau_lbm D2Q9 {n}^2", "cell-updates", r, ms, 72, "valu (hbm for a single step)",
         {"levels_per_pass": 4, "note": "4 time levels per pass: 72 B is the single-step algorithmic figure"})
    lb.close()


if __name__ == "__main__":
    main()
