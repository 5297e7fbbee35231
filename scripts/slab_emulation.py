"""Per-rank compute time of the Z-slab schedule for 8 / 4 / 2 ranks, measured on ONE GPU with the exchange left out
(a 512^2 x nzl slab handle driven through the launch sequence of fluid-sims_amd/slab.py)."""
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
import fluid_sims_amd as f
from importlib import import_module
slab = import_module("fluid_sims_amd.slab")
n = 512
L = f.load(); params = f.Tau3DParams(); L.tau3d_params_default(ctypes.byref(params), n, n, n)
def single_domain():   # single-domain ms per step on this box (the clock drifts by a few per cent between runs: best of two)
    e = f.Tau3D(n); e.init(1); e.set_clock(0.02, 1e-4); e.step_async(10); e.sync()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); e.step_async(10); e.sync(); best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    e.close(); del e
    return best
IDEAL = single_domain()
print("single domain: %.3f ms/step" % IDEAL)
for world in (8, 4, 2):
    nzl = n // world
    be = slab.EngineSlabBackend(f.taueng, params, 0, nzl, 0)
    be.h.init(1); be.h.set_clock(0.02, 1e-4)
    E = max(3, min(8, nzl // 2))
    def step(k):
        for _ in range(k):
            be.begin()            # controller + clock + unpack: one kernel
            be.edges(E)           # (x/y flux kernel over all planes +) edge planes, boundary planes packed by the same launch
            be.interior(E)
            be.end()
    step(5); be.sync()
    el = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); step(20); be.sync(); el = min(el, (time.perf_counter() - t0) / 20)
    print("world %d: slab %d planes, %.3f ms/step per rank (no comm)  -> %.1f Gcell/s aggregate if comm hides, ideal %.3f ms (%.0f %%)" % (world, nzl, el * 1e3, n**3 / el / 1e9, IDEAL / world, 100 * IDEAL / world / (el * 1e3)))
    del be
