"""Per-rank compute time of the Z-slab schedule for 8 / 4 / 2 ranks, measured on ONE GPU with the exchange left out
(a 512^2 x nzl slab handle driven through the launch sequence of fluid-sims_amd/slab.py)."""
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
import fluid_sims_amd as f
from importlib import import_module
slab = import_module("fluid_sims_amd.slab")
n = 512
L = f.load(); params = f.Tau3DParams(); L.tau3d_params_default(ctypes.byref(params), n, n, n)
for world in (8, 4, 2):
    nzl = n // world
    be = slab.EngineSlabBackend(f.taueng, params, 0, nzl, 0)
    be.h.init(1); be.h.set_clock(0.02, 1e-4)
    E = max(3, min(8, nzl // 2))
    def step(k):
        for _ in range(k):
            be.clock_begin(); be.unpack(0)
            be.step_edges(E); be.pack(1)
            be.step_range(E, nzl - E)
            be.clock_end()
    step(5); be.sync()
    t0 = time.perf_counter(); step(20); be.sync(); el = (time.perf_counter() - t0) / 20
    print("world %d: slab %d planes, %.3f ms/step per rank (no comm)  -> %.1f Gcell/s aggregate if comm hides, ideal %.3f ms" % (world, nzl, el * 1e3, n**3 / el / 1e9, 8.40 / world))
    del be
