#!/usr/bin/env python
"""overlap_probe.py — do k_flux_xy (VALU-bound) and k_update_z (co-limited by HBM) hide each other's stalls when they share the chip?
Two independent 512^3 engines, 20 steps each: on ONE stream (back to back) against on TWO streams (free to overlap).  Evidence for /
against pipelining xy(n+1) beside z(n) inside one engine (DESIGN: round 6)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fluid_sims_amd as f

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def make(stream):
    e = f.Tau3D(n, n, n, stream=ctypes.c_void_p(stream.cuda_stream))
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step_async(10)
    e.sync()
    return e


def timed(fn, syncs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    for s in syncs:
        s()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for label, sa, sb in (("one stream ", s1, s1), ("two streams", s1, s2)):
    a, b = make(sa), make(sb)
    best = 1e9
    for rep in range(3):
        def go():
            for k in range(steps):       # interleave the submissions so that neither queue runs dry
                a.step_async(1)
                b.step_async(1)
        ms = timed(go, (a.sync, b.sync))
        best = min(best, ms)
    print(f"{label}: {best / steps:.3f} ms per pair of steps, {2 * n ** 3 * steps / best / 1e6:.2f} Gcell/s aggregate")
    a.close(); b.close()
