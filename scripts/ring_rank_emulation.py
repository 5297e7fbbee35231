"""What ONE rank of a 2 / 4 / 8-way 512^3 run computes per step, emulated on one GPU: a 512 x 512 x nzl slab handle under the C
ring with a world of one, for each transport (the halos go to the rank itself: same launches, same bytes; what it cannot show is
the xGMI links).  Steps 5..12 after the impulsive start (a periodic domain this thin leaves the sane range after ~20 steps).
  python scripts/ring_rank_emulation.py [reps]"""
import sys
import time

sys.path.insert(0, ".")
import fluid_sims_amd as f  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 512


def run(nzl, transport, pipeline=True):
    import os
    os.environ["TAU3D_RING_PIPELINE"] = "1" if pipeline else "0"   # (read at tau3d_ring_create)
    best = 1e9
    for _ in range(reps):
        e = f.Tau3D(n, n, nzl)
        e.init(1)
        e.set_clock(0.02, 1e-4)
        if transport is None:
            step, fin = e.step_async, e.sync
            ring = None
        else:
            ring = f.Tau3DRing(e, 0, 1, transport)
            ring.prime()
            step, fin = ring.step, ring.finish
        step(5)
        fin()
        t0 = time.perf_counter()
        step(8)
        fin()
        best = min(best, (time.perf_counter() - t0) / 8 * 1e3)
        assert e.field_range()[2], "left the fast window"
        if ring:
            ring.close()
        e.close()
    return best


e = f.Tau3D(n)
e.init(1)
e.set_clock(0.02, 1e-4)
e.step(5)
t0 = time.perf_counter()
e.step(8)
full = (time.perf_counter() - t0) / 8 * 1e3
e.close()
print(f"512^3 single domain: {full:.3f} ms/step")
for world in (2, 4, 8):
    nzl = n // world
    row = {"plain periodic slab": run(nzl, None), "ring local copies": run(nzl, f.RING_LOCAL), "ring rccl-to-self": run(nzl, f.RING_RCCL),
           "ring ipc (direct halos) + rccl all-reduce, edge / interior launches": run(nzl, f.RING_IPC, pipeline=False),
           "ring ipc, pipelined step (default)": run(nzl, f.RING_IPC)}
    print(f"world {world}: {nzl} planes, share of the full step {full / world:.3f} ms | " +
          " | ".join(f"{k} {v:.3f} ms -> x{full / v:.2f}" for k, v in row.items()), flush=True)
