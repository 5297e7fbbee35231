"""What ONE rank of a 2 / 4 / 8-way 512^3 run computes per step, emulated on one GPU: a 512 x 512 x nzl slab handle under the C
ring with a world of one, for each transport (the halos go to the rank itself: same launches, same bytes; what it cannot show is
the xGMI links).  Steps 5..12 after the impulsive start (a periodic domain this thin leaves the sane range after ~20 steps).
  python scripts/ring_rank_emulation.py [reps]
  python scripts/ring_rank_emulation.py [reps] --inject-allreduce-us 0,20,40,80 [--world 8]
The second form prices the all-reduce: a spin kernel of N us is enqueued behind every all-reduce on the stream it sits on
(TAU3D_RING_INJECT_AR_US, csrc/ring.hip) — a world of one pays ~10 us for it, eight ranks over xGMI plausibly 20-80 — for the round-5
default schedule (x/y fluxes ahead of the ONE all-reduce) and the round-4 one (TAU3D_RING_SPEC=0: two all-reduces on the direct
transport, the first gating the next step's x/y launch)."""
import os
import sys
import time

sys.path.insert(0, ".")
import fluid_sims_amd as f  # noqa: E402

argv = sys.argv[1:]
inject, world_arg = None, 8
if "--inject-allreduce-us" in argv:
    i = argv.index("--inject-allreduce-us")
    inject = [int(x) for x in argv[i + 1].split(",")]
    del argv[i:i + 2]
if "--world" in argv:
    i = argv.index("--world")
    world_arg = int(argv[i + 1])
    del argv[i:i + 2]
reps = int(argv[0]) if argv else 3
n = 512


def run(nzl, transport, pipeline=True, spec=True, inject_us=0):
    os.environ["TAU3D_RING_PIPELINE"] = "1" if pipeline else "0"   # (read at tau3d_ring_create)
    os.environ["TAU3D_RING_SPEC"] = "1" if spec else "0"
    os.environ["TAU3D_RING_INJECT_AR_US"] = str(inject_us)
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


if inject is not None:
    nzl = n // world_arg
    plain = run(nzl, None)
    print(f"one rank of a {world_arg}-way 512^3 run: 512 x 512 x {nzl} planes; the same slab as a plain periodic domain {plain:.4f} ms/step "
          f"(best of {reps}, 8 timed steps)")
    print(f"{'injected us per all-reduce':>28s} | {'round-5 schedule (default)':>36s} | {'round-4 schedule (TAU3D_RING_SPEC=0)':>40s}")
    base = {}
    for us in inject:
        row = []
        for spec in (True, False):
            ms = {t: run(nzl, tr, spec=spec, inject_us=us) for t, tr in (("ipc", f.RING_IPC), ("rccl", f.RING_RCCL))}
            if us == inject[0]:
                base[spec] = ms
            row.append("  ".join(f"{t} {v:.4f} ms ({(v / base[spec][t] - 1) * 100:+.1f} %)" for t, v in ms.items()))
        print(f"{us:>28d} | {row[0]:>36s} | {row[1]:>40s}", flush=True)
    sys.exit(0)

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
           "ring ipc, round-4 pipelined step (all-reduce first)": run(nzl, f.RING_IPC, spec=False),
           "ring ipc, x/y ahead of the all-reduce (default)": run(nzl, f.RING_IPC)}
    print(f"world {world}: {nzl} planes, share of the full step {full / world:.3f} ms | " +
          " | ".join(f"{k} {v:.3f} ms -> x{full / v:.2f}" for k, v in row.items()), flush=True)
