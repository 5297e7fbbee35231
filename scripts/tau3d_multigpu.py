"""tau3d on N GPUs of one node: the 3D hypersonic run of bin/tau3d, Z-slab decomposed over torch.distributed (RCCL).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/tau3d_multigpu.py \\
         --n 512 --frames 50 --steps-per-frame 2 [--mode 0|1] [--dump-rank0 PATH]

One process per GPU; rank r owns planes [z0, z0 + nzl) (fluid-sims_amd/slab.py).  Rank 0 prints the HUD line of the
reference's frame loop (tau_hypersonic_3d_cuda.cu:1762-1771) after every frame and the aggregate rate at the end.
With one process (no launcher) it runs the same ring with itself as both neighbours."""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--steps-per-frame", type=int, default=2)
    ap.add_argument("--mode", type=int, default=0, help="0: the reference's quiescent start, 1: impulsive start at t = 0.02")
    ap.add_argument("--dump-rank0", default=None, help="write rank 0's slab (6 fields, raw fp32) here at the end")
    a = ap.parse_args(argv)

    import numpy as np
    import torch
    import torch.distributed as dist
    import fluid_sims_amd as f
    from importlib import import_module
    slab = import_module("fluid_sims_amd.slab")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    params = f.Tau3DParams()
    f.load().tau3d_params_default(ctypes.byref(params), a.n, a.n, a.n)
    z0, nzl = slab.slab_bounds(a.n, world, rank)
    be = slab.EngineSlabBackend(f.taueng, params, z0, nzl, local)
    be.h.init(a.mode)
    if a.mode:
        be.h.set_clock(0.02, 1e-4)
    ring = slab.SlabRing(be, rank, world)
    ring.prime()
    if rank == 0:
        print(f"tau3d {a.n}^3 on {world} GPU(s): slabs of {nzl} planes, {a.frames} frames x {a.steps_per_frame} steps", flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for fr in range(a.frames):
        ring.step(a.steps_per_frame)
        if rank == 0 and (fr % 10 == 9 or fr == a.frames - 1):
            ring.finish()
            c = be.clock()
            print(f"frame {fr}  step {c.step}  t={c.t:.6g}  d_tau={c.d_tau:.4g}  dt={c.dt:.4g}  gain={c.gain:.3f}  maxs={c.maxs:.6g}", flush=True)
    ring.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    steps = a.frames * a.steps_per_frame
    if rank == 0:
        print(f"{steps} steps on {a.n}x{a.n}x{a.n} over {world} GPU(s) in {el:.3f} s: {float(a.n) ** 3 * steps / el / 1e9:.3f} Gcell-updates/s", flush=True)
        if a.dump_rank0:
            np.stack(be.h.download()).tofile(a.dump_rank0)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
