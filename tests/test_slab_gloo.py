"""CPU, multi-process (gloo): the Z-slab ring driver (halo exchange, edge/interior split, max
all-reduce, device-side clock protocol) reproduces the single-domain result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, steps, outdir, prime=True):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    import fluid_sims_amd  # noqa: F401  (package shim)
    slab = import_module("fluid_sims_amd.slab")
    from oracle import pyoracle
    from tests.slab_oracle_backend import OracleSlabBackend
    nx, ny, nz = shape
    p = pyoracle.P3()
    pyoracle._lib("libtauoracle3d.so").o3_params_default(__import__("ctypes").byref(p), nx, ny, nz)
    z0, nzl = slab.slab_bounds(nz, world, rank)
    be = OracleSlabBackend(p, z0, nzl)
    be.init(1)
    be.o.clock.t = 0.02
    be.o.clock.d_tau = 1e-4
    ring = slab.SlabRing(be, rank, world)
    if prime:
        ring.prime()
    ring.step(steps)          # primes itself when the caller has not
    ring.finish()
    c = be.clock()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), z0=z0, nzl=nzl, t=c.t, d_tau=c.d_tau, maxs=c.maxs,
             **{f"f{f}": be.cur[f][3:-3] for f in range(6)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (16, 16, 24)), (3, (24, 16, 20)), (4, (16, 8, 24)),
                                         (8, (8, 8, 64))])   # BASELINE's world size: 8 planes per rank, E = 4
def test_slab_ring_equals_single_domain(oracle_built, tmp_path, world, shape):
    steps = 3
    mp.spawn(_worker, args=(world, _free_port(), shape, steps, str(tmp_path)), nprocs=world, join=True)
    nx, ny, nz = shape
    o = oracle_built.Oracle3D(nx, ny, nz)
    st = o.init(1)
    o.clock.t = 0.02
    o.clock.d_tau = 1e-4
    st = o.run(st, steps)
    want = o.interior(st)
    got = [np.empty((nz, ny, nx), np.float32) for _ in range(6)]
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        z0, nzl = int(d["z0"]), int(d["nzl"])
        for f in range(6):
            got[f][z0:z0 + nzl] = d[f"f{f}"]
        assert float(d["t"]) == o.clock.t and float(d["d_tau"]) == o.clock.d_tau
        assert float(d["maxs"]) == o.clock.maxs, "every rank sees the global max wavespeed"
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_slab_ring_primes_itself(oracle_built, tmp_path):
    """SlabRing.step() without a prime() call: the ring exchanges the halos and agrees on the field range by itself"""
    world, shape, steps = 2, (16, 16, 24), 2
    mp.spawn(_worker, args=(world, _free_port(), shape, steps, str(tmp_path), False), nprocs=world, join=True)
    nx, ny, nz = shape
    o = oracle_built.Oracle3D(nx, ny, nz)
    st = o.init(1)
    o.clock.t = 0.02
    o.clock.d_tau = 1e-4
    want = o.interior(o.run(st, steps))
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        z0, nzl = int(d["z0"]), int(d["nzl"])
        for f in range(6):
            assert np.array_equal(d[f"f{f}"], want[f][z0:z0 + nzl])


def test_slab_bounds():
    from importlib import import_module
    import fluid_sims_amd  # noqa: F401
    slab = import_module("fluid_sims_amd.slab")
    for nz, world in [(512, 8), (512, 1), (50, 4), (24, 4)]:
        cover = []
        for r in range(world):
            z0, nzl = slab.slab_bounds(nz, world, r)
            assert nzl >= 6
            cover += list(range(z0, z0 + nzl))
        assert cover == list(range(nz))
    with pytest.raises(ValueError):
        slab.slab_bounds(20, 4, 0)
