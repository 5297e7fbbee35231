"""CPU, multi-process: the row-slab ring of the 2D stencil simulators (fluid-sims_amd/slab2d.py) over gloo with
the CPU oracles as the stepper — every owned row bit-identical to the single-domain oracle run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(kind, nx, ny):
    from oracle import pyoracle as O
    o = O.Oracle2D()
    if kind == "gs":
        u, v = o.gs_init(nx, ny, 1337)
        stepper = lambda a, b, n: o.gs_step(o.gs_params(nx, a.shape[0]), a, b, n)
    else:
        rng = np.random.default_rng(3)
        u = rng.standard_normal((ny, nx)).astype(np.float32)
        v = rng.standard_normal((ny, nx)).astype(np.float32)
        def stepper(a, b, n):
            p = O.LapParams(nx, a.shape[0], 1.0, 1.0, 0.1, 0.2, 1.0)
            return o.lap_step(kind, p, a, b, n)
    return u, v, stepper


def _worker(rank, world, port, kind, nx, ny, H, nsteps, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    slab2d = import_module("fluid_sims_amd.slab2d")
    from tests.row_oracle_backend import OracleRowBackend
    u, v, stepper = _setup(kind, nx, ny)
    y0, nyl = slab2d.row_bounds(ny, world, rank)
    be = OracleRowBackend(stepper, nx, nyl, H)
    be.upload(slab2d.local_rows(u, y0, nyl, H), slab2d.local_rows(v, y0, nyl, H))
    ring = slab2d.RowRing(be, rank, world)
    ring.step(nsteps)
    ring.finish()
    a, b = be.download_owned()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), y0=y0, nyl=nyl, a=a, b=b)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,shape,H,nsteps", [("gs", 2, (48, 40), 4, 11), ("gs", 3, (32, 50), 2, 7),
                                                       ("sw", 2, (40, 36), 4, 8), ("burgers", 4, (24, 41), 3, 6),
                                                       ("gs", 8, (32, 64), 4, 9)])   # 8 ranks, 8 rows each
def test_row_ring_equals_single_domain(oracle_built, tmp_path, kind, world, shape, H, nsteps):
    nx, ny = shape
    mp.spawn(_worker, args=(world, _free_port(), kind, nx, ny, H, nsteps, str(tmp_path)), nprocs=world, join=True)
    u, v, stepper = _setup(kind, nx, ny)
    wu, wv = stepper(u, v, nsteps)
    gu, gv = np.empty_like(wu), np.empty_like(wv)
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        y0, nyl = int(d["y0"]), int(d["nyl"])
        gu[y0:y0 + nyl], gv[y0:y0 + nyl] = d["a"], d["b"]
    assert np.array_equal(gu, wu) and np.array_equal(gv, wv)


def test_row_bounds_and_local_rows():
    from importlib import import_module
    sys.path.insert(0, ROOT)
    s = import_module("fluid_sims_amd.slab2d")
    for ny, world in ((50, 3), (8192, 8), (7, 7)):
        b = [s.row_bounds(ny, world, r) for r in range(world)]
        assert b[0][0] == 0 and sum(n for _, n in b) == ny and all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(world - 1))
    f = np.arange(10 * 3, dtype=np.float32).reshape(10, 3)
    loc = s.local_rows(f, 8, 2, 2)
    assert np.array_equal(loc[:, 0] // 3, [6, 7, 8, 9, 0, 1])
