"""GPU, multi-process: the row-slab ring with the REAL HIP engine on every rank (ranks share the one GPU of the
box; gloo transport with the packed row buffers staged through host memory).  Bit-identical to the single-domain
engine run, fused passes included (H = 4 rows = the four time levels of one fused pass)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(f, kind, nx):
    if kind == "gs":
        return lambda ny_local, stream: f.GrayScott(nx, ny_local, stream=stream)
    return lambda ny_local, stream: f.Laplacian2D(nx, ny_local, kind, 0.1, 0.2, stream=stream)


def _initial(kind, nx, ny):
    rng = np.random.default_rng(11)
    if kind == "gs":
        u = (1.0 - 0.5 * rng.random((ny, nx))).astype(np.float32)
        v = (0.25 * rng.random((ny, nx))).astype(np.float32)
    else:
        u, v = (rng.standard_normal((ny, nx)).astype(np.float32) for _ in range(2))
    return u, v


def _worker(rank, world, port, kind, nx, ny, H, nsteps, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    import fluid_sims_amd as f
    slab2d = import_module("fluid_sims_amd.slab2d")

    class Staged(slab2d.RowRing):
        """same ring; the tensors handed to gloo are host copies of the device buffers"""

        def exchange(self):
            b = self.b
            dev = b.buf
            host = {k: torch.empty(t.shape, dtype=t.dtype) for k, t in dev.items()}
            # pack on the device, move through the host, unpack on the device
            H_, nyl, n = b.H, b.nyl, b.H * b.nx
            fl = b.fields()
            for k, a in enumerate(fl):
                dev[("send", 0)][k * n:(k + 1) * n].copy_(a[H_:2 * H_].reshape(-1))
                dev[("send", 1)][k * n:(k + 1) * n].copy_(a[nyl:nyl + H_].reshape(-1))
            torch.cuda.current_stream().synchronize()
            for s in (0, 1):
                host[("send", s)].copy_(dev[("send", s)])
            ops = [dist.P2POp(dist.isend, host[("send", 0)], self.lo, self.group, tag=0),
                   dist.P2POp(dist.isend, host[("send", 1)], self.hi, self.group, tag=1),
                   dist.P2POp(dist.irecv, host[("recv", 1)], self.hi, self.group, tag=0),
                   dist.P2POp(dist.irecv, host[("recv", 0)], self.lo, self.group, tag=1)]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            for s in (0, 1):
                dev[("recv", s)].copy_(host[("recv", s)])
            for k, a in enumerate(fl):
                a[0:H_].reshape(-1).copy_(dev[("recv", 0)][k * n:(k + 1) * n])
                a[nyl + H_:nyl + 2 * H_].reshape(-1).copy_(dev[("recv", 1)][k * n:(k + 1) * n])
            torch.cuda.current_stream().synchronize()

    u, v = _initial(kind, nx, ny)
    y0, nyl = slab2d.row_bounds(ny, world, rank)
    be = slab2d.EngineRowBackend(_make(f, kind, nx), nx, nyl, H, 0)
    be.upload(slab2d.local_rows(u, y0, nyl, H), slab2d.local_rows(v, y0, nyl, H))
    ring = Staged(be, rank, world)
    ring.step(nsteps)
    ring.finish()
    a, b = be.download_owned()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), y0=y0, nyl=nyl, a=a, b=b)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,shape,H,nsteps", [("gs", 2, (512, 384), 4, 12), ("gs", 4, (2048, 1024), 4, 10),
                                                       ("sw", 2, (768, 512), 4, 8), ("burgers", 3, (512, 300), 4, 9)])
def test_engine_row_ring_equals_single_domain(eng, tmp_path, kind, world, shape, H, nsteps):
    nx, ny = shape
    mp.spawn(_worker, args=(world, _free_port(), kind, nx, ny, H, nsteps, str(tmp_path)), nprocs=world, join=True)
    u, v = _initial(kind, nx, ny)
    e = _make(eng, kind, nx)(ny, None)
    e.upload(u, v)
    e.step(nsteps)
    wu, wv = e.download()
    gu, gv = np.empty_like(wu), np.empty_like(wv)
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        y0, nyl = int(d["y0"]), int(d["nyl"])
        gu[y0:y0 + nyl], gv[y0:y0 + nyl] = d["a"], d["b"]
    assert np.isfinite(wu).all() and np.array_equal(gu, wu) and np.array_equal(gv, wv)
    e.close()


def test_world1_ring_on_device(eng):
    """the production exchange path (device tensors, no staging) with the periodic self-neighbour"""
    from importlib import import_module
    slab2d = import_module("fluid_sims_amd.slab2d")
    nx, ny, H = 1024, 640, 4
    u, v = _initial("gs", nx, ny)
    be = slab2d.EngineRowBackend(_make(eng, "gs", nx), nx, ny, H, 0)
    be.upload(slab2d.local_rows(u, 0, ny, H), slab2d.local_rows(v, 0, ny, H))
    slab2d.RowRing(be, 0, 1).step(14).finish()
    gu, gv = be.download_owned()
    e = eng.GrayScott(nx, ny)
    e.upload(u, v)
    e.step(14)
    wu, wv = e.download()
    assert np.array_equal(gu, wu) and np.array_equal(gv, wv)
