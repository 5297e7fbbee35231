"""GPU, multi-process: the Z-slab ring with the REAL HIP engine on every rank.  The GPU box has one device, so the
ranks share it and the transport is gloo with the packed halo buffers staged through host memory — everything else
is the production path (EngineSlabBackend: device-side pack / unpack, tau3d_step_edges_async, interior range, the max
word, the device clock).  Result: bit-identical to the single-domain engine run."""
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


def _worker(rank, world, port, shape, steps, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes
    from importlib import import_module
    import fluid_sims_amd as f
    slab = import_module("fluid_sims_amd.slab")

    class Staged(slab.EngineSlabBackend):
        """same engine calls; the tensors handed to torch.distributed are host copies of the device buffers"""

        def __init__(self, *a):
            super().__init__(*a)
            self._host = {k: torch.empty(v.shape, dtype=v.dtype) for k, v in self._buf.items()}
            self._hmax = torch.zeros(self._max.shape, dtype=torch.float32)

        def buf(self, kind, side):
            return self._host[(kind, side)]

        def pack(self, which):
            super().pack(which)
            self.h.sync()
            for side in (0, 1):
                self._host[("send", side)].copy_(self._buf[("send", side)])

        def unpack(self, which):
            for side in (0, 1):
                self._buf[("recv", side)].copy_(self._host[("recv", side)])
            torch.cuda.current_stream().synchronize()
            super().unpack(which)

        def max_tensor(self):                     # SlabRing all-reduces this in place just before clock_end
            self.h.sync()
            self._hmax.copy_(self._max)
            return self._hmax

        def clock_end(self):
            self._max.copy_(self._hmax)
            torch.cuda.current_stream().synchronize()
            super().clock_end()

    nx, ny, nz = shape
    L = f.load()
    params = f.Tau3DParams()
    L.tau3d_params_default(ctypes.byref(params), nx, ny, nz)
    z0, nzl = slab.slab_bounds(nz, world, rank)
    be = Staged(f.taueng, params, z0, nzl, 0)
    be.h.init(1)
    be.h.set_clock(0.02, 1e-4)
    ring = slab.SlabRing(be, rank, world)
    ring.prime()
    ring.step(steps)
    ring.finish()
    c = be.clock()
    st = be.h.download()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), z0=z0, nzl=nzl, t=c.t, d_tau=c.d_tau, maxs=c.maxs,
             **{f"f{k}": st[k] for k in range(6)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,steps", [(2, (64, 48, 40), 6), (4, (48, 32, 64), 5), (3, (40, 40, 50), 4)])
def test_engine_ring_equals_single_domain(eng, tmp_path, world, shape, steps):
    mp.spawn(_worker, args=(world, _free_port(), shape, steps, str(tmp_path)), nprocs=world, join=True)
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    c = e.step(steps)
    want = e.download()
    got = [np.empty((nz, ny, nx), np.float32) for _ in range(6)]
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        z0, nzl = int(d["z0"]), int(d["nzl"])
        for k in range(6):
            got[k][z0:z0 + nzl] = d[f"f{k}"]
        assert float(d["t"]) == c.t and float(d["d_tau"]) == c.d_tau and float(d["maxs"]) == c.maxs
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    e.close()
