"""GPU, multi-process: the Z-slab ring with the REAL HIP engine on every rank.  The GPU box has one device, so the
ranks share it and the transport is gloo with the packed halo buffers staged through host memory — everything else
is the production path (EngineSlabBackend: tau3d_slab_begin / edges / interior / end, the max word, the device
clock).  Result: bit-identical to the single-domain engine run."""
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
            self._hmax_valid = False

        def buf(self, kind, side):
            return self._host[(kind, side)]

        def _send_to_host(self):
            self.h.sync()
            for side in (0, 1):
                self._host[("send", side)].copy_(self._buf[("send", side)])

        def _recv_to_device(self):
            for side in (0, 1):
                self._buf[("recv", side)].copy_(self._host[("recv", side)])
            if self._hmax_valid:                     # the all-reduced max words go back before the controller reads them
                self._max.copy_(self._hmax)
                self._hmax_valid = False
            torch.cuda.current_stream().synchronize()

        def pack(self, which):
            super().pack(which)
            self._send_to_host()

        def unpack(self, which):
            self._recv_to_device()
            super().unpack(which)

        def begin(self):
            self._recv_to_device()
            super().begin()

        def edges(self, depth):
            super().edges(depth)                      # writes the send buffers itself
            self._send_to_host()

        def max_tensor(self):                     # SlabRing all-reduces this in place
            self.h.sync()
            self._hmax.copy_(self._max)
            self._hmax_valid = True
            return self._hmax

        def clock(self):
            if self._hmax_valid:
                self._max.copy_(self._hmax)
                self._hmax_valid = False
                torch.cuda.current_stream().synchronize()
            return super().clock()

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


@pytest.mark.parametrize("world,shape,steps", [(2, (64, 48, 40), 6), (4, (48, 32, 64), 5), (3, (40, 40, 50), 4),
                                               (8, (32, 32, 64), 4),        # BASELINE's world size: 8 planes per rank, E = 4
                                               (2, (160, 128, 24), 3),      # planes >= 128^2: the split step (k_flux_xy + k_update_z), whole slab in the edges piece
                                               (2, (128, 128, 40), 3)])     # split step with an interior piece: 20 planes per rank, E = 8
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
