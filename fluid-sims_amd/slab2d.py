"""Row-slab (y) decomposition of the periodic 5-point-stencil simulators — Gray-Scott and the Burgers /
shallow-water viscosity passes — over torch.distributed ("also slab-shardable, 1-row halo, periodic").

Rank r owns rows [y0, y0 + nyl) of the ny x nx grid and keeps H halo rows on each side: a local array of
(nyl + 2H) x nx, full width (x stays periodic and local).  The engine handle is simply created with
ny = nyl + 2H and steps that array as the periodic domain it believes it has.  What wraps around the ends of the
local array is wrong — but a 5-point stencil carries that one row per step, so after k <= H steps only the outer
k rows of each halo are contaminated and every owned row is exactly what the single-domain run computes
(same kernel, same operands: bit-identical).  Every H steps the halos are refreshed from the ring neighbours: the
first H owned rows go to the low neighbour's high halo, the last H owned rows to the high neighbour's low halo.
H = 4 matches the four time levels the engine fuses per pass, so one exchange per fused pass:
2 fields x H x nx x 4 B = 262 KB a side at nx = 8192.

No data-path collective; the path has no global reduction at all (fixed dt).  `bench.py` does not time this path
(BASELINE.json's metric is the 3D solver); `scripts/bench_secondary.py --gs-ring` reports it on one GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def row_bounds(ny, world, rank):
    """contiguous rows of rank `rank`: the first ny % world ranks get one more"""
    base, rem = divmod(ny, world)
    y0 = rank * base + min(rank, rem)
    return y0, base + (1 if rank < rem else 0)


def local_rows(field, y0, nyl, H):
    """(nyl + 2H, nx) local array of a global (ny, nx) field, halos filled periodically"""
    ny = field.shape[0]
    idx = (np.arange(y0 - H, y0 + nyl + H)) % ny
    return np.ascontiguousarray(field[idx])


class _DevMem:
    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class EngineRowBackend:
    """A taueng.GrayScott or taueng.Laplacian2D handle over the local (nyl + 2H) x nx array.  `make(ny_local,
    stream)` builds the handle; everything runs on ONE explicit stream that is also torch's current stream."""

    def __init__(self, make, nx, nyl, H, device=0):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.stream = torch.cuda.Stream(self.dev)
        torch.cuda.set_stream(self.stream)
        import ctypes
        self.h = make(nyl + 2 * H, ctypes.c_void_p(self.stream.cuda_stream))
        self.nx, self.nyl, self.H = nx, nyl, H
        self.buf = {(k, s): torch.empty(2 * H * nx, dtype=torch.float32, device=self.dev)
                    for k in ("send", "recv") for s in (0, 1)}

    def fields(self):
        """the current state as two (nyl + 2H, nx) tensors aliasing the engine's arrays (they swap every step)"""
        shp = (self.nyl + 2 * self.H, self.nx)
        return [torch.as_tensor(_DevMem(p, shp), device=self.dev) for p in self.h.state_ptrs()]

    def upload(self, a, b):
        self.h.upload(a, b)

    def download_owned(self):
        a, b = self.h.download()
        return a[self.H:self.H + self.nyl].copy(), b[self.H:self.H + self.nyl].copy()

    def step(self, n):
        self.h.step_async(n)

    def sync(self):
        self.h.sync()


class RowRing:
    """backend: nyl, H, nx, fields() -> two 2-D tensors of the current state, buf[(kind, side)] flat tensors of
    2 H nx floats, step(n), sync()."""

    def __init__(self, backend, rank, world, group=None):
        self.b, self.rank, self.world, self.group = backend, rank, world, group
        self.lo, self.hi = (rank - 1) % world, (rank + 1) % world
        if backend.nyl < backend.H:
            raise ValueError(f"slab of {backend.nyl} rows is thinner than the halo ({backend.H})")

    def exchange(self):
        """refresh both halos of the current state from the ring neighbours"""
        b, H, nyl, n = self.b, self.b.H, self.b.nyl, self.b.H * self.b.nx
        f = b.fields()
        for k, a in enumerate(f):
            b.buf[("send", 0)][k * n:(k + 1) * n].copy_(a[H:2 * H].reshape(-1))               # first owned rows
            b.buf[("send", 1)][k * n:(k + 1) * n].copy_(a[nyl:nyl + H].reshape(-1))           # last owned rows
        if self.world == 1:
            b.buf[("recv", 1)].copy_(b.buf[("send", 0)])
            b.buf[("recv", 0)].copy_(b.buf[("send", 1)])
        else:
            # same ordering rule as the 3D ring (world = 2: both neighbours are one peer, ops match in issue order)
            ops = [dist.P2POp(dist.isend, b.buf[("send", 0)], self.lo, self.group, tag=0),
                   dist.P2POp(dist.isend, b.buf[("send", 1)], self.hi, self.group, tag=1),
                   dist.P2POp(dist.irecv, b.buf[("recv", 1)], self.hi, self.group, tag=0),
                   dist.P2POp(dist.irecv, b.buf[("recv", 0)], self.lo, self.group, tag=1)]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for k, a in enumerate(f):
            a[0:H].reshape(-1).copy_(b.buf[("recv", 0)][k * n:(k + 1) * n])                   # low halo  <- low neighbour's last rows
            a[nyl + H:nyl + 2 * H].reshape(-1).copy_(b.buf[("recv", 1)][k * n:(k + 1) * n])   # high halo <- high neighbour's first rows

    def step(self, nsteps):
        """nsteps time steps (any count: the last stretch may be shorter than H)"""
        done = 0
        while done < nsteps:
            k = min(self.b.H, nsteps - done)
            self.b.step(k)
            self.exchange()
            done += k
        return self

    def finish(self):
        self.b.sync()
