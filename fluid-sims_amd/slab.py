"""Z-slab ring driver for the 3D hypersonic grid: one process per GPU, torch.distributed for the
exchange (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference is single-GPU (no NCCL/MPI anywhere); this layer is new design
(SURVEY §8e).  nz is split into `world` contiguous slabs; z is periodic
(tau_hypersonic_3d_cuda.cu:1029-1030) so the neighbours form a ring.  Per step and rank:

    clock_begin                          t *= exp(d_tau), dt, gain              (device)
    wait halos(n)                        sent while step n-1 computed its interior
    step edge planes [0,3) , [nzl-3,nzl) need the halos; produce next state's boundary planes
    isend/irecv next-state boundary planes -> neighbours' next-state halos  (comm stream, async)
    step interior planes [3, nzl-3)      overlaps the exchange
    all_reduce(MAX) of the max-wavespeed word (4 bytes)
    clock_end                            d_tau controller (device) + swap

Only 3 planes x 6 fields cross each link per step (18.9 MB at 512^2 planes); each direction of a
neighbour pair has its own xGMI link, so the exchange costs ~0.12 ms against multi-ms slab
compute and hides behind the interior launch.  No other collective is on the data path.

The driver is backend-agnostic: `backend` is the HIP engine handle on the GPU box
(EngineSlabBackend below).  The CPU gloo tests plug the oracle in from tests/ — the product
code in this file never touches the oracle.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def slab_bounds(nz, world, rank):
    """contiguous split of nz planes; every slab needs >= 6 planes (two 3-plane edges)"""
    base, rem = divmod(nz, world)
    z0 = rank * base + min(rank, rem)
    nzl = base + (1 if rank < rem else 0)
    if nzl < 6:
        raise ValueError(f"nz={nz} over {world} ranks leaves a {nzl}-plane slab; need >= 6")
    return z0, nzl


class _DevMem:
    """expose a raw device pointer to torch through __cuda_array_interface__"""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class EngineSlabBackend:
    """The HIP engine (libtaueng) as the slab stepper: halo / max words are aliased as torch
    tensors so torch.distributed can move them; all launches go to torch's current stream."""

    def __init__(self, taueng, params, z0, nzl, device):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        # one explicit (non-default) stream shared by the engine launches, torch copies and the
        # RCCL hand-offs; the default stream's handle is NULL, which the C-ABI reads as "make your own"
        self.stream = torch.cuda.Stream(self.dev)
        torch.cuda.set_stream(self.stream)
        assert self.stream.cuda_stream != 0
        self.h = taueng.Tau3D(params.nx, params.ny, params.nz, params=params, z0=z0, nzl=nzl, device=device,
                              stream=C.c_void_p(self.stream.cuda_stream))
        self.nzl = nzl
        n = 3 * params.ny * params.nx
        self._n = n
        self._t = {}
        for which in (0, 1):
            for f in range(6):
                for side in (0, 1):
                    for kind in ("send", "recv"):
                        p = self.h.halo_ptr(kind, which, f, side)
                        self._t[(kind, which, f, side)] = torch.as_tensor(_DevMem(p, (n,)), device=self.dev)
        self._max = torch.as_tensor(_DevMem(self.h.max_ptr(), (1,)), device=self.dev)
        self._flip = 0

    # the engine swaps its ping-pong sides in clock_end; `which` is relative to the CURRENT side
    def halo_tensor(self, kind, which, field, side):
        return self._t[(kind, which ^ self._flip, field, side)]

    def max_tensor(self):
        return self._max

    def clock_begin(self):
        self.h.clock_begin_async()

    def step_range(self, lo, hi):
        self.h.step_range_async(lo, hi)

    def clock_end(self):
        self.h.clock_end_async()
        self._flip ^= 1

    def sync(self):
        self.h.sync()

    def clock(self):
        return self.h.clock()


class SlabRing:
    """Steps a Z-slab with ring halo exchange; `backend` implements the five calls above."""

    def __init__(self, backend, rank, world, group=None):
        self.b, self.rank, self.world, self.group = backend, rank, world, group
        self.lo = (rank - 1) % world
        self.hi = (rank + 1) % world
        self._pending = []

    # ---- exchange of the boundary planes of state `which` (0 = current, 1 = next)
    def _post_exchange(self, which):
        if self.world == 1:
            # periodic self-neighbour: plain device copies, no communicator involved
            for f in range(6):
                self.b.halo_tensor("recv", which, f, 1).copy_(self.b.halo_tensor("send", which, f, 0))
                self.b.halo_tensor("recv", which, f, 0).copy_(self.b.halo_tensor("send", which, f, 1))
            return []
        ops = []
        for f in range(6):
            # my low-z interior planes -> low neighbour's high halo; my high planes -> high neighbour's low halo
            ops.append(dist.P2POp(dist.isend, self.b.halo_tensor("send", which, f, 0), self.lo, self.group, tag=f))
            ops.append(dist.P2POp(dist.isend, self.b.halo_tensor("send", which, f, 1), self.hi, self.group, tag=6 + f))
        for f in range(6):
            ops.append(dist.P2POp(dist.irecv, self.b.halo_tensor("recv", which, f, 1), self.hi, self.group, tag=f))
            ops.append(dist.P2POp(dist.irecv, self.b.halo_tensor("recv", which, f, 0), self.lo, self.group, tag=6 + f))
        if self.world == 2:
            # both neighbours are the same peer: order the ops so sends/recvs pair up by tag
            pass
        return dist.batch_isend_irecv(ops)

    def _wait(self):
        for r in self._pending:
            r.wait()
        self._pending = []

    def prime(self):
        """exchange the halos of the current state (after init / upload)"""
        self._pending = self._post_exchange(0)
        self._wait()

    def step(self, n=1):
        b, nzl = self.b, self.b.nzl
        for _ in range(n):
            b.clock_begin()
            self._wait()                              # halos of the current state have landed
            b.step_range(0, 3)
            b.step_range(nzl - 3, nzl)
            self._pending = self._post_exchange(1)    # next state's boundary planes, async
            if nzl > 6:
                b.step_range(3, nzl - 3)              # overlaps the exchange
            if self.world > 1:
                dist.all_reduce(b.max_tensor(), op=dist.ReduceOp.MAX, group=self.group)
            b.clock_end()
        return self

    def finish(self):
        self._wait()
        self.b.sync()
