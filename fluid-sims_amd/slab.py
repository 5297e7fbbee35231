"""Z-slab ring driver for the 3D hypersonic grid: one process per GPU, torch.distributed for the
exchange (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference is single-GPU (no NCCL/MPI anywhere); this layer is new design
(SURVEY §8e).  nz is split into `world` contiguous slabs; z is periodic
(tau_hypersonic_3d_cuda.cu:1029-1030) so the neighbours form a ring.  Per step and rank:

    clock_begin                          t *= exp(d_tau), dt, gain              (device)
    wait + unpack halos(n)               received while step n-1 computed its interior
    step edge planes [0,3) , [nzl-3,nzl) need the halos; produce next state's boundary planes
    pack those planes, isend/irecv       2 sends + 2 recvs of one packed buffer each (comm stream, async)
    step interior planes [3, nzl-3)      overlaps the exchange
    all_reduce(MAX) of the max-wavespeed and max-|primitive| words (8 bytes)
    clock_end                            d_tau controller (device) + swap

Only 3 planes x 6 fields cross each link per step (18.9 MB at 512^2 planes); each direction of a
neighbour pair has its own xGMI link, so the exchange costs ~0.12 ms against multi-ms slab
compute and hides behind the interior launch.  No other collective is on the data path.

The driver is backend-agnostic: `backend` is the HIP engine handle on the GPU box
(EngineSlabBackend below).  The CPU gloo tests plug the oracle in from tests/ — the product
code in this file never touches the oracle.
"""
import ctypes as C

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
    """The HIP engine (libtaueng) as the slab stepper: the packed exchange buffers and the max word are
    aliased as torch tensors so torch.distributed can move them; all launches go to ONE explicit stream
    that is also torch's current stream (copies, RCCL hand-offs)."""

    def __init__(self, taueng, params, z0, nzl, device):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        # the default stream's handle is NULL, which the C-ABI reads as "make your own": use a real one
        self.stream = torch.cuda.Stream(self.dev)
        torch.cuda.set_stream(self.stream)
        assert self.stream.cuda_stream != 0
        self.h = taueng.Tau3D(params.nx, params.ny, params.nz, params=params, z0=z0, nzl=nzl, device=device,
                              stream=C.c_void_p(self.stream.cuda_stream))
        self.nzl = nzl
        self._buf = {}
        for kind in ("send", "recv"):
            for side in (0, 1):
                p, n = self.h.halo_buf(kind, side)
                self._buf[(kind, side)] = torch.as_tensor(_DevMem(p, (n,)), device=self.dev)
        self._max = torch.as_tensor(_DevMem(self.h.max_ptr(), (2,)), device=self.dev)   # max wavespeed, max |primitive|

    def buf(self, kind, side):
        return self._buf[(kind, side)]

    def pack(self, which):
        self.h.pack_halos_async(which)

    def unpack(self, which):
        self.h.unpack_halos_async(which)

    def max_tensor(self):
        return self._max

    def clock_begin(self):
        self.h.clock_begin_async()

    def step_range(self, lo, hi):
        self.h.step_range_async(lo, hi)

    def step_edges(self, depth):
        self.h.step_edges_async(depth)

    def clock_end(self):
        self.h.clock_end_async()

    def sync(self):
        self.h.sync()

    def clock(self):
        return self.h.clock()


class SlabRing:
    """Steps a Z-slab with ring halo exchange.  `backend` provides buf(kind, side) (flat tensors: the packed
    3 planes x 6 fields a side sends / receives), pack(which), unpack(which), max_tensor(), clock_begin(),
    step_range(lo, hi), clock_end(), sync(); `which` = 0 current state, 1 next state."""

    def __init__(self, backend, rank, world, group=None):
        self.b, self.rank, self.world, self.group = backend, rank, world, group
        self.lo = (rank - 1) % world
        self.hi = (rank + 1) % world
        self._pending = None
        self.edge = max(3, min(8, backend.nzl // 2))   # planes per edge launch (>= the 3 halo planes, <= half a slab)

    def _post_exchange(self):
        b = self.b
        if self.world == 1:  # periodic self-neighbour: my low planes are my own high halo and vice versa
            b.buf("recv", 1).copy_(b.buf("send", 0))
            b.buf("recv", 0).copy_(b.buf("send", 1))
            return []
        # low-z boundary planes -> low neighbour (they are ITS high halo); high planes -> high neighbour.
        # Order matters for RCCL when both neighbours are the same peer (world = 2): ops between one pair
        # match in issue order, so sends go (side 0, side 1) and receives (side 1, side 0).
        ops = [dist.P2POp(dist.isend, b.buf("send", 0), self.lo, self.group, tag=0),
               dist.P2POp(dist.isend, b.buf("send", 1), self.hi, self.group, tag=1),
               dist.P2POp(dist.irecv, b.buf("recv", 1), self.hi, self.group, tag=0),
               dist.P2POp(dist.irecv, b.buf("recv", 0), self.lo, self.group, tag=1)]
        return dist.batch_isend_irecv(ops)

    def _land(self, which):
        """wait for the exchange in flight and unpack it into the halos of state `which`"""
        if self._pending is None:
            return
        for r in self._pending:
            r.wait()
        self._pending = None
        self.b.unpack(which)

    def prime(self):
        """exchange the halos of the current state (after init / upload)"""
        self.b.pack(0)
        self._pending = self._post_exchange()
        self._land(0)
        if self.world > 1:                             # the field range init / upload measured, over all slabs
            dist.all_reduce(self.b.max_tensor(), op=dist.ReduceOp.MAX, group=self.group)

    def step(self, n=1):
        b, nzl = self.b, self.b.nzl
        for _ in range(n):
            b.clock_begin()
            self._land(0)                              # halos of the current state
            # Edge launches are E planes deep, not just the 3 that are sent: the z-marching kernel re-decodes 5
            # warm-up planes per chunk, so a 3-plane launch is 8 plane iterations for 3 useful ones.  With E = 8
            # a 64-plane slab costs 13 + 13 + 58 iterations instead of 8 + 8 + 98 — the same ~76 % duty as the
            # single-GPU launch — and the interior that hides the exchange is still ~0.6 ms at 512^2 x 48.
            E = self.edge
            if hasattr(b, "step_edges"):
                b.step_edges(E)                        # both edges in one launch (engine backend)
            else:
                b.step_range(0, E)
                b.step_range(nzl - E, nzl)
            b.pack(1)                                  # next state's boundary planes
            self._pending = self._post_exchange()      # async; lands at the start of the next step
            if nzl > 2 * E:
                b.step_range(E, nzl - E)               # overlaps the exchange
            if self.world > 1:
                dist.all_reduce(b.max_tensor(), op=dist.ReduceOp.MAX, group=self.group)
            b.clock_end()
        return self

    def finish(self):
        self._land(0)
        self.b.sync()
