"""Z-slab ring driver for the 3D hypersonic grid: one process per GPU, torch.distributed for the
exchange (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference is single-GPU (no NCCL/MPI anywhere); this layer is new design
(SURVEY §8e).  nz is split into `world` contiguous slabs; z is periodic
(tau_hypersonic_3d_cuda.cu:1029-1030) so the neighbours form a ring.  Per step and rank:

    begin      ONE kernel: d_tau controller of step n-1 (it needs the all-reduced max word), clock of step n
               (t *= exp(d_tau), dt, gain), and the halos received while step n-1 computed its interior unpacked
    edges      edge planes [0,E) , [nzl-E,nzl): [large planes: their x/y flux kernel, then] one launch that steps both edges
               and also writes the new boundary planes into the packed send buffers
    isend/irecv  2 sends + 2 recvs of one packed buffer each (async)
    interior   planes [E, nzl-E) [x/y flux kernel + z kernel]: overlaps the exchange
    all_reduce(MAX) of the max-wavespeed and max-|primitive| words (8 bytes)
    end        swap (host bookkeeping; the controller rides on the next begin)

Five dispatches (four below 128^2 planes: one fused kernel per piece plus a pack kernel) and two collectives per step.  Only 3 planes x 6 fields cross each link per step (18.9 MB at 512^2 planes); each direction of a
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

    # ---- one step = begin, edges, <exchange posted>, interior, <all-reduce>, end  (include/taueng.h)
    def begin(self):
        self.h.slab_begin_async()

    def edges(self, depth):
        self.h.slab_edges_async(depth)

    def interior(self, depth):
        self.h.slab_interior_async(depth)

    def end(self):
        self.h.slab_end_async()

    def sync(self):
        self.h.sync()

    def clock(self):
        return self.h.clock()


class SlabRing:
    """Steps a Z-slab with ring halo exchange.  `backend` provides buf(kind, side) (flat tensors: the packed
    3 planes x 6 fields a side sends / receives), pack(which), unpack(which) (`which` = 0: current state),
    max_tensor(), sync(), clock() and the four pieces of a step:
        begin()      controller of the previous step (it needs the all-reduced max), clock of this one, received halos
                     unpacked into the current state
        edges(E)     planes [0,E) and [nzl-E,nzl) of the new state, boundary planes packed into the send buffers
        interior(E)  planes [E, nzl-E)
        end()        swap"""

    def __init__(self, backend, rank, world, group=None, self_p2p=False):
        # self_p2p: with world == 1, send / receive the halos to / from OURSELVES through torch.distributed (RCCL) instead of a
        # device copy — the N > 1 communication path (batched isend / irecv on the packed buffers, their stream ordering
        # against the step kernels) on a single GPU; the all-reduce of the max words runs too
        self.b, self.rank, self.world, self.group = backend, rank, world, group
        self.self_p2p = bool(self_p2p) and world == 1
        self.lo = (rank - 1) % world
        self.hi = (rank + 1) % world
        self._pending = None
        self._primed = False
        self.edge = max(3, min(8, backend.nzl // 2))   # planes per edge launch (>= the 3 halo planes, <= half a slab)

    def _post_exchange(self):
        b = self.b
        if self.world == 1 and not self.self_p2p:  # periodic self-neighbour: my low planes are my own high halo and vice versa
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

    def _wait(self):
        if self._pending is not None:
            for r in self._pending:
                r.wait()
            self._pending = None

    def prime(self):
        """exchange the halos of the current state and agree on its field range (after init / upload).  step() does it
        itself when the caller has not: without it the first step would read undefined halo planes and every slab could
        pick its own WENO weight form."""
        b = self.b
        b.pack(0)
        self._pending = self._post_exchange()
        self._wait()
        b.unpack(0)
        if self.world > 1 or self.self_p2p:                             # the field range init / upload measured, over all slabs
            dist.all_reduce(b.max_tensor(), op=dist.ReduceOp.MAX, group=self.group)
        self._primed = True                            # (the first begin() unpacks the same buffers once more: idempotent)

    def invalidate(self):
        """the backend's state was re-initialised or uploaded: exchange again before the next step"""
        self._primed = False

    def step(self, n=1):
        b = self.b
        if not self._primed:
            self.prime()
        E = self.edge
        for _ in range(n):
            self._wait()                               # the halos posted during the previous step have landed
            b.begin()
            # Edge launches are E planes deep, not just the 3 that are sent: a marching launch pays a warm-up per chunk,
            # so E = 8 keeps the edge launch at the duty of the interior one; the interior that hides the exchange is
            # still ~0.5 ms at 512^2 x 48.
            b.edges(E)
            self._pending = self._post_exchange()      # async; lands before the next begin()
            b.interior(E)                              # overlaps the exchange
            if self.world > 1 or self.self_p2p:
                dist.all_reduce(b.max_tensor(), op=dist.ReduceOp.MAX, group=self.group)
            b.end()
        return self

    def finish(self):
        self._wait()
        self.b.sync()
