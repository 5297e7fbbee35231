"""CPU slab backend for the gloo tests: the oracle stands in for the HIP engine behind the same
five calls (halo_tensor / max_tensor / clock_begin / step_range / clock_end).  Lives in tests/ —
the product package never imports the oracle."""
import numpy as np
import torch

from oracle import pyoracle


class OracleSlabBackend:
    def __init__(self, params, z0, nzl):
        self.o = pyoracle.Oracle3D(params.nx, params.ny, params.nz, z0=z0, nzl=nzl, params=params)
        self.nzl = nzl
        self.cur = self.o.new_state()
        self.nxt = self.o.new_state()
        self._max = torch.zeros(1, dtype=torch.float32)
        self.plane = params.nx * params.ny

    def init(self, mode):
        self.cur = self.o.init(mode)

    def halo_tensor(self, kind, which, field, side):
        a = (self.cur if which == 0 else self.nxt)[field]
        n = self.nzl
        if kind == "send":
            sl = a[3:6] if side == 0 else a[n:n + 3]
        else:
            sl = a[0:3] if side == 0 else a[n + 3:n + 6]
        return torch.from_numpy(sl.reshape(-1))   # view on the numpy buffer

    def max_tensor(self):
        return self._max

    def clock_begin(self):
        self.o.clock_begin()
        self._max.zero_()

    def step_range(self, lo, hi):
        c = self.o.clock
        m = self.o.step_range(self.cur, self.nxt, c.dt, c.gain, lo, hi)
        self._max[0] = max(float(self._max[0]), m)

    def clock_end(self):
        self.o.clock_end(float(self._max[0]))
        self.cur, self.nxt = self.nxt, self.cur

    def sync(self):
        pass

    def clock(self):
        return self.o.clock
