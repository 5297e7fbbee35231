"""CPU slab backend for the gloo tests: the oracle stands in for the HIP engine behind the same
calls (buf / pack / unpack / max_tensor / begin / edges / interior / end).  Lives in tests/ —
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
        self._end_pending = False
        self.plane = params.nx * params.ny
        self._xn = {(k, sd): np.zeros((6, 3 * self.plane), np.float32) for k in ("send", "recv") for sd in (0, 1)}
        self._x = {key: torch.from_numpy(a.reshape(-1)) for key, a in self._xn.items()}

    def init(self, mode):
        self.cur = self.o.init(mode)

    def buf(self, kind, side):
        return self._x[(kind, side)]

    def pack(self, which):
        st, n = (self.cur if which == 0 else self.nxt), self.nzl
        for f in range(6):
            self._xn[("send", 0)][f] = st[f][3:6].reshape(-1)
            self._xn[("send", 1)][f] = st[f][n:n + 3].reshape(-1)

    def unpack(self, which):
        st, n = (self.cur if which == 0 else self.nxt), self.nzl
        for f in range(6):
            st[f][0:3] = self._xn[("recv", 0)][f].reshape(3, *st[f].shape[1:])
            st[f][n + 3:n + 6] = self._xn[("recv", 1)][f].reshape(3, *st[f].shape[1:])

    def max_tensor(self):
        return self._max

    def _flush(self):
        if self._end_pending:                       # d_tau controller of the previous step, with the all-reduced max
            self.o.clock_end(float(self._max[0]))
            self._end_pending = False

    def begin(self):
        self._flush()
        self.o.clock_begin()
        self._max.zero_()
        self.unpack(0)

    def step_range(self, lo, hi):
        c = self.o.clock
        m = self.o.step_range(self.cur, self.nxt, c.dt, c.gain, lo, hi)
        self._max[0] = max(float(self._max[0]), m)

    def edges(self, depth):
        n = self.nzl
        if 2 * depth >= n:
            self.step_range(0, n)
        else:
            self.step_range(0, depth)
            self.step_range(n - depth, n)
        self.pack(1)

    def interior(self, depth):
        if 2 * depth < self.nzl:
            self.step_range(depth, self.nzl - depth)

    def end(self):
        self.cur, self.nxt = self.nxt, self.cur
        self._end_pending = True

    def sync(self):
        pass

    def clock(self):
        self._flush()
        return self.o.clock
