"""CPU stand-in for the engine in the row-slab ring: the Gray-Scott / Laplacian oracles stepping the local
(nyl + 2H) x nx array as a periodic domain (tests only)."""
import numpy as np
import torch


class OracleRowBackend:
    def __init__(self, stepper, nx, nyl, H):
        """stepper(a, b, n) -> (a', b') on numpy arrays of the local shape, periodic"""
        self.stepper, self.nx, self.nyl, self.H = stepper, nx, nyl, H
        self.a = self.b_ = None
        self.buf = {(k, s): torch.empty(2 * H * nx, dtype=torch.float32) for k in ("send", "recv") for s in (0, 1)}

    def upload(self, a, b):
        self.a, self.b_ = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)

    def fields(self):
        return [torch.from_numpy(self.a), torch.from_numpy(self.b_)]      # share memory with the arrays

    def download_owned(self):
        return self.a[self.H:self.H + self.nyl].copy(), self.b_[self.H:self.H + self.nyl].copy()

    def step(self, n):
        a, b = self.stepper(self.a, self.b_, n)
        self.a, self.b_ = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)

    def sync(self):
        pass
