"""refcpu — THE REFERENCE'S OWN HOST CODE, compiled by oracle/build_ref.sh from pure line cuts of the sources under
/root/reference (no stand-in header, nothing written in place of what is cut away).  TEST INFRASTRUCTURE ONLY: only tests/ and
scripts that write fixtures may import this module; the product path never does.

  RefHypCpu      tau_hypersonic.c:1-674 / tau_hypersonic_simd.c:1-804 minus the raylib include — the CPU 2D solver of BASELINE
                 config C1 with its static init_sim :450 / compute_dt :477 / step_physics :500 (simd: :441 / :556 / :639)
  slice_to_rgba  tau_hypersonic_3d_cuda.cu:1410-1442 (clamp01, safe_log1p, slice_to_rgba)
  th3cs_palette  th3cs.cu:1199-1222 (the frame loop's schlieren -> palette-index lines)

The grid size of the CPU solver is a compile-time #define of the reference: one library per size (build_ref.sh lists them).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
CPU_SIZES = ((300, 300), (256, 256), (96, 64))


def _cpu_path(W, H, simd):
    return os.path.join(REF_DIR, f"libref_hyp_cpu{'_simd' if simd else ''}_{W}x{H}.so")


def available_cpu(W=300, H=300, simd=False):
    return os.path.exists(_cpu_path(W, H, simd))


def available_hostmaps():
    return os.path.exists(os.path.join(REF_DIR, "libref_hostmaps.so"))


class RefHypCpu:
    """the reference solver itself; its state lives in the library's static arrays, so ONE instance per (size, simd)"""

    def __init__(self, W=300, H=300, simd=False):
        L = C.CDLL(_cpu_path(W, H, simd))
        assert (L.ref_w(), L.ref_h()) == (W, H)
        L.ref_compute_dt.restype = C.c_double
        L.ref_time.restype = C.c_double
        L.ref_step.argtypes = [C.c_int]
        L.ref_state.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_set_state.argtypes = [C.c_void_p, C.c_double]
        self.L, self.W, self.H, self.simd = L, W, H, simd
        L.ref_init()

    def init(self):
        self.L.ref_init()

    def step(self, n=1):
        self.L.ref_step(int(n))

    def compute_dt(self):
        return self.L.ref_compute_dt()

    @property
    def t(self):
        return self.L.ref_time()

    def state(self):
        """(H, W, 4) fp64 AoS rho, mx, my, E (Cons, tau_hypersonic.c:24-29) and the (H, W) u8 mask"""
        u = np.empty((self.H, self.W, 4), np.float64)
        m = np.empty((self.H, self.W), np.uint8)
        self.L.ref_state(u.ctypes.data, m.ctypes.data)
        return u, m

    def set_state(self, u, t):
        u = np.ascontiguousarray(u, np.float64)
        assert u.shape == (self.H, self.W, 4)
        self.L.ref_set_state(u.ctypes.data, float(t))


_hm = None


def _hostmaps():
    global _hm
    if _hm is None:
        _hm = C.CDLL(os.path.join(REF_DIR, "libref_hostmaps.so"))
        _hm.ref_slice_to_rgba.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        _hm.ref_th3cs_palette.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _hm


def slice_to_rgba(vol, zslice, log_scale, a_gain):
    """tau_hypersonic_3d_cuda.cu:1416-1442 on a (nz, ny, nx) fp32 volume -> (ny, nx, 4) bytes r, g, b, a (little-endian words
    A<<24 | c<<16 | c<<8 | c)"""
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    out = np.empty((ny, nx), np.uint32)
    _hostmaps().ref_slice_to_rgba(out.ctypes.data, vol.ctypes.data, nx, ny, nz, int(zslice), 1 if log_scale else 0, C.c_float(a_gain))
    return out.view(np.uint8).reshape(ny, nx, 4)


def th3cs_palette(sch, frame=0, frames=1):
    """th3cs.cu:1199-1222 on one frame's schlieren volume -> the uint64 palette indices it stores at [frame * N, (frame+1) * N)"""
    sch = np.ascontiguousarray(sch, np.float32)
    nz, ny, nx = sch.shape
    idx = np.zeros(frames * sch.size, np.uint64)
    _hostmaps().ref_th3cs_palette(sch.ctypes.data, nx, ny, nz, int(frame), idx.ctypes.data)
    return idx.reshape(frames, nz, ny, nx)
