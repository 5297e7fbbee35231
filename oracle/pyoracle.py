"""pyoracle — ctypes access to the CPU oracles.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (fluid-sims_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force=False):
    """Compile the oracle shared objects with gcc (IEEE, no FMA contraction)."""
    args = ["make", "-C", _HERE] + (["-B"] if force else [])
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)


def _lib(name):
    p = os.path.join(_BUILD, name)
    if not os.path.exists(p):
        build()
    return C.CDLL(p)


class P3(C.Structure):
    _fields_ = ([("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32)] +
                [(n, C.c_float) for n in
                 "dx dy dz cfl u_ref R gamma_floor Twall tau_vib theta_v sdf_cx sdf_cy sdf_cz sdf_r "
                 "inflow_r inflow_p inflow_u inflow_v inflow_w".split()] +
                [("sponge_n", C.c_int32), ("sponge_strength", C.c_float),
                 ("sponge_out_n", C.c_int32), ("sponge_out_strength", C.c_float)])


class Clock(C.Structure):
    _fields_ = [(n, C.c_float) for n in "t d_tau dt gain maxs".split()] + [("step", C.c_int32)]


class GSParams(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32)] + [(n, C.c_float) for n in "dx dt Du Dv feed kill".split()]


class LapParams(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32)] + [(n, C.c_float) for n in "dx dy nu dt u0".split()]


HALO = 3


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _ptrs(arrs):
    return (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])


class Oracle3D:
    """3D hypersonic oracle on a Z-slab in halo layout: arrays of shape (nzl+6, ny, nx)."""

    def __init__(self, nx, ny=None, nz=None, z0=0, nzl=None, params=None, lib="libtauoracle3d.so"):
        self.L = _lib(lib)   # "libtauoracle3d_fma.so": the same source with FMA contraction on (tests/test_oracle_spread.py)
        self.L.o3_clock_end.argtypes = [C.POINTER(Clock), C.c_float, C.c_float]
        self.L.o3_step.restype = C.c_float
        self.L.o3_step.argtypes = [C.POINTER(P3), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_float, C.c_float]
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        if params is None:
            params = P3()
            self.L.o3_params_default(C.byref(params), nx, ny, nz)
        self.p = params
        self.z0 = z0
        self.nzl = self.p.nz if nzl is None else nzl
        self.shape_h = (self.nzl + 2 * HALO, self.p.ny, self.p.nx)
        self.solid = np.zeros(self.shape_h, np.uint8)
        self.L.o3_build_solid(C.byref(self.p), z0, self.nzl, _vp(self.solid))
        self.clock = Clock()
        self.L.o3_clock_reset(C.byref(self.clock))

    def new_state(self):
        return [np.zeros(self.shape_h, np.float32) for _ in range(6)]

    def init(self, mode=0):
        st = self.new_state()
        fn = self.L.o3_init_impulsive if mode else self.L.o3_init
        fn(C.byref(self.p), self.nzl, _vp(self.solid), _ptrs(st))
        self.L.o3_clock_reset(C.byref(self.clock))
        return st

    def fill_halo_periodic(self, st):
        self.L.o3_fill_halo_periodic(C.byref(self.p), self.nzl, _ptrs(st))

    def step_range(self, st_in, st_out, dt, gain, lo=0, hi=None):
        hi = self.nzl if hi is None else hi
        return self.L.o3_step(C.byref(self.p), self.z0, self.nzl, lo, hi, _ptrs(st_in), _ptrs(st_out),
                              _vp(self.solid), dt, gain)

    def clock_begin(self):
        self.L.o3_clock_begin(C.byref(self.clock))

    def clock_end(self, maxs):
        self.L.o3_clock_end(C.byref(self.clock), self.p.cfl, maxs)

    def run(self, st, nsteps):
        """single-domain full steps (controller + k_step + swap); returns the final state"""
        other = self.new_state()
        self.L.o3_run(C.byref(self.p), _ptrs(st), _ptrs(other), _vp(self.solid), C.byref(self.clock), nsteps)
        return st if nsteps % 2 == 0 else other

    def vis(self, st, mode):
        """k_vis over the interior planes; the halo planes of st must be current"""
        out = np.empty((self.nzl, self.p.ny, self.p.nx), np.float32)
        scale = np.empty_like(out)
        self.L.o3_vis(C.byref(self.p), self.z0, self.nzl, _ptrs(st), _vp(self.solid), int(mode), _vp(out), _vp(scale))
        return out, scale

    def slice_rgba(self, vol, zslice, log_scale=False, a_gain=1.0):
        vol = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = vol.shape
        px = np.empty((ny, nx), np.uint32)
        mn, mx = C.c_float(), C.c_float()
        self.L.o3_slice_to_rgba(_vp(px), _vp(vol), nx, ny, nz, int(zslice), int(bool(log_scale)), C.c_float(a_gain),
                                C.byref(mn), C.byref(mx))
        return px.view(np.uint8).reshape(ny, nx, 4), mn.value, mx.value

    def palette_indices(self, vol, gamma=0.65):
        """th3cs.cu:1199-1222: (uint8 indices, min, max) of a float volume"""
        vol = np.ascontiguousarray(vol, np.float32)
        out = np.empty(vol.shape, np.uint8)
        mn, mx = C.c_float(), C.c_float()
        self.L.o3_palette_indices.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.POINTER(C.c_float),
                                              C.POINTER(C.c_float)]
        self.L.o3_palette_indices(_vp(vol), vol.size, gamma, _vp(out), C.byref(mn), C.byref(mx))
        return out, mn.value, mx.value

    def outflow_reflection(self, st, nprobe=6):
        self.L.o3_outflow_reflection.restype = C.c_float
        return self.L.o3_outflow_reflection(C.byref(self.p), self.nzl, _ptrs(st), int(nprobe))

    @staticmethod
    def interior(st):
        return [a[HALO:-HALO] for a in st]

    def from_interior(self, fields):
        st = self.new_state()
        for a, f in zip(st, fields):
            a[HALO:-HALO] = np.asarray(f, np.float32).reshape(self.nzl, self.p.ny, self.p.nx)
        return st


class Oracle2D:
    def __init__(self):
        self.L = _lib("libtauoracle2d.so")

    def gs_params(self, nx, ny, **kw):
        p = GSParams()
        self.L.o2_gs_params_default(C.byref(p), nx, ny)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def gs_init(self, nx, ny, seed=1337):
        u = np.empty((ny, nx), np.float32)
        v = np.empty((ny, nx), np.float32)
        self.L.o2_gs_init(nx, ny, C.c_uint32(seed), _vp(u), _vp(v))
        return u, v

    def gs_step(self, p, u, v, nsteps=1):
        u = np.ascontiguousarray(u, np.float32).copy()
        v = np.ascontiguousarray(v, np.float32).copy()
        un, vn = np.empty_like(u), np.empty_like(v)
        for _ in range(nsteps):
            self.L.o2_gs_step(C.byref(p), _vp(u), _vp(v), _vp(un), _vp(vn))
            u, un = un, u
            v, vn = vn, v
        return u, v

    def lap_step(self, kind, p, a, b, npasses=1, oneD=False):
        a = np.ascontiguousarray(a, np.float32).copy()
        b = np.ascontiguousarray(b, np.float32).copy()
        an, bn = np.empty_like(a), np.empty_like(b)
        for _ in range(npasses):
            if kind == "burgers":
                self.L.o2_burgers_visc(C.byref(p), int(oneD), _vp(a), _vp(b), _vp(an), _vp(bn))
            else:
                self.L.o2_sw_visc(C.byref(p), _vp(a), _vp(b), _vp(an), _vp(bn))
            a, an = an, a
            b, bn = bn, b
        return a, b


class H2Params(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32)] + [(n, C.c_double) for n in
               "gamma cfl visc_nu visc_rho visc_e mach geom_x0 geom_cy geom_rb geom_rn geom_theta".split()]


class OracleH2:
    """fp64 oracle of the GPU 2D Euler scheme (tau_hypersonic_cuda.cu); state = 4 arrays (H, W)."""

    def __init__(self, W, H, **kw):
        self.L = _lib("libtauoracleh2.so")
        self.L.o2h_max_wavespeed.restype = C.c_double
        self.L.o2h_dt_from_maxs.restype = C.c_double
        self.L.o2h_dt_from_maxs.argtypes = [C.POINTER(H2Params), C.c_double]
        self.L.o2h_step_dt.argtypes = [C.POINTER(H2Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        self.L.o2h_unit_minmod.restype = C.c_double
        self.L.o2h_unit_minmod.argtypes = [C.c_double, C.c_double]
        self.L.o2h_unit_mc.restype = C.c_double
        self.L.o2h_unit_mc.argtypes = [C.c_double] * 3
        self.p = H2Params()
        self.L.o2h_params_default(C.byref(self.p), W, H)
        for k, v in kw.items():
            setattr(self.p, k, v)
        self.W, self.H = W, H
        self.mask = np.zeros((H, W), np.uint8)
        self.t = 0.0

    def _p4(self, st):
        return (C.c_void_p * 4)(*[a.ctypes.data for a in st])

    def init(self):
        st = [np.zeros((self.H, self.W)) for _ in range(4)]
        self.L.o2h_init(C.byref(self.p), _vp(st[0]), _vp(st[1]), _vp(st[2]), _vp(st[3]), _vp(self.mask))
        self.t = 0.0
        return st

    def apply_inflow(self, st):
        self.L.o2h_apply_inflow(C.byref(self.p), _vp(st[0]), _vp(st[1]), _vp(st[2]), _vp(st[3]), _vp(self.mask))

    def max_wavespeed(self, st):
        return self.L.o2h_max_wavespeed(C.byref(self.p), _vp(st[0]), _vp(st[1]), _vp(st[2]), _vp(st[3]), _vp(self.mask))

    def dt_from_maxs(self, maxs):
        return self.L.o2h_dt_from_maxs(C.byref(self.p), maxs)

    def step_dt(self, st, dt):
        """one step on a state that already carries the inflow column; returns the new state"""
        st = [np.ascontiguousarray(a, np.float64) for a in st]
        out = [np.empty_like(a) for a in st]
        self.L.o2h_step_dt(C.byref(self.p), self._p4(st), self._p4(out), _vp(self.mask), dt)
        return out

    def render(self, st, view_mode):
        st = [np.ascontiguousarray(a, np.float64) for a in st]
        val = np.empty((self.H, self.W))
        mn, mx = C.c_double(), C.c_double()
        self.L.o2h_render_vals(C.byref(self.p), _vp(st[0]), _vp(st[1]), _vp(st[2]), _vp(st[3]), _vp(self.mask),
                               int(view_mode), _vp(val), C.byref(mn), C.byref(mx))
        return val, mn.value, mx.value

    def render_pixels(self, val, vmin, vmax):
        val = np.ascontiguousarray(val, np.float64)
        px = np.empty((self.H, self.W), np.uint32)
        self.L.o2h_render_pixels(self.W, self.H, _vp(self.mask), _vp(val), C.c_double(vmin), C.c_double(vmax), _vp(px))
        return px.view(np.uint8).reshape(self.H, self.W, 4)

    def run(self, st, n):
        other = [np.empty_like(a) for a in st]
        t = C.c_double(self.t)
        self.L.o2h_run(C.byref(self.p), self._p4(st), self._p4(other), _vp(self.mask), n, C.byref(t))
        self.t = t.value
        return st if n % 2 == 0 else other


class SphParams(C.Structure):
    _fields_ = ([("N", C.c_int32)] + [(n, C.c_float) for n in
                "boxX boxY dTau t0 CFL rho0 c0 gammaEOS hMul viscAlpha gravity".split()] +
                [(n, C.c_int32) for n in "useVisc useGrav viscSub seed useXSPH".split()] +
                [("xsphEps", C.c_float), ("rain", C.c_int32)])


class OracleSph:
    """2D WCSPH oracle (tau_sph.cu), linked-list neighbour order = descending particle index."""

    def __init__(self, N, lib="libtauoraclesph.so", **kw):
        L = _lib(lib)        # "libtauoraclesph_asc.so": cell lists walked in ascending particle order
        L.osph_create.restype = C.c_void_p
        L.osph_create.argtypes = [C.POINTER(SphParams)]
        L.osph_destroy.argtypes = [C.c_void_p]
        L.osph_step.argtypes = [C.c_void_p, C.c_int]
        L.osph_substep.argtypes = [C.c_void_p, C.c_float]
        L.osph_dt.restype = C.c_float
        L.osph_dt.argtypes = [C.c_void_p]
        L.osph_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.osph_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.osph_grid.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.osph_get_clock.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.osph_get_accabs.argtypes = [C.c_void_p, C.c_void_p]
        self.L = L
        self.p = SphParams()
        L.osph_params_default(C.byref(self.p), N)
        for k, v in kw.items():
            setattr(self.p, k, v)
        self.N = N
        self.h = L.osph_create(C.byref(self.p))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.osph_destroy(self.h)
            self.h = None

    def grid(self):
        gx, gy = C.c_int(), C.c_int()
        cell, h, m = C.c_float(), C.c_float(), C.c_float()
        self.L.osph_grid(self.h, C.byref(gx), C.byref(gy), C.byref(cell), C.byref(h), C.byref(m))
        return {"Gx": gx.value, "Gy": gy.value, "cell": cell.value, "h": h.value, "mass": m.value}

    def rasterize(self, W, H):
        g = np.empty((2 * H, W), np.int32)
        self.L.osph_rasterize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.L.osph_rasterize(self.h, W, H, g.ctypes.data)
        return g

    def rain_spawned(self):
        self.L.osph_rain_spawned.restype = C.c_long
        self.L.osph_rain_spawned.argtypes = [C.c_void_p]
        return int(self.L.osph_rain_spawned(self.h))

    def set_state(self, pos, vel):
        pos = np.ascontiguousarray(pos, np.float32)
        vel = np.ascontiguousarray(vel, np.float32)
        self.L.osph_set_state(self.h, pos.ctypes.data, vel.ctypes.data)

    def state(self):
        n = self.N
        out = {k: np.empty((n, 2), np.float32) for k in ("pos", "vel", "acc")}
        out["s"] = np.empty(n, np.float32)
        out["press"] = np.empty(n, np.float32)
        out["cell"] = np.empty(n, np.int32)
        self.L.osph_get_state(self.h, out["pos"].ctypes.data, out["vel"].ctypes.data, out["acc"].ctypes.data,
                              out["s"].ctypes.data, out["press"].ctypes.data, out["cell"].ctypes.data)
        out["acc_abs"] = np.empty(n, np.float32)   # sum of |pair term|: the conditioning scale of acc
        self.L.osph_get_accabs(self.h, out["acc_abs"].ctypes.data)
        return out

    def dt(self):
        return self.L.osph_dt(self.h)

    def substep(self, dt):
        self.L.osph_substep(self.h, dt)

    def step(self, n=1):
        self.L.osph_step(self.h, n)

    def clock(self):
        t, tau, s = C.c_float(), C.c_float(), C.c_long()
        self.L.osph_get_clock(self.h, C.byref(t), C.byref(tau), C.byref(s))
        return {"t": t.value, "tau": tau.value, "step": s.value}


class FlowOParams(C.Structure):
    _fields_ = ([("nx", C.c_int32), ("ny", C.c_int32)] + [(n, C.c_float) for n in "dx dy nu u0 g CFL dtau".split()] +
                [(n, C.c_int32) for n in "muscl visc_substeps oneD".split()])


class OracleFlow:
    """Full Burgers / shallow-water step oracle (oracle/stencil2d_oracle.c, second half)."""

    def __init__(self, kind, nx, ny, **kw):
        L = _lib("libtauoracle2d.so")
        vp = C.c_void_p
        PP = C.POINTER(FlowOParams)
        L.o2_burgers_smax.restype = C.c_float
        L.o2_burgers_smax.argtypes = [PP, vp, vp]
        L.o2_burgers_step.argtypes = [PP, C.c_float, vp, vp, vp, vp]
        L.o2_burgers_init.argtypes = [PP, C.c_int, C.c_int] + [C.c_float] * 8 + [vp, vp]
        L.o2_burgers_colehopf_relL2.restype = C.c_double
        L.o2_burgers_colehopf_relL2.argtypes = [PP, C.c_int, C.c_float, vp, C.c_float]
        L.o2_sw_cmax.restype = C.c_float
        L.o2_sw_cmax.argtypes = [PP, vp, vp, vp]
        L.o2_sw_step.argtypes = [PP, C.c_float] + [vp] * 6
        L.o2_sw_init.argtypes = [PP] + [C.c_float] * 8 + [vp] * 3
        self.L, self.kind = L, kind
        d = dict(dx=1.0, dy=1.0, nu=0.1, u0=1.0, g=9.81, CFL=0.45, dtau=1.0, muscl=0, visc_substeps=1, oneD=0)
        if kind == "sw":
            d.update(nu=0.001, CFL=0.5)
        d.update(kw)
        if kind == "burgers" and d["oneD"]:
            ny = 1
        self.p = FlowOParams(nx, ny, d["dx"], d["dy"], d["nu"], d["u0"], d["g"], d["CFL"], d["dtau"], d["muscl"],
                             d["visc_substeps"], d["oneD"])
        self.shape = (ny, nx)

    def init_burgers(self, colehopf=0, ck=4, ca=0.5, amp=1.0, bsig=16.0, swirl=10.0, rc=40.0, offx=0.0, offy=0.0, asym=0.0):
        a, b = np.empty(self.shape, np.float32), np.empty(self.shape, np.float32)
        self.L.o2_burgers_init(C.byref(self.p), colehopf, ck, ca, amp, bsig, swirl, rc, offx, offy, asym, _vp(a), _vp(b))
        return [a, b]

    def init_sw(self, H0=1000.0, bumpAmp=1.0, bumpSigma=1.0, offx=100.0, offy=100.0, asym=10.0, swirl=1.0, swirlRc=100.0):
        f = [np.empty(self.shape, np.float32) for _ in range(3)]
        self.L.o2_sw_init(C.byref(self.p), H0, bumpAmp, bumpSigma, offx, offy, asym, swirl, swirlRc, _vp(f[0]), _vp(f[1]), _vp(f[2]))
        return f

    def metric(self, f):
        f = [np.ascontiguousarray(a, np.float32) for a in f]
        if self.kind == "burgers":
            return self.L.o2_burgers_smax(C.byref(self.p), _vp(f[0]), _vp(f[1]))
        return self.L.o2_sw_cmax(C.byref(self.p), _vp(f[0]), _vp(f[1]), _vp(f[2]))

    def dt_eff(self, f, t):
        """min(t*dtau, CFL*len/max) — tau_burgers.cu:693-694 / tau_shallow_water.cu:689-690"""
        m = np.float32(self.metric(f))
        length = np.float32(1.0) if self.kind == "burgers" else np.float32(min(self.p.dx, self.p.dy))
        return float(min(np.float32(t) * np.float32(self.p.dtau), np.float32(self.p.CFL) * length / m))

    def step(self, f, dt):
        f = [np.ascontiguousarray(a, np.float32) for a in f]
        out = [np.empty_like(a) for a in f]
        if self.kind == "burgers":
            self.L.o2_burgers_step(C.byref(self.p), dt, _vp(f[0]), _vp(f[1]), _vp(out[0]), _vp(out[1]))
        else:
            self.L.o2_sw_step(C.byref(self.p), dt, _vp(f[0]), _vp(f[1]), _vp(f[2]), _vp(out[0]), _vp(out[1]), _vp(out[2]))
        return out

    def colehopf_relL2(self, phu, ck, ca, t_now):
        phu = np.ascontiguousarray(phu, np.float32)
        return self.L.o2_burgers_colehopf_relL2(C.byref(self.p), ck, ca, _vp(phu), t_now)


class LbmParams(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("obstacle", C.c_int32), ("tau", C.c_float), ("drive", C.c_float),
                ("rho0", C.c_float), ("obstacle_radius", C.c_float)]


class OracleLbm:
    """D2Q9 BGK oracle (tau_lbm.cu): populations (9, ny, nx) float32, solid (ny, nx) uint8."""

    def __init__(self, nx=512, ny=256, **kw):
        self.L = _lib("libtauoraclelbm.so")
        self.p = LbmParams()
        self.L.olbm_params_default(C.byref(self.p))
        self.p.nx, self.p.ny = nx, ny
        for k, v in kw.items():
            setattr(self.p, k, v)
        self.solid = np.zeros((ny, nx), np.uint8)

    def init(self):
        f = np.empty((9, self.p.ny, self.p.nx), np.float32)
        self.L.olbm_init(C.byref(self.p), _vp(f), _vp(self.solid))
        return f

    def step(self, f, n=1):
        f = np.ascontiguousarray(f, np.float32)
        g = np.empty_like(f)
        for _ in range(n):
            self.L.olbm_step(C.byref(self.p), _vp(f), _vp(g), _vp(self.solid))
            f, g = g, f
        return f

    def speed(self, f):
        f = np.ascontiguousarray(f, np.float32)
        s = np.empty((self.p.ny, self.p.nx), np.float32)
        self.L.olbm_speed(C.byref(self.p), _vp(f), _vp(self.solid), _vp(s))
        return s
