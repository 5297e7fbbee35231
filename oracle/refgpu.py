"""refgpu — runs THE REFERENCE'S OWN KERNELS (oracle/_ref/*.co, built by oracle/build_ref.sh from the sources under
/root/reference) on the MI355X.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and scripts/ that write evidence under profiles/ may import this module; the product
path (fluid-sims_amd/) never does.  The code objects hold the reference's device code unchanged; what this file restates
is only what the reference's `main`s do around the launches — launch shapes, the order of launches, the handful of host
lines between them — each with its file:line.

Plain ctypes on the HIP runtime: hipModuleLoad / hipModuleGetFunction / hipModuleGetGlobal / hipModuleLaunchKernel.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

_hip = None
_libm = None


class RefError(RuntimeError):
    pass


def available(name="th3cs"):
    return os.path.exists(os.path.join(REF_DIR, name + ".co"))


def hip():
    """the HIP runtime the process already holds (PyTorch's when torch is importable, so libtaueng shares it)"""
    global _hip
    if _hip is None:
        import fluid_sims_amd as f
        _hip = f.taueng._load_hip_runtime() if hasattr(f, "taueng") else None
        if _hip is None:  # pragma: no cover
            _hip = C.CDLL("/opt/rocm/lib/libamdhip64.so", mode=C.RTLD_GLOBAL)
        _hip.hipGetErrorString.restype = C.c_char_p
    return _hip


def libm():
    global _libm
    if _libm is None:
        _libm = C.CDLL("libm.so.6")
        for n in ("expf", "logf", "sqrtf", "powf"):
            getattr(_libm, n).restype = C.c_float
        _libm.expf.argtypes = _libm.logf.argtypes = _libm.sqrtf.argtypes = [C.c_float]
    return _libm


def ck(e, what=""):
    if e != 0:
        raise RefError(f"{what}: HIP error {e}: {hip().hipGetErrorString(e).decode()}")


class DevBuf:
    """a device allocation with numpy up/download"""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = C.c_void_p()
        ck(hip().hipMalloc(C.byref(self.ptr), C.c_size_t(max(self.nbytes, 4))), "hipMalloc")

    @classmethod
    def like(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.put(a)
        return b

    def put(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        ck(hip().hipMemcpy(self.ptr, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1), "hipMemcpy H2D")

    def get(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        ck(hip().hipMemcpy(C.c_void_p(out.ctypes.data), self.ptr, C.c_size_t(out.nbytes), 2), "hipMemcpy D2H")
        return out

    def zero(self):
        ck(hip().hipMemset(self.ptr, 0, C.c_size_t(self.nbytes)), "hipMemset")

    def copy_from_device(self, src_ptr, nbytes=None):
        ck(hip().hipMemcpy(self.ptr, C.c_void_p(src_ptr), C.c_size_t(self.nbytes if nbytes is None else nbytes), 3), "hipMemcpy D2D")

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    __del__ = free


class RefModule:
    """one reference translation unit's device code"""

    def __init__(self, name):
        p = os.path.join(REF_DIR, name + ".co")
        if not os.path.exists(p):
            raise RefError(f"{p} not found — oracle/build_ref.sh builds it where /root/reference exists")
        self.name = name
        self.syms = {}
        with open(os.path.join(REF_DIR, name + ".syms")) as fh:
            for line in fh:
                m, d = line.rstrip("\n").split("\t")
                self.syms[d] = m
        self.mod = C.c_void_p()
        ck(hip().hipModuleLoad(C.byref(self.mod), p.encode()), f"hipModuleLoad({name})")
        self._fn = {}

    def mangled(self, short):
        """a kernel by its plain name, e.g. 'k_step' -> the one symbol whose demangled form starts with 'k_step('"""
        hits = [m for d, m in self.syms.items() if d == short or d.startswith(short + "(")]
        if len(hits) != 1:
            raise RefError(f"{self.name}: {len(hits)} symbols match {short!r}")
        return hits[0]

    def function(self, short):
        if short not in self._fn:
            f = C.c_void_p()
            ck(hip().hipModuleGetFunction(C.byref(f), self.mod, self.mangled(short).encode()), f"hipModuleGetFunction({short})")
            self._fn[short] = f
        return self._fn[short]

    def set_global(self, short, struct):
        """cudaMemcpyToSymbol(sym, &host, sizeof) of the reference's mains"""
        dptr, size = C.c_void_p(), C.c_size_t()
        ck(hip().hipModuleGetGlobal(C.byref(dptr), C.byref(size), self.mod, self.mangled(short).encode()), f"hipModuleGetGlobal({short})")
        if size.value != C.sizeof(struct):
            raise RefError(f"{self.name}:{short} is {size.value} bytes on the device, {C.sizeof(struct)} here")
        ck(hip().hipMemcpy(dptr, C.byref(struct), size, 1), "hipMemcpy to symbol")

    def launch(self, short, grid, block, args, shmem=0):
        """kernel<<<grid, block, shmem>>>(args...) on the null stream; args are ctypes values (c_void_p, c_int, c_float,
        c_double, c_bool or a Structure passed by value)"""
        holders = [a if isinstance(a, (C._SimpleCData, C.Structure)) else C.c_void_p(a) for a in args]
        params = (C.c_void_p * len(holders))(*[C.cast(C.pointer(h), C.c_void_p) for h in holders])
        g = tuple(grid) + (1,) * (3 - len(grid))
        b = tuple(block) + (1,) * (3 - len(block))
        ck(hip().hipModuleLaunchKernel(self.function(short), g[0], g[1], g[2], b[0], b[1], b[2], C.c_uint(shmem), None, params, None),
           f"launch {short}")

    def sync(self):
        ck(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")


def _p(buf):
    return C.c_void_p(buf.ptr.value)


# ------------------------------------------------------------------------------------------------------------------
# 3D hypersonic: the device code of tau_hypersonic_3d_cuda.cu (the file north_star names) and of th3cs.cu (its headless twin)
# ------------------------------------------------------------------------------------------------------------------
class Params3D(C.Structure):
    """th3cs.cu:69-90 == tau_hypersonic_3d_cuda.cu:21-42"""
    _fields_ = ([("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32)] +
                [(n, C.c_float) for n in
                 "dx dy dz cfl u_ref R gamma_floor Twall tau_vib theta_v sdf_cx sdf_cy sdf_cz sdf_r "
                 "inflow_r inflow_p inflow_u inflow_v inflow_w".split()] +
                [("sponge_n", C.c_int32), ("sponge_strength", C.c_float),
                 ("sponge_out_n", C.c_int32), ("sponge_out_strength", C.c_float)])


def params3d_default(nx, ny, nz):
    """the literals of th3cs.cu:1063-1089 == tau_hypersonic_3d_cuda.cu:1531-1557"""
    f = np.float32
    p = Params3D()
    p.nx, p.ny, p.nz = nx, ny, nz
    p.dx, p.dy, p.dz = f(1.0) / f(nx), f(1.0) / f(ny), f(1.0) / f(nz)
    p.cfl, p.u_ref, p.R, p.gamma_floor, p.Twall, p.tau_vib, p.theta_v = 0.3333, 10.0, 10.0, 1.1, 0.02, 2e-4, 0.2
    p.sdf_cx = p.sdf_cy = p.sdf_cz = 0.5
    p.sdf_r = 0.25
    p.inflow_r = p.inflow_p = 0.02
    p.inflow_u, p.inflow_v, p.inflow_w = 100.0, 0.0, 0.0
    p.sponge_n, p.sponge_strength, p.sponge_out_n, p.sponge_out_strength = 24, 0.05, 24, 0.05
    return p


class Ref3D:
    """th3cs.cu:1062-1130 (setup) and :1157-1196 (the step loop with the log-time controller) around the reference's own
    k_build_solid_mask / k_init / k_step, launched as the reference launches them: block (8,8,4), grid = ceil, dynamic LDS
    (bx+6)(by+6)(bz+6)·25 B."""
    WENO_HALO = 3
    BLOCK = (8, 8, 4)

    SOURCES = {"th3cs": "th3cs", "3d_cuda": "tau_hypersonic_3d_cuda"}

    def __init__(self, nx, ny=None, nz=None, params=None, ieee=False, source=None):
        """source: "3d_cuda" = tau_hypersonic_3d_cuda.cu, the file north_star names (oracle/build_ref.sh's line cut 1-1409 minus
        the raylib includes and the Vector3 helpers: k_step :987-1359, k_init :939-985, k_build_solid_mask :759-770, k_vis
        :800-905, k_maxwavespeed_pre :909-937, k_schlieren :1361-1387, k_outflow_reflection_metric :1389-1408); "th3cs" = its
        headless twin (adds k_schlieren_export).  Default: the named file when its code object is there, else the twin."""
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        if source is None:
            source = "3d_cuda" if available("tau_hypersonic_3d_cuda") else "th3cs"
        self.source = source
        stem = self.SOURCES[source]
        self.m = RefModule(stem + ".ieee" if ieee else stem)
        self.p = params3d_default(nx, ny, nz) if params is None else params
        self.m.set_global("P", self.p)
        self.shape = (self.p.nz, self.p.ny, self.p.nx)
        self.N = self.p.nx * self.p.ny * self.p.nz
        self.a = [DevBuf(4 * self.N) for _ in range(6)]
        self.b = [DevBuf(4 * self.N) for _ in range(6)]
        self.solid = DevBuf(self.N)
        self.maxs = DevBuf(4)
        bx, by, bz = self.BLOCK
        self.grid = ((self.p.nx + bx - 1) // bx, (self.p.ny + by - 1) // by, (self.p.nz + bz - 1) // bz)
        h = self.WENO_HALO
        self.smem = (bx + 2 * h) * (by + 2 * h) * (bz + 2 * h) * (6 * 4 + 1)
        self.t = np.float32(1e-5)       # th3cs.cu:1152-1153 == tau_hypersonic_3d_cuda.cu:1635-1636
        self.d_tau = np.float32(1e-3)
        self.m.launch("k_build_solid_mask", self.grid, self.BLOCK, [_p(self.solid)])
        self.m.sync()

    def close(self):
        for b in self.a + self.b + [self.solid, self.maxs]:
            b.free()

    def init(self):
        self.m.launch("k_init", self.grid, self.BLOCK, [_p(x) for x in self.a] + [_p(self.solid)])
        self.m.sync()

    def solid_mask(self):
        return self.solid.get(np.uint8, self.shape)

    def upload(self, fields):
        for buf, f in zip(self.a, fields):
            buf.put(np.asarray(f, np.float32))

    def upload_from_device(self, ptrs):
        for buf, p in zip(self.a, ptrs):
            buf.copy_from_device(p)

    def download(self):
        return [b.get(np.float32, self.shape) for b in self.a]

    def step(self, dt, gain):
        """one k_step launch with the given dt and inflow gain; returns the max wavespeed it wrote; state swapped"""
        self.maxs.put(np.zeros(1, np.float32))
        self.m.launch("k_step", self.grid, self.BLOCK,
                      [_p(x) for x in self.a] + [_p(x) for x in self.b] + [C.c_float(dt), C.c_float(gain), _p(self.maxs), _p(self.solid)],
                      shmem=self.smem)
        self.m.sync()
        self.a, self.b = self.b, self.a
        return float(self.maxs.get(np.float32, (1,))[0])

    def vis(self, mode):
        """k_vis (tau_hypersonic_3d_cuda.cu:800-905, launched :1715-1716) of the current state, mode 0..7 (VisMode :784-794)"""
        out = DevBuf(4 * self.N)
        self.m.launch("k_vis", self.grid, self.BLOCK, [_p(x) for x in self.a] + [_p(self.solid), _p(out), C.c_int(int(mode))])
        self.m.sync()
        v = out.get(np.float32, self.shape)
        out.free()
        return v

    def schlieren_xi(self):
        """k_schlieren (tau_hypersonic_3d_cuda.cu:1361-1387): |grad rho| from xi alone, no solid handling"""
        out = DevBuf(4 * self.N)
        self.m.launch("k_schlieren", self.grid, self.BLOCK, [_p(self.a[0]), _p(out)])
        self.m.sync()
        v = out.get(np.float32, self.shape)
        out.free()
        return v

    def outflow_reflection(self, nprobe=6):
        """k_outflow_reflection_metric (tau_hypersonic_3d_cuda.cu:1389-1408) as the frame loop launches it (:1723-1733, nprobe 6)"""
        self.maxs.put(np.zeros(1, np.float32))
        self.m.launch("k_outflow_reflection_metric", self.grid, self.BLOCK, [_p(x) for x in self.a] + [_p(self.maxs), C.c_int(int(nprobe))])
        self.m.sync()
        return float(self.maxs.get(np.float32, (1,))[0])

    def maxwavespeed_pre(self):
        """k_maxwavespeed_pre (tau_hypersonic_3d_cuda.cu:909-937): max over fluid cells of sum_axes (|u_a| + a) / d_a"""
        self.maxs.put(np.zeros(1, np.float32))
        self.m.launch("k_maxwavespeed_pre", self.grid, self.BLOCK, [_p(x) for x in self.a] + [_p(self.solid), _p(self.maxs)])
        self.m.sync()
        return float(self.maxs.get(np.float32, (1,))[0])

    def schlieren(self):
        """k_schlieren_export (th3cs.cu:641-673; launched :1200-1201) of the current state: |grad rho| per cell, 0 in solids"""
        out = DevBuf(4 * self.N)
        self.m.launch("k_schlieren_export", self.grid, self.BLOCK, [_p(x) for x in self.a] + [_p(self.solid), _p(out)])
        self.m.sync()
        v = out.get(np.float32, self.shape)
        out.free()
        return v

    def run(self, nsteps):
        """th3cs.cu:1160-1196 == tau_hypersonic_3d_cuda.cu:1680-1704, in fp32 with libm's expf as the host code has it"""
        f = np.float32
        m = libm()
        maxs = 0.0
        for _ in range(nsteps):
            self.t = f(self.t * f(m.expf(C.c_float(self.d_tau))))
            dt = f(self.t * self.d_tau)
            ramp = f(self.t / f(0.02))
            gain = f(min(max(ramp, f(0.0)), f(1.0)))
            maxs = self.step(float(dt), float(gain))
            dt_cfl = f(f(self.p.cfl) / f(max(f(maxs), f(1e-9))))
            if dt > f(f(1.10) * dt_cfl):
                self.d_tau = f(self.d_tau * f(0.80))
            elif dt < f(f(0.85) * dt_cfl):
                self.d_tau = f(self.d_tau * f(1.10))
            self.d_tau = f(min(max(self.d_tau, f(1e-7)), f(5e-2)))
        return dict(t=float(self.t), d_tau=float(self.d_tau), maxs=maxs)


# ------------------------------------------------------------------------------------------------------------------
# Gray-Scott: tau_gray_scott.cu step_kernel, launched as :316-326 does
# ------------------------------------------------------------------------------------------------------------------
class RefGrayScott:
    def __init__(self, nx, ny, Du=0.2, Dv=0.1, dt=1.0, dx=1.0, feed=0.03, kill=0.06, ieee=False):
        """defaults: tau_gray_scott.cu:43-61"""
        self.m = RefModule("tau_gray_scott.ieee" if ieee else "tau_gray_scott")
        self.nx, self.ny = nx, ny
        self.k = (Du, Dv, dt, dx, feed, kill)
        n = nx * ny
        self.u0, self.v0, self.u1, self.v1 = (DevBuf(4 * n) for _ in range(4))

    def upload(self, u, v):
        self.u0.put(np.asarray(u, np.float32))
        self.v0.put(np.asarray(v, np.float32))

    def download(self):
        return self.u0.get(np.float32, (self.ny, self.nx)), self.v0.get(np.float32, (self.ny, self.nx))

    def step(self, n=1):
        grid = ((self.nx + 15) // 16, (self.ny + 15) // 16)
        for _ in range(n):
            self.m.launch("step_kernel", grid, (16, 16),
                          [_p(self.u1), _p(self.v1), _p(self.u0), _p(self.v0), C.c_int(self.nx), C.c_int(self.ny)] +
                          [C.c_float(x) for x in self.k])
            self.u0, self.u1 = self.u1, self.u0
            self.v0, self.v1 = self.v1, self.v0
        self.m.sync()

    def close(self):
        for b in (self.u0, self.v0, self.u1, self.v1):
            b.free()


# ------------------------------------------------------------------------------------------------------------------
# SPH: tau_sph.cu — k_clear_heads, k_build_cells, k_density_pressure_cell, k_forces_cell, k_integrate as :676-700 launches them
# ------------------------------------------------------------------------------------------------------------------
class RefSph:
    def __init__(self, N, boxX=1.0, boxY=1.0, rho0=1.0, c0=1.0, gammaEOS=1.0, hMul=2.0, viscAlpha=0.1, gravity=9.81,
                 useVisc=1, useGrav=1, ieee=False, **_):
        """derived quantities as main computes them, in fp32 (tau_sph.cu:573-576, 512-521)"""
        f = np.float32
        self.m = RefModule("tau_sph.ieee" if ieee else "tau_sph")
        self.N = N
        self.box = (f(boxX), f(boxY))
        area = f(f(boxX) * f(boxY))
        self.mass = f(f(f(rho0) * area) / f(N))
        spacing = f(libm().sqrtf(C.c_float(f(area / f(N)))))
        self.h = f(f(hMul) * spacing)
        self.cell = f(f(2.0) * self.h)
        self.Gx = max(1, int(np.ceil(f(f(boxX) / self.cell))))
        self.Gy = max(1, int(np.ceil(f(f(boxY) / self.cell))))
        self.par = dict(rho0=rho0, c0=c0, gammaEOS=gammaEOS, viscAlpha=viscAlpha, gravity=gravity, useVisc=useVisc, useGrav=useGrav)
        self.pos, self.vel, self.acc = (DevBuf(8 * N) for _ in range(3))
        self.s, self.press, self.next = (DevBuf(4 * N) for _ in range(3))
        self.head = DevBuf(4 * self.Gx * self.Gy)

    def upload(self, pos, vel):
        self.pos.put(np.asarray(pos, np.float32))
        self.vel.put(np.asarray(vel, np.float32))

    def substep(self, dt):
        N, BS = self.N, 256
        GS = (N + BS - 1) // BS
        M = self.Gx * self.Gy
        GSm = (M + BS - 1) // BS
        q = self.par
        i32, f32 = C.c_int, C.c_float
        self.m.launch("k_clear_heads", (GSm,), (BS,), [_p(self.head), i32(M)])
        self.m.launch("k_build_cells", (GS,), (BS,), [_p(self.pos), i32(N), _p(self.head), _p(self.next), i32(self.Gx), i32(self.Gy), f32(self.cell)])
        self.m.sync()
        lists = (self.head.get(np.int32, (M,)), self.next.get(np.int32, (N,)))
        self.m.launch("k_density_pressure_cell", (GS,), (BS,),
                      [_p(self.pos), _p(self.s), _p(self.press), _p(self.head), _p(self.next), i32(N), f32(self.mass), f32(self.h),
                       f32(q["rho0"]), f32(q["c0"]), f32(q["gammaEOS"]), i32(self.Gx), i32(self.Gy), f32(self.cell)])
        self.m.launch("k_forces_cell", (GS,), (BS,),
                      [_p(self.pos), _p(self.vel), _p(self.s), _p(self.press), _p(self.acc), _p(self.head), _p(self.next), i32(N),
                       f32(self.mass), f32(self.h), f32(q["viscAlpha"]), f32(q["c0"]), f32(0.0),
                       f32(-(q["gravity"] if q["useGrav"] else 0.0)), C.c_bool(bool(q["useVisc"])), C.c_bool(bool(q["useGrav"])),
                       i32(self.Gx), i32(self.Gy), f32(self.cell)])
        self.m.launch("k_integrate", (GS,), (BS,), [_p(self.pos), _p(self.vel), _p(self.acc), i32(N), f32(dt), f32(self.box[0]), f32(self.box[1])])
        self.m.sync()
        return lists

    def xsph(self, eps):
        """k_xsph_cell + k_apply_xsph as tau_sph.cu:698-704 launches them after k_integrate: the lists of the last build, the
        positions / velocities from after the integrate; acc is the scratch array for the velocity increments"""
        N, BS = self.N, 256
        GS = (N + BS - 1) // BS
        i32, f32 = C.c_int, C.c_float
        self.m.launch("k_xsph_cell", (GS,), (BS,), [_p(self.pos), _p(self.vel), _p(self.s), _p(self.acc), _p(self.head), _p(self.next), i32(N),
                                                    f32(self.mass), f32(self.h), f32(eps), i32(self.Gx), i32(self.Gy), f32(self.cell)])
        self.m.launch("k_apply_xsph", (GS,), (BS,), [_p(self.vel), _p(self.acc), i32(N)])
        self.m.sync()

    def rasterize(self, W, H):
        """k_clear_grid + k_rasterize, tau_sph.cu:747-753: particle counts on a (2H, W) raster, y flipped"""
        size = 2 * H * W
        g = DevBuf(4 * size)
        self.m.launch("k_clear_grid", ((size + 255) // 256,), (256,), [_p(g), C.c_int(size)])
        self.m.launch("k_rasterize", ((self.N + 255) // 256,), (256,), [_p(self.pos), C.c_int(self.N), _p(g), C.c_int(W), C.c_int(H),
                                                                        C.c_float(self.box[0]), C.c_float(self.box[1])])
        self.m.sync()
        out = g.get(np.int32, (2 * H, W))
        g.free()
        return out

    @staticmethod
    def cells_from_lists(head, nxt):
        """particle -> cell from the linked lists k_build_cells wrote (the integer result of the cell build)"""
        cell = np.full(nxt.shape[0], -1, np.int32)
        cur = head.copy()
        cid = np.arange(head.shape[0], dtype=np.int32)
        while True:
            live = cur >= 0
            if not live.any():
                break
            cell[cur[live]] = cid[live]
            cur[live] = nxt[cur[live]]
        return cell

    def state(self):
        N = self.N
        return dict(pos=self.pos.get(np.float32, (N, 2)), vel=self.vel.get(np.float32, (N, 2)), acc=self.acc.get(np.float32, (N, 2)),
                    s=self.s.get(np.float32, (N,)), press=self.press.get(np.float32, (N,)))

    def close(self):
        for b in (self.pos, self.vel, self.acc, self.s, self.press, self.next, self.head):
            b.free()


# ------------------------------------------------------------------------------------------------------------------
# 2D Euler: tau_hypersonic_cuda.cu through the reference's own test seam (W = 8192, H = 1024 are compile-time, fp64)
# ------------------------------------------------------------------------------------------------------------------
class SimConfig(C.Structure):
    """tau_hypersonic_cuda.cu:37-50"""
    _fields_ = [(n, C.c_double) for n in "gamma cfl visc_nu visc_rho visc_e inflow_mach geom_x0 geom_cy geom_Rb geom_Rn geom_theta".split()] + \
               [("steps_per_frame", C.c_int)]


class Soa4(C.Structure):
    """Usoa / Csoa (tau_hypersonic_cuda.cu:109-115): four device pointers, passed by value"""
    _fields_ = [(n, C.c_void_p) for n in ("rho", "mx", "my", "E")]


class RefH2:
    W, H = 8192, 1024

    def __init__(self):
        self.m = RefModule("tau_hypersonic_cuda_tests")
        W, H = self.W, self.H
        c = SimConfig(1.1, 0.25, 5e-2, 5e-2, 2e-2, 25.0, 125.0, H / 2.0, H / 12.0, H / 24.0, np.pi / 4.0, 2)   # default_config :1394-1409
        self.cfg = c
        self.m.set_global("d_cfg", c)
        N = W * H
        self.N = N
        self.threads = 256
        self.blocksN = (N + 255) // 256

        def soa(n):
            bufs = [DevBuf(8 * n) for _ in range(4)]
            return bufs, Soa4(*[b.ptr.value for b in bufs])
        self.Ub, self.U = soa(N)
        self.Tb, self.T = soa(N)
        self.faces = [soa(N) for _ in range(4)]                       # XStateL, XStateR, YStateL, YStateR (tests:454-457)
        self.xflux = soa((W + 1) * H)
        self.yflux = soa(W * (H + 1))
        self.mask = DevBuf(N)
        self.maxspeed = DevBuf(8)
        self.blockmax = DevBuf(8 * self.blocksN)

    def init(self):
        self.m.launch("k_init", (self.blocksN,), (self.threads,), [self.U, _p(self.mask)])
        self.m.sync()

    def upload(self, fields):
        for b, f in zip(self.Ub, fields):
            b.put(np.asarray(f, np.float64))

    def download(self):
        return [b.get(np.float64, (self.H, self.W)) for b in self.Ub]

    def mask_host(self):
        return self.mask.get(np.uint8, (self.H, self.W))

    def step(self, n=1, dt=None):
        """run_hypersonic_steps, tau_hypersonic_cuda_tests.cu:178-243; returns the last dt"""
        W, H, thr = self.W, self.H, self.threads
        tb = (32, 4)
        tiled = ((W + 31) // 32, (H + 3) // 4)
        shm_p = 4 * 34 * 6 * 8 + 34 * 6
        shm_s = 4 * 36 * 8 * 8 + 36 * 8
        c = self.cfg
        for _ in range(n):
            self.m.launch("k_apply_inflow_left", ((H + thr - 1) // thr,), (thr,), [self.U, _p(self.mask)])
            # the two reductions keep their partial maxima in `extern __shared__ double smax[]`: the program launches them with
            # reduceSharedBytes = threads * 8 (tau_hypersonic_cuda.cu:1839-1842).  The reference's TEST harness omits the third
            # launch argument (tau_hypersonic_cuda_tests.cu:207-209): with no LDS behind smax the maximum reads back 0 on this
            # GPU, dt falls through to the diffusion limit 5.0 and the reference's own regression run ends in inf / nan
            # (profiles/r04/reference_programs_mi355x.txt).  The program's launch is the one restated here.
            self.m.launch("k_max_wavespeed_blocks", (self.blocksN,), (thr,), [self.U, _p(self.mask), _p(self.blockmax)], shmem=thr * 8)
            self.m.launch("k_reduce_block_max", (1,), (thr,), [_p(self.blockmax), C.c_int(self.blocksN), _p(self.maxspeed)], shmem=thr * 8)
            self.m.sync()
            maxs = float(self.maxspeed.get(np.float64, (1,))[0])
            if not np.isfinite(maxs) or maxs < 1e-12:
                maxs = 1e-12
            dt_conv = c.cfl / maxs
            nu_max = max(c.visc_nu, c.visc_rho, c.visc_e)
            dt_diff = 0.25 / nu_max if (np.isfinite(nu_max) and nu_max > 1e-12) else dt_conv
            step_dt = min(dt_conv, dt_diff) if dt is None else dt
            half = 0.5 * step_dt
            (_, XL), (_, XR), (_, YL), (_, YR) = self.faces
            self.m.launch("k_predict_face_states", tiled, tb, [self.U, _p(self.mask), XL, XR, YL, YR, C.c_double(half), C.c_double(half)], shmem=shm_p)
            self.m.launch("k_compute_xface_flux", (((W + 1) * H + thr - 1) // thr,), (thr,), [self.U, _p(self.mask), XL, XR, self.xflux[1]])
            self.m.launch("k_compute_yface_flux", ((W * (H + 1) + thr - 1) // thr,), (thr,), [self.U, _p(self.mask), YL, YR, self.yflux[1]])
            self.m.launch("k_step", tiled, tb, [self.U, self.T, _p(self.mask), self.xflux[1], self.yflux[1],
                                                C.c_double(step_dt), C.c_double(step_dt), C.c_double(step_dt)], shmem=shm_s)
            self.Ub, self.Tb = self.Tb, self.Ub
            self.U, self.T = self.T, self.U
        self.m.sync()
        return step_dt, maxs

    def render(self, view_mode):
        """the frame tail of tau_hypersonic_cuda.cu:1892-1921: k_render_vals (+ per-block min / max) -> k_reduce_minmax rounds ->
        k_compute_inv_range -> k_render_pixels.  Returns (pixels (H, W, 4) u8, values (H, W) f64, min, max)."""
        N, thr = self.N, self.threads
        shm = thr * 2 * 8
        tmp, px = DevBuf(8 * N), DevBuf(4 * N)
        bmin, bmax, rmin, rmax = (DevBuf(8 * self.blocksN) for _ in range(4))
        inv = DevBuf(8)
        self.m.launch("k_render_vals", (self.blocksN,), (thr,), [self.U, _p(self.mask), C.c_int(view_mode), _p(tmp), _p(bmin), _p(bmax)], shmem=shm)
        cur_min, cur_max, out_min, out_max = bmin, bmax, rmin, rmax
        cur_n = self.blocksN
        while cur_n > 1:
            out_n = (cur_n + 2 * thr - 1) // (2 * thr)
            self.m.launch("k_reduce_minmax", (out_n,), (thr,), [_p(cur_min), _p(cur_max), _p(out_min), _p(out_max), C.c_int(cur_n)], shmem=shm)
            cur_n = out_n
            nxt_min, nxt_max = out_min, out_max
            out_min = rmin if nxt_min is bmin else bmin
            out_max = rmax if nxt_max is bmax else bmax
            cur_min, cur_max = nxt_min, nxt_max
        self.m.launch("k_compute_inv_range", (1,), (1,), [_p(cur_min), _p(cur_max), _p(inv)])
        self.m.launch("k_render_pixels", (self.blocksN,), (thr,), [_p(self.mask), _p(tmp), _p(cur_min), _p(inv), _p(px)])
        self.m.sync()
        out = (px.get(np.uint8, (self.H, self.W, 4)), tmp.get(np.float64, (self.H, self.W)),
               float(cur_min.get(np.float64, (1,))[0]), float(cur_max.get(np.float64, (1,))[0]))
        for b in (tmp, px, bmin, bmax, rmin, rmax, inv):
            b.free()
        return out

    def close(self):
        for bufs in [self.Ub, self.Tb, self.xflux[0], self.yflux[0]] + [f[0] for f in self.faces]:
            for b in bufs:
                b.free()
        for b in (self.mask, self.maxspeed, self.blockmax):
            b.free()


# ------------------------------------------------------------------------------------------------------------------
# D2Q9 LBM: tau_lbm.cu init_kernel / collide_stream_kernel, launched as :246-263 does
# ------------------------------------------------------------------------------------------------------------------
class LbmParamsRef(C.Structure):
    """tau_lbm.cu:43-55, passed to the kernels by value"""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("steps", C.c_int), ("stride", C.c_int), ("fps_limit", C.c_int),
                ("headless", C.c_bool), ("obstacle", C.c_bool),
                ("tau", C.c_float), ("drive", C.c_float), ("rho0", C.c_float), ("obstacle_radius", C.c_float)]


class RefLbm:
    def __init__(self, nx=512, ny=256, obstacle=True, tau=0.56, drive=1.0e-6, rho0=1.0, obstacle_radius=32.0, ieee=False):
        self.m = RefModule("tau_lbm.ieee" if ieee else "tau_lbm")
        self.P = LbmParamsRef(nx, ny, 0, 4, 0, True, bool(obstacle), tau, drive, rho0, obstacle_radius)
        self.nx, self.ny = nx, ny
        n = nx * ny
        self.f0, self.f1 = DevBuf(36 * n), DevBuf(36 * n)
        self.solid = DevBuf(n)
        self.grid = ((nx + 15) // 16, (ny + 15) // 16)

    def init(self):
        self.m.launch("init_kernel", self.grid, (16, 16), [_p(self.f0), _p(self.solid), self.P])
        self.m.sync()
        self.f1.copy_from_device(self.f0.ptr.value)

    def upload(self, f, solid=None):
        self.f0.put(np.asarray(f, np.float32))
        if solid is not None:
            self.solid.put(np.asarray(solid, np.uint8))

    def download(self):
        return self.f0.get(np.float32, (9, self.ny, self.nx)), self.solid.get(np.uint8, (self.ny, self.nx))

    def step(self, n=1):
        for _ in range(n):
            self.m.launch("collide_stream_kernel", self.grid, (16, 16), [_p(self.f0), _p(self.f1), _p(self.solid), self.P])
            self.f0, self.f1 = self.f1, self.f0
        self.m.sync()

    def close(self):
        for b in (self.f0, self.f1, self.solid):
            b.free()


# ------------------------------------------------------------------------------------------------------------------
# Burgers / shallow water: the convective part of tau_burgers.cu:677-708 and tau_shallow_water.cu:671-700 (the in-place viscosity
# kernels that follow race — SURVEY §2.1 — and are left out: compare with nu = 0)
# ------------------------------------------------------------------------------------------------------------------
class RefFlow:
    BS = (16, 16)

    def __init__(self, kind, nx, ny, dx, dy, u0=1.0, g=9.81, CFL=0.5, muscl=0, oneD=0, fast=False):
        self.kind = kind
        self.m = RefModule({"burgers": "tau_burgers", "sw": "tau_shallow_water"}[kind] + ("" if fast else ".ieee"))
        self.nx, self.ny, self.dx, self.dy = nx, ny, np.float32(dx), np.float32(dy)
        self.u0, self.g, self.CFL, self.muscl, self.oneD = np.float32(u0), np.float32(g), np.float32(CFL), int(muscl), int(oneD)
        n = nx * ny
        self.nf = 2 if kind == "burgers" else 3
        self.f = [DevBuf(4 * n) for _ in range(self.nf)]
        self.flux = [DevBuf(4 * n) for _ in range(2 * self.nf)]
        self.gs = ((nx + 15) // 16, (ny + 15) // 16)
        self.blk = DevBuf(4 * self.gs[0] * self.gs[1])

    def upload(self, fields):
        for b, a in zip(self.f, fields):
            b.put(np.asarray(a, np.float32))

    def download(self):
        return [b.get(np.float32, (self.ny, self.nx)) for b in self.f]

    def dt_eff(self, t, dtau):
        """the CFL part of do_step: wavespeed_block_max + the host max (tau_burgers.cu:678-692 / tau_shallow_water.cu:672-688)"""
        f = np.float32
        i32, f32 = C.c_int, C.c_float
        shm = 16 * 16 * 4
        if self.kind == "burgers":
            self.m.launch("wavespeed_block_max", self.gs, self.BS,
                          [_p(self.f[0]), _p(self.f[1]), f32(self.u0), i32(self.nx), i32(self.ny), f32(f(1.0) / self.dx),
                           f32((f(1.0) / self.dy) if self.ny > 1 else 0.0), _p(self.blk)], shmem=shm)
        else:
            self.m.launch("wavespeed_block_max", self.gs, self.BS,
                          [_p(self.f[0]), _p(self.f[1]), _p(self.f[2]), f32(self.g), i32(self.nx), i32(self.ny), _p(self.blk)], shmem=shm)
        self.m.sync()
        blk = self.blk.get(np.float32, (self.gs[0] * self.gs[1],))
        if self.kind == "burgers":
            smax = max(f(1e-12), blk.max())
            dt_cfl = f(self.CFL / smax)
        else:
            cmax = max(blk.max(), f(1e-12))
            dt_cfl = f(f(self.CFL * min(self.dx, self.dy)) / cmax)
        return float(min(f(f(t) * f(dtau)), dt_cfl))

    def convect(self, dt):
        i32, f32 = C.c_int, C.c_float
        nx, ny = self.nx, self.ny
        if self.kind == "burgers":
            pu, pv = self.f
            Fu, Fv, Gu, Gv = self.flux
            self.m.launch("flux_x_kernel", self.gs, self.BS, [_p(pu), _p(pv), _p(Fu), _p(Fv), i32(nx), i32(ny), f32(self.u0), i32(self.muscl)])
            if not self.oneD:
                self.m.launch("flux_y_kernel", self.gs, self.BS, [_p(pu), _p(pv), _p(Gu), _p(Gv), i32(nx), i32(ny), f32(self.u0), i32(self.muscl)])
            self.m.launch("update_convective", self.gs, self.BS, [_p(pu), _p(pv), _p(Fu), _p(Fv), _p(Gu), _p(Gv), i32(nx), i32(ny), f32(self.dx),
                                                                  f32(self.dy), f32(dt), f32(self.u0), i32(self.oneD)])
        else:
            sg, u, v = self.f
            Fh, Fmx, Fmy, Gh, Gmx, Gmy = self.flux
            self.m.launch("flux_x_kernel", self.gs, self.BS, [_p(sg), _p(u), _p(v), _p(Fh), _p(Fmx), _p(Fmy), i32(nx), i32(ny), f32(self.g)])
            self.m.launch("flux_y_kernel", self.gs, self.BS, [_p(sg), _p(u), _p(v), _p(Gh), _p(Gmx), _p(Gmy), i32(nx), i32(ny), f32(self.g)])
            self.m.launch("update_kernel", self.gs, self.BS, [_p(sg), _p(u), _p(v), _p(Fh), _p(Fmx), _p(Fmy), _p(Gh), _p(Gmx), _p(Gmy), i32(nx), i32(ny),
                                                              f32(self.dx), f32(self.dy), f32(dt), f32(self.g)])
        self.m.sync()

    def viscosity_single_wave(self, nu, dt, block):
        """The reference's in-place viscosity kernel (tau_burgers.cu:490-525 viscosity_step, tau_shallow_water.cu:516-547
        viscosity_uv) launched as ONE workgroup of ONE wave (block = the whole grid, nx * ny = 64 cells).  As the programs launch it
        (16 x 16 blocks over a large grid) its result depends on the order in which threads overwrite the cells their neighbours
        still have to read — no defined answer, SURVEY §2.1.  Inside a single wave64 it has one: every load of the kernel precedes
        its stores in the instruction stream, the stored values depend on the loaded ones (so the wave waits for ALL its lanes'
        loads before the store instruction issues), and nobody else touches the arrays — the in-place update is the Jacobi step,
        computed by the reference's own arithmetic.  That is the regime in which the kernel can referee."""
        i32, f32 = C.c_int, C.c_float
        nx, ny = self.nx, self.ny
        assert nx * ny == 64 and block[0] * block[1] == 64 and block[0] == nx and block[1] == ny, "one wave covering the whole grid"
        if self.kind == "burgers":
            self.m.launch("viscosity_step", (1, 1), block, [_p(self.f[0]), _p(self.f[1]), i32(nx), i32(ny), f32(self.dx), f32(self.dy), f32(nu),
                                                           f32(dt), f32(self.u0), i32(self.oneD)])
        else:
            self.m.launch("viscosity_uv", (1, 1), block, [_p(self.f[1]), _p(self.f[2]), i32(nx), i32(ny), f32(self.dx), f32(self.dy), f32(nu), f32(dt)])
        self.m.sync()

    def close(self):
        for b in self.f + self.flux + [self.blk]:
            b.free()
