"""ctypes binding of libtaueng.so — the host-side mirror of include/taueng.h.

This is plumbing only: every class is a thin handle wrapper whose methods are the C-ABI entry
points one-to-one.  There is NO CPU fallback here: if the HIP library is missing or no gfx950
device is visible, construction raises TauError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class TauError(RuntimeError):
    pass


def lib_path():
    # TAUENG_LIB lets a tuning experiment load a variant build of the same library; default: the in-tree one
    return os.environ.get("TAUENG_LIB") or os.path.join(_HERE, "lib", "libtaueng.so")


_lib = None


class Tau3DParams(C.Structure):
    """tau3d_params (include/tau_params.h) == reference Params, tau_hypersonic_3d_cuda.cu:21-42"""
    _fields_ = ([("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32)] +
                [(n, C.c_float) for n in
                 "dx dy dz cfl u_ref R gamma_floor Twall tau_vib theta_v sdf_cx sdf_cy sdf_cz sdf_r "
                 "inflow_r inflow_p inflow_u inflow_v inflow_w".split()] +
                [("sponge_n", C.c_int32), ("sponge_strength", C.c_float),
                 ("sponge_out_n", C.c_int32), ("sponge_out_strength", C.c_float)])


class Tau3DClock(C.Structure):
    _fields_ = [(n, C.c_float) for n in "t d_tau dt gain maxs".split()] + [("step", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class GSParams(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32)] + [(n, C.c_float) for n in "dx dt Du Dv feed kill".split()]


class H2Params(C.Structure):
    """tauh2_params == reference SimConfig (tau_hypersonic_cuda.cu:37-50) + the W, H it fixes at compile time"""
    _fields_ = [("W", C.c_int32), ("H", C.c_int32)] + [(n, C.c_double) for n in
               "gamma cfl visc_nu visc_rho visc_e mach geom_x0 geom_cy geom_rb geom_rn geom_theta".split()]


class SphParams(C.Structure):
    """tausph_params == reference Params (tau_sph.cu:49-85)"""
    _fields_ = ([("N", C.c_int32)] + [(n, C.c_float) for n in
                "boxX boxY dTau t0 CFL rho0 c0 gammaEOS hMul viscAlpha gravity".split()] +
                [(n, C.c_int32) for n in "useVisc useGrav viscSub seed useXSPH".split()] +
                [("xsphEps", C.c_float), ("rain", C.c_int32)])


class LbmParams(C.Structure):
    """taulbm_params == reference Params (tau_lbm.cu:43-55), physics fields only"""
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("obstacle", C.c_int32), ("tau", C.c_float), ("drive", C.c_float),
                ("rho0", C.c_float), ("obstacle_radius", C.c_float)]


class FlowParams(C.Structure):
    """tauflow_params: Params of tau_burgers.cu:56-91 / tau_shallow_water.cu:54-88 in one block"""
    _fields_ = ([("nx", C.c_int32), ("ny", C.c_int32)] +
                [(n, C.c_float) for n in "dx dy nu u0 g H0 CFL tau0 t0 dtau".split()] +
                [(n, C.c_int32) for n in "muscl visc_substeps oneD".split()] +
                [(n, C.c_float) for n in "amp bsig swirl rc offx offy asym".split()] +
                [("ck", C.c_int32), ("ca", C.c_float)])


class LapParams(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32)] + [(n, C.c_float) for n in "dx dy nu dt u0".split()]


def _load_hip_runtime():
    """libtaueng.so has no DT_NEEDED on the HIP runtime: bind it to the runtime PyTorch ships when
    torch is installed (one runtime for torch streams, RCCL and the engine), else to /opt/rocm's."""
    cands = []
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    except Exception:
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    err = None
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover
            err = e
    raise TauError(f"no HIP runtime (libamdhip64) could be loaded: {err}")


def load():
    """Load libtaueng.so (raises TauError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise TauError(f"{p} not found — run `make -C fluid-sims_amd` (or __graft_entry__.build())")
    _load_hip_runtime()
    L = C.CDLL(p)
    L.tau_last_error.restype = C.c_char_p
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    sig = {
        "tau_device_available": ([], i32), "tau_version": ([], i32),
        "tau3d_params_default": ([C.POINTER(Tau3DParams), i32, i32, i32], None),
        "tau3d_create": ([C.POINTER(vp), C.POINTER(Tau3DParams), i32, i32, i32, vp], i32),
        "tau3d_destroy": ([vp], None),
        "tau3d_init": ([vp, i32], i32),
        "tau3d_upload_state": ([vp, C.POINTER(vp)], i32),
        "tau3d_download_state": ([vp, C.POINTER(vp)], i32),
        "tau3d_download_solid": ([vp, vp], i32),
        "tau3d_upload_planes": ([vp, i32, i32, C.POINTER(vp)], i32),
        "tau3d_download_planes": ([vp, i32, i32, C.POINTER(vp)], i32),
        "tau3d_state_ptrs": ([vp, C.POINTER(vp), C.POINTER(vp)], i32),
        "tau3d_get_clock": ([vp, C.POINTER(Tau3DClock)], i32),
        "tau3d_set_clock": ([vp, C.POINTER(Tau3DClock)], i32),
        "tau3d_step": ([vp, i32, C.POINTER(Tau3DClock)], i32),
        "tau3d_step_async": ([vp, i32], i32),
        "tau3d_step_explicit": ([vp, f32, f32, C.POINTER(f32)], i32),
        "tau3d_clock_begin_async": ([vp], i32),
        "tau3d_step_range_async": ([vp, i32, i32, vp], i32),
        "tau3d_step_edges_async": ([vp, i32, vp], i32),
        "tau3d_clock_end_async": ([vp], i32),
        "tau3d_slab_begin_async": ([vp], i32), "tau3d_slab_edges_async": ([vp, i32], i32),
        "tau3d_slab_xy_async": ([vp], i32), "tau3d_slab_z_async": ([vp], i32), "tau3d_slab_clock_async": ([vp], i32),
        "tau3d_slab_xy_fix_async": ([vp], i32), "tau3d_debug_set_fmax_in": ([vp, C.c_float], i32),
        "tau3d_slab_interior_async": ([vp, i32], i32), "tau3d_slab_end_async": ([vp], i32),
        "tau3d_fill_halo_periodic_async": ([vp], i32),
        "tau3d_halo_send_ptr": ([vp, i32, i32, i32, C.POINTER(vp)], i32),
        "tau3d_halo_recv_ptr": ([vp, i32, i32, i32, C.POINTER(vp)], i32),
        "tau3d_pack_halos_async": ([vp, i32], i32),
        "tau3d_unpack_halos_async": ([vp, i32], i32),
        "tau3d_halo_buf_ptr": ([vp, i32, i32, C.POINTER(vp), C.POINTER(C.c_size_t)], i32),
        "tau3d_max_ptr": ([vp, C.POINTER(vp)], i32),
        "tau3d_state_written": ([vp], i32),
        "tau3d_palette_indices": ([vp, f32, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)], i32),
        "tau3d_field_range": ([vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32)], i32),
        "tau3d_uniform_tiles": ([vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(i32)], i32),
        "tau3d_tile_list_stats": ([vp, C.POINTER(i32), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)], i32),
        "tau3d_sync": ([vp], i32),
        "tau_device_count": ([C.POINTER(i32)], i32),
        "tau_guided_chunks": ([i32, i32, i32, i32, i32, C.POINTER(i32), i32, C.POINTER(i32)], i32),
        "tau3d_slab_info": ([vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)], i32),
        "tau3d_is_split": ([vp], i32), "tau3d_set_split": ([vp, i32], i32),
        "tau3d_slab_bounds": ([i32, i32, i32, C.POINTER(i32), C.POINTER(i32)], i32),
        "tau3d_ring_create": ([C.POINTER(vp), vp, i32, i32, i32, C.c_char_p, C.c_uint64], i32),
        "tau3d_set_halo_direct": ([vp, i32], i32),
        "tau3d_state_group": ([vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(i32)], i32),
        "tau3d_ring_destroy": ([vp], None),
        "tau3d_ring_prime": ([vp], i32), "tau3d_ring_invalidate": ([vp], i32),
        "tau3d_ring_step_async": ([vp, i32], i32), "tau3d_ring_finish": ([vp], i32),
        "tau3d_ring_get_clock": ([vp, C.POINTER(Tau3DClock)], i32), "tau3d_ring_barrier": ([vp], i32),
        "tau3d_ring_timing_enable": ([vp, i32], i32),
        "tau3d_ring_timing_read": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)], i32),
        "tau3d_ring_info": ([vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t], i32),
        "tau3d_vis": ([vp, i32, vp], i32),
        "tau3d_vis_async": ([vp, i32, vp], i32),
        "tau3d_slice_rgba": ([vp, i32, i32, f32, vp, C.POINTER(f32), C.POINTER(f32)], i32),
        "tau3d_outflow_reflection": ([vp, i32, C.POINTER(f32)], i32),
        "tau3d_timing_enable": ([vp, i32], i32),
        "tau3d_timing_read": ([vp, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(C.c_double)], i32),
        "tau3d_timing_read_split": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)], i32),
        "tau3d_timing_span": ([vp, C.POINTER(C.c_double)], i32),
        "tauh2_params_default": ([C.POINTER(H2Params), i32, i32], None),
        "tauh2_create": ([C.POINTER(vp), C.POINTER(H2Params), i32, vp], i32),
        "tauh2_destroy": ([vp], None),
        "tauh2_render": ([vp, i32, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)], i32),
        "tauh2_init": ([vp], i32),
        "tauh2_upload": ([vp, C.POINTER(vp), vp], i32),
        "tauh2_download": ([vp, C.POINTER(vp), vp], i32),
        "tauh2_state_ptrs": ([vp, C.POINTER(vp), C.POINTER(vp)], i32),
        "tauh2_step": ([vp, i32, C.POINTER(C.c_double)], i32),
        "tauh2_step_async": ([vp, i32], i32),
        "tauh2_step_explicit": ([vp, C.c_double], i32),
        "tauh2_get_time": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)], i32),
        "tauh2_sync": ([vp], i32),
        "tauh2_unit_eval": ([vp, C.POINTER(f32)], i32),
        "tauh2_unit_neighbors": ([vp, i32, i32, C.POINTER(f32)], i32),
        "tauh2_body_sdf": ([C.c_double] * 5, C.c_double),
        "tausph_params_default": ([C.POINTER(SphParams), i32], None),
        "tausph_create": ([C.POINTER(vp), C.POINTER(SphParams), i32, vp], i32),
        "tausph_destroy": ([vp], None),
        "tausph_reset_particles": ([vp], i32),
        "tausph_upload": ([vp, vp, vp], i32),
        "tausph_download": ([vp, vp, vp, vp, vp, vp, vp], i32),
        "tausph_state_ptrs": ([vp] + [C.POINTER(vp)] * 5, i32),
        "tausph_grid": ([vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)], i32),
        "tausph_dt": ([vp], f32),
        "tausph_substep_async": ([vp, f32], i32),
        "tausph_step": ([vp, i32], i32),
        "tausph_step_async": ([vp, i32], i32),
        "tausph_get_clock": ([vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(C.c_int64)], i32),
        "tausph_sync": ([vp], i32),
        "taulbm_params_default": ([C.POINTER(LbmParams)], None),
        "taulbm_create": ([C.POINTER(vp), C.POINTER(LbmParams), i32, vp], i32),
        "taulbm_destroy": ([vp], None),
        "taulbm_init": ([vp], i32),
        "taulbm_upload": ([vp, vp, vp], i32),
        "taulbm_download": ([vp, vp, vp], i32),
        "taulbm_state_ptrs": ([vp, C.POINTER(vp), C.POINTER(vp)], i32),
        "taulbm_set_drive": ([vp, f32], i32),
        "taulbm_step": ([vp, i32], i32),
        "taulbm_step_async": ([vp, i32], i32),
        "taulbm_speed": ([vp, vp], i32),
        "taulbm_steps_done": ([vp], C.c_int64),
        "taulbm_sync": ([vp], i32),
        "tausph_rasterize": ([vp, i32, i32, vp], i32),
        "tausph_rain_spawned": ([vp], C.c_int64),
        "tausph_count_pairs": ([vp, C.POINTER(C.c_int64)], i32),
        "tausph_state_written": ([vp], i32),
        "tauflow_params_default": ([C.POINTER(FlowParams), i32, i32, i32], None),
        "tauflow_create": ([C.POINTER(vp), C.POINTER(FlowParams), i32, i32, vp], i32),
        "tauflow_destroy": ([vp], None),
        "tauflow_init": ([vp], i32),
        "tauflow_upload": ([vp, C.POINTER(vp)], i32),
        "tauflow_download": ([vp, C.POINTER(vp)], i32),
        "tauflow_state_ptrs": ([vp, C.POINTER(vp)], i32),
        "tauflow_step": ([vp, i32], i32),
        "tauflow_step_async": ([vp, i32], i32),
        "tauflow_step_explicit": ([vp, f32], i32),
        "tauflow_get_clock": ([vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(C.c_int64)], i32),
        "tauflow_colehopf_relL2": ([vp, f32, C.POINTER(C.c_double)], i32),
        "tauflow_sync": ([vp], i32),
        "tauflow_timer_start": ([vp], i32), "tauflow_timer_stop": ([vp, C.POINTER(C.c_double)], i32),
        "taugs_params_default": ([C.POINTER(GSParams), i32, i32], None),
        "taugs_create": ([C.POINTER(vp), C.POINTER(GSParams), i32, vp], i32),
        "taugs_destroy": ([vp], None),
        "taugs_init_pattern": ([vp, C.c_uint32], i32),
        "taugs_upload": ([vp, vp, vp], i32),
        "taugs_download": ([vp, vp, vp], i32),
        "taugs_state_ptrs": ([vp, C.POINTER(vp), C.POINTER(vp)], i32),
        "taugs_step": ([vp, i32], i32),
        "taugs_step_async": ([vp, i32], i32),
        "taugs_sync": ([vp], i32),
        "taugs_info": ([vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)], i32),
        "taulap_info": ([vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)], i32),
        "taugs_pattern_host": ([i32, i32, C.c_uint32, vp, vp], i32),
        "taurow_bounds": ([i32, i32, i32, C.POINTER(i32), C.POINTER(i32)], i32),
        "taugs_ring_create": ([C.POINTER(vp), vp, i32, i32, i32, i32, C.c_char_p, C.c_uint64], i32),
        "taulap_ring_create": ([C.POINTER(vp), vp, i32, i32, i32, i32, C.c_char_p, C.c_uint64], i32),
        "taurow_ring_step_async": ([vp, i32], i32), "taurow_ring_exchange_async": ([vp], i32), "taurow_ring_finish": ([vp], i32),
        "taurow_ring_barrier": ([vp], i32), "taurow_ring_destroy": ([vp], None),
        "taurow_ring_info": ([vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_long), C.POINTER(i32), C.POINTER(i32)], i32),
        "taugs_ring_step_async": ([vp, i32], i32), "taugs_ring_exchange_async": ([vp], i32), "taugs_ring_finish": ([vp], i32),
        "taugs_ring_barrier": ([vp], i32), "taugs_ring_destroy": ([vp], None),
        "taulap_ring_step_async": ([vp, i32], i32), "taulap_ring_exchange_async": ([vp], i32), "taulap_ring_finish": ([vp], i32),
        "taulap_ring_barrier": ([vp], i32), "taulap_ring_destroy": ([vp], None),
        "taugs_set_levels": ([vp, i32], i32),
        "taulap_create": ([C.POINTER(vp), C.POINTER(LapParams), i32, i32, i32, vp], i32),
        "taulap_destroy": ([vp], None),
        "taulap_upload": ([vp, vp, vp], i32),
        "taulap_download": ([vp, vp, vp], i32),
        "taulap_state_ptrs": ([vp, C.POINTER(vp), C.POINTER(vp)], i32),
        "taulap_set_dt": ([vp, f32], i32),
        "taulap_step": ([vp, i32], i32),
        "taulap_step_async": ([vp, i32], i32),
        "taulap_sync": ([vp], i32),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name, None)
        if fn is None and os.environ.get("TAUENG_ALLOW_MISSING"):   # A/B runs against an older build of the library
            continue
        if fn is None:
            raise TauError(f"{p} does not export {name}: rebuild the library (make -C fluid-sims_amd)")
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


EXPORTS_3D_2D = None  # filled lazily by tests from include/taueng.h


def _ck(rc):
    if rc != 0:
        raise TauError(load().tau_last_error().decode())


def _require_device():
    L = load()
    if not L.tau_device_available():
        raise TauError("no gfx950 (MI355X) device visible to the HIP runtime — libtaueng has no CPU path")
    return L


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


RING_RCCL, RING_HOST, RING_LOCAL, RING_IPC, RING_IPC_HOSTMAX = 0, 1, 2, 3, 4


def slab_bounds(nz, world, rank):
    """tau3d_slab_bounds: (z0, nzl) of rank's contiguous Z-slab"""
    z0, nzl = C.c_int(), C.c_int()
    _ck(load().tau3d_slab_bounds(nz, world, rank, C.byref(z0), C.byref(nzl)))
    return z0.value, nzl.value


def guided_chunks(H, nstrips, slots_per_xcd, lmin, lmax):
    """tau_guided_chunks: the row starts of the marches' guided chunk schedule (host only; len = chunks + 1)"""
    n = C.c_int()
    _ck(load().tau_guided_chunks(H, nstrips, slots_per_xcd, lmin, lmax, None, 0, C.byref(n)))
    out = (C.c_int * (n.value + 1))()
    _ck(load().tau_guided_chunks(H, nstrips, slots_per_xcd, lmin, lmax, out, n.value + 1, C.byref(n)))
    return list(out)


class Tau3DRing:
    """tau3d_ring_*: the Z-slab ring inside the library (csrc/ring.hip) around one slab handle `eng` (a Tau3D created
    with this rank's z0 / nzl).  transport: RING_RCCL (librccl, one device per rank), RING_HOST (staged through the
    rendezvous file, ranks may share a device), RING_LOCAL (world 1, device copies), RING_IPC (direct halos: neighbours' halo
    planes written through hipIpc mappings, RCCL for the 8-byte all-reduce), RING_IPC_HOSTMAX (the same with a host all-reduce).
    job_key: non-zero and unique per launch when world > 1; None derives one from the launcher's environment (torchrun's
    MASTER_PORT / TORCHELASTIC_RUN_ID — the same on every rank of a job, different between jobs)."""

    def __init__(self, eng, rank, world, transport=RING_RCCL, rendezvous=None, job_key=None):
        self._L = eng._L
        self.eng = eng
        self._r = C.c_void_p()
        rv = rendezvous.encode() if rendezvous else None
        if job_key is None:
            import zlib
            tag = "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TAU3D_JOB_KEY"))
            if world > 1 and tag == "|||":
                raise ValueError("Tau3DRing: world > 1 without a launcher environment (MASTER_PORT / TORCHELASTIC_RUN_ID / "
                                 "TAU3D_JOB_KEY): pass job_key=, unique per launch and the same on every rank")
            job_key = (zlib.crc32(tag.encode()) << 32 | zlib.crc32((rendezvous or "").encode())) or 1
        _ck(self._L.tau3d_ring_create(C.byref(self._r), eng._h, rank, world, transport, rv, job_key))

    def close(self):
        if getattr(self, "_r", None):
            self._L.tau3d_ring_destroy(self._r)
            self._r = None

    __del__ = close

    def prime(self):
        _ck(self._L.tau3d_ring_prime(self._r))

    def invalidate(self):
        _ck(self._L.tau3d_ring_invalidate(self._r))

    def step(self, n=1):
        _ck(self._L.tau3d_ring_step_async(self._r, n))
        return self

    def finish(self):
        _ck(self._L.tau3d_ring_finish(self._r))

    def barrier(self):
        _ck(self._L.tau3d_ring_barrier(self._r))

    def clock(self):
        c = Tau3DClock()
        _ck(self._L.tau3d_ring_get_clock(self._r, C.byref(c)))
        return c

    def timing_enable(self, on=True):
        _ck(self._L.tau3d_ring_timing_enable(self._r, 1 if on else 0))

    def timing_read(self):
        """(exchange_ms, allreduce_ms, steps): sums of the event intervals on the communication stream since timing_enable"""
        ex, ar, n = C.c_double(), C.c_double(), C.c_int()
        _ck(self._L.tau3d_ring_timing_read(self._r, C.byref(ex), C.byref(ar), C.byref(n)))
        return ex.value, ar.value, n.value

    def info(self):
        ver, ranks, edge = C.c_int(), C.c_int(), C.c_int()
        lib = C.create_string_buffer(512)
        _ck(self._L.tau3d_ring_info(self._r, C.byref(ver), C.byref(ranks), C.byref(edge), lib, 512))
        return {"rccl_version": ver.value, "comm_ranks": ranks.value, "edge_planes": edge.value, "librccl": lib.value.decode()}


class Tau3D:
    """3D hypersonic handle (tau3d_*).  Fields: xi, phix, phiy, phiz, lam, zet."""

    def __init__(self, nx, ny=None, nz=None, params=None, z0=0, nzl=None, device=0, stream=None):
        L = _require_device()
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        if params is None:
            params = Tau3DParams()
            L.tau3d_params_default(C.byref(params), nx, ny, nz)
        self.params = params
        self.nzl = params.nz if nzl is None else nzl
        self.z0 = z0
        self._h = C.c_void_p()
        _ck(L.tau3d_create(C.byref(self._h), C.byref(params), z0, self.nzl, device, stream))
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.tau3d_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def shape(self):
        return (self.nzl, self.params.ny, self.params.nx)

    def init(self, mode=0):
        _ck(self._L.tau3d_init(self._h, mode))

    def upload(self, fields):
        arrs = [_f32(f).reshape(-1) for f in fields]
        n = int(np.prod(self.shape))
        assert len(arrs) == 6 and all(a.size == n for a in arrs)
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
        _ck(self._L.tau3d_upload_state(self._h, ptrs))

    def download(self):
        arrs = [np.empty(self.shape, np.float32) for _ in range(6)]
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
        _ck(self._L.tau3d_download_state(self._h, ptrs))
        return arrs

    def download_planes(self, lo, hi):
        shp = (hi - lo, self.params.ny, self.params.nx)
        arrs = [np.empty(shp, np.float32) for _ in range(6)]
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
        _ck(self._L.tau3d_download_planes(self._h, lo, hi, ptrs))
        return arrs

    def upload_planes(self, lo, hi, fields):
        arrs = [_f32(f).reshape(-1) for f in fields]
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
        _ck(self._L.tau3d_upload_planes(self._h, lo, hi, ptrs))

    def solid(self):
        s = np.empty(self.shape, np.uint8)
        _ck(self._L.tau3d_download_solid(self._h, s.ctypes.data))
        return s

    def clock(self):
        c = Tau3DClock()
        _ck(self._L.tau3d_get_clock(self._h, C.byref(c)))
        return c

    def set_clock(self, t, d_tau, step=0):
        c = Tau3DClock(t, d_tau, 0.0, 0.0, 0.0, step)
        _ck(self._L.tau3d_set_clock(self._h, C.byref(c)))

    def step(self, n=1):
        c = Tau3DClock()
        _ck(self._L.tau3d_step(self._h, n, C.byref(c)))
        return c

    def step_async(self, n=1):
        _ck(self._L.tau3d_step_async(self._h, n))

    def step_explicit(self, dt, gain):
        m = C.c_float()
        _ck(self._L.tau3d_step_explicit(self._h, dt, gain, C.byref(m)))
        return m.value

    # ---- multi-GPU pieces
    def clock_begin_async(self):
        _ck(self._L.tau3d_clock_begin_async(self._h))

    def step_range_async(self, lo, hi, stream=None):
        _ck(self._L.tau3d_step_range_async(self._h, lo, hi, stream))

    def step_edges_async(self, depth, stream=None):
        _ck(self._L.tau3d_step_edges_async(self._h, depth, stream))

    def clock_end_async(self):
        _ck(self._L.tau3d_clock_end_async(self._h))

    def slab_begin_async(self):
        _ck(self._L.tau3d_slab_begin_async(self._h))

    def slab_edges_async(self, depth):
        _ck(self._L.tau3d_slab_edges_async(self._h, depth))

    def slab_interior_async(self, depth):
        _ck(self._L.tau3d_slab_interior_async(self._h, depth))

    def slab_end_async(self):
        _ck(self._L.tau3d_slab_end_async(self._h))

    def fill_halo_periodic_async(self):
        _ck(self._L.tau3d_fill_halo_periodic_async(self._h))

    def halo_ptr(self, kind, which, field, side):
        p = C.c_void_p()
        fn = self._L.tau3d_halo_send_ptr if kind == "send" else self._L.tau3d_halo_recv_ptr
        _ck(fn(self._h, which, field, side, C.byref(p)))
        return p.value

    def pack_halos_async(self, which):
        _ck(self._L.tau3d_pack_halos_async(self._h, which))

    def unpack_halos_async(self, which):
        _ck(self._L.tau3d_unpack_halos_async(self._h, which))

    def halo_buf(self, kind, side):
        p, n = C.c_void_p(), C.c_size_t()
        _ck(self._L.tau3d_halo_buf_ptr(self._h, {"send": 0, "recv": 1}[kind], side, C.byref(p), C.byref(n)))
        return p.value, n.value

    def max_ptr(self):
        """device address of two floats: max wavespeed, max |primitive| (all-reduced together by a slab ring)"""
        p = C.c_void_p()
        _ck(self._L.tau3d_max_ptr(self._h, C.byref(p)))
        return p.value

    def palette_indices(self, gamma=0.65):
        """8-bit palette indices of the last vis() volume (th3cs.cu:1199-1222); returns (uint8 array, min, max)"""
        out = np.empty((self.nzl, self.params.ny, self.params.nx), np.uint8)
        mn, mx = C.c_float(), C.c_float()
        _ck(self._L.tau3d_palette_indices(self._h, gamma, out.ctypes.data_as(C.c_void_p), C.byref(mn), C.byref(mx)))
        return out, mn.value, mx.value

    def state_written(self):
        _ck(self._L.tau3d_state_written(self._h))

    def debug_set_fmax_in(self, v):
        """test hook: the field-range word an x/y flux launch issued ahead of the clock reads (tau3d_debug_set_fmax_in)"""
        _ck(self._L.tau3d_debug_set_fmax_in(self._h, C.c_float(v)))

    def uniform_tiles(self):
        """(flagged, tiles, enabled) — tau3d_uniform_tiles: k_flux_xy tiles the last step found uniform (divergence exactly zero)"""
        u, n, on = C.c_long(), C.c_long(), C.c_int()
        _ck(self._L.tau3d_uniform_tiles(self._h, C.byref(u), C.byref(n), C.byref(on)))
        return u.value, n.value, bool(on.value)

    def tile_list_stats(self):
        """(mode, listed, tiles, checked, mismatches) — tau3d_tile_list_stats: the predicted-uniform tile list of the split step"""
        m, l, n, c, b = C.c_int(), C.c_long(), C.c_long(), C.c_long(), C.c_long()
        _ck(self._L.tau3d_tile_list_stats(self._h, C.byref(m), C.byref(l), C.byref(n), C.byref(c), C.byref(b)))
        return m.value, l.value, n.value, c.value, b.value

    def field_range(self):
        """(read_max, written_max, fast_form) — see tau3d_field_range"""
        a, b, f = C.c_float(), C.c_float(), C.c_int()
        _ck(self._L.tau3d_field_range(self._h, C.byref(a), C.byref(b), C.byref(f)))
        return a.value, b.value, bool(f.value)

    def sync(self):
        _ck(self._L.tau3d_sync(self._h))

    def set_split(self, on):
        """step = kernel pair k_flux_xy + k_update_z (True) or the fused k_step (False)"""
        _ck(self._L.tau3d_set_split(self._h, 1 if on else 0))

    def is_split(self):
        """True if a step of this handle is the kernel pair k_flux_xy + k_update_z (else the fused k_step)"""
        return bool(self._L.tau3d_is_split(self._h))

    VIS_MODES = ("schlieren_rho", "log_rho", "log_p", "speed", "mach", "vort_mag", "div", "q_criterion")

    def vis(self, mode):
        """the reference's k_vis field (mode 0..7 or a name from VIS_MODES) of the local planes, (nzl, ny, nx)"""
        mode = self.VIS_MODES.index(mode) if isinstance(mode, str) else int(mode)
        out = np.empty((self.nzl, self.params.ny, self.params.nx), np.float32)
        _ck(self._L.tau3d_vis(self._h, mode, out.ctypes.data))
        return out

    def slice_rgba(self, zslice, log_scale=False, a_gain=1.0):
        """slice_to_rgba of the last vis() field: (ny, nx, 4) uint8 RGBA, plus the slice's (min, max)"""
        px = np.empty((self.params.ny, self.params.nx), np.uint32)
        mn, mx = C.c_float(), C.c_float()
        _ck(self._L.tau3d_slice_rgba(self._h, int(zslice), int(bool(log_scale)), float(a_gain), px.ctypes.data,
                                     C.byref(mn), C.byref(mx)))
        return px.view(np.uint8).reshape(self.params.ny, self.params.nx, 4), mn.value, mx.value

    def outflow_reflection(self, nprobe=6):
        v = C.c_float()
        _ck(self._L.tau3d_outflow_reflection(self._h, int(nprobe), C.byref(v)))
        return v.value

    def timing_enable(self, on=True):
        _ck(self._L.tau3d_timing_enable(self._h, int(on)))

    def timing_read(self):
        ms, n, cells = C.c_double(), C.c_int(), C.c_double()
        _ck(self._L.tau3d_timing_read(self._h, C.byref(ms), C.byref(n), C.byref(cells)))
        return ms.value, n.value, cells.value

    def timing_span(self):
        """device ms from the start of the first timed interval to the end of the last one (kernels and gaps)"""
        ms = C.c_double()
        _ck(self._L.tau3d_timing_span(self._h, C.byref(ms)))
        return ms.value

    def timing_read_split(self):
        """(ms in k_flux_xy, ms in k_update_z, intervals) of the timed single-domain split steps"""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        _ck(self._L.tau3d_timing_read_split(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value


class Hypersonic2D:
    """2D Euler handle (tauh2_*): state = rho, mx, my, E fp32 arrays of shape (H, W) + u8 mask."""

    def __init__(self, W, H, device=0, stream=None, **kw):
        L = _require_device()
        p = H2Params()
        L.tauh2_params_default(C.byref(p), W, H)
        for k, v in kw.items():
            setattr(p, k, v)
        self.params = p
        self._h = C.c_void_p()
        _ck(L.tauh2_create(C.byref(self._h), C.byref(p), device, stream))
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.tauh2_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def shape(self):
        return (self.params.H, self.params.W)

    def init(self):
        _ck(self._L.tauh2_init(self._h))

    def upload(self, fields, mask=None):
        arrs = [_f32(f).reshape(-1) for f in fields]
        ptrs = (C.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        _ck(self._L.tauh2_upload(self._h, ptrs, None if m is None else m.ctypes.data))

    def download(self, with_mask=False):
        arrs = [np.empty(self.shape, np.float32) for _ in range(4)]
        ptrs = (C.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        m = np.empty(self.shape, np.uint8) if with_mask else None
        _ck(self._L.tauh2_download(self._h, ptrs, None if m is None else m.ctypes.data))
        return (arrs, m) if with_mask else arrs

    def step(self, n=1):
        t = C.c_double()
        _ck(self._L.tauh2_step(self._h, n, C.byref(t)))
        return t.value

    def step_async(self, n=1):
        _ck(self._L.tauh2_step_async(self._h, n))

    def step_explicit(self, dt):
        _ck(self._L.tauh2_step_explicit(self._h, dt))

    VIEW_MODES = ("log_rho", "log_p", "speed", "log_grad_rho", "asinh_vorticity", "mach", "log_p_over_rho")

    def render(self, view_mode=0):
        """the reference's frame tail (render_vals -> min/max -> pixels): (H, W, 4) uint8 RGBA, the scalar field
        (H, W) float32 and its (min, max) over the fluid"""
        view_mode = self.VIEW_MODES.index(view_mode) if isinstance(view_mode, str) else int(view_mode)
        H, W = self.shape
        px, val = np.empty((H, W), np.uint32), np.empty((H, W), np.float32)
        mn, mx = C.c_double(), C.c_double()
        _ck(self._L.tauh2_render(self._h, view_mode, px.ctypes.data, val.ctypes.data, C.byref(mn), C.byref(mx)))
        return px.view(np.uint8).reshape(H, W, 4), val, mn.value, mx.value

    def unit_eval(self):
        out = (C.c_float * 48)()
        _ck(self._L.tauh2_unit_eval(self._h, out))
        return np.array(out, np.float32)

    def unit_neighbors(self, x, y):
        out = (C.c_float * 9)()
        _ck(self._L.tauh2_unit_neighbors(self._h, x, y, out))
        return np.array(out, np.float32)

    def time(self):
        t, dt, m, s = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        _ck(self._L.tauh2_get_time(self._h, C.byref(t), C.byref(dt), C.byref(m), C.byref(s)))
        return {"t": t.value, "dt": dt.value, "maxs": m.value, "step": s.value}

    def sync(self):
        _ck(self._L.tauh2_sync(self._h))


class Sph2D:
    """2D WCSPH handle (tausph_*): pos/vel/acc as (N, 2) float32, s = ln rho, press."""

    def __init__(self, N, device=0, stream=None, **kw):
        L = _require_device()
        p = SphParams()
        L.tausph_params_default(C.byref(p), N)
        for k, v in kw.items():
            setattr(p, k, v)
        self.params = p
        self._h = C.c_void_p()
        _ck(L.tausph_create(C.byref(self._h), C.byref(p), device, stream))
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.tausph_destroy(self._h)
            self._h = None

    __del__ = close

    def reset_particles(self):
        _ck(self._L.tausph_reset_particles(self._h))

    def upload(self, pos, vel):
        pos, vel = _f32(pos), _f32(vel)
        _ck(self._L.tausph_upload(self._h, pos.ctypes.data, vel.ctypes.data))

    def download(self):
        n = self.params.N
        out = {k: np.empty((n, 2), np.float32) for k in ("pos", "vel", "acc")}
        out["s"] = np.empty(n, np.float32)
        out["press"] = np.empty(n, np.float32)
        out["cell"] = np.empty(n, np.int32)
        _ck(self._L.tausph_download(self._h, out["pos"].ctypes.data, out["vel"].ctypes.data, out["acc"].ctypes.data,
                                    out["s"].ctypes.data, out["press"].ctypes.data, out["cell"].ctypes.data))
        return out

    def grid(self):
        gx, gy = C.c_int(), C.c_int()
        cell, h, m = C.c_float(), C.c_float(), C.c_float()
        _ck(self._L.tausph_grid(self._h, C.byref(gx), C.byref(gy), C.byref(cell), C.byref(h), C.byref(m)))
        return {"Gx": gx.value, "Gy": gy.value, "cell": cell.value, "h": h.value, "mass": m.value}

    def dt(self):
        return self._L.tausph_dt(self._h)

    def substep(self, dt):
        _ck(self._L.tausph_substep_async(self._h, dt))
        self.sync()

    def step(self, n=1):
        _ck(self._L.tausph_step(self._h, n))

    def step_async(self, n=1):
        _ck(self._L.tausph_step_async(self._h, n))

    def clock(self):
        t, tau, s = C.c_float(), C.c_float(), C.c_int64()
        _ck(self._L.tausph_get_clock(self._h, C.byref(t), C.byref(tau), C.byref(s)))
        return {"t": t.value, "tau": tau.value, "step": s.value}

    def sync(self):
        _ck(self._L.tausph_sync(self._h))

    def rasterize(self, W, H):
        """k_rasterize: particle counts on a (2H, W) raster, y flipped (the ncurses view's input)"""
        g = np.empty((2 * H, W), np.int32)
        _ck(self._L.tausph_rasterize(self._h, W, H, g.ctypes.data))
        return g

    def state_written(self):
        """tausph_state_written: positions were written through the device pointers of tausph_state_ptrs — the next sub-step
        counts the particles into their cells again instead of trusting the count the last force pass made"""
        _ck(self._L.tausph_state_written(self._h))

    def rain_spawned(self):
        return int(self._L.tausph_rain_spawned(self._h))

    def count_pairs(self):
        """ordered pairs (i != j) inside the 2h support among the records of the last cell build"""
        n = C.c_int64()
        _ck(self._L.tausph_count_pairs(self._h, C.byref(n)))
        return n.value


class Lbm2D:
    """D2Q9 BGK lattice Boltzmann handle (taulbm_*): populations as (9, ny, nx) float32, solid mask (ny, nx) uint8."""

    def __init__(self, nx=512, ny=256, device=0, stream=None, **kw):
        L = _require_device()
        p = LbmParams()
        L.taulbm_params_default(C.byref(p))
        p.nx, p.ny = nx, ny
        for k, v in kw.items():
            setattr(p, k, v)
        self.params = p
        self._h = C.c_void_p()
        _ck(L.taulbm_create(C.byref(self._h), C.byref(p), device, stream))
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.taulbm_destroy(self._h)
            self._h = None

    __del__ = close

    def init(self):
        _ck(self._L.taulbm_init(self._h))

    def upload(self, f=None, solid=None):
        f = None if f is None else _f32(f)
        solid = None if solid is None else np.ascontiguousarray(solid, np.uint8)
        _ck(self._L.taulbm_upload(self._h, None if f is None else f.ctypes.data, None if solid is None else solid.ctypes.data))

    def download(self):
        f = np.empty((9, self.params.ny, self.params.nx), np.float32)
        solid = np.empty((self.params.ny, self.params.nx), np.uint8)
        _ck(self._L.taulbm_download(self._h, f.ctypes.data, solid.ctypes.data))
        return f, solid

    def set_drive(self, drive):
        _ck(self._L.taulbm_set_drive(self._h, float(drive)))

    def step(self, n=1):
        _ck(self._L.taulbm_step(self._h, n))

    def step_async(self, n=1):
        _ck(self._L.taulbm_step_async(self._h, n))

    def speed(self):
        s = np.empty((self.params.ny, self.params.nx), np.float32)
        _ck(self._L.taulbm_speed(self._h, s.ctypes.data))
        return s

    def sync(self):
        _ck(self._L.taulbm_sync(self._h))


class Flow2D:
    """Full Burgers ('burgers': phi_u, phi_v) or shallow-water ('sw': sigma, u, v) program (tauflow_*)."""

    def __init__(self, kind, nx, ny, device=0, stream=None, **kw):
        L = _require_device()
        self.kind = {"burgers": 0, "sw": 1}[kind]
        self.nf = 2 if self.kind == 0 else 3
        p = FlowParams()
        L.tauflow_params_default(C.byref(p), self.kind, nx, ny)
        for k, v in kw.items():
            setattr(p, k, v)
        if self.kind == 0 and p.oneD:
            p.ny = 1
        self.params = p
        self._h = C.c_void_p()
        _ck(L.tauflow_create(C.byref(self._h), C.byref(p), self.kind, device, stream))
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.tauflow_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def shape(self):
        return (self.params.ny, self.params.nx)

    def init(self):
        _ck(self._L.tauflow_init(self._h))

    def upload(self, fields):
        arrs = [_f32(f).reshape(-1) for f in fields]
        ptrs = (C.c_void_p * 3)(*([a.ctypes.data for a in arrs] + [None] * (3 - len(arrs))))
        _ck(self._L.tauflow_upload(self._h, ptrs))

    def download(self):
        arrs = [np.empty(self.shape, np.float32) for _ in range(self.nf)]
        ptrs = (C.c_void_p * 3)(*([a.ctypes.data for a in arrs] + [None] * (3 - self.nf)))
        _ck(self._L.tauflow_download(self._h, ptrs))
        return arrs

    def step(self, n=1):
        _ck(self._L.tauflow_step(self._h, n))

    def step_async(self, n=1):
        _ck(self._L.tauflow_step_async(self._h, n))

    def step_explicit(self, dt):
        _ck(self._L.tauflow_step_explicit(self._h, dt))

    def clock(self):
        t, tau, dt, w, s = C.c_float(), C.c_float(), C.c_float(), C.c_float(), C.c_int64()
        _ck(self._L.tauflow_get_clock(self._h, C.byref(t), C.byref(tau), C.byref(dt), C.byref(w), C.byref(s)))
        return {"t": t.value, "tau": tau.value, "dt": dt.value, "wavespeed": w.value, "step": s.value}

    def colehopf_relL2(self, t_now):
        r = C.c_double()
        _ck(self._L.tauflow_colehopf_relL2(self._h, t_now, C.byref(r)))
        return r.value

    def timer_start(self):
        """device time of what is enqueued until timer_stop (events on the handle's stream): tauflow_timer_*"""
        _ck(self._L.tauflow_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_double()
        _ck(self._L.tauflow_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def sync(self):
        _ck(self._L.tauflow_sync(self._h))


def row_bounds(ny, world, rank):
    """(y0, nyl): the contiguous rows of rank `rank` (taurow_bounds)"""
    y0, nyl = C.c_int32(), C.c_int32()
    _ck(load().taurow_bounds(ny, world, rank, C.byref(y0), C.byref(nyl)))
    return y0.value, nyl.value


def gs_pattern_host(nx, ny, seed=1337):
    """init_pattern of tau_gray_scott.cu:173-204 as two (ny, nx) host arrays (taugs_pattern_host)"""
    u, v = np.empty((ny, nx), np.float32), np.empty((ny, nx), np.float32)
    _ck(load().taugs_pattern_host(nx, ny, seed, u.ctypes.data, v.ctypes.data))
    return u, v


class RowRing:
    """taugs_ring_* / taulap_ring_*: the row-slab ring inside the library (csrc/ring.hip) around one GrayScott or Laplacian2D
    handle created with ny = nyl + 2 * halo rows.  transport: RING_RCCL, RING_HOST (ranks may share a device), RING_LOCAL (world 1)."""

    def __init__(self, eng, halo, rank, world, transport=RING_RCCL, rendezvous=None, job_key=0):
        self._L = eng._L
        self.eng, self.halo = eng, halo
        self._r = C.c_void_p()
        create = self._L.taugs_ring_create if isinstance(eng, GrayScott) else self._L.taulap_ring_create
        _ck(create(C.byref(self._r), eng._h, halo, rank, world, transport, rendezvous.encode() if rendezvous else None, job_key))

    def close(self):
        if getattr(self, "_r", None):
            self._L.taurow_ring_destroy(self._r)
            self._r = None

    __del__ = close

    def exchange(self):
        _ck(self._L.taurow_ring_exchange_async(self._r))

    def step(self, n):
        _ck(self._L.taurow_ring_step_async(self._r, n))

    def finish(self):
        _ck(self._L.taurow_ring_finish(self._r))

    def barrier(self):
        _ck(self._L.taurow_ring_barrier(self._r))

    def info(self):
        nyl, halo, ver, ranks, ex = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_long()
        _ck(self._L.taurow_ring_info(self._r, C.byref(nyl), C.byref(halo), C.byref(ex), C.byref(ver), C.byref(ranks)))
        return {"nyl": nyl.value, "halo": halo.value, "exchanges": ex.value, "rccl_version": ver.value, "comm_ranks": ranks.value}


class GrayScott:
    """Gray-Scott handle (taugs_*)."""

    def __init__(self, nx, ny, device=0, stream=None, **kw):
        L = _require_device()
        p = GSParams()
        L.taugs_params_default(C.byref(p), nx, ny)
        for k, v in kw.items():
            setattr(p, k, v)
        self.params = p
        self._h = C.c_void_p()
        _ck(L.taugs_create(C.byref(self._h), C.byref(p), device, stream))
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.taugs_destroy(self._h)
            self._h = None

    __del__ = close

    def init_pattern(self, seed=1337):
        _ck(self._L.taugs_init_pattern(self._h, seed))

    def upload(self, u, v):
        u, v = _f32(u), _f32(v)
        _ck(self._L.taugs_upload(self._h, u.ctypes.data, v.ctypes.data))

    def download(self):
        shp = (self.params.ny, self.params.nx)
        u, v = np.empty(shp, np.float32), np.empty(shp, np.float32)
        _ck(self._L.taugs_download(self._h, u.ctypes.data, v.ctypes.data))
        return u, v

    def state_ptrs(self):
        """device addresses of the current u, v arrays (they swap with every step)"""
        u, v = C.c_void_p(), C.c_void_p()
        _ck(self._L.taugs_state_ptrs(self._h, C.byref(u), C.byref(v)))
        return u.value, v.value

    def step(self, n=1):
        _ck(self._L.taugs_step(self._h, n))

    def step_async(self, n=1):
        _ck(self._L.taugs_step_async(self._h, n))

    def set_levels(self, levels):
        """time levels per launch: 0 default (4 fused), 1 one launch per step, 2..4"""
        _ck(self._L.taugs_set_levels(self._h, levels))

    def sync(self):
        _ck(self._L.taugs_sync(self._h))


class Laplacian2D:
    """Viscosity pass handle (taulap_*): kind 'burgers' or 'sw'."""

    def __init__(self, nx, ny, kind, nu, dt, dx=1.0, dy=1.0, u0=1.0, oneD=False, device=0, stream=None):
        L = _require_device()
        p = LapParams(nx, ny, dx, dy, nu, dt, u0)
        self.params = p
        self._h = C.c_void_p()
        k = {"burgers": 0, "sw": 1}[kind]
        _ck(L.taulap_create(C.byref(self._h), C.byref(p), k, int(oneD), device, stream))
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.taulap_destroy(self._h)
            self._h = None

    __del__ = close

    def upload(self, a, b):
        a, b = _f32(a), _f32(b)
        _ck(self._L.taulap_upload(self._h, a.ctypes.data, b.ctypes.data))

    def download(self):
        shp = (self.params.ny, self.params.nx)
        a, b = np.empty(shp, np.float32), np.empty(shp, np.float32)
        _ck(self._L.taulap_download(self._h, a.ctypes.data, b.ctypes.data))
        return a, b

    def state_ptrs(self):
        """device addresses of the current two arrays (they swap with every pass)"""
        a, b = C.c_void_p(), C.c_void_p()
        _ck(self._L.taulap_state_ptrs(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def step(self, n=1):
        _ck(self._L.taulap_step(self._h, n))

    def step_async(self, n=1):
        _ck(self._L.taulap_step_async(self._h, n))

    def sync(self):
        _ck(self._L.taulap_sync(self._h))
