"""GPU: visualisation fields of the 3D solver (tau3d_vis / tau3d_slice_rgba / tau3d_outflow_reflection, through the
C-ABI) against the CPU oracle's restatement of k_vis, slice_to_rgba and k_outflow_reflection_metric."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 2e-6   # measured ~1e-7..5e-7; the contract is 1e-5


def developed(eng, oracle_built, shape, warm):
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    o = oracle_built.Oracle3D(nx, ny, nz)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(warm)
    state = e.download()
    assert all(np.isfinite(a).all() for a in state)
    st = o.from_interior(state)
    o.fill_halo_periodic(st)
    return e, o, st


@pytest.mark.parametrize("shape,warm", [((32, 32, 32), 30), ((48, 40, 24), 25), ((64, 64, 64), 40)])
def test_vis_fields_match_oracle(eng, oracle_built, shape, warm):
    """All eight VisMode fields on a developed bow-shock state.  The gradient modes difference decoded
    primitives, so 1e-5 is taken relative to the operands of those differences (the oracle's `scale`):
    (q+ - q-)/(2 dx) of two values that agree to fp32 rounding is itself only defined to eps*(|q+|+|q-|)/(2 dx)."""
    e, o, st = developed(eng, oracle_built, shape, warm)
    fluid = o.interior([o.solid])[0] == 0
    for mode in range(8):
        got = e.vis(mode)
        want, scale = o.vis(st, mode)
        assert np.isfinite(got).all()
        assert (got[~fluid] == 0).all(), "solid cells show 0 (tau_hypersonic_3d_cuda.cu:810-813)"
        err = np.abs(got.astype(np.float64) - want)[fluid] / np.maximum(scale[fluid].astype(np.float64), 1e-30)
        print("vis mode", mode, e.VIS_MODES[mode], "max err/scale %.2e" % err.max(), "max|field| %.3g" % np.abs(want).max())
        assert err.max() <= TOL, (mode, float(err.max()))
    assert e.vis("mach").max() > 1.0   # the impulsive start is hypersonic: the test is not comparing zeros
    e.close()


@pytest.mark.parametrize("log_scale,a_gain", [(False, 1.0), (True, 0.6), (False, 3.0)])
def test_slice_rgba(eng, oracle_built, log_scale, a_gain):
    """slice_to_rgba of the engine's own field: byte-exact against the oracle mapping the SAME field when the
    ramp is linear; with log scaling device logf and libm logf may differ in the last place, so +-1 per channel."""
    e, o, st = developed(eng, oracle_built, (48, 40, 24), 25)
    vol = e.vis(0)
    for z in (-3, 0, 11, 23, 99):                      # out-of-range slices clamp, :1418
        got, mn, mx = e.slice_rgba(z, log_scale, a_gain)
        want, mn_w, mx_w = o.slice_rgba(vol, z, log_scale, a_gain)
        if not log_scale:
            assert (mn, mx) == (mn_w, mx_w)
            assert np.array_equal(got, want)
        else:
            assert mn == pytest.approx(mn_w, rel=1e-6, abs=1e-7) and mx == pytest.approx(mx_w, rel=1e-6)
            assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 1
        assert (got[..., 0] == got[..., 1]).all() and (got[..., 1] == got[..., 2]).all()   # grey ramp
    assert got[..., 0].max() == 255 and got[..., 0].min() == 0                               # full range used
    e.close()


def test_outflow_reflection_metric(eng, oracle_built):
    e, o, st = developed(eng, oracle_built, (32, 32, 32), 30)
    for nprobe in (1, 6, 40):
        got = e.outflow_reflection(nprobe)
        want = o.outflow_reflection(st, nprobe)
        assert got == pytest.approx(want, rel=1e-5, abs=1e-9)
    e.close()


def test_vis_on_slab_equals_single_domain(eng):
    """A Z-slab handle with exchanged halo planes produces the same field planes as the single domain."""
    nx, ny, nz = 32, 32, 32
    e = eng.Tau3D(nx, ny, nz)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(20)
    full = e.download()
    ref = e.vis(5)
    z0, nzl = 8, 12
    s = eng.Tau3D(nx, ny, nz, z0=z0, nzl=nzl)
    s.init(1)
    s.upload([a[z0:z0 + nzl] for a in full])
    s.upload_planes(-3, 0, [a[z0 - 3:z0] for a in full])
    s.upload_planes(nzl, nzl + 3, [a[z0 + nzl:z0 + nzl + 3] for a in full])
    assert np.array_equal(s.vis(5), ref[z0:z0 + nzl])
    e.close(); s.close()


# ------------------------------------------------------------------ the reference's own k_vis / k_outflow_reflection_metric /
# k_maxwavespeed_pre / k_schlieren (tau_hypersonic_3d_cuda.cu, oracle/_ref line-cut build) and its host slice_to_rgba as referees
@pytest.fixture(scope="module")
def ref3d():
    from oracle import refgpu as r
    if not r.available("tau_hypersonic_3d_cuda"):
        pytest.skip("oracle/_ref/tau_hypersonic_3d_cuda.co absent (oracle/build_ref.sh needs /root/reference) — the k_vis referees are NOT checked")
    return r


@pytest.mark.parametrize("shape,warm", [((32, 32, 32), 30), ((48, 40, 24), 25), ((64, 64, 64), 40), ((96, 80, 40), 40)])
def test_vis_fields_vs_reference_k_vis(eng, oracle_built, ref3d, shape, warm):
    """tau3d_vis modes 0-7 against k_vis (tau_hypersonic_3d_cuda.cu:800-905) launched on the engine's developed state; the
    tolerance is the one of the oracle test above — relative to the operands of the differences a gradient mode takes —
    and the oracle's restatement of k_vis is held to the same kernel in the same breath."""
    e, o, st = developed(eng, oracle_built, shape, warm)
    nx, ny, nz = shape
    r = ref3d.Ref3D(nx, ny, nz, source="3d_cuda")
    r.upload(e.download())
    fluid = r.solid_mask() == 0
    for mode in range(8):
        want = r.vis(mode).astype(np.float64)
        got = e.vis(mode).astype(np.float64)
        orc, scale = o.vis(st, mode)
        sc = np.maximum(scale[fluid].astype(np.float64), 1e-30)
        assert (want[~fluid] == 0).all() and (got[~fluid] == 0).all()
        err = (np.abs(got - want)[fluid] / sc).max()
        err_o = (np.abs(orc - want)[fluid] / sc).max()
        print("k_vis mode", mode, e.VIS_MODES[mode], "engine %.2e oracle %.2e of scale; max|field| %.3g" % (err, err_o, np.abs(want).max()))
        assert err <= 1e-5 and err_o <= 1e-5, (mode, err, err_o)
    # k_schlieren (:1361-1387): |grad rho| from xi alone — mode 0 in fluid cells away from the body and the x boundaries
    want = r.schlieren_xi().astype(np.float64)
    got = e.vis(0).astype(np.float64)
    solid = ~fluid
    near = solid.copy()
    for ax in range(3):
        near |= np.roll(solid, 1, ax) | np.roll(solid, -1, ax)
    far = ~near
    far[:, :, 0] = far[:, :, -1] = False
    rho = np.exp(e.download()[0].astype(np.float64))
    scale = rho.max() / (2.0 / max(shape))
    assert (np.abs(got - want)[far] / scale).max() <= 1e-5
    r.close()
    e.close()


@pytest.mark.parametrize("shape,warm", [((32, 32, 32), 30), ((48, 40, 24), 25), ((64, 64, 64), 40)])
def test_outflow_metric_and_wavespeed_vs_reference_kernels(eng, ref3d, shape, warm):
    """tau3d_outflow_reflection against k_outflow_reflection_metric (:1389-1408), and the max wavespeed the engine's step
    returns against k_maxwavespeed_pre (:909-937) evaluated on the state that step produced... the step's own max is over its
    INPUT state (k_step :1338-1356 reduces the cell it just read), so: pre(state) == maxs of the step taken from that state."""
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(warm)
    r = ref3d.Ref3D(nx, ny, nz, source="3d_cuda")
    r.upload(e.download())
    for nprobe in (0, 1, 6, 40):
        assert e.outflow_reflection(nprobe) == pytest.approx(r.outflow_reflection(nprobe), rel=1e-5, abs=1e-9), nprobe
    pre = r.maxwavespeed_pre()
    m_ref = r.step(2.0e-6, 1.0)
    m_eng = e.step_explicit(2.0e-6, 1.0)
    print(shape, "k_maxwavespeed_pre", pre, "k_step maxs", m_ref, "engine", m_eng)
    assert m_eng == pytest.approx(m_ref, rel=1e-5)
    r.close()
    e.close()


@pytest.mark.parametrize("log_scale,a_gain", [(False, 1.0), (True, 0.55), (True, 2.0), (False, 3.0)])
def test_slice_rgba_vs_reference_host_code(eng, log_scale, a_gain):
    """tau3d_slice_rgba against slice_to_rgba itself (tau_hypersonic_3d_cuda.cu:1416-1442, g++ build of the line cut in
    oracle/_ref/libref_hostmaps.so) fed the engine's own field: linear ramp byte-exact; with the log ramp device logf and libm
    logf may differ in the last place: +-1 per channel, and only in a sliver of the pixels."""
    from oracle import refcpu
    if not refcpu.available_hostmaps():
        pytest.skip("oracle/_ref/libref_hostmaps.so absent — slice_to_rgba is NOT checked against the reference")
    e = eng.Tau3D(48, 40, 24)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(25)
    for mode in (0, 4):
        vol = e.vis(mode)
        for z in (-3, 0, 11, 23, 99):
            got, mn, mx = e.slice_rgba(z, log_scale, a_gain)
            want = refcpu.slice_to_rgba(vol, z, log_scale, a_gain)
            if not log_scale:
                assert np.array_equal(got, want), (mode, z)
            else:
                d = np.abs(got.astype(np.int16) - want.astype(np.int16))
                assert d.max() <= 1 and (d != 0).mean() < 2e-2, (mode, z, int(d.max()), float((d != 0).mean()))
    e.close()


# ------------------------------------------------------------------ 2D solver: the seven view modes + colour ramp
@pytest.mark.parametrize("W,H,warm", [(512, 256, 60), (257, 96, 40), (100, 60, 25)])
def test_render_2d_matches_oracle(eng, oracle_built, W, H, warm):
    """tauh2_render against the fp64 restatement of k_render_vals/k_render_pixels on the same (fp32-valued)
    state.  The fp32 engine forms p = (g-1)(E - rho q^2/2) with relative error kappa*eps, kappa = E/e_int
    (~1e3 in the Mach-25 free stream), so the three modes that show p are compared at TOL * kappa; the
    vorticity mode differences velocities of magnitude U, so its bound is a multiple of U mapped through asinh'."""
    o = oracle_built.OracleH2(W, H)
    o.init()
    e = eng.Hypersonic2D(W, H)
    e.init()
    e.step(warm)
    st = [a.astype(np.float64) for a in e.download()]
    fluid = o.mask == 0
    rho, mx, my, E = st
    kin = 0.5 * (mx * mx + my * my) / rho
    kappa = E / np.maximum(E - kin, 1e-25)
    U = np.sqrt(mx * mx + my * my)[fluid].max() / rho[fluid].min()
    for mode in range(7):
        px, val, mn, mxv = e.render(mode)
        want, mn_w, mx_w = o.render(st, mode)
        err = np.abs(val.astype(np.float64) - want)
        if mode in (1, 5, 6):
            tol = TOL * kappa * np.maximum(1.0, np.abs(want))
        elif mode == 4:
            tol = 0.1 * TOL * 2 * U / np.sqrt(1 + np.sinh(want) ** 2) + 1e-6
        else:
            tol = TOL * np.maximum(1.0, np.abs(want))
        worst = float((err / tol)[fluid].max())
        print("render mode", mode, e.VIEW_MODES[mode], "range [%.4g, %.4g]" % (mn, mxv), "worst err/tol %.3f" % worst)
        assert worst <= 1.0, (mode, worst)
        assert (val[~fluid] == 0).all()
        assert mn == pytest.approx(float(val[fluid].min())) and mxv == pytest.approx(float(val[fluid].max()))
        # the colour ramp itself: oracle mapping of the ENGINE's scalar and range, fp64 vs fp32 ramp -> +-1 per channel
        want_px = o.render_pixels(val, mn, mxv)
        assert np.abs(px.astype(np.int16) - want_px.astype(np.int16)).max() <= 1
        assert (px[~fluid] == np.array([110, 110, 110, 255], np.uint8)).all() and (px[..., 3] == 255).all()
    e.close()
