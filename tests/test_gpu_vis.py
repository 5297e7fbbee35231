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
