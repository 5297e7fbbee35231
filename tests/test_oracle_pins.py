"""CPU: pin the oracles against the reference outputs recorded in tests/golden/ref_checkvalues.json."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_checkvalues.json")))


def _sum(a):
    return float(np.asarray(a, np.float64).sum())


def same9(x, gold):
    """the survey printed fp32 scalars with 9 significant digits (round-trip exact for fp32)"""
    return float("%.9g" % x) == pytest.approx(gold, rel=1e-9)


def test_tau3d_oracle_4steps_matches_reference(oracle_built):
    g = GOLD["tau3d_32cube_4steps"]
    o = oracle_built.Oracle3D(32)
    st = o.run(o.init(0), 4)
    it = o.interior(st)
    assert int(o.interior([o.solid])[0].sum()) == g["solid"]
    assert same9(o.clock.d_tau, g["d_tau"])
    assert same9(o.clock.maxs, g["maxs"])
    assert _sum(it[0]) == pytest.approx(g["sum_xi"], rel=1e-11)
    assert _sum(it[4]) == pytest.approx(g["sum_lam"], rel=1e-11)


@pytest.mark.slow
def test_tau3d_oracle_400steps_matches_reference(oracle_built):
    g = GOLD["tau3d_32cube_400steps"]
    o = oracle_built.Oracle3D(32)
    st = o.run(o.init(0), 400)
    assert same9(o.clock.t, g["t"])
    assert same9(o.clock.d_tau, g["d_tau"])
    assert same9(o.clock.maxs, g["maxs"])
    assert _sum(o.interior(st)[0]) == pytest.approx(g["sum_xi"], rel=1e-11)


def test_tau3d_oracle_slab_equals_single_domain(oracle_built):
    """fixture (iv) of SURVEY §8c: a 2-slab decomposition reproduces the single domain bit for bit."""
    P = oracle_built
    o = P.Oracle3D(16, 16, 32)
    st = o.run(o.init(1), 2)         # impulsive start so the flow is not trivial
    o.fill_halo_periodic(st)
    dt, gain = 2e-5, 1.0
    ref = o.new_state()
    m_ref = o.step_range(st, ref, dt, gain)
    outs, ms = [], []
    for z0 in (0, 16):
        s = P.Oracle3D(16, 16, 32, z0=z0, nzl=16)
        loc = s.new_state()
        for f in range(6):
            for zl in range(-3, 19):
                loc[f][zl + 3] = st[f][(z0 + zl) % 32 + 3]
        out = s.new_state()
        ms.append(s.step_range(loc, out, dt, gain))
        outs.append(out)
    for f in range(6):
        got = np.concatenate([outs[0][f][3:-3], outs[1][f][3:-3]])
        assert np.array_equal(got, ref[f][3:-3])
    assert max(ms) == m_ref


def test_gray_scott_oracle_matches_reference(oracle_built):
    g = GOLD["gray_scott_128sq_100steps"]
    o = oracle_built.Oracle2D()
    p = o.gs_params(128, 128)
    u, v = o.gs_init(128, 128, g["seed"])
    u, v = o.gs_step(p, u, v, 100)
    assert same9(_sum(u), g["sum_u"])
    assert same9(_sum(v), g["sum_v"])


@pytest.mark.parametrize("kind", ["sw", "burgers"])
def test_laplacian_oracle_fourier_mode(oracle_built, kind):
    """No reference output exists for the viscosity passes (parity unpinned against the reference);
    pin analytically: a Fourier mode is an eigenvector of the periodic 5-point Laplacian."""
    o = oracle_built.Oracle2D()
    nx, ny, kx, ky = 64, 48, 3, 2
    x = np.arange(nx)[None, :]
    y = np.arange(ny)[:, None]
    mode = (np.cos(2 * np.pi * kx * x / nx) * np.cos(2 * np.pi * ky * y / ny)).astype(np.float32)
    amp = 0.3
    p = oracle_built.LapParams(nx, ny, 1.0, 1.0, 0.1, 0.2, 1.0)
    lam = (2 * np.cos(2 * np.pi * kx / nx) - 2) + (2 * np.cos(2 * np.pi * ky / ny) - 2)
    want = amp * mode * (1 + p.nu * p.dt * lam)
    if kind == "sw":
        a, b = o.lap_step("sw", p, amp * mode, -amp * mode)
        np.testing.assert_allclose(a, want, atol=2e-7)
        np.testing.assert_allclose(b, -want, atol=2e-7)
    else:
        phi = np.arcsinh(amp * mode / p.u0).astype(np.float32)
        a, b = o.lap_step("burgers", p, phi, -phi)
        np.testing.assert_allclose(p.u0 * np.sinh(a), want, atol=5e-7)
        np.testing.assert_allclose(p.u0 * np.sinh(b), -want, atol=5e-7)


def _seq_sum(x):
    s = 0.0
    for v in x:
        s += v
    return s


def test_tau2d_gpu_scheme_oracle_matches_reference(oracle_built):
    """tau_hypersonic_cuda.cu at 512 x 256, 4 steps, default_config: every recorded digit."""
    g = GOLD["tau2d_cuda_512x256_4steps_tile32x4"]
    o = oracle_built.OracleH2(512, 256)
    st = o.run(o.init(), 4)
    fl = o.mask.reshape(-1) == 0
    assert o.t == g["t"] and int(fl.sum()) == g["fluid"]
    assert _seq_sum(np.maximum(st[0].reshape(-1), 1e-25)[fl]) == g["sum_rho"]
    assert _seq_sum(st[1].reshape(-1)[fl]) == g["sum_mx"]
    assert _seq_sum(st[3].reshape(-1)[fl]) == g["sum_E"]


def test_tau2d_unit_known_answers(oracle_built):
    """known answers of tau_hypersonic_cuda_tests.cu:245-314, 389-442"""
    import ctypes as C
    u = GOLD["unit_known_answers_tau_hypersonic_cuda_tests"]
    o = oracle_built.OracleH2(8, 8)
    L = o.L
    d4 = C.c_double * 4
    out, ref, a = d4(), d4(), C.c_double()
    L.o2h_unit_roundtrip(C.c_double(1.1), d4(*u["roundtrip_cons"]), out)
    np.testing.assert_allclose(list(out), u["roundtrip_cons"], rtol=0, atol=1e-12)
    L.o2h_unit_flux(C.c_double(1.1), d4(*u["flux_prim"]), 0, out, C.byref(a))
    # The reference test expects Fx.E = 102 / Fy.E = -136 (tests:420, 424); with the gamma = 1.1 its
    # own default_config sets, (E + p) u = (5/0.1 + 25 + 5) * 3 = 240 — the reference's expectation is
    # inconsistent with the reference's code (the suite never ran in CI).  The mass and
    # momentum components are pinned to the reference's numbers, the energy one to the formula.
    np.testing.assert_allclose(list(out)[:3], u["flux_x"][:3], rtol=1e-12)
    assert out[3] == pytest.approx((5 / 0.1 + 0.5 * 2 * 25 + 5) * 3.0, rel=1e-14)
    assert a.value == pytest.approx(np.sqrt(1.1 * 5 / 2), rel=1e-14)
    L.o2h_unit_flux(C.c_double(1.1), d4(*u["flux_prim"]), 1, out, C.byref(a))
    np.testing.assert_allclose(list(out)[:3], u["flux_y"][:3], rtol=1e-12)
    assert out[3] == pytest.approx((5 / 0.1 + 0.5 * 2 * 25 + 5) * -4.0, rel=1e-14)
    assert L.o2h_unit_minmod(1.0, 2.0) == u["minmod_1_2"] and L.o2h_unit_minmod(-1.0, 2.0) == u["minmod_m1_2"]
    assert 0 < L.o2h_unit_mc(1.0, 1.2, 1.5) <= 1 and L.o2h_unit_mc(-1.0, 0.2, 1.0) == u["mc_limiter_m1_02_1"]
    for ax in (0, 1):
        L.o2h_unit_hllc(C.c_double(1.1), d4(*u["roundtrip_cons"]), ax, out, ref)
        np.testing.assert_allclose(list(out), list(ref), rtol=0, atol=1e-11)   # HLLC(U,U) = F(U)
    L.o2h_unit_inflow(C.c_double(1.1), C.c_double(25.0), out)
    np.testing.assert_allclose(list(out), [1.0, 25.0 * np.sqrt(1.1), 0.0, 1.0], rtol=1e-15)


def _neighbor_field(W, H, gamma):
    """the hand-built field of tau_hypersonic_cuda_tests.cu:567-600: rest gas, centre mx = 3, right mx = 7, cell above = body"""
    nf = GOLD["unit_known_answers_tau_hypersonic_cuda_tests"]["neighbors_field"]
    x, y = nf["x"], nf["y"]
    rho = np.full((H, W), nf["rest"]["rho"]); mx = np.zeros((H, W)); my = np.zeros((H, W))
    E = np.full((H, W), nf["rest"]["p"] / (gamma - 1.0))
    mask = np.zeros((H, W), np.uint8)
    mx[y, x] = nf["mx_center"]; mx[y, x + 1] = nf["mx_right"]; mask[y + 1, x] = nf["mask_up"]
    return x, y, rho, mx, my, E, mask


def test_tau2d_unit_known_answers_boundary(oracle_built):
    """the remaining reference-held numbers: clamps (tests:255-264, 395-401), enforce_positive_faces (:316-338, 455-478),
    SDF signs (:340-346, 480-484) and the nine neighbour lookups on the hand-built field (:348-371, 567-640)"""
    import ctypes as C
    u = GOLD["unit_known_answers_tau_hypersonic_cuda_tests"]
    o = oracle_built.OracleH2(64, 32)
    L = o.L
    d4, d2, d9 = C.c_double * 4, C.c_double * 2, C.c_double * 9
    out, eps = d4(), d2()
    gamma = 1.1
    L.o2h_unit_clamps(C.c_double(gamma), out, eps)
    eps_rho, eps_p = eps[0], eps[1]
    assert (eps_rho, eps_p) == (1e-25, 1e-25)                      # tau_hypersonic_cuda.cu:32-33
    assert abs(out[0] - eps_rho) <= 1e-30                          # prim_to_cons clamps rho floor
    assert out[1] >= eps_p / (gamma - 1.0)                         # ... keeps positive internal energy
    assert abs(out[2] - u["clamps"]["cons_to_prim_rho"]) <= 1e-12  # cons_to_prim keeps positive rho
    # "cons_to_prim clamps pressure floor" (tests:401) expects p >= EPS_P, but the reference's cons_to_prim floors the
    # INTERNAL ENERGY, p = (gamma-1) max(eint, EPS_P) (tau_hypersonic_cuda.cu:151-152): with its own gamma = 1.1 that is
    # 1e-26 < EPS_P — a second expectation of the never-run suite that its code does not meet (cf. the energy flux
    # above).  Pinned to the code's formula.
    assert out[3] == pytest.approx((gamma - 1.0) * eps_p, rel=1e-12) and out[3] > 0.0
    L.o2h_unit_enforce_positive(0, out)
    assert out[0] >= eps_rho and out[1] >= eps_p and out[2] >= eps_rho and out[3] >= eps_p
    L.o2h_unit_enforce_positive(1, out)
    np.testing.assert_allclose(list(out), u["enforce_positive_no_change"], rtol=0, atol=1e-12)
    sd = d2()
    L.o2h_unit_sdf(sd)
    assert sd[0] < 0.0 < sd[1]                                     # negative inside the body, positive outside
    W, H = 64, 32
    x, y, rho, mx, my, E, mask = _neighbor_field(W, H, gamma)
    P = o.p
    mach = P.mach
    assert (P.W, P.H, P.gamma) == (W, H, gamma)
    nine = d9()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    L.o2h_unit_neighbors(C.byref(P), vp(rho), vp(mx), vp(my), vp(E), vp(mask), x, y, nine)
    infl_mx = mach * np.sqrt(gamma)
    want = [1.0, infl_mx, 1.0, 7.0, -3.0, 1.0, infl_mx, -3.0, 1.0]
    tol = [1e-12, 1e-10, 1e-12, 1e-12, 1e-12, 1e-12, 1e-10, 1e-12, 1e-12]      # CHECK_NEAR tolerances of tests:613-631
    for got, w, t in zip(nine, want, tol):
        assert abs(got - w) <= t


def test_sph_oracle_matches_reference(oracle_built):
    """tau_sph.cu, N = 4096, 3 steps, rain off: every recorded digit"""
    g = GOLD["tau_sph_4096_3steps_norain"]
    o = oracle_built.OracleSph(4096)
    gr = o.grid()
    assert (gr["Gx"], gr["Gy"]) == (g["Gx"], g["Gy"])
    o.step(3)
    st = o.state()
    assert same9(_sum(st["pos"][:, 0]), g["sum_x"]) or float("%.12g" % _sum(st["pos"][:, 0])) == g["sum_x"]
    assert float("%.12g" % _sum(st["pos"][:, 0])) == g["sum_x"]
    assert float("%.12g" % _sum(st["pos"][:, 1])) == g["sum_y"]
    assert float("%.9g" % np.exp(st["s"].astype(np.float64)).mean()) == g["mean_rho"]


@pytest.mark.parametrize("muscl,bound", [(0, 4e-4), (1, 6e-5)])
def test_burgers_oracle_colehopf(oracle_built, muscl, bound):
    """No reference output exists for tau_burgers; analytic pin: the reference's own Cole-Hopf harness
    (tau_burgers.cu:256-273, 720-736) — the restated scheme converges to the exact 1-D solution."""
    o = oracle_built.OracleFlow("burgers", 512, 1, oneD=1, dtau=1e-3, muscl=muscl)
    f = o.init_burgers(colehopf=1, ck=4, ca=0.5)
    t, elapsed = np.float32(1.0), 0.0
    for _ in range(1500):
        dt = o.dt_eff(f, t)
        f = o.step(f, dt)
        elapsed += dt
        t = np.float32(t * np.exp(np.float32(1e-3)))
    assert o.colehopf_relL2(f[0], 4, 0.5, elapsed) < bound


def test_sw_oracle_conservation_and_rest(oracle_built):
    """No reference output exists for tau_sw; analytic pins: sum(h) is conserved by the periodic flux
    form, and a lake at rest stays exactly at rest."""
    o = oracle_built.OracleFlow("sw", 96, 64, dtau=1e-2)
    f = o.init_sw(H0=10.0, bumpAmp=0.5, bumpSigma=6.0, offx=5.0, offy=-3.0, asym=0.3, swirl=0.05, swirlRc=20.0)
    m0 = np.exp(f[0].astype(np.float64)).sum()
    t = np.float32(1.0)
    for _ in range(30):
        f = o.step(f, o.dt_eff(f, t))
        t = np.float32(t * np.exp(np.float32(1e-2)))
    assert np.exp(f[0].astype(np.float64)).sum() == pytest.approx(m0, rel=2e-6)
    rest = [np.full((64, 96), np.log(np.float32(7.0)), np.float32), np.zeros((64, 96), np.float32), np.zeros((64, 96), np.float32)]
    out = o.step(rest, 0.01)
    assert np.abs(out[1]).max() == 0 and np.abs(out[2]).max() == 0
    np.testing.assert_allclose(out[0], rest[0], atol=2e-7)
