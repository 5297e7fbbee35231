"""CPU: known answers for the visualisation restatements in oracle/ (k_vis, slice_to_rgba, the 2D view modes and
colour ramp).  The reference holds no fixtures for these functions and SURVEY §8c recorded no check-values for
them, so they are pinned here against closed forms instead (their shared helpers — decode, wall/inflow/outflow
states, cons_to_prim — are pinned by the step check-values in test_oracle_pins.py)."""
import numpy as np
import pytest


def shear_state(o, c):
    """no body, uniform rho = p = 0.02, u = c * y (cell centres), v = w = 0"""
    st = o.new_state()
    nzh, ny, nx = o.shape_h
    y = (np.arange(ny, dtype=np.float32) + 0.5) * np.float32(o.p.dy)
    u = np.broadcast_to((np.float32(c) * y)[None, :, None], (nzh, ny, nx))
    st[0][:] = np.log(np.float32(0.02))
    st[1][:] = np.arcsinh(u / np.float32(o.p.u_ref)).astype(np.float32)
    st[4][:] = np.log(np.float32(0.02))
    st[5][:] = np.log(np.float32(1e-3))
    return st


def test_kvis_closed_forms(oracle_built):
    P = oracle_built.P3()
    L = oracle_built.Oracle3D(16).L
    L.o3_params_default(oracle_built.C.byref(P), 24, 20, 16)
    P.sdf_r = -1.0                                   # no sphere: every cell is fluid
    o = oracle_built.Oracle3D(24, 20, 16, params=P)
    assert o.solid.sum() == 0
    c = 40.0
    st = shear_state(o, c)
    inner = (slice(None), slice(1, -1), slice(1, -1))   # away from the periodic seam in y and the x ghosts
    f = {m: o.vis(st, m)[0] for m in range(8)}
    assert np.allclose(f[0][inner], 0.0, atol=1e-4)                       # |grad rho| of uniform rho
    assert np.allclose(f[1], np.log(np.float32(1.02)), rtol=1e-6)          # log(1 + rho)
    assert np.allclose(f[2], np.log(np.float32(1.02)), rtol=1e-6)          # log(1 + p)
    y = (np.arange(20) + 0.5) / 20
    assert np.allclose(f[3], np.broadcast_to((c * y)[None, :, None], f[3].shape), rtol=2e-6)   # |u|
    a = np.sqrt(1.1 * 0.02 / 0.02)
    assert np.allclose(f[4], f[3] / a, rtol=2e-6)                          # Mach
    assert np.allclose(f[5][inner], c, rtol=2e-4)                          # |curl u| = |du/dy|
    assert np.allclose(f[6][inner], 0.0, atol=2e-2)                        # div u (du/dx of an x-uniform field)
    # pure shear: ||Omega||^2 = ||S||^2 = c^2 / 2, so Q = 0 — to the rounding of c^2 ~ 1.6e3
    assert np.abs(f[7][inner]).max() <= 1e-3 * c * c


def test_slice_to_rgba_closed_form(oracle_built):
    o = oracle_built.Oracle3D(16)
    nx, ny = 16, 16
    vol = np.zeros((3, ny, nx), np.float32)
    vol[1] = np.linspace(0, 1, nx * ny, dtype=np.float32).reshape(ny, nx)
    px, mn, mx = o.slice_rgba(vol, 1, False, 0.5)
    t = vol[1]
    assert (mn, mx) == (0.0, 1.0)
    assert np.array_equal(px[..., 0], (t * np.float32(255)).astype(np.uint8))
    assert np.array_equal(px[..., 0], px[..., 2])
    assert np.array_equal(px[..., 3], (np.clip(np.float32(0.5) * (t * t), 0, 1) * np.float32(255)).astype(np.uint8))
    px2, _, _ = o.slice_rgba(vol, 99, False, 0.5)         # clamps to the last slice: constant -> t = 0 everywhere
    assert (px2 == 0).all()
    pxl, mnl, mxl = o.slice_rgba(vol, 1, True, 1.0)        # log scale: log1p of the ramp
    assert mnl == 0.0 and mxl == pytest.approx(np.log(2.0), rel=1e-6)


def test_render2d_closed_forms(oracle_built):
    W, H = 64, 32
    o = oracle_built.OracleH2(W, H)
    o.init()
    o.mask[:] = 0
    g = o.p.gamma
    rho = np.full((H, W), 2.0)
    u = np.broadcast_to(np.linspace(1.0, 3.0, W)[None, :], (H, W)).copy()
    v = np.zeros((H, W))
    p = np.full((H, W), 0.5)
    st = [rho, rho * u, rho * v, p / (g - 1) + 0.5 * rho * (u * u + v * v)]
    val, mn, mx = o.render(st, 0)
    assert np.allclose(val, np.log(2.0)) and mn == pytest.approx(np.log(2.0))
    val, mn, mx = o.render(st, 1)
    assert np.allclose(val, np.log(0.5), rtol=1e-12)
    val, mn, mx = o.render(st, 2)
    assert np.allclose(val, u) and (mn, mx) == (pytest.approx(1.0), pytest.approx(3.0))
    val, _, _ = o.render(st, 5)
    assert np.allclose(val, u / np.sqrt(g * 0.5 / 2.0))
    val, _, _ = o.render(st, 6)
    assert np.allclose(val, np.log(0.25))
    val, _, _ = o.render(st, 4)                              # v = 0, u = u(x): no vorticity
    assert np.allclose(val[:, 1:-1], 0.0, atol=1e-12)
    # colour ramp end points and centre, get_color :692-704
    ramp = np.zeros((H, W))
    ramp[0, :3] = [0.0, 0.5, 1.0]
    px = o.render_pixels(ramp, 0.0, 1.0)
    assert tuple(px[0, 0]) == (0, 0, 255, 255) and tuple(px[0, 1]) == (127, 255, 127, 255) and tuple(px[0, 2]) == (255, 0, 0, 255)
    o.mask[3, 3] = 1
    assert tuple(o.render_pixels(ramp, 0.0, 1.0)[3, 3]) == (110, 110, 110, 255)


def test_sph_extras_oracle_closed_forms(oracle_built):
    """XSPH, rain and rasterize restatements (oracle/sph_oracle.cpp) have no reference check-values either."""
    N = 4096
    # XSPH: dvel_i = eps sum_j (m / rhoBar_ij) (v_j - v_i) W_ij is antisymmetric in (i, j), so it adds no net
    # momentum, and it is a smoothing: it cannot increase the velocity variance.  It is parked in acc (:699).
    o = oracle_built.OracleSph(N, useXSPH=1, xsphEps=0.5)
    p = oracle_built.OracleSph(N)
    for k in range(30):
        p.step(1)
    st = p.state()
    o.set_state(st["pos"], st["vel"])
    dt = p.dt()
    o.substep(dt); p.substep(dt)
    a, b = o.state(), p.state()
    dvel = a["vel"].astype(np.float64) - b["vel"]
    assert np.array_equal(a["pos"], b["pos"])                 # XSPH acts after the position update
    assert np.allclose(dvel, a["acc"], rtol=0, atol=2e-7) and np.abs(dvel).max() > 1e-4
    assert np.abs(dvel.sum(axis=0)).max() <= 1e-3 * np.abs(dvel).sum()
    assert a["vel"].astype(np.float64).var(axis=0).sum() < b["vel"].astype(np.float64).var(axis=0).sum()
    # rain: 0.02 N dt drops per sub-step accumulate in a carry (:707-709); drops land in the top band, moving down
    r = oracle_built.OracleSph(N, rain=1)
    r.step(40)
    s = r.state()
    n = r.rain_spawned()
    assert n > 0
    top = s["pos"][:, 1] > 0.85
    assert 0 < top.sum() <= n
    assert (s["pos"][top, 0] >= 0.1).all() and (s["pos"][top, 0] <= 0.9).all() and (s["vel"][top, 1] < 0).all()
    # rasterize: counts sum to N, y flipped (the dam sits at the bottom of the picture)
    g = oracle_built.OracleSph(N).rasterize(40, 12)
    assert g.shape == (24, 40) and g.sum() == N and g[:8].sum() == 0 and g[-6:].sum() > 0


def test_lbm_oracle_closed_forms(oracle_built):
    """oracle/lbm_oracle.c has no reference check-values (SURVEY §8c lists none for tau_lbm.cu): closed forms."""
    nx, ny = 64, 32
    # (a) no drive, no obstacle, fluid at rest: the equilibrium is a fixed point of collide + stream + bounce-back
    o = oracle_built.OracleLbm(nx, ny, obstacle=0, drive=0.0)
    f = o.init()
    w = np.float32([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)
    rest = np.broadcast_to(w[:, None, None], (9, ny, nx)).astype(np.float32).copy()
    assert np.array_equal(o.solid[0], np.ones(nx, np.uint8)) and o.solid[1:-1].sum() == 0      # channel walls only
    g = o.step(rest, 5)
    assert np.abs(g - rest).max() <= 2e-7
    # (b) mass is conserved by every step, with obstacle, shear start and drive
    o = oracle_built.OracleLbm(nx, ny, obstacle_radius=6.0, drive=1e-4)
    f = o.init()
    m0 = f.sum(dtype=np.float64)
    f = o.step(f, 25)
    assert abs(f.sum(dtype=np.float64) - m0) <= 1e-6 * m0
    # (c) a solid cell reflects: fout[opp q] = fin[q]
    opp = [0, 3, 4, 1, 2, 7, 8, 5, 6]
    rng = np.random.default_rng(5)
    r = rng.random((9, ny, nx)).astype(np.float32)
    out = o.step(r, 1)
    sj, si = np.argwhere(o.solid == 1)[7]
    assert all(out[opp[q], sj, si] == r[q, sj, si] for q in range(9))
    # (d) the drive accelerates the channel flow in +x
    o = oracle_built.OracleLbm(nx, ny, obstacle=0, drive=1e-4)
    o.init()                                                  # builds the wall mask
    f = o.step(rest, 60)
    ux = (f[1] + f[5] + f[8] - f[3] - f[6] - f[7])[1:-1].mean()
    assert ux > 1e-3
    # (e) speed: -1 in solids, |u| elsewhere
    s = o.speed(f)
    assert (s[0] == -1).all() and s[ny // 2].mean() == pytest.approx(ux, rel=0.5)
