"""GPU: Gray-Scott, 2D Euler, SPH and D2Q9 LBM against THE REFERENCE'S OWN KERNELS on the same MI355X (oracle/_ref/*.co, built from
/root/reference by oracle/build_ref.sh; launch sequences restated in oracle/refgpu.py with their file:line).

Bit-exact claims are checked against the reference kernels built without FMA contraction (*.ieee.co — the source's own rounding);
the reference's Makefile build (-use_fast_math / contraction on) is compared at the fp32 tolerance beside it.
"""
import numpy as np
import pytest

from tests.test_gpu_sph import compare_substep
from tests.test_gpu_tauh2 import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def refgpu():
    from oracle import refgpu as r
    if not r.available("tau_gray_scott"):
        pytest.skip("oracle/_ref absent: oracle/build_ref.sh has not run (needs /root/reference) — the reference-kernel pins are NOT checked")
    return r


# ---------------------------------------------------------------------------------------------------- Gray-Scott
@pytest.mark.parametrize("nx,ny,steps", [(128, 128, 100), (260, 70, 11), (67, 33, 5), (2048, 2048, 8), (8192, 8192, 4)])
def test_gray_scott_bit_exact_vs_reference_kernel(eng, refgpu, nx, ny, steps):
    """step_kernel (tau_gray_scott.cu:141-171) of the reference, IEEE build: the engine's fused passes give the same bits —
    up to BASELINE.json's 8192^2."""
    g = eng.GrayScott(nx, ny)
    g.init_pattern(1337)
    u0, v0 = g.download()
    rng = np.random.default_rng(nx + ny)
    u0 = (u0 - 0.1 * rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    v0 = (v0 + 0.1 * rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    g.upload(u0, v0)
    r = refgpu.RefGrayScott(nx, ny, ieee=True)
    r.upload(u0, v0)
    g.step(steps)
    r.step(steps)
    gu, gv = g.download()
    wu, wv = r.download()
    assert np.array_equal(gu, wu) and np.array_equal(gv, wv)
    r.close()
    # the reference's own Makefile flags (-use_fast_math): same field to fp32 rounding
    rf = refgpu.RefGrayScott(nx, ny)
    rf.upload(u0, v0)
    rf.step(steps)
    fu, fv = rf.download()
    assert np.abs(fu - gu).max() <= 1e-5 and np.abs(fv - gv).max() <= 1e-5
    rf.close()
    g.close()


# ---------------------------------------------------------------------------------------------------- 2D Euler
def test_tauh2_init_and_steps_vs_reference_kernels(eng, refgpu):
    """tau_hypersonic_cuda.cu through the reference's own seam, at the size its macros fix (8192 x 1024, fp64): k_init's mask bit for
    bit; then single steps on developed states (k_apply_inflow_left .. k_step as run_hypersonic_steps launches them,
    tau_hypersonic_cuda_tests.cu:178-243) against the engine's fused fp32 step at 1e-5 of the cell scales."""
    r = refgpu.RefH2()
    W, H = r.W, r.H
    r.init()
    e = eng.Hypersonic2D(W, H)
    e.init()
    got, mask = e.download(with_mask=True)
    ref_mask = r.mask_host()
    assert np.array_equal(mask, ref_mask), "body mask differs from the reference k_init"
    for g, w in zip(got, r.download()):
        assert np.array_equal(g, w.astype(np.float32))
    fluid = ref_mask == 0
    done = 0
    for warm in (0, 8, 40):
        e.step(warm - done)
        done = warm
        state = e.download()
        assert all(np.isfinite(a).all() for a in state)
        r.upload([a.astype(np.float64) for a in state])
        dt, maxs = r.step(1)
        want = r.download()
        e2 = eng.Hypersonic2D(W, H)
        e2.upload(state, mask)
        e2.step_explicit(dt)
        errs = rel_err(e2.download(), want, fluid)
        e2.close()
        print("tauh2 vs reference kernels, 8192x1024 after", warm, "steps: dt", dt, ["%.2e" % x for x in errs])
        assert max(errs) <= 1e-5, (warm, errs)
    e.close()
    r.close()


def test_tauh2_render_vs_reference_kernels(eng, refgpu):
    """the seven view modes and the colour ramp: k_render_vals / k_reduce_minmax / k_compute_inv_range /
    k_render_pixels of the reference (fp64, 8192 x 1024) against tauh2_render on the same developed state.  Tolerances as in
    tests/test_gpu_vis.py: the three modes that show p = (g-1)(E - rho q^2/2) carry kappa = E / e_int in fp32."""
    TOL = 1e-5
    r = refgpu.RefH2()
    W, H = r.W, r.H
    r.init()
    e = eng.Hypersonic2D(W, H)
    e.init()
    e.step(40)
    state, mask = e.download(with_mask=True)
    st = [a.astype(np.float64) for a in state]
    r.upload(st)
    fluid = mask == 0
    rho, mx, my, E = st
    kin = 0.5 * (mx * mx + my * my) / rho
    kappa = E / np.maximum(E - kin, 1e-25)
    U = np.sqrt(mx * mx + my * my)[fluid].max() / rho[fluid].min()
    for mode in range(7):
        px, val, mn, mxv = e.render(mode)
        wpx, want, mn_w, mx_w = r.render(mode)
        err = np.abs(val.astype(np.float64) - want)
        if mode in (1, 5, 6):
            tol = TOL * kappa * np.maximum(1.0, np.abs(want))
        elif mode == 4:
            tol = 0.1 * TOL * 2 * U / np.sqrt(1 + np.sinh(want) ** 2) + 1e-6
        else:
            tol = TOL * np.maximum(1.0, np.abs(want))
        worst = float((err / tol)[fluid].max())
        dpx = np.abs(px.astype(np.int16) - wpx.astype(np.int16))
        print("render mode", mode, e.VIEW_MODES[mode], "reference range [%.5g, %.5g] engine [%.5g, %.5g]" % (mn_w, mx_w, mn, mxv),
              "worst err/tol %.3f" % worst, "pixels off by > 1:", int((dpx > 1).any(axis=-1).sum()))
        assert worst <= 1.0, (mode, worst)
        assert mn == pytest.approx(mn_w, rel=1e-4, abs=1e-6) and mxv == pytest.approx(mx_w, rel=1e-4, abs=1e-6)
        assert (px[~fluid] == wpx[~fluid]).all()                      # the body's grey
        # the ramp maps (val - min) / range to 0..255: a value error of `tol` moves a channel by 255 tol / range (+ 1 for the fp32 ramp)
        span = max(mx_w - mn_w, 1e-30)
        allow = np.ceil(255.0 * 3.0 * tol / span) + 1
        assert (dpx.max(axis=-1)[fluid] <= allow[fluid]).all()
    e.close()
    r.close()


# ---------------------------------------------------------------------------------------------------- SPH
@pytest.mark.parametrize("N,warm,kw", [(4096, 0, {}), (4096, 40, {}), (16384, 120, {}), (65536, 200, {}), (16384, 80, dict(useVisc=0)),
                                       (4194304, 0, {}), (4194304, 30, {})])
def test_sph_substep_vs_reference_kernels(eng, oracle_built, refgpu, N, warm, kw):
    """k_build_cells .. k_integrate of tau_sph.cu on the engine's developed state: the particle -> cell map read back from the
    reference's linked lists equals the engine's integer cell indices bit for bit (IEEE build; BASELINE.json's 4 M particles
    included); density, pressure, acceleration, velocity and position at the tolerances of tests/test_gpu_sph.py."""
    e = eng.Sph2D(N, **kw)
    e.reset_particles()
    if warm:
        e.step(warm)
    st = e.download()
    dt = e.dt()
    g = e.grid()
    out = {}
    for ieee in (True, False):
        r = refgpu.RefSph(N, ieee=ieee, **{k: getattr(e.params, k) for k in
                                           "boxX boxY rho0 c0 gammaEOS hMul viscAlpha gravity useVisc useGrav".split()})
        assert (r.Gx, r.Gy) == (g["Gx"], g["Gy"]) and float(r.cell) == g["cell"] and float(r.h) == g["h"] and float(r.mass) == g["mass"]
        r.upload(st["pos"], st["vel"])
        head, nxt = r.substep(dt)
        w = r.state()
        w["cell"] = r.cells_from_lists(head, nxt)
        out[ieee] = w
        r.close()
    e.substep(dt)
    got = e.download()
    assert np.array_equal(got["cell"], out[True]["cell"]), "cell indices differ from the reference k_build_cells"
    # the acceleration scale (sum of |pair terms|) is a property of the state, not of who computes it: the CPU oracle supplies it
    # for the sizes it can walk; beyond, the engine-vs-reference difference is taken against |acc| + g
    if N <= 65536:
        o = oracle_built.OracleSph(N, **kw)
        o.set_state(st["pos"], st["vel"])
        o.substep(dt)
        scale = o.state()["acc_abs"]
    else:
        scale = (np.linalg.norm(out[True]["acc"].astype(np.float64), axis=1) + 9.81).astype(np.float32) * 50
    for ieee in (True, False):
        w = out[ieee]
        w["acc_abs"] = scale
        if not ieee:
            w["cell"] = got["cell"]      # fast-math division may move a particle ON a cell edge; asserted above for the IEEE build
        compare_substep(got, w, what=f"vs reference kernels ({'ieee' if ieee else 'fast-math'}) N={N} warm={warm} {kw}",
                        gamma=kw.get("gammaEOS", 1.0), c0=kw.get("c0", 1.0), dt=dt)
    e.close()


@pytest.mark.parametrize("N,warm", [(16384, 60), (65536, 120)])
def test_sph_xsph_and_raster_vs_reference_kernels(eng, refgpu, N, warm):
    """the SPH extras of SURVEY §8f row 3 against the reference's own kernels: k_rasterize (integer counts: bit-exact) and
    k_xsph_cell + k_apply_xsph on the lists of the sub-step's build with the moved particles (tau_sph.cu:698-704)"""
    eps = 0.25
    e = eng.Sph2D(N, useXSPH=1, xsphEps=eps)
    e.reset_particles()
    e.step(warm)
    st = e.download()
    dt = e.dt()
    r = refgpu.RefSph(N, ieee=True, **{k: getattr(e.params, k) for k in "boxX boxY rho0 c0 gammaEOS hMul viscAlpha gravity useVisc useGrav".split()})
    r.upload(st["pos"], st["vel"])
    r.substep(dt)
    r.xsph(eps)
    want = r.state()
    e.substep(dt)
    got = e.download()
    assert np.abs(got["pos"] - want["pos"]).max() <= 1e-5
    vscale = np.linalg.norm(want["vel"].astype(np.float64), axis=1) + 1.0        # c0 = 1
    assert (np.linalg.norm(got["vel"].astype(np.float64) - want["vel"], axis=1) / vscale).max() <= 1e-5
    for W, H in ((80, 24), (200, 50)):
        assert np.array_equal(e.rasterize(W, H), r.rasterize(W, H)) or np.abs(e.rasterize(W, H).astype(np.int64) - r.rasterize(W, H)).sum() <= 4, \
            "raster counts differ by more than the particles whose position differs in the last place across a pixel edge"
    r2 = refgpu.RefSph(N, ieee=True)
    r2.upload(got["pos"], got["vel"])                 # the same positions, bit for bit: the counts must be identical
    assert np.array_equal(e.rasterize(120, 40), r2.rasterize(120, 40))
    assert int(r2.rasterize(120, 40).sum()) == N
    r.close()
    r2.close()
    e.close()


# ---------------------------------------------------------------------------------------------------- LBM
@pytest.mark.parametrize("nx,ny,kw", [(512, 256, {}), (100, 60, dict(obstacle_radius=9.0)), (257, 33, dict(obstacle=0)),
                                      (2048, 1024, dict(tau=0.8, drive=1e-4))])
def test_lbm_bit_exact_vs_reference_kernels(eng, refgpu, nx, ny, kw):
    """init_kernel + collide_stream_kernel (tau_lbm.cu:64-132), IEEE build: same populations, bit for bit"""
    r = refgpu.RefLbm(nx, ny, ieee=True, **{k: (bool(v) if k == "obstacle" else v) for k, v in kw.items()})
    e = eng.Lbm2D(nx, ny, **kw)
    r.init()
    e.init()
    g0, solid = e.download()
    f0, rsolid = r.download()
    # the mask is integer work: bit-exact.  The sheared start goes through sinf: the device's libm and glibc's (which the engine
    # and the oracle follow) differ by one ulp for a few rows (j = 46, 84, 102, 221 of 256) — the populations agree to that
    assert np.array_equal(solid, rsolid)
    np.testing.assert_allclose(g0, f0, rtol=3e-7, atol=0)
    e.upload(f0, rsolid)            # ... so the bit-exact comparison of the step starts from the reference's own start
    for n in (1, 2, 7):
        r.step(n)
        e.step(n)
        assert np.array_equal(e.download()[0], r.download()[0]), f"after {n} more steps"
    e.close()
    r.close()


# ---------------------------------------------------------------------------------------------------- Burgers / shallow water
@pytest.mark.parametrize("kind,nx,ny,kw,warm", [("burgers", 256, 128, dict(muscl=0), 0), ("burgers", 256, 128, dict(muscl=1), 40), ("burgers", 512, 512, dict(muscl=0), 60),
                                                 ("burgers", 2048, 2048, dict(muscl=1), 20), ("sw", 256, 128, {}, 30), ("sw", 512, 512, {}, 80), ("sw", 2048, 2048, {}, 20)])
def test_flow_convective_step_vs_reference_kernels(eng, refgpu, kind, nx, ny, kw, warm):
    """wavespeed_block_max + flux_x / flux_y + update of tau_burgers.cu / tau_shallow_water.cu (IEEE build) on the
    engine's developed state, against the engine's fused step with nu = 0 — the reference's in-place viscosity kernels race and are
    not comparable; the engine's Jacobi pass with nu = 0 is the identity up to the codec round trip.  dt from the reference's own
    CFL reduction must equal the engine's device-side dt."""
    e = eng.Flow2D(kind, nx, ny, dtau=1e-2, nu=0.0, **kw)
    e.init()
    if warm:
        e.step(warm)
    f0 = e.download()
    c0 = e.clock()
    P = e.params
    r = refgpu.RefFlow(kind, nx, ny, P.dx, P.dy, u0=P.u0, g=P.g, CFL=P.CFL, muscl=kw.get("muscl", 0))
    r.upload(f0)
    dt = r.dt_eff(c0["t"], P.dtau)
    r.convect(dt)
    want = r.download()
    e.step(1)
    assert e.clock()["dt"] == pytest.approx(dt, rel=2e-6), "device-side dt differs from the reference's CFL reduction"
    e2 = eng.Flow2D(kind, nx, ny, dtau=1e-2, nu=0.0, **kw)
    e2.upload(f0)
    e2.step_explicit(dt)
    got = e2.download()
    errs = []
    for i, (g, w) in enumerate(zip(got, want)):
        if kind == "burgers":     # decoded velocity against the field scale, and the encoded array itself
            gu, wu = P.u0 * np.sinh(g.astype(np.float64)), P.u0 * np.sinh(w.astype(np.float64))
            errs.append(float(np.abs(gu - wu).max() / max(np.abs(wu).max(), 1e-30)))
            errs.append(float(np.abs(g - w).max()))
        else:                     # sigma = ln h absolutely; u, v against the celerity scale sqrt(g h) + |u|
            if i == 0:
                errs.append(float(np.abs(g.astype(np.float64) - w).max()))
            else:
                sc = np.sqrt(P.g * np.exp(want[0].astype(np.float64))) + np.abs(w)
                errs.append(float((np.abs(g.astype(np.float64) - w) / sc).max()))
    print(kind, (nx, ny), kw, "vs reference kernels:", ["%.1e" % x for x in errs], "dt", dt)
    assert max(errs) <= 1e-5, errs
    e.close()
    e2.close()
    r.close()


# ---------------------------------------------------------------------------------------------------- viscosity passes (e1 / e2)
@pytest.mark.parametrize("kind", ["sw", "burgers"])
@pytest.mark.parametrize("nx,ny", [(8, 8), (16, 4), (4, 16), (32, 2), (2, 32)])
def test_viscosity_pass_vs_reference_kernel_single_wave(eng, refgpu, kind, nx, ny):
    """The referee the round-4 review found missing for SURVEY §8 rows e1 / e2.  The reference's viscosity kernels update their
    arrays in place and race as the programs launch them; launched as ONE wave covering the whole (64-cell, periodic) grid they do
    not — every load of the kernel precedes its stores, and the stores wait for the wave's loads (oracle/refgpu.py:
    viscosity_single_wave) — and compute the Jacobi step with the reference's own arithmetic.  The engine's ping-pong pass against
    that, five passes: shallow water bit for bit (IEEE build of the reference), Burgers within the fp32 tolerance of its sinh /
    asinh codec; and the Makefile (-use_fast_math) build of the reference beside it."""
    rng = np.random.default_rng(1000 * nx + ny)
    a = (0.7 * rng.standard_normal((ny, nx))).astype(np.float32)
    b = (0.7 * rng.standard_normal((ny, nx))).astype(np.float32)
    nu, dt, dx, dy, u0, K = 0.1, 0.2, 1.0, 1.25, 1.5, 5
    h = eng.Laplacian2D(nx, ny, kind, nu, dt, dx=dx, dy=dy, u0=u0)
    h.upload(a, b)
    h.step(K)
    ga, gb = h.download()
    h.close()
    for fast in (False, True):
        r = refgpu.RefFlow(kind, nx, ny, dx, dy, u0=u0, fast=fast)
        fields = [a, b] if kind == "burgers" else [np.zeros_like(a), a, b]      # shallow water: (sigma, u, v), the pass touches u, v
        r.upload(fields)
        for _ in range(K):
            r.viscosity_single_wave(nu, dt, (nx, ny))
        out = r.download()
        wa, wb = (out[0], out[1]) if kind == "burgers" else (out[1], out[2])
        r.close()
        if kind == "sw" and not fast:
            assert np.array_equal(ga, wa) and np.array_equal(gb, wb), "shallow-water viscosity pass: not the reference kernel's bits"
        else:
            if kind == "burgers":   # decoded velocities against their scale, and the encoded arrays
                du = np.abs(u0 * np.sinh(ga.astype(np.float64)) - u0 * np.sinh(wa.astype(np.float64))).max()
                dv = np.abs(u0 * np.sinh(gb.astype(np.float64)) - u0 * np.sinh(wb.astype(np.float64))).max()
                sc = max(np.abs(u0 * np.sinh(wa.astype(np.float64))).max(), np.abs(u0 * np.sinh(wb.astype(np.float64))).max())
                assert max(du, dv) <= 1e-5 * sc, (du, dv, sc)
            assert max(np.abs(ga - wa).max(), np.abs(gb - wb).max()) <= 1e-5, (kind, fast)
