"""GPU: Gray-Scott and the two viscosity passes (through the C-ABI) against the CPU oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_checkvalues.json")))


@pytest.mark.parametrize("nx,ny,steps", [(128, 128, 100), (256, 64, 17), (1024, 96, 9), (260, 70, 11),
                                         (130, 70, 7), (67, 33, 5), (4, 2, 3), (2048, 2048, 3),
                                         # the fused two-steps-per-pass kernel (nx >= 256, nx % 4 == 0): one owned strip is 248
                                         # columns, chunks are 32 rows; exact strip multiples, a one-lane last strip, minimal ny
                                         (256, 8, 2), (496, 40, 4), (744, 9, 6), (252 + 248, 33, 5), (8192, 40, 2),
                                         # small grids: the wave wraps around the row several times, rows wrap several times
                                         (8, 2, 9), (12, 3, 8), (128, 128, 12), (64, 5, 7), (200, 2, 4)])
def test_gray_scott_bit_exact(eng, oracle_built, nx, ny, steps):
    o = oracle_built.Oracle2D()
    p = o.gs_params(nx, ny)
    u0, v0 = o.gs_init(nx, ny, 1337)
    rng = np.random.default_rng(nx * 7919 + ny)
    u0 = (u0 - 0.1 * rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    v0 = (v0 + 0.1 * rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    g = eng.GrayScott(nx, ny)
    g.upload(u0, v0)
    g.step(steps)
    gu, gv = g.download()
    wu, wv = o.gs_step(p, u0, v0, steps)
    assert np.array_equal(gu, wu) and np.array_equal(gv, wv)
    g.close()


def test_gray_scott_reference_checkvalue(eng):
    gd = GOLD["gray_scott_128sq_100steps"]
    g = eng.GrayScott(128, 128)
    g.init_pattern(gd["seed"])
    g.step(100)
    u, v = g.download()
    assert float("%.9g" % u.sum(dtype=np.float64)) == pytest.approx(gd["sum_u"], rel=1e-9)
    assert float("%.9g" % v.sum(dtype=np.float64)) == pytest.approx(gd["sum_v"], rel=1e-9)
    g.close()


@pytest.mark.parametrize("dx", [0.7, 0.5, 2.0, 1.0])   # dx^2 a power of two -> exact-reciprocal path, else IEEE divide
def test_gray_scott_nondefault_params(eng, oracle_built, dx):
    o = oracle_built.Oracle2D()
    kw = dict(dx=dx, dt=0.25 * dx * dx, Du=0.16, Dv=0.08, feed=0.0367, kill=0.0649)
    p = o.gs_params(512, 200, **kw)
    u0, v0 = o.gs_init(512, 200, 42)
    g = eng.GrayScott(512, 200, **kw)
    g.upload(u0, v0)
    g.step(25)
    gu, gv = g.download()
    wu, wv = o.gs_step(p, u0, v0, 25)
    assert np.array_equal(gu, wu) and np.array_equal(gv, wv)
    g.close()


def test_gray_scott_full_size_translation_invariance(eng):
    """BASELINE size 8192^2: the step commutes with periodic shifts (bit-exact), which checks
    every strip/chunk seam and both wrap-arounds at full size."""
    n = 8192
    rng = np.random.default_rng(5)
    u0 = rng.random((n, n), dtype=np.float32)
    v0 = (0.3 * rng.random((n, n), dtype=np.float32)).astype(np.float32)
    g = eng.GrayScott(n, n)
    g.upload(u0, v0)
    g.step(3)
    u1, v1 = g.download()
    sy, sx = 37, 1021
    g.upload(np.roll(u0, (sy, sx), (0, 1)), np.roll(v0, (sy, sx), (0, 1)))
    g.step(3)
    u2, v2 = g.download()
    assert np.array_equal(np.roll(u1, (sy, sx), (0, 1)), u2)
    assert np.array_equal(np.roll(v1, (sy, sx), (0, 1)), v2)
    g.close()


def test_gray_scott_full_size_band_vs_oracle(eng, oracle_built):
    """8192^2: 3 steps on the GPU; a 64-row band is recomputed by the oracle (rows far enough
    from the band edge are exact because the stencil has radius 1 per step)."""
    n, steps, band, pad = 8192, 3, 64, 4
    o = oracle_built.Oracle2D()
    g = eng.GrayScott(n, n)
    g.init_pattern(1337)
    g.step(40)
    u0, v0 = g.download()
    g.step(steps)
    u1, v1 = g.download()
    j0 = n // 2 - 700     # through the edge of the centre square
    rows = slice(j0 - pad, j0 + band + pad)
    p = o.gs_params(n, band + 2 * pad)
    wu, wv = o.gs_step(p, u0[rows], v0[rows], steps)
    assert np.array_equal(wu[pad:-pad], u1[j0:j0 + band])
    assert np.array_equal(wv[pad:-pad], v1[j0:j0 + band])
    g.close()


@pytest.mark.parametrize("nx,ny", [(256, 128), (1000, 37), (66, 50), (2048, 512)])
def test_shallow_water_viscosity_bit_exact(eng, oracle_built, nx, ny):
    o = oracle_built.Oracle2D()
    rng = np.random.default_rng(nx + ny)
    a = rng.standard_normal((ny, nx)).astype(np.float32)
    b = rng.standard_normal((ny, nx)).astype(np.float32)
    p = oracle_built.LapParams(nx, ny, 1.5, 0.75, 0.05, 0.2, 1.0)
    h = eng.Laplacian2D(nx, ny, "sw", nu=p.nu, dt=p.dt, dx=p.dx, dy=p.dy)
    h.upload(a, b)
    h.step(4)
    ga, gb = h.download()
    wa, wb = o.lap_step("sw", p, a, b, 4)
    assert np.array_equal(ga, wa) and np.array_equal(gb, wb)
    h.close()


@pytest.mark.parametrize("nx,ny,oneD", [(256, 128, False), (1000, 37, False), (512, 8, True), (66, 50, False)])
def test_burgers_viscosity_parity(eng, oracle_built, nx, ny, oneD):
    """asinh-encoded fields: compare the decoded velocity u0*sinh(phi) at 1e-5 relative to the
    field scale (the oracle uses libm sinhf/asinhf, the kernel its own series/exp forms)."""
    o = oracle_built.Oracle2D()
    rng = np.random.default_rng(nx * 3 + ny)
    u0 = 2.0
    a = (rng.standard_normal((ny, nx)) * 1.2).astype(np.float32)
    b = (rng.standard_normal((ny, nx)) * 0.01).astype(np.float32)   # small values: series branch
    p = oracle_built.LapParams(nx, ny, 1.0, 1.0, 0.1, 0.2, u0)
    h = eng.Laplacian2D(nx, ny, "burgers", nu=p.nu, dt=p.dt, u0=u0, oneD=oneD)
    h.upload(a, b)
    h.step(3)
    ga, gb = h.download()
    wa, wb = o.lap_step("burgers", p, a, b, 3, oneD=oneD)
    for g, w in ((ga, wa), (gb, wb)):
        ug, uw = u0 * np.sinh(g.astype(np.float64)), u0 * np.sinh(w.astype(np.float64))
        assert np.abs(ug - uw).max() <= 1e-5 * max(np.abs(uw).max(), 1e-30)
        assert np.abs(g - w).max() <= 1e-5
    # small field keeps relative accuracy cell by cell
    ugb, uwb = np.sinh(gb.astype(np.float64)), np.sinh(wb.astype(np.float64))
    assert (np.abs(ugb - uwb) <= 1e-5 * np.abs(uwb) + 1e-9).all()
    h.close()


@pytest.mark.parametrize("nx,ny,passes", [(1024, 64, 8), (256, 40, 9), (2048, 2048, 4)])
def test_burgers_viscosity_fused_passes(eng, oracle_built, nx, ny, passes):
    """taulap_step(n) runs the Burgers pass in fused groups of up to four levels.  Between its levels a group does the
    reference's re-encode / decode round trip (phi = asinh(u/u0), u = u0 sinh(phi)) in registers, so n fused passes are
    BIT-IDENTICAL to n single passes (taulap_step(1) n times: the single-step kernel) — and within 1e-5 of the
    pass-by-pass oracle."""
    o = oracle_built.Oracle2D()
    rng = np.random.default_rng(nx + 3 * ny)
    u0 = 1.5
    a = (rng.standard_normal((ny, nx)) * 1.5).astype(np.float32)
    b = (rng.standard_normal((ny, nx)) * 0.02).astype(np.float32)
    p = oracle_built.LapParams(nx, ny, 1.0, 1.0, 0.1, 0.2, u0)
    h = eng.Laplacian2D(nx, ny, "burgers", nu=p.nu, dt=p.dt, u0=u0)
    h.upload(a, b)
    h.step(passes)
    ga, gb = h.download()
    h.upload(a, b)
    for _ in range(passes):
        h.step(1)                                 # one pass per call: the single-step kernel
    sa, sb = h.download()
    assert np.array_equal(ga, sa) and np.array_equal(gb, sb), "fused Burgers passes must equal single passes bit for bit"
    wa, wb = o.lap_step("burgers", p, a, b, passes)
    # against the oracle the contract is 1e-5 per pass on identical input (test_burgers_viscosity_parity); over a run of
    # passes the codec roundings (v_exp / v_log against libm: ~3e-6 per pass in phi at |phi| ~ 7, measured 1.3e-5 after
    # four) add up linearly
    tol = 5e-6 * passes
    for g, w in ((ga, wa), (gb, wb)):
        ug, uw = u0 * np.sinh(g.astype(np.float64)), u0 * np.sinh(w.astype(np.float64))
        assert np.abs(ug - uw).max() <= tol * max(np.abs(uw).max(), 1e-30)
        assert np.abs(g - w).max() <= tol
    # the small field (|b| ~ 0.02, series branch of sinh/asinh) is accurate against ITS OWN scale, not only against a's
    ugb, uwb = np.sinh(gb.astype(np.float64)), np.sinh(wb.astype(np.float64))
    assert np.abs(ugb - uwb).max() <= tol * np.abs(uwb).max()
    h.close()
