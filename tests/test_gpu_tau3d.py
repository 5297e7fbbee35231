"""GPU: parity of the HIP 3D hypersonic step (through the C-ABI) against the CPU oracle."""
import json
import os

import numpy as np
import pytest

from tests.parity import assert_parity, report

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_checkvalues.json")))


def oracle_one_step(P, o, fields, dt, gain):
    st = o.from_interior(fields)
    o.fill_halo_periodic(st)
    out = o.new_state()
    m = o.step_range(st, out, dt, gain)
    return o.interior(out), m


# the last five have cell centres exactly on the sphere (e.g. z = 37 of 50: 37.5 / 50 - 0.5 = r): an FMA-contracted
# signed distance rounds those the other way (found by a random-shape sweep)
@pytest.mark.parametrize("shape", [(32, 32, 32), (48, 40, 24), (64, 64, 64), (40, 24, 16), (61, 61, 50), (14, 29, 35),
                                   (63, 45, 10), (39, 27, 62), (24, 41, 30)])
def test_init_and_mask_bit_exact(eng, oracle_built, shape):
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    o = oracle_built.Oracle3D(nx, ny, nz)
    for mode in (0, 1):
        e.init(mode)
        got = e.download()
        want = o.interior(o.init(mode))
        assert np.array_equal(e.solid(), o.interior([o.solid])[0]), "solid mask must be bit-exact"
        for g, w in zip(got, want):
            np.testing.assert_allclose(g, w, rtol=0, atol=1e-6)
    e.close()


@pytest.mark.parametrize("shape,mode,warm", [((32, 32, 32), 0, 0), ((32, 32, 32), 1, 30), ((48, 40, 24), 1, 25),
                                             ((64, 64, 64), 1, 40), ((40, 24, 16), 1, 20), ((96, 64, 32), 1, 40)])
def test_single_step_parity(eng, oracle_built, shape, mode, warm):
    """One k_step on identical input (developed by `warm` engine steps from the impulsive start)."""
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    o = oracle_built.Oracle3D(nx, ny, nz)
    e.init(mode)
    if mode:
        e.set_clock(0.02, 1e-4)
    if warm:
        e.step(warm)
    c = e.clock()
    state = e.download()
    # the reference solver itself blows up on coarse anisotropic grids after ~47 impulsive steps
    # (oracle and engine agree on that); parity is only meaningful on a sane input
    assert all(np.isfinite(a).all() and np.abs(a).max() < 30 for a in state)
    dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau))
    gain = 1.0 if mode else 0.0005
    want, m_want = oracle_one_step(oracle_built, o, state, dt, gain)
    m_got = e.step_explicit(dt, gain)
    got = e.download()
    fluid = o.interior([o.solid])[0] == 0
    r = assert_parity(got, want, mask=fluid, what=f"{shape} mode {mode}")
    # solid cells copy through bit-exactly
    for g, s in zip(got, state):
        assert np.array_equal(g[~fluid], s[~fluid])
    assert m_got == pytest.approx(m_want, rel=1e-5)
    print("parity", shape, {k: f"{v:.2e}" for k, v in r.items()})
    e.close()


def test_trajectory_matches_reference_checkvalues(eng):
    """400 controller-driven steps at 32^3 from the reference start: clock and sum(xi) land on the
    reference's recorded outputs (loose: trajectories amplify rounding)."""
    g4, g400 = GOLD["tau3d_32cube_4steps"], GOLD["tau3d_32cube_400steps"]
    e = eng.Tau3D(32)
    e.init(0)
    c = e.step(4)
    st = e.download()
    assert c.d_tau == pytest.approx(g4["d_tau"], rel=1e-6)
    assert c.maxs == pytest.approx(g4["maxs"], rel=1e-5)
    assert float(np.sum(st[0], dtype=np.float64)) == pytest.approx(g4["sum_xi"], rel=1e-6)
    assert float(np.sum(st[4], dtype=np.float64)) == pytest.approx(g4["sum_lam"], rel=1e-6)
    c = e.step(396)
    st = e.download()
    assert c.step == 400
    assert c.t == pytest.approx(g400["t"], rel=2e-3)
    assert float(np.sum(st[0], dtype=np.float64)) == pytest.approx(g400["sum_xi"], rel=2e-3)
    e.close()


def test_ranges_bit_exact(eng):
    """Stepping [0,nz) in one launch or as edge + interior ranges gives identical bits — the
    property the Z-slab overlap schedule relies on."""
    e = eng.Tau3D(64, 32, 48)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(12)
    base = e.download()
    c0 = e.clock()
    e.step(1)
    whole = e.download()
    e.upload(base)
    e.set_clock(c0.t, c0.d_tau, c0.step)
    e.clock_begin_async()
    e.fill_halo_periodic_async()
    e.step_range_async(0, 3)
    e.step_range_async(45, 48)
    e.step_range_async(3, 45)
    e.clock_end_async()
    e.sync()
    parts = e.download()
    for a, b in zip(whole, parts):
        assert np.array_equal(a, b)
    e.close()


def test_deterministic(eng):
    outs = []
    for _ in range(2):
        e = eng.Tau3D(64, 64, 32)
        e.init(1)
        e.set_clock(0.02, 1e-4)
        e.step(20)
        outs.append((e.download(), e.clock().as_dict()))
        e.close()
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    assert outs[0][1] == outs[1][1]


def test_full_size_slab_vs_oracle(eng, oracle_built):
    """BASELINE size 512^3: after a short impulsive warm-up, one step on the GPU; an 8-plane slab
    through the bow-shock region is recomputed by the oracle from the same input planes."""
    n = 512
    e = eng.Tau3D(n)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(12)
    c = e.clock()
    zc = n // 2 - 40   # cuts the sphere (r = 128 cells) and the shock in front of it
    inp = e.download_planes(zc - 3, zc + 8 + 3)
    dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau))
    e.step_explicit(dt, 1.0)
    got = e.download_planes(zc, zc + 8)
    o = oracle_built.Oracle3D(n, n, n, z0=zc, nzl=8)
    st = [np.ascontiguousarray(a) for a in inp]
    out = o.new_state()
    o.step_range(st, out, dt, 1.0)
    want = o.interior(out)
    fluid = o.interior([o.solid])[0] == 0
    assert fluid.sum() < fluid.size, "slab should intersect the body"
    r = assert_parity(got, want, mask=fluid, what="512^3 slab")
    print("512^3 slab parity", {k: f"{v:.2e}" for k, v in r.items()})
    e.close()
