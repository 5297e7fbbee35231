"""GPU: the 3D hypersonic step against THE REFERENCE'S OWN KERNELS running on the same MI355X.

oracle/_ref/tau_hypersonic_3d_cuda.co is the device code of the file north_star names — tau_hypersonic_3d_cuda.cu lines 1-1409
minus the raylib includes (4-5) and the Vector3 helpers (69-101), SURVEY 8c's line cut: k_build_solid_mask :759-770, k_init
:939-985, k_step :987-1359 — and oracle/_ref/th3cs.co that of the reference author's headless twin th3cs.cu; both compiled for
gfx950 by oracle/build_ref.sh from the sources where they lie in /root/reference.  refgpu.Ref3D() uses the named file unless
told otherwise (source="th3cs").  These tests pin, on identical inputs:
  * the CPU oracle (oracle/tau3d_oracle.c) against the reference kernels   -> the oracle is no longer "parity unpinned";
  * the engine (through the C-ABI) against the reference kernels           -> north_star's sentence, literally.
Tolerances are those of tests/parity.py (1e-5 relative on the conserved fields).
"""
import json
import os

import numpy as np
import pytest

from tests.parity import assert_parity, cells_beyond, report

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_checkvalues.json")))


@pytest.fixture(scope="module")
def refgpu():
    from oracle import refgpu as r
    if not (r.available("th3cs") and r.available("tau_hypersonic_3d_cuda")):   # built by __graft_entry__.build() where /root/reference exists; travels with the snapshot
        pytest.skip("oracle/_ref/{tau_hypersonic_3d_cuda,th3cs}.co absent: oracle/build_ref.sh has not run (needs /root/reference) — the reference-kernel pins are NOT checked")
    return r


@pytest.mark.parametrize("shape", [(32, 32, 32), (48, 40, 24), (64, 64, 64), (61, 61, 50), (14, 29, 35), (39, 27, 62), (160, 128, 96)])
def test_mask_and_init_match_the_reference_kernels(eng, oracle_built, refgpu, shape):
    nx, ny, nz = shape
    # the mask is an integer result: bit-exact against the reference kernel built without FMA contraction (the engine's and
    # the oracle's signed distance is mul-then-sub, as the source spells it) ...
    ri = refgpu.Ref3D(nx, ny, nz, ieee=True)
    r = refgpu.Ref3D(nx, ny, nz)
    e = eng.Tau3D(nx, ny, nz)
    o = oracle_built.Oracle3D(nx, ny, nz)
    ref_mask = ri.solid_mask()
    assert np.array_equal(e.solid(), ref_mask), "engine mask differs from k_build_solid_mask"
    assert np.array_equal(o.interior([o.solid])[0], ref_mask), "oracle mask differs from k_build_solid_mask"
    # ... and the reference's default build (hipcc and nvcc alike contract a*b+c) may only differ in cells whose centre lies ON
    # the sphere to rounding (e.g. plane 37 of 50: 37.5/50 - 0.5 = r exactly)
    diff = np.argwhere(r.solid_mask() != ref_mask)
    for z, y, x in diff:
        d = np.sqrt(((x + 0.5) / nx - 0.5) ** 2 + ((y + 0.5) / ny - 0.5) ** 2 + ((z + 0.5) / nz - 0.5) ** 2) - 0.25
        assert abs(d) < 2e-7, (x, y, z, d)
    print(shape, "cells where the contracted reference build flips the mask:", len(diff))
    ri.init()          # k_init writes the wall state into solid cells: compare on the same mask
    e.init(0)
    want = ri.download()
    for name, got in (("engine", e.download()), ("oracle", o.interior(o.init(0)))):
        for g, w in zip(got, want):
            np.testing.assert_allclose(g, w, rtol=0, atol=1e-6, err_msg=name)
    e.close()
    r.close()
    ri.close()


# rcp = True forces the reciprocal WENO weights (TAU3D_WENO_RCP, read at tau3d_create): with split = True that is
# flux_xy_body<false> / update_z_body<false>, the general-form bodies of the benchmarked kernel pair
@pytest.mark.parametrize("shape,mode,warm,split,rcp", [
    ((32, 32, 32), 0, 0, None, False), ((32, 32, 32), 1, 30, None, False), ((48, 40, 24), 1, 25, None, False), ((64, 64, 64), 1, 40, None, False),
    ((96, 64, 32), 1, 40, None, False), ((96, 64, 32), 1, 40, True, False), ((160, 128, 96), 1, 40, True, False), ((256, 192, 128), 1, 40, True, False),
    ((64, 64, 64), 1, 40, None, True), ((96, 64, 32), 1, 40, True, True), ((160, 128, 96), 1, 40, True, True), ((256, 192, 128), 1, 40, True, True)])
def test_single_step_engine_and_oracle_vs_reference_kernel(eng, oracle_built, refgpu, shape, mode, warm, split, rcp, monkeypatch):
    """ONE k_step of the reference on the state the engine developed; the engine's step and the oracle's step on the same
    input must both land within 1e-5 of it."""
    nx, ny, nz = shape
    if rcp:
        monkeypatch.setenv("TAU3D_WENO_RCP", "1")
    e = eng.Tau3D(nx, ny, nz)
    if split is not None:
        e.set_split(split)
    e.init(mode)
    if mode:
        e.set_clock(0.02, 1e-4)
    if warm:
        e.step(warm)
    c = e.clock()
    state = e.download()
    assert all(np.isfinite(a).all() and np.abs(a).max() < 30 for a in state)
    dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau))
    gain = 1.0 if mode else 0.0005
    if warm:
        assert e.field_range()[2] == (not rcp)       # which weight form the step below takes
    r = refgpu.Ref3D(nx, ny, nz)
    r.upload(state)
    m_ref = r.step(dt, gain)
    want = r.download()
    fluid = r.solid_mask() == 0
    # engine
    m_got = e.step_explicit(dt, gain)
    got = e.download()
    re = assert_parity(got, want, mask=fluid, what=f"engine vs reference kernel {shape}")
    assert m_got == pytest.approx(m_ref, rel=1e-5)
    for g, w in zip(got, want):      # solid cells: both copy the input through
        assert np.array_equal(g[~fluid], w[~fluid])
    # oracle (CPU) — small shapes only, it is a scalar loop
    if nx * ny * nz <= 96 * 64 * 32:
        o = oracle_built.Oracle3D(nx, ny, nz)
        st = o.from_interior(state)
        o.fill_halo_periodic(st)
        out = o.new_state()
        m_o = o.step_range(st, out, dt, gain)
        ro = assert_parity(o.interior(out), want, mask=fluid, what=f"oracle vs reference kernel {shape}")
        assert m_o == pytest.approx(m_ref, rel=1e-5)
        print("oracle-vs-ref", shape, {k: f"{v:.1e}" for k, v in ro.items() if k in ("rho", "mx", "E", "rho_ev", "lam/kappa", "zet/kappa")})
    print("engine-vs-ref", shape, {k: f"{v:.1e}" for k, v in re.items() if k in ("rho", "mx", "E", "rho_ev", "lam/kappa", "zet/kappa")})
    e.close()
    r.close()


def test_reference_kernel_trajectory_reproduces_the_recorded_checkvalues(eng, refgpu):
    """The survey's digit strings (tests/golden/ref_checkvalues.json: the reference's device code run on a host emulator) against
    the same code running as a GPU kernel here, and the engine beside both: 4 and 400 controller-driven steps at 32^3."""
    g4, g400 = GOLD["tau3d_32cube_4steps"], GOLD["tau3d_32cube_400steps"]
    r = refgpu.Ref3D(32)
    assert int(r.solid_mask().sum()) == g4["solid"]
    r.init()
    c = r.run(4)
    st = r.download()
    assert c["d_tau"] == pytest.approx(g4["d_tau"], rel=1e-6)
    assert c["maxs"] == pytest.approx(g4["maxs"], rel=1e-5)
    assert float(np.sum(st[0], dtype=np.float64)) == pytest.approx(g4["sum_xi"], rel=1e-6)
    assert float(np.sum(st[4], dtype=np.float64)) == pytest.approx(g4["sum_lam"], rel=1e-6)
    c = r.run(396)
    st = r.download()
    assert c["t"] == pytest.approx(g400["t"], rel=2e-3)
    assert float(np.sum(st[0], dtype=np.float64)) == pytest.approx(g400["sum_xi"], rel=2e-3)
    # the engine from the same start, against the reference kernels' trajectory (loose: 400 steps amplify rounding)
    e = eng.Tau3D(32)
    e.init(0)
    ce = e.step(400)
    se = e.download()
    assert ce.t == pytest.approx(c["t"], rel=2e-3)
    assert float(np.sum(se[0], dtype=np.float64)) == pytest.approx(float(np.sum(st[0], dtype=np.float64)), rel=2e-3)
    e.close()
    r.close()


# Neither 512^3 input of SURVEY §8(d) lives for ever — in the reference's own kernel exactly as in the engine: the reference's
# k_step under its own controller from the same impulsive start follows the engine's clock and max wavespeed to six digits for 60
# steps and overflows at step 65 (scripts/long_run_512_ref.py, profiles/r04/long_run_512_impulsive_reference_vs_engine.txt); the
# engine's lifetimes of both inputs: scripts/long_run_512.py, profiles/r04/long_run_512_*.txt.  The impulsive start (bench.py's headline
# input, timed over steps 25..45) runs away on the near-vacuum lee side of the sphere from step ~55 on (|primitive| 4e11 at step
# 60, max wavespeed 3.4e38 and infinite velocities by step 70, d_tau pinned at its 1e-7 floor).  The reference's own ramped start
# reaches gain 0.9 (t = 0.018) after ~3000 steps with |primitive| <= 660 and goes the same way before step 3250.  So the late,
# developed state that exists is the ramped one: 2500 steps (gain 0.8, bow shock standing, wake formed).
# At the END of bench.py's window (step 45, ten steps before the run-away) the lee-side cells that are about to go are already
# ill-conditioned: the engine and the reference kernel differ by up to 4e-5 in phix / m_x in a few of the 118 M fluid cells
# (87 cells, everything else <= 6e-6) — that case asserts the count and the bound instead of the literal 1e-5.
@pytest.mark.parametrize("start,warm,allow,rcp", [("impulsive", 25, 0, False), ("impulsive", 35, 0, False), ("impulsive", 45, 500, False),
                                                  ("ramped", 2500, 0, False), ("impulsive", 25, 0, True)])
def test_full_size_512_cubed_vs_reference_kernel(eng, refgpu, start, warm, allow, rcp, monkeypatch):
    """BASELINE.json's size, every cell of the 512^3 domain: one step of the kernel pair bench.py times against one k_step of the
    reference on the same developed state."""
    n = 512
    if rcp:        # the reciprocal-weight bodies of the kernel pair, at full size
        monkeypatch.setenv("TAU3D_WENO_RCP", "1")
    e = eng.Tau3D(n)
    assert e.is_split()
    if start == "impulsive":
        e.init(1)
        e.set_clock(0.02, 1e-4)
    else:
        e.init(0)
    e.step(warm)
    c = e.clock()
    state = e.download()
    dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau))
    gain = float(min(max(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) / np.float32(0.02), 0.0), 1.0))
    fr = e.field_range()
    assert max(fr[0], fr[1]) <= 2.5e3 and fr[2] == (not rcp), "the state must be a sane one (inside the fast WENO window)"
    r = refgpu.Ref3D(n)
    r.upload(state)
    m_ref = r.step(dt, gain)
    m_got = e.step_explicit(dt, gain)
    assert m_got == pytest.approx(m_ref, rel=1e-5)
    want = r.download()
    got = e.download()
    solid = r.solid_mask()
    r.close()
    e.close()
    worst = {}
    nbad = ncells = 0
    worst_cell = 0.0
    for z0 in range(0, n, 16):
        sl = slice(z0, z0 + 16)
        fluid = solid[sl] == 0
        g = [a[sl] for a in got]
        w = [a[sl] for a in want]
        try:
            rr = assert_parity(g, w, mask=fluid, what=f"512^3 planes {z0}..{z0 + 15}, {start} {warm} steps")
        except AssertionError:
            rr = report(g, w, fluid)
            nbad += 1
            k, wv = cells_beyond(g, w, fluid)
            ncells += k
            worst_cell = max(worst_cell, wv)
        for k, v in rr.items():
            worst[k] = max(worst.get(k, 0.0), v)
    print(f"512^3 after {warm} steps ({start} start, gain {gain:.3f}), engine vs reference kernel:", {k: f"{v:.2e}" for k, v in worst.items()},
          "| slabs out of tolerance:", nbad, "cells beyond 1e-5:", ncells, "worst", f"{worst_cell:.2e}")
    if allow == 0:
        assert nbad == 0, worst
    else:
        assert ncells <= allow and worst_cell <= 1e-4, (ncells, worst_cell, worst)


@pytest.mark.parametrize("shape,warm", [((32, 32, 32), 30), ((64, 64, 64), 40), ((96, 64, 32), 40)])
def test_schlieren_field_vs_reference_kernel(eng, refgpu, shape, warm):
    """k_schlieren_export of th3cs.cu (what its .4spl frames are made of) on the engine's developed state against tau3d_vis mode 0
    (|grad rho|, the same prim_at_xbc boundary states): relative to the operands of the differences, as tests/test_gpu_vis.py does"""
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(warm)
    state = e.download()
    r = refgpu.Ref3D(nx, ny, nz, source="th3cs")
    r.upload(state)
    want = r.schlieren().astype(np.float64)
    got = e.vis(0).astype(np.float64)
    fluid = r.solid_mask() == 0
    assert (got[~fluid] == 0).all() and (want[~fluid] == 0).all()
    rho = np.exp(state[0].astype(np.float64))
    scale = np.maximum(rho.max() / (2.0 / max(nx, ny, nz)), 1e-30)      # (q+ - q-) / (2 dx) of values that agree to rounding: eps (|q+| + |q-|) / (2 dx)
    err = np.abs(got - want)[fluid].max() / scale
    print("schlieren vs k_schlieren_export", shape, "max err / scale %.2e" % err, "max field %.3g" % want.max())
    assert err <= 2e-6 and want.max() > 0
    e.close()
    r.close()



@pytest.mark.parametrize("shape,warm", [((32, 32, 32), 30), ((61, 61, 50), 25), ((96, 64, 32), 40)])
def test_named_file_and_its_twin_step_alike(eng, refgpu, shape, warm):
    """k_step of tau_hypersonic_3d_cuda.cu (:987-1359) and of th3cs.cu (:716-1058) on the same developed state: the twin differs
    only in dead Tv solves and merged declarations, so the two code objects must produce the same six arrays and the same max
    wavespeed — which is why referees that exist only in th3cs (k_schlieren_export) still speak for the named file."""
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(warm)
    state = e.download()
    e.close()
    out = {}
    for ieee in (False, True):
        for src in ("3d_cuda", "th3cs"):
            r = refgpu.Ref3D(nx, ny, nz, source=src, ieee=ieee)
            assert r.source == src
            r.upload(state)
            m = r.step(2.0e-6, 1.0)
            out[src, ieee] = (m, r.download(), r.solid_mask())
            r.close()
        a, b = out["3d_cuda", ieee], out["th3cs", ieee]
        assert np.array_equal(a[2], b[2])
        same = a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
        worst = max(float(np.abs(x.astype(np.float64) - y).max()) for x, y in zip(a[1], b[1]))
        print(shape, "ieee" if ieee else "default", "builds of the two files:", "bit-identical" if same else f"max |diff| {worst:.2e}")
        assert same, worst
