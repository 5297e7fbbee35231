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


# split = None: the form tau3d_create picks (the fused k_step below 128^2 cells per plane); True: the kernel pair
# k_flux_xy + k_update_z — the step bench.py times at 512^3 — forced through tau3d_set_split
@pytest.mark.parametrize("shape,mode,warm,split", [
    ((32, 32, 32), 0, 0, None), ((32, 32, 32), 1, 30, None), ((48, 40, 24), 1, 25, None), ((64, 64, 64), 1, 40, None),
    ((40, 24, 16), 1, 20, None), ((96, 64, 32), 1, 40, None),
    # (planes of 160 x 128 / 256 x 192 with only 12 / 16 z planes — dz ten times dx — leave the sane range within 40 impulsive
    #  steps, in the oracle as in the engine: the forced-split cases keep the cells near cubic)
    ((96, 64, 32), 1, 40, True), ((160, 128, 96), 1, 40, True), ((256, 192, 128), 1, 40, True), ((48, 40, 24), 1, 25, True),
    ((32, 32, 32), 0, 0, True)])
def test_single_step_parity(eng, oracle_built, shape, mode, warm, split):
    """One step on identical input (developed by `warm` engine steps from the impulsive start)."""
    nx, ny, nz = shape
    e = eng.Tau3D(nx, ny, nz)
    if split is not None:
        e.set_split(split)
        assert e.is_split() == split
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


@pytest.mark.parametrize("warm,zoff", [(12, -40), (40, -4), (50, -4), (50, -40)])
def test_full_size_slab_vs_oracle(eng, oracle_built, warm, zoff):
    """BASELINE size 512^3 (the kernel pair bench.py times): after an impulsive warm-up, one step on the GPU; an 8-plane
    slab is recomputed by the oracle from the same input planes.  (12, -40): off-centre cut through sphere and shock;
    (40, -4): the planes through the sphere's centre — stagnation line and bow-shock stand-off.
    After 50 steps — the last sane state of this start: it runs away after ~55, in the reference's own kernel too
    (profiles/r04/long_run_512_*.txt; rounds 2-4 ran these cases after 60 and 100 steps, i.e. on states between 4e11 and
    3e38) — the impulsive start has evacuated the lee side of the sphere: the first fluid cells behind the wall sit at
    rho = 3e-5 (1/600 of the free stream) between a 0.1-density wall state and a 60-unit velocity jump, and a few of the
    1.7 M cells of a slab land at 1.2 - 1.5e-5 against the oracle.  Those two cases assert exactly that: at most 3 cells
    beyond 1e-5, none beyond 2.5e-5."""
    n = 512
    e = eng.Tau3D(n)
    assert e.is_split()
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(warm)
    c = e.clock()
    zc = n // 2 + zoff   # cuts the sphere (r = 128 cells) and the shock in front of it
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
    if warm == 50:
        from tests.parity import conserved
        worst = np.zeros(fluid.shape)
        for g, w in zip(got[:4], want[:4]):
            worst = np.maximum(worst, np.abs(g.astype(np.float64) - w))
        Ug, _ = conserved(got)
        Uw, sc = conserved(want)
        for g, w, s in zip(Ug[:5], Uw[:5], sc[:5]):
            worst = np.maximum(worst, np.abs(g - w) / s)
        worst = worst[fluid]
        print('cells beyond 1e-5:', int((worst > 1e-5).sum()), 'worst', worst.max())
        assert int((worst > 1e-5).sum()) <= 3 and worst.max() <= 2.5e-5, (int((worst > 1e-5).sum()), worst.max())
        e.close()
        return
    r = assert_parity(got, want, mask=fluid, what="512^3 slab")
    print("512^3 slab parity", {k: f"{v:.2e}" for k, v in r.items()})
    e.close()


# ---- WENO weight form: the step kernel takes the common-denominator weights when the state it reads
# ---- is within 2.5e3 in magnitude (W_FLIM, h3d.hip), the reciprocal form otherwise — both against the same oracle
def _explicit_vs_oracle(eng, oracle_built, shape, fields, dt, gain=1.0, split=None, **par):
    import ctypes
    nx, ny, nz = shape
    P = eng.Tau3DParams()
    eng.load().tau3d_params_default(ctypes.byref(P), nx, ny, nz)
    o = oracle_built.Oracle3D(nx, ny, nz)
    for k, v in par.items():
        setattr(P, k, v)
        setattr(o.p, k, v)
    o = oracle_built.Oracle3D(nx, ny, nz, params=o.p)          # rebuilds the mask with the changed parameters
    e = eng.Tau3D(nx, ny, nz, params=P)
    if split is not None:           # True: the kernel pair k_flux_xy + k_update_z (bench.py's step) whatever the plane size
        e.set_split(split)
        assert e.is_split() == split
    e.init(1)
    e.upload(fields)
    want, m_want = oracle_one_step(oracle_built, o, fields, dt, gain)
    m_got = e.step_explicit(dt, gain)
    got = e.download()
    rng = e.field_range()
    e.close()
    return got, want, o.interior([o.solid])[0] == 0, rng, (m_got, m_want)


def _developed(eng, shape, warm):
    e = eng.Tau3D(*shape)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(warm)
    st = e.download()
    c = e.clock()
    rng = e.field_range()
    e.close()
    return st, float(c.dt), rng


def _forced_reciprocal(fn):
    """run fn() with the engine told to use the reciprocal weights whatever the range (read at tau3d_create)"""
    os.environ["TAU3D_WENO_RCP"] = "1"
    try:
        return fn()
    finally:
        del os.environ["TAU3D_WENO_RCP"]


CONS = ("rho", "mx", "my", "mz", "E")


# split = True runs flux_xy_body<false> / update_z_body<false> — the reciprocal-weight bodies of the benchmarked kernel pair —
# against the oracle (round-3 review: they had only met it through the fused k_step)
@pytest.mark.parametrize("shape,split", [((48, 40, 24), None), ((48, 40, 24), True), ((160, 128, 24), True)])
def test_weight_form_follows_the_field_range(eng, oracle_built, shape, split):
    _explicit = _explicit_vs_oracle
    def _explicit_vs_oracle_(*a, **k):
        return _explicit(*a, split=split, **k)
    st, dt, rng = _developed(eng, shape, 25 if shape[0] < 100 else 12)
    assert rng[2] and 99.9 <= rng[0] < 2.5e3 and 99.9 <= rng[1] < 2.5e3          # Mach-100 run: fast form, |u| = 100 seen
    got, want, fluid, rng, m = _explicit_vs_oracle_(eng, oracle_built, shape, st, dt)
    assert rng[2]
    assert_parity(got, want, mask=fluid, what="fast form")
    got2, _, _, rng2, _ = _forced_reciprocal(lambda: _explicit_vs_oracle_(eng, oracle_built, shape, st, dt))
    assert not rng2[2]
    assert_parity(got2, want, mask=fluid, what="reciprocal form, forced")
    # same flow, pressure and vibrational energy lifted by 1e5 (no vibrational relaxation: exp(theta/T) - 1 at that
    # temperature is all cancellation): out of the fast window -> reciprocal form.  A Mach-0.3 flow now, so the
    # velocities are compared through the momenta (relative to rho (|u| + a)), not through asinh(u / u_ref).
    # (A lift of 1e7 puts e_vib jumps at 2e9, where the reference's own (eps + beta)^2 leaves fp32.)
    big = [a.copy() for a in st]
    big[4] += np.float32(np.log(1e5))
    big[5] += np.float32(np.log(1e5))
    got, want, fluid, rng, m = _explicit_vs_oracle_(eng, oracle_built, shape, big, dt * 1e-3, tau_vib=1e30)
    assert not rng[2] and rng[0] > 2.5e3
    rep = report(got, want, mask=fluid)
    assert all(rep[k] < 1e-5 for k in CONS) and rep["xi"] < 1e-5 and rep["lam"] < 1e-5, rep
    assert m[0] == pytest.approx(m[1], rel=1e-5)


@pytest.mark.parametrize("split", [None, True])
def test_fast_weights_at_the_edge_of_their_window(eng, oracle_built, split):
    """cell-to-cell jumps of the largest admitted size in u, v, w, p and e_vib at once, in every direction: t^4 is at
    the top of fp32 — finite, within tolerance of the oracle, and next to the reciprocal form"""
    shape = (32, 24, 16)
    nx, ny, nz = shape
    rng = np.random.default_rng(5)
    F = 2.3e3
    sgn = lambda: rng.choice([-1.0, 1.0], size=(nz, ny, nx))
    mag = lambda lo: np.where(rng.random((nz, ny, nx)) < 0.5, lo, F)
    r, p, ev = 1.0 + rng.random((nz, ny, nx)), mag(1.0), mag(1e-3)
    u, v, w = (sgn() * mag(1.0) for _ in range(3))
    fields = [np.log(r), np.arcsinh(u / 10.0), np.arcsinh(v / 10.0), np.arcsinh(w / 10.0), np.log(p), np.log(ev)]
    fields = [a.astype(np.float32) for a in fields]
    run = lambda: _explicit_vs_oracle(eng, oracle_built, shape, fields, 1e-8, split=split, tau_vib=1e30)
    got, want, fluid, fr, m = run()
    assert fr[2] and 2e3 < fr[0] <= 2.5e3, fr
    assert all(np.isfinite(g).all() for g in got) and all(np.isfinite(w_).all() for w_ in want)
    rep = report(got, want, mask=fluid)
    got2, _, _, fr2, _ = _forced_reciprocal(run)
    assert not fr2[2]
    rep2 = report(got2, want, mask=fluid)
    print("edge", {k: (f"{rep[k]:.1e}", f"{rep2[k]:.1e}") for k in CONS})
    # input this wild (every face a 5e3 jump) is ill-conditioned for any fp32 evaluation — the reciprocal form, whose
    # arithmetic is the reference's, is itself 3e-5 from the oracle, and re-associating one product moves either
    # form by that much: the bar is the same order of magnitude, and no overflow
    worst, worst2 = max(rep[k] for k in CONS), max(rep2[k] for k in CONS)
    assert worst < 3e-4 and worst < 5 * worst2, (rep, rep2)


@pytest.mark.parametrize("split", [None, True])
def test_fast_weights_keep_small_smooth_corrections(eng, oracle_built, split):
    """the other end of the window: every stencil at the eps floor (t = 1e-6), slopes of 1e-7 — the products
    a_k q_k run through fp32 denormals and must still carry the high-order correction: the error against the oracle
    stays at the rounding level the reciprocal form has"""
    shape = (40, 32, 24)
    nx, ny, nz = shape
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    wave = np.sin(2 * np.pi * (x / nx + 2 * y / ny + z / nz))
    amp = 1e-6
    r = 0.02 * (1 + amp * wave); p = 0.02 * (1 + amp * np.roll(wave, 3, 2)); ev = 1e-3 * (1 + amp * np.roll(wave, 5, 1))
    u = 100.0 * (1 + amp * np.roll(wave, 7, 0)); v = 1e-4 * wave; w = -1e-4 * np.roll(wave, 2, 2)
    fields = [np.log(r), np.arcsinh(u / 10.0), np.arcsinh(v / 10.0), np.arcsinh(w / 10.0), np.log(p), np.log(ev)]
    fields = [a.astype(np.float32) for a in fields]
    run = lambda: _explicit_vs_oracle(eng, oracle_built, shape, fields, 2e-6, split=split, sdf_r=0.0)
    got, want, fluid, fr, m = run()
    assert fr[2]
    got2, _, _, fr2, _ = _forced_reciprocal(run)
    assert not fr2[2]
    rep, rep2 = report(got, want, mask=fluid), report(got2, want, mask=fluid)
    print("smooth", {k: (f"{rep[k]:.1e}", f"{rep2[k]:.1e}") for k in rep})
    for k in ("xi", "phix", "phiy", "phiz", "rho", "mx", "my", "mz", "E"):
        assert rep[k] < 2e-6 and rep[k] <= 1.5 * rep2[k] + 1e-7, (k, rep[k], rep2[k])


def test_lam_zet_error_is_the_oracles_own_two_build_spread(eng, oracle_built):
    """tests/parity.py compares lam = ln p and zet = ln e_vib at 1e-5 * kappa, kappa = (gamma-1) E / p.  The justification
    is measured here, not asserted: on the same developed state the ORACLE built with FMA contraction (how the
    reference's GPU build rounds) differs from the IEEE oracle by more than the literal 1e-5 in lam — and the engine's
    error against the IEEE oracle is of that size, not larger (tests/test_oracle_spread.py is the CPU half)."""
    from tests.test_oracle_spread import two_build_spread_3d
    n = 32
    e = eng.Tau3D(n)
    o = oracle_built.Oracle3D(n)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    e.step(40)
    state = e.download()
    dt, gain = 2.0e-6, 1.0
    st = o.from_interior(state)
    o.fill_halo_periodic(st)
    ieee, fma, fluid = two_build_spread_3d(oracle_built, o, st, dt, gain)
    e.step_explicit(dt, gain)
    got = e.download()
    for name, idx in (("lam", 4), ("zet", 5)):
        spread = np.abs(fma[idx].astype(np.float64) - ieee[idx])[fluid].max()
        err = np.abs(got[idx].astype(np.float64) - ieee[idx])[fluid].max()
        print(f"{name}: engine vs IEEE oracle {err:.2e}, FMA oracle vs IEEE oracle {spread:.2e}")
        assert err <= 3.0 * spread + 1e-6, (name, err, spread)
    assert np.abs(fma[4].astype(np.float64) - ieee[4])[fluid].max() > 1e-5     # the literal 1e-5 is not attainable for lam
    e.close()


@pytest.mark.parametrize("shape", [(32, 32, 32), (160, 128, 12)])      # fused kernel / split step
def test_create_upload_step_without_init(eng, shape):
    """tau3d_create builds the solid mask and defines the state buffers itself: a caller that goes create -> upload ->
    step (never tau3d_init) gets the same result as one that initialised and then uploaded the same state."""
    nx, ny, nz = shape
    a = eng.Tau3D(nx, ny, nz)
    a.init(1)
    a.set_clock(0.02, 1e-4)
    a.step(5)
    state = a.download()
    c = a.clock()
    b = eng.Tau3D(nx, ny, nz)                     # no init
    assert np.array_equal(b.solid(), a.solid())
    b.upload(state)
    b.set_clock(c.t, c.d_tau, c.step)
    ca, cb = a.step(3), b.step(3)
    for x, y in zip(a.download(), b.download()):
        assert np.array_equal(x, y)
    assert (ca.t, ca.d_tau, ca.maxs) == (cb.t, cb.d_tau, cb.maxs)
    a.close()
    b.close()


def test_split_step_equals_fused_step_to_rounding(eng):
    """The two-kernel step (k_flux_xy + k_update_z over the primitive cache, cell-centred WENO on every axis) and the
    fused k_step (face-centred WENO on x / y) are different groupings of the same arithmetic: one step from the same
    state agrees far inside the 1e-5 contract in the well-conditioned fields."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import fluid_sims_amd as f
e = f.Tau3D(96, 64, 24); e.init(1); e.set_clock(0.02, 1e-4); e.step(12)
np.save(sys.argv[1], np.stack(e.download()))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for split in ("0", "1"):
        path = os.path.join("/tmp", f"tau3d_split{split}.npy")
        env = dict(os.environ, TAU3D_SPLIT=split)
        subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=env)
        out[split] = np.load(path)
    r = report(list(out["1"]), list(out["0"]))
    print("split vs fused after 12 steps:", {k: f"{v:.1e}" for k, v in r.items()})
    for k in ("xi", "phix", "phiy", "phiz", "rho", "mx", "E"):
        assert r[k] <= 2e-5, (k, r[k])          # 12 steps of accumulated rounding between two legal groupings


def test_split_step_is_the_same_for_every_chunk_length():
    """k_update_z's chunk length (TAU3D_ZCHUNK planes per workgroup) only decides which workgroup computes a plane: every
    chunk addresses its planes as 32-bit offsets from its own first plane and primes its own ring, so the state after
    several steps must not change by a bit — odd lengths, a length that does not divide nz, one chunk for everything."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import fluid_sims_amd as f
e = f.Tau3D(160, 136, 40); e.init(1); e.set_clock(0.02, 1e-4); e.step(6)
np.save(sys.argv[1], np.stack(e.download()))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for zc in ("0", "1", "7", "16", "40"):
        path = os.path.join("/tmp", f"tau3d_zc{zc}.npy")
        env = dict(os.environ, TAU3D_SPLIT="1", TAU3D_ZCHUNK=zc)
        subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=env)
        outs.append(np.load(path))
    assert np.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_z_halos_written_by_the_step_itself(eng):
    """tau3d_step_async on the split path: k_update_z writes the new state's periodic z halos (no halo copy between steps).
    (a) after a run the halo planes ARE the opposite interior planes; (b) a run whose halos are declared stale half way (any
    plane transfer does: the next step then copies them with k_halo_periodic, as every step did before) ends bit-identical."""
    shape = (160, 128, 24)

    def fresh():
        e = eng.Tau3D(*shape)
        assert e.is_split()
        e.init(1)
        e.set_clock(0.02, 1e-4)
        return e
    a = fresh()
    a.step(7)
    nz = shape[2]
    for got, want in ((a.download_planes(-3, 0), a.download_planes(nz - 3, nz)), (a.download_planes(nz, nz + 3), a.download_planes(0, 3))):
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    want, wclock = a.download(), a.clock().as_dict()
    a.close()
    b = fresh()
    for _ in range(7):
        b.step_async(1)
        b.download_planes(0, 1)
    got, gclock = b.download(), b.clock().as_dict()
    b.close()
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert gclock == wclock


@pytest.mark.parametrize("split", [False, True])
def test_floored_density_face_keeps_the_energy_flux_finite(eng, oracle_built, split):
    """A face whose WENO5 density undershoots below zero is floored at 1e-30 (prim_floor, :565-571): the sound speed there is ~1e15
    and s_K (E* - E_K) is 1e15 times whatever rounding error E* - E_K carries.  The reference survives when its IEEE division
    returns (s_K E_K) / s_K = E_K exactly; the engine used to form U* with a reciprocal-multiply and subtract — an energy flux of
    +-1e7, p at its floor in one cell and 500 in its neighbour (found by scripts/fuzz_ref3d.py).  U* - U_K is now formed directly.
    The state is the y-line of that case (tests/golden/weno_undershoot_yline.json), uniform in x and z, no body."""
    import ctypes
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "weno_undershoot_yline.json")))
    line, dt = np.float32(g["line"]), g["dt"]
    nx, ny, nz = 64, 10, 16
    fields = [np.ascontiguousarray(np.broadcast_to(line[m][None, :, None], (nz, ny, nx))).astype(np.float32) for m in range(6)]
    P = eng.Tau3DParams()
    eng.load().tau3d_params_default(ctypes.byref(P), nx, ny, nz)
    P.sdf_r = -1.0
    P.dx, P.dy, P.dz = 1.0 / 104, 1.0 / 10, 1.0 / 26          # the cell sizes of the run the line comes from
    e = eng.Tau3D(nx, ny, nz, params=P)
    e.set_split(split)
    e.init(1)
    e.upload(fields)
    o = oracle_built.Oracle3D(nx, ny, nz)
    for k in ("sdf_r", "dx", "dy", "dz"):
        setattr(o.p, k, getattr(P, k))
    o = oracle_built.Oracle3D(nx, ny, nz, params=o.p)
    want, _ = oracle_one_step(oracle_built, o, fields, dt, 1.0)
    e.step_explicit(dt, 1.0)
    got = e.download()
    e.close()
    from tests.parity import undershoot_cells
    assert undershoot_cells(fields)[8, 4:6, 32].all(), "the fixture must hold the undershoot it was recorded for"
    sl = (slice(4, 12), slice(None), slice(8, 56))               # away from the x boundaries
    for m, name in enumerate(("xi", "phix", "phiy", "phiz", "lam", "zet")):
        d = np.abs(got[m][sl].astype(np.float64) - want[m][sl])
        # rows 4 and 5 sit on the degenerate face: the reference's own value there is one of several legal ones (its s_M is a ratio
        # below its denominator guard), so those two rows are held to 1e-3 of ln p — finite and sane is the point; every other row
        # to the usual bound
        assert d[:, [0, 1, 2, 3, 6, 7, 8, 9], :].max() <= (2e-3 if name in ("lam", "zet") else 1e-5), (name, float(d.max()))
        assert d[:, 4:6, :].max() <= (5e-2 if name in ("lam", "zet") else 1e-4), (name, float(d[:, 4:6, :].max()))


@pytest.mark.parametrize("shape,mode,steps", [((96, 64, 32), 1, 25), ((160, 128, 96), 1, 30), ((256, 192, 128), 1, 40), ((130, 70, 20), 0, 12),
                                              ((256, 256, 64), 0, 300)])
def test_uniform_region_exits_change_no_bit(eng, monkeypatch, shape, mode, steps):
    """The split step's uniform-region exits (include/taueng.h: tau3d_uniform_tiles) against the same kernels with the exits switched
    off (TAU3D_UNIFORM_EXITS=0): every field of every cell and the clock after every batch of steps, byte for byte — on starts
    that are uniform almost everywhere (the exits take most tiles), on developing bow shocks, and on the ramped start."""
    def run(on):
        if on:
            monkeypatch.delenv("TAU3D_UNIFORM_EXITS", raising=False)
        else:
            monkeypatch.setenv("TAU3D_UNIFORM_EXITS", "0")
        e = eng.Tau3D(*shape)
        e.set_split(True)
        e.init(mode)
        if mode:
            e.set_clock(0.02, 1e-4)
        out = []
        for k in (1, 2, steps - 3):
            e.step(k)
            c = e.clock()
            out.append((e.download(), (c.t, c.d_tau, c.maxs), e.uniform_tiles()))
        e.close()
        return out
    a, b = run(True), run(False)
    for (sa, ca, ua), (sb, cb, ub) in zip(a, b):
        assert ca == cb
        for x, y in zip(sa, sb):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        assert ua[2] is True and ub[2] is False and ub[0] == 0
    frac = [u[0] / u[1] for _, _, u in a]
    print(shape, "mode", mode, "uniform tiles after 1 / 3 /", steps, "steps:", ["%.3f" % f for f in frac])
    assert frac[0] > 0.2            # the exits were taken: the comparison is not between two runs of the full path


@pytest.mark.parametrize("shape,mode,steps", [((96, 64, 32), 1, 25), ((160, 128, 96), 1, 30), ((256, 192, 128), 1, 40), ((256, 256, 64), 0, 120),
                                              ((130, 70, 20), 0, 12)])
def test_predicted_uniform_tile_list_changes_no_bit(eng, monkeypatch, shape, mode, steps):
    """The predicted-uniform tile list (include/taueng.h: tau3d_tile_list_stats): k_flux_xy over the list of the tiles k_tile_predict could
    not clear (TAU3D_TILE_LIST=1, the default) against a k_flux_xy over every tile (=0), and the verifying mode (=2: every prediction
    checked against what k_flux_xy then found) — every field of every cell, the clock and the per-tile "divergence is zero" flags,
    byte for byte; a cell overwritten through tau3d_upload_state in the middle of a predicted region must be seen (the list is
    dropped).  The last shape has ragged tiles: the handle keeps no list and says so."""
    ragged = shape[0] % 32 != 0 or shape[1] % 16 != 0
    def run(tl):
        monkeypatch.setenv("TAU3D_TILE_LIST", str(tl))
        e = eng.Tau3D(*shape)
        e.set_split(True)
        e.init(mode)
        if mode:
            e.set_clock(0.02, 1e-4)
        out = []
        for i, k in enumerate((1, 2, steps - 3, 4, 6)):
            e.step(k)
            c = e.clock()
            st = e.download()
            out.append((st, (c.t, c.d_tau, c.maxs), e.uniform_tiles(), e.tile_list_stats()))
            if i == 2:      # a dent in the far corner of the grid, away from the body and the sponges: uniform there on these starts
                st = [f.copy() for f in st]
                st[4][shape[2] - 5, shape[1] - 9, shape[0] // 2 + 7] += 0.25
                e.upload(st)
        e.close()
        return out
    a, b, v = run(1), run(0), run(2)
    for (sa, ca, ua, la), (sb, cb, ub, lb), (sv, cv, uv, lv) in zip(a, b, v):
        assert ca == cb == cv
        for x, y, w in zip(sa, sb, sv):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
            assert np.array_equal(x.view(np.uint32), w.view(np.uint32))
        assert ua == ub == uv
        assert lb[0] == 0 and la[0] == (0 if ragged else 1) and lv[0] == (0 if ragged else 2)
        assert lv[4] == 0, "a predicted tile did not come out uniform: %r" % (lv,)
    if ragged:
        return
    la, lv = a[-1][3], v[-1][3]
    print(shape, "mode", mode, "listed / tiles:", [(x[3][1], x[3][2]) for x in a], "checked", lv[3])
    if shape[0] >= 160:                                # (three tiles across: the outer two touch ghost columns, the middle one has no flagged neighbour)
        assert lv[3] > 0 and 0 <= min(x[3][1] for x in a) < la[2]   # predictions were made (and checked), a list was shorter than the grid
    assert [x[3][1] for x in a] == [x[3][1] for x in v]   # the same list either way

