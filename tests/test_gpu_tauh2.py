"""GPU: fused fp32 2D Euler step (tauh2_*, through the C-ABI) against the fp64 oracle of the
reference's GPU scheme (tau_hypersonic_cuda.cu)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_checkvalues.json")))
GAMMA = 1.1
TOL = 1e-5   # north_star: conserved fields within 1e-5 relative (fp32 engine vs fp64 oracle, single step)


def scales(st):
    rho, mx, my, E = [np.asarray(a, np.float64) for a in st]
    r = np.maximum(rho, 1e-25)
    u, v = mx / r, my / r
    p = (GAMMA - 1) * np.maximum(E - 0.5 * r * (u * u + v * v), 1e-25)
    a = np.sqrt(GAMMA * p / r)
    mom = r * (np.sqrt(u * u + v * v) + a)
    return [r, mom, mom, np.abs(E)]


def rel_err(got, want, fluid):
    sc = scales(want)
    return [float((np.abs(np.asarray(g, np.float64) - w) / s)[fluid].max()) for g, w, s in zip(got, want, sc)]


@pytest.mark.parametrize("W,H", [(512, 256), (100, 60), (257, 96), (1024, 128)])
def test_init_mask_bit_exact(eng, oracle_built, W, H):
    o = oracle_built.OracleH2(W, H)
    want = o.init()
    e = eng.Hypersonic2D(W, H)
    e.init()
    got, mask = e.download(with_mask=True)
    assert np.array_equal(mask, o.mask), "body mask must be bit-exact"
    for g, w in zip(got, want):
        assert np.array_equal(g, w.astype(np.float32))
    e.close()


@pytest.mark.parametrize("W,H,warm", [(512, 256, 0), (512, 256, 40), (100, 60, 25), (257, 96, 60), (1024, 128, 120)])
def test_single_step_parity(eng, oracle_built, W, H, warm):
    o = oracle_built.OracleH2(W, H)
    o.init()
    e = eng.Hypersonic2D(W, H)
    e.init()
    if warm:
        e.step(warm)
    state = e.download()
    assert all(np.isfinite(a).all() for a in state)
    st64 = [a.astype(np.float64) for a in state]
    o.apply_inflow(st64)
    dt = o.dt_from_maxs(o.max_wavespeed(st64))
    want = o.step_dt(st64, dt)
    e.step_explicit(dt)
    got = e.download()
    fluid = o.mask == 0
    errs = rel_err(got, want, fluid)
    print("tauh2 parity", (W, H, warm), ["%.2e" % x for x in errs])
    assert max(errs) <= TOL, errs
    for g, s in zip(got, state):     # masked cells copy through
        assert np.array_equal(g[~fluid], s[~fluid])
    e.close()


def test_device_dt_matches_oracle(eng, oracle_built):
    """the on-device CFL/diffusion dt (no host round trip) equals the reference's host formula"""
    o = oracle_built.OracleH2(512, 256)
    e = eng.Hypersonic2D(512, 256)
    e.init()
    e.step(30)
    st64 = [a.astype(np.float64) for a in e.download()]
    o.apply_inflow(st64)
    maxs = o.max_wavespeed(st64)
    t0 = e.time()
    e.step(1)
    t1 = e.time()
    assert t1["dt"] == pytest.approx(o.dt_from_maxs(maxs), rel=2e-6)
    assert t1["t"] - t0["t"] == pytest.approx(t1["dt"], rel=1e-6)


def test_trajectory_matches_reference_checkvalues(eng):
    g = GOLD["tau2d_cuda_512x256_4steps_tile32x4"]
    e = eng.Hypersonic2D(512, 256)
    e.init()
    t = e.step(4)
    (rho, mx, my, E), mask = e.download(with_mask=True)
    fl = mask == 0
    assert int(fl.sum()) == g["fluid"]
    assert t == pytest.approx(g["t"], rel=1e-6)
    assert rho[fl].sum(dtype=np.float64) == pytest.approx(g["sum_rho"], rel=1e-6)
    assert mx[fl].sum(dtype=np.float64) == pytest.approx(g["sum_mx"], rel=1e-6)
    assert E[fl].sum(dtype=np.float64) == pytest.approx(g["sum_E"], rel=1e-6)
    e.close()


def test_full_size_band_vs_oracle(eng, oracle_built):
    """BASELINE size 4096^2: one step on the GPU; a 40-row band through the body and its bow shock is
    recomputed by the fp64 oracle (rows >= 4 away from the band edge are exact: stencil radius 2)."""
    n, band, pad = 4096, 40, 6
    e = eng.Hypersonic2D(n, n)
    e.init()
    e.step(150)
    state, mask = e.download(with_mask=True)
    j0 = n // 2 - 690     # the rounded shoulder of the body: v != 0, shock + wall + diffusion all in the band
    rows = slice(j0 - pad, j0 + band + pad)
    o = oracle_built.OracleH2(n, band + 2 * pad)
    o.mask[:] = mask[rows]
    st64 = [np.ascontiguousarray(a[rows], np.float64) for a in state]
    o.apply_inflow(st64)
    full64 = [a.astype(np.float64) for a in state]
    dt = 0.9 * o.dt_from_maxs(o.max_wavespeed(st64))
    want = o.step_dt(st64, dt)
    e.step_explicit(dt)
    got = e.download()
    fluid = (mask[rows] == 0)[pad:-pad]
    assert fluid.sum() < fluid.size
    errs = rel_err([g[rows][pad:-pad] for g in got], [w[pad:-pad] for w in want], fluid)
    print("tauh2 4096^2 band parity", ["%.2e" % x for x in errs])
    assert max(errs) <= TOL, errs
    del full64
    e.close()


def test_deterministic(eng):
    outs = []
    for _ in range(2):
        e = eng.Hypersonic2D(512, 256)
        e.init()
        t = e.step(50)
        outs.append((e.download(), t))
        e.close()
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    assert outs[0][1] == outs[1][1]


@pytest.mark.parametrize("W,H,steps", [(300, 200, 12), (512, 256, 20), (1000, 130, 8), (70, 50, 6)])
def test_marching_step_agrees_with_the_tile_step(eng, tmp_path, W, H, steps):
    """from ~2 M cells on the step runs as a march (one wave per 60-column strip, a five-row window in registers,
    nothing through LDS); TAU_H2_MARCH=2 forces it at these sizes, TAU_H2_MARCH=0 keeps the tile kernel.  Same
    predictor, faces, diffusion and repairs on the same operands — the two differ in where multiply-adds are
    contracted.  Mask and time must agree exactly, the fields to rounding after a few steps (the oracle parity
    tests above run the tile kernel here; the fuzz sweep runs the march against the oracle when forced)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import fluid_sims_amd as f, numpy as np\n"
            "h = f.Hypersonic2D(%d, %d); h.init(); t = h.step(%d)\n"
            "st, m = h.download(with_mask=True); np.savez(sys.argv[1], *st, mask=m, t=np.float64(t if t is not None else 0.0))\n"
            % (root, W, H, steps))
    outs = []
    for march in ("2", "0"):
        out = tmp_path / f"m{march}.npz"
        r = subprocess.run([sys.executable, "-c", code, str(out)], capture_output=True, text=True, env=dict(os.environ, TAU_H2_MARCH=march))
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(np.load(out))
    a, b = outs
    assert np.array_equal(a["mask"], b["mask"])
    assert abs(float(a["t"]) - float(b["t"])) <= 1e-6 * abs(float(b["t"])) + 1e-12
    for k in ("arr_0", "arr_1", "arr_2", "arr_3"):
        assert np.isfinite(a[k]).all()
        scale = max(float(np.abs(b[k]).max()), 1e-30)
        assert float(np.abs(a[k].astype(np.float64) - b[k]).max()) <= 2e-5 * scale, k


def test_march_is_the_same_for_every_workgroup_size(eng, tmp_path):
    """the waves of the marching kernel share nothing, so its workgroup size (TAU_H2_WPB = 1 / 2 / 4 waves, a dispatch
    granularity knob) must not change a bit; 2100 x 1100 runs the march by default, with chunks whose first rows are far
    from row 0 (the loads / stores are 32-bit offsets from each chunk's first row)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import fluid_sims_amd as f, numpy as np\n"
            "h = f.Hypersonic2D(2100, 1100); h.init(); t = h.step(25)\n"
            "st = h.download(); np.savez(sys.argv[1], *st, t=np.float64(t if t is not None else 0.0))\n" % root)
    outs = []
    for wpb in ("1", "2", "4"):
        out = tmp_path / f"w{wpb}.npz"
        r = subprocess.run([sys.executable, "-c", code, str(out)], capture_output=True, text=True, env=dict(os.environ, TAU_H2_WPB=wpb))
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(np.load(out))
    for o in outs[1:]:
        assert float(o["t"]) == float(outs[0]["t"])
        for k in ("arr_0", "arr_1", "arr_2", "arr_3"):
            assert np.array_equal(o[k], outs[0][k]), k


def test_neighbor_lookups_on_the_reference_field(eng):
    """The nine known answers of the reference's neighbour tests (tau_hypersonic_cuda_tests.cu:348-371, 567-640: inflow at
    x < 0, fluid neighbour, NO-SLIP reflection mx 3 -> -3 at a body cell, clamped y) evaluated by the engine's own staging
    rule (h2d::march_load + ghost_sel — what both step kernels stage their tiles / rows with), on the hand-built field."""
    nf = GOLD["unit_known_answers_tau_hypersonic_cuda_tests"]["neighbors_field"]
    W, H = 64, 32
    x, y = nf["x"], nf["y"]
    rho = np.full((H, W), nf["rest"]["rho"], np.float32)
    mx = np.zeros((H, W), np.float32)
    my = np.zeros((H, W), np.float32)
    E = np.full((H, W), nf["rest"]["p"] / (GAMMA - 1.0), np.float32)
    mask = np.zeros((H, W), np.uint8)
    mx[y, x] = nf["mx_center"]
    mx[y, x + 1] = nf["mx_right"]
    mask[y + 1, x] = nf["mask_up"]
    e = eng.Hypersonic2D(W, H)
    e.upload([rho, mx, my, E], mask)
    got = e.unit_neighbors(x, y)
    infl_mx = 25.0 * np.sqrt(np.float32(GAMMA))          # default_config: Mach 25 (tau_hypersonic_cuda.cu:1394-1409)
    want = [1.0, infl_mx, 1.0, 7.0, -3.0, 1.0, infl_mx, -3.0, 1.0]
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)   # fp32 engine: the reference's 1e-12 at double becomes 1 ulp
    e.close()


@pytest.mark.parametrize("W,H,steps", [(2100, 1100, 30), (4096, 2048, 60), (1500, 1500, 200)])
def test_uniform_row_exits_change_no_bit(eng, tmp_path, W, H, steps):
    """k_march_lds skips the predictors and the faces of a trip whose five-row window holds one state in all 64 lanes (the free stream
    ahead of and beside the bow shock: almost every trip of the BASELINE input's first hundreds of steps) and recomputes what the
    skipped trips would have handed on when the stretch ends.  TAUH2_UNIFORM_EXITS=0 evaluates everything: every field of every cell
    and the clock must be byte-identical, on grids that run the march by default, early (mostly free stream) and late (shock layer
    grown) in a run."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import fluid_sims_amd as f, numpy as np\n"
            "h = f.Hypersonic2D(%d, %d); h.init(); t = h.step(%d)\n"
            "st = h.download(); np.savez(sys.argv[1], *st, t=np.float64(t if t is not None else 0.0))\n" % (root, W, H, steps))
    outs = []
    for ex in ("1", "0"):
        out = tmp_path / f"x{ex}.npz"
        r = subprocess.run([sys.executable, "-c", code, str(out)], capture_output=True, text=True, env=dict(os.environ, TAUH2_UNIFORM_EXITS=ex))
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(np.load(out))
    a, b = outs
    assert float(a["t"]) == float(b["t"])
    for k in ("arr_0", "arr_1", "arr_2", "arr_3"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert float(np.abs(a["arr_1"] - a["arr_1"][0, -1]).max()) > 1.0      # a shock layer exists: not two runs of pure free stream
