"""GPU: fused Burgers / shallow-water steps (tauflow_*, through the C-ABI) against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def field_err(got, want, floor=0.0):
    g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(g - w).max() / max(np.abs(w).max(), floor, 1e-30))


@pytest.mark.parametrize("nx,ny,muscl,warm", [(256, 128, 0, 0), (256, 128, 1, 40), (100, 60, 1, 25), (512, 512, 0, 60),
                                              (67, 33, 1, 10)])
def test_burgers_step_parity(eng, oracle_built, nx, ny, muscl, warm):
    kw = dict(dtau=1e-2, muscl=muscl, nu=0.1)
    e = eng.Flow2D("burgers", nx, ny, **kw)
    o = oracle_built.OracleFlow("burgers", nx, ny, **kw)
    e.init()
    f0 = e.download()
    for a, b in zip(f0, o.init_burgers()):
        assert np.array_equal(a, b), "initial field must be bit-exact (host libm on both sides)"
    if warm:
        e.step(warm)
    f = e.download()
    dt = o.dt_eff(f, e.clock()["t"])
    want = o.step(f, dt)
    e.step_explicit(dt)
    got = e.download()
    u0 = 1.0
    for g, w in zip(got, want):   # decoded velocity against the field scale, and the encoded array itself
        assert field_err(u0 * np.sinh(g.astype(np.float64)), u0 * np.sinh(w.astype(np.float64))) <= TOL
        assert np.abs(g - w).max() <= TOL
    e.close()


def test_burgers_device_dt_and_clock(eng, oracle_built):
    kw = dict(dtau=1e-2, muscl=1)
    e = eng.Flow2D("burgers", 256, 128, **kw)
    o = oracle_built.OracleFlow("burgers", 256, 128, **kw)
    e.init()
    e.step(20)
    f = e.download()
    c0 = e.clock()
    want_dt = o.dt_eff(f, c0["t"])
    e.step(1)
    c1 = e.clock()
    assert c1["dt"] == pytest.approx(want_dt, rel=2e-6)
    assert c1["t"] == pytest.approx(c0["t"] * np.exp(np.float32(1e-2)), rel=1e-6) and c1["step"] == 21


def test_burgers_colehopf_harness(eng):
    """the reference's accuracy harness (--colehopf): relative L2 error against the exact solution"""
    e = eng.Flow2D("burgers", 512, 1, oneD=1, dtau=1e-3, muscl=1, nu=0.1)
    e.init()
    elapsed = 0.0
    for _ in range(1500):
        e.step(1)
        elapsed += e.clock()["dt"]
    assert e.colehopf_relL2(elapsed) < 6e-5
    e.close()


def test_burgers_visc_substeps(eng, oracle_built):
    kw = dict(dtau=1e-2, muscl=1, visc_substeps=3)
    e = eng.Flow2D("burgers", 256, 64, **kw)
    o = oracle_built.OracleFlow("burgers", 256, 64, **kw)
    e.init()
    e.step(10)
    f = e.download()
    dt = o.dt_eff(f, e.clock()["t"])
    want = o.step(f, dt)
    e.step_explicit(dt)
    for g, w in zip(e.download(), want):
        assert np.abs(g - w).max() <= TOL
    e.step(3)   # the metric of the final state feeds the next dt
    assert np.isfinite(e.clock()["dt"]) and e.clock()["dt"] > 0
    e.close()


@pytest.mark.parametrize("nx,ny,nu,warm", [(256, 128, 0.001, 0), (256, 128, 0.05, 40), (100, 60, 0.0, 25), (512, 512, 0.001, 60)])
def test_shallow_water_step_parity(eng, oracle_built, nx, ny, nu, warm):
    kw = dict(dtau=1e-2, nu=nu)
    ini = dict(H0=10.0, amp=0.5, bsig=6.0, offx=5.0, offy=-3.0, asym=0.3, swirl=0.05, rc=20.0)
    e = eng.Flow2D("sw", nx, ny, **kw, **ini)
    o = oracle_built.OracleFlow("sw", nx, ny, **kw)
    e.init()
    f0 = e.download()
    for a, b in zip(f0, o.init_sw(H0=10.0, bumpAmp=0.5, bumpSigma=6.0, offx=5.0, offy=-3.0, asym=0.3, swirl=0.05, swirlRc=20.0)):
        assert np.array_equal(a, b)
    if warm:
        e.step(warm)
    f = e.download()
    dt = o.dt_eff(f, e.clock()["t"])
    want = o.step(f, dt)
    e.step_explicit(dt)
    got = e.download()
    c = np.sqrt(9.81 * 10.0)   # gravity-wave speed: the velocity scale of the problem
    assert np.abs(got[0] - want[0]).max() <= TOL                     # sigma = ln h: relative 1e-5 in depth
    assert np.abs(got[1].astype(np.float64) - want[1]).max() <= TOL * c
    assert np.abs(got[2].astype(np.float64) - want[2]).max() <= TOL * c
    e.close()


def test_shallow_water_mass_and_rest(eng):
    e = eng.Flow2D("sw", 512, 256, dtau=1e-2, H0=10.0, amp=0.5, bsig=6.0, offx=5.0, offy=-3.0, asym=0.3, swirl=0.05, rc=20.0)
    e.init()
    m0 = np.exp(e.download()[0].astype(np.float64)).sum()
    e.step(100)
    assert np.exp(e.download()[0].astype(np.float64)).sum() == pytest.approx(m0, rel=5e-6)
    rest = [np.full((256, 512), np.log(np.float32(7.0)), np.float32), np.zeros((256, 512), np.float32), np.zeros((256, 512), np.float32)]
    e.upload(rest)
    e.step(5)
    out = e.download()
    assert np.abs(out[1]).max() == 0 and np.abs(out[2]).max() == 0
    np.testing.assert_allclose(out[0], rest[0], atol=2e-6)
    e.close()


def test_full_size_translation_invariance(eng):
    """4096^2: both programs commute with periodic shifts to rounding —
    every tile seam and both wrap-arounds at full size."""
    n = 4096
    rng = np.random.default_rng(11)
    for kind, nf in (("burgers", 2), ("sw", 3)):
        e = eng.Flow2D(kind, n, n, dtau=1e-2, muscl=1)
        f = [(0.3 * rng.standard_normal((n, n))).astype(np.float32) for _ in range(nf)]
        if kind == "sw":
            f[0] = np.log(10.0 + f[0]).astype(np.float32)
        e.upload(f)
        e.step_explicit(0.01)
        a = e.download()
        sh = (123, 1027)
        e.upload([np.roll(x, sh, (0, 1)) for x in f])
        e.step_explicit(0.01)
        b = e.download()
        for x, y in zip(a, b):
            assert np.array_equal(np.roll(x, sh, (0, 1)), y)
        e.close()


@pytest.mark.parametrize("kind,nx,ny,kw", [("burgers", 256, 128, {}), ("burgers", 100, 61, {}), ("burgers", 512, 16, dict(oneD=1)),
                                           ("sw", 256, 128, {}), ("sw", 97, 50, dict(nu=0.0)), ("sw", 1000, 300, dict(nu=0.05)),
                                           ("burgers", 300, 200, dict(visc_substeps=3)), ("burgers", 256, 128, dict(muscl=1)),
                                           ("burgers", 131, 77, dict(muscl=1)), ("burgers", 512, 16, dict(muscl=1, oneD=1))])
def test_marching_step_agrees_with_the_tile_step(eng, tmp_path, kind, nx, ny, kw):
    """plain Burgers and shallow water take the marching kernel (one wave per 60-column strip, everything in
    registers) from ~2 M cells on — TAU_FLOW_MARCH=2 forces it here, TAU_FLOW_MARCH=0 keeps the LDS-tile kernel.  Same faces, same update formulas — the two differ only
    in where the compiler contracts multiply-adds, so 25 steps apart they agree to ~1e-6 (the parity tests above run
    the marching kernel against the oracle)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import fluid_sims_amd as f, numpy as np\n"
            "e = f.Flow2D(%r, %d, %d, **%r); e.init(); e.step(25)\n"
            "c = e.clock(); np.savez(sys.argv[1], *e.download(), clock=np.array([c['t'], c['dt'], c['wavespeed'], c['step']]))\n"
            % (root, kind, nx, ny, kw))
    outs = []
    for march in ("1", "0"):
        out = tmp_path / f"m{march}.npz"
        r = subprocess.run([sys.executable, "-c", code, str(out)], capture_output=True, text=True, env=dict(os.environ, TAU_FLOW_MARCH={"1": "2", "0": "0"}[march]))
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(np.load(out))
    a, b = outs
    assert a["clock"][3] == b["clock"][3] == 25 and a["clock"][0] == b["clock"][0]
    assert abs(a["clock"][1] - b["clock"][1]) <= 1e-5 * abs(b["clock"][1])
    for k in a.files:
        if k.startswith("arr_"):
            assert np.isfinite(a[k]).all()
            scale = max(float(np.abs(b[k]).max()), 1e-30)
            assert float(np.abs(a[k].astype(np.float64) - b[k]).max()) <= 1e-4 * scale, k


def test_device_timer_brackets_the_launches(eng):
    """tauflow_timer_*: the "GPU" line of the reference's headless summaries (cudaEvent pairs, tau_burgers.cu:790-820) — events on
    the handle's stream around the launches; the program's own summary prints it beside the wall clock"""
    import os
    import re
    import subprocess
    import time
    f = eng.Flow2D("burgers", 1024, 1024, dtau=0.01)
    f.init()
    f.step_async(5)
    f.sync()
    t0 = time.perf_counter()
    f.timer_start()
    f.step_async(50)
    ms = f.timer_stop()
    wall = (time.perf_counter() - t0) * 1e3
    f.close()
    assert 0.0 < ms <= wall * 1.05, (ms, wall)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "bin", "tau_burgers"), "--headless", "--nx", "512", "--ny", "512", "--steps", "200"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    w = re.search(r"Wall:\s+(\d+) frames in ([0-9.]+) s", r.stdout)
    g = re.search(r"GPU:\s+(\d+) frames in ([0-9.]+) s", r.stdout)
    assert w and g and w.group(1) == g.group(1), r.stdout
    assert 0.0 < float(g.group(2)) <= float(w.group(2)) + 1e-3, r.stdout
