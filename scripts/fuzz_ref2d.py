"""Random sweeps of the 2D simulators against THE REFERENCE'S OWN KERNELS on the GPU (oracle/_ref/*.co): Gray-Scott and LBM on random
ragged grids with random parameters (bit-exact against the builds without contraction), SPH with random particle counts and
parameters (cell indices bit-exact, fields at the tolerances of tests/test_gpu_sph.py), 2D Euler at the size the reference fixes
(8192 x 1024) with random SimConfig values and warm-up lengths.   python scripts/fuzz_ref2d.py [seed] [seconds] [which]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import fluid_sims_amd as f  # noqa: E402
from oracle import refgpu, pyoracle  # noqa: E402
from tests.test_gpu_sph import compare_substep  # noqa: E402
from tests.test_gpu_tauh2 import rel_err  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["gs", "lbm", "sph", "h2"]
rng = np.random.default_rng(seed)
count, bad = {}, 0


def gs():
    nx, ny = int(rng.integers(2, 700)), int(rng.integers(2, 300))
    if rng.random() < 0.5:
        nx = 4 * max(1, nx // 4)
    kw = dict(dx=float(rng.choice([1.0, 0.5, 0.7, 2.0])), Du=float(rng.uniform(0.05, 0.2)), Dv=float(rng.uniform(0.02, 0.1)),
              feed=float(rng.uniform(0.02, 0.06)), kill=float(rng.uniform(0.05, 0.07)))
    kw["dt"] = 0.2 * kw["dx"] ** 2 / max(kw["Du"], kw["Dv"]) * float(rng.uniform(0.3, 1.0))
    steps = int(rng.integers(1, 13))
    u0 = rng.random((ny, nx), dtype=np.float32)
    v0 = (0.4 * rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    g = f.GrayScott(nx, ny, **kw)
    g.upload(u0, v0)
    r = refgpu.RefGrayScott(nx, ny, ieee=True, **kw)
    r.upload(u0, v0)
    g.step(steps)
    r.step(steps)
    a, b = g.download(), r.download()
    g.close(); r.close()
    # (a random dt may be unstable: NaN == NaN counts as equal, the bits of everything else must match)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True), \
        f"gray-scott {nx}x{ny} {kw} {steps} steps: not bit-exact ({int((a[0] != b[0]).sum())} cells, finite {bool(np.isfinite(b[0]).all())})"


def lbm():
    nx, ny = int(rng.integers(8, 600)), int(rng.integers(4, 300))
    kw = dict(tau=float(rng.uniform(0.52, 1.5)), drive=float(rng.choice([0.0, 1e-6, 1e-4, 3e-3])), obstacle_radius=float(rng.uniform(1, max(2, ny / 3))))
    obstacle = bool(rng.integers(0, 2))
    r = refgpu.RefLbm(nx, ny, ieee=True, obstacle=obstacle, **kw)
    e = f.Lbm2D(nx, ny, obstacle=int(obstacle), **kw)
    r.init()
    f0, solid = r.download()
    if rng.random() < 0.5:       # random populations near equilibrium and a random extra mask
        w = np.float32([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)[:, None, None]
        f0 = (w * (1.0 + 0.2 * rng.random((9, ny, nx)))).astype(np.float32)
        solid = (solid | (rng.random((ny, nx)) < 0.05)).astype(np.uint8)
        r.upload(f0, solid)
    e.upload(f0, solid)
    n = int(rng.integers(1, 12))
    r.step(n)
    e.step(n)
    a, b = e.download()[0], r.download()[0]
    e.close(); r.close()
    assert np.array_equal(a, b), f"lbm {nx}x{ny} {kw} obstacle={obstacle} {n} steps: not bit-exact"


def sph():
    N = int(rng.choice([1024, 4096, 5000, 16384, 30000, 65536, 200000]))
    kw = dict(viscAlpha=float(rng.choice([0.0, 0.1, 0.5])), useGrav=int(rng.integers(0, 2)), hMul=float(rng.choice([1.5, 2.0, 2.5])))
    if kw["viscAlpha"] == 0.0:
        kw["useVisc"] = 0
    warm = int(rng.integers(0, 40))
    e = f.Sph2D(N, **kw)
    e.reset_particles()
    e.step(warm)
    st = e.download()
    dt = e.dt()
    r = refgpu.RefSph(N, ieee=True, **{k: getattr(e.params, k) for k in "boxX boxY rho0 c0 gammaEOS hMul viscAlpha gravity useVisc useGrav".split()})
    r.upload(st["pos"], st["vel"])
    head, nxt = r.substep(dt)
    w = r.state()
    w["cell"] = r.cells_from_lists(head, nxt)
    r.close()
    o = pyoracle.OracleSph(N, **kw)
    o.set_state(st["pos"], st["vel"])
    o.substep(dt)
    w["acc_abs"] = o.state()["acc_abs"]
    e.substep(dt)
    got = e.download()
    e_pairs = e.count_pairs()
    g_ = e.grid()
    e.close()
    try:
        compare_substep(got, w, what=f"N={N} warm={warm} {kw}", dt=dt)
    except AssertionError as ex:
        # An fp32 sum of n terms taken in arbitrary order is only defined to ~n eps / 2: 1e-5 is n = 335.  With hMul = 2.5 and a
        # compressed state a particle has 500-750 neighbours, and measured against the fp64 sum of the same pairs the REFERENCE is
        # then 1.0e-5 off and the engine 2.6e-6 (scratch check of round 4; the engine adds cell by cell in ascending order).  Past
        # 150 neighbours per particle on average the density is therefore compared with the exact sum instead: the engine must be at least as
        # close to it as the reference is, and within 1e-5.
        nb = e_pairs / N
        if nb <= 150:   # (the MEAN count: the worst particles of a compressed state hold twice as many)
            raise AssertionError(f"N={N} warm={warm} {kw} ({nb:.0f} neighbours per particle): {ex}")
        pos = st["pos"].astype(np.float64)
        h, m = float(np.float32(g_["h"])), float(np.float32(g_["mass"]))
        rg, rr = np.exp(got["s"].astype(np.float64)), np.exp(w["s"].astype(np.float64))
        worst = np.argsort(np.abs(rg / rr - 1))[-256:]            # the particles where engine and reference disagree most
        d = pos[worst, None, :] - pos[None, :, :]
        q = np.sqrt((d ** 2).sum(-1)) / h
        exact = np.where(q < 1, 1 - 1.5 * q ** 2 + 0.75 * q ** 3, np.where(q < 2, 0.25 * (2 - q) ** 3, 0.0)).sum(1) * (m * 10.0 / (7.0 * np.pi * h * h))
        eg = float(np.abs(rg[worst] / exact - 1).max())
        er = float(np.abs(rr[worst] / exact - 1).max())
        if not (eg <= 1e-5 and eg <= er):
            raise AssertionError(f"N={N} warm={warm} {kw}: against the fp64 sum the engine is {eg:.2e} off, the reference {er:.2e}: {ex}")
        LONG.append((nb, eg, er))


LONG = []
H2 = {}


def h2():
    if "r" not in H2:
        H2["r"] = refgpu.RefH2()
    r = H2["r"]
    cfg = dict(gamma=float(rng.choice([1.1, 1.2, 1.4])), mach=float(rng.choice([5.0, 15.0, 25.0])), visc_nu=float(rng.choice([0.0, 0.02, 0.05])),
               visc_rho=float(rng.choice([0.0, 0.05])), visc_e=float(rng.choice([0.0, 0.02])), geom_theta=float(rng.uniform(0.3, 1.0)),
               geom_rb=float(rng.uniform(40, 120)), geom_rn=float(rng.uniform(10, 39)))
    c = r.cfg
    c.gamma, c.inflow_mach, c.visc_nu, c.visc_rho, c.visc_e = cfg["gamma"], cfg["mach"], cfg["visc_nu"], cfg["visc_rho"], cfg["visc_e"]
    c.geom_theta, c.geom_Rb, c.geom_Rn = cfg["geom_theta"], cfg["geom_rb"], cfg["geom_rn"]
    r.m.set_global("d_cfg", c)
    r.init()
    e = f.Hypersonic2D(r.W, r.H, **cfg)
    e.init()
    got, mask = e.download(with_mask=True)
    assert np.array_equal(mask, r.mask_host()), f"2D Euler mask {cfg}"
    warm = int(rng.integers(0, 30))
    e.step(warm)
    state = e.download()
    if not all(np.isfinite(a).all() for a in state):
        e.close()
        return
    r.upload([a.astype(np.float64) for a in state])
    dt, _ = r.step(1)
    want = r.download()
    e2 = f.Hypersonic2D(r.W, r.H, **cfg)
    e2.upload(state, mask)
    e2.step_explicit(dt)
    errs = rel_err(e2.download(), want, mask == 0)
    e.close(); e2.close()
    assert max(errs) <= 1e-5, f"2D Euler {cfg} warm={warm}: {errs}"


fns = dict(gs=gs, lbm=lbm, sph=sph, h2=h2)
t_end = time.time() + seconds
while time.time() < t_end:
    k = which[int(rng.integers(0, len(which)))]
    try:
        fns[k]()
        count[k] = count.get(k, 0) + 1
    except AssertionError as ex:
        bad += 1
        print("FAIL", k, str(ex)[:400], flush=True)
if LONG:
    print(f"sph: {len(LONG)} cases with {min(x for x, _, _ in LONG):.0f}-{max(x for x, _, _ in LONG):.0f} neighbours per particle compared with the fp64 sum: "
          f"engine off by up to {max(x for _, x, _ in LONG):.2e}, reference by up to {max(x for _, _, x in LONG):.2e}")
print(f"fuzz_ref2d seed {seed}: {count} cases compared with the reference kernels, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
