"""One case of scripts/fuzz_ref3d.py under the microscope:  python scripts/fuzz_case3d.py nx ny nz split rcp mode warm
Prints the worst cell of lam / zet (conditioning-scaled, tests/parity.py) for the engine against the reference kernel, and — the
yardstick of tests/test_oracle_spread.py — for the reference's OWN two builds (Makefile flags vs IEEE) against each other, plus
the engine's other step forms (fused / split, fast / reciprocal weights) against the same reference output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import fluid_sims_amd as f  # noqa: E402
from oracle import refgpu  # noqa: E402
from tests.parity import kappa, kappa_zet, undershoot_cells, decode  # noqa: E402

nx, ny, nz, split, rcp, mode, warm = (int(a) for a in sys.argv[1:8])


def engine(split, rcp, st=None, dt=None, gain=None):
    if rcp:
        os.environ["TAU3D_WENO_RCP"] = "1"
    try:
        e = f.Tau3D(nx, ny, nz)
    finally:
        os.environ.pop("TAU3D_WENO_RCP", None)
    e.set_split(bool(split))
    if st is None:
        e.init(mode)
        if mode:
            e.set_clock(0.02, 1e-4)
        e.step(warm)
        st = e.download()
        c = e.clock()
        dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau)) if warm else 2e-6
        gain = 1.0 if mode else float(min(max(c.t / 0.02, 0.0), 1.0))
    else:
        e.init(mode)
        e.upload(st)
    e.step_explicit(dt, gain)
    out = e.download()
    sol = e.solid()
    e.close()
    return st, dt, gain, out, sol


st, dt, gain, got, sol = engine(split, rcp)
refs = {}
for ieee in (False, True):
    r = refgpu.Ref3D(nx, ny, nz, ieee=ieee)
    r.upload(st)
    r.step(dt, gain)
    refs[ieee] = r.download()
    mask_ok = np.array_equal(r.solid_mask(), sol)
    r.close()
    print("reference build", "IEEE" if ieee else "Makefile flags", "mask equal to the engine's:", mask_ok)
want = refs[False]
fluid = (sol == 0) & ~undershoot_cells(st, sol)
kaps = {"lam": kappa(want), "zet": kappa_zet(want)}     # (zet: kappa x theta_v / T where vibration is frozen out, tests/parity.py)


def worst(a, b, name):
    for fld, idx in (("lam", 4), ("zet", 5)):
        kap = kaps[fld]
        d = np.where(fluid, np.abs(np.asarray(a[idx], np.float64) - np.asarray(b[idx], np.float64)) / kap, 0.0)
        i = np.unravel_index(np.argmax(d), d.shape)
        prim = [float(q[i]) for q in decode(want)]
        print(f"{name:44s} {fld}/kappa max {d.max():.3e} at {i}  raw diff {abs(float(a[idx][i]) - float(b[idx][i])):.3e}  kappa {kap[i]:.3e}  "
              f"want (r,u,v,w,p,ev) = {['%.3e' % x for x in prim]}  cells > 1e-5: {int((d > 1e-5).sum())}")


worst(got, want, f"engine split={split} rcp={rcp} vs reference")
worst(refs[True], want, "reference IEEE build vs reference Makefile build")
for s2, r2 in ((0, 0), (0, 1), (1, 0), (1, 1)):
    if (s2, r2) != (split, rcp):
        try:
            _, _, _, g2, _ = engine(s2, r2, st, dt, gain)
            worst(g2, want, f"engine split={s2} rcp={r2} vs reference")
        except Exception as ex:
            print("engine", s2, r2, "failed:", ex)
