"""Random-shape parity sweep of every kernel against its oracle (GPU box).

  python scripts/fuzz_parity.py [seed] [seconds]

Small random grids / particle counts, ragged in every direction; one step from a short warm-up (or from random
populations), compared with the same tolerances as tests/.  This sweep found the FMA-contracted signed distance
that flipped cells lying exactly on the sphere (see h3d::sdf_solid); tests/test_gpu_fuzz.py runs a short fixed-seed
slice of it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_sims_amd as f
from oracle import pyoracle
from tests.parity import assert_parity


def sweep(seed=0, seconds=120.0, max_iter=None, log=print):
  rng = np.random.default_rng(seed)
  bad = 0
  done = [0] * 9                  # comparisons actually made, per kind (a non-finite warm-up state is skipped)
  t_end = time.time() + seconds
  it = 0
  while time.time() < t_end and (max_iter is None or it < max_iter):
    it += 1
    kind = it % 9
    try:
        if kind == 0:   # 3D
            nx, ny, nz = [int(rng.integers(8, 72)) for _ in range(3)]
            e = f.Tau3D(nx, ny, nz); o = pyoracle.Oracle3D(nx, ny, nz)
            e.init(1); e.set_clock(0.02, 1e-4); e.step(int(rng.integers(0, 12)))
            st = e.download()
            if not all(np.isfinite(a).all() and np.abs(a).max() < 30 for a in st): e.close(); continue
            c = e.clock(); dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau))
            s = o.from_interior(st); o.fill_halo_periodic(s); out = o.new_state(); o.step_range(s, out, dt, 1.0)
            e.step_explicit(dt, 1.0); got = e.download()
            assert_parity(got, o.interior(out), mask=o.interior([o.solid])[0] == 0, what=f"3D {nx,ny,nz}")
            e.close()
        elif kind == 1:  # Gray-Scott + laplacians, any nx
            nx, ny = int(rng.integers(2, 300)), int(rng.integers(2, 120))
            o2 = pyoracle.Oracle2D()
            u = rng.random((ny, nx)).astype(np.float32); v = rng.random((ny, nx)).astype(np.float32)
            g = f.GrayScott(nx, ny); g.upload(u, v); g.step(3); gu, gv = g.download(); g.close()
            p = o2.gs_params(nx, ny); wu, wv = u, v
            for _ in range(3): wu, wv = o2.gs_step(p, wu, wv)
            assert np.array_equal(gu, wu) and np.array_equal(gv, wv), f"GS {nx}x{ny}"
        elif kind == 2:  # 2D Euler
            W, H = int(rng.integers(8, 400)), int(rng.integers(8, 200))
            o = pyoracle.OracleH2(W, H); o.init(); e = f.Hypersonic2D(W, H); e.init(); e.step(int(rng.integers(0, 30)))
            st = [a.astype(np.float64) for a in e.download()]
            if not all(np.isfinite(a).all() for a in st): e.close(); continue
            o.apply_inflow(st); dt = o.dt_from_maxs(o.max_wavespeed(st)); want = o.step_dt(st, dt); e.step_explicit(dt); got = e.download()
            fl = o.mask == 0
            r, mx, my, E = want
            rr = np.maximum(r, 1e-25); uu, vv = mx / rr, my / rr
            pp = 0.1 * np.maximum(E - 0.5 * rr * (uu * uu + vv * vv), 1e-25); a = np.sqrt(1.1 * pp / rr)
            mom = rr * (np.sqrt(uu * uu + vv * vv) + a)
            for gq, wq, sc in zip(got, want, [rr, mom, mom, np.abs(E)]):
                err = (np.abs(gq.astype(np.float64) - wq) / sc)[fl].max()
                assert err <= 1e-5, f"tauh2 {W}x{H} err {err:.2e}"
            e.close()
        elif kind == 3:  # LBM
            nx, ny = int(rng.integers(3, 600)), int(rng.integers(3, 80))
            fz = (0.05 + rng.random((9, ny, nx))).astype(np.float32); sol = (rng.random((ny, nx)) < 0.2).astype(np.uint8)
            o = pyoracle.OracleLbm(nx, ny, drive=1e-3); o.solid[:] = sol; e = f.Lbm2D(nx, ny, drive=1e-3); e.upload(fz, sol)
            want = o.step(fz, 2); e.step(2); got, _ = e.download(); e.close()
            assert np.array_equal(got, want), f"LBM {nx}x{ny}"
        elif kind == 5:  # full Burgers step, random size / MUSCL / 1D / substeps
            nx, ny = int(rng.integers(8, 300)), int(rng.integers(8, 120))
            kw = dict(dtau=1e-2, muscl=int(rng.integers(0, 2)), nu=float(rng.choice([0.0, 0.02, 0.1])), visc_substeps=int(rng.integers(1, 3)))
            e = f.Flow2D("burgers", nx, ny, **kw); o = pyoracle.OracleFlow("burgers", nx, ny, **kw)
            e.init(); e.step(int(rng.integers(0, 15))); fl = e.download()
            dt = o.dt_eff(fl, e.clock()["t"]); want = o.step(fl, dt); e.step_explicit(dt)
            # phi = asinh(u/u0): where a shock passes, u_out ~ 1 is what is left of fluxes of size u_in ~ 30, so the
            # 1e-5 is taken against the largest |u| feeding the cell (5 x 5 neighbourhood, periodic), not against u_out
            from scipy.ndimage import maximum_filter
            uin = np.maximum(np.abs(np.sinh(fl[0].astype(np.float64))), np.abs(np.sinh(fl[1].astype(np.float64))))
            scale = np.maximum(maximum_filter(uin, size=5, mode="wrap"), 1.0)
            for g, w in zip(e.download(), want):
                du = np.abs(np.sinh(g.astype(np.float64)) - np.sinh(w.astype(np.float64)))
                worst = (du / np.maximum(scale, np.abs(np.sinh(w.astype(np.float64))))).max()
                assert worst <= 1e-5, f"burgers {nx}x{ny} {kw} err {worst:.2e}"
            e.close()
        elif kind == 6:  # full shallow-water step
            nx, ny = int(rng.integers(8, 300)), int(rng.integers(8, 120))
            kw = dict(dtau=1e-2, nu=float(rng.choice([0.0, 0.001, 0.05])))
            ini = dict(H0=10.0, amp=0.5, bsig=6.0, offx=5.0, offy=-3.0, asym=0.3, swirl=0.05, rc=20.0)
            e = f.Flow2D("sw", nx, ny, **kw, **ini); o = pyoracle.OracleFlow("sw", nx, ny, **kw)
            e.init(); e.step(int(rng.integers(0, 15))); fl = e.download()
            dt = o.dt_eff(fl, e.clock()["t"]); want = o.step(fl, dt); e.step_explicit(dt); got = e.download()
            c = np.sqrt(9.81 * 10.0)
            assert np.abs(got[0] - want[0]).max() <= 1e-5, f"sw {nx}x{ny} sigma"
            assert max(np.abs(got[1].astype(np.float64) - want[1]).max(), np.abs(got[2].astype(np.float64) - want[2]).max()) <= 1e-5 * c, f"sw {nx}x{ny} u,v"
            e.close()
        elif kind == 7:  # 3D visualisation fields on a random shape
            nx, ny, nz = [int(rng.integers(8, 60)) for _ in range(3)]
            e = f.Tau3D(nx, ny, nz); o = pyoracle.Oracle3D(nx, ny, nz)
            e.init(1); e.set_clock(0.02, 1e-4); e.step(int(rng.integers(0, 10)))
            st = e.download()
            if not all(np.isfinite(a).all() and np.abs(a).max() < 30 for a in st): e.close(); continue
            s3 = o.from_interior(st); o.fill_halo_periodic(s3)
            fluid = o.interior([o.solid])[0] == 0
            mode = int(rng.integers(0, 8))
            got = e.vis(mode); want, scale = o.vis(s3, mode)
            err = (np.abs(got.astype(np.float64) - want)[fluid] / np.maximum(scale[fluid].astype(np.float64), 1e-30)).max()
            assert err <= 2e-6 and (got[~fluid] == 0).all(), f"vis mode {mode} {nx,ny,nz} err {err:.2e}"
            e.close()
        elif kind == 8:  # SPH from clustered random positions (overflow / LDS-bypass paths)
            N = int(rng.integers(50, 1500))
            side = float(rng.uniform(0.05, 0.9))
            pos = (0.5 - side / 2 + side * rng.random((N, 2))).astype(np.float32)
            vel = (0.05 * rng.standard_normal((N, 2))).astype(np.float32)
            o = pyoracle.OracleSph(N); e = f.Sph2D(N); e.upload(pos, vel); o.set_state(pos, vel)
            o.substep(1e-4); e.substep(1e-4)
            g, w = e.download(), o.state()
            assert np.array_equal(g["cell"], w["cell"]), f"SPH cluster {N} cells"
            tol = max(1e-5, N * 2.0 ** -24)
            rw, rg = np.exp(w["s"].astype(np.float64)), np.exp(g["s"].astype(np.float64))
            assert (np.abs(rg - rw) / rw).max() <= tol, f"SPH cluster {N} side {side:.2f} rho {(np.abs(rg - rw) / rw).max():.2e}"
            sc = np.maximum(w["acc_abs"].astype(np.float64), 1e-30)
            assert (np.linalg.norm(g["acc"].astype(np.float64) - w["acc"], axis=1) / sc).max() <= tol, f"SPH cluster {N} acc"
            e.close()
        else:           # SPH
            N = int(rng.integers(1, 9000))
            o = pyoracle.OracleSph(N); e = f.Sph2D(N); e.reset_particles(); e.step(int(rng.integers(0, 4)))
            st = e.download(); o.set_state(st["pos"], st["vel"]); dt = e.dt(); o.substep(dt); e.substep(dt)
            g, w = e.download(), o.state()
            assert np.array_equal(g["cell"], w["cell"]), f"SPH {N} cells"
            rw, rg = np.exp(w["s"].astype(np.float64)), np.exp(g["s"].astype(np.float64))
            nb = 400 * 2.0 ** -24
            assert (np.abs(rg - rw) / rw).max() <= max(1e-5, nb), f"SPH {N} rho {(np.abs(rg - rw) / rw).max():.2e}"
            sc = np.maximum(w["acc_abs"].astype(np.float64), 1e-30)
            assert (np.linalg.norm(g["acc"].astype(np.float64) - w["acc"], axis=1) / sc).max() <= max(1e-5, nb), f"SPH {N} acc"
            e.close()
        done[kind] += 1
    except AssertionError as ex:
        bad += 1; log("FAIL " + str(ex)[:300])
    except Exception as ex:
        bad += 1; log("ERROR %d %s %s" % (kind, type(ex).__name__, str(ex)[:300]))
  log("iterations %d failures %d compared per kind %s" % (it, bad, done))
  return it, bad


if __name__ == "__main__":
    sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 120.0)
