"""Random-shape sweep of the 3D step against THE REFERENCE'S OWN k_step on the GPU (oracle/_ref/th3cs.co): ragged planes up to
~300^2 (partial tiles, fewer planes than a chunk), fused or split step, the fast or the FORCED reciprocal WENO weights, a random
number of warm-up steps from the impulsive or the reference start.  The reference kernel runs on the device, so a minute covers
hundreds of shapes (the CPU oracle manages a dozen).

  python scripts/fuzz_ref3d.py [seed] [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import fluid_sims_amd as f  # noqa: E402
from oracle import refgpu  # noqa: E402
from tests.parity import assert_parity, undershoot_cells  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
t_end = time.time() + seconds
n = bad = skipped = flipped = degenerate = 0
kinds = {}
while time.time() < t_end:
    big = rng.random() < 0.3
    nx, ny, nz = (int(rng.integers(8, 301 if big else 120)), int(rng.integers(8, 301 if big else 120)), int(rng.integers(8, 49)))
    split, rcp, mode = bool(rng.integers(0, 2)), rng.random() < 0.35, int(rng.integers(0, 2))
    warm = int(rng.integers(0, 25))
    if rcp:
        os.environ["TAU3D_WENO_RCP"] = "1"
    try:
        e = f.Tau3D(nx, ny, nz)
    finally:
        os.environ.pop("TAU3D_WENO_RCP", None)
    try:
        e.set_split(split)
        e.init(mode)
        if mode:
            e.set_clock(0.02, 1e-4)
        e.step(warm)
        st = e.download()
        if not all(np.isfinite(a).all() and np.abs(a).max() < 30 for a in st):
            skipped += 1
            continue
        c = e.clock()
        dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau)) if warm else 2e-6
        gain = 1.0 if mode else float(min(max(c.t / 0.02, 0.0), 1.0))
        # the reference's Makefile build contracts a*b+c in the signed distance and flips the mask of cells whose centre lies ON the
        # sphere (tests/test_gpu_ref3d.py::test_mask_and_init...): such shapes are compared with the build without contraction,
        # whose mask is the engine's bit for bit — a different mask is a different problem, not a rounding difference
        r = refgpu.Ref3D(nx, ny, nz)
        if not np.array_equal(r.solid_mask(), e.solid()):
            r.close()
            r = refgpu.Ref3D(nx, ny, nz, ieee=True)
            assert np.array_equal(r.solid_mask(), e.solid()), "mask differs from the IEEE build of k_build_solid_mask"
            flipped += 1
        r.upload(st)
        m_ref = r.step(dt, gain)
        m = e.step_explicit(dt, gain)
        assert e.field_range()[2] == (not rcp)
        # cells next to a face whose WENO density / pressure undershoots below zero carry no parity information (tests/parity.py:
        # the reference's own result is an accident of rounding there); they appear on thin anisotropic grids a step or two
        # before the state leaves the sane range
        us = undershoot_cells(st, r.solid_mask())
        want = r.download()
        if not all(np.isfinite(a[r.solid_mask() == 0]).all() for a in want):
            # the REFERENCE's own step returns inf / nan from this input (a thin anisotropic grid a step before its blow-up: encoded
            # values below 30 are still velocities of 1e7): nothing to compare with (round 5: seed 11, 224 x 88 x 46 after 23 steps —
            # four cells non-finite in the reference kernel's output and in the engine's alike)
            r.close()
            skipped += 1
            continue
        degenerate += int(us.any())
        assert_parity(e.download(), want, mask=(r.solid_mask() == 0) & ~us, what=f"{(nx, ny, nz)} split={split} rcp={rcp} mode={mode} warm={warm}")
        assert us.any() or abs(m - m_ref) <= 1e-5 * max(m_ref, 1e-30), f"max wavespeed {m} vs {m_ref}"
        r.close()
        n += 1
        k = ("split" if split else "fused") + ("/rcp" if rcp else "/fast")
        kinds[k] = kinds.get(k, 0) + 1
    except AssertionError as ex:
        bad += 1
        print("FAIL", (nx, ny, nz), split, rcp, mode, warm, str(ex)[:300], flush=True)
    finally:
        e.close()
print(f"fuzz_ref3d seed {seed}: {n} shapes compared with the reference kernel ({kinds}), {skipped} skipped (state left the sane range), {flipped} on the IEEE build (mask cell on the sphere), {degenerate} with undershoot cells excluded, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
