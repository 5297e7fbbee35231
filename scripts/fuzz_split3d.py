"""Random-shape parity sweep of the SPLIT 3D step (k_flux_xy + k_update_z) against the oracle: planes up to ~220^2, ragged in
every direction (partial tiles in x and y, fewer planes than a chunk), a random number of warm-up steps.

  TAU3D_SPLIT=1 python scripts/fuzz_split3d.py [seed] [seconds]"""
import os
import sys
import time

os.environ.setdefault("TAU3D_SPLIT", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_sims_amd as f
from oracle import pyoracle
from tests.parity import assert_parity

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
t_end = time.time() + seconds
n = bad = 0
while time.time() < t_end:
    nx, ny, nz = int(rng.integers(8, 221)), int(rng.integers(8, 221)), int(rng.integers(8, 41))
    try:
        e = f.Tau3D(nx, ny, nz)
        o = pyoracle.Oracle3D(nx, ny, nz)
        e.init(1)
        e.set_clock(0.02, 1e-4)
        e.step(int(rng.integers(0, 10)))
        st = e.download()
        if not all(np.isfinite(a).all() and np.abs(a).max() < 30 for a in st):
            e.close()
            continue
        dt = 2e-6
        s = o.from_interior(st)
        o.fill_halo_periodic(s)
        out = o.new_state()
        o.step_range(s, out, dt, 1.0)
        e.step_explicit(dt, 1.0)
        got = e.download()
        assert_parity(got, o.interior(out), mask=o.interior([o.solid])[0] == 0, what=f"split 3D {nx, ny, nz}")
        e.close()
        n += 1
    except AssertionError as ex:
        bad += 1
        print("FAIL", (nx, ny, nz), str(ex)[:300], flush=True)
print(f"fuzz_split3d seed {seed}: {n} shapes compared, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
