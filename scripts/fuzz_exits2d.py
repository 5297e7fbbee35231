"""Random sweep of the 2D Euler march's uniform-row exits (h2d::k_march_lds<1, true>) against the same step with every trip evaluated
(TAUH2_UNIFORM_EXITS=0): random grids that take the march (>= ~2 M cells) and smaller ones, random numbers of steps in batches, every
field of every cell and the clock byte for byte.
  python scripts/fuzz_exits2d.py [seed] [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_sims_amd as f

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
t_end = time.time() + seconds
n = bad = 0


def run(ex, W, H, batches):
    os.environ["TAUH2_UNIFORM_EXITS"] = ex
    h = f.Hypersonic2D(W, H)
    h.init()
    out = []
    for k in batches:
        t = h.step(k)
        out.append(([a.view(np.uint32).copy() for a in h.download()], t))
    h.close()
    return out


while time.time() < t_end:
    if rng.integers(0, 4) == 0:
        W, H = int(rng.integers(64, 700)), int(rng.integers(64, 700))
    else:
        W, H = int(rng.integers(1200, 3000)), int(rng.integers(1000, 2600))
    batches = [int(rng.integers(1, 120)) for _ in range(int(rng.integers(1, 4)))]
    a, b = run("1", W, H, batches), run("0", W, H, batches)
    ok = all(ta == tb and all(np.array_equal(x, y) for x, y in zip(sa, sb)) for (sa, ta), (sb, tb) in zip(a, b))
    n += 1
    if not ok:
        bad += 1
        print("FAIL", W, H, batches, flush=True)
print(f"fuzz_exits2d seed {seed}: {n} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
