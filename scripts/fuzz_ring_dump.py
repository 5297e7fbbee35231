"""Random sweep of the Z-slab ring against the single domain through bin/tau3d's dumps (every plane of the six fields + the clock,
byte for byte): random grids of whole k_flux_xy tiles (so that slabs keep the predicted-uniform tile list), 2-4 ranks sharing the
device, both transports that allow it, both starts.
  python scripts/fuzz_ring_dump.py [seed] [seconds]"""
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = random.Random(seed)
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
t_end = time.time() + seconds
n = bad = 0
with tempfile.TemporaryDirectory() as tmp:
    while time.time() < t_end:
        nx, ny = 32 * rng.randint(3, 8), 16 * rng.randint(3, 12)
        world = rng.randint(2, 4)
        nz = rng.randint(7, 24) * world + rng.randint(0, world - 1)   # ragged slabs too (tau3d_slab_bounds)
        frames, start = rng.randint(1, 12), rng.randint(0, 1)
        if start == 1:
            frames = min(frames, 8)
        tr = rng.choice(["host", "ipc-host"])
        sched = rng.choice([{}, {}, {"TAU3D_RING_SPEC": "0"}, {"TAU3D_RING_SPEC": "0", "TAU3D_RING_PIPELINE": "0"}])   # the three step schedules
        if rng.randint(0, 4) == 0:   # ragged tiles: no tile list in either run
            nx, ny = rng.randint(40, 260), rng.randint(24, 200)
        grid = ["--nx", str(nx), "--ny", str(ny), "--nz", str(nz), "--frames", str(frames), "--start", str(start)]
        a, b = os.path.join(tmp, "s.bin"), os.path.join(tmp, "r.bin")
        r1 = subprocess.run([os.path.join(ROOT, "bin", "tau3d"), *grid, "--dump", a], capture_output=True, text=True, env=env)
        r2 = subprocess.run([os.path.join(ROOT, "bin", "tau3d"), *grid, "--gpus", str(world), "--transport", tr, "--dump", b],
                            capture_output=True, text=True, env=dict(env, **sched))
        ok = r1.returncode == 0 and r2.returncode == 0 and open(a, "rb").read() == open(b, "rb").read()
        n += 1
        if not ok:
            bad += 1
            print("FAIL", grid, world, tr, sched, r1.returncode, r2.returncode, (r2.stderr or "")[-300:], flush=True)
print(f"fuzz_ring_dump seed {seed}: {n} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
