"""How long do the two 512^3 inputs of SURVEY §8(d) live?  Engine only (the reference kernel agrees step by step where it was
compared: tests/test_gpu_ref3d.py).  Prints clock, max wavespeed and the |primitive| range every few steps.
  python scripts/long_run_512.py impulsive 80 5
  python scripts/long_run_512.py ramped 6000 250
"""
import sys
import time

sys.path.insert(0, ".")
import fluid_sims_amd as f  # noqa: E402

mode, nsteps, every = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 512
e = f.Tau3D(n)
if mode == "impulsive":
    e.init(1)
    e.set_clock(0.02, 1e-4)
else:
    e.init(0)
t0 = time.time()
done = 0
while done < nsteps:
    c = e.step(every)
    done += every
    print(f"step {done:5d} t={c.t:.6g} d_tau={c.d_tau:.4g} dt={c.dt:.4g} gain={c.gain:.4f} maxs={c.maxs:.6g} range={e.field_range()} wall={time.time() - t0:.1f}s",
          flush=True)
