"""The reference's own k_step under its own controller (oracle/refgpu.Ref3D.run) from the impulsive 512^3 start, beside the engine
from the same start: does the input live longer in the reference than in the engine?   python scripts/long_run_512_ref.py [steps] [n]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import fluid_sims_amd as f  # noqa: E402
from oracle import refgpu  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 90
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
e = f.Tau3D(n)
e.init(1)
e.set_clock(0.02, 1e-4)
st = e.download()
r = refgpu.Ref3D(n)
r.upload(st)
r.t, r.d_tau = np.float32(0.02), np.float32(1e-4)
done = 0
while done < steps:
    ce = e.step(5)
    cr = r.run(5)
    done += 5
    xi = r.a[0].get(np.float32, r.shape)
    print(f"step {done:4d}  engine: t={ce.t:.7g} d_tau={ce.d_tau:.4g} maxs={ce.maxs:.6g} range={max(e.field_range()[:2]):.4g}   "
          f"reference kernel: t={cr['t']:.7g} d_tau={cr['d_tau']:.4g} maxs={cr['maxs']:.6g} min rho={float(np.exp(xi.min())):.3g} finite={bool(np.isfinite(xi).all())}", flush=True)
