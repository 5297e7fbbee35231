"""What THE REFERENCE'S OWN KERNELS do on this MI355X at BASELINE.json's sizes (oracle/_ref/*.co: the reference sources compiled
for gfx950 with its Makefile's flags by oracle/build_ref.sh), launched as its mains launch them — next to the engine's number for
the same work on the same box.  Evidence for profiles/ (not a test, not part of bench.py: bench.py may touch oracle/ only in its
cpu_baseline leg).   python scripts/time_reference_kernels.py > profiles/r04/reference_kernels_mi355x.txt
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import fluid_sims_amd as f  # noqa: E402
from oracle import refgpu  # noqa: E402


def timed(fn, sync, n, warm=2):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n


out = {}
# ---- C5: 3D hypersonic 512^3, the developed state bench.py times (25 steps after the impulsive start)
n = 512
e = f.Tau3D(n)
e.init(1)
e.set_clock(0.02, 1e-4)
e.step(25)
state = e.download()
c = e.clock()
dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau))
t_eng = timed(lambda: e.step_async(1), e.sync, 20)
e.close()
r = refgpu.Ref3D(n)
r.upload(state)
t_ref = timed(lambda: r.step(dt, 1.0), r.m.sync, 5, warm=1)     # Ref3D.step syncs itself (8-byte max read-back, as the reference does)
r.close()
cells = n ** 3
out["tau3d_512"] = dict(reference_ms=t_ref * 1e3, reference_gcells=cells / t_ref / 1e9, engine_ms=t_eng * 1e3, engine_gcells=cells / t_eng / 1e9,
                        speedup=t_ref / t_eng, note="th3cs.cu k_step, block (8,8,4), 49 KB dynamic LDS; engine: k_flux_xy + k_update_z + clock")
print(json.dumps({"tau3d_512": out["tau3d_512"]}), flush=True)

# ---- C3: Gray-Scott 8192^2
n = 8192
g = f.GrayScott(n, n)
g.init_pattern(1337)
u, v = g.download()
t_eng = timed(lambda: g.step_async(40), g.sync, 5) / 40
g.close()
r = refgpu.RefGrayScott(n, n)
r.upload(u, v)
t_ref = timed(lambda: r.step(20), r.m.sync, 3) / 20
r.close()
out["gray_scott_8192"] = dict(reference_ms=t_ref * 1e3, reference_gcells=n * n / t_ref / 1e9, engine_ms=t_eng * 1e3, engine_gcells=n * n / t_eng / 1e9,
                              speedup=t_ref / t_eng, note="step_kernel 16x16 blocks, -ffast-math")
print(json.dumps({"gray_scott_8192": out["gray_scott_8192"]}), flush=True)

# ---- C2-like: 2D Euler at the size the reference fixes at compile time (8192 x 1024), fp64 reference vs fp32 engine
r = refgpu.RefH2()
r.init()
r.step(50)
t_ref = timed(lambda: r.step(1), r.m.sync, 20)
r.close()
e = f.Hypersonic2D(8192, 1024)
e.init()
e.step(50)
t_eng = timed(lambda: e.step_async(20), e.sync, 5) / 20
e.close()
cells = 8192 * 1024
out["tauh2_8192x1024"] = dict(reference_ms=t_ref * 1e3, reference_gcells=cells / t_ref / 1e9, engine_ms=t_eng * 1e3, engine_gcells=cells / t_eng / 1e9,
                              speedup=t_ref / t_eng, note="reference: fp64, 6 kernels + a blocking 8-byte copy per step (run_hypersonic_steps); engine: fp32 fused march")
print(json.dumps({"tauh2_8192x1024": out["tauh2_8192x1024"]}), flush=True)

# ---- C4: SPH 4 M particles, lattice start
N = 1 << 22
e = f.Sph2D(N)
e.reset_particles()
st = e.download()
dt = e.dt()
t_eng = timed(lambda: e.step_async(10), e.sync, 5) / 10
r = refgpu.RefSph(N, **{k: getattr(e.params, k) for k in "boxX boxY rho0 c0 gammaEOS hMul viscAlpha gravity useVisc useGrav".split()})
e.close()
r.upload(st["pos"], st["vel"])
t_ref = timed(lambda: r.substep(dt), r.m.sync, 5, warm=1)
r.close()
out["sph_4m"] = dict(reference_ms=t_ref * 1e3, reference_gparticles=N / t_ref / 1e9, engine_ms=t_eng * 1e3, engine_gparticles=N / t_eng / 1e9,
                     speedup=t_ref / t_eng, note="reference sub-step incl. the list read-back this driver adds (4 B x (N+M)); engine: counting-sort build")
print(json.dumps({"sph_4m": out["sph_4m"]}), flush=True)
