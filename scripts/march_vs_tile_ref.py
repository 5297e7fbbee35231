import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
shape = (96, 64, 32)
pre = 2
if sys.argv[1] == "child":
    import fluid_sims_amd as f
    e = f.Tau3D(*shape); e.set_split(True); e.init(1); e.set_clock(0.02, 1e-4)
    e.step(pre)
    s0 = np.stack(e.download())
    c = e.clock()
    dt = float(np.float32(c.t * np.float32(np.exp(np.float32(c.d_tau)))) * np.float32(c.d_tau))
    m = e.step_explicit(dt, 1.0)
    np.savez(sys.argv[2], s0=s0, s1=np.stack(e.download()), dt=dt)
    sys.exit(0)
for name, lib in (("march", None), ("tile", os.path.join(ROOT, "build_var", "tile", "libtaueng.so"))):
    env = dict(os.environ)
    if lib: env["TAUENG_LIB"] = lib
    subprocess.run([sys.executable, __file__, "child", f"/tmp/{name}.npz"], check=True, env=env)
a, b = np.load("/tmp/march.npz"), np.load("/tmp/tile.npz")
assert np.array_equal(a["s0"], b["s0"]) and a["dt"] == b["dt"]
from oracle import refgpu
r = refgpu.Ref3D(*shape)
r.upload(list(a["s0"]))
r.step(float(a["dt"]), 1.0)
ref = np.stack(r.download())
solid = r.solid_mask()
d = np.abs(a["s1"].astype(np.float64) - b["s1"])
idx = np.argwhere(d > 0)
print("differing entries", len(idx))
for f_, z, y, x in idx[:24]:
    print((f_, z, y, x), "march %.7f tile %.7f ref %.7f" % (a["s1"][f_, z, y, x], b["s1"][f_, z, y, x], ref[f_, z, y, x]),
          "solid nb x-1,x+1,y-1,y+1,z-1,z+1:", solid[z, y, x - 1], solid[z, y, x + 1], solid[z, y - 1, x], solid[z, y + 1, x], solid[z - 1, y, x], solid[z + 1, y, x])
