import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shape = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (96, 64, 32)
if sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import fluid_sims_amd as f
    e = f.Tau3D(*shape); e.set_split(True); e.init(1); e.set_clock(0.02, 1e-4)
    e.step(int(sys.argv[4]))
    np.save(sys.argv[3], np.stack(e.download()))
    sys.exit(0)
import numpy as np
steps = sys.argv[3] if len(sys.argv) > 3 else "1"
for name, lib in (("march", None), ("tile", os.path.join(ROOT, "build_var", "tile", "libtaueng.so"))):
    env = dict(os.environ)
    if lib: env["TAUENG_LIB"] = lib
    subprocess.run([sys.executable, __file__, "child", ",".join(map(str, shape)), f"/tmp/{name}.npy", steps], check=True, env=env)
a, b = np.load("/tmp/march.npy"), np.load("/tmp/tile.npy")
d = np.abs(a.astype(np.float64) - b)
print("shape", shape, "steps", steps, "max diff per field", d.reshape(6, -1).max(1))
idx = np.argwhere(d > 0)
print("differing entries", len(idx), "of", d.size)
if len(idx):
    f_, z, y, x = idx.T
    print("x range", x.min(), x.max(), "y range", y.min(), y.max(), "z range", z.min(), z.max())
    import collections
    print("by y:", sorted(collections.Counter(y.tolist()).items())[:40])
    print("by x:", sorted(collections.Counter(x.tolist()).items())[:40])
    w = np.argmax(d); print("worst", np.unravel_index(w, d.shape), a.flat[w], b.flat[w])
