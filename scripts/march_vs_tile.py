#!/usr/bin/env python
"""march_vs_tile.py — k_flux_xy's y-marching body (round 6) against the tile kernel of rounds 2-5, bit for bit.
The same start, the same steps, through the in-tree library (march) and a variant build with -DTAU3D_XY_MARCH=0 (tile):
  scripts/variant_build.sh tile "-DTAU3D_XY_MARCH=0"; python scripts/march_vs_tile.py
Every field of every cell after every listed step count must be byte-identical (same arithmetic, same operands, same association)."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [((96, 64, 32), 1, 12), ((61, 61, 50), 1, 9), ((14, 29, 35), 1, 6), ((160, 128, 96), 1, 20), ((200, 136, 24), 1, 15), ((64, 300, 16), 1, 10),
         ((256, 192, 128), 1, 30), ((130, 70, 20), 0, 8), ((512, 512, 512), 1, 30)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import fluid_sims_amd as f
    out = {}
    for shape, mode, steps in CASES:
        e = f.Tau3D(*shape)
        e.set_split(True)
        e.init(mode)
        if mode:
            e.set_clock(0.02, 1e-4)
        for k in (1, steps - 1):
            e.step(k)
        st = e.download()
        c = e.clock()
        h = hashlib.sha256()
        for a in st:
            h.update(np.ascontiguousarray(a).tobytes())
        out[str(shape)] = [h.hexdigest(), c.t, c.d_tau, c.maxs, float(np.abs(st[1]).max())]
        e.close()
    print("RESULT " + json.dumps(out))
    sys.exit(0)
res = {}
for name, lib in (("march", None), ("tile", os.path.join(ROOT, "build_var", "tile", "libtaueng.so"))):
    env = dict(os.environ)
    if lib:
        env["TAUENG_LIB"] = lib
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    res[name] = json.loads(line[0][7:])
bad = 0
for k in res["march"]:
    same = res["march"][k] == res["tile"][k]
    bad += not same
    print(("identical " if same else "DIFFERENT ") + k, res["march"][k][:1][0][:16], res["tile"][k][0][:16], "maxs", res["march"][k][3], "max|phix|", res["march"][k][4])
sys.exit(1 if bad else 0)
