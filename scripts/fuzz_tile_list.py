"""Random sweep of the predicted-uniform tile list (include/taueng.h: tau3d_tile_list_stats) — the split 3D step with the list and
k_update_z's use of the predictions (TAU3D_TILE_LIST=1, the default) against the same step with neither (=0), every field of every
cell, the clock and the per-tile flags byte for byte, the verifying mode (=2: 0 mismatches), and the same step with every face of
every cell evaluated (TAU3D_UNIFORM_EXITS=0: fields and clock) — over random grids of whole tiles (3-10 tiles across, 3-16 up, 8-100 planes), both starts, with and without the body, random batches of steps, random chunk lengths
of the z march (TAU3D_ZCHUNK) and a state write (tau3d_upload_state of a dented state) at a random point.

  python scripts/fuzz_tile_list.py [seed] [seconds]"""
import ctypes
import os
import sys
import time

os.environ["TAU3D_SPLIT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_sims_amd as f

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
t_end = time.time() + seconds
n = bad = checked = skipped_any = 0


def run(tl, shape, mode, body, batches, dent_at, dent, zchunk):
    os.environ["TAU3D_TILE_LIST"] = str(max(tl, 0))
    if tl < 0:                                   # -1: every face of every cell evaluated, as the reference does
        os.environ["TAU3D_UNIFORM_EXITS"] = "0"
    else:
        os.environ.pop("TAU3D_UNIFORM_EXITS", None)
    if zchunk:
        os.environ["TAU3D_ZCHUNK"] = str(zchunk)
    else:
        os.environ.pop("TAU3D_ZCHUNK", None)
    p = f.Tau3DParams()
    f.load().tau3d_params_default(ctypes.byref(p), *shape)
    if not body:
        p.sdf_r = -1.0
    e = f.Tau3D(*shape, params=p)
    e.init(mode)
    if mode:
        e.set_clock(0.02, 1e-4)
    out = []
    for i, k in enumerate(batches):
        e.step(k)
        c = e.clock()
        st = e.download()
        out.append(([a.view(np.uint32).copy() for a in st], (c.t, c.d_tau, c.maxs), e.uniform_tiles(), e.tile_list_stats()))
        if i == dent_at:
            st = [a.copy() for a in st]
            st[dent[0]][dent[1]] += dent[2]
            e.upload(st)
    e.close()
    return out


while time.time() < t_end:
    shape = (32 * int(rng.integers(3, 11)), 16 * int(rng.integers(3, 17)), int(rng.integers(8, 101)))
    if rng.integers(0, 4) == 0:      # ragged tiles: the handle keeps no list (mode 0) — what is compared is the exits on / off
        shape = (int(rng.integers(40, 300)), int(rng.integers(24, 200)), int(rng.integers(8, 60)))
    mode, body = int(rng.integers(0, 2)), bool(rng.integers(0, 4))
    batches = [int(rng.integers(1, 8)) for _ in range(int(rng.integers(2, 7)))]
    if mode == 1 and sum(batches) > 30:
        batches = batches[:3]
    dent_at = int(rng.integers(0, len(batches)))
    dent = (int(rng.integers(0, 6)), (int(rng.integers(0, shape[2])), int(rng.integers(0, shape[1])), int(rng.integers(0, shape[0]))),
            float(rng.choice([0.25, -0.125, 1e-3])))
    zchunk = int(rng.choice([0, 0, 8, 16, 33, 64, 100]))
    a, b, v, o = (run(tl, shape, mode, body, batches, dent_at, dent, zchunk) for tl in (1, 0, 2, -1))
    ok = True
    for (sa, ca, ua, la), (sb, cb, ub, lb), (sv, cv, uv, lv), (so, co, uo, lo) in zip(a, b, v, o):
        same = ca == cb == cv == co and ua == ub == uv and all(np.array_equal(x, y) and np.array_equal(x, w) and np.array_equal(x, q)
                                                                for x, y, w, q in zip(sa, sb, sv, so))
        ok = ok and same and lv[4] == 0
    checked += v[-1][3][3]
    skipped_any += any(0 <= x[3][1] < x[3][2] for x in a)
    n += 1
    if not ok:
        bad += 1
        print("FAIL", shape, "mode", mode, "body", body, "batches", batches, "dent", dent_at, dent, "zchunk", zchunk, [x[3] for x in v], flush=True)
print(f"fuzz_tile_list seed {seed}: {n} cases, {bad} failures; {skipped_any} cases with a list shorter than the grid, {checked} predictions verified", flush=True)
sys.exit(1 if bad else 0)
