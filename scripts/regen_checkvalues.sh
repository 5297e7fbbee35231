#!/bin/bash
# scripts/regen_checkvalues.sh — regenerates the CPU-solver check-values of tests/golden/ref_checkvalues.json and the field
# fixture tests/golden/cpu2d_ref_96x64_steps12_13.npz from the reference's own source and compares them with the committed data.
#
# BUILD CONTAINER ONLY (needs /root/reference; listed in .gpurunignore).  The programs it runs are oracle/_ref/libref_hyp_cpu*.so:
# oracle/build_ref.sh's pure line cuts of tau_hypersonic.c (lines 1-674) and tau_hypersonic_simd.c (1-804) minus the raylib
# include, compiled by gcc with the reference Makefile's flags — no stand-in header (the declarations-only raylib.h this script
# used to write is gone).  The CUDA entries of the JSON (2D / 3D / Gray-Scott / SPH) came from the survey's host-side block
# emulator (Appendix A); those are pinned on the GPU against oracle/_ref/*.co instead (tests/test_gpu_ref*.py).
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "${TAU_REFERENCE:-/root/reference}" ] || { echo "regen_checkvalues: no reference tree (build container only)"; exit 2; }
sh "$ROOT/oracle/build_ref.sh"
cd "$ROOT"
python3 - <<'PY'
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import refcpu

def sums(r):                      # the survey's harness: left-to-right sums over fluid cells
    u, m = r.state()
    fl = m.ravel() == 0
    acc = lambda a: float(np.add.accumulate(a.ravel()[fl])[-1])
    return dict(t=r.t, fluid=int(fl.sum()), sum_rho=acc(u[..., 0]), sum_mx=acc(u[..., 1]), sum_E=acc(u[..., 3]))

def run(W, H, n, simd=False):
    r = refcpu.RefHypCpu(W, H, simd=simd)
    r.step(n)
    return sums(r)

gold = json.load(open("tests/golden/ref_checkvalues.json"))
g3, g2 = gold["tau_hypersonic_cpu_300sq"], gold["tau_hypersonic_cpu_256sq_8steps"]
a1, a24, b8, s10 = run(300, 300, 1), run(300, 300, 24), run(256, 256, 8), run(300, 300, 10, simd=True)
checks = [("300^2 t after 1 step", a1["t"], g3["t_1step"]), ("300^2 t after 24", a24["t"], g3["t_24steps"]),
          ("300^2 fluid", a24["fluid"], g3["fluid"]), ("300^2 sum rho", a24["sum_rho"], g3["sum_rho_24"]),
          ("300^2 sum mx", a24["sum_mx"], g3["sum_mx_24"]), ("300^2 sum E", a24["sum_E"], g3["sum_E_24"]),
          ("256^2 t after 8", b8["t"], g2["t"]), ("256^2 fluid", b8["fluid"], g2["fluid"]), ("256^2 sum rho", b8["sum_rho"], g2["sum_rho"]),
          ("SIMD file 300^2 sum rho after 10 (tests/test_drivers.py)", s10["sum_rho"], 82947.469425548319)]
bad = 0
for name, a, b in checks:
    ok = a == b
    bad += not ok
    print(("ok   " if ok else "DIFF ") + f"{name}: regenerated {a!r} committed {b!r}")
# field fixture (SURVEY 8c fixture ii for config C1): 96 x 64 after 12 and 13 steps
r = refcpu.RefHypCpu(96, 64)
r.step(12)
u0, m = r.state(); t0 = r.t
r.step(1)
u1, _ = r.state(); t1 = r.t
new = dict(W=96, H=64, steps0=12, steps1=13, t0=t0, t1=t1, U0=np.moveaxis(u0, 2, 0), U1=np.moveaxis(u1, 2, 0), mask=m)
path = "tests/golden/cpu2d_ref_96x64_steps12_13.npz"
if os.path.exists(path):
    old = np.load(path)
    same = all(np.array_equal(old[k], np.asarray(v)) for k, v in new.items())
    bad += not same
    print(("ok   " if same else "DIFF ") + "96x64 field fixture (steps 12 -> 13) against the committed tests/golden file")
else:
    np.savez_compressed(path, **new); print("wrote", path)
sys.exit(1 if bad else 0)
PY
