"""GPU: a short fixed-seed slice of scripts/fuzz_parity.py — random ragged shapes for every kernel, each compared
with its oracle at the tolerances of the dedicated tests (bit-exact for Gray-Scott, LBM, masks and SPH cell indices)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_shapes(eng, oracle_built, seed):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    msgs = []
    it, bad = m.sweep(seed=seed, seconds=90.0, max_iter=90, log=msgs.append)
    assert it == 90 and bad == 0, "\n".join(msgs)


def test_split_3d_step_on_random_shapes():
    """a short slice of scripts/fuzz_split3d.py: the two-kernel 3D step on ragged planes up to ~220^2 against the oracle
    (a 150 s run of it compared 626 shapes without a failure)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_split3d.py"), "7", "12"], capture_output=True,
                       text=True, cwd=ROOT, env=dict(os.environ, TAU3D_SPLIT="1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "0 failures" in r.stdout


def test_3d_step_on_random_shapes_vs_the_reference_kernel():
    """a short slice of scripts/fuzz_ref3d.py: ragged shapes, fused / split step, fast / FORCED reciprocal WENO weights, against the
    reference's own k_step running on the same GPU (oracle/_ref/th3cs.co)"""
    import subprocess
    import sys
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "th3cs.co")):
        pytest.skip("oracle/_ref/th3cs.co absent: oracle/build_ref.sh has not run (needs /root/reference)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_ref3d.py"), "11", "25"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1500:]
    assert "0 failures" in r.stdout and "split/rcp" in r.stdout and "split/fast" in r.stdout


def test_2d_simulators_on_random_cases_vs_the_reference_kernels():
    """a short slice of scripts/fuzz_ref2d.py: Gray-Scott and LBM on random ragged grids / parameters (bit-exact), SPH with random
    particle counts and parameters, 2D Euler with random SimConfig values at 8192 x 1024 — against the reference's own kernels"""
    import subprocess
    import sys
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "tau_sph.co")):
        pytest.skip("oracle/_ref absent: oracle/build_ref.sh has not run (needs /root/reference)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_ref2d.py"), "13", "40"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]
    assert "0 failures" in r.stdout


def test_predicted_uniform_tile_list_on_random_grids():
    """a short slice of scripts/fuzz_tile_list.py: the predicted-uniform tile list on / off / verifying over random grids of whole tiles,
    both starts, random step batches, z-march chunk lengths and a state write mid-run — bytes of every field, clock, tile flags
    (a 240 s run: 1 880 cases, 23.5 M predictions verified, no failure: profiles/r06/fuzz_tile_list.txt)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_tile_list.py"), "5", "15"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1500:]
    assert " 0 failures" in r.stdout


def test_ring_dumps_on_random_grids():
    """a short slice of scripts/fuzz_ring_dump.py: bin/tau3d over 2-4 ranks sharing the device against the single domain, whole dumps
    byte for byte, on random grids of whole tiles with thin and ragged slabs (a 300 s run of it found the stale uniform-plane count
    behind a wave that leaves the body — `urun` in update_z_body — that no fixed shape had shown; since the fix: 248 cases, 0 failures)"""
    import subprocess
    import sys
    if not os.path.exists(os.path.join(ROOT, "bin", "tau3d")):
        pytest.skip("bin/tau3d not built (make tau3d)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_ring_dump.py"), "3", "20"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1500:]
    assert " 0 failures" in r.stdout
