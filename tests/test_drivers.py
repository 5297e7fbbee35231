"""The plain-C drivers that keep the reference's Makefile target names: CLI validation (CPU) and
end-to-end runs through the C-ABI (GPU)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_checkvalues.json")))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(os.path.join(BIN, "tau_sph")):
        subprocess.run(["make", "-C", ROOT, "-j4"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return BIN


def run(*args, **kw):
    return subprocess.run(list(args), capture_output=True, text=True, cwd=ROOT, **kw)


def test_reference_target_names_exist(built):
    for t in ("tau_hypersonic", "tau_hypersonic_simd", "tau_2d_hypersonic_cuda", "tau_hypersonic_cuda_tests", "tau3d",
              "tgs", "tau_sph", "tau_burgers", "tau_sw", "tau_lbm", "th3cs"):
        assert os.access(os.path.join(built, t), os.X_OK), t


@pytest.mark.parametrize("args,msg", [
    (["--gamma", "1.0"], "Invalid --gamma: 1 (must be > 1)."),
    (["--cfl", "0"], "Invalid --cfl: 0 (must be > 0)."),
    (["--visc-nu", "-1"], "Invalid --visc-nu: -1 (must be >= 0)."),
    (["--mach", "-3"], "Invalid --mach: -3 (must be > 0)."),
    (["--steps-per-frame", "0"], "Invalid --steps-per-frame: 0 (must be in [1, 1024])."),
    (["--steps-per-frame", "2000"], "Invalid --steps-per-frame: 2000 (must be in [1, 1024])."),
    (["--geom-theta", "1.6"], "Invalid --geom-theta: 1.6 (must be in (0, pi/2))."),
    (["--geom-rb", "1", "--geom-rn", "50"], "Require geom-rb >= geom-rn*cos(theta)."),
    (["--tile-bx", "0"], "Invalid tile dimensions"),
    (["--mach", "abc"], "Invalid value for --mach: abc"),
    (["--bogus"], "Unknown or incomplete argument: --bogus"),
    (["--mach"], "Unknown or incomplete argument: --mach"),
])
def test_tau2d_cli_validation_matches_reference(built, args, msg):
    """same rules and messages as tau_hypersonic_cuda.cu:1458-1639; error -> usage on stderr, exit 1"""
    r = run(os.path.join(built, "tau_2d_hypersonic_cuda"), *args)
    assert r.returncode == 1
    assert msg in r.stderr and "Usage:" in r.stderr


def test_cpu_drivers_reproduce_reference(built):
    g = GOLD["tau_hypersonic_cpu_256sq_8steps"]
    r = run(os.path.join(built, "tau_hypersonic"), "--W", "256", "--H", "256", "--steps", "8")
    assert r.returncode == 0
    assert "t=%.17g" % g["t"] in r.stdout and "fluid=%d" % g["fluid"] in r.stdout and "sum_rho=%.17g" % g["sum_rho"] in r.stdout
    r = run(os.path.join(built, "tau_hypersonic_simd"), "--steps", "10")
    assert "sum_rho=82947.469425548319" in r.stdout   # the SIMD file's own value


def test_tests_binary_usage(built):
    assert run(os.path.join(built, "tau_hypersonic_cuda_tests"), "--nope").returncode == 2


def test_gpu_programs_refuse_without_gpu(built):
    import fluid_sims_amd as f
    if f.load().tau_device_available():
        pytest.skip("a GPU is visible")
    for t in ("tau3d", "tgs", "tau_sph"):
        r = run(os.path.join(built, t))
        assert r.returncode == 1 and "no CPU path" in r.stderr


def test_getopt_long_forms_without_gpu(built):
    """the reference's getopt_long tables accept `--opt=value` and unambiguous abbreviations (tau_gray_scott.cu:84-104,
    tau_sph.cu:395-418): parsing gets past them (to the no-GPU refusal on this box), unknown options get getopt's message"""
    import fluid_sims_amd as f
    if f.load().tau_device_available():
        pytest.skip("a GPU is visible")
    for cmd in (["tgs", "--nx=64", "--ny", "32", "--st=5", "--F=0.03"], ["tau_sph", "--n=4096", "--dTau=1e-3", "-s", "7"],
                ["tau_burgers", "--nx=64", "--ny=64"], ["tau_lbm", "--nx=64", "--no-obstacle"]):
        r = run(os.path.join(built, cmd[0]), *cmd[1:])
        assert r.returncode == 1 and "no CPU path" in r.stderr and "unrecognized" not in r.stderr, (cmd, r.stderr)
    r = run(os.path.join(built, "tgs"), "--bogus")
    assert "unrecognized option '--bogus'" in r.stderr


def test_tau3d_multi_rank_refuses_without_gpu(built):
    """tau3d --gpus N forks its ranks before touching the HIP runtime; each rank reports the missing device itself"""
    import fluid_sims_amd as f
    if f.load().tau_device_available():
        pytest.skip("a GPU is visible")
    r = run(os.path.join(built, "tau3d"), "--n", "32", "--frames", "1", "--gpus", "2", "--transport", "host")
    assert r.returncode == 1 and r.stderr.count("no CPU path") == 2 and "rank 1 exited with 1" in r.stderr
    assert run(os.path.join(built, "tau3d"), "--gpus", "0").returncode == 1
    assert "rccl | ipc | host | ipc-host" in run(os.path.join(built, "tau3d"), "--transport", "mpi").stderr


def test_tgs_multi_rank_refuses_without_gpu(built):
    """tgs --gpus N (the row-slab ring in the library, round 5) forks its ranks before touching the HIP runtime as well"""
    import fluid_sims_amd as f
    if f.load().tau_device_available():
        pytest.skip("a GPU is visible")
    r = run(os.path.join(built, "tgs"), "--nx", "64", "--ny", "64", "--steps", "4", "--gpus", "2", "--transport", "host")
    assert r.returncode == 1 and r.stderr.count("no CPU path") == 2 and "rank 1 exited with 1" in r.stderr
    assert run(os.path.join(built, "tgs"), "--gpus", "0").returncode == 1
    assert "rccl | host" in run(os.path.join(built, "tgs"), "--transport", "mpi").stderr
    assert run(os.path.join(built, "tgs"), "--halo", "0").returncode == 1


@pytest.mark.gpu
def test_tgs_getopt_forms(built):
    """`--nx=64 --st=20` (getopt_long forms of the reference, tau_gray_scott.cu:95-104) run the same problem as the spaced forms"""
    a = run(os.path.join(built, "tgs"), "--nx=64", "--ny=48", "--st=20", "--seed=5")
    b = run(os.path.join(built, "tgs"), "--nx", "64", "--ny", "48", "--steps", "20", "--seed", "5")
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert re.search(r"sum u = \S+, sum v = \S+", a.stdout).group(0) == re.search(r"sum u = \S+, sum v = \S+", b.stdout).group(0)
    assert "20 steps on 64x48" in a.stdout


@pytest.mark.gpu
def test_tgs_end_to_end(built):
    g = GOLD["gray_scott_128sq_100steps"]
    r = run(os.path.join(built, "tgs"), "--headless", "--steps", "100")
    assert r.returncode == 0, r.stderr
    m = re.search(r"sum u = (\S+), sum v = (\S+)", r.stdout)
    assert float(m.group(1)) == pytest.approx(g["sum_u"], rel=1e-9) and float(m.group(2)) == pytest.approx(g["sum_v"], rel=1e-9)


@pytest.mark.gpu
def test_tau3d_end_to_end(built, tmp_path):
    g = GOLD["tau3d_32cube_4steps"]
    d = tmp_path / "s.bin"
    r = run(os.path.join(built, "tau3d"), "--n", "32", "--frames", "2", "--dump", str(d))
    assert r.returncode == 0, r.stderr
    raw = open(d, "rb").read()
    hdr, body = raw.split(b"\n", 1)
    a = np.frombuffer(body, np.float32).reshape(6, 32, 32, 32)
    assert b"steps=4" in hdr
    assert float(a[0].sum(dtype=np.float64)) == pytest.approx(g["sum_xi"], rel=1e-6)
    assert "Gcell-updates/s" in r.stdout


@pytest.mark.gpu
def test_make_test_round_trip(built, tmp_path):
    """the reference's `make test`: write a baseline, verify the same run against it (Makefile:39-43)"""
    b = str(tmp_path / "base.txt")
    exe = os.path.join(built, "tau_hypersonic_cuda_tests")
    r1 = run(exe, "--steps", "24", "--W", "1024", "--H", "256", "--write-baseline", "--baseline", b)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    r2 = run(exe, "--steps", "24", "--W", "1024", "--H", "256", "--verify-baseline", "--baseline", b)
    assert r2.returncode == 0 and " 0 failed" in r2.stdout, r2.stdout + r2.stderr
    assert len(open(b).read().split("\n")) >= 12


@pytest.mark.gpu
def test_tau2d_and_sph_end_to_end(built):
    r = run(os.path.join(built, "tau_2d_hypersonic_cuda"), "--W", "512", "--H", "256", "--frames", "2")
    assert r.returncode == 0 and "step 4" in r.stdout, r.stdout + r.stderr
    m = re.search(r"step 4  t=(\S+)", r.stdout)
    assert float(m.group(1)) == pytest.approx(GOLD["tau2d_cuda_512x256_4steps_tile32x4"]["t"], rel=1e-5)
    r = run(os.path.join(built, "tau_sph"), "--n", "4096", "--headless", "--steps", "3")
    assert r.returncode == 0 and "grid=16x16" in r.stdout and "rain=on" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_burgers_and_sw_end_to_end(built):
    r = run(os.path.join(built, "tau_burgers"), "--headless", "--colehopf", "--nx", "512", "--dtau", "1e-3", "--muscl",
            "--steps", "1500")
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"relative L2 error at elapsed t=\S+: (\S+)", r.stdout)
    assert float(m.group(1)) < 6e-5
    r = run(os.path.join(built, "tau_sw"), "--headless", "--nx", "256", "--ny", "256", "--steps", "50", "--dtau", "0.01")
    # the reference's own summary lines (tau_shallow_water.cu:774-780)
    assert r.returncode == 0 and "Headless benchmark (stride=5):" in r.stdout and "Simulated steps: 50" in r.stdout, r.stdout + r.stderr


def read_ppm(path):
    raw = open(path, "rb").read()
    magic, dims, maxv, body = raw.split(b"\n", 3)
    w, h = map(int, dims.split())
    assert magic == b"P6" and maxv == b"255" and len(body) == 3 * w * h
    return np.frombuffer(body, np.uint8).reshape(h, w, 3)


@pytest.mark.gpu
def test_headless_images(built, tmp_path):
    """--ppm stands in for the raylib window: the 3D slice is the grey ramp of slice_to_rgba, the 2D frame
    the blue-green-red ramp with the body in grey 110 (tau_hypersonic_cuda.cu:692-704, 1265-1268)."""
    p3 = str(tmp_path / "s.ppm")
    r = run(os.path.join(built, "tau3d"), "--n", "32", "--frames", "15", "--start", "1", "--vis", "0", "--ppm", p3)
    assert r.returncode == 0 and "outflow |dp|=" in r.stdout, r.stdout + r.stderr
    im = read_ppm(p3)
    assert im.shape == (32, 32, 3) and (im[..., 0] == im[..., 1]).all() and im.max() == 255 and im.min() == 0
    p2 = str(tmp_path / "f.ppm")
    r = run(os.path.join(built, "tau_2d_hypersonic_cuda"), "--W", "512", "--H", "256", "--frames", "30", "--view", "5",
            "--ppm", p2)
    assert r.returncode == 0 and "view mode 5" in r.stdout, r.stdout + r.stderr
    im = read_ppm(p2)
    assert im.shape == (256, 512, 3)
    body = (im == 110).all(axis=2)
    assert 0 < body.sum() < 0.2 * body.size            # the sphere-cone body is there, in grey
    assert (im[:, 0, 0] == 255).all() and (im[:, 0, 2] == 0).all()   # Mach view: the inflow column sits at the red end
    r = run(os.path.join(built, "tau_2d_hypersonic_cuda"), "--W", "64", "--H", "64", "--frames", "1", "--view", "9",
            "--ppm", p2)
    assert r.returncode == 1 and "view mode 9 outside 0..6" in r.stderr


@pytest.mark.gpu
def test_tau_sph_rain_xsph_and_raster(built, tmp_path):
    """reference defaults: rain on (tau_sph.cu:76); --muscl switches XSPH on (:480-482); --pgm = the ncurses raster"""
    pg = str(tmp_path / "r.pgm")
    r = run(os.path.join(built, "tau_sph"), "--n", "16384", "--headless", "--steps", "40", "--muscl", "--pgm", pg)
    assert r.returncode == 0 and "rain=on xsph=on eps=0.25" in r.stdout, r.stdout + r.stderr
    assert int(re.search(r"rain: (\d+) drops", r.stdout).group(1)) > 0
    raw = open(pg, "rb").read()
    magic, dims, maxv, body = raw.split(b"\n", 3)
    assert magic == b"P5" and dims == b"80 48" and len(body) == 80 * 48
    im = np.frombuffer(body, np.uint8).reshape(48, 80)
    assert im[30:].mean() > 10 * max(im[:10].mean(), 0.1)      # the fluid sits at the bottom (y flipped), rain is sparse
    r = run(os.path.join(built, "tau_sph"), "--n", "4096", "--headless", "--steps", "3", "--no-rain")
    assert "rain=off xsph=off" in r.stdout


@pytest.mark.gpu
def test_tau_lbm_end_to_end(built, tmp_path):
    pg = str(tmp_path / "l.pgm")
    r = run(os.path.join(built, "tau_lbm"), "--nx", "256", "--ny", "128", "--headless", "--steps", "400", "--radius", "16",
            "--drive", "1e-4", "--pgm", pg)
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(r"LBM D2Q9: 400 steps, 32768 cells, [0-9.]+ MLUPS", r.stdout)     # tau_lbm.cu:297-299
    raw = open(pg, "rb").read()
    magic, dims, maxv, body = raw.split(b"\n", 3)
    im = np.frombuffer(body, np.uint8).reshape(128, 256)
    assert dims == b"256 128" and (im[0] == 255).all() and (im[-1] == 255).all()        # channel walls
    assert im[64, 72] == 255 and im[64, 200] < 255                                       # cylinder at 0.28 nx; fluid behind it
    r = run(os.path.join(built, "tau_lbm"), "--nx", "4", "--tau", "0.1", "--headless", "--steps", "2")
    assert "2 steps, 4096 cells" in r.stdout                                             # nx, ny clamp to >= 16 ... ny default 256
