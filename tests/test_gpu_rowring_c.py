"""GPU: the row-slab ring of the 2D stencil handles INSIDE the library (csrc/ring.hip: taugs_ring_* / taulap_ring_*) and the
plain-C driver `tgs --gpus N` — no Python, no torch.distributed on the data path (round-4 review, "What's missing" 5).

The GPU box has one device: N forked ranks share it over the host-staged transport (the ring's ordering, the rendezvous and the
per-rank gather are the production code; only the transport differs from RCCL), and a world of one runs RCCL send / recv to
itself.  Everything must be bit-identical to the single-domain run: a 5-point stencil carries the wrap-around error of a local
periodic array one row per step, and the halos are refreshed every `halo` steps."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TGS = os.path.join(ROOT, "bin", "tgs")


def run(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TAU3D_RING_TIMEOUT="60")
    return subprocess.run(list(args), capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)


def dump_of(tmp_path, name, *flags):
    path = str(tmp_path / name)
    r = run(TGS, *flags, "--dump", path)
    assert r.returncode == 0, r.stdout + r.stderr
    return open(path, "rb").read(), r.stdout


def summary(out):
    line = [l for l in out.splitlines() if " steps on " in l][-1]
    return line.split("sum u")[1]


@pytest.mark.parametrize("nx,ny,steps", [(128, 128, 50), (256, 96, 37), (1024, 512, 21)])
def test_tgs_row_ring_is_bit_identical(eng, tmp_path, nx, ny, steps):
    grid = ["--nx", str(nx), "--ny", str(ny), "--steps", str(steps)]
    want, out1 = dump_of(tmp_path, "single.bin", *grid)
    assert len(want) > 2 * 4 * nx * ny
    for world, halo in ((2, 4), (3, 4), (4, 8), (8, 4), (2, 1)):
        if ny // world < halo:
            continue
        got, out = dump_of(tmp_path, f"w{world}h{halo}.bin", *grid, "--gpus", str(world), "--transport", "host", "--halo", str(halo))
        assert f"row ring: {world} ranks, host-staged transport, {halo}-row halos" in out
        assert got == want, f"world {world}, halo {halo}: dump differs from the single-domain run"
        assert summary(out) == summary(out1)        # sum u / sum v to nine digits, as the reference prints them


def test_tgs_rccl_needs_one_device_per_rank(eng):
    import ctypes
    n = ctypes.c_int()
    eng.load().tau_device_count(ctypes.byref(n))
    if n.value >= 2:
        pytest.skip("this box has several devices")
    r = run(TGS, "--nx", "64", "--ny", "64", "--steps", "4", "--gpus", "2")
    assert r.returncode != 0 and "needs 2 devices" in r.stderr


@pytest.mark.parametrize("kind", ["gs", "sw", "burgers"])
@pytest.mark.parametrize("transport", ["local", "rccl"])
def test_row_ring_binding_world1(eng, kind, transport):
    """RowRing (the ctypes mirror) with a world of one: the slab's own first / last rows are its halos — device copies, and RCCL
    send / recv to itself — for Gray-Scott and both viscosity passes, with a step count that is not a multiple of the halo"""
    nx, ny, H, steps = 192, 160, 4, 11
    rng = np.random.default_rng(5)
    if kind == "gs":
        a = (1.0 - 0.5 * rng.random((ny, nx))).astype(np.float32)
        b = (0.25 * rng.random((ny, nx))).astype(np.float32)
        make = lambda rows: eng.GrayScott(nx, rows)
    else:
        a, b = (rng.standard_normal((ny, nx)).astype(np.float32) for _ in range(2))
        make = lambda rows: eng.Laplacian2D(nx, rows, kind, 0.1, 0.2)
    ref = make(ny)
    ref.upload(a, b)
    ref.step(steps)
    wa, wb = ref.download()
    ref.close()
    e = make(ny + 2 * H)
    idx = np.arange(-H, ny + H) % ny
    e.upload(np.ascontiguousarray(a[idx]), np.ascontiguousarray(b[idx]))
    ring = eng.RowRing(e, H, 0, 1, {"local": eng.RING_LOCAL, "rccl": eng.RING_RCCL}[transport])
    ring.step(steps)
    ring.finish()
    info = ring.info()
    ga, gb = e.download()
    ring.close()
    e.close()
    assert info["nyl"] == ny and info["halo"] == H and info["exchanges"] == 3
    if transport == "rccl":
        assert info["rccl_version"] > 0 and info["comm_ranks"] == 1
    assert np.array_equal(ga[H:H + ny], wa) and np.array_equal(gb[H:H + ny], wb)


def test_row_ring_refuses_bad_arguments(eng):
    e = eng.GrayScott(64, 10)                    # 10 rows with 4-row halos: 2 owned rows < halo
    with pytest.raises(eng.TauError, match="at least the halo depth"):
        eng.RowRing(e, 4, 0, 1, eng.RING_LOCAL)
    with pytest.raises(eng.TauError, match="rccl, host or local"):
        eng.RowRing(e, 1, 0, 1, eng.RING_IPC)
    with pytest.raises(eng.TauError, match="rendezvous"):
        eng.RowRing(e, 1, 0, 2, eng.RING_HOST)
    e.close()
