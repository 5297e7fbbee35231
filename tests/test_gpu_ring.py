"""GPU: the Z-slab ring INSIDE the library (csrc/ring.hip, tau3d_ring_*) and the plain-C driver `tau3d --gpus N`.

The GPU box has one device.  What runs here:
  * `bin/tau3d --gpus N --transport host`: N forked ranks sharing the device, halos staged through the shared rendezvous
    file — the ring's ordering (prime, begin / edges / exchange / interior / all-reduce / end, events between the two
    streams), the rendezvous and the per-rank dump are the production code; only the transport differs from RCCL;
  * the RCCL transport with a communicator of one (ncclSend / ncclRecv to itself + ncclAllReduce), from C and from Python;
  * `--gpus 2` over RCCL on this one-device box fails from INSIDE the ranks with a clear message.
All results must be bit-identical to the single-domain run (same kernels, same inputs: SURVEY §8e fixture iv)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAU3D = os.path.join(ROOT, "bin", "tau3d")


def run(*args, **kw):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("TAU3D_RING_TIMEOUT", "60")
    return subprocess.run(list(args), capture_output=True, text=True, cwd=ROOT, env=env, timeout=600, **kw)


def dump_of(tmp_path, name, *flags):
    path = str(tmp_path / name)
    r = run(TAU3D, *flags, "--dump", path)
    assert r.returncode == 0, r.stdout + r.stderr
    return open(path, "rb").read(), r.stdout


SHAPES = [((32, 32, 64), 6), ((64, 48, 40), 5),
          ((160, 128, 48), 3)]      # planes >= 128^2: the split step (k_flux_xy + k_update_z), with an interior piece at world 2


# host: packed buffers staged through the rendezvous file.  ipc-host: the DIRECT transport — every rank maps its neighbours' state
# allocations (hipIpcGetMemHandle / hipIpcOpenMemHandle through the rendezvous file) and copies its new boundary planes straight
# into their halo planes on the exchange stream; only the 8-byte all-reduce is staged by the host here, because RCCL refuses ranks
# that share a device (with one device per rank the same copies run beside RCCL's all-reduce: --transport ipc).
@pytest.mark.parametrize("transport,banner", [("host", "host-staged transport"), ("ipc-host", "direct halos (IPC-mapped neighbours), host all-reduce")])
@pytest.mark.parametrize("shape,frames", SHAPES)
def test_tau3d_host_ring_is_bit_identical(eng, tmp_path, shape, frames, transport, banner):
    nx, ny, nz = shape
    grid = ["--nx", str(nx), "--ny", str(ny), "--nz", str(nz), "--frames", str(frames), "--start", "1"]
    want, _ = dump_of(tmp_path, "single.bin", *grid)
    assert len(want) > 6 * 4 * nx * ny * nz
    for world in (2, 3, 4, 8):
        if nz // world < 6:
            continue
        got, out = dump_of(tmp_path, f"w{world}.bin", *grid, "--gpus", str(world), "--transport", transport)
        assert f"ring: {world} ranks, {banner}" in out
        assert got == want, f"world {world}: dump differs from the single-domain run"


def test_tau3d_host_ring_at_the_benchmarked_plane_size(eng, tmp_path):
    """512 x 512 planes, two ranks of 64 planes each — the slab one GPU of the 8-way 512^3 run owns, edges + interior + exchange as
    there — against the 512 x 512 x 128 single domain: the whole dump (every plane of the six fields and the clock with its max
    wavespeed) byte for byte.  (Round-3 review: the only 512^2-plane record, taken before the -ffp-contract=on fix, showed a 1-ulp
    difference in maxs.)"""
    grid = ["--nx", "512", "--ny", "512", "--nz", "128", "--frames", "3", "--start", "1"]
    frames = lambda o: [l for l in o.splitlines() if l.startswith("frame ")]
    want, out1 = dump_of(tmp_path, "single.bin", *grid)
    assert len(want) > 6 * 4 * 512 * 512 * 128
    got, out = dump_of(tmp_path, "w2.bin", *grid, "--gpus", "2", "--transport", "host")
    assert "ring: 2 ranks, host-staged transport" in out
    assert got == want, "2 ranks x 64 planes of 512^2: dump differs from the single-domain run"
    got, out = dump_of(tmp_path, "w2ipc.bin", *grid, "--gpus", "2", "--transport", "ipc-host")
    assert "direct halos" in out
    assert got == want, "2 ranks x 64 planes of 512^2, direct halos: dump differs from the single-domain run"
    assert frames(out) == frames(out1)
    # the same through RCCL talking to itself (one rank owning all 128 planes: slab_begin / edges / interior / all-reduce)
    got, out = dump_of(tmp_path, "ring1.bin", *grid, "--ring")
    assert got == want
    assert frames(out) == frames(out1) and len(frames(out1)) >= 1, (frames(out1), frames(out))   # t, d_tau, dt, maxs to 9 digits
    print("single:", frames(out1)[-1], "| ring:", frames(out)[-1])


def test_tau3d_rccl_self_ring_is_bit_identical(eng, tmp_path):
    grid = ["--nx", "64", "--ny", "48", "--nz", "40", "--frames", "5", "--start", "1"]
    want, _ = dump_of(tmp_path, "single.bin", *grid)
    got, out = dump_of(tmp_path, "ring.bin", *grid, "--ring")
    assert "RCCL" in out and "communicator of 1" in out
    assert got == want
    # the direct transport with a world of one: own boundary planes into own halo planes, RCCL (communicator of one) all-reduce
    got, out = dump_of(tmp_path, "ringipc.bin", *grid, "--ring", "--transport", "ipc")
    assert "direct halos" in out and "RCCL" in out
    assert got == want


def test_tau3d_rccl_needs_one_device_per_rank(eng, tmp_path):
    import fluid_sims_amd as f
    n = __import__("ctypes").c_int()
    f.load().tau_device_count(__import__("ctypes").byref(n))
    if n.value >= 2:
        pytest.skip("this box has several devices")
    r = run(TAU3D, "--n", "32", "--frames", "1", "--gpus", "2")
    assert r.returncode != 0
    assert "needs 2 devices" in r.stderr


@pytest.mark.parametrize("transport", ["rccl", "local", "ipc"])
def test_python_binding_ring_world1(eng, transport):
    """Tau3DRing (the ctypes mirror of tau3d_ring_*) with a world of one: RCCL to itself, and plain device copies"""
    nx, ny, nz, steps = 48, 40, 36, 7
    ref = eng.Tau3D(nx, ny, nz)
    ref.init(1)
    ref.set_clock(0.02, 1e-4)
    ref.step(steps)
    want, wc = ref.download(), ref.clock()
    ref.close()
    e = eng.Tau3D(nx, ny, nz)
    e.init(1)
    e.set_clock(0.02, 1e-4)
    ring = eng.Tau3DRing(e, 0, 1, {"rccl": eng.RING_RCCL, "local": eng.RING_LOCAL, "ipc": eng.RING_IPC}[transport])
    ring.step(steps)
    ring.finish()
    c = ring.clock()
    got = e.download()
    info = ring.info()
    ring.close()
    e.close()
    if transport in ("rccl", "ipc"):
        assert info["rccl_version"] > 0 and info["comm_ranks"] == 1 and "rccl" in info["librccl"]
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert (c.t, c.d_tau, c.maxs, c.step) == (wc.t, wc.d_tau, wc.maxs, wc.step)


def test_ring_speculative_xy_launch_is_repeated_when_the_weight_form_flips(eng):
    """Round 5: the ring issues the x/y fluxes of a step AHEAD of the all-reduced field range (csrc/ring.hip: ring_step_spec).  Here
    the range word that launch reads is poisoned before every step (beyond the fast window -> it takes the reciprocal weight
    form), the clock's commit then puts the true range back and records the flip, and k_flux_xy_fix repeats the fluxes in the fast
    form: the state must come out byte for byte as in the plain step loop.  (Without the repeat the reciprocal-form divergence
    would be used: the two forms agree to ~1e-7, not to the bit — the control below.)"""
    nx, ny, nz, steps = 160, 128, 24, 4          # planes >= 128^2: the kernel pair
    ref = eng.Tau3D(nx, ny, nz)
    ref.init(1)
    ref.set_clock(0.02, 1e-4)
    ref.step(steps)
    want, wc = ref.download(), ref.clock()
    ref.close()

    def ring_run(poison, fix=True):
        if not fix:
            os.environ["TAU3D_DEBUG_NO_XY_FIX"] = "1"
        try:
            e = eng.Tau3D(nx, ny, nz)
            e.init(1)
            e.set_clock(0.02, 1e-4)
            ring = eng.Tau3DRing(e, 0, 1, eng.RING_LOCAL)
            ring.prime()
            for _ in range(steps):
                if poison:
                    ring.finish()
                    e.debug_set_fmax_in(1e30)
                ring.step(1)
            ring.finish()
            c = ring.clock()
            got = e.download()
            ring.close()
            e.close()
            return got, c
        finally:
            os.environ.pop("TAU3D_DEBUG_NO_XY_FIX", None)

    got, c = ring_run(False)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    got, c = ring_run(True)
    assert all(np.array_equal(a, b) for a, b in zip(got, want)), "a repeated x/y launch must leave no trace"
    assert (c.t, c.d_tau, c.maxs, c.step) == (wc.t, wc.d_tau, wc.maxs, wc.step)
    got, c = ring_run(True, fix=False)           # control: the poison does bite when the repeat is switched off
    assert not all(np.array_equal(a, b) for a, b in zip(got, want)), "the poisoned launch was expected to differ without the repeat"
    err = max(float(np.max(np.abs(a - b))) for a, b in zip(got, want))
    assert err < 1e-3, err                      # ... and differs by rounding only (both weight forms are the same scheme)


@pytest.mark.parametrize("env", [{"TAU3D_RING_SPEC": "0"}, {"TAU3D_RING_PIPELINE": "0"}])
def test_tau3d_ring_earlier_schedules_still_bit_identical(eng, tmp_path, env):
    """the round-4 pipelined schedule (all-reduce first) and the round-2 edge / interior schedule stay selectable and exact"""
    grid = ["--nx", "160", "--ny", "128", "--nz", "48", "--frames", "3", "--start", "1"]
    want, _ = dump_of(tmp_path, "single.bin", *grid)
    for transport in ("host", "ipc-host"):
        path = str(tmp_path / f"{transport}.bin")
        r = subprocess.run([TAU3D, *grid, "--gpus", "2", "--transport", transport, "--dump", path], capture_output=True, text=True, cwd=ROOT,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TAU3D_RING_TIMEOUT="60", **env), timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(path, "rb").read() == want, (transport, env)


def test_bench_force_slab_c_ring(eng):
    """bench.py's N > 1 code path on one GPU: the C ring with RCCL to itself, one JSON line"""
    import json
    import sys
    r = run(sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--grid", "128", "--steps", "4", "--warmup", "2",
            "--force-slab", "--self-p2p", "--no-cpu-baseline", "--no-configs", "--no-variants")
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["ring"]["transport"] == "rccl" and j["ring"]["comm_ranks"] == 1 and j["value"] > 0
    assert j["ring"]["matches_single_domain"] is True and j["ring"]["clock_matches"] is True and j["ring"]["differing_planes"] == 0


def test_ring_rendezvous_times_out_instead_of_hanging(eng, tmp_path):
    """a rank whose peers never show up gets an error after TAU3D_RING_TIMEOUT seconds (here: rank 1 of 2 waiting for rank 0's
    rendezvous file, then rank 0 of 2 waiting at the start barrier) — never a hang inside ncclCommInitRank"""
    import sys
    code = r"""
import sys, ctypes
sys.path.insert(0, %r)
import fluid_sims_amd as f
rank = int(sys.argv[1])
p = f.Tau3DParams(); f.load().tau3d_params_default(ctypes.byref(p), 32, 32, 32)
z0, nzl = f.slab_bounds(32, 2, rank)
e = f.Tau3D(32, 32, 32, params=p, z0=z0, nzl=nzl)
e.init(1)
try:
    f.Tau3DRing(e, rank, 2, f.RING_HOST, rendezvous=sys.argv[2], job_key=12345)
except f.TauError as ex:
    print("TauError:", ex); sys.exit(7)
sys.exit(0)
""" % ROOT
    env = dict(os.environ, TAU3D_RING_TIMEOUT="3")
    for rank, words in ((1, "waited"), (0, "timed out")):
        r = subprocess.run([sys.executable, "-c", code, str(rank), str(tmp_path / f"rv{rank}")], capture_output=True, text=True,
                           cwd=ROOT, env=env, timeout=120)
        assert r.returncode == 7 and words in r.stdout, (rank, r.stdout, r.stderr)


@pytest.mark.parametrize("transport,word", [("host", "host-staged"), ("ipc-host", "direct halos")])
def test_bench_two_ranks_end_to_end_on_one_gpu(eng, transport, word):
    """`python bench.py --gpus 2` as the driver calls it: bench.py spawns its own ranks under torch.distributed.run, each rank
    builds its slab and the library ring, rank 0 prints ONE JSON line with n_gpus = 2 and the per-rank kernel times.  On this
    one-GPU box the ranks share the device (--ring-transport host, process group over gloo); with RCCL the same command is
    the measured N = 2 configuration."""
    import json
    import sys
    r = run(sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "4", "--warmup", "2",
            "--ring-transport", transport)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["decomposition"] == "z-slab x2" and j["value"] > 0
    assert len(j["roofline"]["per_rank_kernel_ms_per_step"]) == 2 and all(t > 0 for t in j["roofline"]["per_rank_kernel_ms_per_step"])
    assert word in j["ring"]["transport"]
    # the run proved itself: every rank's slab == its own single-domain recompute, bit for bit, and the per-rank event times are there
    assert j["ring"]["matches_single_domain"] is True and j["ring"]["clock_matches"] is True and j["ring"]["first_differing_plane"] is None
    pr = j["ring"]["per_rank_event_ms_per_step"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["xy_ms"] > 0 and r["z_ms"] > 0 and r["exchange_ms"] > 0 and r["allreduce_ms"] > 0 for r in pr), pr


@pytest.mark.parametrize("ranks,grid", [(2, 128), (8, 128)])
def test_bench_auto_transport_probe_on_one_gpu(eng, ranks, grid):
    """`bench.py --gpus N` with the default --ring-transport auto, the command the driver's scaling run issues: both candidate
    transports are built, timed over a few warm-up steps and closed again, the faster one is rebuilt for the timed region, and
    after it every rank checks its slab against its own single-domain recompute (ring.matches_single_domain).  On this one-GPU box
    the candidates are the two that let ranks share a device (TAU_BENCH_AUTO_SHARED: ipc-host against host; with a device per
    rank: ipc against rccl)."""
    import json
    import sys
    env = dict(os.environ, TAU_BENCH_AUTO_SHARED="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TAU3D_RING_TIMEOUT="120")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--grid", str(grid), "--steps", "4", "--warmup", "2"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    probes = j["ring"]["auto_probe_ms_per_step"]
    assert j["n_gpus"] == ranks and len(probes) == 2 and all(isinstance(v, float) and v > 0 for v in probes.values()), probes
    fastest = min(probes, key=probes.get)
    assert j["ring"]["transport"] == fastest
    assert j["ring"]["matches_single_domain"] is True and j["ring"]["clock_matches"] is True and j["ring"]["differing_planes"] == 0
    assert len(j["ring"]["per_rank_event_ms_per_step"]) == ranks


def test_bench_verification_catches_a_single_flipped_bit(eng):
    """the check can fail: TAU_BENCH_VERIFY_SELFTEST flips one bit of one word of the last rank's downloaded slab before the
    comparison -> matches_single_domain false, the plane named, exit code 4 (and the ring / process group still torn down)"""
    import json
    import sys
    env = dict(os.environ, TAU_BENCH_VERIFY_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TAU3D_RING_TIMEOUT="60")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "3", "--warmup", "1",
                        "--ring-transport", "host"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    # the ranks exit 4; torch.distributed.run (which bench.py --gpus 2 re-launches itself under) reports a failed rank as 1
    assert r.returncode != 0 and "exitcode  : 4" in r.stderr, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["ring"]["matches_single_domain"] is False and j["ring"]["clock_matches"] is True
    assert j["ring"]["first_differing_plane"] == 64 + 32 and j["ring"]["differing_planes"] == 1
    assert "does NOT reproduce" in r.stderr
    # --no-verify skips it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "3", "--warmup", "1",
                        "--ring-transport", "host", "--no-verify"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["ring"]["matches_single_domain"] is None


def test_bench_probe_survives_a_candidate_that_fails_while_stepping(eng):
    """auto: the first candidate builds on every rank and then throws on one rank while stepping (TAU_BENCH_FAIL_PROBE) — it is
    reported unavailable, closed, and the other candidate runs the timed region; the run still verifies"""
    import json
    import sys
    import fluid_sims_amd as f
    env = dict(os.environ, TAU_BENCH_AUTO_SHARED="1", TAU_BENCH_FAIL_PROBE=str(f.RING_IPC_HOSTMAX), HSA_ENABLE_IPC_MODE_LEGACY="0",
               TAU3D_RING_TIMEOUT="20")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    probes = j["ring"]["auto_probe_ms_per_step"]
    bad = [k for k, v in probes.items() if isinstance(v, str)]
    # (rank 0 prints the line: what it saw is its own peer-is-gone error or "another rank failed", not rank 1's exception text)
    assert len(bad) == 1 and "ipc-host" in bad[0] and probes[bad[0]].startswith("unavailable"), probes
    assert "host-staged" in j["ring"]["transport"] and j["ring"]["matches_single_domain"] is True


def test_ring_refuses_a_zero_job_key_and_reports_a_failed_peer(eng, tmp_path):
    """world > 1 needs a non-zero job key (it tells this job's rendezvous file from a stale one); and a rank that finds the file
    of a job whose rank 0 has already given up learns so from the status word at once instead of waiting out a timeout"""
    import sys
    code = r"""
import sys, ctypes
sys.path.insert(0, %r)
import fluid_sims_amd as f
rank, key = int(sys.argv[1]), int(sys.argv[3])
p = f.Tau3DParams(); f.load().tau3d_params_default(ctypes.byref(p), 32, 32, 32)
z0, nzl = f.slab_bounds(32, 2, rank)
e = f.Tau3D(32, 32, 32, params=p, z0=z0, nzl=nzl)
e.init(1)
try:
    f.Tau3DRing(e, rank, 2, f.RING_HOST, rendezvous=sys.argv[2], job_key=key)
except f.TauError as ex:
    print("TauError:", ex); sys.exit(7)
sys.exit(0)
""" % ROOT
    env = dict(os.environ, TAU3D_RING_TIMEOUT="3")
    go = lambda rank, key: subprocess.run([sys.executable, "-c", code, str(rank), str(tmp_path / "rv"), str(key)], capture_output=True,
                                          text=True, cwd=ROOT, env=env, timeout=120)
    r = go(0, 0)
    assert r.returncode == 7 and "non-zero job key" in r.stdout, (r.stdout, r.stderr)
    r = go(0, 777)                       # rank 0 alone: times out at the start barrier, says so in the file and leaves it for its peers
    assert r.returncode == 7 and "timed out" in r.stdout, (r.stdout, r.stderr)
    assert os.path.exists(tmp_path / "rv")
    import time
    # A rank 1 that starts AFTER that rank 0 gave up cannot tell its file from one an earlier launch left under the same key
    # (round-4 advice: such a leftover made the ranks of the NEXT launch fail with "rank 0 failed"): it keeps waiting for a
    # fresh file, and says what it saw when it gives up.
    r = go(1, 777)
    assert r.returncode == 7 and "waited" in r.stdout and "rank 0 had failed" in r.stdout, (r.stdout, r.stderr)
    # ... and the next launch under the same key is not poisoned by the leftover: both ranks come up (rank 0 removes it first)
    import threading
    res = {}
    th = [threading.Thread(target=lambda k=k: res.__setitem__(k, go(k, 777))) for k in (1, 0)]
    th[0].start(); time.sleep(1.0); th[1].start()      # rank 1 first: it finds the leftover, refuses it, then maps the new file
    for t in th:
        t.join()
    assert res[0].returncode == 0 and res[1].returncode == 0, (res[0].stdout, res[1].stdout)
    r = go(1, 778)                       # another job's key under the same path: that file is not this job's -> waits, then gives up
    assert r.returncode == 7 and "waited" in r.stdout, (r.stdout, r.stderr)


def test_bench_falls_back_to_the_torch_ring_when_the_c_ring_cannot_be_built(eng):
    """every rank agrees (all-reduce of a flag) before the C ring is used; if one cannot build it, all of them run the
    torch.distributed ring of rounds 1-2 instead and the JSON line says so"""
    import json
    import sys
    env = dict(os.environ, TAU_BENCH_FAIL_C_RING="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "3", "--warmup", "1",
                        "--ring-transport", "host"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["value"] > 0
    assert "python" in j["ring"]["driver"] and "TAU_BENCH_FAIL_C_RING" in j["ring"]["fallback_from_c_ring"]
