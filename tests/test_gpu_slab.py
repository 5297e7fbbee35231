"""GPU: the slab driver on the HIP engine — the self-neighbour (world = 1) ring must equal the
single-domain entry point bit for bit, and the aliased halo / max tensors must be usable by RCCL."""
import ctypes
import os
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(eng, n):
    p = eng.Tau3DParams()
    eng.load().tau3d_params_default(ctypes.byref(p), *n)
    return p


def test_ring_world1_equals_step(eng):
    import torch
    slab = import_module("fluid_sims_amd.slab")
    n = (64, 32, 48)
    ref = eng.Tau3D(*n)
    ref.init(1)
    ref.set_clock(0.02, 1e-4)
    c_ref = ref.step(9)
    want = ref.download()

    be = slab.EngineSlabBackend(eng.taueng, _params(eng, n), 0, n[2], 0)
    be.h.init(1)
    be.h.set_clock(0.02, 1e-4)
    ring = slab.SlabRing(be, 0, 1)
    ring.prime()
    ring.step(9)
    ring.finish()
    got = be.h.download()
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    c = be.clock()
    assert (c.t, c.d_tau, c.maxs, c.step) == (c_ref.t, c_ref.d_tau, c_ref.maxs, c_ref.step)
    # the packed buffers alias engine memory: pack the current state and read it back through torch
    be.pack(0)
    be.sync()
    lo = be.buf("send", 0).cpu().numpy().reshape(6, 3, n[1], n[0])
    hi = be.buf("send", 1).cpu().numpy().reshape(6, 3, n[1], n[0])
    for f in range(6):
        assert np.array_equal(lo[f], got[f][0:3]) and np.array_equal(hi[f], got[f][n[2] - 3:])
    torch.cuda.synchronize()


def test_rccl_on_aliased_tensors(eng):
    """world-size-1 RCCL group: all_reduce(MAX) on the engine's max word and a broadcast of a halo
    tensor run on the aliased device memory (the N > 1 path uses exactly these tensors)."""
    import torch
    import torch.distributed as dist
    slab = import_module("fluid_sims_amd.slab")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = (32, 32, 32)
        be = slab.EngineSlabBackend(eng.taueng, _params(eng, n), 0, 32, 0)
        be.h.init(1)
        be.h.set_clock(0.02, 1e-4)
        be.begin()
        be.edges(8)
        be.interior(8)
        m = be.max_tensor()                      # [max wavespeed, max |primitive|] of the step in flight
        assert m.shape == (2,)
        before, fbefore = (float(v) for v in m.tolist())
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        dist.broadcast(be.buf("send", 0), src=0)
        torch.cuda.synchronize()
        assert before > 0 and fbefore >= 99.9 and m.tolist() == [before, fbefore]   # inflow u = 100
        be.end()
        be.sync()
        assert be.clock().maxs == before
    finally:
        dist.destroy_process_group()


def test_multigpu_driver_script_world1(eng, tmp_path):
    """scripts/tau3d_multigpu.py with one process (the ring is its own neighbour): same state as bin/tau3d's loop"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "r0.bin"
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "tau3d_multigpu.py"), "--n", "32", "--frames", "3",
                        "--steps-per-frame", "2", "--dump-rank0", str(out)], capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "6 steps on 32x32x32 over 1 GPU(s)" in r.stdout and "step 6" in r.stdout
    got = np.fromfile(out, np.float32).reshape(6, 32, 32, 32)
    e = eng.Tau3D(32)
    e.init(0)
    e.step(6)
    for g, w in zip(got, e.download()):
        assert np.array_equal(g, w)
    e.close()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("n", [(64, 48, 40), (160, 128, 24)])     # fused step / split step
def test_ring_over_rccl_send_recv_to_self(eng, n):
    """The N > 1 communication path on one GPU: a world-of-one RCCL group, the ring's batched isend / irecv of the packed halo
    buffers addressed to OURSELVES (low planes -> own high halo and vice versa: what a periodic z means for one slab) and the
    all-reduce of the max words — real ncclSend / ncclRecv kernels on the aliased engine memory, ordered against the step
    kernels by torch's stream semantics.  Bit-identical to the single-domain entry point."""
    import torch
    import torch.distributed as dist
    slab = import_module("fluid_sims_amd.slab")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29579"
    ref = eng.Tau3D(*n)
    ref.init(1)
    ref.set_clock(0.02, 1e-4)
    c_ref = ref.step(7)
    want = ref.download()
    ref.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        be = slab.EngineSlabBackend(eng.taueng, _params(eng, n), 0, n[2], 0)
        be.h.init(1)
        be.h.set_clock(0.02, 1e-4)
        ring = slab.SlabRing(be, 0, 1, self_p2p=True)
        ring.step(7)                      # primes itself: pack, exchange, unpack, all-reduce of the field range
        ring.finish()
        got = be.h.download()
        c = be.clock()
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        assert (c.t, c.d_tau, c.maxs, c.step) == (c_ref.t, c_ref.d_tau, c_ref.maxs, c_ref.step)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
