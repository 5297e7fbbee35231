"""CPU: the C-ABI library loads and exports every symbol include/taueng.h declares; the product
path refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "taueng.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tau[a-z0-9]*_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    import fluid_sims_amd as f
    if not os.path.exists(f.lib_path()):
        import __graft_entry__ as g
        g.build()
    L = f.load()   # binds the HIP runtime first (the library has no DT_NEEDED on it)
    names = declared_symbols()
    assert len(names) > 40
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in taueng.h but not exported: {missing}"


def test_binding_covers_header():
    import fluid_sims_amd as f
    L = f.load()
    for n in declared_symbols():
        fn = getattr(L, n)
        if n not in ("tau_last_error",):
            assert fn.argtypes is not None, f"{n}: binding has no signature"


def test_no_cpu_fallback():
    import fluid_sims_amd as f
    L = f.load()
    if L.tau_device_available():
        pytest.skip("a GPU is visible")
    for ctor in (lambda: f.Tau3D(32), lambda: f.GrayScott(64, 64), lambda: f.Laplacian2D(64, 64, "sw", 0.1, 0.1)):
        with pytest.raises(f.TauError):
            ctor()


def test_slab_bounds_without_gpu():
    """tau3d_slab_bounds (the C ring's partition, csrc/ring.hip) against fluid-sims_amd/slab.py's: contiguous, covering, >= 6 planes"""
    import fluid_sims_amd as f
    from importlib import import_module
    slab = import_module("fluid_sims_amd.slab")
    for nz, world in ((512, 8), (64, 3), (50, 4), (48, 8), (13, 2)):
        got = [f.slab_bounds(nz, world, r) for r in range(world)]
        assert got == [slab.slab_bounds(nz, world, r) for r in range(world)]
        assert got[0][0] == 0 and sum(n for _, n in got) == nz and all(a[0] + a[1] == b[0] for a, b in zip(got, got[1:]))
    with pytest.raises(f.TauError, match="need >= 6"):
        f.slab_bounds(40, 8, 0)
