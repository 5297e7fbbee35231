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


def test_guided_chunk_schedule_without_gpu():
    """tau_guided_chunks (host logic of tau::guided_chunks, csrc/tau_common.hip): the chunks tile [0, H) in order, every band of
    H / 8 rows gets the same number of chunks with lengths descending from `remaining x strips / slots` to the minimum, none
    beyond the maximum (the 2D march addresses a chunk's band with 32-bit offsets: 56 rows)"""
    import fluid_sims_amd as f
    for H, W, slots, lmin, lmax in ((4096, 4096, 512, 6, 56), (8192, 8192, 512, 6, 56), (1024, 8192, 512, 6, 56), (3000, 3000, 512, 6, 56),
                                    (8192, 8192, 1024, 12, 64), (37, 128, 512, 6, 56), (5, 64, 512, 6, 56), (1, 8, 512, 6, 56)):
        nstrips = (W + 59) // 60
        t = f.guided_chunks(H, nstrips, slots, lmin, lmax)
        assert t[0] == 0 and t[-1] == H and all(a <= b for a, b in zip(t, t[1:]))
        lens = [b - a for a, b in zip(t, t[1:])]
        assert max(lens) <= lmax + lmin // 2 + 1 and (len(lens) % 8) == 0
        per = len(lens) // 8
        for b in range(8):
            band = [x for x in lens[b * per:(b + 1) * per] if x > 0]
            assert sum(lens[b * per:(b + 1) * per]) == H * (b + 1) // 8 - H * b // 8
            # descending apart from the last chunk of a band (which absorbs a short remainder or is clipped by a shorter band)
            assert all(x >= y for x, y in zip(band[:-1], band[1:-1])), (H, W, band)
        if H >= 2048:
            first = min(lmax, max(lmin, round((H + 7) // 8 * nstrips / slots)))
            assert lens[0] == first
    with pytest.raises(f.TauError):
        f.guided_chunks(0, 4, 512, 6, 56)
