"""GPU: error behaviour of the C-ABI — every failure is a non-zero status + tau_last_error() (raised as TauError
by the ctypes mirror), nothing aborts, and a failed create leaves nothing behind (the next create succeeds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_out_of_memory_create_is_clean(eng):
    for make in (lambda: eng.Tau3D(4096, 4096, 4096),            # 12 x 275 GB
                 lambda: eng.Lbm2D(1 << 17, 1 << 17),            # 2 x 618 GB
                 lambda: eng.Hypersonic2D(1 << 18, 1 << 17)):
        with pytest.raises(eng.TauError, match="hipMalloc|memory|out of"):
            make()
    e = eng.Tau3D(32)                                            # the device is still usable
    e.init(0)
    e.step(2)
    assert np.isfinite(e.download()[0]).all()
    e.close()


def test_bad_arguments_report_instead_of_aborting(eng):
    with pytest.raises(eng.TauError, match="at least 8"):
        eng.Tau3D(4)
    with pytest.raises(eng.TauError, match="slab"):
        eng.Tau3D(32, 32, 32, z0=30, nzl=8)
    with pytest.raises(eng.TauError, match="tau must exceed 0.5"):
        eng.Lbm2D(64, 64, tau=0.4)
    with pytest.raises(eng.TauError, match="positive"):
        eng.Sph2D(0)
    with pytest.raises(eng.TauError, match="2x2"):
        eng.GrayScott(1, 64)
    e = eng.Tau3D(32)
    e.init(1)
    with pytest.raises(eng.TauError, match="no visualisation field"):
        e.slice_rgba(0)
    with pytest.raises(eng.TauError, match="outside 0..7"):
        e.vis(8)
    with pytest.raises(eng.TauError, match="bad plane range"):
        e.step_range_async(5, 5)
    e.vis(0)
    e.slice_rgba(3)
    e.close()
    h = eng.Hypersonic2D(64, 64)
    h.init()
    with pytest.raises(eng.TauError, match="outside 0..6"):
        h.render(7)
    with pytest.raises(eng.TauError, match="positive"):
        h.step_explicit(0.0)
    h.close()
    s = eng.Tau3D(32, 32, 32, z0=0, nzl=16)
    with pytest.raises(eng.TauError, match="single-domain call on a slab handle"):
        s.step(1)
    s.close()


def test_order_of_calls_is_checked(eng):
    e = eng.Tau3D(32)
    e.init(0)
    with pytest.raises(eng.TauError, match="call tau3d_vis first"):
        e.palette_indices()
    with pytest.raises(eng.TauError, match="call tau3d_vis first"):
        e.slice_rgba(3)
    e.vis(0)
    idx, mn, mx = e.palette_indices()
    assert idx.shape == (32, 32, 32) and mn <= mx
    e.close()
    from importlib import import_module
    slab2d = import_module("fluid_sims_amd.slab2d")
    be = slab2d.EngineRowBackend(lambda ny, s: eng.GrayScott(64, ny, stream=s), 64, 2, 4, 0)
    with pytest.raises(ValueError, match="thinner than the halo"):
        slab2d.RowRing(be, 0, 1)
