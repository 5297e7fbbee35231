"""fluid-sims_amd — MI355X-native explicit time-stepping engine (libtaueng) and its host mirror.

The directory name carries a hyphen (it is the name the project was given); import it through
the ``fluid_sims_amd`` shim at the repository root, or load this package by path.
"""
from . import taueng  # noqa: F401
from .taueng import (  # noqa: F401
    TauError, lib_path, load, Tau3D, Tau3DRing, slab_bounds, guided_chunks, RING_RCCL, RING_HOST, RING_LOCAL, RING_IPC, RING_IPC_HOSTMAX, Hypersonic2D, Sph2D, Flow2D, Lbm2D, GrayScott, Laplacian2D, Tau3DParams, Tau3DClock, RowRing, row_bounds, gs_pattern_host,
)
