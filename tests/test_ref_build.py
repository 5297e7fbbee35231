"""CPU: oracle/_ref — the reference's own sources built for gfx950 by oracle/build_ref.sh — holds what the GPU tests launch.
(No compute here: the code objects are only inspected.  Skipped where the build has not run, i.e. outside the build container.)"""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")

WANT = {
    "th3cs": ["k_build_solid_mask(", "k_init(", "k_step(", "P"],
    "th3cs.ieee": ["k_build_solid_mask(", "k_step("],
    "tau_hypersonic_cuda_tests": ["k_init(", "k_apply_inflow_left(", "k_max_wavespeed_blocks(", "k_reduce_block_max(", "k_predict_face_states(",
                                  "k_compute_xface_flux(", "k_compute_yface_flux(", "k_step(", "d_cfg"],
    "tau_gray_scott": ["step_kernel("], "tau_gray_scott.ieee": ["step_kernel("],
    "tau_sph": ["k_clear_heads(", "k_build_cells(", "k_density_pressure_cell(", "k_forces_cell(", "k_integrate("],
    "tau_sph.ieee": ["k_build_cells("],
    "tau_lbm.ieee": ["init_kernel(", "collide_stream_kernel("],
    "tau_burgers": ["viscosity_step("], "tau_shallow_water": ["viscosity_uv("],
}


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "MANIFEST")), reason="oracle/_ref has not been built (oracle/build_ref.sh needs /root/reference)")
def test_reference_objects_hold_the_kernels_the_gpu_tests_launch():
    manifest = open(os.path.join(REF, "MANIFEST")).read().split()
    for name, kernels in WANT.items():
        assert name + ".co" in manifest
        co = open(os.path.join(REF, name + ".co"), "rb").read(20)
        assert co[:4] == b"\x7fELF" and co[18:20] == (224).to_bytes(2, "little"), f"{name}.co is not an AMDGPU code object"
        syms = [l.rstrip("\n").split("\t")[1] for l in open(os.path.join(REF, name + ".syms"))]
        for k in kernels:
            assert any(s == k or s.startswith(k) for s in syms), (name, k)
    for prog in ("tgs", "tau_sph", "tau_lbm", "tau_burgers", "tau_sw", "tau_hypersonic_cuda_tests"):
        p = os.path.join(REF, "bin", prog)
        assert os.path.exists(p) and os.access(p, os.X_OK)


def test_the_recipe_writes_nothing_for_the_reference():
    """oracle/build_ref.sh compiles what hipify-perl makes of the reference's files and nothing else: no -I of a stub directory, no
    -include, no header or source of this repo on its compile lines"""
    sh = open(os.path.join(ROOT, "oracle", "build_ref.sh")).read()
    lines = [l for l in sh.splitlines() if "HIPCC" in l and "-c " in l or ("HIPCC" in l and "-o" in l)]
    assert lines
    for l in lines:
        assert " -I" not in l and "-include" not in l and "$HERE" not in l.replace('"$OUT', "").replace("$OUT", "")
