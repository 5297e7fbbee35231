"""CPU: oracle/_ref — the reference's own sources built for gfx950 by oracle/build_ref.sh — holds what the GPU tests launch.
(No compute here: the code objects are only inspected.  Skipped where the build has not run, i.e. outside the build container.)"""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")

WANT = {
    # the file north_star names, from SURVEY 8c's line cut (1-1409 minus raylib includes and Vector3 helpers)
    "tau_hypersonic_3d_cuda": ["k_build_solid_mask(", "k_init(", "k_step(", "k_vis(", "k_maxwavespeed_pre(", "k_schlieren(",
                               "k_outflow_reflection_metric(", "P"],
    "tau_hypersonic_3d_cuda.ieee": ["k_build_solid_mask(", "k_step("],
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
    for lib in ("libref_hostmaps.so", "libref_hyp_cpu_300x300.so", "libref_hyp_cpu_256x256.so", "libref_hyp_cpu_96x64.so",
                "libref_hyp_cpu_simd_300x300.so", "libref_hyp_cpu_simd_256x256.so", "libref_hyp_cpu_simd_96x64.so"):
        assert lib in manifest and open(os.path.join(REF, lib), "rb").read(4) == b"\x7fELF"
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


def test_the_line_cuts_are_cuts_not_rewrites():
    """the files that include raylib.h are built from `sed -n 'A,Bp'` ranges of the reference's own text with lines DELETED
    (`sed '/raylib.h/d'`, `-e '4,5d' -e '69,101d'`) — the only substitution anywhere is the survey's W/H #define patch — and the
    harnesses #include the cut rather than paste anything in its place"""
    sh = open(os.path.join(ROOT, "oracle", "build_ref.sh")).read()
    assert "sed -n '1,674p' \"$REF/tau_hypersonic.c\"" in sh and "sed -n '1,804p' \"$REF/tau_hypersonic_simd.c\"" in sh
    assert "sed -n '1,1409p' \"$REF/tau_hypersonic_3d_cuda.cu\" | sed -e '4,5d' -e '69,101d'" in sh
    subs = [l for l in sh.splitlines() if "sed" in l and "s/" in l and not l.lstrip().startswith("#")]
    assert len(subs) == 1 and "#define W 300" in subs[0] and "#define H 300" in subs[0], subs
    assert "typedef struct" not in sh and "InitWindow" not in sh      # no stand-in declarations
