#!/usr/bin/env python3
"""instr_budget.py [--sq-xy N --sq-z N] — per-phase VALU instruction budget of the 3D step's kernel pair.

Each phase function of fluid-sims_amd/csrc/h3d.hip (decode, the cell-centred WENO, the one-sided ring form, HLLC per axis,
the update, the encode) is wrapped in a probe kernel that loads its operands, calls it, and stores its results; the probe is
compiled with the flags of the split translation unit (Makefile: build/h3d_split.o) and its VALU instructions are counted by
issue class (scripts/isa_mix.py: full / half / transcendental, cycles from profiles/r02/valu_calib.txt).  A probe that only
moves the same operands gives the addressing overhead, which is subtracted.  Unit cost x the number of times a cell pays for
the phase (tile geometry of k_flux_xy / chunk geometry of k_update_z, stated below) gives the phase's share; what is left of
the executed count (SQ_INSTS_VALU / cells / 64 from profiles/r0x/pmc_sq.txt, given on the command line) is staging,
addressing, selects and control.

Static counts of straight-line code (HLLC's supersonic early returns are counted at weight 1: the full path, which is what
a subsonic face executes; a face that returns early executes less — the free stream's x faces)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from isa_mix import classify  # noqa: E402

CSRC = os.path.join(HERE, "..", "fluid-sims_amd", "csrc")

PROBES = r'''
#define TAU3D_SPLIT_TU
#define TAU_EXPERIMENT
#define TAU3D_FAST_ONLY
#include "h3d.hip"
namespace h3d {
#define LD(p, i) (p)[(size_t)(i) * n + t]
#define PROBE_HEAD const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
// operands in, results out, nothing in between: the addressing / load / store overhead of a probe with NI loads and NO stores
template <int NI, int NO> __global__ void p_move(const float *__restrict__ in, float *__restrict__ out, size_t n) {
  PROBE_HEAD
  float v[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) v[i] = LD(in, i);
#pragma unroll
  for (int o = 0; o < NO; o++) LD(out, o) = v[o % NI];
}
template __global__ void p_move<6, 6>(const float *, float *, size_t);
template __global__ void p_move<5, 2>(const float *, float *, size_t);
template __global__ void p_move<5, 1>(const float *, float *, size_t);
template __global__ void p_move<12, 6>(const float *, float *, size_t);
template __global__ void p_move<24, 8>(const float *, float *, size_t);

__global__ void p_decode(const Args A, const float *__restrict__ in, float *__restrict__ out, size_t n) {
  PROBE_HEAD
  const float uref = vreg(A.u_ref);
#pragma unroll
  for (int m = 0; m < 6; m++) LD(out, m) = decode_field(uref, m, LD(in, m));
}
template <bool FAST> __global__ void p_weno_cell(const float *__restrict__ in, float *__restrict__ out, size_t n) {
  PROBE_HEAD
  float L, R;
  weno_cell<FAST>(LD(in, 0), LD(in, 1), LD(in, 2), LD(in, 3), LD(in, 4), L, R);
  LD(out, 0) = L; LD(out, 1) = R;
}
template __global__ void p_weno_cell<true>(const float *, float *, size_t);
template __global__ void p_weno_cell<false>(const float *, float *, size_t);
template <bool FAST> __global__ void p_weno_side(const float *__restrict__ in, float *__restrict__ out, size_t n) {
  PROBE_HEAD
  LD(out, 0) = weno_cell_side<FAST, true>(LD(in, 0), LD(in, 1), LD(in, 2), LD(in, 3), LD(in, 4));
}
template __global__ void p_weno_side<true>(const float *, float *, size_t);
template <int AXIS> __global__ void p_hllc(const Args A, const float *__restrict__ in, float *__restrict__ out, size_t n) {
  PROBE_HEAD
  const Gas G = gas_vgpr(A);
  Prim L, R;
#pragma unroll
  for (int m = 0; m < 6; m++) { L.q[m] = LD(in, m); R.q[m] = LD(in, 6 + m); }
  prim_floor(L);
  prim_floor(R);
  const Cons F = hllc(G, L, R, AXIS);
#pragma unroll
  for (int m = 0; m < 6; m++) LD(out, m) = F.c[m];
}
template __global__ void p_hllc<0>(const Args, const float *, float *, size_t);
template __global__ void p_hllc<1>(const Args, const float *, float *, size_t);
template __global__ void p_hllc<2>(const Args, const float *, float *, size_t);
__global__ void p_update(const Args A, const float *__restrict__ in, float *__restrict__ out, size_t n) {
  PROBE_HEAD
  const Gas G = gas_vgpr(A);
  const UpdK K = updk_vgpr(A, G);
  const float dt = vreg(A.clk->dt), inv_dz = vreg(A.inv_dz);
  float own[6], D[6], lo[6], hi[6], E[6];
#pragma unroll
  for (int m = 0; m < 6; m++) { own[m] = LD(in, m); D[m] = LD(in, 6 + m); lo[m] = LD(in, 12 + m); hi[m] = LD(in, 18 + m); }
  float smax = 0.f, fmx = 0.f;
  update_cell(A, K, own, D, lo, hi, dt, inv_dz, A.clk->gain, (int)(t & 511), E, smax, fmx);
#pragma unroll
  for (int m = 0; m < 6; m++) LD(out, m) = E[m];
  LD(out, 6) = smax; LD(out, 7) = fmx;
}
}  // namespace h3d
'''

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-ffp-contract=on", "-mllvm",
         "-amdgpu-sched-strategy=max-ilp", "-S", "--cuda-device-only", "-Wno-everything"]
CYC = {"full": 2.3, "half": 4.5, "trans": 10.0, "lane": 4.5}


def kernel_counts(asm):
    """{demangled-ish kernel name: Counter(class -> count)} for every probe kernel in the assembly text"""
    lines = asm.split("\n")
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN3h3d\S+):", l)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        c = Counter()
        for b in lines[i:end]:
            k = classify(b)
            if k and (k in ("full", "trans", "lane") or k.startswith("half")):
                c["half" if k.startswith("half") else k] += 1
        out[name] = c
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sq-xy", type=float, default=None, help="executed VALU instructions per cell of k_flux_xy (SQ_INSTS_VALU)")
    ap.add_argument("--sq-z", type=float, default=None, help="the same for k_update_z")
    ap.add_argument("--defs", default="", help="extra -D flags for the compile (a variant under test)")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.hip")
        open(src, "w").write(PROBES)
        s = os.path.join(td, "probe.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + a.defs.split() + ["-I", CSRC, src, "-o", s], check=True)
        kc = kernel_counts(open(s).read())

    def get(sub):
        ks = [k for k in kc if sub in k]
        assert len(ks) == 1, (sub, ks)
        return kc[ks[0]]

    def net(sub, move):
        c = get(sub).copy()
        c.subtract(get(move))
        return c

    units = {
        "decode (6 fields)": net("p_decode", "p_move<6, 6>"),
        "weno_cell<fast> (both edge states of a cell, one variable)": net("p_weno_cell<true>", "p_move<5, 2>"),
        "weno_cell<reciprocal>": net("p_weno_cell<false>", "p_move<5, 2>"),
        "weno ring form <fast> (one edge state)": net("p_weno_side<true>", "p_move<5, 1>"),
        "prim_floor x2 + hllc, x face": net("p_hllc<0>", "p_move<12, 6>"),
        "prim_floor x2 + hllc, y face": net("p_hllc<1>", "p_move<12, 6>"),
        "prim_floor x2 + hllc, z face": net("p_hllc<2>", "p_move<12, 6>"),
        "update_cell (update, repairs, Landau-Teller, sponges, maxima, encode)": net("p_update", "p_move<24, 8>"),
    }

    def fmt(c, mult=1.0):
        tot = sum(c.values())
        cyc = sum(CYC[k] * v for k, v in c.items())
        return f"{tot * mult:8.1f}  (full {c['full'] * mult:7.1f}  half {c['half'] * mult:6.1f}  trans {c['trans'] * mult:5.1f})  ~{cyc * mult:7.0f} cycles"

    print("unit costs (VALU instructions of one call, probe minus its operand moves)")
    for k, c in units.items():
        print(f"  {k:75s} {fmt(c)}")

    # multiplicities per cell, 512^3 (k_flux_xy: 32 x 16 tile of one plane, 3-cell x / y halo without corners, ring of 2 x 48
    # cells x 6 variables; k_update_z: 64-plane chunks, 6 extra planes decoded per chunk, one extra z face per chunk)
    XT, YT, H = 32, 16, 3
    cells = XT * YT
    halo = 2 * H * (XT + YT)
    ring = 2 * (XT + YT)
    far = XT + YT
    ZC = 64
    wc = units["weno_cell<fast> (both edge states of a cell, one variable)"]
    rows_xy = [
        ("decode: own cell + halo cells", units["decode (6 fields)"], (cells + halo) / cells),
        ("WENO x (6 variables)", wc, 6.0),
        ("WENO y (6 variables)", wc, 6.0),
        ("WENO ring cells (one lane per ring cell and variable)", units["weno ring form <fast> (one edge state)"], ring * 6 / cells),
        ("x face: floors + HLLC", units["prim_floor x2 + hllc, x face"], 1.0),
        ("y face: floors + HLLC", units["prim_floor x2 + hllc, y face"], 1.0),
        ("far faces (one lane per face)", units["prim_floor x2 + hllc, x face"], far / cells),
    ]
    rows_z = [
        ("decode: plane z+4 (+ 6 planes per 64-plane chunk)", units["decode (6 fields)"], (ZC + 6) / ZC),
        ("WENO z (6 variables; + 1 cell + 1 side per chunk)", wc, 6.0 * (ZC + 2) / ZC),
        ("z face: floors + HLLC (+ 1 per chunk)", units["prim_floor x2 + hllc, z face"], (ZC + 1) / ZC),
        ("update_cell", units["update_cell (update, repairs, Landau-Teller, sponges, maxima, encode)"], 1.0),
    ]
    for title, rows, sq in (("k_flux_xy", rows_xy, a.sq_xy), ("k_update_z", rows_z, a.sq_z)):
        print(f"\n{title}: per cell = unit cost x multiplicity")
        tot = Counter()
        for name, c, mult in rows:
            print(f"  {name:58s} x{mult:6.3f} {fmt(c, mult)}")
            for k, v in c.items():
                tot[k] += v * mult
        n = sum(tot.values())
        print(f"  {'sum of the phases':58s}         {fmt(tot)}")
        if sq:
            print(f"  executed (SQ_INSTS_VALU / cell): {sq:.0f}  =  sum of the phases {sq - n:+.0f} ({100 * (sq - n) / sq:+.0f} %)")
            print("    (the static sum is an UPPER bound of the arithmetic — it takes every wave-uniform branch: both sinh forms of each decoded\n"
                  "     velocity [fsinh_wave runs one], the entropy fix [skipped away from sonic lines], HLLC's star-state path behind the supersonic\n"
                  "     return [the free stream's x faces leave at it], tiles inside the body — and a LOWER bound of the rest: staging, addressing,\n"
                  "     selects, the divergence and control are not in it.  Executed below the sum = the branches pay more than the rest costs.)")


if __name__ == "__main__":
    main()
