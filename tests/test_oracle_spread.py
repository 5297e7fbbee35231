"""How far apart do two LEGAL fp32 evaluations of the reference's own arithmetic lie?  (CPU only.)

tests/parity.py compares `lam = ln p` and `zet = ln e_vib` at 1e-5 * kappa, kappa = (gamma-1) E / p, instead of the literal
1e-5, and tests/test_gpu_sph.py compares `acc` against the sum of |pair terms|: both on the argument that the reference's
formulation itself cancels there (p is the difference of two O(E) numbers at Mach 100; the SPH force is a cancelling
sum).  VERDICT r01 asked for that to be demonstrated rather than asserted.  These tests build the ORACLE twice —

  3D step : IEEE (-ffp-contract=off, the oracle every parity test uses)   vs   -ffp-contract=fast -mfma
            (how the reference's GPU build rounds: nvcc contracts by default)
  SPH     : cell lists walked in descending particle order (the emulated reference)   vs   ascending order
            (the engine's order; a GPU's atomicExch order is arbitrary)

— and measure the spread between the two on one step from the same developed state: it exceeds the literal 1e-5 in the
conditioned quantities, by the factor the scaled tolerances allow, and stays far inside 1e-5 everywhere else.  The GPU
tests then require the engine's error to be of the size of this spread (test_gpu_tau3d.py, test_gpu_sph.py)."""
import numpy as np

from tests import parity


def developed_3d(oracle_built, n=32, steps=40):
    o = oracle_built.Oracle3D(n)
    st = o.init(1)
    o.clock.t = 0.02
    o.clock.d_tau = 1e-4
    st = o.run(st, steps)
    o.fill_halo_periodic(st)
    return o, st


def two_build_spread_3d(oracle_built, o, st, dt, gain=1.0):
    """(IEEE result, FMA-contracted result, fluid mask) of ONE step from `st` (halo layout, halos current)"""
    p = o.p
    b = oracle_built.Oracle3D(p.nx, p.ny, p.nz, lib="libtauoracle3d_fma.so")
    oa, ob = o.new_state(), b.new_state()
    o.step_range(st, oa, dt, gain)
    b.step_range(st, ob, dt, gain)
    return o.interior(oa), o.interior(ob), o.interior([o.solid])[0] == 0


def test_3d_lam_zet_spread_between_two_builds_of_the_oracle(oracle_built):
    o, st = developed_3d(oracle_built)
    ia, ib, fluid = two_build_spread_3d(oracle_built, o, st, 2e-6)
    r = parity.report(ib, ia, fluid)
    print("IEEE vs FMA oracle, 32^3 developed:", {k: f"{v:.2e}" for k, v in r.items()})
    # well-conditioned quantities: two legal evaluations agree far inside the 1e-5 contract
    for k in ("xi", "phix", "phiy", "phiz", "rho", "mx", "my", "mz", "E"):
        assert r[k] <= 2e-6, (k, r[k])
    # lam = ln p: the two builds of the REFERENCE'S OWN arithmetic differ by more than the literal 1e-5 ...
    kap = parity.kappa(ia)
    d_lam = np.abs(ib[4].astype(np.float64) - ia[4])[fluid]
    assert d_lam.max() > 5e-5 and int((d_lam > 1e-5).sum()) > 100, d_lam.max()
    # ... by the conditioning factor, and by no more: spread / kappa is an fp32 rounding
    assert (d_lam / kap[fluid]).max() <= 1e-6
    assert kap[fluid].max() > 400            # Mach 100 free stream: E / (p / (gamma - 1)) ~ 500
    # zet inherits part of it through e_eq(T(p)) over one relaxation step
    d_zet = np.abs(ib[5].astype(np.float64) - ia[5])[fluid]
    assert 1e-6 < d_zet.max() and (d_zet / kap[fluid]).max() <= 1e-6


def sph_pair(oracle_built, N=4096, steps=300):
    """two SPH oracles on the same state `steps` sub-steps into the dam break: list order descending / ascending"""
    a = oracle_built.OracleSph(N)
    b = oracle_built.OracleSph(N, lib="libtauoraclesph_asc.so")
    a.step(steps)
    s = a.state()
    b.set_state(s["pos"], s["vel"])
    a.set_state(s["pos"], s["vel"])
    dt = a.dt()
    a.substep(dt)
    b.substep(dt)
    sa = a.state()
    return sa, b.state(), sa["acc_abs"]


def test_sph_acc_spread_between_the_two_summation_orders(oracle_built):
    sa, sb, accabs = sph_pair(oracle_built)
    # integer results and well-conditioned sums agree to rounding
    rho_a, rho_b = np.exp(sa["s"].astype(np.float64)), np.exp(sb["s"].astype(np.float64))
    assert (np.abs(rho_a - rho_b) / rho_a).max() <= 1e-5      # (4e-6 at ~20 rho0, ~1000 neighbours per particle)
    d = np.abs(sa["acc"].astype(np.float64) - sb["acc"]).max(axis=1)
    mag = np.linalg.norm(sa["acc"].astype(np.float64), axis=1)
    rel_to_acc = d / np.maximum(mag, 1e-30)
    rel_to_sum = d / np.maximum(accabs.astype(np.float64), 1e-30)
    print(f"SPH order spread: max |d acc| / |acc| = {rel_to_acc.max():.2e}, / sum|pair terms| = {rel_to_sum.max():.2e}")
    # relative to the net force the two legal orders differ by far more than 1e-5 once the column has settled and the
    # net force is what is left of the cancelling pair terms (6 steps into the run it is still 3e-6) ...
    assert rel_to_acc.max() > 1e-4 and int((rel_to_acc > 1e-5).sum()) > 100
    # ... relative to what was summed they agree to a few fp32 roundings
    assert rel_to_sum.max() <= 2e-6
