#!/usr/bin/env python
"""bench.py — headline benchmark: Gcell-updates/s of the 3D hypersonic step on a 512^3 fp32 grid.

  python bench.py --gpus N --steps K --warmup W          (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one full time step of the reference loop (tau_hypersonic_3d_cuda.cu:1680-1711: log-time
clock, k_step over the whole grid, d_tau controller, swap) on synthetic input already resident in
HBM: the SURVEY §8d "developed flow" start (every fluid cell at the Mach-100 inflow state, t = 0.02 so
the inflow gain is 1, sphere r = 0.25 at the centre), W warm-up steps so the bow shock exists and the
controller has settled, then exactly K timed steps bracketed by barrier + device sync.

N > 1: the SAME 512^3 grid is Z-slab partitioned over the ranks (strong scaling, as BASELINE.json's
metric states: "512^3 at 1/2/4/8 MI355X"); per step each rank exchanges 3 boundary planes x 6 fields
with both ring neighbours (RCCL send/recv over xGMI, overlapped with the interior planes) and one
8-byte all-reduce(max) (max wavespeed + max |primitive|) feeds the device-side d_tau controller.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`; at N = 1 the
line also carries `configs`: the other four BASELINE.json configurations (2D Euler 4096^2, Gray-Scott 8192^2,
SPH 4 194 304 particles, the CPU program at 256^2), each with its own event-timed roofline.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CELL = 49.0     # 6 fp32 in + 1 u8 mask + 6 fp32 out
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E nominal (MI355X_MICROARCH.md)


def cpu_baseline(n, state_slab, dt, z0, planes, max_seconds=25.0):
    """Oracle (C restatement of the reference step, 1 thread) timed on a bounded sample of the SAME
    workload: a `planes`-plane Z-slab of the warmed-up 512^3 state.  Reported baseline, not a target."""
    from oracle import pyoracle
    o = pyoracle.Oracle3D(n, n, n, z0=z0, nzl=planes)
    out = o.new_state()
    cells = 0
    t0 = time.perf_counter()
    reps = 0
    while True:
        o.step_range(state_slab, out, dt, 1.0)
        reps += 1
        cells += planes * n * n
        if time.perf_counter() - t0 > max_seconds * 0.5 or reps >= 8:
            break
    el = time.perf_counter() - t0
    return {"value": cells / el / 1e9, "unit": "Gcell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{planes}-plane Z-slab (z0={z0}) of the warmed-up {n}^3 state, {reps} oracle steps, "
                      f"{el:.1f} s, gcc -O2 IEEE fp32, 1 thread (the reference is single-threaded)"}


def cpu_baseline_2d(steps=100, n=300):
    """What north_star names: tau_hypersonic_simd.c (fp64, AVX2 compute_dt, -O3 -mavx2 -mfma) on the host,
    300^2 as shipped, 1 thread — the restated CPU program of BASELINE config 1 (fluid-sims_amd/cpu/)."""
    from importlib import import_module
    m = import_module("fluid_sims_amd.cpu2d")
    s = m.CpuHypersonic2D(n, n, simd=True)
    t0 = time.perf_counter()
    s.step(steps)
    el = time.perf_counter() - t0
    return {"value": n * n * steps / el / 1e9, "unit": "Gcell-updates/s", "cores": 1, "kind": "port",
            "sample": f"tau_hypersonic_simd restated, 2D {n}x{n} fp64, {steps} steps, {el:.2f} s, 1 thread"}


def cpu_baseline_2d_all_cores(steps=60, n=300):
    """SURVEY §8d's optional row: the same single-threaded program as independent replicas on every host core
    (one solver handle per thread; the C step releases the GIL), aggregate rate."""
    import threading
    from importlib import import_module
    m = import_module("fluid_sims_amd.cpu2d")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container's CPU quota is what it can really use, whatever the affinity mask says
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    cores = min(cores, 64)   # bounds the run time when the quota cannot be read
    sims = [m.CpuHypersonic2D(n, n, simd=True) for _ in range(cores)]
    gate = threading.Barrier(cores + 1)

    def work(s):
        gate.wait()
        s.step(steps)

    th = [threading.Thread(target=work, args=(s,)) for s in sims]
    for t in th:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    return {"value": cores * n * n * steps / el / 1e9, "unit": "Gcell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{cores} independent replicas of tau_hypersonic_simd restated (one per core), 2D {n}x{n} fp64, "
                      f"{steps} steps each, {el:.2f} s"}


def input_variants(f, n, steps=6):
    """SURVEY §8d asks for two more inputs beside the headline one: (i) the reference's own quiescent k_init
    start after 50 controller warm-up steps, and a no-body variant that bounds the branch-free throughput."""
    out = {}
    for name, mode, warm, body in (("reference_ic_50_warmup", 0, 50, True), ("developed_no_body", 1, 10, False)):
        p = f.Tau3DParams()
        f.load().tau3d_params_default(ctypes.byref(p), n, n, n)
        if not body:
            p.sdf_r = -1.0
        e = f.Tau3D(n, n, n, params=p)
        e.init(mode)
        if mode:
            e.set_clock(0.02, 1e-4)
        e.step_async(warm)
        e.sync()
        t0 = time.perf_counter()
        e.step_async(steps)
        e.sync()
        el = time.perf_counter() - t0
        out[name] = {"value": round(float(n) ** 3 * steps / el / 1e9, 3), "unit": "Gcell-updates/s", "steps": steps, "warmup": warm}
        e.close()
    return out


def _event_timed(torch, stream, enqueue, sync):
    """ms of `enqueue()` on `stream`, from events recorded ON that stream (the handles launch on it)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync()
    e0.record(stream)
    enqueue()
    e1.record(stream)
    e1.synchronize()
    sync()
    return e0.elapsed_time(e1)


def _roof(kernel, ms_per_launch, units_per_launch, bytes_per_unit, bound, note=None):
    gbs = units_per_launch * bytes_per_unit / (ms_per_launch * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
         "traffic": None, "kernel": kernel, "avg_launch_ms": round(ms_per_launch, 5),
         "algorithmic_bytes_per_launch": units_per_launch * bytes_per_unit, "binding_bound": bound}
    if note:
        r["note"] = note
    return r


def other_configs(f, torch, dev):
    """BASELINE.json configs C1-C4 with SURVEY §8d's inputs and step counts, each event-timed on its handle's stream."""
    import ctypes as C
    out = []
    stream = torch.cuda.Stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    # ---- C2: tau_hypersonic_cuda 2D 4096^2 fp32, k_init geometry scaled to the grid, 50 warm-up + 200 timed steps
    n = 4096
    e = f.Hypersonic2D(n, n, stream=sp)
    e.init()
    e.step_async(50)
    ms = _event_timed(torch, stream, lambda: e.step_async(200), e.sync)
    out.append({"config": f"tau_hypersonic_cuda 2D {n}x{n} fp32 (one fused kernel per step)", "steps": 200, "warmup": 50,
                "value": round(n * n * 200 / ms / 1e6, 3), "unit": "Gcell-updates/s", "ms_per_step": round(ms / 200, 5),
                "roofline": _roof("h2d::k_march", ms / 200, n * n, 33, "valu")})
    e.close()

    # ---- C3: tau_gray_scott 8192^2, init_pattern(seed 1337), 1000 steps: four time levels per pass (the default), and the
    #      single-step kernel (one launch per step)
    n = 8192
    g = f.GrayScott(n, n, stream=sp)
    g.init_pattern(1337)
    g.step_async(40)
    ms = _event_timed(torch, stream, lambda: g.step_async(1000), g.sync)
    out.append({"config": f"tau_gray_scott {n}x{n}, 4 time levels per pass (default)", "steps": 1000, "warmup": 40,
                "value": round(n * n * 1000 / ms / 1e6, 2), "unit": "Gcell-updates/s", "ms_per_step": round(ms / 1000, 5),
                "roofline": _roof("st2::k_fused<GS,4> (one launch = 4 steps)", ms / 250, 4 * n * n, 16, "valu + hbm",
                                  "16 B per update is the single-step algorithmic figure; a 4-level pass moves ~4.6 B per "
                                  "update (profiles/r01i), so frac > 1 is traffic removed, not bandwidth exceeded")})

    def single():
        for _ in range(1000):
            g.step_async(1)
    ms = _event_timed(torch, stream, single, g.sync)
    out.append({"config": f"tau_gray_scott {n}x{n}, one step per launch", "steps": 1000, "warmup": 0,
                "value": round(n * n * 1000 / ms / 1e6, 2), "unit": "Gcell-updates/s", "ms_per_step": round(ms / 1000, 5),
                "roofline": _roof("st2::k_march<GS>", ms / 1000, n * n, 16, "hbm")})
    g.close()

    # ---- C4: tau_sph 4 194 304 particles, reset_particles(seed 69420), rain off, 200 sub-steps: on the lattice start
    #      and on the developed (collapsing dam) state 1500 sub-steps later
    N = 1 << 22
    s = f.Sph2D(N, stream=sp)
    s.reset_particles()
    s.step_async(20)
    for label, pre in (("lattice start (after 20 sub-steps)", 0), ("developed state (after 1500 more sub-steps)", 1500)):
        if pre:
            s.step_async(pre)
        ms = _event_timed(torch, stream, lambda: s.step_async(200), s.sync)
        out.append({"config": f"tau_sph {N} particles, {label}", "steps": 200,
                    "value": round(N * 200 / ms / 1e6, 3), "unit": "Gparticle-sub-steps/s", "ms_per_step": round(ms / 200, 5),
                    "roofline": _roof("sph sub-step (cell build + k_density + k_forces)", ms / 200, N, 100,
                                      "valu (pair evaluation)")})
    s.close()

    # ---- C1: tau_hypersonic.c restated (fp64 scalar CPU program), 256^2, 100 steps after init_sim, 1 thread
    from importlib import import_module
    m = import_module("fluid_sims_amd.cpu2d")
    c = m.CpuHypersonic2D(256, 256, simd=False)
    t0 = time.perf_counter()
    c.step(100)
    el = time.perf_counter() - t0
    out.append({"config": "tau_hypersonic.c restated, 2D 256x256 fp64 scalar, CPU, 1 thread", "steps": 100,
                "value": round(256 * 256 * 100 / el / 1e9, 6), "unit": "Gcell-updates/s", "ms_per_step": round(el * 10, 4),
                "roofline": None})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=512, help="grid edge (headline: 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the two extra SURVEY 8d inputs (reference IC, no body)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json configs (2D Euler, Gray-Scott, SPH, CPU)")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the Z-slab ring driver even on one GPU (self-neighbour halo copies): exercises the N>1 code path")
    ap.add_argument("--self-p2p", action="store_true",
                    help="with --force-slab on one GPU: exchange the halos with OURSELVES through RCCL send/recv and all-reduce the "
                         "max words (a world-of-one process group) instead of device copies")
    args = ap.parse_args()

    import numpy as np
    import torch
    import fluid_sims_amd as f

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (libtaueng has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n = args.n
    if world > 1 or (args.force_slab and args.self_p2p):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29581")
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from importlib import import_module
    slab = import_module("fluid_sims_amd.slab")

    L = f.load()
    params = f.Tau3DParams()
    L.tau3d_params_default(ctypes.byref(params), n, n, n)
    z0, nzl = slab.slab_bounds(n, world, rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if world == 1 and not args.force_slab:
        eng = f.Tau3D(n, n, n, params=params, device=local)
        eng.init(1)
        eng.set_clock(0.02, 1e-4)
        step = lambda k: eng.step_async(k)          # noqa: E731
        sync = eng.sync
        h = eng
    else:
        be = slab.EngineSlabBackend(f.taueng, params, z0, nzl, local)
        be.h.init(1)
        be.h.set_clock(0.02, 1e-4)
        ring = slab.SlabRing(be, rank, world, self_p2p=args.self_p2p)
        ring.prime()
        step = lambda k: ring.step(k)               # noqa: E731
        sync = ring.finish
        h = be.h

    step(args.warmup)
    sync()
    h.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    step(args.steps)
    sync()
    barrier()
    el = time.perf_counter() - t0
    k_ms, k_launches, k_cells = h.timing_read()
    xy_ms, z_ms, n_split = h.timing_read_split()
    h.timing_enable(False)

    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())

    clk = h.clock()
    cells_total = float(n) ** 3 * args.steps
    value = cells_total / el / 1e9

    if rank == 0:
        # The step is two kernels over the same planes (k_flux_xy, then k_update_z): the events bracket the pair, so
        # achieved = algorithmic bytes of a step / the summed duration of both.  `traffic` is NOT measured in this run:
        # it is the PMC figure of the committed profile (profiles/k_step_traffic.json names the passes it came from).
        k_s = k_ms * 1e-3
        achieved = ALGO_BYTES_PER_CELL * k_cells / k_s / 1e9 if k_s > 0 else 0.0
        traffic, tsrc = None, None
        tpath = os.path.join(ROOT, "profiles", "k_step_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic, tsrc = tj.get("hbm_bytes_per_launch"), tj.get("source")
            except Exception:
                traffic = None
        split = n * n >= 128 * 128 and os.environ.get("TAU3D_SPLIT", "1") != "0"
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": f"profile-derived, not this run: {tsrc}" if traffic else None,
                "kernel": "h3d::k_flux_xy + h3d::k_update_z (one step = the pair)" if split else "h3d::k_step",
                "launches": args.steps, "event_intervals": k_launches,   # a Z-slab step is two timed intervals (edges, interior)
                "avg_launch_ms": round(k_ms / max(args.steps, 1), 4),    # kernel time of ONE step on this GPU (rank 0)
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL * k_cells / max(args.steps, 1),
                "per_gpu": world > 1,                                     # N > 1: rank 0's slab (its cells / its kernel time)
                "kernels": ([{"name": "h3d::k_flux_xy", "avg_launch_ms": round(xy_ms / n_split, 4)},
                             {"name": "h3d::k_update_z", "avg_launch_ms": round(z_ms / n_split, 4)}] if n_split else None),
                "note": "the step is FP32-VALU bound (WENO5 + HLLC, ~2.16 k VALU instructions per 64 cells at ~3.3 cycles each against "
                        "a ~2.3-cycle full-rate issue and a ~3.0-cycle floor for the instruction mix, profiles/r02/valu_calib.txt, "
                        "pmc_sq.txt); the HBM fraction is reported because BASELINE.json's metric asks for it"}
        out = {"metric": "Gcell-updates/s, 3D hypersonic 512^3 fp32", "value": round(value, 4),
               "unit": "Gcell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"tau_hypersonic_3d {n}^3 fp32, sphere r=0.25, Mach-100 inflow, "
                                      f"developed-flow start (SURVEY 8d input ii)",
                          "grid": [n, n, n], "decomposition": f"z-slab x{world}" if world > 1 else "single domain",
                          "halo_planes": 3, "t": clk.t, "d_tau": clk.d_tau, "maxs": clk.maxs},
               "roofline": roof}
        if world == 1 and not args.no_variants and not args.force_slab:
            try:
                out["other_inputs"] = input_variants(f, n)
            except Exception as e:  # extras never take the headline down
                out["other_inputs"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            planes = 8
            zc = n // 2 - 40 if n >= 128 else 0
            planes = min(planes, n)
            h.fill_halo_periodic_async()                   # the sample may reach into the halo planes: make them current
            h.sync()
            st = h.download_planes(zc - 3, zc + planes + 3)
            cpu_sample = ([np.ascontiguousarray(a) for a in st], clk.dt, zc, planes)
        if world == 1 and not args.no_configs and not args.force_slab:
            h.close()                                      # the 512^3 state (20 GB with the primitive cache) is not needed below
            h = None
            try:
                out["configs"] = other_configs(f, torch, dev)
            except Exception as e:  # extras never take the headline down
                out["configs"] = [{"error": str(e)}]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, *cpu_sample)
            try:
                out["cpu_baseline_2d_simd"] = cpu_baseline_2d()
                out["cpu_baseline_2d_simd_256"] = cpu_baseline_2d(n=256)      # BASELINE config 1 size
                out["cpu_baseline_2d_simd_all_cores"] = cpu_baseline_2d_all_cores()
            except Exception as e:  # the 2D CPU program is an extra, never fatal for the headline
                out["cpu_baseline_2d_simd"] = {"error": str(e)}
        try:   # RCCL's version banner (NCCL_DEBUG=VERSION on the GPU boxes) sits in the C stdio buffer until exit: push it out
            import ctypes as _ct   # first, so that the JSON line is the LAST line of stdout
            _ct.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)

    if world > 1 or (args.force_slab and args.self_p2p):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
