#!/usr/bin/env python
"""bench.py — headline benchmark: Gcell-updates/s of the 3D hypersonic step on a 512^3 fp32 grid.

  python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 it re-launches itself under
                                                           torch.distributed.run, one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the same, launched from outside)

A "step" is one full time step of the reference loop (tau_hypersonic_3d_cuda.cu:1680-1711: log-time
clock, k_step over the whole grid, d_tau controller, swap) on synthetic input already resident in
HBM: the SURVEY §8d "developed flow" start (every fluid cell at the Mach-100 inflow state, t = 0.02 so
the inflow gain is 1, sphere r = 0.25 at the centre), W warm-up steps so the bow shock exists and the
controller has settled, then exactly K timed steps bracketed by barrier + device sync.

N > 1: the SAME 512^3 grid is Z-slab partitioned over the ranks (strong scaling, as BASELINE.json's
metric states: "512^3 at 1/2/4/8 MI355X"); per step each rank exchanges 3 boundary planes x 6 fields
with both ring neighbours (RCCL send/recv over xGMI, overlapped with the interior planes) and one
8-byte all-reduce(max) (max wavespeed + max |primitive|) feeds the device-side d_tau controller.  The ring
is the library's (tau3d_ring_*, csrc/ring.hip: librccl called from C on a private stream); torch.distributed
only provides the launch, the barrier and the max-over-ranks of the contract.  --ring-driver python selects the
older torch.distributed ring of fluid-sims_amd/slab.py instead.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`; at N = 1 the
line also carries `configs`: the other four BASELINE.json configurations (2D Euler 4096^2, Gray-Scott 8192^2,
SPH 4 194 304 particles, the CPU program at 256^2), each with its own event-timed roofline.
"""
import argparse
import ctypes
import json
import math
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


def input_variants(f, torch, dev, n, steps=20):
    """SURVEY §8d asks for two more inputs beside the headline one: (i) the reference's own quiescent k_init
    start after 50 controller warm-up steps, and a no-body variant that bounds the branch-free throughput.
    Event-timed on the handle's stream, `steps` steps each."""
    out = {}
    stream = torch.cuda.Stream(dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    # developed_late: the LATE developed state that exists at 512^3 — the reference's own ramped start after 2500 steps (inflow gain
    # 0.8, bow shock standing, wake formed; the impulsive start of the headline runs away after ~55 steps and the ramped one before
    # step 3250, in the reference's own kernel as in this engine: profiles/r04/long_run_512_*.txt, tests/test_gpu_ref3d.py).
    # forced_reciprocal_weights: the headline input with TAU3D_WENO_RCP=1 — the general-form bodies flux_xy_body<false> /
    # update_z_body<false> a state beyond |primitive| 2.5e3 would take (no sane 512^3 state does).
    late_warm = 2500 if n >= 512 else 400
    # headline_uniform_exits_off: the headline input with TAU3D_UNIFORM_EXITS=0 — every face of every cell evaluated, as the reference
    # does; the uniform-region exits (include/taueng.h: tau3d_uniform_tiles) give the same bits in less time wherever the flow is still
    # the undisturbed inflow state.  developed_late_exits_off: the same for the late state.
    for name, mode, warm, body, rcp, exits in (("reference_ic_50_warmup", 0, 50, True, False, True), ("developed_no_body", 1, 10, False, False, True),
                                               ("developed_late", 0, late_warm, True, False, True), ("forced_reciprocal_weights", 1, 25, True, True, True),
                                               ("headline_uniform_exits_off", 1, 10, True, False, False),
                                               ("developed_late_exits_off", 0, late_warm, True, False, False)):
        p = f.Tau3DParams()
        f.load().tau3d_params_default(ctypes.byref(p), n, n, n)
        if not body:
            p.sdf_r = -1.0
        if rcp:
            os.environ["TAU3D_WENO_RCP"] = "1"       # read by tau3d_create
        if not exits:
            os.environ["TAU3D_UNIFORM_EXITS"] = "0"  # read by tau3d_create
        try:
            e = f.Tau3D(n, n, n, params=p, stream=sp)
        finally:
            os.environ.pop("TAU3D_WENO_RCP", None)
            os.environ.pop("TAU3D_UNIFORM_EXITS", None)
        e.init(mode)
        if mode:
            e.set_clock(0.02, 1e-4)
        e.step_async(warm)
        ms = _event_timed(torch, stream, lambda: e.step_async(steps), e.sync)
        c = e.clock()
        ut = e.uniform_tiles()
        fr = e.field_range()
        out[name] = {"value": round(float(n) ** 3 * steps / ms / 1e6, 3), "unit": "Gcell-updates/s", "steps": steps, "warmup": warm,
                     "timing": "HIP events on the handle's stream", "weno_form": "fast (common denominator)" if fr[2] else "reciprocal",
                     "max_abs_primitive": round(float(max(fr[0], fr[1])), 1) if max(fr[0], fr[1]) < 3e38 else "inf",
                     "state_sane": bool(fr[2] or rcp) and math.isfinite(float(c.maxs)) and float(max(fr[0], fr[1])) <= 6e4,
                     "t": c.t, "gain": round(c.gain, 4),
                     "uniform_exits": bool(ut[2]), "uniform_tile_fraction": round(ut[0] / max(ut[1], 1), 4),
                     "tile_list": _tile_list(e)}
        e.close()
    return out


def _tile_list(e):
    """predicted-uniform tile list (include/taueng.h: tau3d_tile_list_stats): the share of k_flux_xy's tiles the step after the last
    timed one would have launched — the others were flagged uniform from the flags of the step before, without being read"""
    mode, listed, tiles, _, _ = e.tile_list_stats()
    return {"mode": mode, "listed_tile_fraction": round(listed / max(tiles, 1), 4) if mode == 1 and listed >= 0 else None}


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


def _profile_traffic(keys):
    """HBM bytes per launch of the named kernels from the committed PMC profile (profiles/k_step_traffic.json, regenerated from
    profiles/<round>/secondary_fetch.txt + secondary_write.txt by scripts/make_traffic_json.py) — or None"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "k_step_traffic.json")))
        sec = tj["secondary_hbm_bytes_per_launch"]
        return sum(sec[k] for k in keys), tj.get("secondary_source")
    except Exception:
        return None, None


def _roof(kernel, ms_per_launch, units_per_launch, bytes_per_unit, bound, note=None, traffic_keys=None, traffic_field=None):
    gbs = units_per_launch * bytes_per_unit / (ms_per_launch * 1e-3) / 1e9
    traffic, tsrc = _profile_traffic(traffic_keys) if traffic_keys else (None, None)
    if traffic_field:   # a ready-made per-launch total of the committed profile (e.g. the SPH developed state's own passes)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "k_step_traffic.json")))
            traffic, tsrc = tj[traffic_field[0]], tj.get(traffic_field[1])
        except Exception:
            traffic, tsrc = None, None
    r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
         "traffic": traffic, "kernel": kernel, "avg_launch_ms": round(ms_per_launch, 5),
         "algorithmic_bytes_per_launch": units_per_launch * bytes_per_unit, "binding_bound": bound}
    if traffic is not None:
        r["traffic_source"] = f"profile-derived, not this run: {tsrc}"
    if note:
        r["note"] = note
    return r


def other_configs(f, torch, dev):
    """BASELINE.json configs C1-C4 with SURVEY §8d's inputs and step counts, each event-timed on its handle's stream."""
    import ctypes as C
    out = []
    stream = torch.cuda.Stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    # ---- C2: tau_hypersonic_cuda 2D 4096^2 fp32, k_init geometry scaled to the grid, 50 warm-up + 200 timed steps — the
    #      engine as it ships (uniform-region exits: a trip of the march whose five-row window holds one state skips its predictors
    #      and faces, same bits), the same input with every trip evaluated (TAUH2_UNIFORM_EXITS=0), and the state 3000 steps later,
    #      when the shock layer has grown across the domain
    n = 4096
    for label, exits, warm in (("", True, 50), (", every trip evaluated (TAUH2_UNIFORM_EXITS=0)", False, 50), (", after 3000 steps", True, 3000)):
        if not exits:
            os.environ["TAUH2_UNIFORM_EXITS"] = "0"     # read by tauh2_create
        try:
            e = f.Hypersonic2D(n, n, stream=sp)
        finally:
            os.environ.pop("TAUH2_UNIFORM_EXITS", None)
        e.init()
        e.step_async(warm)
        ms = _event_timed(torch, stream, lambda: e.step_async(200), e.sync)
        kname = "h2d::k_march_lds<1, true>" if exits else "h2d::k_march_lds<1, false>"
        out.append({"config": f"tau_hypersonic_cuda 2D {n}x{n} fp32 (one fused kernel per step){label}", "steps": 200, "warmup": warm,
                    "value": round(n * n * 200 / ms / 1e6, 3), "unit": "Gcell-updates/s", "ms_per_step": round(ms / 200, 5),
                    "uniform_exits": exits,
                    "roofline": _roof(kname, ms / 200, n * n, 33, "valu" if not exits else "valu in the shock layer, load latency in the free stream",
                                      traffic_keys=[kname] if warm == 50 else None)})
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
                "roofline": _roof("st2::k_fused<GS,4> (one launch = 4 steps)", ms / 250, n * n, 16, "valu + hbm",
                                  "per LAUNCH: the compulsory traffic of a pass is one read and one write of u, v (16 B per cell) "
                                  "whatever the number of time levels it advances; per cell-update the pass moves a quarter of that",
                                  traffic_keys=["st2::k_fused<0, 4>"])})
    g.set_levels(1)                                  # the reference's structure: one launch per step (bit-identical results)
    g.step_async(8)
    ms = _event_timed(torch, stream, lambda: g.step_async(1000), g.sync)   # ONE call: 1000 launches enqueued back to back
    out.append({"config": f"tau_gray_scott {n}x{n}, one step per launch", "steps": 1000, "warmup": 8,
                "value": round(n * n * 1000 / ms / 1e6, 2), "unit": "Gcell-updates/s", "ms_per_step": round(ms / 1000, 5),
                "roofline": _roof("st2::k_march<GS>", ms / 1000, n * n, 16, "hbm", traffic_keys=["st2::k_march<0>"])})
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
        s.sync()
        pairs0 = s.count_pairs()                       # ordered pairs inside the 2h support at the start of the timed window ...
        ms = _event_timed(torch, stream, lambda: s.step_async(200), s.sync)
        pairs1 = s.count_pairs()                       # ... and at its end: each sub-step's density AND force pass evaluate every one of them
        pair_rate = 0.5 * (pairs0 + pairs1) * 2 * 200 / ms / 1e6
        out.append({"config": f"tau_sph {N} particles, {label}", "steps": 200,
                    "value": round(N * 200 / ms / 1e6, 3), "unit": "Gparticle-sub-steps/s", "ms_per_step": round(ms / 200, 5),
                    "pair_interactions": {"value": round(pair_rate, 1), "unit": "G pair-interactions/s",
                                          "neighbours_per_particle": round(0.5 * (pairs0 + pairs1) / N, 1),
                                          "note": "ordered pairs (i, j) within 2h x 2 neighbour passes per sub-step; pair count = mean of the "
                                                  "counts before and after the timed window (tausph_count_pairs)"},
                    "roofline": _roof("sph sub-step (counting-sort cell build + k_density + k_forces)", ms / 200, N, 100,
                                      "valu (pair evaluation)",
                                      traffic_keys=(["sph::k_tile_sums", "sph::k_scan", "sph::k_scatter", "sph::k_rank_gather", "sph::k_density<1>",
                                                     "sph::k_forces<1>"] if not pre else None),
                                      traffic_field=(("sph_developed_hbm_bytes_per_substep", "sph_developed_source") if pre else None))})
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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_spawn(n):
    """--gpus N without a launcher: run this very command under torch.distributed.run, one rank per GPU.  Whether the node
    has N devices is checked INSIDE the ranks (each says what it sees), not here."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's intra-node transport needs it on these hosts
    env.setdefault("NCCL_DEBUG", "VERSION")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", dest="n", type=int, default=512,
                    help="grid edge (headline: 512).  (Not --n: torch.distributed.run's own parser takes that for an abbreviation of --nnodes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the two extra SURVEY 8d inputs (reference IC, no body)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json configs (2D Euler, Gray-Scott, SPH, CPU)")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the Z-slab ring even on one GPU (self-neighbour halo copies): exercises the N>1 code path")
    ap.add_argument("--self-p2p", action="store_true",
                    help="with --force-slab on one GPU: exchange the halos with OURSELVES through RCCL send/recv and all-reduce the "
                         "max words (a communicator of one) instead of device copies")
    ap.add_argument("--ring-transport", choices=("auto", "rccl", "ipc", "host", "ipc-host"), default="auto",
                    help="how the halos travel at N > 1.  ipc: every rank writes its boundary planes straight into its neighbours' halo "
                         "planes (hipIpc-mapped state, hipMemcpyAsync: SDMA over xGMI, no CU), RCCL only for the 8-byte all-reduce; "
                         "rccl: ncclSend / ncclRecv of packed buffers + ncclAllReduce; auto (default): both are built and timed over a "
                         "few untimed warm-up steps, the faster one runs the timed region (a transport that cannot be set up on every "
                         "rank is skipped).  host / ipc-host: halos (resp. only the all-reduce) staged through shared memory so that "
                         "the ranks may SHARE a device — this script's N > 1 path on a one-GPU box (process group over gloo); not a "
                         "performance configuration")
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1 (or --force-slab): skip the check that follows the timed region — every rank recomputes warmup + steps "
                         "single-domain steps of the whole grid from the same start on its own device and compares its slab with "
                         "those planes byte for byte (ring.matches_single_domain; a mismatch exits 4)")
    ap.add_argument("--ring-driver", choices=("c", "python"), default="c",
                    help="c: the library's ring (tau3d_ring_*, librccl called from C); python: fluid-sims_amd/slab.py over torch.distributed")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_spawn(args.gpus))

    if args.gpus > 1:   # before the HIP runtime comes up in this rank (the launcher's environment normally carries it already)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import fluid_sims_amd as f

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: drop the launcher (bench.py spawns its own ranks) or make them agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (libtaueng has no CPU path)")
    ndev = torch.cuda.device_count()
    # TAU_BENCH_AUTO_SHARED=1 (tests on a one-GPU box): `auto` probes ipc-host against host instead of ipc against rccl — the same
    # build / time / close / rebuild sequence with ranks that share the device
    auto_shared = args.ring_transport == "auto" and bool(os.environ.get("TAU_BENCH_AUTO_SHARED"))
    shared = args.ring_transport in ("host", "ipc-host") or auto_shared
    if ndev < world and not shared:
        if ndev == 1 and os.environ.get("TAU_BENCH_ONE_VISIBLE_DEVICE_PER_RANK"):
            local = 0    # a launcher that shows every rank exactly its own device; the library compares device identities across ranks
        else:
            raise SystemExit(f"bench.py --gpus {world}: rank {rank} sees {ndev} device(s); the Z-slab ring needs {world} devices, "
                             f"one MI355X per rank")
    if shared:
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n = args.n
    use_ring = world > 1 or args.force_slab
    py_ring = use_ring and args.ring_driver == "python"
    need_pg = world > 1 or (py_ring and args.self_p2p)
    if need_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29581")
        if world > 1 and shared:
            dist.init_process_group("gloo")
        elif world > 1:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    L = f.load()
    params = f.Tau3DParams()
    L.tau3d_params_default(ctypes.byref(params), n, n, n)
    z0, nzl = f.slab_bounds(n, world, rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    ring_info = None
    closer = None
    ring_obj = None
    if not use_ring:
        eng = f.Tau3D(n, n, n, params=params, device=local)
        eng.init(1)
        eng.set_clock(0.02, 1e-4)
        step = lambda k: eng.step_async(k)          # noqa: E731
        sync = eng.sync
        get_clock = eng.clock
        h = eng
    else:
        def python_ring(why=None):
            from importlib import import_module
            slab = import_module("fluid_sims_amd.slab")
            be = slab.EngineSlabBackend(f.taueng, params, z0, nzl, local)
            be.h.init(1)
            be.h.set_clock(0.02, 1e-4)
            ring = slab.SlabRing(be, rank, world, self_p2p=args.self_p2p and need_pg)
            ring.prime()
            info = {"driver": "python (torch.distributed batch_isend_irecv + all_reduce)"}
            if why:
                info["fallback_from_c_ring"] = why
            return (lambda k: ring.step(k)), ring.finish, be.h.clock, be.h, info, None, None

        TRANSPORT_NAMES = {f.RING_RCCL: "rccl", f.RING_LOCAL: "local device copies", f.RING_IPC: "ipc",
                           f.RING_HOST: "host-staged (ranks share a device: not a performance configuration)",
                           f.RING_IPC_HOSTMAX: "ipc-host (direct halos, host all-reduce; ranks share a device: not a performance configuration)"}

        def c_ring(transport):
            # the library's ring: every rank passes the same rendezvous path and job key (agreed through torch.distributed)
            key = torch.randint(1, 2 ** 62, (1,), dtype=torch.int64, device="cpu" if shared else dev)
            if world > 1:
                dist.broadcast(key, 0)
            key = int(key.item())
            eng = f.Tau3D(n, n, n, params=params, z0=z0, nzl=nzl, device=local)
            eng.init(1)
            eng.set_clock(0.02, 1e-4)
            err = None
            ring = None
            try:
                if os.environ.get("TAU_BENCH_FAIL_C_RING"):
                    raise RuntimeError("TAU_BENCH_FAIL_C_RING is set (test of the fallback)")
                ring = f.Tau3DRing(eng, rank, world, transport,
                                   rendezvous=f"/dev/shm/tau3d_bench_{key & 0xffffffffff:x}" if world > 1 else None, job_key=key)
                ring.prime()
            except Exception as e:  # a rank that cannot build the ring tells the others (below): all of them fall back together
                err = f"rank {rank}: {e}"
            if world > 1:   # agree: the C ring runs only if EVERY rank has it
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device="cpu" if shared else dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and err is None:
                    err = "another rank could not create the C ring"
            if err:
                if ring is not None:
                    ring.close()
                eng.close()
                return None, err
            info = dict(ring.info(), driver="c (tau3d_ring_*: librccl from libtaueng)", transport=TRANSPORT_NAMES[transport],
                        step_schedule=("edge / interior launches (TAU3D_RING_PIPELINE=0)" if os.environ.get("TAU3D_RING_PIPELINE", "1") == "0"
                                       else "pipelined, all-reduce first, two all-reduces on the direct transports (TAU3D_RING_SPEC=0)"
                                       if os.environ.get("TAU3D_RING_SPEC", "1") == "0"
                                       else "x/y fluxes of all planes ahead of the all-reduce; exchange, then ONE all-reduce, beside them"))

            def close_all():
                ring.close()
                eng.close()
            return ((lambda k: ring.step(k)), ring.finish, ring.clock, eng, info, close_all, ring), None

        def probe(transport, steps=10):
            """ms per step of one candidate over `steps` steps after the same warm-up, max over ranks (untimed region of the bench)"""
            got, why = c_ring(transport)
            if got is None:
                return None, why
            st, sy, _, _, _, cl, _ = got
            # a candidate that builds but then fails while stepping (a HIP / RCCL error out of the library) must not take the run
            # down: every rank says how it went, the candidate counts only if ALL of them got through, and it is closed either way
            # Every rank walks the SAME sequence of process-group collectives whatever happens in between.
            ms, err = float("inf"), None

            def attempt(fn):
                nonlocal err
                if err is None:
                    try:
                        fn()
                    except Exception as e:
                        err = f"rank {rank}: {e}"

            def warm():
                if os.environ.get("TAU_BENCH_FAIL_PROBE") == str(transport) and rank == world - 1:
                    raise RuntimeError("TAU_BENCH_FAIL_PROBE is set (test: a candidate that fails while stepping)")
                st(args.warmup)
                sy()
            def always(fn):      # a collective: issued even by a rank that already failed, so that the others are not left waiting
                nonlocal err
                try:
                    fn()
                except Exception as e:
                    err = err or f"rank {rank}: {e}"

            attempt(warm)
            always(barrier)
            t0 = time.perf_counter()
            attempt(lambda: (st(steps), sy()))
            if err is None:
                ms = (time.perf_counter() - t0) / steps * 1e3
            always(barrier)
            try:
                t = torch.tensor([ms if err is None else float("inf")], dtype=torch.float64, device="cpu" if shared else dev)
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t.item())
            except Exception as e:
                err = err or f"rank {rank}: {e}"
                ms = float("inf")
            try:
                cl()
            except Exception as e:
                err = err or f"rank {rank}: close: {e}"
            if not math.isfinite(ms):
                return None, err or "another rank failed while stepping this transport"
            return ms, None

        want = args.ring_transport
        if world == 1:
            cands = [f.RING_IPC if want == "ipc" else (f.RING_RCCL if args.self_p2p else f.RING_LOCAL)]
        elif want == "auto":
            cands = [f.RING_IPC_HOSTMAX, f.RING_HOST] if auto_shared else [f.RING_IPC, f.RING_RCCL]
        else:
            cands = [{"rccl": f.RING_RCCL, "ipc": f.RING_IPC, "host": f.RING_HOST, "ipc-host": f.RING_IPC_HOSTMAX}[want]]
        probes = {}
        if not py_ring and len(cands) > 1:
            for tr in cands:
                ms, why_not = probe(tr)
                probes[TRANSPORT_NAMES[tr]] = round(ms, 4) if ms is not None else f"unavailable: {why_not}"
            ok = [tr for tr in cands if isinstance(probes[TRANSPORT_NAMES[tr]], float)]
            cands = [min(ok, key=lambda tr: probes[TRANSPORT_NAMES[tr]])] if ok else [f.RING_RCCL]
        got, why = (None, None) if py_ring else c_ring(cands[0])
        if got is not None and probes:
            got[4]["auto_probe_ms_per_step"] = probes
        if got is None:
            if why and rank == 0:
                print(f"bench.py: C ring unavailable ({why}); falling back to the torch.distributed ring", file=sys.stderr)
            got = python_ring(why)
        step, sync, get_clock, h, ring_info, closer, ring_obj = got

    step(args.warmup)
    sync()
    h.timing_enable(True)
    if ring_obj is not None:
        ring_obj.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    step(args.steps)
    sync()
    barrier()
    el = time.perf_counter() - t0
    ring_times = ring_obj.timing_read() if ring_obj is not None else None
    if ring_obj is not None:
        ring_obj.timing_enable(False)
    k_ms, k_launches, k_cells = h.timing_read()
    span_ms = h.timing_span()          # HIP events on the launch stream: first launch of the timed region -> end of the last
    xy_ms, z_ms, n_split = h.timing_read_split()
    h.timing_enable(False)
    is_split = h.is_split()

    per_rank_ms = None
    if world > 1:
        cdev = "cpu" if shared else dev
        tmax = torch.tensor([el], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
        km = torch.zeros(world, dtype=torch.float64, device=cdev)
        km[rank] = k_ms / max(args.steps, 1)
        dist.all_reduce(km, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 4) for x in km.tolist()]
        sp = torch.tensor([span_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(sp, op=dist.ReduceOp.MAX)
        span_ms = float(sp.item())

    clk = get_clock()
    frange = h.field_range()
    utiles = h.uniform_tiles()
    tlist = _tile_list(h)

    # ---- the ring run proves itself: slab == single domain, byte for byte (SURVEY 8e's parity oracle, on the hardware and with
    # the transport that was just timed).  The grid fits one GPU, so EVERY rank recomputes warmup + steps single-domain steps from
    # the same start on its own device and compares its own planes locally — no data crosses the ring it is checking.
    verify = None
    per_rank = None
    if use_ring:
        cdev = "cpu" if shared else dev
        if not args.no_verify:
            t0v = time.perf_counter()
            ok, first_bad, nbad, clock_ok, verr = 1, 2 ** 31 - 1, 0, 1, None
            try:
                mine = h.download()
                one = f.Tau3D(n, n, n, params=params, device=local)
                one.init(1)
                one.set_clock(0.02, 1e-4)
                one.step_async(args.warmup + args.steps)
                one.sync()
                c1 = one.clock()
                want = one.download_planes(z0, z0 + nzl)
                one.close()
                if os.environ.get("TAU_BENCH_VERIFY_SELFTEST") and rank == world - 1:   # test of the check itself: one bit of one word
                    mine[3].view(np.uint32)[nzl // 2, 5, 7] ^= 1
                bad_planes = np.zeros(nzl, bool)
                for a, b in zip(mine, want):       # bit patterns, so that -0.0 / NaN payloads count too
                    bad_planes |= (a.view(np.uint32) != b.view(np.uint32)).reshape(nzl, -1).any(axis=1)
                nbad = int(bad_planes.sum())
                if nbad:
                    ok, first_bad = 0, z0 + int(np.argmax(bad_planes))
                clock_ok = int((c1.t, c1.d_tau, c1.maxs) == (clk.t, clk.d_tau, clk.maxs))
                if not clock_ok:
                    ok = 0
            except Exception as e:
                ok, verr = 0, f"rank {rank}: {e}"
            v = torch.tensor([ok, clock_ok, first_bad], dtype=torch.int64, device=cdev)
            tot = torch.tensor([nbad], dtype=torch.int64, device=cdev)
            if world > 1:
                dist.all_reduce(v, op=dist.ReduceOp.MIN)
                dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            v = [int(x) for x in v.tolist()]
            verify = {"matches_single_domain": bool(v[0]), "clock_matches": bool(v[1]),
                      "first_differing_plane": (v[2] if v[2] < 2 ** 31 - 1 else None), "differing_planes": int(tot.item()),
                      "compared": f"{n}^3 x 6 fields, every rank's planes against its own single-domain recompute of "
                                  f"{args.warmup + args.steps} steps, bit patterns", "seconds": round(time.perf_counter() - t0v, 2)}
            if verr:
                verify["error"] = verr
        # per-rank event times of the timed steps: x/y kernel, z kernel (handle's stream), exchange, all-reduce (ring's stream)
        rt = ring_times or (0.0, 0.0, 0)
        row = torch.zeros(world, 4, dtype=torch.float64, device=cdev)
        row[rank] = torch.tensor([xy_ms / max(n_split, 1), z_ms / max(n_split, 1), rt[0] / max(rt[2], 1), rt[1] / max(rt[2], 1)],
                                 dtype=torch.float64)
        if world > 1:
            dist.all_reduce(row, op=dist.ReduceOp.SUM)
        per_rank = [{"rank": i, "xy_ms": round(r[0], 4), "z_ms": round(r[1], 4), "exchange_ms": round(r[2], 4),
                     "allreduce_ms": round(r[3], 4)} for i, r in enumerate(row.tolist())]
    cells_total = float(n) ** 3 * args.steps
    value = cells_total / el / 1e9
    # sane = finite, positive wavespeed and every |primitive| bounded — whichever WENO weight form the steps ran (the reciprocal
    # form is a legitimate run: TAU3D_WENO_RCP=1, or a state between the fast window's 2.5e3 and 6e4)
    state_sane = math.isfinite(float(clk.maxs)) and float(clk.maxs) > 0.0 and float(max(frange[0], frange[1])) <= 6e4

    if rank == 0:
        # The step is two kernels over the same planes (k_flux_xy, then k_update_z): the events bracket the pair, so
        # achieved = algorithmic bytes of a step / the summed duration of both.  `traffic` is NOT measured in this run:
        # it is the PMC figure of the committed profile (profiles/k_step_traffic.json names the passes it came from).
        k_s = k_ms * 1e-3
        achieved = ALGO_BYTES_PER_CELL * k_cells / k_s / 1e9 if k_s > 0 else 0.0
        traffic, tsrc, valu = None, None, None
        tpath = os.path.join(ROOT, "profiles", "k_step_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic, tsrc = tj.get("hbm_bytes_per_launch"), tj.get("source")
                valu = tj.get("valu")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic if n == 512 and world == 1 else None,
                "traffic_source": f"profile-derived, not this run: {tsrc}" if traffic and n == 512 and world == 1 else None,
                "kernel": ("h3d::k_flux_xy + h3d::k_tile_predict + h3d::k_update_z + h3d::k_fill_z (one step; single domain: k_flux_xy over the "
                           "list of tiles not predicted uniform, k_fill_z the fully predicted chunks)") if is_split else "h3d::k_step",
                "launches": args.steps, "event_intervals": k_launches,   # a Z-slab step is two timed intervals (edges, interior)
                "avg_launch_ms": round(k_ms / max(args.steps, 1), 4),    # kernel time of ONE step on this GPU (rank 0)
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL * k_cells / max(args.steps, 1),
                "per_gpu": world > 1,                                     # N > 1: rank 0's slab (its cells / its kernel time)
                # (events on the handle's stream: before k_flux_xy / between it and k_tile_predict / after k_update_z)
                "kernels": ([{"name": "h3d::k_flux_xy", "avg_launch_ms": round(xy_ms / n_split, 4)},
                             {"name": "h3d::k_tile_predict + h3d::k_update_z + h3d::k_fill_z", "avg_launch_ms": round(z_ms / n_split, 4)}] if n_split else None),
                "note": "where the flow is disturbed the step is FP32-VALU bound (WENO5 + HLLC), where it is predicted uniform it is six "
                        "stores per cell; the HBM fraction is reported because BASELINE.json's metric asks for it (algorithmic bytes: every "
                        "cell read and written once per kernel, whatever was skipped), the VALU block beside it counts issued instructions"}
        if per_rank_ms:
            roof["per_rank_kernel_ms_per_step"] = per_rank_ms
        # the same K steps as the launch stream's own events saw them (max over ranks): no host clock, no process barrier
        roof["event_span_ms_per_step"] = round(span_ms / max(args.steps, 1), 4)
        roof["event_span_gcells"] = round(cells_total / (span_ms * 1e-3) / 1e9, 4) if span_ms > 0 else None
        if valu and n == 512 and k_s > 0:
            # second bound: wave-instructions issued per step (SQ_INSTS_VALU of the committed PMC pass) against what the chip
            # issues in the measured kernel time: 256 CUs x 4 SIMDs, one wave64 FMA-pipe instruction per 2 cycles
            insts = float(valu["sq_insts_valu_per_step"]) * (k_cells / max(args.steps, 1)) / float(n) ** 3
            clock_ghz = float(valu.get("clock_ghz", 2.4))
            peak = 256 * 4 * clock_ghz * 1e9 / 2.0
            ach = insts / (k_s / max(args.steps, 1))
            out_valu = {"bound": "valu", "achieved": round(ach / 1e9, 1), "peak": round(peak / 1e9, 1), "unit": "G wave-instr/s",
                        "frac": round(ach / peak, 4), "cycles_per_instr": round(256 * 4 * clock_ghz * 1e9 / ach, 3),
                        "source": f"instruction count profile-derived ({valu.get('source')}), time from this run's events; "
                                  f"peak = 1024 SIMDs x {clock_ghz} GHz / 2 cycles per wave64 instruction"}
        else:
            out_valu = None
        out = {"metric": "Gcell-updates/s, 3D hypersonic 512^3 fp32", "value": round(value, 4),
               "unit": "Gcell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"tau_hypersonic_3d {n}^3 fp32, sphere r=0.25, Mach-100 inflow, "
                                      f"developed-flow start (SURVEY 8d input ii)",
                          "grid": [n, n, n], "decomposition": f"z-slab x{world}" if use_ring else "single domain",
                          "halo_planes": 3, "t": clk.t, "d_tau": clk.d_tau, "maxs": clk.maxs,
                          # which body of the kernels the timed steps ran (tau3d_field_range): the common-denominator WENO weights
                          # while every |primitive| <= 2.5e3, else the reciprocal form
                          "weno_form": "fast (common denominator)" if frange[2] else "reciprocal",
                          "max_abs_primitive": round(float(max(frange[0], frange[1])), 1) if max(frange[0], frange[1]) < 3e38 else "inf",
                          "timed_steps_after_start": [args.warmup, args.warmup + args.steps],
                          # the impulsive start runs away after ~55 steps (in the reference's kernel as here): a timed window that
                          # crosses into the runaway is not the workload — state at the END of the window: fast form still
                          # taken, finite wavespeed, every |primitive| <= 6e4
                          "state_sane": state_sane,
                          # uniform-region exits (include/taueng.h): the share of k_flux_xy's tiles (rank 0's planes) whose cells all held
                          # the undisturbed inflow state in the last timed step — their x/y divergence is exactly zero and was not
                          # computed; other_inputs.headline_uniform_exits_off is the same input with every face evaluated
                          "uniform_exits": bool(utiles[2]), "uniform_tile_fraction": round(utiles[0] / max(utiles[1], 1), 4),
                          # ... and of the tiles that ARE uniform, most are known to be from the flags of the step before
                          # (tau3d_tile_list_stats): k_flux_xy is launched over the list of the others (N > 1: rank 0's slab)
                          "tile_list": tlist},
               "roofline": roof}
        if out_valu:
            out["roofline_valu"] = out_valu
        if ring_info:
            out["ring"] = ring_info
            if verify is not None:
                out["ring"].update(verify)
            else:
                out["ring"]["matches_single_domain"] = None     # --no-verify
            if per_rank:
                out["ring"]["per_rank_event_ms_per_step"] = per_rank
                out["ring"]["per_rank_event_note"] = ("xy / z: kernels on the slab handle's stream; exchange / allreduce: the ring's "
                                                      "communication stream (the all-reduce waits for the slowest rank's exchange)")
        if world == 1 and not args.no_variants and not use_ring:
            try:
                out["other_inputs"] = input_variants(f, torch, dev, n)
                late = out["other_inputs"].get("developed_late")
                if late:   # the harder number, always beside the headline: the late developed state (ramped start, 2500 steps)
                    out["value_late"] = late["value"]
                    out["value_late_state_sane"] = late.get("state_sane")
                off = out["other_inputs"].get("headline_uniform_exits_off")
                if off:    # every face of every cell evaluated (what the reference's k_step does): the number the exits do not touch
                    out["value_uniform_exits_off"] = off["value"]
            except Exception as e:  # extras never take the headline down
                out["other_inputs"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline and not use_ring:
            planes = 8
            zc = n // 2 - 40 if n >= 128 else 0
            planes = min(planes, n)
            h.fill_halo_periodic_async()                   # the sample may reach into the halo planes: make them current
            h.sync()
            st = h.download_planes(zc - 3, zc + planes + 3)
            cpu_sample = ([np.ascontiguousarray(a) for a in st], clk.dt, zc, planes)
        if world == 1 and not args.no_configs and not use_ring:
            h.close()                                      # the 512^3 state is not needed below
            h = None
            try:
                out["configs"] = other_configs(f, torch, dev)
            except Exception as e:  # extras never take the headline down
                out["configs"] = [{"error": str(e)}]
        if world == 1 and not args.no_cpu_baseline and not use_ring:
            # north_star names tau_hypersonic_simd.c as the CPU baseline: the restated program (fluid-sims_amd/cpu/), 300^2 as
            # shipped, one thread.  The 3D oracle on a slab of the headline state — the same workload — is kept beside it.
            try:
                out["cpu_baseline"] = cpu_baseline_2d()
                out["cpu_baseline_2d_simd_256"] = cpu_baseline_2d(n=256)      # BASELINE config 1 size
                out["cpu_baseline_2d_simd_all_cores"] = cpu_baseline_2d_all_cores()
            except Exception as e:
                out["cpu_baseline"] = {"error": str(e)}
            try:
                out["cpu_baseline_3d_oracle"] = cpu_baseline(n, *cpu_sample)
            except Exception as e:
                out["cpu_baseline_3d_oracle"] = {"error": str(e)}
        try:   # RCCL's version banner (NCCL_DEBUG=VERSION on the GPU boxes) sits in the C stdio buffer until exit: push it out
            import ctypes as _ct   # first, so that the JSON line is the LAST line of stdout
            _ct.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)

    rc = 0
    if not state_sane:   # every rank sees the same all-reduced clock block
        if rank == 0:
            print(f"bench.py: the timed window [{args.warmup}, {args.warmup + args.steps}) crossed into the runaway of the impulsive "
                  f"start (fast form {bool(frange[2])}, maxs {clk.maxs}, max |primitive| {max(frange[0], frange[1])}): not a valid "
                  f"measurement — use fewer steps / less warm-up", file=sys.stderr)
        rc = 3
    if verify is not None and not verify["matches_single_domain"]:
        if rank == 0:
            print(f"bench.py: the Z-slab run does NOT reproduce the single-domain run ({verify}): the number above is not a valid "
                  f"measurement", file=sys.stderr)
        rc = 4
    try:                 # tear down on every path: a non-zero exit still closes the ring and the process group
        if use_ring and closer is not None:
            closer()
        if need_pg:
            dist.destroy_process_group()
    finally:
        if rc:
            sys.exit(rc)


if __name__ == "__main__":
    main()
