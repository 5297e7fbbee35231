import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes as C
import fluid_sims_amd as f
dev = torch.device("cuda", 0)
mode = os.environ.get("TAU3D_PERSIST", "auto")
out = {}
for n in (32, 48, 64, 96, 128):
    stream = torch.cuda.Stream(dev)
    e = f.Tau3D(n, stream=C.c_void_p(stream.cuda_stream))
    e.init(1); e.set_clock(0.02, 1e-4); e.step_async(50); e.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); e.step_async(200); e1.record(stream); e1.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    e.step_async(3); e.step_async(1); e.step_async(2)
    c = e.clock(); st = e.download()
    print(f"persist={mode} n={n}: {us:.1f} us/step ({n**3/us/1e3:.2f} Gcell/s) step {c.step} t {c.t:.9g} d_tau {c.d_tau:.9g} maxs {c.maxs:.9g}", flush=True)
    np.savez(f"/tmp/pers_{mode}_{n}.npz", *st, clk=np.array([c.t, c.d_tau, c.maxs, c.step]))
    e.close()
