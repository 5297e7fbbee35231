import sys, ctypes
sys.path.insert(0, '.')
import fluid_sims_amd as f
mode = sys.argv[1]
n, nzl = 512, 64
e = f.Tau3D(n, n, nzl); e.init(1); e.set_clock(0.02, 1e-4)
if mode == "plain":
    e.step(5); e.step_async(8); e.sync()
else:
    ring = f.Tau3DRing(e, 0, 1, {"ipc": f.RING_IPC, "local": f.RING_LOCAL, "rccl": f.RING_RCCL}[mode]); ring.prime()
    ring.step(5); ring.finish(); ring.step(8); ring.finish(); ring.close()
e.close()
