import sys, time
sys.path.insert(0, '.')
import fluid_sims_amd as f
n = 4096
e = f.Hypersonic2D(n, n); e.init(); e.step_async(50); e.sync()
e.step_async(100); e.sync(); e.close()
