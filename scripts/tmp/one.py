import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import fluid_sims_amd as f
p = f.Tau3DParams(); f.load().tau3d_params_default(ctypes.byref(p), 512, 512, 512)
p.sdf_r = -1.0
e = f.Tau3D(512, 512, 512, params=p)
e.init(1); e.set_clock(0.02, 1e-4)
e.step(30)
print(e.tile_list_stats(), e.uniform_tiles())
e.close()
