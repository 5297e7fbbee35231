import os, sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo')
import fluid_sims_amd as f
def run(env, body, mode, warm, label):
    for k in ('TAU3D_TILE_LIST','TAU3D_Z_SKIP','TAU3D_UNIFORM_EXITS','TAU3D_Z_EXP'): os.environ.pop(k, None)
    os.environ.update(env)
    p = f.Tau3DParams(); f.load().tau3d_params_default(ctypes.byref(p), 512, 512, 512)
    if not body: p.sdf_r = -1.0
    e = f.Tau3D(512, 512, 512, params=p)
    e.init(mode)
    if mode: e.set_clock(0.02, 1e-4)
    e.step(warm)
    e.timing_enable(True)
    e.step(20)
    r = e.timing_read_split()
    print(label, env, 'xy %.3f z %.3f ms' % (r[0]/r[2], r[1]/r[2]), 'list', e.tile_list_stats()[1:3], 'uniform', e.uniform_tiles()[:2], flush=True)
    e.close()
for env in ({}, {'TAU3D_Z_EXP':'1'}, {'TAU3D_Z_EXP':'8'}, {'TAU3D_Z_EXP':'4'}):
    run(env, False, 1, 10, 'no body, developed start')
