import sys; sys.path.insert(0,'.')
import numpy as np
from gym_auv_amd import _lib, scenarios
from tests.test_parity_gpu import _adversarial_scene
from tests.helpers import make_sim
V=20000
cfg, scene = _adversarial_scene(V, 101)
res=[]
for mode in (0,1):
    sim = make_sim(cfg, scene, lidar_mode=mode); sim.reset(scene["init_state"])
    res.append((sim.read("ranges"), sim.read("hit"))); sim.close()
(r0,h0),(r1,h1)=res
bad = (r0!=r1)|(h0!=h1)
print("bad rays", bad.sum(), "envs", bad.any(1).sum())
envs=np.nonzero(bad.any(1))[0][:8]
np.set_printoptions(precision=6, linewidth=200)
for e in envs:
    rays=np.nonzero(bad[e])[0]
    print("env",e,"rays",rays[:10], "raster", r0[e,rays[:5]], h0[e,rays[:5]], "brute", r1[e,rays[:5]], h1[e,rays[:5]])
    hb=np.unique(h1[e,rays]); print("  brute ids", hb, " raster ids", np.unique(h0[e,rays]))
    for o in hb:
        if o<6: print("  circle",o, scene["circles"][e,o])
        else: print("  poly",o, scene["poly_pose"][e,o-6], scene["poly_nv"][e,o-6], scene["poly_verts"][e,o-6].ravel())
    print("  psi", scene["init_state"][2,e])
