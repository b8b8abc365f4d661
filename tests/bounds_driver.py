"""Run under AUVSIM_LIB=<the AUV_DEBUG_BOUNDS build>: stress scenes through the
kernels, then print the number of violated on-chip index bounds (must be 0)."""
import ctypes as C
import sys

import numpy as np

from gym_auv_amd import _lib, cnn, params, scenarios
from gym_auv_amd.sim import AuvSim


def run(cfg, scene, steps, tile=0, block=0, gen=None):
    sim = AuvSim(cfg)
    if tile:
        sim.set_option(_lib.OPT_RASTER_TILE, tile)
    if block:
        sim.set_option(_lib.OPT_SENSE_BLOCK, block)
    if gen is not None:
        sim.generate_scene(*gen)
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
        sim.set_option(_lib.OPT_REGENERATE, 1)
        sim.reset(None)
    else:
        sim.set_scene(scene)
        sim.reset(scene["init_state"])
    V = cfg["n_envs"]
    acts = scenarios.random_actions(V, steps, seed=3)
    acts[:, 0] = 1.0
    for k in range(steps):
        sim.step_host(acts[k])
    sim.sync()
    sim.close()


def main():
    lib = _lib.load()
    if not hasattr(lib, "auv_debug_violations"):
        print("not a debug build")
        return 2
    # crowded scenes: vessels inside hulls, pair lists that overflow, every tile / workgroup shape
    for V, R, Cn, P, NV, ns, tile, block in [(300, 180, 8, 8, 5, None, 0, 0), (130, 1000, 2, 3, 7, 7, 1, 512),
                                             (67, 180, 16, 5, 16, 16, 0, 64), (259, 90, 7, 1, 4, 4, 5, 128),
                                             (1000, 64, 1, 1, 3, 3, 64, 64), (40, 720, 128, 128, 8, 8, 0, 0),
                                             (33, 720, 40, 40, 8, 8, 2, 1024),
                                             (1027, 180, 8, 8, 5, None, 4, 256)]:   # the compile-time shape, narrow LDS image
        cfg = params.make_config(V, Cn, P, NV, n_rays=R, max_steps=6)
        scene = scenarios.make_scene(V, Cn, P, NV, seed=V, extent=50.0, n_sides=ns, width=(6.0, 40.0))
        scene["circles"][:, :, 2] *= 2.0          # giant circles: most vessels start inside one
        run(cfg, scene, 8, tile, block)
    cfg = params.make_config(4099, 8, 8, 5, max_steps=7)
    run(cfg, None, 30, gen=(5, dict(extent=60.0, clearance=0.5, r_hi=60.0)))
    v = C.c_int64(-1)
    lib.auv_debug_violations.argtypes = [C.POINTER(C.c_int64)]
    rc = lib.auv_debug_violations(C.byref(v))
    print("violations", v.value, "rc", rc)
    return 0 if (rc == 0 and v.value == 0) else 1


if __name__ == "__main__":
    sys.exit(main())
