"""Shared glue for the parity tests: build the device sim and the CPU oracle
from the same config/scene and compare every output buffer bit for bit."""
import numpy as np

from gym_auv_amd import params
from gym_auv_amd.sim import AuvSim
from oracle import auv_oracle as O

# device buffer name -> oracle attribute
PAIRS = {
    "state": "state", "ranges": "ranges", "hit": "hit", "perception": "perception",
    "navigation": "navigation", "obst_dist": "obst_dist", "risk": "risk", "meta": "meta",
    "own": "own", "collision": "collision", "reached_goal": "reached_goal",
    "truncated": "truncated", "done": "done", "min_dist": "min_dist", "min_idx": "min_idx",
    "along": "along", "t_step": "t_step", "circles": "circles", "poly_pose": "poly_pose",
    "reward": "reward", "episode": "episode",
}


def make_oracle(cfg, scene):
    env = O.OracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"],
                  scene.get("poly_heading"))
    env.set_path(scene["path"])
    return env


def make_sim(cfg, scene, device=0, lidar_mode=None):
    sim = AuvSim(cfg, device=device)
    if lidar_mode is not None:
        from gym_auv_amd import _lib

        sim.set_option(_lib.OPT_LIDAR_MODE, lidar_mode)
    sim.set_scene(scene)
    return sim


def same_bits(a, b):
    """Exact equality, with -0.0 == +0.0 and equal infinities allowed."""
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and bool(np.array_equal(a, b))


def assert_all_equal(sim, env, where=""):
    bad = []
    for name, attr in PAIRS.items():
        got = sim.read(name)
        want = getattr(env, attr)
        if not same_bits(got, np.asarray(want).reshape(got.shape)):
            w = np.asarray(want).reshape(got.shape)
            nbad = int((got != w).sum())
            bad.append(f"{name}: {nbad}/{got.size} differ")
    assert not bad, f"device != oracle {where}: " + "; ".join(bad)
