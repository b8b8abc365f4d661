"""Moving obstacles against the reference's own data: advance, the five logged
columns and the risk, through the simulation.

`tests/golden/ship_golden.npz` (made by make_ship_golden.py from
data/metadata_MovingObstaclesNoRules.csv and risk_MovingObstaclesNoRules.csv)
holds 134 moving obstacles, each logged in 4 consecutive rows through which the
vessel's state and command are known (windows of turn_golden.npz).  Per
obstacle the fixture carries its heading, its speed s and its size (the
constant K - 3.255 the reference subtracts from the centre distance; an integer
3 .. 18 m for all of them), recovered from 6 equations in 2 unknowns.

Here one env per obstacle gets a ship of that beam where the first row saw it,
sailing along its heading at s, and the vessel runs the 15 steps.  At every
logged row the env's metadata record [heading_ts, dist, speed, bearing, course]
(utils/explore_dataset.py:45) and its risk have to be the logged ones: 402 rows
x 6 numbers that test the obstacle advance, the vessel track, the reference's
conventions for every column and its risk index as evaluated inside the step."""
import os

import numpy as np

from gym_auv_amd import params
from oracle import auv_oracle as O
from oracle import c_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ship_golden.npz")
ROWS, LOG_EVERY = 4, 5
TOL = {"heading_ts": 1e-12, "dist": 2e-8, "speed": 1e-9, "bearing": 1e-9, "course": 1e-8, "risk": 1e-9}


def ships():
    g = np.load(GOLDEN)
    return {k: g[k] for k in g.files}


def scene_for(g, cfg):
    """One ship per env: the reference's five-vertex hull of beam w (stern at
    -w/2, shoulders at +w/2, bow at 3w/2), centred where row 0 saw it."""
    n = len(g["offset"])
    w = g["offset"] - cfg["vessel_radius"]
    rec0 = g["records"][:, 0]
    ang = rec0[:, 3] + g["state"][2]
    rng = rec0[:, 1] + g["offset"]
    h = g["heading_ts"]
    pose = np.zeros((n, 1, 6))
    pose[:, 0, 0] = rng * np.cos(ang)
    pose[:, 0, 1] = rng * np.sin(ang)
    pose[:, 0, 2], pose[:, 0, 3] = np.cos(h), np.sin(h)
    pose[:, 0, 4], pose[:, 0, 5] = g["speed_ts"] * np.cos(h), g["speed_ts"] * np.sin(h)
    verts = np.zeros((n, 1, 5, 2))
    for k, (x, y) in enumerate([(-0.5, -0.5), (0.5, -0.5), (1.5, 0.0), (0.5, 0.5), (-0.5, 0.5)]):
        verts[:, 0, k, 0], verts[:, 0, k, 1] = x * w, y * w
    psi = g["state"][2]
    path = np.zeros((n, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = psi, np.cos(psi), np.sin(psi), 1e6
    return {"circles": np.zeros((n, 0, 5)), "poly_pose": pose, "poly_verts": verts,
            "poly_nv": np.full((n, 1), 5, dtype=np.int32), "poly_heading": h[:, None].copy(), "path": path}


def check_row(g, k, own, meta, risk):
    rec = g["records"][:, k]
    assert np.abs(own[:, 0] - g["own"][:, k, 0]).max() <= 1e-13
    assert np.abs(O.wrap_angle(own[:, 1] - g["own"][:, k, 1])).max() <= 1e-13
    for c, name in enumerate(("heading_ts", "dist", "speed", "bearing", "course")):
        diff = meta[:, 0, c] - rec[:, c]
        if name in ("heading_ts", "bearing", "course"):
            diff = O.wrap_angle(diff)
        assert np.abs(diff).max() <= TOL[name], (k, name, np.abs(diff).max())
    assert np.abs(risk[:, 0] - g["risk"][:, k]).max() <= TOL["risk"], (k, np.abs(risk[:, 0] - g["risk"][:, k]).max())


def test_fixture_shape_and_integer_sizes():
    g = ships()
    n = len(g["offset"])
    assert n == 134 and g["records"].shape == (n, ROWS, 5) and g["risk"].shape == (n, ROWS)
    size = g["offset"] - 3.255
    assert np.abs(size - np.rint(size)).max() < 1e-9 and np.rint(size).min() == 3 and np.rint(size).max() == 18
    assert g["speed_ts"].min() > 0.05 and g["speed_ts"].max() < 1.3
    assert (g["risk"] > 0.3).sum() > 50 and (g["records"][:, :, 1] < 30.0).sum() > 20      # close encounters among them


def test_numpy_oracle_reproduces_the_logged_ship_records():
    g = ships()
    n = len(g["offset"])
    cfg = params.make_config(n, 0, 1, 5)
    scene = scene_for(g, cfg)
    env = O.OracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.reset(g["state"].copy())
    check_row(g, 0, env.own, env.meta, env.risk)
    act = np.stack([g["surge"], g["steer"]])
    for k in range(1, ROWS):
        for _ in range(LOG_EVERY):
            env.step(act)
        check_row(g, k, env.own, env.meta, env.risk)


def test_c_oracle_reproduces_the_logged_ship_records():
    g = ships()
    n = len(g["offset"])
    cfg = params.make_config(n, 0, 1, 5)
    scene = scene_for(g, cfg)
    env = c_oracle.COracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.reset(g["state"].copy())
    act = np.stack([g["surge"], g["steer"]])
    for k in range(1, ROWS):
        for _ in range(LOG_EVERY):
            env.step(act)
        check_row(g, k, env.own, env.meta, env.risk)
    env.close()
