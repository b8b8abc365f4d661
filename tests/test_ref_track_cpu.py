"""Position integration, obstacle distance and bearing against the reference's
own data.

`tests/golden/track_golden.npz` (made by make_track_golden.py from
data/metadata_MovingObstaclesNoRules.csv) holds 5,230 sightings of a static
obstacle in two consecutive logged rows between which the vessel's state and
command are known (the rows covered by surge_golden.npz and turn_golden.npz).
Each is a sample of three things at once: how the reference integrates the
position over the 5 steps in between, that its `bearing` column
(utils/explore_dataset.py:8,45) is the body-frame direction of the obstacle's
centre, and that its `dist` column is the centre distance minus radius minus a
clearance.  The clearance is 3.255 m: with it every one of the radii comes out
as an integer (15 .. 41 m), which is what the fixture's `offset` shows.

Here one env per sighting gets a circle where the first row saw it, runs the 5
steps, and has to report the second row's dist and bearing to 1e-9, the rest
of its record, and the risk the reference logged for it, to 1e-9 (the
obstacle's position is itself derived from logged numbers, which costs a few
digits; the residual of the derivation is below 1e-8 m by construction)."""
import os

import numpy as np

from gym_auv_amd import params
from oracle import auv_oracle as O
from oracle import c_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "track_golden.npz")
TOL_DIST = 2e-8      # m, on distances of 5 .. 176 m
TOL_BEARING = 1e-9   # rad
TOL_RISK = 1e-9
LOG_EVERY = 5


def sightings():
    g = np.load(GOLDEN)
    return {k: g[k] for k in g.files}


def scene_for(g, cfg):
    """One circle per env, placed where row A saw it: centre distance
    dist + offset along bearing + heading; radius = offset - clearance."""
    n = len(g["offset"])
    ang = g["bearing_a"] + g["state"][2]
    rng = g["dist_a"] + g["offset"]
    circles = np.zeros((n, 1, 5))
    circles[:, 0, 0] = rng * np.cos(ang)
    circles[:, 0, 1] = rng * np.sin(ang)
    circles[:, 0, 2] = g["offset"] - cfg["vessel_radius"]
    psi = g["state"][2]
    path = np.zeros((n, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = psi, np.cos(psi), np.sin(psi), 1e6
    return {"circles": circles, "poly_pose": np.zeros((n, 0, 6)), "poly_verts": np.zeros((n, 0, 5, 2)),
            "poly_nv": np.zeros((n, 0), dtype=np.int32), "poly_heading": np.zeros((n, 0)), "path": path}


def check(g, obst_dist, meta, collision, risk=None, own=None):
    assert np.abs(obst_dist[:, 0] - g["dist_b"]).max() <= TOL_DIST
    assert np.abs(O.wrap_angle(meta[:, 0, 3] - g["bearing_b"])).max() <= TOL_BEARING
    assert np.abs(meta[:, 0, 1] - g["dist_b"]).max() <= TOL_DIST          # the logged dist column is that distance
    assert not collision.any()                                             # none of the logged sightings is a contact
    # the rest of the second row's record as the reference logged it for a static obstacle
    # (utils/explore_dataset.py:46: heading_ts 0, speed = speed_OS; course 0 or -pi), and its risk
    rec = g["record_b"]
    assert not meta[:, 0, 0].any() and not rec[:, 0].any()
    assert np.abs(meta[:, 0, 2] - rec[:, 2]).max() <= 1e-13
    assert np.array_equal(meta[:, 0, 4], rec[:, 4])
    if own is not None:
        assert np.array_equal(own[:, 0], meta[:, 0, 2])                   # speed column == speed_OS, bit for bit
    if risk is not None:
        assert np.abs(risk[:, 0] - g["risk_b"]).max() <= TOL_RISK


def test_fixture_shape_and_integer_radii():
    g = sightings()
    n = len(g["offset"])
    assert n == 5230 and g["state"].shape == (6, n)
    radius = g["offset"] - 3.255
    assert np.abs(radius - np.rint(radius)).max() < 1e-7
    assert np.rint(radius).min() == 15 and np.rint(radius).max() == 41
    assert params.make_config(1, 0, 0)["vessel_radius"] == 3.255
    assert (g["surge"] == 1.0).sum() > 3000 and (np.abs(g["state"][5]) > 1e-3).sum() > 500   # mostly under way, many turning


def test_numpy_oracle_tracks_the_reference_obstacles():
    g = sightings()
    n = len(g["offset"])
    cfg = params.make_config(n, 1, 0)
    scene = scene_for(g, cfg)
    env = O.OracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.reset(g["state"].copy())
    # before the first step the env reports row A's numbers
    assert np.abs(env.obst_dist[:, 0] - g["dist_a"]).max() <= 1e-10
    assert np.abs(O.wrap_angle(env.meta[:, 0, 3] - g["bearing_a"])).max() <= 1e-12
    act = np.stack([g["surge"], g["steer"]])
    for _ in range(LOG_EVERY):
        env.step(act)
    check(g, env.obst_dist, env.meta, env.collision, env.risk, env.own)


def test_c_oracle_tracks_the_reference_obstacles():
    g = sightings()
    n = len(g["offset"])
    cfg = params.make_config(n, 1, 0)
    scene = scene_for(g, cfg)
    env = c_oracle.COracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(scene["circles"], scene["poly_pose"], np.zeros((n, 0, 1, 2)), scene["poly_nv"], None)
    env.set_path(scene["path"])
    env.reset(g["state"].copy())
    act = np.stack([g["surge"], g["steer"]])
    for _ in range(LOG_EVERY):
        env.step(act)
    check(g, env.obst_dist, env.meta, env.collision, env.risk, env.own)
    env.close()
