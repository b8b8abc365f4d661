"""Synthetic MovingObstaclesNoRules-style scenes, generated on the host.

A scene is a dict of numpy arrays (V envs):
  circles      [V,C,5]   cx, cy, radius, vx, vy       (radius <= 0: absent)
  poly_pose    [V,P,6]   X, Y, cos h, sin h, vx, vy
  poly_verts   [V,P,NV,2] body-frame vertices, counter-clockwise
  poly_nv      [V,P]     vertices used (0: absent)
  poly_heading [V,P]     h (reported as heading_ts in the metadata rows)
  init_state   [6,V]     x, y, psi, u, v, r
  path         [V,6]     x0, y0, angle, cos angle, sin angle, length
All generators take a seed for numpy.random.default_rng; generation and
upload are never part of a timed region.
"""
from __future__ import annotations

import numpy as np

from .params import make_config


def ship_polygon(width):
    """Five-vertex ship outline of the given beam (array broadcast over the
    leading dims): a square stern section with a pointed bow, centred so the
    pose position is mid-hull.  Returns [...,5,2], counter-clockwise."""
    w = np.asarray(width, dtype=np.float64)[..., None]
    x = np.concatenate([-w, w * 0.0, w, w * 0.0, -w], axis=-1)
    y = np.concatenate([-w / 2, -w / 2, w * 0.0, w / 2, w / 2], axis=-1)
    return np.stack([x, y], axis=-1)


def ngon(radii):
    """Star-shaped polygon from per-vertex radii [...,N] at equal angles."""
    r = np.asarray(radii, dtype=np.float64)
    n = r.shape[-1]
    a = np.arange(n) * (2.0 * np.pi / n)
    return np.stack([r * np.cos(a), r * np.sin(a)], axis=-1)


def _push_out(cx, cy, clear):
    """Move centres radially so that they are at least `clear` from the origin."""
    d = np.sqrt(cx * cx + cy * cy)
    d = np.where(d == 0.0, 1.0, d)
    f = np.where(d < clear, clear / d, 1.0)
    return cx * f, cy * f


def _straight_path(V, angle, length=800.0):
    path = np.zeros((V, 6))
    path[:, 2] = angle
    path[:, 3] = np.cos(angle)
    path[:, 4] = np.sin(angle)
    path[:, 5] = length
    return path


def waypoint_path(points, n_points=None):
    """Segment records [V,K-1,6] = (x0, y0, angle, cos, sin, length) of the
    polylines through `points` [V,K,2] (what `AuvSim.set_path` and the oracles'
    `set_path` take), and the number of segments in use per env ([V]; from
    `n_points` [V] waypoints in use, default all K).  Waypoints in use must
    differ from their successors (a segment without length is refused at
    upload) and should turn by less than 90 degrees (a vessel leaves a segment
    when its projection passes the segment's end)."""
    pts = np.asarray(points, dtype=np.float64)
    d = pts[:, 1:] - pts[:, :-1]
    length = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])
    safe = np.where(length > 0.0, length, 1.0)
    ca, sa = d[..., 0] / safe, d[..., 1] / safe
    seg = np.stack([pts[:, :-1, 0], pts[:, :-1, 1], np.arctan2(d[..., 1], d[..., 0]), ca, sa, length], axis=-1)
    n = np.full(len(pts), pts.shape[1] - 1, dtype=np.int32) if n_points is None else np.asarray(n_points, dtype=np.int32) - 1
    return np.ascontiguousarray(seg), n


def _moving_polys(rng, V, P, NV, extent, width_lo, width_hi, speed_lo, speed_hi, n_sides=None):
    X = rng.uniform(-extent, extent, (V, P))
    Y = rng.uniform(-extent, extent, (V, P))
    h = rng.uniform(-np.pi, np.pi, (V, P))
    spd = rng.uniform(speed_lo, speed_hi, (V, P))
    w = rng.uniform(width_lo, width_hi, (V, P))
    X, Y = _push_out(X, Y, 2.0 * w + 25.0)
    ch, sh = np.cos(h), np.sin(h)
    pose = np.stack([X, Y, ch, sh, spd * ch, spd * sh], axis=-1)
    if n_sides is None:
        verts = ship_polygon(w)
    else:
        verts = ngon(w[..., None] * rng.uniform(0.6, 1.0, (V, P, n_sides)))
    nv = np.full((V, P), verts.shape[2], dtype=np.int32)
    if verts.shape[2] < NV:
        pad = np.zeros((V, P, NV - verts.shape[2], 2))
        verts = np.concatenate([verts, pad], axis=2)
    return pose, np.ascontiguousarray(verts), nv, h


def _circles(rng, V, C, extent, r_lo=5.0, r_hi=30.0):
    cx = rng.uniform(-extent, extent, (V, C))
    cy = rng.uniform(-extent, extent, (V, C))
    rad = rng.uniform(r_lo, r_hi, (V, C))
    cx, cy = _push_out(cx, cy, rad + 25.0)
    z = np.zeros((V, C))
    return np.stack([cx, cy, rad, z, z], axis=-1)


def make_scene(V, C, P, NV=5, seed=0, extent=200.0, n_sides=None, moving=True,
               width=(5.0, 20.0), speed=(0.3, 1.2)):
    """Vessel at the origin with a random heading; C static circles and P
    polygons (moving ships by default) scattered over [-extent, extent]^2."""
    rng = np.random.default_rng(seed)
    circles = _circles(rng, V, C, extent) if C else np.zeros((V, 0, 5))
    if P:
        pose, verts, nv, h = _moving_polys(rng, V, P, NV, extent, width[0], width[1],
                                           speed[0] if moving else 0.0,
                                           speed[1] if moving else 0.0, n_sides)
    else:
        pose, verts = np.zeros((V, 0, 6)), np.zeros((V, 0, NV, 2))
        nv, h = np.zeros((V, 0), dtype=np.int32), np.zeros((V, 0))
    psi = rng.uniform(-np.pi, np.pi, V)
    init = np.zeros((6, V))
    init[2] = psi
    return {
        "circles": np.ascontiguousarray(circles), "poly_pose": np.ascontiguousarray(pose),
        "poly_verts": verts, "poly_nv": nv, "poly_heading": h,
        "init_state": init, "path": _straight_path(V, psi),
    }


# Generator parameters that follow what the reference's logged rows show of its scenario
# (data/metadata_MovingObstaclesNoRules.csv; sizes and speeds recovered by tests/golden/make_scan_golden.py):
# whole-number circle radii 15..41 m and ship sizes 1..18 m with bell-shaped laws, ship speeds up to ~0.9 m/s
# peaking near 0.5, and an obstacle density that lists 2-3 obstacles within the 150 m sensor range on average
# (16 obstacles over an 850 m square).
REFERENCE_LIKE = dict(extent=425.0, r_lo=15.0, r_hi=41.0, w_lo=1.0, w_hi=18.0, s_lo=0.05, s_hi=0.9, clearance=25.0,
                      path_length=800.0, size_law=1)

# the five benchmark configurations: vessels, rays, scene, seed
WORKLOADS = {
    "cfg1_single_episode": dict(V=1, R=180, C=17, P=17, NV=5, seed=0, extent=400.0),
    "cfg2_4096_static": dict(V=4096, R=180, C=16, P=0, NV=5, seed=1, extent=200.0),
    "cfg3_65536_moving": dict(V=65536, R=180, C=8, P=8, NV=5, seed=2, extent=200.0),
    "cfg4_8x65536_moving": dict(V=65536, R=180, C=8, P=8, NV=5, seed=3, extent=200.0),
    "cfg5_dense_720": dict(V=4096, R=720, C=128, P=128, NV=8, seed=4, extent=400.0,
                           n_sides=8),
}


def make_workload(name, n_envs=None, rank=0):
    """(config dict, scene dict) for one of WORKLOADS; `n_envs` overrides the
    vessel count (parity tests run the same scene family at small sizes) and
    `rank` offsets the seed so that every GPU simulates different envs."""
    w = dict(WORKLOADS[name])
    V = int(n_envs) if n_envs is not None else w["V"]
    cfg = make_config(V, w["C"], w["P"], w["NV"], n_rays=w["R"])
    scene = make_scene(V, w["C"], w["P"], w["NV"], seed=w["seed"] + 1000 * rank,
                       extent=w["extent"], n_sides=w.get("n_sides"))
    return cfg, scene


def random_actions(V, steps, seed=0):
    """[steps,2,V] actions ~ U([0,1] x [-1,1])."""
    rng = np.random.default_rng(seed)
    a = np.empty((steps, 2, V))
    a[:, 0] = rng.uniform(0.0, 1.0, (steps, V))
    a[:, 1] = rng.uniform(-1.0, 1.0, (steps, V))
    return a
