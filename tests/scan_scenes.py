"""Scenes of the reference's logged rows, rebuilt from tests/golden/scan_golden.npz
(made by tests/golden/make_scan_golden.py from the reference's metadata file).

One env per logged row: the vessel at the origin with the logged heading, every
listed obstacle at its logged bearing and distance with its recovered size --
a static obstacle as a circle of that radius, a moving one as a five-vertex
ship of that beam headed along heading_ts.  Casting the LiDAR over such an env
gives the scan the reference's sensor saw when it logged the row, up to the
conventions under test (ray order, ship outline).
"""
import os

import numpy as np

from gym_auv_amd import params

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scan_golden.npz")
CLEARANCE = 3.255
SLOTS = 8   # obstacles per logged row: at most 8 (utils/dataloader.py:108)


def ship_outline(w, kind="centred"):
    """Five-vertex hull of beam w, 2 w long, counter-clockwise, [...,5,2].
    "centred": the pose is mid-hull (gym_auv_amd.scenarios.ship_polygon);
    "stern": the pose is the centre of the square stern section, the bow 1.5 w ahead of it."""
    w = np.asarray(w, dtype=np.float64)[..., None]
    z = w * 0.0
    if kind == "centred":
        x = np.concatenate([-w, z, w, z, -w], axis=-1)
    elif kind == "stern":
        x = np.concatenate([-w / 2, w / 2, 1.5 * w, w / 2, -w / 2], axis=-1)
    else:
        raise ValueError(kind)
    y = np.concatenate([-w / 2, -w / 2, z, w / 2, w / 2], axis=-1)
    return np.stack([x, y], axis=-1)


def ray_table(n_rays=180, shift=0.0, mirrored=False):
    """The build's ray table (gym_auv_amd.params.ray_table) turned by `shift`
    rays, or run clockwise from the same start."""
    i = np.arange(n_rays, dtype=np.float64)
    step = 2.0 * np.pi / n_rays
    ang = -np.pi + (i + 1.0 + shift) * step
    if mirrored:
        ang = -ang
    return np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1))


def load(rows="all"):
    """The fixture; rows="test" keeps the rows of the reference's test split
    (the last 30 % of the file, utils/dataloader.py:63-67)."""
    g = dict(np.load(GOLDEN))
    keep = np.ones(len(g["row"]), dtype=bool)
    if rows == "test":
        keep = g["row"] >= int(0.7 * int(g["n_rows_file"]))
    g["keep"] = keep
    return g


def build(g, outline="centred"):
    """(cfg, scene, labels) of the kept rows."""
    idx = np.where(g["keep"])[0]
    V = len(idx)
    circles = np.zeros((V, SLOTS, 5))
    pose = np.zeros((V, SLOTS, 6))
    pose[:, :, 2] = 1.0
    verts = np.zeros((V, SLOTS, 5, 2))
    nv = np.zeros((V, SLOTS), dtype=np.int32)
    heading = np.zeros((V, SLOTS))
    psi = g["heading_os"][idx]
    for e, r in enumerate(idx):
        lo, hi = int(g["first"][r]), int(g["first"][r + 1])
        nc = npol = 0
        for o in range(lo, hi):
            size = float(g["size"][o])
            D = g["dist"][o] + size + CLEARANCE
            a = g["bearing"][o] + psi[e]
            cx, cy = D * np.cos(a), D * np.sin(a)
            if g["static"][o]:
                circles[e, nc] = (cx, cy, size, 0.0, 0.0)
                nc += 1
            else:
                h = g["heading_ts"][o]
                pose[e, npol] = (cx, cy, np.cos(h), np.sin(h), 0.0, 0.0)
                verts[e, npol] = ship_outline(size, outline)
                nv[e, npol] = 5
                heading[e, npol] = h
                npol += 1
    init = np.zeros((6, V))
    init[2] = psi
    path = np.zeros((V, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = psi, np.cos(psi), np.sin(psi), 1e6
    cfg = params.make_config(V, SLOTS, SLOTS, 5)
    scene = {"circles": circles, "poly_pose": pose, "poly_verts": verts, "poly_nv": nv, "poly_heading": heading,
             "init_state": init, "path": path}
    return cfg, scene, g["label"][idx]


def reorder(scans, shift=0, mirrored=False):
    """The scans [..., R] a ray table turned by `shift` whole rays (and, with
    `mirrored`, run clockwise) would have produced: the same set of rays in
    another order, so no new cast is needed."""
    R = scans.shape[-1]
    i = np.arange(R)
    j = (R - 2 - i - shift) % R if mirrored else (i + shift) % R
    return scans[..., j]


def split_masks(g):
    """Boolean masks (train, validation, test) over the kept rows: the sequential
    70/30 -> 70/30 split of utils/dataloader.py:63-67."""
    rows = g["row"][g["keep"]]
    n = int(g["n_rows_file"])
    n_tr_val = int(0.7 * n)
    n_tr = int(0.7 * n_tr_val)
    return rows < n_tr, (rows >= n_tr) & (rows < n_tr_val), rows >= n_tr_val
