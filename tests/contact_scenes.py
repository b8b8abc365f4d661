"""Scenes of the reference's near-contact records (tests/golden/contact_golden.npz,
made by tests/golden/make_contact_golden.py from its metadata file): one env per
record, the vessel at the origin with the logged heading and the one obstacle
where the record puts it, a static one as a circle of the recovered radius, a
moving one as a ship of the recovered beam headed along heading_ts."""
import os

import numpy as np

from gym_auv_amd import params, scenarios

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "contact_golden.npz")
CLEARANCE = 3.255


def load():
    return dict(np.load(GOLDEN))


def vessel_in_obstacle_frame(g):
    """(ahead, abeam) of the vessel seen from every obstacle, metres, in the obstacle's body frame."""
    rec = g["record"]
    D = rec[:, 1] + g["size"] + CLEARANCE
    a = rec[:, 3] + g["heading_os"]                       # world direction vessel -> obstacle
    rx, ry = -D * np.cos(a), -D * np.sin(a)
    h = rec[:, 0]
    return rx * np.cos(h) + ry * np.sin(h), -rx * np.sin(h) + ry * np.cos(h)


def build(g, hull=None, contact_dist=None, places=None):
    """(cfg, scene) with one env per record.  hull(w) -> [n,5,2] body-frame vertices (default
    scenarios.ship_polygon); places: (ahead, abeam) arrays replacing the logged places of the vessel in the
    obstacle's frame (the obstacle is then put accordingly; static records keep their logged place)."""
    n = len(g["row"])
    psi, rec, size = g["heading_os"], g["record"], g["size"].astype(np.float64)
    D = rec[:, 1] + size + CLEARANCE
    a = rec[:, 3] + psi
    cx, cy = D * np.cos(a), D * np.sin(a)
    h = rec[:, 0]
    if places is not None:
        ahead, abeam = places
        mv = ~g["static"]
        # vessel at the origin: obstacle centre = -(ahead * heading + abeam * left normal)
        cx = np.where(mv, -(ahead * np.cos(h) - abeam * np.sin(h)), cx)
        cy = np.where(mv, -(ahead * np.sin(h) + abeam * np.cos(h)), cy)
    circles = np.zeros((n, 1, 5))
    pose = np.zeros((n, 1, 6))
    pose[:, :, 2] = 1.0
    verts = np.zeros((n, 1, 5, 2))
    nv = np.zeros((n, 1), dtype=np.int32)
    heading = np.zeros((n, 1))
    st = g["static"]
    circles[st, 0, 0], circles[st, 0, 1], circles[st, 0, 2] = cx[st], cy[st], size[st]
    mv = ~st
    pose[mv, 0, 0], pose[mv, 0, 1], pose[mv, 0, 2], pose[mv, 0, 3] = cx[mv], cy[mv], np.cos(h[mv]), np.sin(h[mv])
    verts[mv, 0] = (scenarios.ship_polygon if hull is None else hull)(size[mv])
    nv[mv, 0] = 5
    heading[mv, 0] = h[mv]
    init = np.zeros((6, n))
    init[2] = psi
    path = np.zeros((n, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = psi, np.cos(psi), np.sin(psi), 1e6
    cfg = params.make_config(n, 1, 1, 5) if contact_dist is None else params.make_config(n, 1, 1, 5, contact_dist=contact_dist)
    return cfg, {"circles": circles, "poly_pose": pose, "poly_verts": verts, "poly_nv": nv, "poly_heading": heading,
                 "init_state": init, "path": path}


def passes(g):
    """Runs of consecutive rows about the same obstacle: lists of record numbers."""
    out, cur = [], [0]
    for i in range(1, len(g["row"])):
        same = g["row"][i] == g["row"][i - 1] + 1 and g["size"][i] == g["size"][i - 1] and g["static"][i] == g["static"][i - 1]
        if same:
            cur.append(i)
        else:
            out.append(cur)
            cur = [i]
    out.append(cur)
    return out
