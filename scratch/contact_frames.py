"""Near-contact records of the reference's log in the obstacle's body frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import scan_scenes as S
g = S.load("all")
c = np.load(os.path.join(ROOT, "tests/golden/cri_golden.npz"))
row_of_kept = {int(r): i for i, r in enumerate(g["row"])}
first_rec = np.searchsorted(c["row"], np.arange(int(g["n_rows_file"]) + 1))
CLR = 3.255
print("row  kind size  dist   centre  | vessel in obstacle frame: ahead, abeam (m) | in sizes | hull(2w x w centred) dist-3.255 | stern-kind")
def hull_dist(px, py, verts):
    # signed distance from point to polygon boundary (negative inside), vertices CCW
    n = len(verts); dmin = np.inf; inside = True
    for k in range(n):
        a = verts[k]; b = verts[(k + 1) % n]; e = b - a
        t = np.clip(((px - a[0]) * e[0] + (py - a[1]) * e[1]) / (e @ e), 0, 1)
        q = a + t * e
        dmin = min(dmin, np.hypot(px - q[0], py - q[1]))
        if e[0] * (py - a[1]) - e[1] * (px - a[0]) < 0: inside = False
    return -dmin if inside else dmin
for q in np.where(c["dist"] < 5.0)[0]:
    row = int(c["row"][q])
    static = c["heading_ts"][q] == 0 and c["rel_speed"][q] == c["own_speed"][q]
    size = None
    if row in row_of_kept:
        r = row_of_kept[row]
        o = int(g["first"][r]) + (q - first_rec[row])
        size = int(g["size"][o])
        assert abs(g["dist"][o] - c["dist"][q]) < 1e-12
    if size is None:
        print(row, "static" if static else "moving", "size unknown", "%.3f" % c["dist"][q]); continue
    D = c["dist"][q] + size + CLR
    a = c["bearing"][q] + c["own_heading"][q]           # world direction vessel -> obstacle
    rel = -D * np.array([np.cos(a), np.sin(a)])           # vessel relative to the obstacle, world frame
    h = c["heading_ts"][q]
    ahead = rel[0] * np.cos(h) + rel[1] * np.sin(h); abeam = -rel[0] * np.sin(h) + rel[1] * np.cos(h)
    if static:
        print(f"{row:5d} static r={size:2d} {c['dist'][q]:7.3f} {D:7.2f} | circle: hull distance - clearance = {c['dist'][q]:.3f}")
    else:
        w = size
        d1 = hull_dist(ahead, abeam, S.ship_outline(float(w), "centred")) - CLR
        d2 = hull_dist(ahead, abeam, S.ship_outline(float(w), "stern")) - CLR
        print(f"{row:5d} moving w={size:2d} {c['dist'][q]:7.3f} {D:7.2f} | {ahead:7.2f} {abeam:7.2f} | {ahead/w:5.2f} {abeam/w:5.2f} | {d1:7.3f} | {d2:7.3f}")
