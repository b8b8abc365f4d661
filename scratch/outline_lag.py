"""Is the ships' apparent displacement ahead a matter of size (outline) or of time (a lag between the logged
record and the logged scan)?  Offsets along the heading in units of the ship's size, of its speed (seconds), and in
metres, with the hull kept at 2w x w; per network the error over the offset, and the same per row-in-episode mod 5."""
import os, sys, time, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiprocessing import Pool
from oracle import auv_oracle as O
from oracle import cnn_oracle as CN
from tests import scan_scenes as S
NETS = ("shallow", "deep2", "deep")
_G = {}

def setup():
    g = S.load("all")
    c = np.load(os.path.join(ROOT, "tests/golden/cri_golden.npz"))
    tr, va, te = S.split_masks(g)
    has_ship = np.array([(~g["static"][g["first"][r]:g["first"][r + 1]]).any() for r in range(len(g["row"]))])
    mask = tr & has_ship
    g["keep"] = mask
    cfg, scene, label = S.build(g)
    # ship speed from the record: o = s' + rel_speed (cos course, sin course), |o| = o . h
    first_rec = np.searchsorted(c["row"], np.arange(int(g["n_rows_file"]) + 1))
    sizes = np.zeros(scene["poly_nv"].shape); speeds = np.zeros(scene["poly_nv"].shape)
    for e, r in enumerate(np.where(mask)[0]):
        row = int(g["row"][r]); k = 0
        lo, hi = int(g["first"][r]), int(g["first"][r + 1])
        base = first_rec[row]
        assert first_rec[row + 1] - base == hi - lo
        for j, o in enumerate(range(lo, hi)):
            if not g["static"][o]:
                q = base + j
                assert abs(c["heading_ts"][q] - g["heading_ts"][o]) < 1e-12
                psi, hts = c["own_heading"][q], c["heading_ts"][q]
                s = c["own_speed"][q] * np.array([np.cos(psi), np.sin(psi)])
                h = np.array([np.cos(hts), np.sin(hts)])
                d = hts - psi
                if d > np.pi or d <= -np.pi:
                    s = 2 * (s @ h) * h - s
                ov = s + c["rel_speed"][q] * np.array([np.cos(c["course"][q]), np.sin(c["course"][q])])
                sizes[e, k] = g["size"][o]; speeds[e, k] = ov @ h; k += 1
    gold = np.load(os.path.join(os.path.dirname(S.GOLDEN), "cnn_golden.npz"))
    rows = g["row"][mask]
    _G.update(cfg=cfg, scene=scene, label=label, sizes=sizes, speeds=speeds, rows=rows, W={k: CN.golden_weights(gold, k) for k in NETS})

def one(p):
    unit, off, lam = p
    sc, cfg, label = _G["scene"], _G["cfg"], _G["label"]
    w = _G["sizes"]
    shift = off * (w if unit == "w" else _G["speeds"] if unit == "s" else np.ones_like(w))   # metres along the heading
    pose = sc["poly_pose"].copy()
    pose[..., 0] += shift * pose[..., 2]
    pose[..., 1] += shift * pose[..., 3]
    ww = w[..., None]; z = ww * 0
    x = np.concatenate([-lam / 2 * ww, z, lam / 2 * ww, z, -lam / 2 * ww], axis=-1)
    y = np.concatenate([-ww / 2, -ww / 2, z, ww / 2, ww / 2], axis=-1)
    verts = np.stack([x, y], axis=-1) * (sc["poly_nv"] > 0)[..., None, None]
    rng, _ = O.lidar(sc["init_state"], sc["circles"], pose, verts, sc["poly_nv"], S.ray_table(), cfg["sensor_range"])
    xx = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(len(label), 1, 180).astype(np.float32)
    return p, np.stack([CN.MODELS[k](xx, _G["W"][k], "risk")[:, 0].astype(np.float64) for k in NETS], axis=1)

if __name__ == "__main__":
    setup()
    sp = _G["speeds"][_G["sizes"] > 0]
    print("ship speeds: min %.3f median %.3f max %.3f; sizes median %.1f" % (sp.min(), np.median(sp), sp.max(), np.median(_G["sizes"][_G["sizes"] > 0])))
    grid = [("w", o, 2.0) for o in (0, 0.5, 1, 1.5, 2, 3, 4)] + [("s", o, 2.0) for o in (-20, -10, -5, 5, 10, 15, 20, 25, 30, 40, 60)] + \
           [("m", o, 2.0) for o in (5, 10, 15, 20, 30)]
    with Pool(6) as pool:
        res = dict(pool.map(one, grid, chunksize=2))
    label = _G["label"]
    for unit in ("w", "s", "m"):
        print("--- offsets in", {"w": "ship sizes", "s": "seconds of the ship's own motion", "m": "metres"}[unit])
        for p in grid:
            if p[0] == unit or (p[1] == 0):
                e = ((res[p] - label[:, None]) ** 2).mean(axis=0)
                print(f"  {p[1]:6.1f}: " + " ".join(f"{v:.4f}" for v in e))
    # per phase of the 25-step list refresh: row-in-episode mod 5 needs episode starts; use file row of own speed 0 -> crude: skip
