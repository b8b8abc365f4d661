"""Two-parameter fit of the ships' outline against the reference's shipped networks.

Hull family: beam w, length lam*w, mid-hull delta*w ahead of the logged pose, bow taper over the front half.
(lam, delta) = (2, 0) is scenarios.ship_polygon, (2, 0.5) has the pose at the centre of a square stern section.
For every (lam, delta) of a grid: cast the rebuilt training rows that list a ship, feed the scans to the three
networks, take the squared error against the labels; report the grid, the minimum and the set of grid points not
significantly worse than it (paired z < 2)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import auv_oracle as O
from oracle import cnn_oracle as CN
from tests import scan_scenes as S

NETS = ("shallow", "deep2", "deep")

def hull(w, lam, delta):
    w = np.asarray(w, dtype=np.float64)[..., None]
    z = w * 0.0
    x = np.concatenate([(delta - lam / 2) * w, delta * w, (delta + lam / 2) * w, delta * w, (delta - lam / 2) * w], axis=-1)
    y = np.concatenate([-w / 2, -w / 2, z, w / 2, w / 2], axis=-1)
    return np.stack([x, y], axis=-1)

def main(split="train"):
    g = S.load("all")
    tr, va, te = S.split_masks(g)
    has_ship = np.array([(~g["static"][g["first"][r]:g["first"][r + 1]]).any() for r in range(len(g["row"]))])
    mask = {"train": tr, "test": te, "all": tr | va | te}[split] & has_ship
    g["keep"] = mask
    cfg, scene, label = S.build(g)
    sizes = np.zeros(scene["poly_nv"].shape)
    idx = np.where(mask)[0]
    for e, r in enumerate(idx):
        lo, hi = int(g["first"][r]), int(g["first"][r + 1])
        k = 0
        for o in range(lo, hi):
            if not g["static"][o]:
                sizes[e, k] = g["size"][o]; k += 1
    gold = np.load(os.path.join(os.path.dirname(S.GOLDEN), "cnn_golden.npz"))
    W = {k: CN.golden_weights(gold, k) for k in NETS}
    lams = [1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 5.0]
    deltas = [-0.5, -0.25, 0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0]
    err = {}
    t0 = time.time()
    for lam in lams:
        for de in deltas:
            verts = hull(sizes, lam, de) * (scene["poly_nv"] > 0)[..., None, None]
            rng, _ = O.lidar(scene["init_state"], scene["circles"], scene["poly_pose"], verts, scene["poly_nv"], S.ray_table(), cfg["sensor_range"])
            x = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(len(label), 1, 180).astype(np.float32)
            err[(lam, de)] = np.stack([(CN.MODELS[k](x, W[k], "risk")[:, 0].astype(np.float64) - label) ** 2 for k in NETS], axis=1)
        print(f"lam {lam} done, {time.time()-t0:.0f}s", flush=True)
    n = len(label)
    print(f"{split}: {n} rows with a ship")
    for i, k in enumerate(NETS):
        print(f"--- {k}: mean squared error; rows lam, columns delta {deltas}")
        for lam in lams:
            print(f"{lam:4.1f} " + " ".join(f"{err[(lam, de)][:, i].mean():.4f}" for de in deltas))
        best = min(err, key=lambda p: err[p][:, i].mean())
        ok = []
        for p in err:
            d = err[p][:, i] - err[best][:, i]
            z = d.mean() / (d.std() / np.sqrt(n) + 1e-300)
            if z < 2.0: ok.append(p)
        print("minimum at", best, "mse", err[best][:, i].mean(), "; not worse at z<2:", sorted(ok))
    tot = {p: err[p].mean() for p in err}
    best = min(tot, key=tot.get)
    print("all three networks together: minimum at", best)
    return err

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "train")
