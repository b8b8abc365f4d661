"""Three-parameter fit of the ships' outline (beam, length, offset ahead, all in units of the logged size) against
the reference's shipped networks on the rebuilt training rows that list a ship; grid points in parallel."""
import os, sys, time, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiprocessing import Pool
from oracle import auv_oracle as O
from oracle import cnn_oracle as CN
from tests import scan_scenes as S
NETS = ("shallow", "deep2", "deep")

def hull(w, beam, lam, delta):
    w = np.asarray(w, dtype=np.float64)[..., None]
    z = w * 0.0
    x = np.concatenate([(delta - lam / 2) * w, delta * w, (delta + lam / 2) * w, delta * w, (delta - lam / 2) * w], axis=-1)
    y = np.concatenate([-beam * w / 2, -beam * w / 2, z, beam * w / 2, beam * w / 2], axis=-1)
    return np.stack([x, y], axis=-1)

_G = {}
def setup(split):
    g = S.load("all")
    tr, va, te = S.split_masks(g)
    has_ship = np.array([(~g["static"][g["first"][r]:g["first"][r + 1]]).any() for r in range(len(g["row"]))])
    mask = {"train": tr, "test": te, "all": tr | va | te}[split] & has_ship
    g["keep"] = mask
    cfg, scene, label = S.build(g)
    sizes = np.zeros(scene["poly_nv"].shape)
    for e, r in enumerate(np.where(mask)[0]):
        k = 0
        for o in range(int(g["first"][r]), int(g["first"][r + 1])):
            if not g["static"][o]:
                sizes[e, k] = g["size"][o]; k += 1
    gold = np.load(os.path.join(os.path.dirname(S.GOLDEN), "cnn_golden.npz"))
    _G.update(cfg=cfg, scene=scene, label=label, sizes=sizes, W={k: CN.golden_weights(gold, k) for k in NETS})

def one(p):
    beam, lam, de = p
    sc, cfg, label = _G["scene"], _G["cfg"], _G["label"]
    verts = hull(_G["sizes"], beam, lam, de) * (sc["poly_nv"] > 0)[..., None, None]
    rng, _ = O.lidar(sc["init_state"], sc["circles"], sc["poly_pose"], verts, sc["poly_nv"], S.ray_table(), cfg["sensor_range"])
    x = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(len(label), 1, 180).astype(np.float32)
    pred = np.stack([CN.MODELS[k](x, _G["W"][k], "risk")[:, 0].astype(np.float64) for k in NETS], axis=1)
    return p, pred

if __name__ == "__main__":
    split = sys.argv[1] if len(sys.argv) > 1 else "train"
    setup(split)
    beams = [1.0, 1.5, 2.0, 3.0]
    lams = [2.0, 3.0, 4.0, 5.0, 6.0, 8.0]
    deltas = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0]
    grid = list(itertools.product(beams, lams, deltas))
    t0 = time.time()
    with Pool(6) as pool:
        res = dict(pool.map(one, grid, chunksize=4))
    label = _G["label"]; n = len(label)
    print(f"{split}: {n} rows, {len(grid)} grid points, {time.time()-t0:.0f}s")
    np.savez_compressed(f"/tmp/outline_fit3_{split}.npz", grid=np.array(grid), pred=np.stack([res[p] for p in grid]), label=label)
    for i, k in enumerate(NETS):
        mse = {p: ((res[p][:, i] - label) ** 2).mean() for p in grid}
        bias = {p: (res[p][:, i] - label).mean() for p in grid}
        best = min(mse, key=mse.get)
        print(f"--- {k}: minimum at beam,lam,delta = {best}: mse {mse[best]:.4f} bias {bias[best]:+.4f}; default (1,2,0): {mse[(1.0,2.0,0.0)]:.4f} bias {bias[(1.0,2.0,0.0)]:+.4f}; stern (1,2,.5): {mse[(1.0,2.0,0.5)]:.4f}")
        for b in beams:
            print(f" beam {b}: rows lam {lams}, cols delta {deltas}")
            for lam in lams:
                print("   " + " ".join(f"{mse[(b, lam, de)]:.4f}" for de in deltas))
