"""Does the ships' apparent length go with their size or with their speed?  Stern at -s0 sizes, bow at b0 sizes +
T seconds of own motion ahead of the logged pose, beam = size."""
import os, sys, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scratch"))
from multiprocessing import Pool
import outline_lag as L
from oracle import auv_oracle as O
from oracle import cnn_oracle as CN
from tests import scan_scenes as S
NETS = L.NETS

def one(p):
    s0, b0, T = p
    G = L._G
    sc, cfg, label = G["scene"], G["cfg"], G["label"]
    w = G["sizes"][..., None]; v = G["speeds"][..., None]; z = w * 0
    stern, bow = -s0 * w, b0 * w + T * v
    mid = 0.5 * (stern + bow)
    x = np.concatenate([stern, mid, bow, mid, stern], axis=-1)
    y = np.concatenate([-w / 2, -w / 2, z, w / 2, w / 2], axis=-1)
    verts = np.stack([x, y], axis=-1) * (sc["poly_nv"] > 0)[..., None, None]
    rng, _ = O.lidar(sc["init_state"], sc["circles"], sc["poly_pose"], verts, sc["poly_nv"], S.ray_table(), cfg["sensor_range"])
    xx = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(len(label), 1, 180).astype(np.float32)
    return p, np.stack([CN.MODELS[k](xx, G["W"][k], "risk")[:, 0].astype(np.float64) for k in NETS], axis=1)

if __name__ == "__main__":
    L.setup()
    grid = list(itertools.product([0.5, 1.0, 1.5], [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], [0.0, 10.0, 20.0, 40.0, 80.0]))
    with Pool(6) as pool:
        res = dict(pool.map(one, grid, chunksize=3))
    label = L._G["label"]
    for s0 in (0.5, 1.0, 1.5):
        print(f"--- stern at -{s0} sizes; rows bow0 (sizes) [1..6], columns T (s) [0,10,20,40,80]; shallow | deep2 | deep")
        for b0 in (1.0, 2.0, 3.0, 4.0, 5.0, 6.0):
            e = [((res[(s0, b0, T)] - label[:, None]) ** 2).mean(axis=0) for T in (0.0, 10.0, 20.0, 40.0, 80.0)]
            print("  " + " | ".join(" ".join(f"{x[i]:.4f}" for x in e) for i in range(3)))
