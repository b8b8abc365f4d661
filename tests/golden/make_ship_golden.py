"""Builds tests/golden/ship_golden.npz from the reference's dataset files.

    python tests/golden/make_ship_golden.py [/root/reference]

Needs turn_golden.npz (make_turn_golden.py): through each of its windows of 4
logged rows the vessel's state and command are known, so its track is.  A
moving obstacle logged in all 4 rows with one heading_ts
(utils/explore_dataset.py:45-53) is then seen at (dist, bearing) from 4 known
places.  If it sails straight along heading_ts at a constant speed s, and
`dist` is its centre distance minus a constant K, then for k = 1..3

    p_k + (dist_k + K) e(bearing_k + psi_k) = p_0 + (dist_0 + K) e(bearing_0 + psi_0) + 5 k s e(heading_ts)

which is 6 equations for the 2 unknowns (K, s).  The script solves them by least
squares for every such obstacle and keeps those met to 1e-8 m (every candidate
is): this pins the obstacles' constant-velocity advance and the centre-distance
form of a ship's `dist`.  K - 3.255 (the ship's size) is an integer for every
one of them.  The fixture holds, per obstacle: the vessel state at the first
row, the command, (heading_ts, s, K), and the 4 logged records with their risks
(data/risk_MovingObstaclesNoRules.csv) -- data only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import auv_oracle as orc  # noqa: E402

ROWS, LOG_EVERY = 4, 5


def main():
    import importlib
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    meta = [np.array(line.split(), dtype=np.float64)
            for line in open(os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv"))]
    risk = [np.array(line.split(), dtype=np.float64)
            for line in open(os.path.join(ref, "data", "risk_MovingObstaclesNoRules.csv")) if line.strip()]
    cfg = importlib.import_module("gym-auv_amd.params").make_config(1, 0, 0)
    t = np.load(os.path.join(HERE, "turn_golden.npz"))
    n = t["start"].shape[1]
    st = np.zeros((6, n))
    st[2], st[3:] = t["heading"][0], t["start"]
    track = [st.copy()]
    act = np.stack([t["surge"], t["steer"]])
    for _ in range(ROWS - 1):
        for _ in range(LOG_EVERY):
            st = orc.step_dynamics(st, act, cfg)
        track.append(st.copy())

    out = {k: [] for k in ("row", "state", "surge", "steer", "heading_ts", "speed_ts", "offset", "records", "own", "risk")}
    worst = 0.0
    for j in range(n):
        i = int(t["row"][j])
        recs = []
        for k in range(ROWS):
            a = meta[i + k]
            recs.append([(a[2 + 5 * o:7 + 5 * o], risk[i + k][o]) for o in range((len(a) - 2) // 5)
                         if not (a[2 + 5 * o] == 0 and a[4 + 5 * o] == a[0])])
        for r0 in recs[0]:
            chain = [r0]
            for k in range(1, ROWS):
                m = [r for r in recs[k] if r[0][0] == r0[0][0]]
                if len(m) != 1:
                    break
                chain.append(m[0])
            if len(chain) < ROWS:
                continue
            h = r0[0][0]
            eh = np.array([np.cos(h), np.sin(h)])
            e = [np.array([np.cos(c[0][3] + track[k][2, j]), np.sin(c[0][3] + track[k][2, j])]) for k, c in enumerate(chain)]
            A, b = [], []
            for k in range(1, ROWS):
                dp = track[k][:2, j] - track[0][:2, j]
                for c in range(2):
                    A.append([e[k][c] - e[0][c], -LOG_EVERY * k * eh[c]])
                    b.append(-(dp[c] + chain[k][0][1] * e[k][c] - chain[0][0][1] * e[0][c]))
            A, b = np.array(A), np.array(b)
            x = np.linalg.lstsq(A, b, rcond=None)[0]
            res = np.abs(A @ x - b).max()
            if res > 1e-8:
                continue
            worst = max(worst, res)
            out["row"].append(i)
            out["state"].append(track[0][:, j])
            out["surge"].append(t["surge"][j])
            out["steer"].append(t["steer"][j])
            out["heading_ts"].append(h)
            out["speed_ts"].append(x[1])
            out["offset"].append(x[0])
            out["records"].append([c[0] for c in chain])
            out["own"].append([meta[i + k][:2] for k in range(ROWS)])
            out["risk"].append([c[1] for c in chain])
    K = np.array(out["offset"])
    size = K - 3.255
    print(len(K), "ships; worst residual", worst, "m; size off an integer by at most", np.abs(size - np.rint(size)).max(),
          "; sizes", np.rint(size).min(), "..", np.rint(size).max())
    dst = os.path.join(HERE, "ship_golden.npz")
    np.savez_compressed(dst, row=np.array(out["row"], dtype=np.int32), state=np.array(out["state"]).T,
                        **{k: np.array(out[k]) for k in ("surge", "steer", "heading_ts", "speed_ts", "offset", "records", "own", "risk")})
    print("wrote", dst)


if __name__ == "__main__":
    main()
