"""Builds tests/golden/track_golden.npz from the reference's dataset file.

    python tests/golden/make_track_golden.py [/root/reference]

Needs surge_golden.npz and turn_golden.npz (make_surge_golden.py,
make_turn_golden.py): between the rows they cover, the vessel's full state and
its command are known, so its displacement over the 5 steps between two logged
rows can be integrated.  A static obstacle logged in both rows
(heading_ts == 0 and speed == speed_OS, utils/explore_dataset.py:46) is seen at
(dist, bearing) before and after; if `dist` is the centre distance minus a
constant and `bearing` the body-frame direction of the centre, then

    (dist_A + K) e(bearing_A + psi_A) - (dist_B + K) e(bearing_B + psi_B) = p_B - p_A

which is two equations for the single unknown K (obstacle radius plus whatever
clearance the reference subtracts).  The script solves K by least squares for
every pair of static records of two consecutive rows and keeps the pair when
both equations are met to 1e-8 m: 5,230 obstacle sightings that pin, all at
once, the integration of the position, the body-frame bearing and the
centre-distance form of `dist`.  K - 3.255 is an integer between 18 and 45 for
every one of them.  Data only: rows, the vessel state before, the command, the
two sightings, K, and the second row's full record with its logged risk
(data/risk_MovingObstaclesNoRules.csv).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import auv_oracle as orc  # noqa: E402

LOG_EVERY = 5
SURGE_RATIO = 0.678685866220174      # five RKF45 steps of u' = -(2/25.8) u, see make_surge_golden.py


def advance(st, surge, steer, cfg, steps=LOG_EVERY):
    act = np.stack([surge, steer])
    for _ in range(steps):
        st = orc.step_dynamics(st, act, cfg)
    return st


def main():
    import importlib
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    rows = [np.array(line.split(), dtype=np.float64)
            for line in open(os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv"))]
    risks = [np.array(line.split(), dtype=np.float64)
             for line in open(os.path.join(ref, "data", "risk_MovingObstaclesNoRules.csv")) if line.strip()]
    cfg = importlib.import_module("gym-auv_amd.params").make_config(1, 0, 0)

    # segments (row A -> row A+1) with a known state at A and a known constant command
    seg_row, seg_state, seg_surge, seg_steer = [], [], [], []
    g = np.load(os.path.join(HERE, "surge_golden.npz"))
    thrust = 2.0 * (g["speed_after"] - SURGE_RATIO * g["speed_before"]) / (1.0 - SURGE_RATIO)
    for i, psi, u, f in zip(g["row"], g["heading"], g["speed_before"], thrust):
        seg_row.append(int(i))
        seg_state.append([0.0, 0.0, psi, u, 0.0, 0.0])
        seg_surge.append(min(max(f / 2.0, 0.0), 1.0))
        seg_steer.append(0.0)
    t = np.load(os.path.join(HERE, "turn_golden.npz"))
    n = t["start"].shape[1]
    st = np.zeros((6, n))
    st[2], st[3:] = t["heading"][0], t["start"]
    for k in range(3):
        for j in range(n):
            seg_row.append(int(t["row"][j]) + k)
            seg_state.append([0.0, 0.0, t["heading"][k][j], st[3][j], st[4][j], st[5][j]])
            seg_surge.append(float(t["surge"][j]))
            seg_steer.append(float(t["steer"][j]))
        st = advance(st, t["surge"], t["steer"], cfg)
    # windows overlap: keep one segment per row
    _, first = np.unique(np.array(seg_row), return_index=True)
    seg_row = np.array(seg_row)[first]
    A = np.array(seg_state)[first].T.copy()
    surge = np.array(seg_surge)[first]
    steer = np.array(seg_steer)[first]
    B = advance(A.copy(), surge, steer, cfg)
    dp = B[:2] - A[:2]

    out = {k: [] for k in ("row", "state", "surge", "steer", "dist_a", "bearing_a", "dist_b", "bearing_b", "offset", "record_b", "risk_b")}
    for s, i in enumerate(seg_row):
        a, b = rows[i], rows[i + 1]
        stat_a = [np.append(a[2 + 5 * o:7 + 5 * o], risks[i][o]) for o in range((len(a) - 2) // 5)
                  if a[2 + 5 * o] == 0 and a[4 + 5 * o] == a[0]]
        stat_b = [np.append(b[2 + 5 * o:7 + 5 * o], risks[i + 1][o]) for o in range((len(b) - 2) // 5)
                  if b[2 + 5 * o] == 0 and b[4 + 5 * o] == b[0]]
        for ra in stat_a:
            for rb in stat_b:
                ea = np.array([np.cos(ra[3] + A[2, s]), np.sin(ra[3] + A[2, s])])
                eb = np.array([np.cos(rb[3] + B[2, s]), np.sin(rb[3] + B[2, s])])
                m = ea - eb
                rhs = dp[:, s] + rb[1] * eb - ra[1] * ea
                if m @ m < 1e-12:
                    continue
                K = (m @ rhs) / (m @ m)
                if np.abs(K * m - rhs).max() < 1e-8:
                    out["row"].append(i)
                    out["state"].append(A[:, s])
                    out["surge"].append(surge[s])
                    out["steer"].append(steer[s])
                    out["dist_a"].append(ra[1])
                    out["bearing_a"].append(ra[3])
                    out["dist_b"].append(rb[1])
                    out["bearing_b"].append(rb[3])
                    out["offset"].append(K)
                    out["record_b"].append(rb[:5])
                    out["risk_b"].append(rb[5])
    K = np.array(out["offset"])
    frac = np.abs((K - 3.255) - np.rint(K - 3.255))
    print(len(K), "sightings over", len(set(out["row"])), "row pairs; K - 3.255 off an integer by at most", frac.max(),
          "; radii", np.rint(K - 3.255).min(), "..", np.rint(K - 3.255).max())
    dst = os.path.join(HERE, "track_golden.npz")
    np.savez_compressed(dst, row=np.array(out["row"], dtype=np.int32), state=np.array(out["state"]).T,
                        **{k: np.array(out[k]) for k in ("surge", "steer", "dist_a", "bearing_a", "dist_b", "bearing_b", "offset", "record_b", "risk_b")})
    print("wrote", dst)


if __name__ == "__main__":
    main()
