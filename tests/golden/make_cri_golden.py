"""Builds tests/golden/cri_golden.npz from the reference's dataset files.

    python tests/golden/make_cri_golden.py [/root/reference]

Reads data/metadata_MovingObstaclesNoRules.csv and
data/risk_MovingObstaclesNoRules.csv (row-aligned text, `%.18e`): a metadata
row is [speed_OS, heading_OS, (heading_ts, dist, speed, bearing, course) x n]
(utils/explore_dataset.py:8,40-53), the risk row holds the n per-obstacle
risks.  The fixture keeps every per-obstacle record (23,680; 14,102 of them
static: heading_ts == 0 and speed == speed_OS, utils/explore_dataset.py:46) as
eight float64 columns: the inputs and the risk the reference's simulator logged
for them.  It is data only; nothing of the reference's code is stored.
"""
import os
import sys

import numpy as np


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    meta_path = os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv")
    risk_path = os.path.join(ref, "data", "risk_MovingObstaclesNoRules.csv")
    meta = [line.split() for line in open(meta_path)]
    risk = [line.split() for line in open(risk_path) if line.strip()]
    assert len(meta) == len(risk)
    cols = {k: [] for k in ("row", "own_speed", "own_heading", "heading_ts", "dist", "rel_speed", "bearing",
                            "course", "risk")}
    n_static = 0
    for i, (m, r) in enumerate(zip(meta, risk)):
        m = np.array(m, dtype=np.float64)
        r = np.array(r, dtype=np.float64)
        n = (len(m) - 2) // 5
        assert n == len(r)
        for o in range(n):
            heading_ts, dist, speed, bearing, course = m[2 + 5 * o:7 + 5 * o]
            n_static += int(heading_ts == 0 and speed == m[0])
            cols["row"].append(i)
            cols["own_speed"].append(m[0])
            cols["own_heading"].append(m[1])
            cols["heading_ts"].append(heading_ts)
            cols["dist"].append(dist)
            cols["rel_speed"].append(speed)
            cols["bearing"].append(bearing)
            cols["course"].append(course)
            cols["risk"].append(r[o])
    out = {k: np.array(v, dtype=np.int32 if k == "row" else np.float64) for k, v in cols.items()}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cri_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, "records", len(out["risk"]), "of them static", n_static)


if __name__ == "__main__":
    main()
