"""Builds tests/golden/surge_golden.npz from the reference's dataset file.

    python tests/golden/make_surge_golden.py [/root/reference]

data/metadata_MovingObstaclesNoRules.csv logs [speed_OS, heading_OS, ...]
(utils/explore_dataset.py:8) once per 5 simulator steps.  Wherever two
consecutive rows carry bit-identical headings the vessel ran straight between
them (no sway, no yaw), so the pair (speed before, speed after) is one sample of
the reference simulator's surge dynamics over 5 steps.  The fixture keeps all
such pairs with a speed above 1e-3 m/s: data only.
"""
import os
import sys

import numpy as np


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    rows = [line.split() for line in open(os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv"))]
    speed = np.array([float(r[0]) for r in rows])
    heading = np.array([float(r[1]) for r in rows])
    keep = (heading[1:] == heading[:-1]) & (speed[:-1] > 1e-3)
    idx = np.nonzero(keep)[0]
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "surge_golden.npz")
    np.savez_compressed(dst, row=idx.astype(np.int32), heading=heading[idx], speed_before=speed[idx],
                        speed_after=speed[idx + 1])
    print("wrote", dst, len(idx), "straight-line samples")


if __name__ == "__main__":
    main()
