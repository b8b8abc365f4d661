"""Builds tests/golden/turn_golden.npz from the reference's dataset file.

    python tests/golden/make_turn_golden.py [/root/reference]

data/metadata_MovingObstaclesNoRules.csv logs [speed_OS, heading_OS, ...]
(utils/explore_dataset.py:8) once per 5 simulator steps.  Wherever the vessel
turned with the thruster held at one of its limits (0 or full) and the rudder
at zero or hard over to either side, over 4 consecutive rows (15 steps), the 7
logged numbers (4 speeds, 3 heading increments) are a function of only 3
unknowns, the body velocities (u, v, r) at the first row -- so every such window
over-determines the sway-yaw dynamics, and with the rudder hard over the yaw
moment limit too, by 4 equations.

The script finds those windows: for every run of 4 rows with changing heading
it solves for (u, v, r) by Gauss-Newton on the oracle's `step_dynamics` with each
of the six commands {0, 1} x {0, +1, -1}, and keeps the window if all 7 numbers
are met to 1e-13.  (A window with any other command history, or a model whose sway-yaw
block differs from the reference's, leaves residuals of 1e-3 .. 1e-9.)  The
fixture holds the rows, the command and the fitted start; tests integrate
from the start and compare with the logged rows.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import auv_oracle as orc  # noqa: E402

ROWS = 4          # rows per window
LOG_EVERY = 5     # simulator steps per logged row


def model_cfg():
    import importlib
    params = importlib.import_module("gym-auv_amd.params")
    return params.make_config(1, 0, 0)


def predict(p, psi0, surge, steer, cfg):
    """p [3,W] = (u, v, r) at the first row -> speeds [ROWS,W], heading
    increments [ROWS-1,W] (unwrapped) over the window under the action
    (surge, steer)."""
    W = p.shape[1]
    st = np.zeros((6, W))
    st[2] = psi0
    st[3:] = p
    act = np.stack([np.full(W, surge), np.full(W, steer)])
    sp = [np.sqrt(st[3] * st[3] + st[4] * st[4])]
    dh = []
    turned = np.zeros(W)
    for _ in range(ROWS - 1):
        for _ in range(LOG_EVERY):
            before = st[2].copy()
            st = orc.step_dynamics(st, act, cfg)
            turned = turned + orc.wrap_angle(st[2] - before)
        sp.append(np.sqrt(st[3] * st[3] + st[4] * st[4]))
        dh.append(turned.copy())
    return np.stack(sp), np.stack(dh)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    rows = [line.split() for line in open(os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv"))]
    speed = np.array([float(r[0]) for r in rows])
    heading = np.array([float(r[1]) for r in rows])
    cfg = model_cfg()

    idx = []
    for i in range(len(speed) - ROWS + 1):
        if np.any(speed[i:i + ROWS] < 1e-3) or np.any(np.diff(heading[i:i + ROWS]) == 0.0):
            continue
        idx.append(i)
    idx = np.array(idx)
    S = np.stack([speed[idx + k] for k in range(ROWS)])
    H = np.stack([heading[idx + k] for k in range(ROWS)])
    DH = orc.wrap_angle(H[1:] - H[:1])          # a 15-step turn stays far below pi
    psi0 = H[0]

    taken = np.zeros(len(idx), dtype=bool)
    start = np.zeros((3, len(idx)))
    cmd = np.zeros((2, len(idx)))
    for surge in (0.0, 1.0):
        for steer in (0.0, 1.0, -1.0):
            def residual(p):
                sp, dh = predict(p, psi0, surge, steer, cfg)
                return np.concatenate([sp - S, dh - DH])            # [7,W]

            best = None
            for v0 in (0.0, 0.02, -0.02):
                p = np.stack([S[0], np.full(len(idx), v0), DH[0] / LOG_EVERY])
                for _ in range(40):
                    r0 = residual(p)
                    J = np.empty((r0.shape[0], 3, len(idx)))
                    for k in range(3):
                        q = p.copy()
                        q[k] += 1e-7
                        J[:, k] = (residual(q) - r0) / 1e-7
                    JT = np.transpose(J, (2, 1, 0))                  # [W,3,7]
                    A = JT @ np.transpose(JT, (0, 2, 1)) + 1e-30 * np.eye(3)
                    g = JT @ r0.T[:, :, None]
                    p = np.clip(p - np.linalg.solve(A, g)[:, :, 0].T, -5.0, 5.0)
                e = np.abs(residual(p)).max(axis=0)
                if best is None:
                    best, best_e = p.copy(), e.copy()
                else:
                    better = e < best_e
                    best[:, better] = p[:, better]
                    best_e[better] = e[better]
            keep = best_e < 1e-13
            print("command", (surge, steer), ":", int(keep.sum()), "of", len(idx), "windows; worst residual kept",
                  best_e[keep].max() if keep.any() else None)
            assert not (keep & taken).any()
            taken |= keep
            start[:, keep] = best[:, keep]
            cmd[0, keep], cmd[1, keep] = surge, steer
    keep = taken
    start = start[:, keep]
    dst = os.path.join(HERE, "turn_golden.npz")
    np.savez_compressed(dst, row=idx[keep].astype(np.int32), surge=cmd[0, keep], steer=cmd[1, keep], start=start,
                        speed=S[:, keep], heading=H[:, keep])
    print("wrote", dst, int(keep.sum()), "windows")


if __name__ == "__main__":
    main()
