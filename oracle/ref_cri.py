"""The reference's collision-risk index for static obstacles (TEST INFRASTRUCTURE ONLY).

The formula is not in the reference checkout; what is there are its inputs and
outputs: data/metadata_MovingObstaclesNoRules.csv and
data/risk_MovingObstaclesNoRules.csv, row-aligned, 23,680 per-obstacle records
(layout: utils/explore_dataset.py:8,40-53; label reduction:
utils/dataloader.py:104-120).  The restatement below was recovered from those
records and reproduces every one of the 14,102 static-obstacle records
(heading_ts == 0 and speed == speed_OS, utils/explore_dataset.py:46) to
2e-15 (`tests/golden/cri_static_golden.npz`, `tests/test_ref_cri_cpu.py`).
The 9,578 moving-obstacle records are NOT reproduced (their bearing and
speed-ratio terms differ and were not recovered): for moving obstacles this
index is an extrapolation and stays parity unpinned.

    risk = 0.2 u_B(bearing) + 0.3 u_DCPA u_TCPA + 0.3 u_D(dist)

* u_B: 1 for bearing in [-45, 10] deg; ((90 - b)/80)^2 on (10, 90);
  ((b + 270)/225)^2 below -45; 0 from 90 to 180.
* The closest-point-of-approach pair uses the reference's own frames: the
  obstacle position p = dist * (cos, sin)(bearing + heading_OS), i.e. the body
  bearing turned into the world frame, against the unit direction (cos, sin) of
  the logged relative course (for a static obstacle that course is 0 when
  heading_OS > 0 and -pi otherwise).  With L = |p . v| and DCPA = |p x v|:
  u_DCPA = 1 up to D1 = 32 m, linear to 0 at D2 = 160 m;
  closing (p . v < 0): t1 = sqrt(D1^2 - DCPA^2) (DCPA <= D1) else D1 - DCPA,
  t2 = sqrt(D2^2 - DCPA^2), u_TCPA = 1 for L <= t1, (t2 - L)/(t2 - t1) up to t2;
  opening: u_TCPA = 1 - L/D1.  The pair is 0 when the relative speed is 0.
* u_D: 1 up to 20.08 m, ((45.18 - d)/25.1)^2 up to 45.18 m, 0 beyond.

Same arithmetic contract as `auv_oracle.py` (binary64, unfused, this operation
order, `sincos` / `atan2` of the contract), so the HIP and C implementations
match it bit for bit.
"""
from __future__ import annotations

import numpy as np

from .auv_oracle import sincos, atan2, wrap_angle

RAD2DEG = 57.29577951308232
CRI_D1 = 32.0
CRI_D2 = 160.0
CRI_DA = 20.08
CRI_DB = 45.18
CRI_W_B = 0.2
CRI_W_CT = 0.3
CRI_W_D = 0.3


def bearing_membership(b):
    bd = np.asarray(b, dtype=np.float64) * RAD2DEG
    q1 = (90.0 - bd) / 80.0
    q2 = (bd + 270.0) / 225.0
    return np.where(bd < -45.0, q2 * q2,
                    np.where(bd <= 10.0, 1.0,
                             np.where(bd < 90.0, q1 * q1, 0.0)))


def cri_core(px, py, vx, vy, moving, d, b):
    """px, py: obstacle position; vx, vy: unit relative-course direction;
    moving: relative speed > 0; d: surface distance; b: body bearing (rad)."""
    along = px * vx + py * vy
    cross = px * vy - py * vx
    dc = np.abs(cross)
    ln = np.abs(along)
    near = dc <= CRI_D1
    u_dc = np.where(near, 1.0, np.maximum((CRI_D2 - dc) / (CRI_D2 - CRI_D1), 0.0))
    t2 = np.sqrt(np.maximum(CRI_D2 * CRI_D2 - dc * dc, 0.0))
    t1 = np.where(near, np.sqrt(np.maximum(CRI_D1 * CRI_D1 - dc * dc, 0.0)), CRI_D1 - dc)
    den = t2 - t1
    q = (t2 - ln) / np.where(den > 0.0, den, 1.0)
    u_ta = np.where(ln <= t1, 1.0, np.minimum(np.maximum(q, 0.0), 1.0))
    u_tr = np.maximum(1.0 - ln / CRI_D1, 0.0)
    u_t = np.where(along < 0.0, u_ta, u_tr)
    ct = np.where(moving, (CRI_W_CT * u_dc) * u_t, 0.0)
    qd = np.maximum((CRI_DB - d) / (CRI_DB - CRI_DA), 0.0)
    u_d = np.where(d <= CRI_DA, 1.0, qd * qd)
    return (CRI_W_B * bearing_membership(b) + ct) + CRI_W_D * u_d


def cri_rows(own_heading, dist, rel_speed, bearing, course):
    """Risk of dataset records: one value per (heading_OS, dist, speed, bearing, course)."""
    own_heading = np.asarray(own_heading, dtype=np.float64)
    dist = np.asarray(dist, dtype=np.float64)
    sw, cw = sincos(np.asarray(bearing, dtype=np.float64) + own_heading)
    sc, cc = sincos(np.asarray(course, dtype=np.float64))
    return cri_core(dist * cw, dist * sw, cc, sc, np.asarray(rel_speed) > 0.0, dist, bearing)


def risk_reference(state, circles, poly_pose, dist, sensor_range):
    """Per-obstacle risk [V, C+P] of a simulator state with the index above.

    Obstacles that do not move take the reference's static-obstacle course
    (0 / -pi by the sign of the vessel's heading); moving ones take the
    direction of their velocity relative to the vessel (extrapolation, see the
    module docstring).  p is the hull distance along the direction of the
    obstacle's centre; 0 outside sensor range."""
    V = state.shape[1]
    C = circles.shape[1]
    P = poly_pose.shape[1]
    px, py, psi, u, v = state[0], state[1], state[2], state[3], state[4]
    s, c = sincos(psi)
    vox = c * u - s * v
    voy = s * u + c * v
    risk = np.zeros((V, C + P), dtype=np.float64)
    for o in range(C + P):
        if o < C:
            cx, cy = circles[:, o, 0], circles[:, o, 1]
            vx, vy = circles[:, o, 3], circles[:, o, 4]
        else:
            k = o - C
            cx, cy = poly_pose[:, k, 0], poly_pose[:, k, 1]
            vx, vy = poly_pose[:, k, 4], poly_pose[:, k, 5]
        mx = cx - px
        my = cy - py
        rvx = vx - vox
        rvy = vy - voy
        d = dist[:, o]
        n = np.sqrt(mx * mx + my * my)
        has_n = n > 0.0
        ns = np.where(has_n, n, 1.0)
        ux = np.where(has_n, mx / ns, 1.0)
        uy = np.where(has_n, my / ns, 0.0)
        vv = rvx * rvx + rvy * rvy
        still = (vx == 0.0) & (vy == 0.0)
        sp = np.sqrt(vv)
        sps = np.where(vv > 0.0, sp, 1.0)
        dvx = np.where(still, np.where(psi > 0.0, 1.0, -1.0), rvx / sps)
        dvy = np.where(still, 0.0, rvy / sps)
        b = wrap_angle(atan2(my, mx) - psi)
        with np.errstate(invalid="ignore"):   # absent slots have d = +inf; they are gated to 0
            r = cri_core(d * ux, d * uy, dvx, dvy, vv > 0.0, d, b)
        risk[:, o] = np.where(d < sensor_range, r, 0.0)
    return risk
