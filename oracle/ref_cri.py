"""The reference's collision-risk index (TEST INFRASTRUCTURE ONLY).

The formula is not in the reference checkout; what is there are its inputs and
outputs: data/metadata_MovingObstaclesNoRules.csv and
data/risk_MovingObstaclesNoRules.csv, row-aligned, 23,680 per-obstacle records
(layout: utils/explore_dataset.py:8,40-53; label reduction:
utils/dataloader.py:104-120).  The restatement below was recovered from those
records and reproduces ALL of them -- 14,102 static-obstacle records
(heading_ts == 0 and speed == speed_OS, utils/explore_dataset.py:46) and 9,578
moving-obstacle records -- to 2e-15 (`tests/golden/cri_golden.npz`,
`tests/test_ref_cri_cpu.py`).

    risk = max(0, 0.2 u_B(bearing) + 0.3 u_DCPA u_TCPA + 0.3 u_D(dist) + 0.2 u_A)

* u_B: 1 for bearing in [-45, 10] deg; ((90 - b)/80)^2 on (10, 90);
  ((b + 270)/225)^2 below -45; 0 from 90 to 180.
* The closest-point-of-approach pair uses the reference's own frames: the
  obstacle position p = dist * (cos, sin)(bearing + heading_OS), i.e. the body
  bearing turned into the world frame, against the unit direction (cos, sin) of
  the logged relative course.  With L = |p . v| and DCPA = |p x v|:
  u_DCPA = 1 up to D1 = 32 m, linear to 0 at D2 = 160 m;
  closing (p . v < 0): t1 = sqrt(D1^2 - DCPA^2) (DCPA <= D1) else D1 - DCPA,
  t2 = sqrt(D2^2 - DCPA^2), u_TCPA = 1 for L <= t1, (t2 - L)/(t2 - t1) up to t2;
  opening: u_TCPA = 1 - L/D1.  The pair is 0 when the relative speed is 0.
* u_D: 1 up to 20.08 m, ((45.18 - d)/25.1)^2 up to 45.18 m, 0 beyond.
* u_A (moving obstacles only): cos(a)|cos(a)| of the vessel's bearing a seen
  from the obstacle, relative to the obstacle's heading:
  a = bearing + heading_OS + pi - heading_ts.  +1 dead ahead of the obstacle,
  -1 dead astern.
* The logged relative course (`reference_course`): a static obstacle has 0 when
  heading_OS > 0 and -pi otherwise; a moving one has
  heading_ts + atan2(s * speed_OS * sin D, speed_ts - speed_OS * cos D) with
  D = heading_ts - heading_OS taken as is, s = +1 for |D| <= pi and -1 beyond
  (the vessel's velocity mirrored about the obstacle's heading when the raw
  difference leaves (-pi, pi]).  All 9,578 moving records follow it.

Same arithmetic contract as `auv_oracle.py` (binary64, unfused, this operation
order, `sincos` / `atan2` of the contract), so the HIP and C implementations
match it bit for bit.
"""
from __future__ import annotations

import numpy as np

from .auv_oracle import PI, sincos, atan2, wrap_angle

RAD2DEG = 57.29577951308232
CRI_D1 = 32.0
CRI_D2 = 160.0
CRI_DA = 20.08
CRI_DB = 45.18
CRI_W_B = 0.2
CRI_W_CT = 0.3
CRI_W_D = 0.3
CRI_W_A = 0.2


def bearing_membership(b):
    bd = np.asarray(b, dtype=np.float64) * RAD2DEG
    q1 = np.abs(90.0 - bd) / 80.0
    q2 = np.abs(bd + 270.0) / 225.0
    return np.where(bd < -45.0, q2 * q2,
                    np.where(bd <= 10.0, 1.0,
                             np.where(bd < 90.0, q1 * q1, 0.0)))


def cri_terms(along, cross, moving, d, b, aspect):
    """along = p . v, cross = p x v for the obstacle position p and the unit
    relative-course direction v; moving: relative speed > 0; d: surface
    distance; b: body bearing (rad); aspect: u_A (0 for a static obstacle).
    Divisions by 128 and 32 are written as the exact products they are."""
    dc = np.abs(cross)
    ln = np.abs(along)
    near = dc <= CRI_D1
    u_dc = np.where(near, 1.0, np.maximum((CRI_D2 - dc) * 0.0078125, 0.0))
    t2 = np.sqrt(np.maximum(CRI_D2 * CRI_D2 - dc * dc, 0.0))
    t1 = np.where(near, np.sqrt(np.maximum(CRI_D1 * CRI_D1 - dc * dc, 0.0)), CRI_D1 - dc)
    den = t2 - t1
    q = (t2 - ln) / np.where(den > 0.0, den, 1.0)
    u_ta = np.where(ln <= t1, 1.0, np.minimum(np.maximum(q, 0.0), 1.0))
    u_tr = np.maximum(1.0 - ln * 0.03125, 0.0)
    u_t = np.where(along < 0.0, u_ta, u_tr)
    ct = np.where(moving, (CRI_W_CT * u_dc) * u_t, 0.0)
    span = CRI_DB - d
    qd = np.where(span > 0.0, span / (CRI_DB - CRI_DA), 0.0)
    u_d = np.where(d <= CRI_DA, 1.0, qd * qd)
    total = (ct + CRI_W_D * u_d) + (CRI_W_B * bearing_membership(b) + CRI_W_A * aspect)
    return np.maximum(total, 0.0)


def cri_core(px, py, vx, vy, moving, d, b, aspect):
    """`cri_terms` for a position p and a unit direction v."""
    return cri_terms(px * vx + py * vy, px * vy - py * vx, moving, d, b, aspect)


def cri_rows(own_speed, own_heading, heading_ts, dist, rel_speed, bearing, course):
    """Risk of dataset records: one value per obstacle record (speed_OS,
    heading_OS | heading_ts, dist, speed, bearing, course) of a metadata row."""
    own_speed = np.asarray(own_speed, dtype=np.float64)
    own_heading = np.asarray(own_heading, dtype=np.float64)
    heading_ts = np.asarray(heading_ts, dtype=np.float64)
    dist = np.asarray(dist, dtype=np.float64)
    rel_speed = np.asarray(rel_speed, dtype=np.float64)
    bearing = np.asarray(bearing, dtype=np.float64)
    bw = bearing + own_heading
    sw, cw = sincos(bw)
    sc, cc = sincos(np.asarray(course, dtype=np.float64))
    static = (heading_ts == 0.0) & (rel_speed == own_speed)       # utils/explore_dataset.py:46
    _, ca = sincos(bw - heading_ts)                                # cos(a) = -cos(a - pi)
    aspect = np.where(static, 0.0, 0.0 - ca * np.abs(ca))
    return cri_core(dist * cw, dist * sw, cc, sc, rel_speed > 0.0, dist, bearing, aspect)


def reference_course(own_speed, own_heading, heading_ts, speed_ts):
    """The relative course the reference logs for a moving obstacle (rad, not wrapped)."""
    dl = heading_ts - own_heading
    s = np.where(np.abs(dl) > np.pi, -1.0, 1.0)
    return heading_ts + np.arctan2(s * own_speed * np.sin(dl), speed_ts - own_speed * np.cos(dl))


def reference_motion(ovx, ovy, vox, voy, psi):
    """Relative motion of an obstacle the way the reference forms it; returns
    (still, ss, wx, wy, ww, rr): `still` = obstacle without a velocity, ss the
    vessel's speed, (wx, wy) a vector along the relative course with squared
    length ww, rr the squared relative speed of a moving obstacle.

    The vessel's velocity is its speed along its heading; an obstacle without
    a velocity is static (course along +x for a positive heading, -x
    otherwise); one with a velocity o is headed along it, and its relative
    velocity is o minus the vessel's velocity, the latter mirrored about o when
    heading_ts - heading_OS leaves (-pi, pi].  That condition is evaluated
    without angles: with X = (cos psi, sin psi) x o, it holds iff psi < 0 < o_y
    and X < 0, or o_y < 0 < psi and X > 0."""
    spsi, cpsi = sincos(psi)
    ss = np.sqrt(vox * vox + voy * voy)
    sx = ss * cpsi
    sy = ss * spsi
    still = (ovx == 0.0) & (ovy == 0.0)
    oo = ovx * ovx + ovy * ovy
    X = cpsi * ovy - spsi * ovx
    mirror = ((spsi < 0.0) & (ovy > 0.0) & (X < 0.0)) | ((spsi > 0.0) & (ovy < 0.0) & (X > 0.0))
    dot = sx * ovx + sy * ovy
    f = (2.0 * dot) / np.where(still, 1.0, oo)
    ex = np.where(mirror, f * ovx - sx, sx)
    ey = np.where(mirror, f * ovy - sy, sy)
    rx = ovx - ex
    ry = ovy - ey
    rr = rx * rx + ry * ry
    wx = np.where(still, np.where(psi > 0.0, 1.0, -1.0), rx)
    wy = np.where(still, 0.0, ry)
    ww = np.where(still, 1.0, rr)
    return still, ss, wx, wy, ww, rr


def _own_velocity(state):
    psi, u, v = state[2], state[3], state[4]
    s, c = sincos(psi)
    return c * u - s * v, s * u + c * v


def _obstacle(circles, poly_pose, o):
    C = circles.shape[1]
    if o < C:
        return circles[:, o, 0], circles[:, o, 1], circles[:, o, 3], circles[:, o, 4]
    k = o - C
    return poly_pose[:, k, 0], poly_pose[:, k, 1], poly_pose[:, k, 4], poly_pose[:, k, 5]


def reference_distances(state, circles, poly_pose, poly_verts, poly_nv, dist, clearance):
    """The `dist` column the way the reference logs it: centre distance minus
    the obstacle's size minus a clearance (3.255 m), for ships as for circles.
    A circle's size is its radius, so its column is the hull distance `dist`
    already holds; a ship's size is its beam, the extent of its body-frame
    vertices across its heading (the reference's ship of width w spans
    -w/2 .. w/2 there).  Pinned by tests/golden/track_golden.npz and
    ship_golden.npz.  Absent slots keep +inf."""
    out = np.array(dist, dtype=np.float64, copy=True)
    C = circles.shape[1]
    NV = poly_verts.shape[2]
    px, py = state[0], state[1]
    for q in range(poly_pose.shape[1]):
        nv = poly_nv[:, q]
        hi = np.full(nv.shape, -np.inf)
        lo = np.full(nv.shape, np.inf)
        for k in range(NV):
            y = poly_verts[:, q, k, 1]
            act = k < nv
            hi = np.where(act & (y > hi), y, hi)
            lo = np.where(act & (y < lo), y, lo)
        mx = poly_pose[:, q, 0] - px
        my = poly_pose[:, q, 1] - py
        with np.errstate(invalid="ignore"):
            d = (np.sqrt(mx * mx + my * my) - (hi - lo)) - clearance
        out[:, C + q] = np.where(nv > 0, d, np.inf)
    return out


def reference_metadata(state, circles, poly_pose, meta):
    """The heading_ts, speed and course columns of `meta` [V, O, 5] in the
    reference's conventions (the bearing is the same in both; the dist column is
    `reference_distances`, set by the caller):
    heading_ts = direction of the obstacle's velocity (0 at rest), speed = the
    relative speed (the vessel's own at rest), course = world-frame direction of
    the relative velocity (0 / -pi at rest)."""
    vox, voy = _own_velocity(state)
    psi = state[2]
    out = meta.copy()
    for o in range(circles.shape[1] + poly_pose.shape[1]):
        _, _, ovx, ovy = _obstacle(circles, poly_pose, o)
        still, ss, wx, wy, _, rr = reference_motion(ovx, ovy, vox, voy, psi)
        out[:, o, 0] = np.where(still, 0.0, atan2(ovy, ovx))
        out[:, o, 2] = np.where(still, ss, np.sqrt(rr))
        out[:, o, 4] = np.where(still, np.where(psi > 0.0, 0.0, 0.0 - PI), atan2(wy, wx))
    return out


def risk_reference(state, circles, poly_pose, dist):
    """Per-obstacle risk [V, C+P] of a simulator state with the index above;
    inputs formed as in `reference_motion`; dist = `reference_distances`.  The
    obstacle position p is that distance along the direction of the obstacle's centre m, so
    p . v = dist (m . w)/(|m||w|) and likewise p x v; the aspect is
    -(m . o)|m . o| / (|m|^2 |o|^2).  Every present obstacle has a risk, whatever
    its distance (the reference logs risks of up to 0.41 for obstacles beyond its
    sensor range); absent slots (dist = +inf) have 0."""
    V = state.shape[1]
    px, py, psi = state[0], state[1], state[2]
    vox, voy = _own_velocity(state)
    O = circles.shape[1] + poly_pose.shape[1]
    risk = np.zeros((V, O), dtype=np.float64)
    for o in range(O):
        cx, cy, ovx, ovy = _obstacle(circles, poly_pose, o)
        still, ss, wx, wy, ww, rr = reference_motion(ovx, ovy, vox, voy, psi)
        mx = cx - px
        my = cy - py
        d = dist[:, o]
        mm = mx * mx + my * my
        den = np.sqrt(mm * ww)
        dens = np.where(den > 0.0, den, 1.0)
        moving = np.where(still, ss > 0.0, rr > 0.0)
        mo = mx * ovx + my * ovy
        ad = mm * (ovx * ovx + ovy * ovy)
        aspect = np.where(ad > 0.0, 0.0 - (mo * np.abs(mo)) / np.where(ad > 0.0, ad, 1.0), 0.0)
        b = wrap_angle(atan2(my, mx) - psi)
        with np.errstate(invalid="ignore"):   # absent slots have d = +inf; they are gated to 0
            along = d * ((mx * wx + my * wy) / dens)
            cross = d * ((mx * wy - my * wx) / dens)
            r = cri_terms(along, cross, moving, d, b, aspect)
        risk[:, o] = np.where(d < np.inf, r, 0.0)
    return risk
